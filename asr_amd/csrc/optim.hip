// Optimizer step of the training loop on flat fp32 arenas: global gradient norm, clipping and
// AdamW in two passes over HBM, without a host round trip in between.
//
// Reference arithmetic: DistributedTrainer.train_epoch, src/training/distributed.py:207-231 --
//   total_norm = sqrt(sum_p ||p.grad||_2^2)               (:210-215, one .item() per parameter there)
//   NaN norm -> the step is skipped                        (:218-221)
//   clip_grad_norm_(parameters, max_norm)                  (src/training/utils.py:653-665):
//     grads *= min(1, max_norm / (total_norm + 1e-6))
//   torch.optim.AdamW(lr, betas, eps = 1e-8, weight_decay) (src/training/distributed.py:92-98):
//     p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//     p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Parameters, gradients and both moments are flat arrays in the same order (asr_amd.optim flattens
// a model's parameters once; gradients come from the bucket arena of asr_amd.parallel), so a step is
// three launches whatever the number of parameter tensors.  All three kernels are HBM-bound streams:
// 4 n bytes read for the norm, 16 n read + 12 n written for the update.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

constexpr int kOptThreads = 256;
constexpr int kOptMaxBlocks = 1024;  // partial sums of the norm; fixed for a given n: the sum's order is too

__device__ __forceinline__ float block_sum(float s, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = s;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < kOptThreads / 64; ++w) t += red[w];
  return t;  // valid in thread 0
}

__global__ __launch_bounds__(kOptThreads) void k_grad_sumsq(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[kOptThreads / 64];
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float s = 0.f;
  for (long i = blockIdx.x * (long)kOptThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kOptThreads) {
    const float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
    const float v = g[(n4 << 2) + threadIdx.x];
    s += v * v;
  }
  const float t = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// norm2[0] = sum of the partials in double, in a fixed tree order: thread t adds partials t, t + 256, ...,
// then the xor tree of each wave, then the four wave sums in order
__global__ __launch_bounds__(kOptThreads) void k_grad_sumsq_final(const float* __restrict__ partial, int nb,
                                                                  float* __restrict__ norm2) {
  __shared__ double red[kOptThreads / 64];
  double t = 0.0;
  for (int i = threadIdx.x; i < nb; i += kOptThreads) t += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < kOptThreads / 64; ++w) tot += red[w];
    norm2[0] = (float)tot;
  }
}

struct AdamW {
  float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, max_norm;
};

__device__ __forceinline__ void adamw1(float& p, float g, float& m, float& v, const AdamW& h, float coef) {
  g *= coef;
  p *= 1.f - h.lr * h.wd;
  m = h.beta1 * m + (1.f - h.beta1) * g;
  v = h.beta2 * v + (1.f - h.beta2) * g * g;
  const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
  p -= (h.lr / h.bc1) * (m / denom);
}

__global__ __launch_bounds__(kOptThreads) void k_adamw(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long n, AdamW h,
                                                       const float* __restrict__ norm2) {
  float coef = 1.f;
  if (norm2) {
    const float norm = sqrtf(norm2[0]);
    if (norm != norm) return;  // NaN gradients: the step is skipped (uniform over the grid)
    coef = fminf(1.f, h.max_norm / (norm + 1e-6f));
  }
  const long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long i = blockIdx.x * (long)kOptThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kOptThreads) {
    float4 pp = p4[i], mm = m4[i], vv = v4[i];
    const float4 gg = g4[i];
    adamw1(pp.x, gg.x, mm.x, vv.x, h, coef);
    adamw1(pp.y, gg.y, mm.y, vv.y, h, coef);
    adamw1(pp.z, gg.z, mm.z, vv.z, h, coef);
    adamw1(pp.w, gg.w, mm.w, vv.w, h, coef);
    p4[i] = pp;
    m4[i] = mm;
    v4[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    adamw1(p[i], g[i], m[i], v[i], h, coef);
  }
}

// ---- the same step with its clock in device memory -----------------------------------------------------------------
// A step replayed from a HIP graph cannot take the learning rate and the step count as launch arguments: they
// are frozen at capture.  Here both live in a small device array the kernels advance themselves:
//   clock[0]  scheduler steps taken        clock[1]  optimizer steps taken (a NaN norm does not count, as in the
//   clock[2..5]  this step's {lr, 1 - b1^t, sqrt(1 - b2^t), clip coefficient or NaN = skip}, floats     reference)
// k_opt_clock (one thread, double arithmetic like the host's) runs between the norm and the update.
// The learning rate of optimizer step k is what the scheduler set after step k - 1 (the loop calls
// optimizer.step() and then lr_scheduler.step(), src/training/distributed.py:229-234; the scheduler's formula:
// src/training/utils.py:246-263): the constructor's rate for the first step, then
//   s <= warmup: max_lr * s / warmup        else  max_lr * decay_rate ** ((s - warmup) // decay_steps),  s = k - 1.
struct LrSchedule {
  double lr0, max_lr, decay_rate;
  int warmup_steps, decay_steps, scheduled;
};

__global__ void k_opt_clock(int* __restrict__ clock, LrSchedule sc, float beta1, float beta2, float max_norm,
                            const float* __restrict__ norm2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float* f = reinterpret_cast<float*>(clock);
  const int s = clock[0];
  double lr = sc.lr0;
  if (sc.scheduled && s > 0) {
    if (s <= sc.warmup_steps)
      lr = sc.max_lr * ((double)s / (double)sc.warmup_steps);
    else
      lr = sc.max_lr * pow(sc.decay_rate, (double)((s - sc.warmup_steps) / max(sc.decay_steps, 1)));
  }
  clock[0] = s + 1;
  float coef = 1.f;
  if (norm2) {
    const float norm = sqrtf(norm2[0]);
    coef = norm != norm ? norm : fminf(1.f, max_norm / (norm + 1e-6f));
  }
  int t = clock[1];
  if (coef == coef) clock[1] = ++t;
  f[2] = (float)lr;
  f[3] = (float)(1.0 - pow((double)beta1, (double)t));
  f[4] = (float)sqrt(1.0 - pow((double)beta2, (double)t));
  f[5] = coef;
}

__global__ __launch_bounds__(kOptThreads) void k_adamw_clock(float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v, long n,
                                                             AdamW h, const int* __restrict__ clock) {
  const float* f = reinterpret_cast<const float*>(clock);
  const float coef = f[5];
  if (coef != coef) return;  // NaN gradients: the step is skipped (uniform over the grid)
  h.lr = f[2];
  h.bc1 = f[3];
  h.bc2_sqrt = f[4];
  const long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long i = blockIdx.x * (long)kOptThreads + threadIdx.x; i < n4; i += (long)gridDim.x * kOptThreads) {
    float4 pp = p4[i], mm = m4[i], vv = v4[i];
    const float4 gg = g4[i];
    adamw1(pp.x, gg.x, mm.x, vv.x, h, coef);
    adamw1(pp.y, gg.y, mm.y, vv.y, h, coef);
    adamw1(pp.z, gg.z, mm.z, vv.z, h, coef);
    adamw1(pp.w, gg.w, mm.w, vv.w, h, coef);
    p4[i] = pp;
    m4[i] = mm;
    v4[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (n4 << 2) + threadIdx.x;
    adamw1(p[i], g[i], m[i], v[i], h, coef);
  }
}

static inline bool al16p(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
static inline int opt_blocks(long n) {
  const long b = ((n >> 2) + kOptThreads - 1) / kOptThreads;
  return (int)(b < 1 ? 1 : (b > kOptMaxBlocks ? kOptMaxBlocks : b));
}

}  // namespace lmufft

using namespace lmufft;

extern "C" {

size_t lmufft_grad_norm_workspace_bytes(void) { return sizeof(float) * kOptMaxBlocks; }

int lmufft_grad_norm2_f32(const float* grads, int64_t n, float* norm2, void* workspace, size_t workspace_bytes,
                          void* stream) {
  if (n <= 0) return set_error("grad norm: n = %lld", (long long)n);
  if (!grads || !norm2 || !workspace) return set_error("grad norm: NULL pointer");
  if (workspace_bytes < lmufft_grad_norm_workspace_bytes())
    return set_error("grad norm: workspace of %zu bytes needed, %zu given", lmufft_grad_norm_workspace_bytes(), workspace_bytes);
  if (!al16p(grads)) return set_error("grad norm: the gradient arena must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nb = opt_blocks(n);
  float* partial = (float*)workspace;
  {
    ProfScope ps("k_grad_sumsq", st);
    hipLaunchKernelGGL(k_grad_sumsq, dim3(nb), dim3(kOptThreads), 0, st, grads, (long)n, partial);
    hipLaunchKernelGGL(k_grad_sumsq_final, dim3(1), dim3(kOptThreads), 0, st, partial, nb, norm2);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_grad_sumsq failed: %s", hipGetErrorString(e));
  return 0;
}

int lmufft_adamw_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int32_t step, const float* norm2,
                          float max_norm, void* stream) {
  if (n <= 0) return set_error("adamw: n = %lld", (long long)n);
  if (!params || !grads || !exp_avg || !exp_avg_sq) return set_error("adamw: NULL pointer");
  if (step < 1) return set_error("adamw: step counts from 1, got %d", step);
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) return set_error("adamw: betas must lie in [0, 1)");
  if (norm2 && !(max_norm > 0.f)) return set_error("adamw: max_norm must be positive when a norm is given");
  if (!al16p(params) || !al16p(grads) || !al16p(exp_avg) || !al16p(exp_avg_sq))
    return set_error("adamw: the arenas must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  AdamW h;
  h.lr = lr;
  h.beta1 = beta1;
  h.beta2 = beta2;
  h.eps = eps;
  h.wd = weight_decay;
  h.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  h.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  h.max_norm = max_norm;
  {
    ProfScope ps("k_adamw", st);
    hipLaunchKernelGGL(k_adamw, dim3(opt_blocks(n)), dim3(kOptThreads), 0, st, params, grads, exp_avg, exp_avg_sq, (long)n, h,
                       norm2);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_adamw failed: %s", hipGetErrorString(e));
  return 0;
}

int lmufft_adamw_step_clock_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                int32_t* clock, const lmufft_lr_schedule* schedule, float beta1, float beta2, float eps,
                                float weight_decay, const float* norm2, float max_norm, void* stream) {
  if (n <= 0) return set_error("adamw: n = %lld", (long long)n);
  if (!params || !grads || !exp_avg || !exp_avg_sq || !clock || !schedule) return set_error("adamw: NULL pointer");
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) return set_error("adamw: betas must lie in [0, 1)");
  if (norm2 && !(max_norm > 0.f)) return set_error("adamw: max_norm must be positive when a norm is given");
  if (schedule->scheduled && (schedule->warmup_steps < 0 || schedule->decay_steps <= 0))
    return set_error("adamw: schedule needs warmup_steps >= 0 and decay_steps > 0");
  if (!al16p(params) || !al16p(grads) || !al16p(exp_avg) || !al16p(exp_avg_sq))
    return set_error("adamw: the arenas must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  LrSchedule sc{schedule->lr0,          schedule->max_lr,      schedule->decay_rate,
                schedule->warmup_steps, schedule->decay_steps, schedule->scheduled};
  AdamW h;
  h.lr = 0.f;
  h.beta1 = beta1;
  h.beta2 = beta2;
  h.eps = eps;
  h.wd = weight_decay;
  h.bc1 = h.bc2_sqrt = 1.f;
  h.max_norm = max_norm;
  {
    ProfScope ps("k_adamw", st);
    hipLaunchKernelGGL(k_opt_clock, dim3(1), dim3(64), 0, st, clock, sc, beta1, beta2, max_norm, norm2);
    hipLaunchKernelGGL(k_adamw_clock, dim3(opt_blocks(n)), dim3(kOptThreads), 0, st, params, grads, exp_avg, exp_avg_sq,
                       (long)n, h, (const int*)clock);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_adamw_clock failed: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"

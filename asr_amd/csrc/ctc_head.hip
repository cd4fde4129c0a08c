// CTC head for inference on gfx950: vocabulary projection + log-softmax + greedy decode,
// without the per-frame host round trips of the reference.
//
// Reference arithmetic: CTCDecoder.forward, src/models/asr_model.py:34-50
//   (log_probs = log_softmax(enc W^T + b)), and LMUASRModel.decode / _greedy_decode,
//   src/models/asr_model.py:158-213 (per utterance: argmax over the vocabulary for the first
//   input_lengths[i] frames, drop blanks (= vocab_size - 1) and consecutive repeats) -- there a
//   Python loop with one .item() per frame.
//
// k_ctc_logits: one wave per frame; lane v owns vocabulary entry v (V <= 64).  W^T sits in LDS as
//   [k][V + 1] (lanes hit distinct banks), the frame's encoder row in a wave-private LDS strip
//   (broadcast reads).  Max, sum-exp and the arg-max (first maximum, like torch.argmax on the
//   CPU) are xor-tree reductions over the wave.  HBM: reads enc once, writes log-probs once.
// k_ctc_collapse: one workgroup per utterance; keep-flags, block prefix sum, compaction.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>

#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

typedef __bf16 ctc_bf16;
constexpr int kCtcThreads = 256;
constexpr int kCtcMaxHd = 1024;

template <class TIn>
__global__ __launch_bounds__(kCtcThreads) void k_ctc_logits(const TIn* __restrict__ enc, const float* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ log_probs, int* __restrict__ best,
                                                            long frames, int Hd, int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ldw = V + 1;
  float* Wl = lds;                                  // [Hd][V + 1]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* er = lds + (size_t)Hd * ldw + (size_t)wave * Hd;  // this wave's encoder row
  for (int i = threadIdx.x; i < V * Hd; i += kCtcThreads) {
    const int v = i / Hd, k = i % Hd;
    Wl[k * ldw + v] = W[i];
  }
  __syncthreads();
  const float bv = lane < V ? bias[lane] : 0.f;
  for (long f = (long)blockIdx.x * 4 + wave; f < frames; f += (long)gridDim.x * 4) {
    for (int k = lane; k < Hd; k += 64) er[k] = (float)enc[f * Hd + k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private strip: program order suffices
    float acc = bv;
    if (lane < V) {
#pragma unroll 8
      for (int k = 0; k < Hd; ++k) acc = fmaf(er[k], Wl[k * ldw + lane], acc);
    }
    // log-softmax over the V lanes
    float mx = lane < V ? acc : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float ex = lane < V ? expf(acc - mx) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ex += __shfl_xor(ex, o);
    const float lp = acc - mx - logf(ex);
    // arg-max of the log-probabilities, first maximum on ties
    float bvv = lane < V ? lp : -INFINITY;
    int bi = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bvv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bvv || (ov == bvv && oi < bi)) {
        bvv = ov;
        bi = oi;
      }
    }
    if (log_probs && lane < V) log_probs[f * V + lane] = lp;
    if (lane == 0) best[f] = bi;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the strip is rewritten in the next round
  }
}

// tokens[b][0 .. counts[b]) = collapsed arg-max path of utterance b; the rest of the row is -1
__global__ __launch_bounds__(kCtcThreads) void k_ctc_collapse(const int* __restrict__ best,
                                                              const int* __restrict__ lengths, int T, int blank,
                                                              int* __restrict__ tokens, int* __restrict__ counts) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const int* bp = best + (long)b * T;
  int* out = tokens + (long)b * T;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += kCtcThreads) {
    const int t = t0 + tid;
    const int tok = t < len ? bp[t] : blank;
    const int prev = (t > 0 && t < len) ? bp[t - 1] : -1;
    const int keep = (t < len && tok != blank && tok != prev) ? 1 : 0;
    // inclusive scan inside the wave, then across the four waves
    int s = keep;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(s, o);
      if (lane >= o) s += v;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int before = carry_s;
    for (int w2 = 0; w2 < wave; ++w2) before += wsum[w2];
    if (keep) out[before + s - 1] = tok;
    __syncthreads();
    if (tid == 0) carry_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  const int total = carry_s;
  for (int t = total + tid; t < T; t += kCtcThreads) out[t] = -1;
  if (tid == 0) counts[b] = total;
}

}  // namespace lmufft

using namespace lmufft;

extern "C" {

size_t lmufft_ctc_workspace_bytes(int32_t B, int32_t T) {
  if (B <= 0 || T <= 0) return 0;
  return (sizeof(int32_t) * (size_t)B * T + 255) & ~size_t(255);
}

int lmufft_ctc_greedy(const void* enc, int dtype, const float* W, const float* bias, const int32_t* lengths,
                      int32_t B, int32_t T, int32_t Hd, int32_t V, int32_t blank, float* log_probs, int32_t* tokens,
                      int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
  if (B <= 0 || T <= 0 || Hd <= 0 || V <= 0) return set_error("ctc: bad shape (B=%d T=%d Hd=%d V=%d)", B, T, Hd, V);
  if (V > 64) return set_error("ctc: vocabulary of %d entries; the kernel holds one entry per lane (<= 64)", V);
  if (Hd > kCtcMaxHd) return set_error("ctc: encoder width %d exceeds %d", Hd, kCtcMaxHd);
  if (blank < 0 || blank >= V) return set_error("ctc: blank index %d outside the vocabulary", blank);
  if (dtype != LMUFFT_F32 && dtype != LMUFFT_BF16) return set_error("ctc: unknown dtype %d", dtype);
  if (!enc || !W || !bias || !tokens || !counts) return set_error("ctc: NULL pointer");
  const size_t need = lmufft_ctc_workspace_bytes(B, T);
  if (!workspace || workspace_bytes < need)
    return set_error("ctc: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  int* best = (int*)workspace;
  const long frames = (long)B * T;
  const size_t lds = sizeof(float) * ((size_t)Hd * (V + 1) + 4 * (size_t)Hd);
  if (lds > 160 * 1024) return set_error("ctc: %zu bytes of LDS needed", lds);
  const void* fn = dtype == LMUFFT_F32 ? (const void*)k_ctc_logits<float> : (const void*)k_ctc_logits<ctc_bf16>;
  if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return set_error("cannot reserve %zu bytes of LDS for k_ctc_logits", lds);
  const int grid = (int)std::min<long>((frames + 3) / 4, 1024);
  {
    ProfScope ps("k_ctc_logits", st);
    if (dtype == LMUFFT_F32)
      hipLaunchKernelGGL(k_ctc_logits<float>, dim3(grid), dim3(kCtcThreads), lds, st, (const float*)enc, W, bias,
                         log_probs, best, frames, Hd, V);
    else
      hipLaunchKernelGGL(k_ctc_logits<ctc_bf16>, dim3(grid), dim3(kCtcThreads), lds, st, (const ctc_bf16*)enc, W, bias,
                         log_probs, best, frames, Hd, V);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_ctc_logits failed: %s", hipGetErrorString(e));
  {
    ProfScope ps("k_ctc_collapse", st);
    hipLaunchKernelGGL(k_ctc_collapse, dim3(B), dim3(kCtcThreads), 0, st, (const int*)best, lengths, T, blank, tokens,
                       counts);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_ctc_collapse failed: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"

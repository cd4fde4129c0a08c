// Residual add + LayerNorm on gfx950: one pass over HBM instead of the reference's add kernel
// followed by a LayerNorm kernel; and its backward for training (k_layernorm_bwd, further down).
//
// Reference arithmetic: MultiHeadAttention.forward, src/models/lmu_encoder.py:97
//   (output = layer_norm(output + residual)) and LMUEncoder.forward,
//   src/models/lmu_encoder.py:274 (x = layer_norm(x)); torch.nn.LayerNorm: biased variance
//   over the last dimension, eps inside the square root, fp32 statistics.
//
// k_add_layernorm: 32 lanes own one row (two rows per wave); a lane keeps its share of the row
//   (up to 8 groups of 4 elements, C <= 1024) in registers, so the row is read once; mean and the
//   centred second moment are two xor-tree reductions over the 32 lanes.  HBM-bound: reads x
//   (+ res) once, writes out once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

typedef float ln_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kLnThreads = 256;
constexpr int kLnMaxGroups = 8;  // groups of 4 elements per lane: C <= 32 * 4 * 8

template <bool BF16>
__device__ __forceinline__ ln_f32x4 ln_load4(const void* base, long idx) {
  if (BF16) {
    const uint2 p = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    ln_f32x4 v;
    v[0] = __uint_as_float(p.x << 16);
    v[1] = __uint_as_float(p.x & 0xffff0000u);
    v[2] = __uint_as_float(p.y << 16);
    v[3] = __uint_as_float(p.y & 0xffff0000u);
    return v;
  } else {
    return *reinterpret_cast<const ln_f32x4*>(reinterpret_cast<const float*>(base) + idx);
  }
}

__device__ __forceinline__ unsigned ln_bf16_bits(float f) {  // round to nearest even
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN stays NaN
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <bool BF16>
__device__ __forceinline__ void ln_store4(void* base, long idx, const ln_f32x4& v) {
  if (BF16) {
    uint2 p;
    p.x = ln_bf16_bits(v[0]) | (ln_bf16_bits(v[1]) << 16);
    p.y = ln_bf16_bits(v[2]) | (ln_bf16_bits(v[3]) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + idx) = p;
  } else {
    *reinterpret_cast<ln_f32x4*>(reinterpret_cast<float*>(base) + idx) = v;
  }
}

// Dropout after the layer norm (`self.dropout_layer(layer_norm(x))`, src/models/lmu_encoder.py:274-277, training):
// element (row, column) is kept by the shared hash (HashDropout: rows of C columns), scaled by 1 / (1 - p).
__device__ __forceinline__ ln_f32x4 ln_dropout4(const ln_f32x4& v, const HashDropout& dr, const HashRow& hr, int col) {
  const uint32_t x0 = hash_pair(hr, (unsigned)col >> 1), x1 = hash_pair(hr, ((unsigned)col >> 1) + 1);
  ln_f32x4 o;
  o[0] = hash_keep_lo(dr, x0) ? v[0] * dr.inv_keep : 0.f;
  o[1] = hash_keep_hi(dr, x0) ? v[1] * dr.inv_keep : 0.f;
  o[2] = hash_keep_lo(dr, x1) ? v[2] * dr.inv_keep : 0.f;
  o[3] = hash_keep_hi(dr, x1) ? v[3] * dr.inv_keep : 0.f;
  return o;
}

// RES: 0 none, 1 f32, 2 bf16
template <bool XB, int RES, bool OB, int NG, bool DROP = false>
__global__ __launch_bounds__(kLnThreads) void k_add_layernorm(const void* x, const void* res,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, long rows,
                                                              int C, void* out, HashDropout dr = HashDropout{0, 0, 0, 1.f},
                                                              void* twin = nullptr) {
  hash_resolve_epoch(dr);
  const int sub = threadIdx.x & 31;                                  // lane inside the row's half-wave
  const long row0 = (long)blockIdx.x * (kLnThreads / 32) + (threadIdx.x >> 5);
  const long row_step = (long)gridDim.x * (kLnThreads / 32);
  const int groups = C / 4;
  ln_f32x4 g[NG], bt[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int c = sub + 32 * j;
    if (c < groups) {
      g[j] = *reinterpret_cast<const ln_f32x4*>(gamma + 4 * c);
      bt[j] = *reinterpret_cast<const ln_f32x4*>(beta + 4 * c);
    }
  }
  const float inv_c = 1.f / (float)C;
  for (long row = row0; row < rows; row += row_step) {
    ln_f32x4 v[NG];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int c = sub + 32 * j;
      if (c < groups) {
        v[j] = ln_load4<XB>(x, row * C + 4 * c);
        if (RES == 1) v[j] += ln_load4<false>(res, row * C + 4 * c);
        if (RES == 2) v[j] += ln_load4<true>(res, row * C + 4 * c);
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int c = sub + 32 * j;
      if (c < groups) {
        v[j] -= mean;
        q += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * inv_c + eps);
    [[maybe_unused]] const HashRow hr = hash_row(dr, (unsigned long long)row, (unsigned)C >> 1);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int c = sub + 32 * j;
      if (c < groups) {
        ln_f32x4 y = v[j] * rstd * g[j] + bt[j];
        if (DROP) y = ln_dropout4(y, dr, hr, 4 * c);
        ln_store4<OB>(out, row * C + 4 * c, y);
        if (!OB && twin) ln_store4<true>(twin, row * C + 4 * c, y);  // the same values rounded to bf16
      }
    }
  }
}

template <bool XB, int RES, bool OB>
static void launch_ln(int ng, dim3 grid, hipStream_t st, const void* x, const void* res, const float* gamma,
                      const float* beta, float eps, long rows, int C, void* out, const HashDropout& dr, void* twin) {
#define LN_CASE(NGV)                                                                                                  \
  do {                                                                                                                \
    if (dr.threshold)                                                                                                 \
      hipLaunchKernelGGL((k_add_layernorm<XB, RES, OB, NGV, true>), grid, dim3(kLnThreads), 0, st, x, res, gamma,     \
                         beta, eps, rows, C, out, dr, twin);                                                          \
    else                                                                                                              \
      hipLaunchKernelGGL((k_add_layernorm<XB, RES, OB, NGV, false>), grid, dim3(kLnThreads), 0, st, x, res, gamma,    \
                         beta, eps, rows, C, out, dr, twin);                                                          \
  } while (0)
  if (ng <= 1) LN_CASE(1);
  else if (ng <= 2) LN_CASE(2);
  else if (ng <= 3) LN_CASE(3);
  else if (ng <= 4) LN_CASE(4);
  else if (ng <= 6) LN_CASE(6);
  else LN_CASE(8);
#undef LN_CASE
}

// ---------------------------------------------------------------------------
// Backward of y = LayerNorm(x + res) * gamma + beta (what autograd derives for src/models/lmu_encoder.py:97, 274):
//   xhat = (x + res - mean) * rstd;  gy = dy * gamma;
//   ds = rstd * (gy - mean_c(gy) - xhat * mean_c(gy * xhat))     (gradient of the sum: of x and of res alike)
//   dgamma = sum_rows dy * xhat;  dbeta = sum_rows dy.
// Row layout as in the forward (32 lanes own a row held in registers); mean and rstd are recomputed from
// x (+ res) with the forward's arithmetic, so nothing but its inputs has to be kept.  A lane accumulates
// dgamma / dbeta of its columns over the rows it visits; the eight row slots of a workgroup are summed through
// LDS and every workgroup leaves one partial [2][C] in the workspace, which k_layernorm_param_reduce adds in
// workgroup order: deterministic.  dtypes are run-time flags (the kernel is HBM-bound).
// ---------------------------------------------------------------------------
__device__ __forceinline__ ln_f32x4 ln_load4_rt(const void* base, long idx, bool bf16) {
  return bf16 ? ln_load4<true>(base, idx) : ln_load4<false>(base, idx);
}
__device__ __forceinline__ void ln_store4_rt(void* base, long idx, const ln_f32x4& v, bool bf16) {
  if (bf16) ln_store4<true>(base, idx, v);
  else ln_store4<false>(base, idx, v);
}

constexpr int kLnBwdMaxBlocks = 1024;  // four workgroups per CU (what its registers allow): the row loop is a chain of load, reductions, store

template <int NG>
__global__ __launch_bounds__(kLnThreads) void k_layernorm_bwd(const void* x, int xb, const void* res, int rb, const void* dy,
                                                              int gb, const float* __restrict__ gamma, float eps, long rows,
                                                              int C, void* dx, void* dres, float* __restrict__ partial,
                                                              HashDropout dr, const void* dy2, int g2b, int relu_x) {
  // relu_x: x is the output of a ReLU whose backward is the next thing to happen to dx -- dx is stored already
  // multiplied by (x > 0) (dres is not)
  hash_resolve_epoch(dr);
  extern __shared__ __attribute__((aligned(16))) float red[];  // [8 row slots][2][C]
  const int sub = threadIdx.x & 31, slot = threadIdx.x >> 5;
  const long row0 = (long)blockIdx.x * (kLnThreads / 32) + slot;
  const long row_step = (long)gridDim.x * (kLnThreads / 32);
  const int groups = C / 4;
  ln_f32x4 g[NG], dg[NG], db[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int c = sub + 32 * j;
    if (c < groups) g[j] = *reinterpret_cast<const ln_f32x4*>(gamma + 4 * c);
    dg[j] = ln_f32x4(0.f);
    db[j] = ln_f32x4(0.f);
  }
  const float inv_c = 1.f / (float)C;
  for (long row = row0; row < rows; row += row_step) {
    ln_f32x4 v[NG], gy[NG];
    float s = 0.f;
    unsigned keep = relu_x ? 0u : ~0u;  // bit 4 j + e: x[row][4 (sub + 32 j) + e] > 0
    const HashRow hr = hash_row(dr, (unsigned long long)row, (unsigned)C >> 1);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int c = sub + 32 * j;
      if (c < groups) {
        v[j] = ln_load4_rt(x, row * C + 4 * c, xb);
        if (relu_x) {
#pragma unroll
          for (int e = 0; e < 4; ++e) keep |= (v[j][e] > 0.f ? 1u : 0u) << (4 * j + e);
        }
        if (res) v[j] += ln_load4_rt(res, row * C + 4 * c, rb);
        gy[j] = ln_load4_rt(dy, row * C + 4 * c, gb);
        if (dy2) gy[j] += ln_load4_rt(dy2, row * C + 4 * c, g2b);  // the gradient of the output's second copy
        if (dr.threshold) gy[j] = ln_dropout4(gy[j], dr, hr, 4 * c);  // through the dropout
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int c = sub + 32 * j;
      if (c < groups) {
        v[j] -= mean;
        q += (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * inv_c + eps);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int c = sub + 32 * j;
      if (c < groups) {
        v[j] *= rstd;  // xhat
        db[j] += gy[j];
        dg[j] += gy[j] * v[j];
        gy[j] *= g[j];
        a += (gy[j][0] + gy[j][1]) + (gy[j][2] + gy[j][3]);
        const ln_f32x4 t = gy[j] * v[j];
        b += (t[0] + t[1]) + (t[2] + t[3]);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
    }
    a *= inv_c;
    b *= inv_c;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int c = sub + 32 * j;
      if (c < groups) {
        const ln_f32x4 d = (gy[j] - a - v[j] * b) * rstd;
        ln_f32x4 dm = d;
#pragma unroll
        for (int e = 0; e < 4; ++e) dm[e] = ((keep >> (4 * j + e)) & 1u) ? d[e] : 0.f;
        ln_store4_rt(dx, row * C + 4 * c, dm, xb);
        if (dres) ln_store4_rt(dres, row * C + 4 * c, d, rb);
      }
    }
  }
  // the workgroup's partial sums of dgamma, dbeta
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int c = sub + 32 * j;
    if (c < groups) {
      *reinterpret_cast<ln_f32x4*>(&red[(slot * 2 + 0) * C + 4 * c]) = dg[j];
      *reinterpret_cast<ln_f32x4*>(&red[(slot * 2 + 1) * C + 4 * c]) = db[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += kLnThreads) {
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < kLnThreads / 32; ++sl) t += red[sl * 2 * C + i];
    partial[(long)blockIdx.x * 2 * C + i] = t;
  }
}

// out[i] = sum over the workgroups' partials; i < C: dgamma, else dbeta.  One wave per column: lane l adds the
// partials l, l + 64, ... (all loads in flight together), then the xor tree -- a fixed order.
__global__ __launch_bounds__(kLnThreads) void k_layernorm_param_reduce(const float* __restrict__ partial, int nblocks, int C,
                                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int i = blockIdx.x * (kLnThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= 2 * C) return;
  float t = 0.f;
  for (int b = lane; b < nblocks; b += 64) t += partial[(long)b * 2 * C + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  if (lane == 0) {
    if (i < C) dgamma[i] = t;
    else dbeta[i - C] = t;
  }
}

// the same sums for several layer norms in one launch (the table travels in the kernel's arguments)
constexpr int kLnSumManyMax = 32;
struct LnSumBatch {
  const float* partial[kLnSumManyMax];
  float* dgamma[kLnSumManyMax];
  float* dbeta[kLnSumManyMax];
  int nblocks[kLnSumManyMax], C[kLnSumManyMax];
  int first_block[kLnSumManyMax + 1];
  int count;
};
__global__ __launch_bounds__(kLnThreads) void k_layernorm_param_reduce_many(LnSumBatch) {
  typedef const __attribute__((address_space(4))) LnSumBatch* table_ptr;
  table_ptr b = (table_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  int it = 0;
  for (int i = 1; i < b->count; ++i) it += (int)blockIdx.x >= b->first_block[i] ? 1 : 0;  // uniform
  const int C = b->C[it], nblocks = b->nblocks[it];
  const float* partial = b->partial[it];
  const int i = ((int)blockIdx.x - b->first_block[it]) * (kLnThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= 2 * C) return;
  float t = 0.f;
  for (int k = lane; k < nblocks; k += 64) t += partial[(long)k * 2 * C + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  if (lane == 0) {
    if (i < C) b->dgamma[it][i] = t;
    else b->dbeta[it][i - C] = t;
  }
}

}  // namespace lmufft

using namespace lmufft;

static int ln_bwd_blocks(int64_t rows) {
  const long blocks = (rows + kLnThreads / 32 - 1) / (kLnThreads / 32);
  return (int)std::min<long>(blocks, kLnBwdMaxBlocks);
}

static int add_layernorm_any(const void* x, int x_dtype, const void* res, int res_dtype, const float* gamma,
                            const float* beta, float eps, int64_t rows, int32_t C, void* out, int out_dtype,
                            const HashDropout& dr, void* twin, void* stream) {
  if (rows < 0 || C <= 0) return set_error("layernorm: bad shape (rows=%lld C=%d)", (long long)rows, C);
  if (C % 4 != 0 || C > 32 * 4 * kLnMaxGroups)
    return set_error("layernorm: width %d (needs a multiple of 4, at most %d)", C, 32 * 4 * kLnMaxGroups);
  auto known = [](int d) { return d == LMUFFT_F32 || d == LMUFFT_BF16; };
  if (!known(x_dtype) || !known(out_dtype) || (res && !known(res_dtype))) return set_error("layernorm: unknown dtype");
  if (rows == 0) return 0;
  if (!x || !gamma || !beta || !out) return set_error("layernorm: NULL pointer");
  if (twin && (out_dtype != LMUFFT_F32 || (reinterpret_cast<uintptr_t>(twin) & 7) != 0 || twin == x || twin == res))
    return set_error("layernorm: the bf16 twin needs a float32 `out`, 8-byte alignment and memory of its own");
  auto aligned = [](const void* p, int dtype) {
    return (reinterpret_cast<uintptr_t>(p) & (dtype == LMUFFT_F32 ? 15 : 7)) == 0;
  };
  if (!aligned(x, x_dtype) || !aligned(out, out_dtype) || (res && !aligned(res, res_dtype)) ||
      !aligned(gamma, LMUFFT_F32) || !aligned(beta, LMUFFT_F32))
    return set_error("layernorm: rows must start on 16-byte (f32) / 8-byte (bf16) boundaries");
  // in place is fine when the element sizes agree (a lane overwrites exactly what it has read)
  if ((out == x && out_dtype != x_dtype) || (res && out == res && out_dtype != res_dtype))
    return set_error("layernorm: out may alias x or res only with the same dtype");
  hipStream_t st = (hipStream_t)stream;
  const int ng = (C / 4 + 31) / 32;
  const long blocks = (rows + kLnThreads / 32 - 1) / (kLnThreads / 32);
  const dim3 grid((unsigned)std::min<long>(blocks, 256 * 16));
  const bool xb = x_dtype == LMUFFT_BF16, ob = out_dtype == LMUFFT_BF16;
  const int rk = !res ? 0 : (res_dtype == LMUFFT_BF16 ? 2 : 1);
  {
    ProfScope ps("k_add_layernorm", st);
#define LN_GO(XBV, RV, OBV) launch_ln<XBV, RV, OBV>(ng, grid, st, x, res, gamma, beta, eps, (long)rows, C, out, dr, twin)
#define LN_RES(XBV, OBV)          \
  do {                            \
    if (rk == 0) LN_GO(XBV, 0, OBV); \
    else if (rk == 1) LN_GO(XBV, 1, OBV); \
    else LN_GO(XBV, 2, OBV);      \
  } while (0)
    if (xb && ob) LN_RES(true, true);
    else if (xb) LN_RES(true, false);
    else if (ob) LN_RES(false, true);
    else LN_RES(false, false);
#undef LN_RES
#undef LN_GO
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_add_layernorm failed: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int lmufft_add_layernorm(const void* x, int x_dtype, const void* res, int res_dtype, const float* gamma,
                                    const float* beta, float eps, int64_t rows, int32_t C, void* out, int out_dtype,
                                    void* stream) {
  return add_layernorm_any(x, x_dtype, res, res_dtype, gamma, beta, eps, rows, C, out, out_dtype,
                           make_hash_dropout(0.f, 0, 0), nullptr, stream);
}

// The registered dropout epoch counter of every device (lmufft_dropout_epoch).
namespace lmufft {
static const uint32_t* g_dropout_epoch[64] = {nullptr};
const uint32_t* dropout_epoch_ptr() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  return g_dropout_epoch[dev];
}
}  // namespace lmufft

extern "C" int lmufft_dropout_epoch(const uint32_t* epoch_dev) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return lmufft::set_error("dropout epoch: no current device");
  if (dev < 0 || dev >= 64) return lmufft::set_error("dropout epoch: device index %d out of range", dev);
  if (epoch_dev && (reinterpret_cast<uintptr_t>(epoch_dev) & 3)) return lmufft::set_error("dropout epoch: pointer must be 4-byte aligned");
  lmufft::g_dropout_epoch[dev] = epoch_dev;
  return 0;
}

extern "C" int lmufft_add_layernorm_dropout(const void* x, int x_dtype, const void* res, int res_dtype, const float* gamma,
                                            const float* beta, float eps, int64_t rows, int32_t C, void* out,
                                            int out_dtype, float p_drop, uint32_t seed0, uint32_t seed1, void* out_bf16,
                                            void* stream) {
  if (!(p_drop >= 0.f && p_drop < 1.f)) return set_error("layernorm: dropout probability %g outside [0, 1)", (double)p_drop);
  if (out == x || (res && out == res)) {
    if (p_drop > 0.f) return set_error("layernorm with dropout: the backward recomputes from x and res; out may not alias them");
  }
  return add_layernorm_any(x, x_dtype, res, res_dtype, gamma, beta, eps, rows, C, out, out_dtype,
                           make_hash_dropout(p_drop, seed0, seed1), out_bf16, stream);
}

extern "C" size_t lmufft_layernorm_bwd_workspace_bytes(int32_t C) {
  return C > 0 ? sizeof(float) * 2 * (size_t)C * kLnBwdMaxBlocks : 0;
}

extern "C" int lmufft_layernorm_bwd(const void* x, int x_dtype, const void* res, int res_dtype, const void* dy, int dy_dtype,
                                    const float* gamma, float eps, int64_t rows, int32_t C, void* dx, void* dres,
                                    float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
  return lmufft_layernorm_dropout_bwd(x, x_dtype, res, res_dtype, dy, dy_dtype, nullptr, 0, gamma, eps, rows, C, 0.f, 0, 0,
                                      dx, dres, dgamma, dbeta, workspace, workspace_bytes, stream);
}

extern "C" int lmufft_layernorm_dropout_bwd(const void* x, int x_dtype, const void* res, int res_dtype, const void* dy,
                                            int dy_dtype, const void* dy2, int dy2_dtype, const float* gamma, float eps,
                                            int64_t rows, int32_t C,
                                            float p_drop, uint32_t seed0, uint32_t seed1, void* dx, void* dres,
                                            float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  return lmufft_layernorm_dropout_bwd_relu(x, x_dtype, res, res_dtype, dy, dy_dtype, dy2, dy2_dtype, gamma, eps, rows, C,
                                           p_drop, seed0, seed1, dx, dres, dgamma, dbeta, workspace, workspace_bytes,
                                           stream, 0);
}

extern "C" int lmufft_layernorm_dropout_bwd_relu(const void* x, int x_dtype, const void* res, int res_dtype,
                                                 const void* dy, int dy_dtype, const void* dy2, int dy2_dtype,
                                                 const float* gamma, float eps, int64_t rows, int32_t C, float p_drop,
                                                 uint32_t seed0, uint32_t seed1, void* dx, void* dres, float* dgamma,
                                                 float* dbeta, void* workspace, size_t workspace_bytes, void* stream,
                                                 int relu_x) {
  if (relu_x && res && !dres)
    return set_error("layernorm backward: relu_x with a residual needs a dres buffer (dx alone is masked)");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return set_error("layernorm backward: dropout probability %g outside [0, 1)", (double)p_drop);
  const HashDropout dr = make_hash_dropout(p_drop, seed0, seed1);
  if (rows <= 0 || C <= 0) return set_error("layernorm backward: bad shape (rows=%lld C=%d)", (long long)rows, C);
  if (C % 4 != 0 || C > 32 * 4 * kLnMaxGroups)
    return set_error("layernorm backward: width %d (needs a multiple of 4, at most %d)", C, 32 * 4 * kLnMaxGroups);
  auto known = [](int d) { return d == LMUFFT_F32 || d == LMUFFT_BF16; };
  if (!known(x_dtype) || !known(dy_dtype) || (res && !known(res_dtype)) || (dy2 && !known(dy2_dtype)))
    return set_error("layernorm backward: unknown dtype");
  if (!x || !dy || !gamma || !dx || !workspace || (!dgamma != !dbeta)) return set_error("layernorm backward: NULL pointer");
  if (dres && !res) return set_error("layernorm backward: dres without res");
  if (workspace_bytes < lmufft_layernorm_bwd_workspace_bytes(C))
    return set_error("layernorm backward: workspace of %zu bytes needed, %zu given", lmufft_layernorm_bwd_workspace_bytes(C),
                     workspace_bytes);
  auto aligned = [](const void* p, int dtype) {
    return (reinterpret_cast<uintptr_t>(p) & (dtype == LMUFFT_F32 ? 15 : 7)) == 0;
  };
  if (!aligned(x, x_dtype) || !aligned(dx, x_dtype) || !aligned(dy, dy_dtype) || (res && !aligned(res, res_dtype)) ||
      (dres && !aligned(dres, res_dtype)) || (dy2 && !aligned(dy2, dy2_dtype)) || !aligned(gamma, LMUFFT_F32) ||
      !aligned(workspace, LMUFFT_F32))
    return set_error("layernorm backward: rows must start on 16-byte (f32) / 8-byte (bf16) boundaries");
  hipStream_t st = (hipStream_t)stream;
  const int ng = (C / 4 + 31) / 32;
  const int nb = ln_bwd_blocks(rows);
  const size_t lds = sizeof(float) * (kLnThreads / 32) * 2 * (size_t)C;
  float* partial = (float*)workspace;
  {
    ProfScope ps("k_layernorm_bwd", st);
#define LN_BWD(NGV)                                                                                                    \
  hipLaunchKernelGGL((k_layernorm_bwd<NGV>), dim3(nb), dim3(kLnThreads), lds, st, x, x_dtype == LMUFFT_BF16, res,         \
                     res_dtype == LMUFFT_BF16, dy, dy_dtype == LMUFFT_BF16, gamma, eps, (long)rows, C, dx, dres, partial, dr, \
                     dy2, dy2_dtype == LMUFFT_BF16, relu_x ? 1 : 0)
    if (ng <= 1) LN_BWD(1);
    else if (ng <= 2) LN_BWD(2);
    else if (ng <= 3) LN_BWD(3);
    else if (ng <= 4) LN_BWD(4);
    else if (ng <= 6) LN_BWD(6);
    else LN_BWD(8);
#undef LN_BWD
    if (dgamma)  // (NULL: the workgroups' partial sums stay in the workspace for lmufft_layernorm_param_sum_many)
      hipLaunchKernelGGL(k_layernorm_param_reduce, dim3((2 * C + kLnThreads / 64 - 1) / (kLnThreads / 64)), dim3(kLnThreads), 0, st,
                         (const float*)partial, nb, C, dgamma, dbeta);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_layernorm_bwd failed: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int lmufft_layernorm_param_sum_many(const lmufft_ln_sum_item* items, int32_t count, void* stream) {
  if (count < 0 || (count > 0 && !items)) return set_error("layernorm parameter sums: bad item list");
  for (int i = 0; i < count; ++i) {
    const lmufft_ln_sum_item& it = items[i];
    if (it.rows <= 0 || it.C <= 0 || it.C % 4 != 0 || it.C > 32 * 4 * kLnMaxGroups)
      return set_error("layernorm parameter sums: bad shape (rows=%lld C=%d)", (long long)it.rows, it.C);
    if (!it.dgamma || !it.dbeta || !it.workspace) return set_error("layernorm parameter sums: NULL pointer");
  }
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < count; i0 += kLnSumManyMax) {
    LnSumBatch b;
    b.count = std::min(count - i0, kLnSumManyMax);
    b.first_block[0] = 0;
    for (int i = 0; i < b.count; ++i) {
      const lmufft_ln_sum_item& it = items[i0 + i];
      b.partial[i] = (const float*)it.workspace;
      b.dgamma[i] = it.dgamma;
      b.dbeta[i] = it.dbeta;
      b.nblocks[i] = ln_bwd_blocks(it.rows);
      b.C[i] = it.C;
      b.first_block[i + 1] = b.first_block[i] + (2 * it.C + kLnThreads / 64 - 1) / (kLnThreads / 64);
    }
    {
      ProfScope ps("k_layernorm_bwd", st);
      hipLaunchKernelGGL(k_layernorm_param_reduce_many, dim3(b.first_block[b.count]), dim3(kLnThreads), 0, st, b);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error("launch of k_layernorm_param_reduce_many failed: %s", hipGetErrorString(e));
  }
  return 0;
}

// Residual add + LayerNorm for inference on gfx950: one pass over HBM instead of the
// reference's add kernel followed by a LayerNorm kernel.
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

// RES: 0 none, 1 f32, 2 bf16
template <bool XB, int RES, bool OB, int NG>
__global__ __launch_bounds__(kLnThreads) void k_add_layernorm(const void* x, const void* res,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, long rows,
                                                              int C, void* out) {
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
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int c = sub + 32 * j;
      if (c < groups) ln_store4<OB>(out, row * C + 4 * c, v[j] * rstd * g[j] + bt[j]);
    }
  }
}

template <bool XB, int RES, bool OB>
static void launch_ln(int ng, dim3 grid, hipStream_t st, const void* x, const void* res, const float* gamma,
                      const float* beta, float eps, long rows, int C, void* out) {
#define LN_CASE(NGV)                                                                                              \
  hipLaunchKernelGGL((k_add_layernorm<XB, RES, OB, NGV>), grid, dim3(kLnThreads), 0, st, x, res, gamma, beta, eps, \
                     rows, C, out)
  if (ng <= 1) LN_CASE(1);
  else if (ng <= 2) LN_CASE(2);
  else if (ng <= 3) LN_CASE(3);
  else if (ng <= 4) LN_CASE(4);
  else if (ng <= 6) LN_CASE(6);
  else LN_CASE(8);
#undef LN_CASE
}

}  // namespace lmufft

using namespace lmufft;

extern "C" int lmufft_add_layernorm(const void* x, int x_dtype, const void* res, int res_dtype, const float* gamma,
                                    const float* beta, float eps, int64_t rows, int32_t C, void* out, int out_dtype,
                                    void* stream) {
  if (rows < 0 || C <= 0) return set_error("layernorm: bad shape (rows=%lld C=%d)", (long long)rows, C);
  if (C % 4 != 0 || C > 32 * 4 * kLnMaxGroups)
    return set_error("layernorm: width %d (needs a multiple of 4, at most %d)", C, 32 * 4 * kLnMaxGroups);
  auto known = [](int d) { return d == LMUFFT_F32 || d == LMUFFT_BF16; };
  if (!known(x_dtype) || !known(out_dtype) || (res && !known(res_dtype))) return set_error("layernorm: unknown dtype");
  if (rows == 0) return 0;
  if (!x || !gamma || !beta || !out) return set_error("layernorm: NULL pointer");
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
#define LN_GO(XBV, RV, OBV) launch_ln<XBV, RV, OBV>(ng, grid, st, x, res, gamma, beta, eps, (long)rows, C, out)
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

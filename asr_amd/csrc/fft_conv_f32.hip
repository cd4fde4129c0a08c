// Causal convolution of the LMUFFT layer by LDS-resident FFT (fp32), gfx950.
//
// Reference arithmetic: src/lmu.py:303-312 -- rfft(u, 2T) * rfft(H, 2T) -> irfft,
// first T samples.  Here the transform length is the power of two N >= 2T - 1
// (any such N gives the same linear convolution) and TWO memory channels share
// one complex transform: with u and H real,
//     ifft( fft(u) * (fft(H_2p) + i fft(H_2p+1)) ) = m_2p + i m_2p+1 ,
// so a batch item costs one forward FFT of u and M/2 inverse complex FFTs.
// The adjoint uses the mirrored filters: with z = gm_2p + i gm_2p+1 and
// F = rev(H_2p) - i rev(H_2p+1), Re(z * F)[T-1+s] = sum_k corr(gm_k, H_k)[s],
// and because the transform is linear the channel pairs are summed in the
// spectrum: M/2 forward FFTs and ONE inverse FFT per batch item and k group.
//
// One wave owns one 1024-point (or smaller) transform: 64 lanes x N/64 points,
// Stockham stages of radix 16, 16, N/256 with the butterflies in registers and
// the exchanges through ONE wave-private LDS buffer (a stage reads all its
// inputs into registers before it writes), index padded by one element per 16
// (conflict-free ds_read_b64 / ds_write_b64).  A wave's LDS operations complete
// in program order, so the stages of a transform need no workgroup barrier and
// the eight waves of a workgroup run their transforms independently.
// The first stage reads straight from its source (global u, the LDS copy of a
// gm tile, or the spectrum product) and the last stage writes straight to its
// sink (the [T][k] output tile, or the spectrum accumulator), so a transform
// makes two LDS round trips.  Output tiles are transposed through LDS so that
// m is written, and gm read, in 128-byte row segments.
//
// Kernels: k_fft_u (spectrum of every u[b], B small transforms) -> k_conv_fwd_fft;
// k_conv_bwd_fft (partial spectra per 32-channel group) -> k_fft_gu (sum of the
// groups, one inverse transform per batch item, ReLU mask of the encoder).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

constexpr int kFftWaves = 8;                     // one transform per wave at a time
constexpr int kFftThreads = 64 * kFftWaves;
constexpr int kFftPairs = 16;      // channel pairs per workgroup (32 memory channels, 128-byte rows)
constexpr int kFftCols = 2 * kFftPairs;
constexpr int kTileLd = kFftCols + 1;  // LDS row stride of the [T][32] tile in floats

__device__ __forceinline__ float2 operator+(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 operator-(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {  // a * conj(b)
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
template <bool INV>
__device__ __forceinline__ float2 twiddle(float2 a, float2 w) {  // a * w (forward) or a * conj(w) (inverse)
  return INV ? cmul_conj(a, w) : cmul(a, w);
}
template <bool INV>
__device__ __forceinline__ float2 mul_mi(float2 a) {  // a * (-i) forward, a * (+i) inverse
  return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// ---- small DFTs in registers, natural order in and out ---------------------
template <bool INV>
__device__ __forceinline__ void dft2(float2 (&v)[2]) {
  const float2 a = v[0];
  v[0] = a + v[1];
  v[1] = a - v[1];
}
template <bool INV>
__device__ __forceinline__ void dft4(float2& a, float2& b, float2& c, float2& d) {
  const float2 t0 = a + c, t1 = a - c, t2 = b + d, t3 = mul_mi<INV>(b - d);
  a = t0 + t2;
  b = t1 + t3;
  c = t0 - t2;
  d = t1 - t3;
}
template <bool INV>
__device__ __forceinline__ void dft4(float2 (&v)[4]) {
  dft4<INV>(v[0], v[1], v[2], v[3]);
}
// n = 2*n1 + n2, k = k1 + 4*k2:  W8^(nk) = W4^(n1 k1) W8^(n2 k1) W2^(n2 k2)
template <bool INV>
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
  const float h = 0.70710678118654752f;
  dft4<INV>(v[0], v[2], v[4], v[6]);  // n2 = 0 -> Y0[k1] in v[0], v[2], v[4], v[6]
  dft4<INV>(v[1], v[3], v[5], v[7]);  // n2 = 1 -> Y1[k1] in v[1], v[3], v[5], v[7]
  v[3] = twiddle<INV>(v[3], make_float2(h, -h));   // W8^1
  v[5] = mul_mi<INV>(v[5]);                        // W8^2 = -i
  v[7] = twiddle<INV>(v[7], make_float2(-h, -h));  // W8^3
  float2 o[8];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    o[k1] = v[2 * k1] + v[2 * k1 + 1];
    o[k1 + 4] = v[2 * k1] - v[2 * k1 + 1];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = o[i];
}
// n = 4*n1 + n2, k = k1 + 4*k2:  W16^(nk) = W4^(n1 k1) W16^(n2 k1) W4^(n2 k2)
template <bool INV>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
  const float c = 0.92387953251128674f, s = 0.38268343236508977f, h = 0.70710678118654752f;
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) dft4<INV>(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);  // Y[n2][k1] in v[4*k1 + n2]
  // W16^(n2*k1), element v[4*k1 + n2]
  v[4 * 1 + 1] = twiddle<INV>(v[4 * 1 + 1], make_float2(c, -s));    // W^1
  v[4 * 1 + 2] = twiddle<INV>(v[4 * 1 + 2], make_float2(h, -h));    // W^2
  v[4 * 1 + 3] = twiddle<INV>(v[4 * 1 + 3], make_float2(s, -c));    // W^3
  v[4 * 2 + 1] = twiddle<INV>(v[4 * 2 + 1], make_float2(h, -h));    // W^2
  v[4 * 2 + 2] = mul_mi<INV>(v[4 * 2 + 2]);                         // W^4 = -i
  v[4 * 2 + 3] = twiddle<INV>(v[4 * 2 + 3], make_float2(-h, -h));   // W^6
  v[4 * 3 + 1] = twiddle<INV>(v[4 * 3 + 1], make_float2(s, -c));    // W^3
  v[4 * 3 + 2] = twiddle<INV>(v[4 * 3 + 2], make_float2(-h, -h));   // W^6
  v[4 * 3 + 3] = twiddle<INV>(v[4 * 3 + 3], make_float2(-c, s));    // W^9
  float2 o[16];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    float2 a = v[4 * k1], b = v[4 * k1 + 1], cc = v[4 * k1 + 2], d = v[4 * k1 + 3];
    dft4<INV>(a, b, cc, d);  // over n2 -> k2
    o[k1] = a;
    o[k1 + 4] = b;
    o[k1 + 8] = cc;
    o[k1 + 12] = d;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = o[i];
}
template <int R, bool INV>
__device__ __forceinline__ void dftR(float2 (&v)[R]) {
  if constexpr (R == 2) dft2<INV>(v);
  if constexpr (R == 4) dft4<INV>(v);
  if constexpr (R == 8) dft8<INV>(v);
  if constexpr (R == 16) dft16<INV>(v);
}

__device__ __forceinline__ int fpad(int i) { return i + (i >> 4); }

// Orders this wave's LDS traffic: everything issued so far has completed, and the compiler
// moves no memory access across this point.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// One Stockham stage of an N-point transform done by ONE wave.
//   src(i, slot) -> element i of the stage input  (i in [0, N)); slot = q*R + r counts the lane's
//                  inputs in a fixed order (a compile-time constant after unrolling)
//   dst(i, v, slot) -> element i of the stage output; slot = q*R + r counts the lane's outputs
//                  in a fixed order (the same for every transform of that stage)
//   tw[k] = exp(-2 pi i k / N), k in [0, N)
// Butterfly j (of N/R) reads i = j + r*N/R, multiplies by W^(r * (j % NS) * N/(NS*R)),
// takes the R-point DFT and writes i = (j / NS)*NS*R + (j % NS) + r*NS.
// All inputs are read before the first output is written (src and dst may be one buffer).
template <int N, int R, int NS, bool INV, class Src, class Dst>
__device__ __forceinline__ void fft_stage(const float2* __restrict__ tw, Src&& src, Dst&& dst) {
  constexpr int NB = N / R;
  constexpr int NQ = (NB + 63) / 64;
  const int lane = threadIdx.x & 63;
  float2 v[NQ][R];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = lane + 64 * q;
    if (NB % 64 != 0 && j >= NB) break;
#pragma unroll
    for (int r = 0; r < R; ++r) v[q][r] = src(j + r * NB, q * R + r);
  }
  wave_lds_fence();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = lane + 64 * q;
    if (NB % 64 != 0 && j >= NB) break;
    const int kk = j & (NS - 1);
    if (NS > 1) {
      const int tstep = kk * (N / (NS * R));
#pragma unroll
      for (int r = 1; r < R; ++r) v[q][r] = twiddle<INV>(v[q][r], tw[(r * tstep) & (N - 1)]);
    }
    dftR<R, INV>(v[q]);
    const int j0 = (j / NS) * (NS * R) + kk;
#pragma unroll
    for (int r = 0; r < R; ++r) dst(j0 + r * NS, v[q][r], q * R + r);
  }
  wave_lds_fence();
}

// Whole transform by one wave: radix 16, 16, then N/256 (absent for N = 256).  buf is a
// wave-private LDS buffer of fpad(N) float2.  No workgroup barrier inside.
template <int N, bool INV, class Src, class Dst>
__device__ __forceinline__ void wave_fft(const float2* __restrict__ tw, float2* buf, Src&& src, Dst&& dst) {
  constexpr int R3 = N / 256;
  static_assert(R3 == 1 || R3 == 2 || R3 == 4, "N in {256, 512, 1024}");
  fft_stage<N, 16, 1, INV>(tw, src, [&](int i, float2 v, int) { buf[fpad(i)] = v; });
#ifdef LMUFFT_FFT_SHUFFLE
  // Measured alternative (off by default, see the note at the end of this function): the exchange between the
  // second radix-16 stage and the radix-4 stage through v_permlane32_swap / v_permlane16_swap instead of LDS.
  if constexpr (R3 == 4) {
    float2 keep[16];
    fft_stage<N, 16, 16, INV>(tw, [&](int i, int) { return buf[fpad(i)]; }, [&](int, float2 v, int slot) { keep[slot] = v; });
    // Lane (row b = lane / 16, c = lane % 16) holds elements 256 b + c + 16 r in register r.  The radix-4 butterfly
    // at position p = c + 16 r takes the elements p + 256 r' from the rows r' = 0..3, register r: the four lanes
    // c, 16 + c, 32 + c, 48 + c transpose register group (r / 4) against row, after which lane (b, c) owns the
    // butterflies p = c + 16 (4 b + q), q = 0..3.
    const int lane = threadIdx.x & 63, brow = lane >> 4, c = lane & 15;
    float2 X[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int comp = 0; comp < 2; ++comp) {
        unsigned g[4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) g[gi] = __float_as_uint(comp ? keep[4 * gi + q].y : keep[4 * gi + q].x);
        auto s02 = __builtin_amdgcn_permlane32_swap(g[0], g[2], false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(g[1], g[3], false, false);
        auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        const unsigned x[4] = {t01[0], t01[1], t23[0], t23[1]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (comp) X[q][r].y = __uint_as_float(x[r]);
          else X[q][r].x = __uint_as_float(x[r]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = c + 16 * (4 * brow + q);
#pragma unroll
      for (int r = 1; r < 4; ++r) X[q][r] = twiddle<INV>(X[q][r], tw[(r * j) & (N - 1)]);
      dft4<INV>(X[q]);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst(j + r * 256, X[q][r], q * 4 + r);
    }
    wave_lds_fence();
    return;
  }
#endif
  if constexpr (R3 == 1) {
    fft_stage<N, 16, 16, INV>(tw, [&](int i, int) { return buf[fpad(i)]; }, dst);
  } else {
    fft_stage<N, 16, 16, INV>(tw, [&](int i, int) { return buf[fpad(i)]; },
                              [&](int i, float2 v, int) { buf[fpad(i)] = v; });
    fft_stage<N, R3, 256, INV>(tw, [&](int i, int) { return buf[fpad(i)]; }, dst);
  }
}

template <int N>
__device__ __forceinline__ void fill_twiddles(float2* tw) {
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    float sn, cs;
    sincospif(2.0f * (float)k / (float)N, &sn, &cs);
    tw[k] = make_float2(cs, -sn);
  }
}
// the same table, copied from the prepared buffer (cheaper than N sincos per workgroup)
template <int N>
__device__ __forceinline__ void load_twiddles(float2* tw, const float2* __restrict__ table) {
  for (int k = threadIdx.x; k < N; k += blockDim.x) tw[k] = table[k];
}

// LDS carve (floats): tw[2N] | per wave one buffer [waves * 2*fpad(N)] | tile [T * kTileLd]
template <int N>
struct FftLds {
  static constexpr int PADN = N + N / 16;
  static constexpr int TW = 0;
  static constexpr int WORK = TW + 2 * N;
  static constexpr int TILE = WORK + kFftWaves * 2 * PADN;
  static size_t bytes(int T, int pairs = kFftPairs) { return sizeof(float) * ((size_t)TILE + (size_t)T * (2 * pairs + 1)); }
  static size_t bytes_single() { return sizeof(float) * (size_t)(2 * N + 4 * PADN); }  // one-wave kernels
};

// ---------------------------------------------------------------------------
// spectra of the filters, once per (H, T):
//   G [p][f] = fft(H_2p + i H_2p+1)[f] / N                  (forward conv)
//   Gr[p][f] = fft(rev(H_2p) - i rev(H_2p+1))[f] / N        (adjoint), rev(h)[j] = h[T-1-j]
// one workgroup (one wave) per pair and kind
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void k_fft_prep(const float* __restrict__ H, float2* __restrict__ G, int M, int T,
                                                 int ldH, int npairs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* tw = reinterpret_cast<float2*>(lds);
  float2* buf = tw + N;
  const int p = blockIdx.x, adj = blockIdx.y;
  fill_twiddles<N>(tw);
  __syncthreads();
  if (p == 0 && adj == 0)  // the twiddle table follows the spectra
    for (int k = threadIdx.x; k < N; k += 64) G[(long)2 * npairs * N + k] = tw[k];
  const float* h0 = H + (long)(2 * p) * ldH;
  const float* h1 = (2 * p + 1 < M) ? H + (long)(2 * p + 1) * ldH : nullptr;
  float2* out = G + ((long)adj * npairs + p) * N;
  const float scale = 1.0f / (float)N;
  wave_fft<N, false>(
      tw, buf,
      [&](int i, int) {
        if (i >= T) return make_float2(0.f, 0.f);
        const int j = adj ? T - 1 - i : i;
        const float a = h0[j], b = h1 ? h1[j] : 0.f;
        return make_float2(a, adj ? -b : b);
      },
      [&](int i, float2 v, int) { out[i] = make_float2(v.x * scale, v.y * scale); });
}

// spectrum of u[b] (zero-padded to N), one wave per batch item: U[b][f]
template <int N>
__global__ __launch_bounds__(64) void k_fft_u(const float* __restrict__ u, const float2* __restrict__ table,
                                              float2* __restrict__ U, int T) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* tw = reinterpret_cast<float2*>(lds);
  float2* buf = tw + N;
  const int b = blockIdx.x;
  load_twiddles<N>(tw, table);
  __syncthreads();
  const float* ub = u + (long)b * T;
  float2* Ub = U + (long)b * N;
  wave_fft<N, false>(
      tw, buf,
      [&](int i, int) {  // clamped address + select: the sixteen loads of a lane stay in flight together
        const float v = ub[min(i, T - 1)];
        return make_float2(i < T ? v : 0.f, 0.f);
      },
      [&](int i, float2 v, int) { Ub[i] = v; });
}

// ---------------------------------------------------------------------------
// forward: m[b][t][k0 .. k0+31] for one batch item and 16 channel pairs
// ---------------------------------------------------------------------------
// PAIRS channel pairs per workgroup (16: 128-byte rows of m; 24: three transforms per wave, which at batch 64 and
// 96 pairs is one workgroup per CU with every SIMD at six transforms -- 16 gave 1.5 rounds of workgroups, i.e. eight
// transforms on the busiest SIMDs).  FUSE_U: the workgroup's first wave takes the spectrum of u[b] itself (the
// workgroups of a batch item each do, writing the same numbers to the same place) instead of a launch of 64
// one-wave workgroups before this one.
template <int N, int PAIRS, bool FUSE_U>
__global__ __launch_bounds__(kFftThreads) void k_conv_fwd_fft(const float* __restrict__ u, float2* __restrict__ U,
                                                              const float2* __restrict__ G,
                                                              const float2* __restrict__ table,
                                                              float* __restrict__ m, int T, int M, int npairs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using L = FftLds<N>;
  constexpr int COLS = 2 * PAIRS, LD = COLS + 1;
  float2* tw = reinterpret_cast<float2*>(lds + L::TW);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2* buf = reinterpret_cast<float2*>(lds + L::WORK) + (size_t)wave * L::PADN;
  float* tile = lds + L::TILE;
  const int b = blockIdx.y, p0 = blockIdx.x * PAIRS;
  float2* Ub = U + (long)b * N;

  load_twiddles<N>(tw, table);
  __syncthreads();
  if (FUSE_U) {
    if (wave == 0) {
      const float* ub = u + (long)b * T;
      wave_fft<N, false>(
          tw, buf,
          [&](int i, int) {  // clamped address + select: the sixteen loads of a lane stay in flight together
            const float v = ub[min(i, T - 1)];
            return make_float2(i < T ? v : 0.f, 0.f);
          },
          [&](int i, float2 v, int) { Ub[i] = v; });
      __threadfence_block();
    }
    __syncthreads();
  }
  // ifft(U * G_p) -> columns 2*pl, 2*pl+1 of the tile; the waves take the pairs in turn
  for (int pl = wave; pl < PAIRS; pl += kFftWaves) {
    const int p = p0 + pl;
    if (p >= npairs) break;
    const float2* Gp = G + (long)p * N;
    // the product spectrum, formed with coalesced loads and parked in the wave's buffer, which
    // the transform then works on in place
#pragma unroll
    for (int k = 0; k < N / 64; ++k) buf[fpad(lane + 64 * k)] = cmul(Ub[lane + 64 * k], Gp[lane + 64 * k]);
    wave_lds_fence();
    wave_fft<N, true>(
        tw, buf, [&](int i, int) { return buf[fpad(i)]; },
        [&](int i, float2 v, int) {
          if (i < T) {
            tile[i * LD + 2 * pl] = v.x;
            tile[i * LD + 2 * pl + 1] = v.y;
          }
        });
  }
  __syncthreads();
  // tile -> m[b][t][k0 + c], whole row segments
  const int k0 = 2 * p0;
  float* mb = m + (long)b * T * M;
  for (int idx = threadIdx.x; idx < T * COLS; idx += kFftThreads) {
    const int t = idx / COLS, c = idx % COLS;
    if (k0 + c < M) mb[(long)t * M + k0 + c] = tile[t * LD + c];
  }
}

// ---------------------------------------------------------------------------
// adjoint, first kernel: spectrum of the contribution of the 32 channels k0 .. k0+31 of one
// batch item, Sp[(kgroup * B + b)][f] = sum_pairs fft(gm_2p + i gm_2p+1)[f] * Gr_p[f]
// ---------------------------------------------------------------------------
template <int N, int PAIRS>
__global__ __launch_bounds__(kFftThreads) void k_conv_bwd_fft(const float* __restrict__ gm,
                                                              const float2* __restrict__ Gr,
                                                              const float2* __restrict__ table,
                                                              float2* __restrict__ Sp, int B, int T, int M,
                                                              int npairs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using L = FftLds<N>;
  float2* tw = reinterpret_cast<float2*>(lds + L::TW);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2* buf = reinterpret_cast<float2*>(lds + L::WORK) + (size_t)wave * L::PADN;
  float* tile = lds + L::TILE;
  constexpr int COLS = 2 * PAIRS, LD = COLS + 1;
  const int b = blockIdx.y, p0 = blockIdx.x * PAIRS, k0 = 2 * p0;
  const float* gmb = gm + (long)b * T * M;

  load_twiddles<N>(tw, table);
  for (int idx = threadIdx.x; idx < T * COLS; idx += kFftThreads) {
    const int t = idx / COLS, c = idx % COLS;
    tile[t * LD + c] = (k0 + c < M) ? gmb[(long)t * M + k0 + c] : 0.f;
  }
  __syncthreads();

  // a lane accumulates the spectrum elements lane, lane + 64, ...: every transform parks its
  // result in the wave's buffer and the product with the mirrored-filter spectrum is formed from
  // there with coalesced loads
  constexpr int PER = N / 64;
  float2 acc[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) acc[i] = make_float2(0.f, 0.f);

  for (int pl = wave; pl < PAIRS; pl += kFftWaves) {
    const int p = p0 + pl;
    if (p >= npairs) break;
    const float2* Gp = Gr + (long)p * N;
    wave_fft<N, false>(
        tw, buf,
        [&](int i, int) {  // clamped address + select keeps the reads of a lane in flight together
          const float* t2 = tile + min(i, T - 1) * LD + 2 * pl;
          const float re = t2[0], im = t2[1];
          return make_float2(i < T ? re : 0.f, i < T ? im : 0.f);
        },
        [&](int i, float2 v, int) { buf[fpad(i)] = v; });
    wave_lds_fence();
    float2 gq[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) gq[q] = Gp[lane + 64 * q];
#pragma unroll
    for (int q = 0; q < PER; ++q) acc[q] = acc[q] + cmul(buf[fpad(lane + 64 * q)], gq[q]);
    wave_lds_fence();  // the buffer is the next transform's workspace
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) buf[fpad(lane + 64 * q)] = acc[q];
  __syncthreads();
  // the waves' spectra summed in a fixed order -> Sp, coalesced
  const float2* w0 = reinterpret_cast<const float2*>(lds + L::WORK);
  float2* out = Sp + ((long)blockIdx.x * B + b) * N;
  for (int i = threadIdx.x; i < N; i += kFftThreads) {
    const int pi = fpad(i);
    float2 s = w0[pi];
#pragma unroll
    for (int wv = 1; wv < kFftWaves; ++wv) s = s + w0[(size_t)wv * L::PADN + pi];
    out[i] = s;
  }
}

// adjoint, second kernel: gu[b][s] = Re ifft(sum_groups Sp[g][b])[T-1+s], times (u[b][s] > 0) when
// u is given (the encoder's ReLU); one wave per batch item
template <int N>
__global__ __launch_bounds__(64) void k_fft_gu(const float2* __restrict__ Sp, const float2* __restrict__ table,
                                               const float* __restrict__ u, float* __restrict__ gu, int B, int T,
                                               int groups) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* tw = reinterpret_cast<float2*>(lds);
  float2* buf = tw + N;
  float2* sum = buf + FftLds<N>::PADN;
  const int b = blockIdx.x, lane = threadIdx.x;
  load_twiddles<N>(tw, table);
  // the groups' spectra summed in group order, coalesced, all loads of a group in flight together
  float2 s[N / 64];
#pragma unroll
  for (int q = 0; q < N / 64; ++q) s[q] = Sp[(long)b * N + lane + 64 * q];
  for (int g = 1; g < groups; ++g) {
    float2 t[N / 64];
#pragma unroll
    for (int q = 0; q < N / 64; ++q) t[q] = Sp[((long)g * B + b) * N + lane + 64 * q];
#pragma unroll
    for (int q = 0; q < N / 64; ++q) s[q] = s[q] + t[q];
  }
#pragma unroll
  for (int q = 0; q < N / 64; ++q) sum[fpad(lane + 64 * q)] = s[q];
  __syncthreads();
  const float* ub = u ? u + (long)b * T : nullptr;
  float* out = gu + (long)b * T;
  wave_fft<N, true>(
      tw, buf, [&](int i, int) { return sum[fpad(i)]; },
      [&](int i, float2 v, int) {
        if (i >= T - 1 && i < 2 * T - 1) {
          const int sidx = i - (T - 1);
          out[sidx] = (ub && !(ub[sidx] > 0.f)) ? 0.f : v.x;
        }
      });
}

// ---------------------------------------------------------------------------
// Long sequences (2T - 1 > 1024): uniformly partitioned convolution on the same one-wave
// transforms.  N points, L = N / 2 samples per block; the filter is cut into partitions
// h_q = H[:, qL .. qL + L), the input into overlapping segments x_j = u[(j-1)L .. (j+1)L):
//     m[jL .. jL + L) = last L samples of ifft( sum_{q <= j} fft(x_j-q) * fft(h_q) ) ,
// and for the adjoint, with segments z_c = gm[cL .. cL + N) of a channel pair and the partitions
// mirrored inside themselves, F_q = rev_L(h_2p,q) - i rev_L(h_2p+1,q):
//     gu[jL .. jL + L) = Re ifft( sum_{c >= j} fft(z_c) * fft(F_c-j) )[L - 1 .. 2L - 1) .
// The products are summed in the spectrum, so the number of inverse transforms stays one per
// (batch item, channel pair, block); the spectra of u blocks, filter partitions and the partial
// sums are L2-resident (a few MB).
// ---------------------------------------------------------------------------
// The forward takes 1024-point transforms (blocks of 512 samples: half as many partition products
// as with 512 points, and the products' operands come from L2 for every batch item and channel
// pair); the adjoint keeps 512 points, because a workgroup stages a whole N-frame segment of 32
// channels in LDS.  The two directions therefore have their own partitions of the filter.
constexpr int kPartNF = 1024;            // forward
constexpr int kPartNA = 512;             // adjoint
constexpr int kPartJB = 6;  // output blocks per adjoint workgroup (accumulators in registers)

// spectra of the filter partitions of one direction: G[q * npairs + p][f]; `table` receives exp(-2 pi i k / N)
template <int N>
__global__ __launch_bounds__(64) void k_fft_prep_part(const float* __restrict__ H, float2* __restrict__ G,
                                                      float2* __restrict__ table, int M, int T, int ldH, int npairs,
                                                      int adj) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* tw = reinterpret_cast<float2*>(lds);
  float2* buf = tw + N;
  constexpr int L = N / 2;
  const int p = blockIdx.x, q = blockIdx.y;
  fill_twiddles<N>(tw);
  __syncthreads();
  if (p == 0 && q == 0)
    for (int k = threadIdx.x; k < N; k += 64) table[k] = tw[k];
  const float* h0 = H + (long)(2 * p) * ldH;
  const float* h1 = (2 * p + 1 < M) ? H + (long)(2 * p + 1) * ldH : nullptr;
  float2* out = G + ((long)q * npairs + p) * N;
  const float scale = 1.0f / (float)N;
  wave_fft<N, false>(
      tw, buf,
      [&](int i, int) {
        const int tap = q * L + (adj ? L - 1 - i : i);
        if (i >= L || tap >= T) return make_float2(0.f, 0.f);
        const float a = h0[tap], b = h1 ? h1[tap] : 0.f;
        return make_float2(a, adj ? -b : b);
      },
      [&](int i, float2 v, int) { out[i] = make_float2(v.x * scale, v.y * scale); });
}

// spectra of the overlapping input segments: U[b * nb + j][f] = fft(u[b][(j-1)L .. (j+1)L))
template <int N>
__global__ __launch_bounds__(64) void k_fft_u_part(const float* __restrict__ u, const float2* __restrict__ table,
                                                   float2* __restrict__ U, int T, int nb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* tw = reinterpret_cast<float2*>(lds);
  float2* buf = tw + N;
  constexpr int L = N / 2;
  const int b = blockIdx.x, j = blockIdx.y;
  load_twiddles<N>(tw, table);
  __syncthreads();
  const float* ub = u + (long)b * T;
  float2* Ub = U + ((long)b * nb + j) * N;
  const int s0 = (j - 1) * L;
  wave_fft<N, false>(
      tw, buf,
      [&](int i, int) {
        const int s = s0 + i;
        const float v = ub[min(max(s, 0), T - 1)];
        return make_float2((s >= 0 && s < T) ? v : 0.f, 0.f);
      },
      [&](int i, float2 v, int) { Ub[i] = v; });
}

// forward: m[b][jL .. jL + L)[k0 .. k0 + 31] for one batch item, block and 16 channel pairs
template <int N>
__global__ __launch_bounds__(kFftThreads) void k_conv_fwd_fft_part(const float2* __restrict__ U,
                                                                   const float2* __restrict__ G,
                                                                   const float2* __restrict__ table,
                                                                   float* __restrict__ m, int T, int M, int npairs,
                                                                   int nb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using Lds = FftLds<N>;
  constexpr int L = N / 2;
  float2* tw = reinterpret_cast<float2*>(lds + Lds::TW);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2* buf = reinterpret_cast<float2*>(lds + Lds::WORK) + (size_t)wave * Lds::PADN;
  float* tile = lds + Lds::TILE;
  const int b = blockIdx.y, j = blockIdx.z, p0 = blockIdx.x * kFftPairs;
  const float2* Ub = U + (long)b * nb * N;
  const int rows = min(L, T - j * L);

  load_twiddles<N>(tw, table);
  __syncthreads();
  for (int pl = wave; pl < kFftPairs; pl += kFftWaves) {
    const int p = p0 + pl;
    if (p >= npairs) break;
    // sum of the partition products, formed with coalesced loads (a lane owns elements lane,
    // lane + 64, ...) and parked in the wave's buffer, which the transform then works on in place
    constexpr int PER = N / 64;
    float2 s[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) s[k] = make_float2(0.f, 0.f);
    for (int q = 0; q <= j; ++q) {
      const float2* Uq = Ub + (long)(j - q) * N + lane;
      const float2* Gq = G + ((long)q * npairs + p) * N + lane;
      float2 uq[PER], gq[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        uq[k] = Uq[64 * k];
        gq[k] = Gq[64 * k];
      }
#pragma unroll
      for (int k = 0; k < PER; ++k) s[k] = s[k] + cmul(uq[k], gq[k]);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) buf[fpad(lane + 64 * k)] = s[k];
    wave_lds_fence();
    wave_fft<N, true>(
        tw, buf, [&](int i, int) { return buf[fpad(i)]; },
        [&](int i, float2 v, int) {
          if (i >= L && i - L < rows) {
            tile[(i - L) * kTileLd + 2 * pl] = v.x;
            tile[(i - L) * kTileLd + 2 * pl + 1] = v.y;
          }
        });
  }
  __syncthreads();
  const int k0 = 2 * p0;
  float* mb = m + ((long)b * T + (long)j * L) * M;
  for (int idx = threadIdx.x; idx < rows * kFftCols; idx += kFftThreads) {
    const int t = idx / kFftCols, c = idx % kFftCols;
    if (k0 + c < M) mb[(long)t * M + k0 + c] = tile[t * kTileLd + c];
  }
}

// adjoint, first kernel: for one batch item, 32 channels and the output blocks j0 .. j0 + JB - 1,
// Sp[((g * B + b) * nb + j)][f] = sum_{c >= j} sum_pairs fft(z_c)[f] * Gr_{c-j}[f]
template <int N>
__global__ __launch_bounds__(kFftThreads) void k_conv_bwd_fft_part(const float* __restrict__ gm,
                                                                   const float2* __restrict__ Gr,
                                                                   const float2* __restrict__ table,
                                                                   float2* __restrict__ Sp, int B, int T, int M,
                                                                   int npairs, int nb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using Lds = FftLds<N>;
  constexpr int L = N / 2;
  float2* tw = reinterpret_cast<float2*>(lds + Lds::TW);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2* buf = reinterpret_cast<float2*>(lds + Lds::WORK) + (size_t)wave * Lds::PADN;
  float* tile = lds + Lds::TILE;  // [N][kTileLd]: one segment of 32 channels
  const int b = blockIdx.y, p0 = blockIdx.x * kFftPairs, k0 = 2 * p0;
  const int j0 = blockIdx.z * kPartJB, j1 = min(nb, j0 + kPartJB);
  const float* gmb = gm + (long)b * T * M;

  // a lane accumulates the spectrum elements lane, lane + 64, ...: the transform parks its result
  // in the wave's buffer and the products with the mirrored-filter spectra are formed from there,
  // with coalesced loads that are all in flight together (inside the last stage's sink they
  // were issued one by one, in that stage's strided order)
  constexpr int PER = N / 64;
  float2 acc[kPartJB][PER];
#pragma unroll
  for (int a = 0; a < kPartJB; ++a)
#pragma unroll
    for (int i = 0; i < PER; ++i) acc[a][i] = make_float2(0.f, 0.f);

  load_twiddles<N>(tw, table);
  for (int c = j0; c < nb; ++c) {
    __syncthreads();  // the previous segment's transforms have read the tile
    const int t0 = c * L;
    for (int idx = threadIdx.x; idx < N * kFftCols; idx += kFftThreads) {
      const int tl = idx / kFftCols, cc = idx % kFftCols;
      const int tt = t0 + tl;
      tile[tl * kTileLd + cc] = (tt < T && k0 + cc < M) ? gmb[(long)tt * M + k0 + cc] : 0.f;
    }
    __syncthreads();
    for (int pl = wave; pl < kFftPairs; pl += kFftWaves) {
      const int p = p0 + pl;
      if (p >= npairs) break;
      wave_fft<N, false>(
          tw, buf,
          [&](int i, int) {
            const float* t2 = tile + i * kTileLd + 2 * pl;
            return make_float2(t2[0], t2[1]);
          },
          [&](int i, float2 v, int) { buf[fpad(i)] = v; });
      wave_lds_fence();
      float2 z[PER];
#pragma unroll
      for (int q = 0; q < PER; ++q) z[q] = buf[fpad(lane + 64 * q)];
#pragma unroll
      for (int a = 0; a < kPartJB; ++a) {
        const int j = j0 + a;
        if (j < j1 && j <= c) {  // (wave-uniform)
          const float2* Gq = Gr + ((long)(c - j) * npairs + p) * N + lane;
          float2 gq[PER];
#pragma unroll
          for (int q = 0; q < PER; ++q) gq[q] = Gq[64 * q];
#pragma unroll
          for (int q = 0; q < PER; ++q) acc[a][q] = acc[a][q] + cmul(z[q], gq[q]);
        }
      }
      wave_lds_fence();  // the buffer is the next transform's workspace
    }
  }
  // per output block: park the wave's sum in its buffer, add the waves in a fixed order -> Sp
  const float2* w0 = reinterpret_cast<const float2*>(lds + Lds::WORK);
#pragma unroll
  for (int a = 0; a < kPartJB; ++a) {
    const int j = j0 + a;
    if (j >= j1) break;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) buf[fpad(lane + 64 * q)] = acc[a][q];
    __syncthreads();
    float2* out = Sp + (((long)blockIdx.x * B + b) * nb + j) * N;
    for (int i = threadIdx.x; i < N; i += kFftThreads) {
      const int pi = fpad(i);
      float2 s = w0[pi];
#pragma unroll
      for (int wv = 1; wv < kFftWaves; ++wv) s = s + w0[(size_t)wv * Lds::PADN + pi];
      out[i] = s;
    }
  }
}

// adjoint, second kernel: gu[b][jL .. jL + L) = Re ifft(sum_groups Sp[g][b][j])[L - 1 .. 2L - 1), masked by u > 0
template <int N>
__global__ __launch_bounds__(64) void k_fft_gu_part(const float2* __restrict__ Sp, const float2* __restrict__ table,
                                                    const float* __restrict__ u, float* __restrict__ gu, int B, int T,
                                                    int groups, int nb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* tw = reinterpret_cast<float2*>(lds);
  float2* buf = tw + N;
  float2* sum = buf + FftLds<N>::PADN;
  constexpr int L = N / 2;
  const int b = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
  load_twiddles<N>(tw, table);
  float2 s[N / 64];
#pragma unroll
  for (int q = 0; q < N / 64; ++q) s[q] = Sp[((long)b * nb + j) * N + lane + 64 * q];
  for (int g = 1; g < groups; ++g) {
    float2 t[N / 64];
#pragma unroll
    for (int q = 0; q < N / 64; ++q) t[q] = Sp[(((long)g * B + b) * nb + j) * N + lane + 64 * q];
#pragma unroll
    for (int q = 0; q < N / 64; ++q) s[q] = s[q] + t[q];
  }
#pragma unroll
  for (int q = 0; q < N / 64; ++q) sum[fpad(lane + 64 * q)] = s[q];
  __syncthreads();
  const float* ub = u ? u + (long)b * T : nullptr;
  float* out = gu + (long)b * T;
  wave_fft<N, true>(
      tw, buf, [&](int i, int) { return sum[fpad(i)]; },
      [&](int i, float2 v, int) {
        const int sidx = j * L + i - (L - 1);
        if (i >= L - 1 && i < 2 * L - 1 && sidx < T) out[sidx] = (ub && !(ub[sidx] > 0.f)) ? 0.f : v.x;
      });
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline int cdivf(long a, long b) { return (int)((a + b - 1) / b); }

int fft_length(int T) {
  for (int n = 256; n <= 1024; n <<= 1)
    if (n >= 2 * T - 1) return n;
  return 0;  // longer sequences take the partitioned kernels
}

// layout of the prepared buffer of a long sequence:
//   G [nbF][npairs][kPartNF] | Gr [nbA][npairs][kPartNA] | twiddles of kPartNF | twiddles of kPartNA
struct PartLayout {
  int nbF, nbA;  // blocks of kPartNF / 2 and of kPartNA / 2 samples; 0: T is outside the range
  size_t offGr, offTabF, offTabA, total;  // in float2
};
static PartLayout part_layout(int M, int T) {
  PartLayout l{0, 0, 0, 0, 0, 0};
  if (fft_length(T) || T <= 0 || T > 16384) return l;
  const size_t npairs = (size_t)((M + 1) / 2);
  l.nbF = cdivf(T, kPartNF / 2);
  l.nbA = cdivf(T, kPartNA / 2);
  l.offGr = (size_t)l.nbF * npairs * kPartNF;
  l.offTabF = l.offGr + (size_t)l.nbA * npairs * kPartNA;
  l.offTabA = l.offTabF + kPartNF;
  l.total = l.offTabA + kPartNA;
  return l;
}
bool fft_route_available(int T) { return fft_length(T) != 0 || part_layout(2, T).nbF != 0; }

size_t fftprep_bytes(int M, int T) {
  const int N = fft_length(T);
  const size_t npairs = (size_t)((M + 1) / 2);
  if (N) return sizeof(float2) * (2 * npairs + 1) * N;  // G | Gr | twiddle table
  return sizeof(float2) * part_layout(M, T).total;
}

#define FFT_DISPATCH(N_, CALL)  \
  switch (N_) {                 \
    case 256: { constexpr int NN = 256; CALL; } break;   \
    case 512: { constexpr int NN = 512; CALL; } break;   \
    default: { constexpr int NN = 1024; CALL; } break;   \
  }

#define FFT_LAUNCH_CHECK(name)                                                                        \
  do {                                                                                                \
    hipError_t e_ = hipGetLastError();                                                                \
    if (e_ != hipSuccess) return set_error("launch of %s failed: %s", name, hipGetErrorString(e_));   \
  } while (0)

template <class K>
static int reserve_lds(K kernel, size_t lds, const char* name) {
  if (lds > 160 * 1024) return set_error("%s needs %zu bytes of LDS (sequence too long for the FFT kernels)", name, lds);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return set_error("cannot reserve %zu bytes of LDS for %s", lds, name);
  return 0;
}

static int no_fft_route(int T) { return set_error("the FFT convolution supports T <= 16384 (got %d)", T); }

int fft_prepare(const lmufft_shape& s, const float* H, void* prep, hipStream_t st) {
  const int N = fft_length(s.T);
  const int npairs = (s.M + 1) / 2;
  if (!N) {
    const PartLayout l = part_layout(s.M, s.T);
    if (!l.nbF) return no_fft_route(s.T);
    float2* base = (float2*)prep;
    const size_t ldsF = FftLds<kPartNF>::bytes_single(), ldsA = FftLds<kPartNA>::bytes_single();
    if (int rc = reserve_lds(k_fft_prep_part<kPartNF>, ldsF, "k_fft_prep")) return rc;
    if (int rc = reserve_lds(k_fft_prep_part<kPartNA>, ldsA, "k_fft_prep")) return rc;
    {
      ProfScope ps("k_fft_prep", st);
      hipLaunchKernelGGL(k_fft_prep_part<kPartNF>, dim3(npairs, l.nbF), dim3(64), ldsF, st, H, base, base + l.offTabF,
                         s.M, s.T, s.ldH, npairs, 0);
      hipLaunchKernelGGL(k_fft_prep_part<kPartNA>, dim3(npairs, l.nbA), dim3(64), ldsA, st, H, base + l.offGr,
                         base + l.offTabA, s.M, s.T, s.ldH, npairs, 1);
    }
    FFT_LAUNCH_CHECK("k_fft_prep");
    return 0;
  }
  const dim3 grid(npairs, 2);
  FFT_DISPATCH(N, {
    const size_t lds = FftLds<NN>::bytes_single();
    if (int rc = reserve_lds(k_fft_prep<NN>, lds, "k_fft_prep")) return rc;
    ProfScope ps("k_fft_prep", st);
    hipLaunchKernelGGL(k_fft_prep<NN>, grid, dim3(64), lds, st, H, (float2*)prep, s.M, s.T, s.ldH, npairs);
  });
  FFT_LAUNCH_CHECK("k_fft_prep");
  return 0;
}

int fft_conv_groups(int M) { return cdivf((M + 1) / 2, kFftPairs); }  // the most any route uses (workspace sizing)

// Channel pairs per workgroup of the one-transform kernels: 24 where that takes fewer (rounds of workgroups) x
// (transforms per wave) than 16 and its [T][49] tile fits the LDS.
static int fft_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  return (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
}
template <int N>
static int fft_one_pairs(int npairs, int B, int T) {
  if (FftLds<N>::bytes(T, 24) > 160 * 1024) return 16;
  const int cus = fft_cus();
  const long c24 = (long)cdivf((long)cdivf(npairs, 24) * B, cus) * 3, c16 = (long)cdivf((long)cdivf(npairs, 16) * B, cus) * 2;
  return c24 < c16 ? 24 : 16;
}

// scratch of the FFT route: spectra of u (forward) or per-group partial spectra (adjoint)
size_t fft_conv_workspace_bytes(const lmufft_shape& s) {
  const int N = fft_length(s.T);
  if (N) return sizeof(float2) * (size_t)std::max(1, fft_conv_groups(s.M)) * s.B * N;
  const PartLayout l = part_layout(s.M, s.T);
  if (!l.nbF) return 0;
  return sizeof(float2) * std::max((size_t)s.B * l.nbF * kPartNF,  // forward: spectra of the u segments
                                   (size_t)std::max(1, fft_conv_groups(s.M)) * s.B * l.nbA * kPartNA);
}

int fft_conv_forward(const lmufft_shape& s, const float* u, const void* prep, float* m, void* ws, hipStream_t st) {
  const int N = fft_length(s.T);
  const int npairs = (s.M + 1) / 2;
  float2* U = (float2*)ws;
  if (!N) {
    const PartLayout l = part_layout(s.M, s.T);
    if (!l.nbF) return no_fft_route(s.T);
    constexpr int NN = kPartNF;
    const int nb = l.nbF;
    const float2* table = (const float2*)prep + l.offTabF;
    const size_t lds1 = FftLds<NN>::bytes_single();
    if (int rc = reserve_lds(k_fft_u_part<NN>, lds1, "k_fft_u")) return rc;
    {
      ProfScope ps("k_fft_u", st);
      hipLaunchKernelGGL(k_fft_u_part<NN>, dim3(s.B, nb), dim3(64), lds1, st, u, table, U, s.T, nb);
    }
    FFT_LAUNCH_CHECK("k_fft_u");
    const size_t lds = FftLds<NN>::bytes(NN / 2);
    if (int rc = reserve_lds(k_conv_fwd_fft_part<NN>, lds, "k_conv_fwd_fft")) return rc;
    {
      ProfScope ps("k_conv_fwd_fft", st);
      hipLaunchKernelGGL(k_conv_fwd_fft_part<NN>, dim3(cdivf(npairs, kFftPairs), s.B, nb), dim3(kFftThreads), lds, st,
                         (const float2*)U, (const float2*)prep, table, m, s.T, s.M, npairs, nb);
    }
    FFT_LAUNCH_CHECK("k_conv_fwd_fft");
    return 0;
  }
  const float2* table = (const float2*)prep + (size_t)2 * npairs * N;
  FFT_DISPATCH(N, {
    const int P = fft_one_pairs<NN>(npairs, s.B, s.T);
    const size_t lds = FftLds<NN>::bytes(s.T, P);
    const dim3 grid(cdivf(npairs, P), s.B);
    ProfScope ps("k_conv_fwd_fft", st);
    if (P == 24) {
      if (int rc = reserve_lds(k_conv_fwd_fft<NN, 24, true>, lds, "k_conv_fwd_fft")) return rc;
      hipLaunchKernelGGL((k_conv_fwd_fft<NN, 24, true>), grid, dim3(kFftThreads), lds, st, u, U, (const float2*)prep,
                         table, m, s.T, s.M, npairs);
    } else {
      if (int rc = reserve_lds(k_conv_fwd_fft<NN, 16, true>, lds, "k_conv_fwd_fft")) return rc;
      hipLaunchKernelGGL((k_conv_fwd_fft<NN, 16, true>), grid, dim3(kFftThreads), lds, st, u, U, (const float2*)prep,
                         table, m, s.T, s.M, npairs);
    }
  });
  FFT_LAUNCH_CHECK("k_conv_fwd_fft");
  return 0;
}

// gu[b][s] (masked by u > 0 when u != NULL); ws as sized by fft_conv_workspace_bytes
int fft_conv_backward(const lmufft_shape& s, const float* gm, const void* prep, const float* u, float* gu, void* ws,
                      hipStream_t st) {
  const int N = fft_length(s.T);
  const int npairs = (s.M + 1) / 2;
  const int groups = fft_conv_groups(s.M);
  float2* Sp = (float2*)ws;
  if (!N) {
    const PartLayout l = part_layout(s.M, s.T);
    if (!l.nbF) return no_fft_route(s.T);
    constexpr int NN = kPartNA;
    const int nb = l.nbA;
    const float2* Gr = (const float2*)prep + l.offGr;
    const float2* table = (const float2*)prep + l.offTabA;
    const size_t lds = FftLds<NN>::bytes(NN);
    if (int rc = reserve_lds(k_conv_bwd_fft_part<NN>, lds, "k_conv_bwd_fft")) return rc;
    {
      ProfScope ps("k_conv_bwd_fft", st);
      hipLaunchKernelGGL(k_conv_bwd_fft_part<NN>, dim3(groups, s.B, cdivf(nb, kPartJB)), dim3(kFftThreads), lds, st, gm,
                         Gr, table, Sp, s.B, s.T, s.M, npairs, nb);
    }
    FFT_LAUNCH_CHECK("k_conv_bwd_fft");
    const size_t lds1 = FftLds<NN>::bytes_single();
    if (int rc = reserve_lds(k_fft_gu_part<NN>, lds1, "k_fft_gu")) return rc;
    {
      ProfScope ps("k_fft_gu", st);
      hipLaunchKernelGGL(k_fft_gu_part<NN>, dim3(s.B, nb), dim3(64), lds1, st, (const float2*)Sp, table, u, gu, s.B,
                         s.T, groups, nb);
    }
    FFT_LAUNCH_CHECK("k_fft_gu");
    return 0;
  }
  const float2* Gr = (const float2*)prep + (size_t)npairs * N;
  const float2* table = (const float2*)prep + (size_t)2 * npairs * N;
  FFT_DISPATCH(N, {
    const int P = fft_one_pairs<NN>(npairs, s.B, s.T);
    const int ngroups = cdivf(npairs, P);
    const size_t lds = FftLds<NN>::bytes(s.T, P);
    {
      ProfScope ps("k_conv_bwd_fft", st);
      if (P == 24) {
        if (int rc = reserve_lds(k_conv_bwd_fft<NN, 24>, lds, "k_conv_bwd_fft")) return rc;
        hipLaunchKernelGGL((k_conv_bwd_fft<NN, 24>), dim3(ngroups, s.B), dim3(kFftThreads), lds, st, gm, Gr, table, Sp,
                           s.B, s.T, s.M, npairs);
      } else {
        if (int rc = reserve_lds(k_conv_bwd_fft<NN, 16>, lds, "k_conv_bwd_fft")) return rc;
        hipLaunchKernelGGL((k_conv_bwd_fft<NN, 16>), dim3(ngroups, s.B), dim3(kFftThreads), lds, st, gm, Gr, table, Sp,
                           s.B, s.T, s.M, npairs);
      }
    }
    const size_t lds1 = FftLds<NN>::bytes_single();
    if (int rc = reserve_lds(k_fft_gu<NN>, lds1, "k_fft_gu")) return rc;
    ProfScope ps("k_fft_gu", st);
    hipLaunchKernelGGL(k_fft_gu<NN>, dim3(s.B), dim3(64), lds1, st, (const float2*)Sp, table, u, gu, s.B, s.T,
                       ngroups);
  });
  FFT_LAUNCH_CHECK("k_conv_bwd_fft");
  return 0;
}

}  // namespace lmufft

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
// Stockham autosort stages of radix 16, 16, N/256 with the butterflies in
// registers and the exchanges through a wave-private LDS buffer whose index is
// padded by one element per 16 (conflict-free ds_read_b64 / ds_write_b64).
// The first stage reads straight from its source (global u, the LDS copy of a
// gm tile, or the spectrum product) and the last stage writes straight to its
// sink (the [T][k] output tile, or the spectrum accumulator), so a transform
// makes two LDS round trips.  Output tiles are transposed through LDS so that
// m is written, and gm read, in 128-byte row segments.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

constexpr int kFftThreads = 256;   // 4 waves, one transform each at a time
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

// One Stockham stage of an N-point transform done by ONE wave.
//   src(i)      -> element i of the stage input   (i in [0, N))
//   dst(i, v, slot) -> element i of the stage output; slot = q*R + r counts the lane's outputs
//                  in a fixed order (the same for every transform of that stage)
//   tw[k] = exp(-2 pi i k / N), k in [0, N)
// Butterfly j (of N/R) reads i = j + r*N/R, multiplies by W^(r * (j % NS) * N/(NS*R)),
// takes the R-point DFT and writes i = (j / NS)*NS*R + (j % NS) + r*NS.
template <int N, int R, int NS, bool INV, class Src, class Dst>
__device__ __forceinline__ void fft_stage(const float2* __restrict__ tw, Src&& src, Dst&& dst) {
  constexpr int NB = N / R;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < (NB + 63) / 64; ++q) {
    const int j = lane + 64 * q;
    if (NB % 64 != 0 && j >= NB) break;
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src(j + r * NB);
    const int kk = j & (NS - 1);
    if (NS > 1) {
      const int tstep = kk * (N / (NS * R));
#pragma unroll
      for (int r = 1; r < R; ++r) v[r] = twiddle<INV>(v[r], tw[(r * tstep) & (N - 1)]);
    }
    dftR<R, INV>(v);
    const int j0 = (j / NS) * (NS * R) + kk;
#pragma unroll
    for (int r = 0; r < R; ++r) dst(j0 + r * NS, v[r], q * R + r);
  }
}

// Whole transform: radix 16, 16, then N/256 (absent for N = 256).  bufA / bufB are
// wave-private LDS buffers of fpad(N) float2.  The caller places a workgroup
// barrier before reading what the last stage wrote; barriers between the stages
// are inside (every wave of the workgroup must call this the same number of
// times; `active` = false makes the wave only keep the barriers company).
template <int N, bool INV, class Src, class Dst>
__device__ __forceinline__ void wave_fft(bool active, const float2* __restrict__ tw, float2* bufA, float2* bufB,
                                         Src&& src, Dst&& dst) {
  constexpr int R3 = N / 256;
  static_assert(R3 == 1 || R3 == 2 || R3 == 4, "N in {256, 512, 1024}");
  if (active) fft_stage<N, 16, 1, INV>(tw, src, [&](int i, float2 v, int) { bufA[fpad(i)] = v; });
  __syncthreads();
  if constexpr (R3 == 1) {
    if (active) fft_stage<N, 16, 16, INV>(tw, [&](int i) { return bufA[fpad(i)]; }, dst);
  } else {
    if (active)
      fft_stage<N, 16, 16, INV>(tw, [&](int i) { return bufA[fpad(i)]; },
                                [&](int i, float2 v, int) { bufB[fpad(i)] = v; });
    __syncthreads();
    if (active) fft_stage<N, R3, 256, INV>(tw, [&](int i) { return bufB[fpad(i)]; }, dst);
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

// LDS carve (floats): tw[2N] | U[2*fpad(N)] | per wave A,B [4 * 2*fpad(N)] each | tile [T * kTileLd]
template <int N>
struct FftLds {
  static constexpr int PADN = N + N / 16;
  static constexpr int TW = 0;
  static constexpr int U = TW + 2 * N;
  static constexpr int WORK = U + 2 * PADN;
  static constexpr int TILE = WORK + 4 * 2 * 2 * PADN;
  static size_t bytes(int T) { return sizeof(float) * ((size_t)TILE + (size_t)T * kTileLd); }
};

// ---------------------------------------------------------------------------
// spectra of the filters, once per (H, T):
//   G [p][f] = fft(H_2p + i H_2p+1)[f] / N                  (forward conv)
//   Gr[p][f] = fft(rev(H_2p) - i rev(H_2p+1))[f] / N        (adjoint), rev(h)[j] = h[T-1-j]
// one workgroup (one active wave) per pair and kind
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void k_fft_prep(const float* __restrict__ H, float2* __restrict__ G, int M, int T,
                                                 int ldH, int npairs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* tw = reinterpret_cast<float2*>(lds);
  float2* bufA = tw + N;
  float2* bufB = bufA + FftLds<N>::PADN;
  const int p = blockIdx.x, adj = blockIdx.y;
  fill_twiddles<N>(tw);
  __syncthreads();
  const float* h0 = H + (long)(2 * p) * ldH;
  const float* h1 = (2 * p + 1 < M) ? H + (long)(2 * p + 1) * ldH : nullptr;
  float2* out = G + ((long)adj * npairs + p) * N;
  const float scale = 1.0f / (float)N;
  wave_fft<N, false>(
      true, tw, bufA, bufB,
      [&](int i) {
        if (i >= T) return make_float2(0.f, 0.f);
        const int j = adj ? T - 1 - i : i;
        const float a = h0[j], b = h1 ? h1[j] : 0.f;
        return make_float2(a, adj ? -b : b);
      },
      [&](int i, float2 v, int) { out[i] = make_float2(v.x * scale, v.y * scale); });
}

// ---------------------------------------------------------------------------
// forward: m[b][t][k0 .. k0+31] for one batch item and 16 channel pairs
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kFftThreads) void k_conv_fwd_fft(const float* __restrict__ u,
                                                              const float2* __restrict__ G,
                                                              float* __restrict__ m, int T, int M, int npairs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using L = FftLds<N>;
  float2* tw = reinterpret_cast<float2*>(lds + L::TW);
  float2* U = reinterpret_cast<float2*>(lds + L::U);
  const int wave = threadIdx.x >> 6;
  float2* bufA = reinterpret_cast<float2*>(lds + L::WORK) + (size_t)wave * 2 * L::PADN;
  float2* bufB = bufA + L::PADN;
  float* tile = lds + L::TILE;
  const int b = blockIdx.y, p0 = blockIdx.x * kFftPairs;
  const float* ub = u + (long)b * T;

  fill_twiddles<N>(tw);
  __syncthreads();
  // spectrum of u (wave 0)
  wave_fft<N, false>(
      wave == 0, tw, bufA, bufB, [&](int i) { return make_float2(i < T ? ub[i] : 0.f, 0.f); },
      [&](int i, float2 v, int) { U[fpad(i)] = v; });
  __syncthreads();
  // four pairs per wave: ifft(U * G_p) -> columns 2*pl, 2*pl+1 of the tile
  for (int q = 0; q < kFftPairs / 4; ++q) {
    const int pl = q * 4 + wave;
    const int p = p0 + pl;
    const bool active = p < npairs;
    const float2* Gp = G + (long)(active ? p : 0) * N;
    wave_fft<N, true>(
        active, tw, bufA, bufB, [&](int i) { return cmul(U[fpad(i)], Gp[i]); },
        [&](int i, float2 v, int) {
          if (i < T) {
            tile[i * kTileLd + 2 * pl] = v.x;
            tile[i * kTileLd + 2 * pl + 1] = v.y;
          }
        });
    __syncthreads();  // bufA / bufB are reused by the next pair
  }
  // tile -> m[b][t][k0 + c], 128-byte row segments
  const int k0 = 2 * p0;
  float* mb = m + (long)b * T * M;
  for (int idx = threadIdx.x; idx < T * kFftCols; idx += kFftThreads) {
    const int t = idx / kFftCols, c = idx % kFftCols;
    if (k0 + c < M) mb[(long)t * M + k0 + c] = tile[t * kTileLd + c];
  }
}

// ---------------------------------------------------------------------------
// adjoint: partial gu[b][s] from the 32 channels k0 .. k0+31 of one batch item
// slab layout part[(kgroup * B + b) * T + s]
// ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(kFftThreads) void k_conv_bwd_fft(const float* __restrict__ gm,
                                                              const float2* __restrict__ Gr,
                                                              float* __restrict__ part, int B, int T, int M,
                                                              int npairs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using L = FftLds<N>;
  float2* tw = reinterpret_cast<float2*>(lds + L::TW);
  float2* S = reinterpret_cast<float2*>(lds + L::U);  // summed spectrum
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2* bufA = reinterpret_cast<float2*>(lds + L::WORK) + (size_t)wave * 2 * L::PADN;
  float2* bufB = bufA + L::PADN;
  float* tile = lds + L::TILE;
  const int b = blockIdx.y, p0 = blockIdx.x * kFftPairs, k0 = 2 * p0;
  const float* gmb = gm + (long)b * T * M;

  fill_twiddles<N>(tw);
  for (int idx = threadIdx.x; idx < T * kFftCols; idx += kFftThreads) {
    const int t = idx / kFftCols, c = idx % kFftCols;
    tile[t * kTileLd + c] = (k0 + c < M) ? gmb[(long)t * M + k0 + c] : 0.f;
  }
  __syncthreads();

  // every lane accumulates the spectrum elements that the last stage hands it:
  // the (index -> lane, slot) map of a stage is the same for every transform
  constexpr int R3_ = N / 256;
  constexpr int RL_ = R3_ == 1 ? 16 : R3_;                  // radix of the last stage
  constexpr int PER = ((N / RL_ + 63) / 64) * RL_;          // outputs of the last stage per lane
  float2 acc[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) acc[i] = make_float2(0.f, 0.f);

  for (int q = 0; q < kFftPairs / 4; ++q) {
    const int pl = q * 4 + wave;
    const int p = p0 + pl;
    const bool active = p < npairs;
    const float2* Gp = Gr + (long)(active ? p : 0) * N;
    wave_fft<N, false>(
        active, tw, bufA, bufB,
        [&](int i) {
          return i < T ? make_float2(tile[i * kTileLd + 2 * pl], tile[i * kTileLd + 2 * pl + 1])
                       : make_float2(0.f, 0.f);
        },
        [&](int i, float2 v, int slot) { acc[slot] = acc[slot] + cmul(v, Gp[i]); });
    __syncthreads();
  }
  // the four waves' spectra, in the index order of the last stage, are summed into S
  {
    // replay the last stage's index sequence to know which element each slot holds
    constexpr int R3 = N / 256;
    constexpr int RL = R3 == 1 ? 16 : R3;
    constexpr int NSL = R3 == 1 ? 16 : 256;
    constexpr int NB = N / RL;
    float2* mine = bufA;  // wave-private
#pragma unroll
    for (int q = 0; q < (NB + 63) / 64; ++q) {
      const int j = lane + 64 * q;
      if (NB % 64 != 0 && j >= NB) break;
      const int kk = j & (NSL - 1);
      const int j0 = (j / NSL) * (NSL * RL) + kk;
#pragma unroll
      for (int r = 0; r < RL; ++r) mine[fpad(j0 + r * NSL)] = acc[q * RL + r];
    }
  }
  __syncthreads();
  {
    float2* w0 = reinterpret_cast<float2*>(lds + L::WORK);
    for (int i = threadIdx.x; i < N; i += kFftThreads) {
      const int pi = fpad(i);
      S[pi] = ((w0[pi] + w0[2 * L::PADN + pi]) + w0[4 * L::PADN + pi]) + w0[6 * L::PADN + pi];
    }
  }
  __syncthreads();
  float* out = part + ((long)blockIdx.x * B + b) * T;
  wave_fft<N, true>(
      wave == 0, tw, bufA, bufB, [&](int i) { return S[fpad(i)]; },
      [&](int i, float2 v, int) {
        if (i >= T - 1 && i < 2 * T - 1) out[i - (T - 1)] = v.x;
      });
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline int cdivf(long a, long b) { return (int)((a + b - 1) / b); }

int fft_length(int T) {
  for (int n = 256; n <= 1024; n <<= 1)
    if (n >= 2 * T - 1) return n;
  return 0;  // not supported by the LDS-resident kernels
}

size_t fftprep_bytes(int M, int T) {
  const int N = fft_length(T);
  if (!N) return 0;
  return sizeof(float2) * 2 * (size_t)((M + 1) / 2) * N;
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

int fft_prepare(const lmufft_shape& s, const float* H, void* prep, hipStream_t st) {
  const int N = fft_length(s.T);
  if (!N) return set_error("the LDS-resident FFT convolution supports T <= 512 (got %d)", s.T);
  const int npairs = (s.M + 1) / 2;
  const dim3 grid(npairs, 2);
  FFT_DISPATCH(N, {
    const size_t lds = sizeof(float2) * (size_t)(NN + 2 * FftLds<NN>::PADN);
    if (int rc = reserve_lds(k_fft_prep<NN>, lds, "k_fft_prep")) return rc;
    ProfScope ps("k_fft_prep", st);
    hipLaunchKernelGGL(k_fft_prep<NN>, grid, dim3(64), lds, st, H, (float2*)prep, s.M, s.T, s.ldH, npairs);
  });
  FFT_LAUNCH_CHECK("k_fft_prep");
  return 0;
}

int fft_conv_forward(const lmufft_shape& s, const float* u, const void* prep, float* m, hipStream_t st) {
  const int N = fft_length(s.T);
  if (!N) return set_error("the LDS-resident FFT convolution supports T <= 512 (got %d)", s.T);
  const int npairs = (s.M + 1) / 2;
  const dim3 grid(cdivf(npairs, kFftPairs), s.B);
  FFT_DISPATCH(N, {
    const size_t lds = FftLds<NN>::bytes(s.T);
    if (int rc = reserve_lds(k_conv_fwd_fft<NN>, lds, "k_conv_fwd_fft")) return rc;
    ProfScope ps("k_conv_fwd_fft", st);
    hipLaunchKernelGGL(k_conv_fwd_fft<NN>, grid, dim3(kFftThreads), lds, st, u, (const float2*)prep, m, s.T, s.M,
                       npairs);
  });
  FFT_LAUNCH_CHECK("k_conv_fwd_fft");
  return 0;
}

int fft_conv_groups(int M) { return cdivf((M + 1) / 2, kFftPairs); }

// part: [groups][B][T] floats
int fft_conv_backward(const lmufft_shape& s, const float* gm, const void* prep, float* part, hipStream_t st) {
  const int N = fft_length(s.T);
  if (!N) return set_error("the LDS-resident FFT convolution supports T <= 512 (got %d)", s.T);
  const int npairs = (s.M + 1) / 2;
  const dim3 grid(cdivf(npairs, kFftPairs), s.B);
  const float2* Gr = (const float2*)prep + (size_t)npairs * N;
  FFT_DISPATCH(N, {
    const size_t lds = FftLds<NN>::bytes(s.T);
    if (int rc = reserve_lds(k_conv_bwd_fft<NN>, lds, "k_conv_bwd_fft")) return rc;
    ProfScope ps("k_conv_bwd_fft", st);
    hipLaunchKernelGGL(k_conv_bwd_fft<NN>, grid, dim3(kFftThreads), lds, st, gm, Gr, part, s.B, s.T, s.M, npairs);
  });
  FFT_LAUNCH_CHECK("k_conv_bwd_fft");
  return 0;
}

}  // namespace lmufft

// fp32 kernels of the LMUFFT layer for gfx950 (MI355X) and their C ABI.
//
// Reference arithmetic: src/lmu.py:297-320 (forward); the backward is what
// autograd derives from those lines.  Kernel inventory (all fp32, f32-MFMA):
//
//   k_encode_fwd   u = relu(x . w_u + b_u)                      HBM-bound
//   k_conv_fwd     m = Toeplitz(u) * H^T  (causal conv)         MFMA f32
//   k_proj_fwd     h = relu([m;x] W_h^T + b_h), h_n             MFMA f32
//   k_dgrad        gm = gz W_h[:, :M]; gx = gz W_h[:, M:] + ga w_u    MFMA f32
//   k_wgrad        gW_h, gb_h partial slabs (split over rows)   MFMA f32
//   k_conv_bwd     gu partial = diag-sums of gm * H             MFMA f32
//   k_gu_reduce    gu = sum of partial slabs, ga = gu*(u>0)
//   k_encode_bwd   gw_u, gb_u partial slabs                     HBM-bound
//   k_reduce_slabs deterministic slab sums for parameter grads
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

#include "gemm_core_f32.h"
#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

// ---------------------------------------------------------------------------
// u = relu(x . w_u + b_u)                                    src/lmu.py:300
// one wave per row, 16-byte loads, wave-level shuffle reduction
// ---------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_encode_fwd(const float* __restrict__ x,
                                                         const float* __restrict__ w_u,
                                                         const float* __restrict__ b_u,
                                                         float* __restrict__ u, int rows, int D) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float bias = b_u[0];
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float* xr = x + (long)row * D;
    float s = 0.f;
    if (VEC) {
      for (int c = lane * 4; c < D; c += 256) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w_u + c);
        s += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
      }
    } else {
      for (int c = lane; c < D; c += 64) s += xr[c] * w_u[c];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) u[row] = fmaxf(s + bias, 0.f);
  }
}

// ---------------------------------------------------------------------------
// m[b,t,k] = sum_{tau<=t} u[b,t-tau] H[k,tau]               src/lmu.py:303-312
// Per batch item a GEMM: A[t][tau] = u[t-tau] (lower-triangular Toeplitz,
// generated on the fly), B[tau][k] = H[k][tau].  Only the causal part of the
// K range is visited.  Grid: (k tiles, batch, t tiles) with the deepest
// (latest) t tiles dispatched first.
// ---------------------------------------------------------------------------
using ConvFwdTile = Tile<64, 192, 32, 2, 2, KC, KC>;

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_conv_fwd(const float* __restrict__ u,
                                                       const float* __restrict__ H,
                                                       float* __restrict__ m, int T, int M, int ldH) {
  using TL = ConvFwdTile;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n0 = blockIdx.x * TL::BN;
  const int b = blockIdx.y;
  const int tb = gridDim.z - 1 - blockIdx.z;
  const int t0 = tb * TL::BM;
  const float* ub = u + (long)b * T;

  const float* zf = reinterpret_cast<const float*>(kZeroBytes);
  auto la = plain_load([=](int r, int k) -> f32x4 {
    const int t = t0 + r;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = t - (k + j);
      v[j] = *((t < T && s >= 0) ? ub + s : zf);
    }
    return v;
  });
  auto lb = plain_load([=](int r, int k) -> f32x4 {
    const int n = n0 + r;
    return load4<VEC>(H, (long)n * ldH + k, k, T, n < M);
  });

  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tile_zero<TL>(acc);
  const int kend = min(T, t0 + TL::BM);
  gemm_tile<TL>(smem, acc, la, lb, 0, kend, w);

  float* mb = m + (long)b * T * M;
  tile_epilogue<TL>(acc, w, [&](int mm, int nn, float v) {
    const int t = t0 + mm, n = n0 + nn;
    if (t < T && n < M) mb[(long)t * M + n] = v;
  });
}

// ---------------------------------------------------------------------------
// h = relu([m;x] W_h^T + b_h); h_n = h[:, T-1, :]           src/lmu.py:315-318
// The concatenation is never materialised: the A loader reads columns < M
// from m and the rest from x.
// ---------------------------------------------------------------------------
using ProjTile = Tile<64, 128, 32, 2, 2, KC, KC>;

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_proj_fwd(const float* __restrict__ m,
                                                       const float* __restrict__ x,
                                                       const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ h, float* __restrict__ h_n,
                                                       int rows, int T, int M, int D, int Hd) {
  using TL = ProjTile;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the column tiles of one row tile share [m;x]: same XCD
  const XcdSlot slot = xcd_slot((Hd + TL::BN - 1) / TL::BN);
  const int n0 = slot.member * TL::BN;
  const int m0 = slot.group * TL::BM;
  if (m0 >= rows) return;
  const int K = M + D;

  auto la = plain_load([=](int r, int k) -> f32x4 {
    const long row = m0 + r;
    if (VEC) {  // one 16-byte load from whichever panel holds column k
      const float* p = k < M ? m + row * M + k : x + row * D + (k - M);
      p = (row < rows && k < K) ? p : reinterpret_cast<const float*>(kZeroBytes);
      return *reinterpret_cast<const f32x4*>(p);
    } else {
      if (row >= rows) return zero4();
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k + j;
        v[j] = kk < M ? m[row * M + kk] : (kk < K ? x[row * D + (kk - M)] : 0.f);
      }
      return v;
    }
  });
  auto lb = plain_load([=](int r, int k) -> f32x4 {
    const int n = n0 + r;
    return load4<VEC>(W, (long)n * K + k, k, K, n < Hd);
  });

  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tile_zero<TL>(acc);
  gemm_tile<TL>(smem, acc, la, lb, 0, K, w);

  float bcol[TL::TN];
  preload_cols<TL>(bcol, w, [&](int nn) { return load_or_zero(bias + n0 + nn, n0 + nn < Hd); });
  tile_epilogue_ix<TL>(acc, w, [&](int, int tn, int, int mm, int nn, float v) {
    const long row = m0 + mm;
    const int n = n0 + nn;
    if (row < rows && n < Hd) h[row * Hd + n] = fmaxf(v + bcol[tn], 0.f);
  });
  // h_n: the last frame of every utterance that ends inside this row tile
  const long tile_end = min((long)rows, (long)m0 + TL::BM);
  for (long rl = ((long)m0 / T) * T + T - 1; rl < tile_end; rl += T) {
    const int ml = (int)(rl - m0);
    tile_epilogue_ix<TL>(acc, w, [&](int, int tn, int, int mm, int nn, float v) {
      const int n = n0 + nn;
      if (mm == ml && n < Hd) h_n[(rl / T) * Hd + n] = fmaxf(v + bcol[tn], 0.f);
    });
  }
}

// gz[row][n] = (h > 0) ? gh (+ gh_n on the last frame) : 0, four n at once.
// fetch issues the loads of h, gh (and gh_n); the masking runs in finish.
template <bool VEC>
struct GzSource {
  const float* __restrict__ h;
  const float* __restrict__ gh;
  const float* __restrict__ gh_n;
  int T, Hd;
  struct raw_t {
    f32x4 h, g, gl;
  };
  __device__ __forceinline__ raw_t fetch_at(long row, int n, bool ok) const {
    raw_t r;
    r.h = load4<VEC>(h, row * Hd + n, n, Hd, ok);
    r.g = gh ? load4<VEC>(gh, row * Hd + n, n, Hd, ok) : zero4();
    r.gl = zero4();
    if (gh_n) {  // (uniform) the last frame also receives the gradient of h_n
      const unsigned b = (unsigned)row / (unsigned)T;
      const bool last = (unsigned)row - b * (unsigned)T == (unsigned)(T - 1);
      r.gl = load4<VEC>(gh_n, (long)b * Hd + n, n, Hd, ok && last);
    }
    return r;
  }
  __device__ __forceinline__ f32x4 finish(const raw_t& r) const {
    f32x4 g = r.g + r.gl;
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = r.h[j] > 0.f ? g[j] : 0.f;
    return g;
  }
};

// A operand of the data gradient: tile row = frame, k = hidden unit
template <bool VEC>
struct GzRowsLoad {
  GzSource<VEC> src;
  long m0;
  int rows;
  typedef typename GzSource<VEC>::raw_t raw_t;
  __device__ __forceinline__ raw_t fetch(int r, int k) const { return src.fetch_at(m0 + r, k, m0 + r < rows); }
  __device__ __forceinline__ f32x4 finish(const raw_t& v) const { return src.finish(v); }
};

// A operand of the weight gradient: tile row = hidden unit (4 consecutive), k = frame
template <bool VEC>
struct GzColsLoad {
  GzSource<VEC> src;
  int n0, kend;
  typedef typename GzSource<VEC>::raw_t raw_t;
  __device__ __forceinline__ raw_t fetch(int r, int k) const { return src.fetch_at((long)k, n0 + r, k < kend); }
  __device__ __forceinline__ f32x4 finish(const raw_t& v) const { return src.finish(v); }
};

// ---------------------------------------------------------------------------
// out[row][j - j_begin] = sum_n gz[row][n] W_h[n][j]  (+ ga[row] * w_u[j - j_begin])
// for the column range [j_begin, j_begin + ncols) of W_h.  Called twice:
//   columns [0, M)    -> gm  (input of the adjoint convolution)
//   columns [M, M+D)  -> gx, with the encoder's rank-1 term ga w_u fused into
//                        the epilogue, so gx is written exactly once.
// ---------------------------------------------------------------------------
using DgradTile = Tile<64, 192, 32, 2, 2, KC, NC>;

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_dgrad(const float* __restrict__ h,
                                                    const float* __restrict__ gh,
                                                    const float* __restrict__ gh_n,
                                                    const float* __restrict__ W, float* __restrict__ out,
                                                    const float* __restrict__ ga,
                                                    const float* __restrict__ w_u, int rows, int T, int Hd,
                                                    int ldw, int j_begin, int ncols) {
  using TL = DgradTile;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the column tiles of one row tile share h and gh: same XCD
  const XcdSlot slot = xcd_slot((ncols + TL::BN - 1) / TL::BN);
  const int n0 = slot.member * TL::BN;
  const int m0 = slot.group * TL::BM;
  if (m0 >= rows) return;

  const GzRowsLoad<VEC> la{{h, gh, gh_n, T, Hd}, (long)m0, rows};
  // B[k = hidden index][j]: W_h rows are contiguous along j -> NC layout
  auto lb = plain_load([=](int r, int k) -> f32x4 {
    const int j = n0 + r;
    return load4<VEC>(W, (long)k * ldw + j_begin + j, j, ncols, k < Hd);
  });

  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tile_zero<TL>(acc);
  gemm_tile<TL>(smem, acc, la, lb, 0, Hd, w);

  float wcol[TL::TN], grow[TL::TM][16];
  preload_cols<TL>(wcol, w, [&](int nn) { return load_or_zero(w_u + n0 + nn, ga && n0 + nn < ncols); });
  preload_rows<TL>(grow, w, [&](int mm) { return load_or_zero(ga + m0 + mm, ga && m0 + mm < rows); });
  tile_epilogue_ix<TL>(acc, w, [&](int tm, int tn, int i, int mm, int nn, float v) {
    const long row = m0 + mm;
    const int j = n0 + nn;
    if (row < rows && j < ncols) out[row * ncols + j] = v + grow[tm][i] * wcol[tn];
  });
}

// ---------------------------------------------------------------------------
// gW_h[n][j] = sum_rows gz[row][n] * [m;x][row][j];  gb_h[n] = sum_rows gz[row][n]
// The row range is split over gridDim.z workgroups; each writes one slab
// [Hd*(M+D) + Hd] that k_reduce_slabs sums in a fixed order.
// ---------------------------------------------------------------------------
using WgradTile = Tile<128, 192, 32, 2, 2, NC, NC>;

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_wgrad(const float* __restrict__ h,
                                                    const float* __restrict__ gh,
                                                    const float* __restrict__ gh_n,
                                                    const float* __restrict__ m, const float* __restrict__ x,
                                                    float* __restrict__ slabs, int rows, int rows_per_split,
                                                    int T, int M, int D, int Hd) {
  using TL = WgradTile;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int MD = M + D;
  // all output tiles of one row range share its slices of gz and [m;x]: same XCD
  const int tiles_j = (MD + TL::BN - 1) / TL::BN, tiles_n = (Hd + TL::BM - 1) / TL::BM;
  const XcdSlot slot = xcd_slot(tiles_j * tiles_n);
  const int tile_j = slot.member % tiles_j;
  const int j0 = tile_j * TL::BN;
  const int n0 = (slot.member / tiles_j) * TL::BM;
  const int kbeg = slot.group * rows_per_split;
  if (kbeg >= rows) return;
  const int kend = min(rows, kbeg + rows_per_split);
  const long bias_off = ((long)Hd * MD + 3) & ~3L;  // segments start on multiples of four floats
  float* slab = slabs + slot.group * (bias_off + ((Hd + 3) & ~3));

  const GzColsLoad<VEC> la{{h, gh, gh_n, T, Hd}, n0, kend};
  auto lb = plain_load([=](int r, int k) -> f32x4 {
    const long row = k;
    const int j = j0 + r;
    if (VEC) {
      const float* p = j < M ? m + row * M + j : x + row * D + (j - M);
      p = (k < kend && j < MD) ? p : reinterpret_cast<const float*>(kZeroBytes);
      return *reinterpret_cast<const f32x4*>(p);
    } else {
      if (k >= kend) return zero4();
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jj = j + q;
        v[q] = jj < M ? m[row * M + jj] : (jj < MD ? x[row * D + (jj - M)] : 0.f);
      }
      return v;
    }
  });

  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tile_zero<TL>(acc);

  // same loop as gemm_tile, with the bias-gradient column sums taken from the
  // staged A tile (LDS image [k][n]) by the workgroups of the first j tile
  float* As = smem;
  float* Bs = smem + TL::A_FLOATS;
  typename decltype(la)::raw_t ra[TL::A_VECS];
  typename decltype(lb)::raw_t rb[TL::B_VECS];
  float colsum = 0.f;
  const bool do_bias = tile_j == 0 && threadIdx.x < TL::BM;
  if (kbeg < kend) {
    tile_fetch<TL::BM, TL::BK, TL::ALAY>(ra, la, kbeg);
    tile_fetch<TL::BN, TL::BK, TL::BLAY>(rb, lb, kbeg);
  }
  for (int k0 = kbeg; k0 < kend; k0 += TL::BK) {
    tile_store<TL::BM, TL::BK, TL::ALAY>(As, ra, la);
    tile_store<TL::BN, TL::BK, TL::BLAY>(Bs, rb, lb);
    __syncthreads();
    if (k0 + TL::BK < kend) {
      tile_fetch<TL::BM, TL::BK, TL::ALAY>(ra, la, k0 + TL::BK);
      tile_fetch<TL::BN, TL::BK, TL::BLAY>(rb, lb, k0 + TL::BK);
    }
    if (do_bias) {
#pragma unroll 8
      for (int k = 0; k < TL::BK; ++k) colsum += As[k * TL::BM + threadIdx.x];
    }
    tile_compute<TL>(As, Bs, acc, w);
    __syncthreads();
  }

  tile_epilogue<TL>(acc, w, [&](int mm, int nn, float v) {
    const int n = n0 + mm, j = j0 + nn;
    if (n < Hd && j < MD) slab[(long)n * MD + j] = v;
  });
  if (do_bias && n0 + (int)threadIdx.x < Hd) slab[bias_off + n0 + threadIdx.x] = colsum;
}

// ---------------------------------------------------------------------------
// gu[b,s] = sum_{t>=s} sum_k gm[b,t,k] H[k,t-s]
// Per (batch, t tile): P[i][tau] = sum_k gm[t0+i][k] H[k][tau] on MFMA, then
// every P element is added to the diagonal s = t0 + i - tau it belongs to
// (through LDS, one thread per diagonal, in a fixed order).  Each workgroup writes its partial
// gu for s < t0+BM; k_gu_reduce sums the partials in a fixed order.
// ---------------------------------------------------------------------------
using ConvBwdTile = Tile<64, 128, 32, 2, 2, KC, NC>;

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_conv_bwd(const float* __restrict__ gm,
                                                       const float* __restrict__ H,
                                                       float* __restrict__ gu_part, int T, int M, int ldH) {
  using TL = ConvBwdTile;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gu_acc = smem + TL::SMEM_FLOATS;  // [t0 + BM] diagonals
  const int b = blockIdx.y;
  const int ntt = gridDim.z;
  float* ptile = gu_acc + ntt * TL::BM;
  const int tb = ntt - 1 - blockIdx.z;
  const int t0 = tb * TL::BM;
  const int send = min(T, t0 + TL::BM);  // diagonals 0 .. send-1
  const float* gmb = gm + (long)b * T * M;

  for (int s = threadIdx.x; s < send; s += kThreads) gu_acc[s] = 0.f;
  // (the barriers inside gemm_tile order this against the diagonal sums)

  auto la = plain_load([=](int r, int k) -> f32x4 {
    const int t = t0 + r;
    return load4<VEC>(gmb, (long)t * M + k, k, M, t < T);
  });

  WavePos<TL> w;
  for (int tau0 = 0; tau0 < send; tau0 += TL::BN) {
    auto lb = plain_load([=](int r, int k) -> f32x4 {
      const int tau = tau0 + r;
      return load4<VEC>(H, (long)k * ldH + tau, tau, T, k < M);
    });
    f32x16 acc[TL::TM][TL::TN];
    tile_zero<TL>(acc);
    gemm_tile<TL>(smem, acc, la, lb, 0, M, w);
    add_tile_diagonals<TL>(acc, w, ptile, gu_acc, t0 - tau0, T - 1 - t0);
  }
  float* out = gu_part + ((long)b * ntt + tb) * T;
  for (int s = threadIdx.x; s < send; s += kThreads) out[s] = gu_acc[s];
}

// gu[b,t] = sum over the t tiles tb >= t / tile_t of their partial slab;
// optional ReLU mask of the encoder: ga = gu * (u > 0)
__global__ __launch_bounds__(kThreads) void k_gu_reduce(const float* __restrict__ gu_part,
                                                        const float* __restrict__ u,
                                                        float* __restrict__ out, int B, int T, int ntt,
                                                        int tile_t) {
  const long n = (long)B * T;
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
    const int b = (int)(i / T), t = (int)(i % T);
    float s = 0.f;
    for (int tb = t / tile_t; tb < ntt; ++tb) s += gu_part[((long)b * ntt + tb) * T + t];
    if (u) s = u[i] > 0.f ? s : 0.f;
    out[i] = s;
  }
}

// ---------------------------------------------------------------------------
// partial gw_u = sum_rows ga[row] x[row,:], partial gb_u = sum_rows ga[row]
// One wave per row; rows with ga == 0 (ReLU-dead frames, about half of them)
// do not touch x.  Slab per workgroup: [D rounded up to 4 | gb_u + 3 unused].
// ---------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_encode_bwd(const float* __restrict__ ga,
                                                         const float* __restrict__ x, float* __restrict__ part,
                                                         int rows, int D, int rows_per_block) {
  __shared__ float red[4][kThreads];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  const int Dp = (D + 3) & ~3;
  float* slab = part + (long)blockIdx.x * (Dp + 4);

  for (int c0 = 0; c0 < D; c0 += 256) {
    const int c = c0 + lane * 4;
    f32x4 acc = zero4();
    float gsum = 0.f;
    for (int row = r0 + wave; row < r1; row += 4) {
      const float g = ga[row];
      gsum += g;
      if (g != 0.f && c < D) acc += g * load4<VEC>(x, (long)row * D + c, c, D);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][lane * 4 + j] = acc[j];
    __syncthreads();
    const int cc = c0 + threadIdx.x;
    if (cc < D) slab[cc] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (c0 == 0) {
      __syncthreads();
      if (lane == 0) red[wave][0] = gsum;
      __syncthreads();
      if (threadIdx.x == 0) slab[Dp] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    }
  }
}

// Sum of S slabs (and of the partial copies inside a slab), added in a fixed order
// (deterministic).  A thread owns four consecutive outputs and keeps eight slabs in flight;
// in a segment with several copies per slab, eight neighbouring lanes share the four outputs,
// each summing the copies r, r + 8, ... and a fixed xor tree combining them.
struct SlabBounds {
  long tb[5];    // boundaries of the segments in thread slots (multiples of 8)
  long so[4];    // offset of every segment inside a slab
  long pn[4];    // padded copy length
  int reps[4];
  int lanes[4];  // 1 or 8 lanes per four outputs
};
__device__ __forceinline__ int slab_segment(const long* bounds, long i) {
  return (i >= bounds[1]) + (i >= bounds[2]) + (i >= bounds[3]);
}
template <class T>
__device__ __forceinline__ T pick4(const T* a, int q) {
  return q == 0 ? a[0] : (q == 1 ? a[1] : (q == 2 ? a[2] : a[3]));
}
__device__ __forceinline__ void slab_store4(const SlabOutputs& o, int seg, long local, const float (&v)[4]) {
  float* dst = pick4(o.out, seg);
  if (!dst) return;  // a segment whose sum nobody wants
  const long nn = pick4(o.n, seg);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (local + j < nn) dst[local + j] = v[j];
}

__device__ __forceinline__ void reduce_slabs_body(const float* __restrict__ in, int S, long stride, const SlabOutputs& o,
                                                  const SlabBounds& sb, int block, int nblocks) {
  // tb[4] and the grid stride are multiples of 8: the eight lanes of a group leave the loop together
  for (long t = block * (long)kThreads + threadIdx.x; t < sb.tb[4]; t += (long)nblocks * kThreads) {
    const int seg = slab_segment(sb.tb, t);
    const long lt = t - pick4(sb.tb, seg);
    const int L = pick4(sb.lanes, seg);
    const long local = (L == 8 ? lt >> 3 : lt) * 4;
    const int r0 = L == 8 ? (int)(lt & 7) : 0;
    const long pn = pick4(sb.pn, seg);
    const int reps = pick4(sb.reps, seg);
    f32x4 s = f32x4(0.f);
    if (local < pn) {
      const float* src = in + pick4(sb.so, seg) + local + (long)r0 * pn;
      for (int r = r0; r < reps; r += L, src += (long)L * pn) {
        int k = 0;
        for (; k + 8 <= S; k += 8) {
          f32x4 v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(src + (k + q) * stride);
#pragma unroll
          for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; k < S; ++k) s += *reinterpret_cast<const f32x4*>(src + k * stride);
      }
    }
    float o4[4] = {s[0], s[1], s[2], s[3]};
#pragma unroll
    for (int m = 1; m < 8; m <<= 1)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float other = __shfl_xor(o4[j], m);
        o4[j] += L == 8 ? other : 0.f;
      }
    if (r0 == 0 && local < pn) slab_store4(o, seg, local, o4);
  }
}

__global__ __launch_bounds__(kThreads) void k_reduce_slabs(const float* __restrict__ in, int S, long stride,
                                                           SlabOutputs o, SlabBounds sb) {
  reduce_slabs_body(in, S, stride, o, sb, (int)blockIdx.x, (int)gridDim.x);
}

// Several independent slab sums in one launch (the weight gradients of a training step's Linear layers, summed
// together at the end of the backward instead of one launch each).  The table travels in the kernel's arguments and
// is read there with scalar loads; a workgroup belongs to the item whose block range holds it.
constexpr int kReduceManyMax = 12;
struct ReduceBatch {
  const float* in[kReduceManyMax];
  long stride[kReduceManyMax];
  SlabOutputs o[kReduceManyMax];
  SlabBounds sb[kReduceManyMax];
  int S[kReduceManyMax];
  int first_block[kReduceManyMax + 1];
  int count;
};
static_assert(sizeof(ReduceBatch) <= 4096, "fits the kernel-argument segment");
__global__ __launch_bounds__(kThreads) void k_reduce_slabs_many(ReduceBatch) {
  typedef const __attribute__((address_space(4))) ReduceBatch* table_ptr;
  table_ptr b = (table_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  int it = 0;
  for (int i = 1; i < b->count; ++i) it += (int)blockIdx.x >= b->first_block[i] ? 1 : 0;  // uniform
  SlabOutputs o;
  SlabBounds sb;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    o.out[q] = b->o[it].out[q];
    o.n[q] = b->o[it].n[q];
    o.reps[q] = b->o[it].reps[q];
    sb.so[q] = b->sb[it].so[q];
    sb.pn[q] = b->sb[it].pn[q];
    sb.reps[q] = b->sb[it].reps[q];
    sb.lanes[q] = b->sb[it].lanes[q];
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) sb.tb[q] = b->sb[it].tb[q];
  const int first = b->first_block[it];
  reduce_slabs_body(b->in[it], b->S[it], b->stride[it], o, sb, (int)blockIdx.x - first, b->first_block[it + 1] - first);
}

// Same sum for FEW outputs and MANY slabs (the encoder's weight gradient: D + 1 outputs, one
// slab per 64 frames): a workgroup owns 32 slab positions, its 8 thread rows split the slabs;
// partial sums are combined through LDS in a fixed order.
__global__ __launch_bounds__(kThreads) void k_reduce_slabs_tall(const float* __restrict__ in, int S, long stride,
                                                                SlabOutputs o, SlabBounds sb) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
  const long i = blockIdx.x * 32L + col;  // one copy per segment here: slab offset == output index
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < stride) {
    int k = part;
    for (; k + 24 < S; k += 32) {
      s0 += in[(long)k * stride + i];
      s1 += in[(long)(k + 8) * stride + i];
      s2 += in[(long)(k + 16) * stride + i];
      s3 += in[(long)(k + 24) * stride + i];
    }
    for (; k < S; k += 8) s0 += in[(long)k * stride + i];
  }
  red[part][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0 && (col & 3) == 0 && i < stride) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += red[q][col + j];
      v[j] = s;
    }
    const int seg = slab_segment(sb.so, i);
    slab_store4(o, seg, i - pick4(sb.so, seg), v);
  }
}

// out[i] = sum_p part[p * n + i], optionally masked by u[i] > 0 (ReLU of the encoder)
__global__ __launch_bounds__(kThreads) void k_sum_slabs_masked(const float* __restrict__ part,
                                                               const float* __restrict__ u, float* __restrict__ out,
                                                               long n, int P) {
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 4 <= P; p += 4) {
      s0 += part[(long)p * n + i];
      s1 += part[(long)(p + 1) * n + i];
      s2 += part[(long)(p + 2) * n + i];
      s3 += part[(long)(p + 3) * n + i];
    }
    for (; p < P; ++p) s0 += part[(long)p * n + i];
    float s = (s0 + s1) + (s2 + s3);
    if (u) s = u[i] > 0.f ? s : 0.f;
    out[i] = s;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

constexpr int kEncRowsPerBlock = 64;

static int wgrad_splits(long rows, int Hd, int MD, int* rows_per_split) {
  const int tiles = cdiv(Hd, WgradTile::BM) * cdiv(MD, WgradTile::BN);
  // whole row ranges per XCD (32 CUs each), one workgroup per CU at most
  int S = 8 * std::max(1, 32 / tiles);
  int rps = cdiv(rows, S);
  rps = cdiv(rps, WgradTile::BK) * WgradTile::BK;
  S = cdiv(rows, rps);
  *rows_per_split = rps;
  return S;
}

struct BwdPlan {
  int S, rps, ntt, enc_blocks;
  size_t off_wslab, off_gupart, off_ga, off_encpart, total;
};

static BwdPlan plan_backward(const lmufft_shape& s) {
  BwdPlan p;
  const long rows = (long)s.B * s.T;
  const int MD = s.M + s.D;
  p.S = wgrad_splits(rows, s.Hd, MD, &p.rps);
  p.ntt = cdiv(s.T, ConvBwdTile::BM);
  p.enc_blocks = cdiv(rows, kEncRowsPerBlock);
  size_t off = 0;
  p.off_wslab = off;
  off += align256(sizeof(float) * (size_t)p.S * (slab_seg_pad((long)s.Hd * MD) + slab_seg_pad(s.Hd)));
  p.off_gupart = off;
  off += align256(std::max(sizeof(float) * (size_t)s.B * p.ntt * s.T, fft_conv_workspace_bytes(s)));
  p.off_ga = off;
  off += align256(sizeof(float) * (size_t)rows);
  p.off_encpart = off;
  off += align256(sizeof(float) * (size_t)p.enc_blocks * (slab_seg_pad(s.D) + 4));
  p.total = off;
  return p;
}

static int check_shape(const lmufft_shape* s) {
  if (!s) return set_error("shape is NULL");
  if (s->B <= 0 || s->T <= 0 || s->D <= 0 || s->M <= 0 || s->Hd <= 0)
    return set_error("shape: B, T, D, M, Hd must be positive (got B=%d T=%d D=%d M=%d Hd=%d)", s->B, s->T,
                     s->D, s->M, s->Hd);
  if (s->T > s->ldH)
    return set_error("sequence length %d exceeds the impulse response length (seq_len) %d", s->T, s->ldH);
  if ((long)s->B * s->T >= (1L << 31) / 4) return set_error("B*T too large");
  return 0;
}

#define PROF(name) ProfScope prof_scope_(name, st)
#define LAUNCH_CHECK(name)                                                            \
  do {                                                                                \
    hipError_t e_ = hipGetLastError();                                                \
    if (e_ != hipSuccess) return set_error("launch of %s failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

static SlabBounds slab_bounds(const SlabOutputs& o, bool* single_out) {
  SlabBounds sb;
  bool single = true;
  sb.tb[0] = 0;
  for (int q = 0; q < 4; ++q) {
    sb.pn[q] = slab_seg_pad(o.n[q]);
    sb.reps[q] = slab_reps(o, q);
    sb.lanes[q] = sb.reps[q] > 1 ? 8 : 1;
    sb.so[q] = slab_seg_offset(o, q);
    const long slots = sb.pn[q] / 4 * sb.lanes[q];
    sb.tb[q + 1] = sb.tb[q] + ((slots + 7) & ~7L);
    single = single && sb.reps[q] == 1;
  }
  if (single_out) *single_out = single;
  return sb;
}

int launch_reduce_slabs_many(const float* const* in, const int* S, const SlabOutputs* o, int count, hipStream_t st) {
  for (int i0 = 0; i0 < count; i0 += kReduceManyMax) {
    ReduceBatch b;
    b.count = std::min(count - i0, kReduceManyMax);
    b.first_block[0] = 0;
    for (int i = 0; i < b.count; ++i) {
      if (!aligned16(in[i0 + i])) return set_error("workspace must be 16-byte aligned");
      b.in[i] = in[i0 + i];
      b.S[i] = S[i0 + i];
      b.o[i] = o[i0 + i];
      b.stride[i] = slab_stride(o[i0 + i]);
      b.sb[i] = slab_bounds(o[i0 + i], nullptr);
      b.first_block[i + 1] = b.first_block[i] + (int)min(cdiv(b.sb[i].tb[4], kThreads), 1024L);
    }
    {
      PROF("k_reduce_slabs");
      hipLaunchKernelGGL(k_reduce_slabs_many, dim3(b.first_block[b.count]), dim3(kThreads), 0, st, b);
    }
    LAUNCH_CHECK("k_reduce_slabs_many");
  }
  return 0;
}

int launch_reduce_slabs(const float* in, int S, const SlabOutputs& o, hipStream_t st) {
  if (!aligned16(in)) return set_error("workspace must be 16-byte aligned");
  const long stride = slab_stride(o);
  bool single = true;
  const SlabBounds sb = slab_bounds(o, &single);
  if (single && stride <= 4096 && S >= 64) {  // few outputs, many slabs
    {
      PROF("k_reduce_slabs_tall");
      hipLaunchKernelGGL(k_reduce_slabs_tall, dim3(cdiv(stride, 32)), dim3(kThreads), 0, st, in, S, stride, o, sb);
    }
    LAUNCH_CHECK("k_reduce_slabs_tall");
    return 0;
  }
  const dim3 grid(min(cdiv(sb.tb[4], kThreads), 4096));
  {
    PROF("k_reduce_slabs");
    hipLaunchKernelGGL(k_reduce_slabs, grid, dim3(kThreads), 0, st, in, S, stride, o, sb);
  }
  LAUNCH_CHECK("k_reduce_slabs");
  return 0;
}

int launch_gu_reduce(const float* gu_part, const float* u, float* out, int B, int T, int ntt, int tile_t,
                     hipStream_t st) {
  const long n = (long)B * T;
  {
    PROF("k_gu_reduce");
    hipLaunchKernelGGL(k_gu_reduce, dim3(min(cdiv(n, kThreads), 2048)), dim3(kThreads), 0, st, gu_part, u, out, B,
                       T, ntt, tile_t);
  }
  LAUNCH_CHECK("k_gu_reduce");
  return 0;
}

int launch_sum_slabs_masked(const float* part, const float* u, float* out, long n, int P, hipStream_t st) {
  {
    PROF("k_sum_slabs_masked");
    hipLaunchKernelGGL(k_sum_slabs_masked, dim3(min(cdiv(n, kThreads), 2048)), dim3(kThreads), 0, st, part, u, out,
                       n, P);
  }
  LAUNCH_CHECK("k_sum_slabs_masked");
  return 0;
}

// FFT route when the caller supplied prepared spectra and the kernels can take T; Toeplitz otherwise
static bool use_fft(const lmufft_shape& s, int conv_algo, const void* fftprep) {
  if (conv_algo == LMUFFT_CONV_TOEPLITZ) return false;
  if (conv_algo == LMUFFT_CONV_FFT) return true;
  return fftprep != nullptr && fft_route_available(s.T);
}
static int check_algo(const lmufft_shape& s, int conv_algo, const void* fftprep) {
  if (conv_algo != LMUFFT_CONV_AUTO && conv_algo != LMUFFT_CONV_TOEPLITZ && conv_algo != LMUFFT_CONV_FFT)
    return set_error("unknown convolution algorithm %d", conv_algo);
  if (conv_algo == LMUFFT_CONV_FFT) {
    if (!fftprep) return set_error("the FFT convolution needs the prepared spectra (lmufft_fftprep_f32)");
    if (!fft_route_available(s.T)) return set_error("the FFT convolution does not support T = %d", s.T);
  }
  return 0;
}

static int conv_forward_toeplitz(const lmufft_shape& s, const float* u, const float* H, float* m,
                                 hipStream_t st) {
  using TL = ConvFwdTile;
  const dim3 grid(cdiv(s.M, TL::BN), s.B, cdiv(s.T, TL::BM));
  const size_t lds = TL::SMEM_FLOATS * sizeof(float);
  const bool vec = (s.ldH % 4 == 0) && (s.T % 4 == 0) && aligned16(H);
  {
  PROF("k_conv_fwd");
  if (vec)
    hipLaunchKernelGGL(k_conv_fwd<true>, grid, dim3(kThreads), lds, st, u, H, m, s.T, s.M, s.ldH);
  else
    hipLaunchKernelGGL(k_conv_fwd<false>, grid, dim3(kThreads), lds, st, u, H, m, s.T, s.M, s.ldH);
  }
  LAUNCH_CHECK("k_conv_fwd");
  return 0;
}

// gu_part: [B][ntt][T] scratch; out = gu (u == NULL) or ga = gu*(u>0)
static int conv_backward_toeplitz(const lmufft_shape& s, const float* gm, const float* H, const float* u,
                                  float* gu_part, float* out, hipStream_t st) {
  using TL = ConvBwdTile;
  const int ntt = cdiv(s.T, TL::BM);
  const dim3 grid(1, s.B, ntt);
  const size_t lds = (TL::SMEM_FLOATS + (size_t)ntt * TL::BM + (size_t)TL::BM * (TL::BN + 1)) * sizeof(float);
  if (lds > 160 * 1024) return set_error("sequence too long for the Toeplitz backward kernel (T=%d)", s.T);
  const bool vec = (s.ldH % 4 == 0) && (s.T % 4 == 0) && (s.M % 4 == 0) && aligned16(H) && aligned16(gm);
  if (lds > 64 * 1024) {
    const void* fn = vec ? (const void*)k_conv_bwd<true> : (const void*)k_conv_bwd<false>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return set_error("cannot reserve %zu bytes of LDS for k_conv_bwd", lds);
  }
  {
    PROF("k_conv_bwd");
    if (vec)
      hipLaunchKernelGGL(k_conv_bwd<true>, grid, dim3(kThreads), lds, st, gm, H, gu_part, s.T, s.M, s.ldH);
    else
      hipLaunchKernelGGL(k_conv_bwd<false>, grid, dim3(kThreads), lds, st, gm, H, gu_part, s.T, s.M, s.ldH);
  }
  LAUNCH_CHECK("k_conv_bwd");
  return launch_gu_reduce(gu_part, u, out, s.B, s.T, ntt, TL::BM, st);
}

}  // namespace lmufft

using namespace lmufft;

extern "C" {

int lmufft_abi_version(void) { return LMUFFT_ABI_VERSION; }

const char* lmufft_last_error(void) { return last_error_buffer(); }

size_t lmufft_forward_workspace_bytes(const lmufft_shape* s, int dtype) {
  if (dtype == LMUFFT_BF16) return check_shape(s) ? 0 : forward_workspace_bf16(*s);
  if (!s || s->B <= 0 || s->T <= 0 || s->M <= 0) return 0;
  // the FFT route keeps the spectra of u here; never return 0 so callers can always allocate
  return std::max((size_t)256, align256(fft_conv_workspace_bytes(*s)));
}

size_t lmufft_backward_workspace_bytes(const lmufft_shape* s, int dtype) {
  if (check_shape(s)) return 0;
  if (dtype == LMUFFT_BF16) return backward_workspace_bf16(*s);
  return plan_backward(*s).total;
}

size_t lmufft_conv_workspace_bytes(const lmufft_shape* s, int dtype, int conv_algo) {
  (void)conv_algo;
  if (!s || s->B <= 0 || s->T <= 0) return 0;
  if (dtype == LMUFFT_BF16) return s->M > 0 ? conv_workspace_bf16(*s) : 0;
  const int ntt = cdiv(s->T, ConvBwdTile::BM);
  return align256(std::max(sizeof(float) * (size_t)s->B * ntt * s->T, fft_conv_workspace_bytes(*s)));
}

size_t lmufft_fftprep_bytes(const lmufft_shape* s) {
  if (!s || s->M <= 0 || s->T <= 0) return 0;
  return fftprep_bytes(s->M, s->T);
}

int lmufft_fftprep_f32(const lmufft_shape* s, const float* H, void* fftprep, void* stream) {
  if (!s || s->M <= 0 || s->T <= 0 || s->T > s->ldH) return set_error("fftprep: bad shape");
  if (!H || !fftprep) return set_error("fftprep: NULL pointer");
  return fft_prepare(*s, H, fftprep, (hipStream_t)stream);
}

int lmufft_conv_forward_f32(const lmufft_shape* s, const float* u, const float* H, const void* fftprep, float* m,
                            int conv_algo, void* workspace, size_t workspace_bytes, void* stream) {
  if (!s || s->B <= 0 || s->T <= 0 || s->M <= 0) return set_error("conv: bad shape");
  if (s->T > s->ldH) return set_error("sequence length %d exceeds the impulse response length %d", s->T, s->ldH);
  if (!u || !H || !m) return set_error("conv: NULL pointer");
  if (int rc = check_algo(*s, conv_algo, fftprep)) return rc;
  if (use_fft(*s, conv_algo, fftprep)) {
    const size_t need = fft_conv_workspace_bytes(*s);
    if (!workspace || workspace_bytes < need)
      return set_error("conv forward (FFT): workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    return fft_conv_forward(*s, u, fftprep, m, workspace, (hipStream_t)stream);
  }
  return conv_forward_toeplitz(*s, u, H, m, (hipStream_t)stream);
}

int lmufft_conv_backward_f32(const lmufft_shape* s, const float* gm, const float* H, const void* fftprep,
                             float* gu, int conv_algo, void* workspace, size_t workspace_bytes, void* stream) {
  if (!s || s->B <= 0 || s->T <= 0 || s->M <= 0) return set_error("conv: bad shape");
  if (s->T > s->ldH) return set_error("sequence length %d exceeds the impulse response length %d", s->T, s->ldH);
  if (!gm || !H || !gu) return set_error("conv: NULL pointer");
  if (int rc = check_algo(*s, conv_algo, fftprep)) return rc;
  const size_t need = lmufft_conv_workspace_bytes(s, LMUFFT_F32, conv_algo);
  if (!workspace || workspace_bytes < need)
    return set_error("conv backward: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  if (use_fft(*s, conv_algo, fftprep)) {
    return fft_conv_backward(*s, gm, fftprep, nullptr, gu, workspace, st);
  }
  return conv_backward_toeplitz(*s, gm, H, nullptr, (float*)workspace, gu, st);
}

// ---------------------------------------------------------------------------
// Allocation-free variant of lmufft_conv_backward_f32: the partial slabs (Toeplitz route) or
// partial spectra (FFT route) live in a file-scope device array instead of a caller workspace.
// Host code never passes the array's NAME to a kernel or a runtime call -- on the host that symbol
// is a stand-in, not a device address; the address is asked for once per device and kept.
// ---------------------------------------------------------------------------
static __device__ __align__(256) unsigned char g_noalloc_scratch[LMUFFT_NOALLOC_SCRATCH_BYTES];
static constexpr int kNoAllocMaxDevices = 64;
static void* g_noalloc_ptr[kNoAllocMaxDevices];  // per device: address of ITS copy of the array
static std::mutex g_noalloc_mutex;

// Address of the current device's scratch array (looked up on the first call on that device).
static int noalloc_scratch(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return set_error("noalloc: hipGetDevice failed: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= kNoAllocMaxDevices) return set_error("noalloc: device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_noalloc_mutex);
  if (!g_noalloc_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch));
    if (e != hipSuccess || !p)
      return set_error("noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
    g_noalloc_ptr[dev] = p;
  }
  *out = g_noalloc_ptr[dev];
  return 0;
}

int lmufft_conv_backward_noalloc_f32(const lmufft_shape* s, const float* gm, const float* H, const void* fftprep,
                                     float* gu, int conv_algo, void* stream) {
  if (!s || s->B <= 0 || s->T <= 0 || s->M <= 0) return set_error("conv: bad shape");
  if (s->T > s->ldH) return set_error("sequence length %d exceeds the impulse response length %d", s->T, s->ldH);
  if (!gm || !H || !gu) return set_error("conv: NULL pointer");
  if (int rc = check_algo(*s, conv_algo, fftprep)) return rc;
  const size_t need = lmufft_conv_workspace_bytes(s, LMUFFT_F32, conv_algo);
  if (need > (size_t)LMUFFT_NOALLOC_SCRATCH_BYTES)
    return set_error("conv backward (no-alloc): this shape needs %zu bytes of scratch, the built-in buffer holds %zu; "
                     "use lmufft_conv_backward_f32 with a workspace",
                     need, (size_t)LMUFFT_NOALLOC_SCRATCH_BYTES);
  void* scratch = nullptr;
  if (int rc = noalloc_scratch(&scratch)) return rc;
  hipStream_t st = (hipStream_t)stream;
  // Nothing in the scratch has to start at zero: both routes write every partial they later read
  // (k_conv_bwd writes diagonals 0..send-1 of each slab and k_gu_reduce reads only those; the FFT
  // kernels write every partial spectrum of the call's shape), so the previous call's contents
  // are simply overwritten.
  if (use_fft(*s, conv_algo, fftprep)) return fft_conv_backward(*s, gm, fftprep, nullptr, gu, scratch, st);
  return conv_backward_toeplitz(*s, gm, H, nullptr, (float*)scratch, gu, st);
}

int lmufft_forward_f32(const lmufft_shape* s, const float* x, const float* w_u, const float* b_u,
                       const float* W_h, const float* b_h, const float* H, const void* fftprep, float* u, float* m,
                       float* h, float* h_n, int conv_algo, void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace;
  (void)workspace_bytes;
  if (int rc = check_shape(s)) return rc;
  if (!x || !w_u || !b_u || !W_h || !b_h || !H || !u || !m || !h || !h_n)
    return set_error("forward: NULL pointer");
  if (int rc = check_algo(*s, conv_algo, fftprep)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const long rows = (long)s->B * s->T;
  const int MD = s->M + s->D;

  {  // eq. 18
    const bool vec = (s->D % 4 == 0) && aligned16(x) && aligned16(w_u);
    const dim3 grid(min(cdiv(rows, 4), 4096));
    {
    PROF("k_encode_fwd");
    if (vec)
      hipLaunchKernelGGL(k_encode_fwd<true>, grid, dim3(kThreads), 0, st, x, w_u, b_u, u, (int)rows, s->D);
    else
      hipLaunchKernelGGL(k_encode_fwd<false>, grid, dim3(kThreads), 0, st, x, w_u, b_u, u, (int)rows, s->D);
    }
    LAUNCH_CHECK("k_encode_fwd");
  }
  if (use_fft(*s, conv_algo, fftprep)) {  // eq. 26
    const size_t need = fft_conv_workspace_bytes(*s);
    if (!workspace || workspace_bytes < need)
      return set_error("forward (FFT): workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    if (int rc = fft_conv_forward(*s, u, fftprep, m, workspace, st)) return rc;
  } else {
    if (int rc = conv_forward_toeplitz(*s, u, H, m, st)) return rc;
  }
  {  // eq. 20
    using TL = ProjTile;
    const bool vec = (s->M % 4 == 0) && (s->D % 4 == 0) && aligned16(m) && aligned16(x) && aligned16(W_h);
    const dim3 grid(xcd_grid(cdiv(rows, TL::BM), cdiv(s->Hd, TL::BN)));
    const size_t lds = TL::SMEM_FLOATS * sizeof(float);
    {
    PROF("k_proj_fwd");
    if (vec)
      hipLaunchKernelGGL(k_proj_fwd<true>, grid, dim3(kThreads), lds, st, m, x, W_h, b_h, h, h_n, (int)rows,
                         s->T, s->M, s->D, s->Hd);
    else
      hipLaunchKernelGGL(k_proj_fwd<false>, grid, dim3(kThreads), lds, st, m, x, W_h, b_h, h, h_n, (int)rows,
                         s->T, s->M, s->D, s->Hd);
    }
    LAUNCH_CHECK("k_proj_fwd");
    (void)MD;
  }
  return 0;
}

int lmufft_backward_f32(const lmufft_shape* s, const float* x, const float* w_u, const float* W_h,
                        const float* H, const void* fftprep, const float* u, const float* m, const float* h,
                        const float* gh,
                        const float* gh_n, float* gx, float* gw_u, float* gb_u, float* gW_h, float* gb_h,
                        float* gm, int conv_algo, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_shape(s)) return rc;
  if (!x || !w_u || !W_h || !H || !u || !m || !h || !gw_u || !gb_u || !gW_h || !gb_h || !gm)
    return set_error("backward: NULL pointer");
  if (int rc = check_algo(*s, conv_algo, fftprep)) return rc;
  const BwdPlan p = plan_backward(*s);
  if (!workspace || workspace_bytes < p.total)
    return set_error("backward: workspace of %zu bytes needed, %zu given", p.total, workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* wslab = (float*)(ws + p.off_wslab);
  float* gupart = (float*)(ws + p.off_gupart);
  float* ga = (float*)(ws + p.off_ga);
  float* encpart = (float*)(ws + p.off_encpart);
  const long rows = (long)s->B * s->T;
  const int MD = s->M + s->D;
  const bool vec_h = (s->Hd % 4 == 0) && aligned16(h) && (!gh || aligned16(gh)) && (!gh_n || aligned16(gh_n));

  auto dgrad = [&](const char* name, float* out, int j_begin, int ncols, const float* ga_, const float* wu_) -> int {
    using TL = DgradTile;
    const bool vec = vec_h && (MD % 4 == 0) && (j_begin % 4 == 0) && (ncols % 4 == 0) && aligned16(W_h);
    const dim3 grid(xcd_grid(cdiv(rows, TL::BM), cdiv(ncols, TL::BN)));
    const size_t lds = TL::SMEM_FLOATS * sizeof(float);
    {
      PROF(name);
      if (vec)
        hipLaunchKernelGGL(k_dgrad<true>, grid, dim3(kThreads), lds, st, h, gh, gh_n, W_h, out, ga_, wu_,
                           (int)rows, s->T, s->Hd, MD, j_begin, ncols);
      else
        hipLaunchKernelGGL(k_dgrad<false>, grid, dim3(kThreads), lds, st, h, gh, gh_n, W_h, out, ga_, wu_,
                           (int)rows, s->T, s->Hd, MD, j_begin, ncols);
    }
    LAUNCH_CHECK(name);
    return 0;
  };

  // gm = gz W_h[:, :M]
  if (int rc = dgrad("k_dgrad_m", gm, 0, s->M, nullptr, nullptr)) return rc;
  // ga = (adjoint conv of gm) * (u > 0)
  if (use_fft(*s, conv_algo, fftprep)) {
    if (int rc = fft_conv_backward(*s, gm, fftprep, u, ga, gupart, st)) return rc;
  } else {
    if (int rc = conv_backward_toeplitz(*s, gm, H, u, gupart, ga, st)) return rc;
  }
  // gx = gz W_h[:, M:] + ga w_u
  if (gx)
    if (int rc = dgrad("k_dgrad_x", gx, s->M, s->D, ga, w_u)) return rc;
  {  // gW_h, gb_h slabs
    using TL = WgradTile;
    const bool vec = vec_h && (s->M % 4 == 0) && (s->D % 4 == 0) && aligned16(m) && aligned16(x);
    const dim3 grid(xcd_grid(p.S, cdiv(MD, TL::BN) * cdiv(s->Hd, TL::BM)));
    const size_t lds = TL::SMEM_FLOATS * sizeof(float);
    {
      PROF("k_wgrad");
      if (vec)
        hipLaunchKernelGGL(k_wgrad<true>, grid, dim3(kThreads), lds, st, h, gh, gh_n, m, x, wslab, (int)rows,
                           p.rps, s->T, s->M, s->D, s->Hd);
      else
        hipLaunchKernelGGL(k_wgrad<false>, grid, dim3(kThreads), lds, st, h, gh, gh_n, m, x, wslab, (int)rows,
                           p.rps, s->T, s->M, s->D, s->Hd);
    }
    LAUNCH_CHECK("k_wgrad");
    const long n0 = (long)s->Hd * MD;
    if (int rc = launch_reduce_slabs(wslab, p.S, SlabOutputs{{gW_h, gb_h, nullptr, nullptr}, {n0, s->Hd, 0, 0}, {1, 1, 1, 1}}, st))
      return rc;
  }
  {  // gw_u, gb_u
    const bool vec = (s->D % 4 == 0) && aligned16(x);
    const dim3 grid(p.enc_blocks);
    {
      PROF("k_encode_bwd");
      if (vec)
        hipLaunchKernelGGL(k_encode_bwd<true>, grid, dim3(kThreads), 0, st, ga, x, encpart, (int)rows, s->D,
                           kEncRowsPerBlock);
      else
        hipLaunchKernelGGL(k_encode_bwd<false>, grid, dim3(kThreads), 0, st, ga, x, encpart, (int)rows, s->D,
                           kEncRowsPerBlock);
    }
    LAUNCH_CHECK("k_encode_bwd");
    if (int rc = launch_reduce_slabs(encpart, p.enc_blocks, SlabOutputs{{gw_u, gb_u, nullptr, nullptr}, {s->D, 1, 0, 0}, {1, 1, 1, 1}},
                                     st))
      return rc;
  }
  return 0;
}

}  // extern "C"

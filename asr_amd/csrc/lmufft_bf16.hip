// bf16 kernels of the LMUFFT layer for gfx950 (MI355X) and their C ABI.
//
// Activations (x, m, h, gh, gx, gm) are bf16 in HBM; every sum is accumulated
// in fp32 (v_mfma_f32_32x32x16_bf16 or fp32 VALU); parameters and their
// gradients stay fp32 ("master weights"), u and gu stay fp32.  bf16 copies of
// W_h (both orientations) are made per call, bf16 copies of H once per module
// (lmufft_hprep_bf16).
//
// Reference arithmetic: src/lmu.py:297-320; the reference itself cannot run
// this path in bf16 on a GPU (torch.fft rejects bf16), so parity is defined
// against the fp32 reference on bf16-rounded inputs, rtol 2e-2.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "gemm_core_bf16.h"
#include "gemm_glds_bf16.h"
#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
static inline int cdivb(long a, long b) { return (int)((a + b - 1) / b); }
static inline int up8(int v) { return (v + 7) & ~7; }
static inline size_t align256b(size_t v) { return (v + 255) & ~size_t(255); }
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#ifdef LMUFFT_STAMPS
// Debug build only (-DLMUFFT_STAMPS=<kernel id>: 1 weight gradient, 2 forward convolution,
// 3 adjoint convolution, 4 projection, 5 encoder gradient from gz):
// per-workgroup timestamps (100 MHz wall clock in slots 0..3, shader-clock sums in slots 4..7),
// read back through lmufft_debug_stamps().
__device__ long long g_stamps[8 * 4096];
__device__ __forceinline__ int stamp_block() {
  return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
}
#define LMUFFT_STAMP(id, i) \
  if (LMUFFT_STAMPS == (id) && threadIdx.x == 0 && stamp_block() < 4096) g_stamps[stamp_block() * 8 + (i)] = wall_clock64()
#define LMUFFT_STAMP_ADD(id, i, v) \
  if (LMUFFT_STAMPS == (id) && threadIdx.x == 0 && stamp_block() < 4096) g_stamps[stamp_block() * 8 + (i)] += (v)
#define LMUFFT_CLOCK() clock64()
#else
#define LMUFFT_STAMP(id, i)
#define LMUFFT_STAMP_ADD(id, i, v)
#define LMUFFT_CLOCK() 0
#endif

// ---------------------------------------------------------------------------
// operand preparation
// ---------------------------------------------------------------------------
// Wb[n][k] (ld = ldW, zero padded) and WbT[j][n] (ld = ldWT, zero padded) from fp32 W[Hd][MD]
struct WeightPrep {
  const float* W;  // NULL: nothing to prepare
  bf16_t *Wb, *WbT;
  int Hd, MD, ldW, ldWT;
};
__device__ __forceinline__ void prep_weights_blocks(const WeightPrep& q, int block, int nblocks) {
  const float* __restrict__ W = q.W;
  bf16_t* __restrict__ Wb = q.Wb;
  bf16_t* __restrict__ WbT = q.WbT;
  const int Hd = q.Hd, MD = q.MD, ldW = q.ldW, ldWT = q.ldWT;
  // a thread writes eight consecutive bf16 (one 16-byte store); ldW and ldWT are multiples of 8.
  // Its eight loads are unconditional (load_or_zero), so they are in flight together: as eight
  // branches with a wait each they were the tail of the encoder kernel these blocks ride on.
  // Wb: eight consecutive k of one row.  WbT: eight consecutive n of one column j, with
  // consecutive threads on consecutive j, so that the strided reads of W coalesce across lanes.
  const int g1 = ldW / 8, g2 = ldWT / 8;
  const long n1 = Wb ? (long)Hd * g1 : 0;
  const long n2 = WbT ? (long)MD * g2 : 0;
  for (long i = block * (long)kThreads + threadIdx.x; i < n1 + n2; i += (long)nblocks * kThreads) {
    bf16x8 o;
    if (i < n1) {
      const int n = (int)(i / g1), k0 = (int)(i % g1) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)load_or_zero(W + (long)n * MD + k0 + e, k0 + e < MD);
      *reinterpret_cast<bf16x8*>(Wb + (long)n * ldW + k0) = o;
    } else {
      const long r = i - n1;
      const int j = (int)(r % MD), n0 = (int)(r / MD) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)load_or_zero(W + (long)(n0 + e) * MD + j, n0 + e < Hd);
      *reinterpret_cast<bf16x8*>(WbT + (long)j * ldWT + n0) = o;
    }
  }
}
__global__ __launch_bounds__(kThreads) void k_prep_weights(WeightPrep q) {
  prep_weights_blocks(q, blockIdx.x, gridDim.x);
}

// Hb[k][tau] (ld = ldHb) and HbT[tau][k] (ld = ldHT, ldHb rows) from fp32 H[M][ldH]
__global__ __launch_bounds__(kThreads) void k_prep_H(const float* __restrict__ H, bf16_t* __restrict__ Hb,
                                                     bf16_t* __restrict__ HbT, int M, int ldH, int ldHb, int ldHT) {
  const long n1 = (long)M * ldHb;
  const long n2 = (long)ldHb * ldHT;
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < n1 + n2; i += (long)gridDim.x * kThreads) {
    if (i < n1) {
      const int k = (int)(i / ldHb), t = (int)(i % ldHb);
      Hb[i] = (bf16_t)(t < ldH ? H[(long)k * ldH + t] : 0.f);
    } else {
      const long q = i - n1;
      const int t = (int)(q / ldHT), k = (int)(q % ldHT);
      HbT[q] = (bf16_t)((t < ldH && k < M) ? H[(long)k * ldH + t] : 0.f);
    }
  }
}

// Q^T[tau][n] = sum_k H^T[tau][k] W_h[n][k] (k < M) in bf16, for the backward's Q route (see k_gu_q_bf16): one
// 64 x 64 tile per workgroup on the matrix cores.
using QPrepTile = TileB<64, 64, 64, 2, 2>;
struct QPrep {
  const bf16_t *HbT, *Wb;
  bf16_t* QT;
  int rowsQ, rowsHT, Hd, M, ldHT, ldW, ldQ;
};
constexpr int kQPrepLdsElems = 2 * 64 * (64 + 8);
// the same tile with the whole memory dimension (up to 192) as ONE k-tile: a launch of its own, where the three
// dependent trips to memory of the 64-deep form would be the launch's length
using QPrepWideTile = TileB<64, 64, 192, 2, 2>;
constexpr int kQPrepWideLdsElems = 2 * 64 * (192 + 8);
template <class TL>
__device__ __forceinline__ void qprep_block(const QPrep& q, int block, bf16_t* qsmem) {
  const int tiles_n = (q.Hd + TL::BN - 1) / TL::BN;
  const int tau0 = (block / tiles_n) * TL::BM, n0 = (block % tiles_n) * TL::BN;
  auto la = plain_load_b([=](int r, int k) -> bf16x8 {
    const int tau = tau0 + r;
    return bload8<true>(q.HbT, (long)tau * q.ldHT + k, k, q.ldHT, tau < q.rowsHT);
  });
  auto lb = plain_load_b([=](int r, int k) -> bf16x8 {
    const int n = n0 + r;
    return bload8<true>(q.Wb, (long)n * q.ldW + k, k, q.M, n < q.Hd);
  });
  typedef LdsOperandKC<TL::BM, TL::BK, decltype(la)> OpA;
  typedef LdsOperandKC<TL::BN, TL::BK, decltype(lb)> OpB;
  static_assert(OpA::ELEMS + OpB::ELEMS == 2 * 64 * (TL::BK + 8), "LDS carve");
  OpA oa(qsmem, la);
  OpB ob(qsmem + OpA::ELEMS, lb);
  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  gemm_loop_kc<TL>(oa, ob, acc, 0, q.M, w);
  tileb_epilogue<TL>(acc, w, [&](int mm, int nn, float v) {
    const int tau = tau0 + mm, n = n0 + nn;
    if (tau < q.rowsQ && n < q.ldQ) q.QT[(long)tau * q.ldQ + n] = (bf16_t)(n < q.Hd ? v : 0.f);
  });
}

// Several layers' Q^T in one launch: workgroups first[i] .. first[i + 1] make q[i]
constexpr int kQBatchMax = 4;
struct QPrepBatch {
  QPrep q[kQBatchMax];
  int first[kQBatchMax + 1];
  int count;
};
template <class TL>
__device__ __forceinline__ void qprep_batch_block(const QPrepBatch& qb, int block, bf16_t* qsmem) {
#pragma unroll
  for (int i = 0; i < kQBatchMax; ++i)  // (unrolled: q[i] is read with a constant index, straight from the arguments)
    if (i < qb.count && block >= qb.first[i] && block < qb.first[i + 1]) qprep_block<TL>(qb.q[i], block - qb.first[i], qsmem);
}

// ---------------------------------------------------------------------------
// u = relu(x . w_u + b_u), x bf16, fp32 accumulate             src/lmu.py:300
// ---------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_encode_fwd_bf16(const bf16_t* __restrict__ x,
                                                              const float* __restrict__ w_u,
                                                              const float* __restrict__ b_u,
                                                              float* __restrict__ u, int rows, int D,
                                                              WeightPrep prep, int prep_blocks) {
  // the first prep_blocks workgroups convert W_h for the projection and the backward (independent
  // work that would otherwise be a launch of its own); the others encode
  if ((int)blockIdx.x < prep_blocks) {
    prep_weights_blocks(prep, blockIdx.x, prep_blocks);
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float bias = b_u[0];
  const int eblock = blockIdx.x - prep_blocks, eblocks = gridDim.x - prep_blocks;
  if (VEC) {
    // sixteen lanes per row, four rows per wave: a row is 16-byte loads of 256 contiguous bytes per
    // pass, a lane has all its loads in flight at once, and the reduction is four shuffles
    // A workgroup takes 32 consecutive rows per step, a lane group rows g and g + 16 of them: the loads of both
    // are issued before either is reduced.
    const int sub = lane & 15;
    for (long row = (long)eblock * 32 + wave * 4 + (lane >> 4); row < rows; row += (long)eblocks * 32) {
      const long row2 = row + 16;
      const bool two = row2 < rows;
      const bf16_t* xr = x + row * D;
      const bf16_t* xr2 = x + (two ? row2 : row) * D;
      float s = 0.f, s2 = 0.f;
      // four 128-column passes at a time with every load issued before the first use (a plain loop over
      // the passes waits for each pass's loads in turn: three dependent trips to memory per row at D = 384);
      // passes beyond D read a line of zeros
      const bf16_t* zx = reinterpret_cast<const bf16_t*>(kZeroBytes);
      const float* zw = reinterpret_cast<const float*>(kZeroBytes);
      for (int c0 = sub * 8; c0 < D; c0 += 512) {
        bf16x8 xv[4], xv2[4];
        f32x4 w0[4], w1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = c0 + 128 * i;
          const bool ok = c < D;
          xv[i] = *reinterpret_cast<const bf16x8*>(ok ? xr + c : zx);
          xv2[i] = *reinterpret_cast<const bf16x8*>(ok ? xr2 + c : zx);
          w0[i] = *reinterpret_cast<const f32x4*>(ok ? w_u + c : zw);
          w1[i] = *reinterpret_cast<const f32x4*>(ok ? w_u + c + 4 : zw);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s += (float)xv[i][j] * w0[i][j] + (float)xv[i][4 + j] * w1[i][j];
            s2 += (float)xv2[i][j] * w0[i][j] + (float)xv2[i][4 + j] * w1[i][j];
          }
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        s2 += __shfl_xor(s2, off);
      }
      if (sub == 0) {
        u[row] = fmaxf(s + bias, 0.f);
        if (two) u[row2] = fmaxf(s2 + bias, 0.f);
      }
    }
  } else {
    for (int row = eblock * 4 + wave; row < rows; row += eblocks * 4) {
      const bf16_t* xr = x + (long)row * D;
      float s = 0.f;
      for (int c = lane; c < D; c += 64) s += (float)xr[c] * w_u[c];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (lane == 0) u[row] = fmaxf(s + bias, 0.f);
    }
  }
}

// ---------------------------------------------------------------------------
// m[b,t,k] = sum_{tau<=t} u[b,t-tau] H[k,tau]                src/lmu.py:303-312
// Toeplitz GEMM.  The A operand A[t][tau] = u[t-tau] is never staged as a
// tile: u (time-reversed, bf16) sits in LDS in 8 copies, copy c shifted by c
// elements and the copies interleaved at 16-byte granularity, so that every
// lane's eight consecutive tau are ONE aligned ds_read_b128 and the 16 lanes
// that the LDS serves together hit 16 different 16-byte slots.
// ---------------------------------------------------------------------------
using ConvFwdTileB = TileB<128, 96, 64, 4, 1>;

template <class TL>
struct ToeplitzOperand {
  const bf16_t* U8;
  __device__ __forceinline__ void fetch(int) {}
  __device__ __forceinline__ void store() {}
  __device__ __forceinline__ bf16x8 frag(int row, int k0, int kl, int hh) const {
    const int d = TL::BM - 1 - row;         // distance of this row from the tile's last frame
    const int c = (8 - (d & 7)) & 7;        // copy whose shift makes the read aligned
    const int Q = ((d + c) >> 3) + ((k0 + kl) >> 3) + hh;
    return *reinterpret_cast<const bf16x8*>(&U8[(Q * 8 + c) * 8]);
  }
};

__global__ __launch_bounds__(kThreads) void k_conv_fwd_bf16(const float* __restrict__ u,
                                                            const bf16_t* __restrict__ Hb,
                                                            bf16_t* __restrict__ m, int T, int M, int ldHb) {
  using TL = ConvFwdTileB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int n0 = blockIdx.x * TL::BN;
  const int b = blockIdx.y;
  const int tb = gridDim.z - 1 - blockIdx.z;
  // The partial tile is the FIRST one (t0 < 0 there, rows before frame 0 are dropped): a tile's
  // k-loop runs over its causal past, so the short loop goes with the tile that has few frames --
  // T = 400: tiles end at 16 / 144 / 272 / 400 (1 + 3 + 5 + 7 k-tiles) instead of 128 / 256 / 384 /
  // 400 (2 + 4 + 6 + 7, the last seven for sixteen frames).
  const int t0 = tb * TL::BM - (TL::BM - T % TL::BM) % TL::BM;
  const int kend = t0 + TL::BM;  // <= T by construction
  const int kend_pad = cdiv_dev(kend, TL::BK) * TL::BK;

  auto lb = plain_load_b([=](int r, int k) -> bf16x8 {
    const int n = n0 + r;
    return bload8<true>(Hb, (long)n * ldHb + k, k, ldHb, n < M);
  });
  typedef LdsOperandKC<TL::BN, TL::BK, decltype(lb)> OpB;
  OpB ob(smem, lb);
  bf16_t* U8 = smem + OpB::ELEMS;

  // time-reversed u around this tile: position p <-> frame s = (t0 + BM - 1) - p.  The frames
  // come into LDS with one coalesced load per thread (one memory round trip), and the eight
  // shifted copies are cut from there.
  const float* ub = u + (long)b * T;
  const int smax = t0 + TL::BM - 1;
  const int nchunks = (kend_pad / 8 + TL::BM / 8) * 8;
  float* uf = reinterpret_cast<float*>(U8 + nchunks * 8);  // u[0 .. kend) of this batch item
  for (int s = threadIdx.x; s < kend; s += kThreads) uf[s] = ub[s];
  LMUFFT_STAMP(2, 0);
  __syncthreads();
  for (int ch = threadIdx.x; ch < nchunks; ch += kThreads) {
    const int Q = ch >> 3, c = ch & 7;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = 8 * Q + j - c;
      const int s = smax - p;
      v[j] = (bf16_t)((p >= 0 && s >= 0 && s < kend) ? uf[s] : 0.f);
    }
    *reinterpret_cast<bf16x8*>(&U8[ch * 8]) = v;
  }
  LMUFFT_STAMP(2, 1);
  // (the first barrier inside gemm_loop_kc orders the fill against the reads)

  ToeplitzOperand<TL> oa{U8};
  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  gemm_loop_kc<TL>(oa, ob, acc, 0, kend, w);

  LMUFFT_STAMP(2, 2);
  bf16_t* mb = m + (long)b * T * M;
  if (M % 8 == 0) {  // rows of m are 16-byte aligned: whole row segments through LDS (free after the loop)
    store_tile_bf16<TL>(acc, w, smem, mb, M, t0, n0, T, M, [](int, int, int, int, int, float v) { return v; });
  } else {
    tileb_epilogue<TL>(acc, w, [&](int mm, int nn, float v) {
      const int t = t0 + mm, n = n0 + nn;
      if (t >= 0 && t < T && n < M) mb[(long)t * M + n] = (bf16_t)v;
    });
  }
  LMUFFT_STAMP(2, 3);
  LMUFFT_STAMP_ADD(2, 4, kend - g_stamps[stamp_block() * 8 + 4]);
}

// ---------------------------------------------------------------------------
// h = relu([m;x] W_h^T + b_h); h_n = h[:, T-1, :]           src/lmu.py:315-318
// ---------------------------------------------------------------------------
using ProjTileB = TileB<128, 128, 64, 2, 2>;

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_proj_fwd_bf16(const bf16_t* __restrict__ m,
                                                            const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ Wb,
                                                            const float* __restrict__ bias,
                                                            bf16_t* __restrict__ h, bf16_t* __restrict__ h_n,
                                                            int rows, int T, int M, int D, int Hd, int ldW) {
  using TL = ProjTileB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int n0 = blockIdx.x * TL::BN;
  const int m0 = blockIdx.y * TL::BM;
  const int K = M + D;

  auto la = plain_load_b([=](int r, int k) -> bf16x8 {
    const long row = m0 + r;
    if (VEC) {  // one 16-byte load from whichever panel holds column k
      const bf16_t* p = k < M ? m + row * M + k : x + row * D + (k - M);
      p = (row < rows && k < K) ? p : reinterpret_cast<const bf16_t*>(kZeroBytes);
      return *reinterpret_cast<const bf16x8*>(p);
    } else {
      if (row >= rows) return bzero8();
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kk = k + j;
        v[j] = kk < M ? m[row * M + kk] : (kk < K ? x[row * D + (kk - M)] : (bf16_t)0.f);
      }
      return v;
    }
  });
  auto lb = plain_load_b([=](int r, int k) -> bf16x8 {
    const int n = n0 + r;
    return bload8<true>(Wb, (long)n * ldW + k, k, ldW, n < Hd);
  });
  typedef LdsOperandKC<TL::BM, TL::BK, decltype(la)> OpA;
  typedef LdsOperandKC<TL::BN, TL::BK, decltype(lb)> OpB;
  OpA oa(smem, la);
  OpB ob(smem + OpA::ELEMS, lb);

  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  gemm_loop_kc<TL>(oa, ob, acc, 0, K, w);

  float bcol[TL::TN];
  preload_cols<TL>(bcol, w, [&](int nn) { return load_or_zero(bias + n0 + nn, n0 + nn < Hd); });
  tile_epilogue_ix<TL>(acc, w, [&](int, int tn, int, int mm, int nn, float v) {
    const long row = m0 + mm;
    const int n = n0 + nn;
    if (row < rows && n < Hd) h[row * Hd + n] = (bf16_t)fmaxf(v + bcol[tn], 0.f);
  });
  // h_n: the last frame of every utterance that ends inside this row tile
  const long tile_end = min((long)rows, (long)m0 + TL::BM);
  for (long rl = ((long)m0 / T) * T + T - 1; rl < tile_end; rl += T) {
    const int ml = (int)(rl - m0);
    tile_epilogue_ix<TL>(acc, w, [&](int, int tn, int, int mm, int nn, float v) {
      const int n = n0 + nn;
      if (mm == ml && n < Hd) h_n[(rl / T) * Hd + n] = (bf16_t)fmaxf(v + bcol[tn], 0.f);
    });
  }
}

// Same GEMM with LDS-DMA staging and 16-byte stores; needs M % 8 == D % 8 == Hd % 8 == 0.
// Tile 128 rows x 384 columns on eight waves: with hidden_size <= 384 every row
// of [m;x] -- the operand that comes from HBM -- passes through a CU's load path
// exactly once, and W_h (L2-resident) once per 128 rows.
using ProjGldsTile = TileB<128, 384, 64, 2, 4>;
struct LnEpilogue {  // y == NULL: none
  const float *gamma, *beta;
  float eps;
  void* y;
  int y_bf16;
};
// Epilogue of the projection: h = relu(acc + b_h) through LDS as whole row segments, the relu bit mask, h_n and the
// optional layer norm.  Tile row 0 is global row m0 (= utterance * T + frame; may lie before the rows the workgroup
// owns); rows in [row_begin, row_end) are stored.  The loop has ended with a barrier: all of LDS is free.
template <class TL>
__device__ __forceinline__ void proj_epilogue(const f32x16 (&acc)[TL::TM][TL::TN], const WavePos<TL>& w,
                                              unsigned char* smem_raw, const float* __restrict__ bias,
                                              bf16_t* __restrict__ h, bf16_t* __restrict__ h_n,
                                              unsigned char* __restrict__ hmask, const LnEpilogue& ln, long m0,
                                              long row_begin, long row_end, int T, int Hd, int n0) {
  float bcol[TL::TN];
  preload_cols<TL>(bcol, w, [&](int nn) { return load_or_zero(bias + n0 + nn, n0 + nn < Hd); });
  bf16_t* tile = reinterpret_cast<bf16_t*>(smem_raw);
  // relu pattern for the backward, one bit per element: hmask[row][Hd / 8] bytes, bit j of byte c = h[row][8 c + j] > 0,
  // written by the lane that stores those 8 elements (the backward then reads 1/16 of the bytes of h to mask the
  // upstream gradient)
  const int CHM = Hd / 8;
  store_tile_bf16<TL>(
      acc, w, tile, h, Hd, m0, n0, row_end, Hd,
      [&](int, int tn, int, int, int, float v) { return fmaxf(v + bcol[tn], 0.f); },
      [&](long row, int col, const bf16x8& v) {
        if (hmask) {
          unsigned bits = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) bits |= ((float)v[j] > 0.f ? 1u : 0u) << j;
          hmask[row * CHM + (col >> 3)] = (unsigned char)bits;
        }
      },
      row_begin);
  // h_n: the last frame of every utterance that ends inside this row tile (tile still in LDS)
  constexpr int LDT = TL::BN + 8;
  for (long rl = (row_begin / T) * T + T - 1; rl < row_end; rl += T) {
    const int ml = (int)(rl - m0);
    for (int c = threadIdx.x; c < TL::BN / 8; c += TL::THREADS) {
      const int col = n0 + c * 8;
      if (col < Hd)
        *reinterpret_cast<bf16x8*>(h_n + (rl / T) * Hd + col) = *reinterpret_cast<const bf16x8*>(&tile[ml * LDT + c * 8]);
    }
  }
  // Layer norm of the rows of h that follows the layer in the encoder (src/models/lmu_encoder.py:274),
  // taken from the bf16 tile still in LDS -- the same bf16-rounded h a separate layer-norm kernel
  // would read back from HBM; the tile spans the whole row (the host checks Hd <= BN).  Four lanes
  // share a row: fp32 mean, centred second moment, two shuffles each.
  if (ln.y) {
    const int rl = threadIdx.x >> 2, q = threadIdx.x & 3;
    static_assert(TL::THREADS == 4 * TL::BM, "four lanes per tile row");
    const long row = m0 + rl;
    const int chunks = Hd / 8;
    float s = 0.f;
    for (int c = q; c < chunks; c += 4) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(&tile[rl * LDT + c * 8]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)v[j];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    const float mean = s / (float)Hd;
    float ss = 0.f;
    for (int c = q; c < chunks; c += 4) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(&tile[rl * LDT + c * 8]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = (float)v[j] - mean;
        ss += d * d;
      }
    }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    const float rstd = rsqrtf(ss / (float)Hd + ln.eps);
    if (row >= row_begin && row < row_end) {
      for (int c = q; c < chunks; c += 4) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(&tile[rl * LDT + c * 8]);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(ln.gamma + c * 8);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(ln.gamma + c * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln.beta + c * 8);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(ln.beta + c * 8 + 4);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o[j] = ((float)v[j] - mean) * rstd * (j < 4 ? g0[j] : g1[j - 4]) + (j < 4 ? b0[j] : b1[j - 4]);
        if (ln.y_bf16) {
          bf16x8 ob;
#pragma unroll
          for (int j = 0; j < 8; ++j) ob[j] = (bf16_t)o[j];
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(ln.y) + row * Hd + c * 8) = ob;
        } else {
          float* yp = reinterpret_cast<float*>(ln.y) + row * Hd + c * 8;
          *reinterpret_cast<f32x4*>(yp) = f32x4{o[0], o[1], o[2], o[3]};
          *reinterpret_cast<f32x4*>(yp + 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
      }
    }
  }
}

constexpr int kProjStages = 2;

__global__ __launch_bounds__(ProjGldsTile::THREADS) void k_proj_fwd_glds_bf16(
    const bf16_t* __restrict__ m, const bf16_t* __restrict__ x, const bf16_t* __restrict__ Wb,
    const float* __restrict__ bias, bf16_t* __restrict__ h, bf16_t* __restrict__ h_n, int rows, int T, int M, int D,
    int Hd, int ldW, LnEpilogue ln, unsigned char* __restrict__ hmask) {
  using TL = ProjGldsTile;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n0 = blockIdx.x * TL::BN;
  const long m0 = (long)blockIdx.y * TL::BM;
  const int K = M + D;

  // A rows are [m[row] ; x[row]]: element k lives at m + row*M + k (k < M) or x + row*D + (k - M)
  typedef GldsOperand<TL::BM, TL::NW, true, TL::BK> OpA;
  typedef GldsOperand<TL::BN, TL::NW, false, TL::BK> OpB;
  OpA oa(
      smem_raw,
      [&](int r, const bf16_t*& q0, const bf16_t*& q1) {
        const long row = m0 + r;
        if (row >= rows) return false;
        q0 = m + row * M;
        q1 = x + row * D - M;
        return true;
      },
      M, K);
  OpB ob(
      smem_raw + kProjStages * OpA::STAGE_BYTES,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        const int n = n0 + r;
        if (n >= Hd) return false;
        q0 = Wb + (long)n * ldW;
        return true;
      },
      ldW, ldW);

  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  LMUFFT_STAMP(4, 0);
  gemm_loop_glds<TL>(oa, ob, acc, 0, K, w);
  LMUFFT_STAMP(4, 1);

  proj_epilogue<TL>(acc, w, smem_raw, bias, h, h_n, hmask, ln, m0, m0, min((long)rows, m0 + TL::BM), T, Hd, n0);
}

// ---------------------------------------------------------------------------
// Convolution and projection in ONE kernel: the m tile of a workgroup's frames is formed in LDS by the
// Toeplitz GEMM and the projection reads it from there as the first k-tiles of its A operand.  m is
// still written (the backward reads it) but never read back, and one launch with its ramp-up and tail
// is gone: at batch 64 x 400 frames the two kernels took 12 + 25 us, this one 2x us.
//
// Workgroup = (utterance b, frame tile j): the ntile = ceil(T / 128) tiles of an utterance have
// FT = ceil(T / ntile) frames each (T = 400: four tiles of 100 -- 256 workgroups at batch 64, one per
// CU); the MFMA tile is the 128 frames ENDING at the tile's last frame, t0 = tend - 128, so rows before
// the tile's first frame are computed and dropped.  512 threads.
//   conv phase   128 frames x 192 memory columns, K = the tend lags that have a causal pair; A is the
//                Toeplitz operand of k_conv_fwd_bf16 (shifted copies of u in LDS), B = Hb rows by LDS-DMA
//                through a four-stage ring with counted waits (three k-tiles in flight; six stages were no
//                faster: with every CU in this phase at once the chip's LDS-DMA ingest is the limit);
//   hand-over    the accumulators go to LDS as bf16 in the layout of the projection's A stages
//                ([128 rows][64 k], 16-byte chunks swizzled) -- three k-tiles -- and from there to HBM;
//   projection   128 x 384, K = M + D in k-tiles of 64: W_h rows by LDS-DMA (two stages); A k-tiles below
//                M are the m tiles, the others are x rows DMA'd into the same three-slot ring (slot
//                kt % 3: its previous tenant was consumed two barriers earlier);
//   epilogue     that of k_proj_fwd_glds_bf16.
// Same k order and MFMA grouping as the two separate kernels: m and h come out bit-identical.
// Needs M % 64 == 0, M <= 192, Hd <= 384, T <= 512, D % 8 == 0, Hd % 8 == 0, 16-byte aligned tensors.
// ---------------------------------------------------------------------------
using FusedConvTile = TileB<128, 192, 64, 4, 2>;
constexpr int kFusedHbStages = 4;
constexpr int kFusedMaxT = 512;
constexpr int kFusedLdsRing = 2 * ProjGldsTile::BN * kGldsRowBytes;                  // behind the two W stages
constexpr int kFusedLdsU = kFusedLdsRing + 3 * ProjGldsTile::BM * kGldsRowBytes;     // behind the A ring
constexpr int kFusedLdsTotal = kFusedLdsU + 16384;
static_assert(kFusedHbStages * FusedConvTile::BN * kGldsRowBytes <= kFusedLdsRing, "the Hb ring lives in the W stages");
static_assert(kFusedLdsTotal <= 160 * 1024, "one workgroup's LDS");
static_assert(kFusedLdsTotal >= ProjGldsTile::BM * (ProjGldsTile::BN + 8) * 2, "epilogue tile fits");
static_assert((kFusedMaxT / 8 + 16) * 128 + kFusedMaxT * 4 <= 16384, "u copies fit");

__global__ __launch_bounds__(ProjGldsTile::THREADS) void k_conv_proj_fwd_bf16(
    const float* __restrict__ u, const bf16_t* __restrict__ Hb, int ldHb, const bf16_t* __restrict__ x,
    const bf16_t* __restrict__ Wb, const float* __restrict__ bias, bf16_t* __restrict__ m, bf16_t* __restrict__ h,
    bf16_t* __restrict__ h_n, unsigned char* __restrict__ hmask, int T, int M, int D, int Hd, int ldW, int ntile,
    int FT, LnEpilogue ln) {
  using TC = FusedConvTile;
  using TP = ProjGldsTile;
  static_assert(TC::THREADS == TP::THREADS && TC::BM == TP::BM, "one workgroup shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // the longest tiles of an utterance first
  const int b = (int)blockIdx.x / ntile, j = ntile - 1 - (int)blockIdx.x % ntile;
  const int tstart = j * FT, tend = min(T, tstart + FT), t0 = tend - TC::BM;
  const int kend = tend;  // lags 0 .. tend - 1
  const int kend_pad = cdiv_dev(kend, TC::BK) * TC::BK;

  LMUFFT_STAMP(6, 0);
  // ---- conv phase ----
  typedef GldsOperand<TC::BN, TC::NW, false, TC::BK> OpH;
  OpH oh(
      smem_raw,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        if (r >= M) return false;
        q0 = Hb + (long)r * ldHb;
        return true;
      },
      ldHb, ldHb);
  constexpr int NSH = kFusedHbStages;
  const int nkt = kend_pad / TC::BK;  // k-tiles of the conv phase (2 .. 8): only these are fetched
#pragma unroll
  for (int st = 0; st < NSH - 1; ++st)
    if (st < nkt) oh.issue(st, st * TC::BK);

  // time-reversed u around the tile, eight shifted copies (see k_conv_fwd_bf16).  (The barriers here also wait
  // for the Hb tiles just issued -- the compiler orders LDS writes behind outstanding LDS-DMA; giving this work
  // to four waves that issue no DMA, or loading u first, measured slower.)
  bf16_t* U8 = reinterpret_cast<bf16_t*>(smem_raw + kFusedLdsU);
  const int nchunks = (kend_pad / 8 + TC::BM / 8) * 8;
  float* uf = reinterpret_cast<float*>(U8 + nchunks * 8);
  const float* ub = u + (long)b * T;
  const int smax = t0 + TC::BM - 1;
  for (int sidx = threadIdx.x; sidx < kend; sidx += TC::THREADS) uf[sidx] = ub[sidx];
  __syncthreads();
  for (int ch = threadIdx.x; ch < nchunks; ch += TC::THREADS) {
    const int Q = ch >> 3, c = ch & 7;
    bf16x8 v;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int pp = 8 * Q + jj - c;
      const int sidx = smax - pp;
      v[jj] = (bf16_t)((pp >= 0 && sidx >= 0 && sidx < kend) ? uf[sidx] : 0.f);
    }
    *reinterpret_cast<bf16x8*>(&U8[ch * 8]) = v;
  }
  __syncthreads();
  LMUFFT_STAMP(6, 1);

  // the projection's operands (address arithmetic done while the first Hb tiles fly)
  const int K = M + D;
  const long grow0 = (long)b * T + t0;  // global row of tile row 0
  typedef GldsOperand<TP::BM, TP::NW, false, TP::BK> OpX;
  typedef GldsOperand<TP::BN, TP::NW, false, TP::BK> OpW;
  OpX ox(
      smem_raw + kFusedLdsRing,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        const int t = t0 + r;
        if (t < tstart || t >= tend) return false;
        q0 = x + (grow0 + r) * D - M;  // element k of the A row, M <= k < M + D
        return true;
      },
      K, K);
  OpW ow(
      smem_raw,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        if (r >= Hd) return false;
        q0 = Wb + (long)r * ldW;
        return true;
      },
      ldW, ldW);
  {
    ToeplitzOperand<TC> oa{U8};
    WavePos<TC> w;
    f32x16 acc[TC::TN];
#pragma unroll
    for (int tn = 0; tn < TC::TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;
    static_assert(TC::TM == 1, "one row tile per wave");
    static_assert(NSH == 4, "the counted waits below are written for three tiles in flight");
    int stage = 0, pf = NSH - 1;
    for (int k0 = 0, kt = 0; k0 < kend; k0 += TC::BK, ++kt) {
      // tile kt has landed when at most the tiles behind it are outstanding: two of them, or what is left of the
      // sequence at its end (no dummy tiles are fetched to keep the count constant: with two to seven k-tiles per
      // workgroup three dummies were up to 60 % of the phase's LDS-DMA instructions)
      const int behind = nkt - 1 - kt;  // uniform
      if (behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OpH::NI * 2) : "memory");
      else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OpH::NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + NSH - 1 < nkt) oh.issue(pf, k0 + (NSH - 1) * TC::BK);
#pragma unroll
      for (int kl = 0; kl < TC::BK; kl += 16) {
        const bf16x8 af = oa.frag(w.wm0 + w.r, k0, kl, w.hh);
#pragma unroll
        for (int tn = 0; tn < TC::TN; ++tn) {
          const bf16x8 bfr = oh.frag(stage, w.wn0 + tn * 32 + w.r, kl, w.hh);
          acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[tn], 0, 0, 0);
        }
      }
      stage = stage + 1 == NSH ? 0 : stage + 1;
      pf = pf + 1 == NSH ? 0 : pf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing k-tiles
    __syncthreads();                                  // the Hb ring is free: the W stages take its place
    LMUFFT_STAMP(6, 2);

    // ---- hand-over: the m tile as the projection's first A k-tiles ----
    unsigned char* ring = smem_raw + kFusedLdsRing;
#pragma unroll
    for (int tn = 0; tn < TC::TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = w.wm0 + (i & 3) + 8 * (i >> 2) + 4 * w.hh;
        const int n = w.wn0 + tn * 32 + w.r;
        if (n < M)
          *reinterpret_cast<bf16_t*>(ring + (n >> 6) * (TP::BM * kGldsRowBytes) + row * kGldsRowBytes +
                                     ((((n & 63) >> 3) ^ ((row >> 1) & 7)) << 4) + (n & 7) * 2) = (bf16_t)acc[tn][i];
      }
  }

  // ---- projection ----
  ow.issue(0, 0);  // behind the ring writes: in front of them the compiler would make the writes wait for it
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // the m tile is in the ring
  LMUFFT_STAMP(6, 3);
  WavePos<TP> w;
  f32x16 acc[TP::TM][TP::TN];
  tileb_zero<TP>(acc);
  const int KM = M / TP::BK, KT = cdiv_dev(K, TP::BK);
  int wst = 0, slot = 0, nslot = 1;  // ring slot of k-tile kt and of kt + 1
  for (int kt = 0; kt < KT; ++kt, wst ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) {
      ow.issue(wst ^ 1, (kt + 1) * TP::BK);
      if (kt + 1 >= KM) ox.issue(nslot, (kt + 1) * TP::BK);
    }
    if (kt == 0) {
      // m to HBM as whole 16-byte row segments, from the ring (its m tiles stay until the x tiles replace them, from
      // iteration 2 on).  Here, behind the loads of k-tile 1, the stores complete under that round trip; in front
      // of the loop the first vmcnt(0) would wait for them.
      const unsigned char* ring = smem_raw + kFusedLdsRing;
      const int CM = M / 8;
      bf16_t* mb = m + (long)b * T * M;
      for (int idx = threadIdx.x; idx < TP::BM * CM; idx += TP::THREADS) {
        const int row = idx / CM, c = idx - row * CM;
        const int t = t0 + row;
        if (t >= tstart && t < tend)
          *reinterpret_cast<bf16x8*>(mb + (long)t * M + c * 8) = *reinterpret_cast<const bf16x8*>(
              ring + (c >> 3) * (TP::BM * kGldsRowBytes) + row * kGldsRowBytes + (((c & 7) ^ ((row >> 1) & 7)) << 4));
      }
    }
#pragma unroll
    for (int kl = 0; kl < TP::BK; kl += 16) {
      bf16x8 af[TP::TM], bfr[TP::TN];
#pragma unroll
      for (int tm = 0; tm < TP::TM; ++tm) af[tm] = ox.frag(slot, w.wm0 + tm * 32 + w.r, kl, w.hh);
#pragma unroll
      for (int tn = 0; tn < TP::TN; ++tn) bfr[tn] = ow.frag(wst, w.wn0 + tn * 32 + w.r, kl, w.hh);
#pragma unroll
      for (int tm = 0; tm < TP::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TP::TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
    }
    slot = nslot;
    nslot = nslot == 2 ? 0 : nslot + 1;
  }
  __syncthreads();
  LMUFFT_STAMP(6, 5);
  proj_epilogue<TP>(acc, w, smem_raw, bias, h, h_n, hmask, ln, grow0, (long)b * T + tstart, (long)b * T + tend, T, Hd, 0);
  LMUFFT_STAMP(6, 6);
  LMUFFT_STAMP_ADD(6, 4, j - g_stamps[stamp_block() * 8 + 4]);
}

// ---------------------------------------------------------------------------
// gz = (h > 0) ? gh (+ gh_n on the last frame) : 0
// One pass over h and gh; the three GEMMs of the backward read gz afterwards
// (the weight-gradient kernel also takes gb_h = sum_rows gz from its staged tiles).
// ---------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_relu_bwd_bf16(const bf16_t* __restrict__ h,
                                                            const bf16_t* __restrict__ gh,
                                                            const bf16_t* __restrict__ gh_n,
                                                            bf16_t* __restrict__ gz, int rows, int T, int Hd) {
  if (VEC) {
    const unsigned CH = (unsigned)Hd / 8;  // 16-byte chunks per row
    const long total = (long)rows * CH;
    for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
      const unsigned row = (unsigned)(i / CH), c = (unsigned)(i - (long)row * CH);
      const long off = (long)row * Hd + c * 8;
      const bf16x8 hv = *reinterpret_cast<const bf16x8*>(h + off);
      bf16x8 g = gh ? *reinterpret_cast<const bf16x8*>(gh + off) : bzero8();
      if (gh_n) {
        const unsigned b = row / (unsigned)T;
        if (row - b * (unsigned)T == (unsigned)(T - 1)) {
          const bf16x8 gl = *reinterpret_cast<const bf16x8*>(gh_n + (long)b * Hd + c * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) g[j] = (bf16_t)((float)g[j] + (float)gl[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = (float)hv[j] > 0.f ? g[j] : (bf16_t)0.f;
      *reinterpret_cast<bf16x8*>(gz + off) = g;
    }
  } else {
    const long total = (long)rows * Hd;
    for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
      const long row = i / Hd;
      const int n = (int)(i - row * Hd);
      float g = gh ? (float)gh[i] : 0.f;
      if (gh_n && row % T == T - 1) g += (float)gh_n[(row / T) * Hd + n];
      gz[i] = (float)h[i] > 0.f ? (bf16_t)g : (bf16_t)0.f;
    }
  }
}

// ---------------------------------------------------------------------------
// Encoder gradient straight from the masked upstream gradient ("Q route", T <= kGuQMaxT).
// The reference's backward goes gz -> gm = gz W_h[:, :M] -> gu = adjoint convolution of gm with H
// (autograd of src/lmu.py:303-316).  Both steps are linear, so they compose:
//   gu[b][s] = sum_{t >= s} sum_n gz[b][t][n] Q[n][t - s],   Q[n][tau] = sum_k W_h[n][k] H[k][tau],
// one GEMM of gz against the constant Q^T [T][Hd] followed by sums along the diagonals t - tau = s --
// gm is never formed, written or read back (two kernels and ~65 MB of HBM traffic per layer at
// batch 64 x 400 frames become one kernel reading gz).  Q is rounded to bf16 once per step, as W_h and H are
// on the two-step route, where gm is rounded besides.
//
// k_relu_bwd_q_bf16: the gz pass; its first workgroups make Q^T on the matrix cores from the bf16 copies of
// H^T (hprep) and W_h (wprep) -- 30 MFLOP that would otherwise be a launch of their own.
// ---------------------------------------------------------------------------
constexpr int kGuQMaxT = 512;

__global__ __launch_bounds__(kThreads) void k_relu_bwd_q_bf16(const bf16_t* __restrict__ h,
                                                              const unsigned char* __restrict__ hmask,
                                                              const bf16_t* __restrict__ gh,
                                                              const bf16_t* __restrict__ gh_n,
                                                              bf16_t* __restrict__ gz, int rows, int T, int Hd,
                                                              QPrepBatch qb) {
  __shared__ __attribute__((aligned(16))) bf16_t qsmem[kQPrepLdsElems];
  const int qblocks = qb.first[qb.count];
  if ((int)blockIdx.x < qblocks) {
    qprep_batch_block<QPrepTile>(qb, (int)blockIdx.x, qsmem);
    return;
  }
  const unsigned CH = (unsigned)Hd / 8;  // 16-byte chunks per row
  const long total = (long)rows * CH;
  const long nb = (long)gridDim.x - qblocks;
  for (long i = ((long)blockIdx.x - qblocks) * kThreads + threadIdx.x; i < total; i += nb * kThreads) {
    const unsigned row = (unsigned)(i / CH), c = (unsigned)(i - (long)row * CH);
    const long off = (long)row * Hd + c * 8;
    unsigned keep;  // bit j: h[row][8 c + j] > 0 -- from the forward's bit mask when there is one (1/16 of the bytes)
    if (hmask) {
      keep = hmask[i];
    } else {
      const bf16x8 hv = *reinterpret_cast<const bf16x8*>(h + off);
      keep = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) keep |= ((float)hv[j] > 0.f ? 1u : 0u) << j;
    }
    bf16x8 g = gh ? *reinterpret_cast<const bf16x8*>(gh + off) : bzero8();
    if (gh_n) {
      const unsigned b = row / (unsigned)T;
      if (row - b * (unsigned)T == (unsigned)(T - 1)) {
        const bf16x8 gl = *reinterpret_cast<const bf16x8*>(gh_n + (long)b * Hd + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = (bf16_t)((float)g[j] + (float)gl[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = ((keep >> j) & 1u) ? g[j] : (bf16_t)0.f;
    *reinterpret_cast<bf16x8*>(gz + off) = g;
  }
}

// Q^T alone (the upstream gradient arrives masked, see LMUFFT_BWD_GH_MASKED): one workgroup per 64 x 64 tile
__global__ __launch_bounds__(kThreads) void k_qprep_bf16(QPrepBatch qb) {
  __shared__ __attribute__((aligned(16))) bf16_t qsmem[kQPrepWideLdsElems];
  qprep_batch_block<QPrepWideTile>(qb, (int)blockIdx.x, qsmem);
}

// k_gu_q_bf16: workgroup = (utterance b, tile of 128 frames t -- right-aligned, so that the short tile is the
// first one --, chunk c of 256 lags); P[i][tau] = sum_n gz[b][t0 + i][n] Q^T[tau][n] for the chunk's lags below
// t0 + 128, then sums along the diagonals i - tau = s - t0.  A GEMM tile of 128 x 256 on eight waves (a wave owns 32
// frames x 128 lags: four 32 x 32 accumulators), K = Hd in k-tiles of 32 through a three-stage LDS-DMA ring with
// counted waits; column tiles past the causal limit are skipped.  74 KB of LDS: two workgroups share a CU, so the
// 6 B (tile, chunk) pairs of batch 64 x 400 frames are all resident at once (with one 147 KB workgroup per CU and a
// six-stage ring the long pairs ran no faster -- the kernel is bound by what the chip's CUs ingest together, about
// 7.5 TB/s of L2 and HBM reads through the LDS-DMA path -- and the short pairs waited for a second round).
// The diagonal sums happen in registers (sixteen lane rotations per 32 x 32 tile, see below); the waves' 63 sums
// per tile meet in a small LDS array and are added in a fixed order into gu_part[slot][b][s], slot = the
// (tile, chunk) pair's number inside the utterance; the input-gradient GEMM adds the slots.
using GuQTile = TileB<128, 256, 32, 4, 2>;
constexpr int kGuQStages = 3;

constexpr int kGuQMaxSlots = 32;
struct GuQPlan {  // the (tile, chunk) pairs of an utterance, longest first
  int nslot;
  unsigned char tile[kGuQMaxSlots], chunk[kGuQMaxSlots];
};
static GuQPlan guq_plan(int T) {
  const int BM = GuQTile::BM, BN = GuQTile::BN, pad = (BM - T % BM) % BM;
  GuQPlan p;
  p.nslot = 0;
  int work[kGuQMaxSlots];
  for (int j = 0; j * BM - pad < T; ++j) {
    const int t0 = j * BM - pad, tend = t0 + BM;
    for (int c = 0; c * BN < tend && p.nslot < kGuQMaxSlots; ++c) {
      int tiles = 0;  // 32 x 32 tiles with a causal pair
      for (int rm = 0; rm < BM / 32; ++rm) {
        const int lags = std::min(std::min(BN, tend - c * BN), t0 + 32 * rm + 32 - c * BN);
        tiles += lags > 0 ? (lags + 31) / 32 : 0;
      }
      int k = p.nslot++;
      while (k > 0 && work[k - 1] < tiles) {  // insertion sort, descending
        work[k] = work[k - 1];
        p.tile[k] = p.tile[k - 1];
        p.chunk[k] = p.chunk[k - 1];
        --k;
      }
      work[k] = tiles;
      p.tile[k] = (unsigned char)j;
      p.chunk[k] = (unsigned char)c;
    }
  }
  return p;
}

__global__ __launch_bounds__(GuQTile::THREADS)
void k_gu_q_bf16(const bf16_t* __restrict__ gz, const bf16_t* __restrict__ QT, float* __restrict__ part, int B, int T,
                 int Hd, int ldQ, int rowsQ, GuQPlan plan) {
  using TL = GuQTile;
  constexpr int NS = kGuQStages;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // the grid runs slot after slot, longest first: the short ones start on the CUs whose first workgroup has finished
  const int slot = (int)blockIdx.x / B, b = (int)blockIdx.x % B;
  const int j = plan.tile[slot], c = plan.chunk[slot];
  const int t0 = j * TL::BM - (TL::BM - T % TL::BM) % TL::BM;  // < 0 for a partial first tile
  const int tend = t0 + TL::BM;                    // <= T
  const int lag0 = c * TL::BN;                     // first lag of the chunk
  const int nlag = min(TL::BN, tend - lag0);       // lags of the chunk that have a causal pair (> 0 by construction)
  const bf16_t* gzb = gz + (long)b * T * Hd;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

  // LDS-DMA staging, k-tiles of 32: a piece = one wave-instruction = 16 rows x 64 bytes; row r's 16-byte chunk d sits
  // at position d ^ ((r >> 2) & 3) (conflict-free ds_read_b128 fragments), applied on the source side.  Wave w
  // stages the frames' piece w and the lags' pieces w and w + 8.  A piece with no needed row in it is never issued
  // (its LDS rows keep whatever they held); inside an issued piece rows outside the operand re-read its nearest
  // valid row.  Neither touches the result: a frame t < lag0 or a lag beyond the causal limit only meets diagonals
  // s = t - lag < 0, which nobody reads, and a diagonal is needed as a whole or not at all.
  constexpr int kStageA = TL::BM * 64, kStageB = TL::BN * 64, kStage = kStageA + kStageB;
  static_assert(TL::NW == 8 && TL::BM == 128 && TL::BN == 256 && TL::BK == 32, "piece assignment");
  const bf16_t* src[3];
  bool live[3];
  {
    const int rr = lane >> 2, pos = lane & 3;
    int row = 16 * wave + rr;                       // frames
    live[0] = t0 + 16 * wave + 15 >= lag0;          // (lag0 >= 0: frames before the chunk's first lag meet no causal pair)
    src[0] = gzb + (long)max(t0 + row, 0) * Hd + 8 * (pos ^ ((row >> 2) & 3));
#pragma unroll
    for (int i = 0; i < 2; ++i) {                   // lags
      row = 16 * (wave + 8 * i) + rr;
      live[1 + i] = 16 * (wave + 8 * i) < nlag;
      src[1 + i] = QT + (long)(lag0 + min(row, nlag - 1)) * ldQ + 8 * (pos ^ ((row >> 2) & 3));
    }
  }
  const int npiece = __builtin_amdgcn_readfirstlane((live[0] ? 1 : 0) + (live[1] ? 1 : 0) + (live[2] ? 1 : 0));
  auto issue = [&](int stage, int k0) {
    const int kk = k0 < Hd ? k0 : 0;                // k-tiles past the end re-read the first one (nobody reads them)
    unsigned char* base = smem_raw + stage * kStage;
    if (live[0])
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[0] + kk),
                                       (__attribute__((address_space(3))) void*)(base + wave * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (live[1 + i])
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[1 + i] + kk),
                                         (__attribute__((address_space(3))) void*)(base + kStageA + (wave + 8 * i) * 1024),
                                         16, 0, 0);
  };
  auto wait_tile = [&]() {  // all but the newest NS - 2 k-tiles of this wave's pieces have landed
    switch (npiece) {
      case 0: break;
      case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * (NS - 2)) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NS - 2)) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NS - 2)) : "memory"); break;
    }
  };
  auto frag = [&](int stage, int off, int row, int kl, int hh) -> bf16x8 {
    const int d = (kl >> 3) + hh;
    return *reinterpret_cast<const bf16x8*>(smem_raw + stage * kStage + off + row * 64 + ((d ^ ((row >> 2) & 3)) << 4));
  };

  WavePos<TL> w;
  LMUFFT_STAMP(5, 0);
  const int rmw = wave / TL::WGN, cw = wave % TL::WGN;  // this wave's row tile and column half
  // column tiles of this wave that hold a causal pair: lag0 + WN cw + 32 tn <= t0 + 32 rmw + 31
  int ntn = (min(nlag, t0 + 32 * rmw + 32 - lag0) - TL::WN * cw + 31) / 32;
  ntn = __builtin_amdgcn_readfirstlane(ntn < 0 ? 0 : (ntn > TL::TN ? TL::TN : ntn));
  f32x16 acc[TL::TN];
#pragma unroll
  for (int tn = 0; tn < TL::TN; ++tn)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;

  {  // K loop: NS - 1 k-tiles in flight, counted waits, raw barriers; only the Hd / BK real k-tiles are fetched
    const int nkt = Hd / TL::BK;
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
      if (st < nkt) issue(st, st * TL::BK);
    int stage = 0, pf = NS - 1;
    for (int k0 = 0, kt = 0; k0 < Hd; k0 += TL::BK, ++kt) {
      if (kt + NS - 2 < nkt) wait_tile();  // NS - 2 newer tiles are outstanding behind this one ...
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... or none (NS = 3), at the end of the sequence
      __builtin_amdgcn_s_barrier();
      if (kt + NS - 1 < nkt) issue(pf, k0 + (NS - 1) * TL::BK);
#pragma unroll
      for (int kl = 0; kl < TL::BK; kl += 16) {
        const bf16x8 af = frag(stage, 0, w.wm0 + w.r, kl, w.hh);
#pragma unroll
        for (int tn = 0; tn < TL::TN; ++tn)
          if (tn < ntn) {
            const bf16x8 bfr = frag(stage, kStageA, w.wn0 + tn * 32 + w.r, kl, w.hh);
            acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[tn], 0, 0, 0);
          }
      }
      stage = stage + 1 == NS ? 0 : stage + 1;
      pf = pf + 1 == NS ? 0 : pf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing k-tiles
    __syncthreads();                                  // the stages are free
  }
  LMUFFT_STAMP(5, 1);

  // Diagonal sums of a 32 x 32 product, in registers.  Accumulator register i of lane (r, hh) is the element
  // (row rho = (i & 3) + 8 (i >> 2) + 4 hh, lag r); fetched from lane (r' + rho) mod 32 of the same half it lands,
  // in lane r', on the diagonal delta = row - lag = -r' (no wrap) or 32 - r' (wrap): sixteen lane rotations and two
  // running sums per lane replace an LDS image of the tile and a pass over it.
  float* wsum = reinterpret_cast<float*>(smem_raw);  // [row tile][column tile][half hh][64]: sums of delta = idx - 31
  {
    int rot[16];
    unsigned wrapmask = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int v = w.r + (i & 3) + 8 * (i >> 2) + 4 * w.hh;
      rot[i] = ((lane & 32) | (v & 31)) << 2;
      wrapmask |= (v >= 32 ? 1u : 0u) << i;
    }
    float pv[TL::TN][16];
#pragma unroll
    for (int tn = 0; tn < TL::TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float a = acc[tn][i];  // (a bit_cast applied to the vector element itself read element 0 sixteen times)
        pv[tn][i] = __int_as_float(__builtin_amdgcn_ds_bpermute(rot[i], __float_as_int(a)));
      }
#pragma unroll
    for (int tn = 0; tn < TL::TN; ++tn) {
      float d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool wrap = (wrapmask >> i) & 1u;
        d1 += wrap ? 0.f : pv[tn][i];
        d2 += wrap ? pv[tn][i] : 0.f;
      }
      // delta = -r (d1) and 32 - r (d2; r = 0: none) of this half's rows; the two halves are added by the readers
      float* dst = wsum + ((rmw * (TL::WGN * TL::TN) + cw * TL::TN + tn) * 2 + w.hh) * 64;
      dst[31 - w.r] = d1;
      dst[63 - w.r] = d2;
    }
  }
  __syncthreads();
  LMUFFT_STAMP(5, 2);
  // frame s collects, from every row tile rm, the (at most two) column tiles whose diagonal delta = s - t0 - 32 rm
  // + lag0 + 32 tn lies in [-31, 31]; fixed order
  float* out = part + ((long)slot * B + b) * T;
  for (int s = threadIdx.x; s < T; s += TL::THREADS) {
    float sum = 0.f;
    if (s < tend) {
#pragma unroll
      for (int rm = 0; rm < TL::WGM; ++rm) {
        // rows 32 rm + rho, lags lag0 + 32 tn + r: s = t0 + row - lag, so delta = rho - r = 32 tn - x with
        const int x = 32 * rm + t0 - s - lag0;
        const int tlo = x >> 5;                         // the tile with delta in [-31, 0] (floor)
        const int dlo = 32 * tlo - x;
        constexpr int CT = TL::BN / 32;  // column tiles of the workgroup
        if (tlo >= 0 && tlo < CT) {
          const float* p = wsum + (rm * CT + tlo) * 128 + dlo + 31;
          sum += p[0] + p[64];
        }
        if (tlo + 1 >= 0 && tlo + 1 < CT && dlo <= -1) {
          const float* p = wsum + (rm * CT + tlo + 1) * 128 + dlo + 63;
          sum += p[0] + p[64];
        }
      }
    }
    out[s] = sum;
  }
  LMUFFT_STAMP(5, 3);
  LMUFFT_STAMP_ADD(5, 4, slot - g_stamps[stamp_block() * 8 + 4]);
}


// ---------------------------------------------------------------------------
// out[row][j] = sum_n gz[row][n] W_h[n][j_begin + j]  (+ ga[row] w_u[j])
// called for the memory columns (-> gm) and for the input columns (-> gx)
// ---------------------------------------------------------------------------
template <class TL, bool VEC>
__global__ __launch_bounds__(kThreads) void k_dgrad_bf16(const bf16_t* __restrict__ gz,
                                                         const bf16_t* __restrict__ WbT, bf16_t* __restrict__ out,
                                                         const float* __restrict__ ga,
                                                         const float* __restrict__ w_u, int rows, int Hd,
                                                         int ldWT, int j_begin, int ncols) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int n0 = blockIdx.x * TL::BN;
  const int m0 = blockIdx.y * TL::BM;

  auto la = plain_load_b([=](int r, int k) -> bf16x8 {
    const long row = m0 + r;
    return bload8<VEC>(gz, row * Hd + k, k, Hd, row < rows);
  });
  auto lb = plain_load_b([=](int r, int k) -> bf16x8 {
    const int j = n0 + r;
    return bload8<true>(WbT, (long)(j_begin + j) * ldWT + k, k, ldWT, j < ncols);
  });
  typedef LdsOperandKC<TL::BM, TL::BK, decltype(la)> OpA;
  typedef LdsOperandKC<TL::BN, TL::BK, decltype(lb)> OpB;
  OpA oa(smem, la);
  OpB ob(smem + OpA::ELEMS, lb);

  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  gemm_loop_kc<TL>(oa, ob, acc, 0, Hd, w);

  float wcol[TL::TN], grow[TL::TM][16];
  preload_cols<TL>(wcol, w, [&](int nn) { return load_or_zero(w_u + n0 + nn, ga && n0 + nn < ncols); });
  preload_rows<TL>(grow, w, [&](int mm) { return load_or_zero(ga + m0 + mm, ga && m0 + mm < rows); });
  tile_epilogue_ix<TL>(acc, w, [&](int tm, int tn, int i, int mm, int nn, float v) {
    const long row = m0 + mm;
    const int j = n0 + nn;
    if (row < rows && j < ncols) out[row * ncols + j] = (bf16_t)(v + grow[tm][i] * wcol[tn]);
  });
}

// LDS-DMA form; needs Hd % 8 == 0, ncols % 8 == 0 and 16-byte aligned rows
// partial slabs of the adjoint convolution (second form): part[p][row], p < P <= kGaMaxParts
constexpr int kGaMaxParts = 32;
struct GaParts {
  const float* part;  // NULL: ga comes as an array
  const float* u;
  float* ga_out;
  long stride;
  int P;
};

template <class TL>
__global__ __launch_bounds__(TL::THREADS) void k_dgrad_glds_bf16(const bf16_t* __restrict__ gz,
                                                                 const bf16_t* __restrict__ WbT,
                                                                 bf16_t* __restrict__ out,
                                                                 const float* __restrict__ ga,
                                                                 const float* __restrict__ w_u, int rows_all, int Hd,
                                                                 int ldWT, int j_begin, int ncols, GaParts gp,
                                                                 int rpw, int col_tiles,
                                                                 const bf16_t* __restrict__ relu_src,
                                                                 const unsigned char* __restrict__ relu_bits) {
  // A workgroup owns rpw <= BM rows (rpw is chosen so that the grid fills the chip: the tile's remaining rows
  // are computed on zeros and not stored) and one column tile; the column tiles of a row range are neighbours
  // on one XCD (1-D grid, xcd_slot), so that the rows they share come from HBM once.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const XcdSlot slot = xcd_slot(col_tiles);
  const int n0 = slot.member * TL::BN;
  const long m0 = (long)slot.group * rpw;
  if (m0 >= rows_all) return;
  const int rows = (int)min((long)rows_all, m0 + rpw);  // rows at or beyond it belong to the next workgroup

  // Encoder-activation gradient from the adjoint convolution's partial slabs, for this tile's rows:
  // ga[row] = (u[row] > 0) sum_p part[p][row].  The partial loads are issued here and summed after
  // the GEMM loop, so their latency hides behind it; the column-tile-0 workgroup also writes ga
  // for the weight-gradient kernel.  (Replaces a separate reduction kernel.)
  constexpr int kGaGroups = TL::THREADS / TL::BM;
  constexpr int kGaPer = (kGaMaxParts + kGaGroups - 1) / kGaGroups;
  float ga_part[kGaPer];
  const int ga_row = threadIdx.x % TL::BM, ga_grp = threadIdx.x / TL::BM;
  if (gp.part) {
#pragma unroll
    for (int i = 0; i < kGaPer; ++i) {
      const int p = ga_grp + i * kGaGroups;
      ga_part[i] = load_or_zero(gp.part + (long)p * gp.stride + m0 + ga_row, p < gp.P && m0 + ga_row < rows);
    }
  }

  // relu_src / relu_bits: out[row][j] is kept only where relu_src[row][j] > 0 (relu_src = the [rows_all][ncols]
  // tensor whose ReLU backward is the next thing to happen to this gradient; relu_bits, if given, holds that
  // comparison already: bit j of byte [row][c] for column 8 c + j).  What matches the 16-byte pieces of `out` this
  // thread stores is fetched here, so that it arrives behind the GEMM loop.
  constexpr int LDT = TL::BN + 8, CH = TL::BN / 8;
  constexpr int PER = TL::BM * CH / TL::THREADS;
  static_assert(TL::BM * CH % TL::THREADS == 0, "whole chunks per thread");
  bf16x8 src[PER];
  unsigned keep[PER];
  if (relu_bits) {
#pragma unroll
    for (int n = 0; n < PER; ++n) {
      const int idx = threadIdx.x + n * TL::THREADS;
      const long row = m0 + idx / CH;
      const int col = n0 + (idx % CH) * 8;
      keep[n] = (row < rows && col < ncols) ? relu_bits[row * (ncols / 8) + col / 8] : 0u;
    }
  } else if (relu_src) {
#pragma unroll
    for (int n = 0; n < PER; ++n) {
      const int idx = threadIdx.x + n * TL::THREADS;
      const long row = m0 + idx / CH;
      const int col = n0 + (idx % CH) * 8;
      src[n] = (row < rows && col < ncols) ? *reinterpret_cast<const bf16x8*>(relu_src + row * ncols + col) : bzero8();
    }
  }

  GldsOperand<TL::BM, TL::NW, false> oa(
      smem_raw,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        const long row = m0 + r;
        if (row >= rows) return false;
        q0 = gz + row * Hd;
        return true;
      },
      Hd, Hd);
  GldsOperand<TL::BN, TL::NW, false> ob(
      smem_raw + 2 * GldsOperand<TL::BM, TL::NW, false>::STAGE_BYTES,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        const int j = n0 + r;
        if (j >= ncols) return false;
        q0 = WbT + (long)(j_begin + j) * ldWT;
        return true;
      },
      ldWT, ldWT);

  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  gemm_loop_glds<TL>(oa, ob, acc, 0, Hd, w);

  float wcol[TL::TN], grow[TL::TM][16];
  if (gp.part) {  // (the loop ended with a barrier: the stages are free)
    float* sc = reinterpret_cast<float*>(smem_raw);  // [kGaGroups][BM] partial sums, then [BM] ga
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kGaPer; ++i) s += ga_part[i];
    sc[ga_grp * TL::BM + ga_row] = s;
    __syncthreads();
    float v = 0.f;
    if (threadIdx.x < TL::BM) {
#pragma unroll
      for (int g = 0; g < kGaGroups; ++g) v += sc[g * TL::BM + ga_row];
      const long row = m0 + ga_row;
      v = (row < rows && gp.u[row] > 0.f) ? v : 0.f;
      if (row < rows && n0 == 0) gp.ga_out[row] = v;
    }
    __syncthreads();
    if (threadIdx.x < TL::BM) sc[ga_row] = v;
    __syncthreads();
    preload_cols<TL>(wcol, w, [&](int nn) { return load_or_zero(w_u + n0 + nn, n0 + nn < ncols); });
    preload_rows<TL>(grow, w, [&](int mm) { return sc[mm]; });
    __syncthreads();  // the epilogue reuses the same LDS
  } else {
    preload_cols<TL>(wcol, w, [&](int nn) { return load_or_zero(w_u + n0 + nn, ga && n0 + nn < ncols); });
    preload_rows<TL>(grow, w, [&](int mm) { return load_or_zero(ga + m0 + mm, ga && m0 + mm < rows); });
  }
  bf16_t* tile = reinterpret_cast<bf16_t*>(smem_raw);
  auto value = [&](int tm, int tn, int i, int, int, float v) { return v + grow[tm][i] * wcol[tn]; };
  if (!relu_src && !relu_bits) {
    store_tile_bf16<TL>(acc, w, tile, out, ncols, m0, n0, rows, ncols, value);
    return;
  }
  if (!relu_bits) {
#pragma unroll
    for (int n = 0; n < PER; ++n) {
      keep[n] = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) keep[n] |= ((float)src[n][j] > 0.f ? 1u : 0u) << j;
    }
  }
  // the tile goes to LDS as usual; the store pass applies the mask fetched at the start
  store_tile_bf16<TL>(acc, w, tile, nullptr, 0, 0, 0, 0, 0, value);
#pragma unroll
  for (int n = 0; n < PER; ++n) {
    const int idx = threadIdx.x + n * TL::THREADS;
    const int mm = idx / CH, c = idx % CH;
    const long row = m0 + mm;
    const int col = n0 + c * 8;
    if (row < rows && col < ncols) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(&tile[mm * LDT + c * 8]);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ((keep[n] >> j) & 1u) ? v[j] : (bf16_t)0.f;
      *reinterpret_cast<bf16x8*>(out + row * ncols + col) = v;
    }
  }
}

using DgradMTileB = TileB<64, 192, 64, 2, 2>;    // memory columns (M = 192): one column tile
using DgradXTileB = TileB<128, 128, 64, 2, 2>;   // register-staged fallback for the input columns
// input columns, LDS-DMA form, eight waves.  256 x 192: a workgroup's operand ingest per output is smallest
// when its rows and columns are about equal in number (the whole weight panel passes through every CU);
// 128 x 384 pulled 393 KB per 128 rows, this tile with 200-row ranges 301 KB per 200.
using DgradXGldsTile = TileB<256, 192, 64, 4, 2>;

// ---------------------------------------------------------------------------
// gu partial slabs: P[i][tau] = sum_k gm[t0+i][k] H[k][tau], summed along the
// diagonals s = t0 + i - tau (through LDS, in a fixed order), one slab per (batch, t tile)
// ---------------------------------------------------------------------------
using ConvBwdTileB = TileB<64, 128, 64, 2, 2>;

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_conv_bwd_bf16(const bf16_t* __restrict__ gm,
                                                            const bf16_t* __restrict__ HbT,
                                                            float* __restrict__ gu_part, int T, int M, int ldHT,
                                                            int rowsHT) {
  using TL = ConvBwdTileB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int b = blockIdx.y;
  const int ntt = gridDim.z;
  const int tb = ntt - 1 - blockIdx.z;
  const int t0 = tb * TL::BM;
  const int send = min(T, t0 + TL::BM);
  const bf16_t* gmb = gm + (long)b * T * M;

  auto la = plain_load_b([=](int r, int k) -> bf16x8 {
    const int t = t0 + r;
    return bload8<VEC>(gmb, (long)t * M + k, k, M, t < T);
  });
  typedef LdsOperandKC<TL::BM, TL::BK, decltype(la)> OpA;
  constexpr int kBElems = TL::BN * (TL::BK + 8);
  float* gu_acc = reinterpret_cast<float*>(smem + OpA::ELEMS + kBElems);
  float* ptile = gu_acc + ntt * TL::BM;
  for (int s = threadIdx.x; s < send; s += kThreads) gu_acc[s] = 0.f;

  WavePos<TL> w;
  for (int tau0 = 0; tau0 < send; tau0 += TL::BN) {
    auto lb = plain_load_b([=](int r, int k) -> bf16x8 {
      const int tau = tau0 + r;
      return bload8<true>(HbT, (long)tau * ldHT + k, k, ldHT, tau < rowsHT);
    });
    typedef LdsOperandKC<TL::BN, TL::BK, decltype(lb)> OpB;
    static_assert(OpB::ELEMS == kBElems, "LDS carve");
    OpA oa(smem, la);
    OpB ob(smem + OpA::ELEMS, lb);
    f32x16 acc[TL::TM][TL::TN];
    tileb_zero<TL>(acc);
    gemm_loop_kc<TL>(oa, ob, acc, 0, M, w);
    add_tile_diagonals<TL>(acc, w, ptile, gu_acc, t0 - tau0, T - 1 - t0);
  }
  float* out = gu_part + ((long)b * ntt + tb) * T;
  for (int s = threadIdx.x; s < send; s += kThreads) out[s] = gu_acc[s];
}

// ---------------------------------------------------------------------------
// Adjoint convolution, second form (used when M % 8 == 0):
//   gu[b][s] = sum_{t >= s} sum_k gm[b][t][k] H[k][t - s]
// as a GEMM whose OUTPUT is gu itself, so no diagonal sums are needed:
//   rows = batch items   A[b][(t,k)] = gm[b][t][k]   (contiguous in HBM; the
//                        fragments are loaded straight into registers)
//   cols = frames s      B[(t,k)][s] = H^T[t - s][k] (a sliding window of H^T
//                        rows kept in LDS, zero rows where t < s)
//   K    = (t, k), t restricted to this workgroup's frame range.
// Grid (P, s tiles of 64, batch tiles of 32): the frames t >= s0 of an s tile
// are cut into P contiguous ranges; each workgroup writes its 32 x 64 partial
// tile to slab p, k_sum_slabs_masked adds the P slabs in order.  The four waves
// take every fourth frame of the range and are summed through LDS in wave order,
// so the result is deterministic.
// ---------------------------------------------------------------------------
constexpr int kCb2St = 64;  // frames s per tile
constexpr int kCb2Bt = 32;  // batch items per tile

template <int KS>  // k-steps of 16 per frame: KS = ceil(M / 16)
__global__ __launch_bounds__(kThreads) void k_conv_bwd2_bf16(const bf16_t* __restrict__ gm,
                                                             const bf16_t* __restrict__ HbT,
                                                             float* __restrict__ part, int B, int T, int M,
                                                             int ldHT, int rowsHT, int P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int KP = KS * 16, LDW = KP + 8;
  const int p = blockIdx.x, s0 = blockIdx.y * kCb2St, b0 = blockIdx.z * kCb2Bt;
  const int Lp = (T - s0 + P - 1) / P;
  const int ta = s0 + p * Lp, tb = min(T, ta + Lp);
  const int nrows = max(tb - ta, 0) + kCb2St - 1;
  bf16_t* window = reinterpret_cast<bf16_t*>(smem_raw);          // [nrows][LDW], row 0 <-> tau = ta - s0 - 63
  float* red = reinterpret_cast<float*>(smem_raw + (size_t)(Lp + kCb2St - 1) * LDW * sizeof(bf16_t));  // [4][32][64]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(kZeroBytes);
  LMUFFT_STAMP(3, 0);

  const int tau_lo = ta - s0 - (kCb2St - 1);
  for (int idx = threadIdx.x; idx < nrows * (KP / 8); idx += kThreads) {
    const int row = idx / (KP / 8), c = idx % (KP / 8);
    const int tau = tau_lo + row;
    const bool ok = tau >= 0 && tau < rowsHT && c * 8 < ldHT;
    *reinterpret_cast<bf16x8*>(&window[row * LDW + c * 8]) =
        *reinterpret_cast<const bf16x8*>(ok ? HbT + (long)tau * ldHT + c * 8 : zero);
  }
  __syncthreads();
  LMUFFT_STAMP(3, 1);

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;

  const bool brow_ok = b0 + r < B;
  const bf16_t* gbase = gm + (long)(b0 + r) * T * M;
  auto load_frame = [&](bf16x8 (&a)[KS], int t) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 16 + 8 * hh;
      a[ks] = *reinterpret_cast<const bf16x8*>((brow_ok && k < M) ? gbase + (long)t * M + k : zero);
    }
  };
  bf16x8 cur[KS], nxt[KS];
  int t = ta + wave;
  if (t < tb) load_frame(cur, t);
  for (; t < tb; t += 4) {
    const bool more = t + 4 < tb;
    if (more) load_frame(nxt, t + 4);
    const int wrow = t - ta + (kCb2St - 1) - r;  // window row of tau = t - s for s = s0 + r
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 16 + 8 * hh;
      const bf16x8 b0f = *reinterpret_cast<const bf16x8*>(&window[wrow * LDW + k]);
      const bf16x8 b1f = *reinterpret_cast<const bf16x8*>(&window[(wrow - 32) * LDW + k]);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[ks], b0f, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[ks], b1f, acc[1], 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) cur[ks] = nxt[ks];
    }
  }

  LMUFFT_STAMP(3, 2);
  // wave partials -> LDS -> ordered sum -> slab p
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int bl = (i & 3) + 8 * (i >> 2) + 4 * hh;
      red[(wave * kCb2Bt + bl) * kCb2St + tn * 32 + r] = acc[tn][i];
    }
  __syncthreads();
  float* slab = part + (long)p * B * T;
  for (int idx = threadIdx.x; idx < kCb2Bt * kCb2St; idx += kThreads) {
    const int bl = idx / kCb2St, sl = idx % kCb2St;
    const float v = ((red[(0 * kCb2Bt + bl) * kCb2St + sl] + red[(1 * kCb2Bt + bl) * kCb2St + sl]) +
                     red[(2 * kCb2Bt + bl) * kCb2St + sl]) +
                    red[(3 * kCb2Bt + bl) * kCb2St + sl];
    if (b0 + bl < B && s0 + sl < T) slab[(long)(b0 + bl) * T + s0 + sl] = v;
  }
  LMUFFT_STAMP(3, 3);
  LMUFFT_STAMP_ADD(3, 4, s0 - g_stamps[stamp_block() * 8 + 4]);
}

// ---------------------------------------------------------------------------
// gW_h[n][j] = sum_rows gz[row][n] [m;x][row][j]
// k = frame index: both operands are k-strided in HBM -> transposed LDS reads.
// The row range is split over workgroups; each row range writes one fp32 slab
//   [ gW_h : Hd*(M+D) | gb_h : Hd | gw_u : D | gb_u : 1 ]   (segments padded to 4)
// whose small segments come from the staged tiles for free:
//   gb_h[n] = sum_rows gz[row][n]        column sums of the A tile  (first j tile)
//   gw_u[d] = sum_rows ga[row] x[row][d] ga-weighted column sums of the B tile (first n tile)
//   gb_u    = sum_rows ga[row]
// The slabs are summed afterwards in a fixed order.
// ---------------------------------------------------------------------------
using WgradTileB = TileB<128, 192, 64, 2, 2>;
static_assert(WgradTileB::BK <= 64, "one wave stages the ga slice of a k tile");

template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_wgrad_bf16(const bf16_t* __restrict__ gz,
                                                         const bf16_t* __restrict__ m,
                                                         const bf16_t* __restrict__ x,
                                                         const float* __restrict__ ga, float* __restrict__ slabs,
                                                         int rows, int rows_per_split, int M, int D, int Hd,
                                                         int tile_blocks = 1) {
  using TL = WgradTileB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int MD = M + D;
  LMUFFT_STAMP(1, 0);
  LMUFFT_STAMP_ADD(1, 4, -g_stamps[stamp_block() * 8 + 4]);
  LMUFFT_STAMP_ADD(1, 5, -g_stamps[stamp_block() * 8 + 5]);
  LMUFFT_STAMP_ADD(1, 6, -g_stamps[stamp_block() * 8 + 6]);
  // all output tiles of one row range share its slices of gz and [m;x]: same XCD.  When there are more tiles
  // than an XCD should take of one range (tile_blocks > 1), a range's tiles form that many groups.
  const int tiles_j = (MD + TL::BN - 1) / TL::BN, tiles_n = (Hd + TL::BM - 1) / TL::BM;
  const int inner = tiles_j * tiles_n / tile_blocks;
  XcdSlot slot = xcd_slot(inner);
  slot.member += (slot.group % tile_blocks) * inner;
  slot.group /= tile_blocks;
  const int tile_j = slot.member % tiles_j, tile_n = slot.member / tiles_j;
  const int j0 = tile_j * TL::BN;
  const int n0 = tile_n * TL::BM;
  const int kbeg = slot.group * rows_per_split;
  if (kbeg >= rows) return;
  const int kend = min(rows, kbeg + rows_per_split);
  const long off_gbh = ((long)Hd * MD + 3) & ~3L, off_gwu = off_gbh + (long)tiles_j * TL::WGN * ((Hd + 3) & ~3),
             off_gbu = off_gwu + (long)tiles_n * TL::WGM * ((D + 3) & ~3);
  float* slab = slabs + slot.group * (off_gbu + 4);

  auto la = plain_load_b([=](int r, int k) -> bf16x8 {
    const int n = n0 + r;
    return bload8<VEC>(gz, (long)k * Hd + n, n, Hd, k < kend);
  });
  auto lb = plain_load_b([=](int r, int k) -> bf16x8 {
    const long row = k;
    const int j = j0 + r;
    if (VEC) {
      const bf16_t* p = j < M ? m + row * M + j : x + row * D + (j - M);
      p = (k < kend && j < MD) ? p : reinterpret_cast<const bf16_t*>(kZeroBytes);
      return *reinterpret_cast<const bf16x8*>(p);
    } else {
      if (k >= kend) return bzero8();
      bf16x8 v;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int jj = j + q;
        v[q] = jj < M ? m[row * M + jj] : (jj < MD ? x[row * D + (jj - M)] : (bf16_t)0.f);
      }
      return v;
    }
  });
  typedef LdsOperandTR<TL::BM, TL::BK, 32, decltype(la)> OpA;
  typedef LdsOperandTR<TL::BN, TL::BK, 32, decltype(lb)> OpB;
  OpA oa(smem, la);
  OpB ob(smem + OpA::ELEMS, lb);

  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  // The small gradients ride on the MFMA pipe:
  //   gb_h: A fragments times a fragment of ones;
  //   gw_u: a fragment whose 32 rows all hold ga[k] (split into two bf16 so that ga keeps
  //         16 mantissa bits) times the B fragments.
  // An A fragment is held by WGN waves of each of the tiles_j workgroups of a tile row, so
  // those RB = tiles_j * WGN waves take turns, one k-step of 16 each, and leave RB partial
  // copies of gb_h in the slab; likewise RU = tiles_n * WGM copies of gw_u.
  const int tid = threadIdx.x;
  const int RB = tiles_j * TL::WGN, RU = tiles_n * TL::WGM;
  const int slot_b = tile_j * TL::WGN + w.wn0 / TL::WN, slot_u = tile_n * TL::WGM + w.wm0 / TL::WM;
  int cnt_b = slot_b, cnt_u = slot_u;  // k-steps until this wave's next turn
  f32x16 accb[TL::TM], accu[TL::TN];
#pragma unroll
  for (int t = 0; t < TL::TM; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) accb[t][i] = 0.f;
#pragma unroll
  for (int t = 0; t < TL::TN; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) accu[t][i] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.f;
  // Pipeline over k-tiles: two LDS stages and two register sets.  In the round of tile i the wave
  // writes tile i + 1 from one register set to the other stage, refills that set with the global
  // loads of tile i + 3, then multiplies tile i -- so the loads of two k-tiles (80 KB per
  // workgroup) are in flight during every round and each has two rounds to land; with one tile in
  // flight the loop ran at the pace of one memory round trip per k-tile.  (Spreading the loads
  // between the k-steps' MFMAs does not help: a wave that waits for the vector-memory queue to
  // take its load cannot issue MFMAs either, so the issue time is paid wherever the loads stand.)  The loop is unrolled by
  // two so that the register sets are named statically.  One barrier per k-tile.
  // With VEC the staging addresses are kept as running pointers (a thread's vectors keep their
  // columns from tile to tile; only the row advances), so a round spends a handful of VALU
  // instructions per vector on addressing instead of re-deriving rows and panels.
  constexpr int kStageElems = OpA::ELEMS + OpB::ELEMS + 2 * TL::BK;  // + the ga slice (BK floats)
  auto stage_A = [&](int st) { return smem + st * kStageElems; };
  auto stage_B = [&](int st) { return smem + st * kStageElems + OpA::ELEMS; };
  auto stage_ga = [&](int st) { return reinterpret_cast<float*>(smem + st * kStageElems + OpA::ELEMS + OpB::ELEMS); };
  bf16x8 ra[2][OpA::NV], rb[2][OpB::NV];
  float ga_reg[2] = {0.f, 0.f}, gbu = 0.f;
  const bf16_t* const zero_line = reinterpret_cast<const bf16_t*>(kZeroBytes);
  const bf16_t* pa[OpA::NV];
  const bf16_t* pb[OpB::NV];
  int inc_b[OpB::NV];
  unsigned col_ok = 0;  // bit i: A vector i inside Hd; bit 8 + i: B vector i inside M + D
  if (VEC) {
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) {
      const int idx = tid + i * kThreads;
      const int c = n0 + (idx % (TL::BM / 8)) * 8, kr = idx / (TL::BM / 8);
      pa[i] = gz + (long)(kbeg + kr) * Hd + c;
      col_ok |= (c < Hd ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < OpB::NV; ++i) {
      const int idx = tid + i * kThreads;
      const int j = j0 + (idx % (TL::BN / 8)) * 8, kr = idx / (TL::BN / 8);
      pb[i] = j < M ? m + (long)(kbeg + kr) * M + j : x + (long)(kbeg + kr) * D + (j - M);
      inc_b[i] = TL::BK * (j < M ? M : D);
      col_ok |= (j < MD ? 1u : 0u) << (8 + i);
    }
  }
  // tiles are fetched in order (with VEC the pointers stand at tile k0); rows at or beyond kend
  // come from the zero line, so rounds always store and fetch and the staging has no branches
  auto fetch = [&](auto set_c, int k0) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) {
      const int idx = tid + i * kThreads;
      const int kr = idx / (TL::BM / 8);
      if (VEC) {
        const bf16_t* p = (((col_ok >> i) & 1u) && k0 + kr < kend) ? pa[i] : zero_line;
        ra[set][i] = *reinterpret_cast<const bf16x8*>(p);
        pa[i] += (long)TL::BK * Hd;
      } else {
        ra[set][i] = la.fetch((idx % (TL::BM / 8)) * 8, k0 + kr);
      }
    }
#pragma unroll
    for (int i = 0; i < OpB::NV; ++i) {
      const int idx = tid + i * kThreads;
      const int kr = idx / (TL::BN / 8);
      if (VEC) {
        const bf16_t* p = (((col_ok >> (8 + i)) & 1u) && k0 + kr < kend) ? pb[i] : zero_line;
        rb[set][i] = *reinterpret_cast<const bf16x8*>(p);
        pb[i] += inc_b[i];
      } else {
        rb[set][i] = lb.fetch((idx % (TL::BN / 8)) * 8, k0 + kr);
      }
    }
    if (tid < TL::BK) ga_reg[set] = (ga && k0 + tid < kend) ? ga[k0 + tid] : 0.f;
  };
  auto store = [&](auto set_c, int st) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
    bf16_t* sa = stage_A(st);
    bf16_t* sb = stage_B(st);
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i) {
      const int idx = tid + i * kThreads;
      *reinterpret_cast<bf16x8*>(&sa[(idx / (TL::BM / 8)) * OpA::LD + (idx % (TL::BM / 8)) * 8]) = ra[set][i];
    }
#pragma unroll
    for (int i = 0; i < OpB::NV; ++i) {
      const int idx = tid + i * kThreads;
      *reinterpret_cast<bf16x8*>(&sb[(idx / (TL::BN / 8)) * OpB::LD + (idx % (TL::BN / 8)) * 8]) = rb[set][i];
    }
    if (tid < TL::BK) {
      stage_ga(st)[tid] = ga_reg[set];
      gbu += ga_reg[set];
    }
  };
  // the part of a tile's LDS stores that is issued during k-step ks: the stores of a whole tile in one go
  // (40 KB through the LDS store path) stood in front of the round's first fragment reads
  auto store_part = [&](auto set_c, int st, int ks) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
    constexpr int KSN = TL::BK / 16;
    bf16_t* sa = stage_A(st);
    bf16_t* sb = stage_B(st);
#pragma unroll
    for (int i = 0; i < OpA::NV; ++i)
      if (i % KSN == ks) {
        const int idx = tid + i * kThreads;
        *reinterpret_cast<bf16x8*>(&sa[(idx / (TL::BM / 8)) * OpA::LD + (idx % (TL::BM / 8)) * 8]) = ra[set][i];
      }
#pragma unroll
    for (int i = 0; i < OpB::NV; ++i)
      if (i % KSN == ks) {
        const int idx = tid + i * kThreads;
        *reinterpret_cast<bf16x8*>(&sb[(idx / (TL::BN / 8)) * OpB::LD + (idx % (TL::BN / 8)) * 8]) = rb[set][i];
      }
    if (ks == KSN - 1 && tid < TL::BK) {
      stage_ga(st)[tid] = ga_reg[set];
      gbu += ga_reg[set];
    }
  };
  auto read_frags = [&](bf16x8 (&af)[TL::TM], bf16x8 (&bfr)[TL::TN], int kl) {
#pragma unroll
    for (int tm = 0; tm < TL::TM; ++tm) af[tm] = oa.frag32(w.wm0 + tm * 32, kl);
#pragma unroll
    for (int tn = 0; tn < TL::TN; ++tn) bfr[tn] = ob.frag32(w.wn0 + tn * 32, kl);
  };
  // round of the tile at k0, which sits in stage st; the tile at k0 + BK waits in register set
  // `set_c` and that set is refilled with the tile at k0 + 3 BK
  auto round = [&](auto set_c, int k0, int st) __attribute__((always_inline)) {
    [[maybe_unused]] const long long c0 = LMUFFT_CLOCK();
    oa.S = stage_A(st);
    ob.S = stage_B(st);
    const float* gaS = stage_ga(st);
    bf16x8 fa[2][TL::TM], fb[2][TL::TN];
    read_frags(fa[0], fb[0], 0);
    [[maybe_unused]] const long long c1 = LMUFFT_CLOCK();
#pragma unroll
    for (int ks = 0; ks < TL::BK / 16; ++ks) {
      const int kl = ks * 16;
      bf16x8(&af)[TL::TM] = fa[ks & 1];
      bf16x8(&bfr)[TL::TN] = fb[ks & 1];
      if (ks + 1 < TL::BK / 16) read_frags(fa[(ks + 1) & 1], fb[(ks + 1) & 1], kl + 16);
      store_part(set_c, st ^ 1, ks);  // that stage was last read before the previous barrier
#pragma unroll
      for (int tm = 0; tm < TL::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TL::TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
      if (cnt_b == 0) {  // wave-uniform
        cnt_b = RB;
#pragma unroll
        for (int tm = 0; tm < TL::TM; ++tm)
          accb[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], ones, accb[tm], 0, 0, 0);
      }
      --cnt_b;
      if (cnt_u == 0) {  // wave-uniform
        cnt_u = RU;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gaS[kl + 8 * w.hh]);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(&gaS[kl + 8 * w.hh + 4]);
        bf16x8 ghi, glo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gv = j < 4 ? g0[j] : g1[j - 4];
          ghi[j] = (bf16_t)gv;
          glo[j] = (bf16_t)(gv - (float)ghi[j]);
        }
#pragma unroll
        for (int tn = 0; tn < TL::TN; ++tn) {
          accu[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ghi, bfr[tn], accu[tn], 0, 0, 0);
        }
#pragma unroll
        for (int tn = 0; tn < TL::TN; ++tn) {
          accu[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(glo, bfr[tn], accu[tn], 0, 0, 0);
        }
      }
      --cnt_u;
    }
    fetch(set_c, k0 + 3 * TL::BK);  // the set's registers are free once its tile has gone to LDS
    [[maybe_unused]] const long long c2 = LMUFFT_CLOCK();
    __syncthreads();
    [[maybe_unused]] const long long c3 = LMUFFT_CLOCK();
    LMUFFT_STAMP_ADD(1, 4, c2 - c1);
    LMUFFT_STAMP_ADD(1, 5, c1 - c0);
    LMUFFT_STAMP_ADD(1, 6, c3 - c2);
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  fetch(Set0{}, kbeg);
  fetch(Set1{}, kbeg + TL::BK);
  store(Set0{}, 0);
  fetch(Set0{}, kbeg + 2 * TL::BK);
  __syncthreads();
  LMUFFT_STAMP(1, 1);
  int k0 = kbeg;
  for (; k0 + TL::BK < kend; k0 += 2 * TL::BK) {
    round(Set1{}, k0, 0);
    round(Set0{}, k0 + TL::BK, 1);
  }
  if (k0 < kend) round(Set1{}, k0, 0);  // odd number of tiles
  LMUFFT_STAMP(1, 2);

  // gW_h tile -> slab through LDS: 16-byte stores of whole row segments (the operand area is free
  // after the loop's last barrier; the launch reserves BM * (BN + 4) floats)
  {
    constexpr int LDT = TL::BN + 4;
    float* tile = reinterpret_cast<float*>(smem);
    tileb_epilogue<TL>(acc, w, [&](int mm, int nn, float v) { tile[mm * LDT + nn] = v; });
    __syncthreads();
    const bool vec_out = (MD % 4 == 0);
    for (int idx = tid; idx < TL::BM * (TL::BN / 4); idx += kThreads) {
      const int mm = idx / (TL::BN / 4), c = (idx % (TL::BN / 4)) * 4;
      const int n = n0 + mm, j = j0 + c;
      if (n >= Hd) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[mm * LDT + c]);
      float* dst = slab + (long)n * MD + j;
      if (vec_out && j + 3 < MD) {
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (j + q < MD) dst[q] = v[q];
      }
    }
  }
  if (w.r == 0) {  // every column of accb holds the row sums; column 0 writes them
    float* dst = slab + off_gbh + (long)slot_b * ((Hd + 3) & ~3);
#pragma unroll
    for (int tm = 0; tm < TL::TM; ++tm)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + w.wm0 + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * w.hh;
        if (n < Hd) dst[n] = accb[tm][i];
      }
  }
  if (w.hh == 0) {  // every row of accu holds gw_u; row 0 (register 0, lanes 0..31) writes it
    float* dst = slab + off_gwu + (long)slot_u * ((D + 3) & ~3);
#pragma unroll
    for (int tn = 0; tn < TL::TN; ++tn) {
      const int j = j0 + w.wn0 + tn * 32 + w.r;
      if (j >= M && j < MD) dst[j - M] = accu[tn][0];
    }
  }
  if (tile_n == 0 && tile_j == 0 && tid < 64) {  // gb_u = sum of ga over the row range
    float v = gbu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (tid == 0) slab[off_gbu] = v;
  }
  LMUFFT_STAMP(1, 3);
}

// ---------------------------------------------------------------------------
// The same product and the same slabs on eight waves (used when every row is 16-byte friendly).
// k_wgrad_bf16 runs one wave per SIMD -- its two operand stages fill the LDS a workgroup can have, and a second
// workgroup per CU would double the slabs -- so nobody covers a wave's LDS and memory latency.  Here a workgroup is
// two groups of four waves; a group has the tile and wave layout of k_wgrad_bf16 (a wave owns 64 x 96), its own two
// LDS stages of 32 rows, its own two register sets, and takes every other 32-row k-tile of the workgroup's row range
// into accumulators of its own.  The fragment reads per MFMA are those of the four-wave kernel (eight waves on ONE
// staged tile would read four fragments for three MFMAs instead of five for six, and the transposing reads are the
// limit); the second wave on a SIMD comes from the k split instead.  After the loop the second group's sums pass
// through LDS and are added to the first group's, which writes the slab.
// The small gradients are VALU sums over the fragments a wave holds anyway, again in turns of one k-step per wave:
//   gb_h: the eight k values of an A fragment lane;  gw_u: the eight k values of a B fragment lane times ga[k]
// (five registers instead of the 80 accumulator registers of the MFMA form, which would not leave two waves a SIMD's
// register file; ga enters in fp32).
// ---------------------------------------------------------------------------
constexpr int kWgPairBK = 32;
constexpr int kWgPairThreads = 512;
using WgradPairTile = TileB<WgradTileB::BM, WgradTileB::BN, kWgPairBK, WgradTileB::WGM, WgradTileB::WGN>;
constexpr int kWgPairLdA = WgradPairTile::BM + 32, kWgPairLdB = WgradPairTile::BN + 32;
constexpr int kWgPairStageElems = kWgPairBK * (kWgPairLdA + kWgPairLdB) + 2 * kWgPairBK;  // A | B | ga (BK floats)
constexpr int kWgPairExFloats = WgradPairTile::TM * WgradPairTile::TN * 16 + WgradPairTile::TM + WgradPairTile::TN + 1;
constexpr size_t kWgPairLds =
    std::max(std::max(sizeof(bf16_t) * 4 * (size_t)kWgPairStageElems, sizeof(float) * 256 * (size_t)kWgPairExFloats),
             sizeof(float) * (size_t)WgradPairTile::BM * (WgradPairTile::BN + 4));
static_assert((kWgPairLdA * 2) % 256 == 64 && (kWgPairLdB * 2) % 256 == 192, "four k rows land on distinct bank quarters");
static_assert(WgradTileB::BK == 2 * kWgPairBK, "a row range is a whole number of tile pairs");

__global__ __launch_bounds__(kWgPairThreads) void k_wgrad_pair_bf16(const bf16_t* __restrict__ gz,
                                                                    const bf16_t* __restrict__ m,
                                                                    const bf16_t* __restrict__ x,
                                                                    const float* __restrict__ ga,
                                                                    float* __restrict__ slabs, int rows,
                                                                    int rows_per_split, int M, int D, int Hd,
                                                                    int tile_blocks) {
  using TL = WgradPairTile;
  constexpr int GT = 256;                                     // threads of a group
  constexpr int NVA = TL::BM * TL::BK / 8 / GT, NVB = TL::BN * TL::BK / 8 / GT;  // staging vectors per thread
  constexpr int KSN = TL::BK / 16;
  constexpr int LDA = kWgPairLdA, LDB = kWgPairLdB;
  constexpr int kElemsA = TL::BK * LDA;
  static_assert(NVA * GT * 8 == TL::BM * TL::BK && NVB * GT * 8 == TL::BN * TL::BK, "whole vectors per thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int MD = M + D;
  const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  const int tid = threadIdx.x & (GT - 1), lane = threadIdx.x & 63;
  const int tiles_j = (MD + TL::BN - 1) / TL::BN, tiles_n = (Hd + TL::BM - 1) / TL::BM;
  const int inner = tiles_j * tiles_n / tile_blocks;
  XcdSlot slot = xcd_slot(inner);
  slot.member += (slot.group % tile_blocks) * inner;
  slot.group /= tile_blocks;
  const int tile_j = slot.member % tiles_j, tile_n = slot.member / tiles_j;
  const int j0 = tile_j * TL::BN;
  const int n0 = tile_n * TL::BM;
  const int kbeg = slot.group * rows_per_split;
  if (kbeg >= rows) return;
  const int kend = min(rows, kbeg + rows_per_split);
  const long off_gbh = ((long)Hd * MD + 3) & ~3L, off_gwu = off_gbh + (long)tiles_j * TL::WGN * ((Hd + 3) & ~3),
             off_gbu = off_gwu + (long)tiles_n * TL::WGM * ((D + 3) & ~3);
  float* slab = slabs + slot.group * (off_gbu + 4);

  WavePos<TL> w;  // position inside the group
  {
    const int gw = (tid >> 6);
    w.wm0 = (gw / TL::WGN) * TL::WM;
    w.wn0 = (gw % TL::WGN) * TL::WN;
  }
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  float accb[TL::TM], accu[TL::TN];
#pragma unroll
  for (int t = 0; t < TL::TM; ++t) accb[t] = 0.f;
#pragma unroll
  for (int t = 0; t < TL::TN; ++t) accu[t] = 0.f;
  // turns: the k-step with global number gk belongs to the wave with slot gk % RB (gb_h) / gk % RU (gw_u) among the
  // waves that hold the same fragment; this group's tiles are the 32-row tiles 2 i + grp of the row range
  const int RB = tiles_j * TL::WGN, RU = tiles_n * TL::WGM;
  const int slot_b = tile_j * TL::WGN + w.wn0 / TL::WN, slot_u = tile_n * TL::WGM + w.wm0 / TL::WM;
  int pos_b = (grp * KSN) % RB, pos_u = (grp * KSN) % RU;

  bf16_t* const gsmem = smem + grp * 2 * kWgPairStageElems;
  auto stage_A = [&](int st) { return gsmem + st * kWgPairStageElems; };
  auto stage_B = [&](int st) { return gsmem + st * kWgPairStageElems + kElemsA; };
  auto stage_ga = [&](int st) { return reinterpret_cast<float*>(gsmem + st * kWgPairStageElems + kElemsA + TL::BK * LDB); };
  // fragment addressing of ds_read_b64_tr_b16: in each 16-lane group lane 4 q + p addresses k row q, elements
  // 4 p .. 4 p + 3, and receives the four k values of column (lane & 15)
  const int fq = (lane & 15) >> 2, fp = lane & 3, fg1 = (lane >> 4) & 1;
  const int foffA = (8 * w.hh + fq) * LDA + 16 * fg1 + 4 * fp, foffB = (8 * w.hh + fq) * LDB + 16 * fg1 + 4 * fp;
  auto frag = [&](const bf16_t* a0, int ld) -> bf16x8 {
    typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(a0 + 4 * ld));
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = lo[j];
      v[4 + j] = hi[j];
    }
    return v;
  };

  // staging: running pointers, rows at or beyond kend come from the zero line (see k_wgrad_bf16)
  bf16x8 ra[2][NVA], rb[2][NVB];
  float ga_reg[2] = {0.f, 0.f}, gbu = 0.f;
  const bf16_t* const zero_line = reinterpret_cast<const bf16_t*>(kZeroBytes);
  const bf16_t* pa[NVA];
  const bf16_t* pb[NVB];
  int inc_b[NVB];
  unsigned col_ok = 0;  // bit i: A vector i inside Hd; bit 8 + i: B vector i inside M + D
  const int kg = kbeg + grp * TL::BK;  // the group's first tile; its tile i starts at kg + 2 BK i
#pragma unroll
  for (int i = 0; i < NVA; ++i) {
    const int idx = tid + i * GT;
    const int c = n0 + (idx % (TL::BM / 8)) * 8, kr = idx / (TL::BM / 8);
    pa[i] = gz + (long)(kg + kr) * Hd + c;
    col_ok |= (c < Hd ? 1u : 0u) << i;
  }
#pragma unroll
  for (int i = 0; i < NVB; ++i) {
    const int idx = tid + i * GT;
    const int j = j0 + (idx % (TL::BN / 8)) * 8, kr = idx / (TL::BN / 8);
    pb[i] = j < M ? m + (long)(kg + kr) * M + j : x + (long)(kg + kr) * D + (j - M);
    inc_b[i] = 2 * TL::BK * (j < M ? M : D);
    col_ok |= (j < MD ? 1u : 0u) << (8 + i);
  }
  auto fetch = [&](auto set_c, int k0) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
      const int kr = (tid + i * GT) / (TL::BM / 8);
      const bf16_t* p = (((col_ok >> i) & 1u) && k0 + kr < kend) ? pa[i] : zero_line;
      ra[set][i] = *reinterpret_cast<const bf16x8*>(p);
      pa[i] += (long)2 * TL::BK * Hd;
    }
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
      const int kr = (tid + i * GT) / (TL::BN / 8);
      const bf16_t* p = (((col_ok >> (8 + i)) & 1u) && k0 + kr < kend) ? pb[i] : zero_line;
      rb[set][i] = *reinterpret_cast<const bf16x8*>(p);
      pb[i] += inc_b[i];
    }
    if (tid < TL::BK) ga_reg[set] = (ga && k0 + tid < kend) ? ga[k0 + tid] : 0.f;
  };
  // the part of a tile's LDS stores that is issued during k-step ks (ks < 0: all of them)
  auto store_part = [&](auto set_c, int st, int ks) __attribute__((always_inline)) {
    constexpr int set = decltype(set_c)::value;
    bf16_t* sa = stage_A(st);
    bf16_t* sb = stage_B(st);
#pragma unroll
    for (int i = 0; i < NVA; ++i)
      if (ks < 0 || i % KSN == ks) {
        const int idx = tid + i * GT;
        *reinterpret_cast<bf16x8*>(&sa[(idx / (TL::BM / 8)) * LDA + (idx % (TL::BM / 8)) * 8]) = ra[set][i];
      }
#pragma unroll
    for (int i = 0; i < NVB; ++i)
      if (ks < 0 || i % KSN == ks) {
        const int idx = tid + i * GT;
        *reinterpret_cast<bf16x8*>(&sb[(idx / (TL::BN / 8)) * LDB + (idx % (TL::BN / 8)) * 8]) = rb[set][i];
      }
    if ((ks < 0 || ks == KSN - 1) && tid < TL::BK) {
      stage_ga(st)[tid] = ga_reg[set];
      gbu += ga_reg[set];
    }
  };
  // round of the tile at k0, which sits in stage st; the group's next tile waits in register set `set_c`, and
  // that set is refilled with the tile three ahead
  auto round = [&](auto set_c, int k0, int st) __attribute__((always_inline)) {
    const bf16_t* sA = stage_A(st) + foffA + w.wm0;
    const bf16_t* sB = stage_B(st) + foffB + w.wn0;
    const float* gaS = stage_ga(st);
    bf16x8 fa[2][TL::TM], fb[2][TL::TN];
    auto read_frags = [&](int buf, int kl) __attribute__((always_inline)) {
#pragma unroll
      for (int tm = 0; tm < TL::TM; ++tm) fa[buf][tm] = frag(sA + kl * LDA + tm * 32, LDA);
#pragma unroll
      for (int tn = 0; tn < TL::TN; ++tn) fb[buf][tn] = frag(sB + kl * LDB + tn * 32, LDB);
    };
    read_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
      const int kl = ks * 16;
      bf16x8(&af)[TL::TM] = fa[ks & 1];
      bf16x8(&bfr)[TL::TN] = fb[ks & 1];
      if (ks + 1 < KSN) read_frags((ks + 1) & 1, kl + 16);
      store_part(set_c, st ^ 1, ks);  // that stage was last read before the previous barrier
#pragma unroll
      for (int tm = 0; tm < TL::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TL::TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
      if (pos_b == slot_b) {  // wave-uniform
#pragma unroll
        for (int tm = 0; tm < TL::TM; ++tm) {
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) sum += (float)af[tm][j];
          accb[tm] += sum;
        }
      }
      pos_b = pos_b + 1 == RB ? 0 : pos_b + 1;
      if (pos_u == slot_u) {  // wave-uniform
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gaS[kl + 8 * w.hh]);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(&gaS[kl + 8 * w.hh + 4]);
#pragma unroll
        for (int tn = 0; tn < TL::TN; ++tn) {
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) sum += (j < 4 ? g0[j] : g1[j - 4]) * (float)bfr[tn][j];
          accu[tn] += sum;
        }
      }
      pos_u = pos_u + 1 == RU ? 0 : pos_u + 1;
    }
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {  // the other group's tile
      pos_b = pos_b + 1 == RB ? 0 : pos_b + 1;
      pos_u = pos_u + 1 == RU ? 0 : pos_u + 1;
    }
    fetch(set_c, k0 + 3 * 2 * TL::BK);  // the set's registers are free once its tile has gone to LDS
    __syncthreads();
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  const int ntile = (kend - kbeg + TL::BK - 1) / TL::BK;  // 32-row tiles of the range
  const int nround = (ntile + 1) / 2;                     // both groups run every round (tiles past kend are zeros)
  fetch(Set0{}, kg);
  fetch(Set1{}, kg + 2 * TL::BK);
  store_part(Set0{}, 0, -1);
  fetch(Set0{}, kg + 4 * TL::BK);
  __syncthreads();
  {
    int i = 0;
    for (; i + 1 < nround; i += 2) {
      round(Set1{}, kg + 2 * TL::BK * i, 0);
      round(Set0{}, kg + 2 * TL::BK * (i + 1), 1);
    }
    if (i < nround) round(Set1{}, kg + 2 * TL::BK * i, 0);  // odd number of rounds
  }

  // the second group's sums -> LDS -> added to the first group's (the operand stages are free after the loop's last
  // barrier); value v of thread tid sits at ex[v * 256 + tid]
  float* ex = reinterpret_cast<float*>(smem_raw);
  if (grp == 1) {
#pragma unroll
    for (int tm = 0; tm < TL::TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TL::TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) ex[((tm * TL::TN + tn) * 16 + i) * GT + tid] = acc[tm][tn][i];
    constexpr int base = TL::TM * TL::TN * 16;
#pragma unroll
    for (int tm = 0; tm < TL::TM; ++tm) ex[(base + tm) * GT + tid] = accb[tm];
#pragma unroll
    for (int tn = 0; tn < TL::TN; ++tn) ex[(base + TL::TM + tn) * GT + tid] = accu[tn];
    ex[(base + TL::TM + TL::TN) * GT + tid] = gbu;
  }
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int tm = 0; tm < TL::TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TL::TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tm][tn][i] += ex[((tm * TL::TN + tn) * 16 + i) * GT + tid];
    constexpr int base = TL::TM * TL::TN * 16;
#pragma unroll
    for (int tm = 0; tm < TL::TM; ++tm) accb[tm] += ex[(base + tm) * GT + tid];
#pragma unroll
    for (int tn = 0; tn < TL::TN; ++tn) accu[tn] += ex[(base + TL::TM + tn) * GT + tid];
    gbu += ex[(base + TL::TM + TL::TN) * GT + tid];
  }
  __syncthreads();

  // gW_h tile -> slab through LDS: 16-byte stores of whole row segments, by all eight waves
  {
    constexpr int LDT = TL::BN + 4;
    float* tile = reinterpret_cast<float*>(smem_raw);
    if (grp == 0) tileb_epilogue<TL>(acc, w, [&](int mm, int nn, float v) { tile[mm * LDT + nn] = v; });
    __syncthreads();
    const bool vec_out = (MD % 4 == 0);
    for (int idx = threadIdx.x; idx < TL::BM * (TL::BN / 4); idx += kWgPairThreads) {
      const int mm = idx / (TL::BN / 4), c = (idx % (TL::BN / 4)) * 4;
      const int n = n0 + mm, j = j0 + c;
      if (n >= Hd) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[mm * LDT + c]);
      float* dst = slab + (long)n * MD + j;
      if (vec_out && j + 3 < MD) {
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (j + q < MD) dst[q] = v[q];
      }
    }
  }
  if (grp == 0) {
    // a lane's sums cover the k values 8 hh .. 8 hh + 7 of every k-step: the two halves of the wave are added here
    float* dstb = slab + off_gbh + (long)slot_b * ((Hd + 3) & ~3);
#pragma unroll
    for (int tm = 0; tm < TL::TM; ++tm) {
      const float v = accb[tm] + __shfl_xor(accb[tm], 32);
      const int n = n0 + w.wm0 + tm * 32 + w.r;
      if (w.hh == 0 && n < Hd) dstb[n] = v;
    }
    float* dstu = slab + off_gwu + (long)slot_u * ((D + 3) & ~3);
#pragma unroll
    for (int tn = 0; tn < TL::TN; ++tn) {
      const float v = accu[tn] + __shfl_xor(accu[tn], 32);
      const int j = j0 + w.wn0 + tn * 32 + w.r;
      if (w.hh == 0 && j >= M && j < MD) dstu[j - M] = v;
    }
    if (tile_n == 0 && tile_j == 0 && tid < 64) {  // gb_u = sum of ga over the row range
      float v = gbu;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (tid == 0) slab[off_gbu] = v;
    }
  }
}

#ifdef LMUFFT_STAMPS
extern "C" int lmufft_debug_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * std::min(n, 8 * 4096));
}
#endif


struct HprepLayout {
  int ldHb, ldHT;
  size_t off_Hb, off_HbT, total;
};
static HprepLayout hprep_layout(int M, int ldH) {
  HprepLayout l;
  l.ldHb = up8(ldH);
  l.ldHT = up8(M);
  l.off_Hb = 0;
  l.off_HbT = align256b(sizeof(bf16_t) * (size_t)M * l.ldHb);
  l.total = l.off_HbT + align256b(sizeof(bf16_t) * (size_t)l.ldHb * l.ldHT);
  return l;
}

struct WprepLayout {
  int ldW, ldWT;
  size_t off_Wb, off_WbT, total;
};
static WprepLayout wprep_layout(const lmufft_shape& s) {
  WprepLayout l;
  const int MD = s.M + s.D;
  l.ldW = up8(MD);
  l.ldWT = up8(s.Hd);
  l.off_Wb = 0;
  l.off_WbT = align256b(sizeof(bf16_t) * (size_t)s.Hd * l.ldW);
  l.total = l.off_WbT + align256b(sizeof(bf16_t) * (size_t)MD * l.ldWT);
  return l;
}


struct BwdPlanB {
  int S, rps, ntt, cb2_P;
  long slab_floats;
  size_t off_gz, off_wslab, off_gupart, off_ga, off_scan, scan_bytes, off_q, total;
  int q_rows, q_ld, q_slots;  // Q route: rows and row stride of Q^T, (frame tile, lag chunk) pairs per utterance
};
// The Q route (k_gu_q_bf16) takes sequences of up to kGuQMaxT frames with 16-byte friendly sizes.
static bool q_route_shape_ok(const lmufft_shape& s) {
  return s.T <= kGuQMaxT && s.M % 8 == 0 && s.Hd % GuQTile::BK == 0 && s.Hd / 8 <= kThreads;  // whole k-tiles
}
// slab of the weight-gradient kernel: gW_h | RB copies of gb_h | RU copies of gw_u | gb_u
static SlabOutputs wgrad_outputs(const lmufft_shape& s, float* gW_h, float* gb_h, float* gw_u, float* gb_u) {
  using TL = WgradTileB;
  const int MD = s.M + s.D;
  const int tiles_j = cdivb(MD, TL::BN), tiles_n = cdivb(s.Hd, TL::BM);
  return SlabOutputs{{gW_h, gb_h, gw_u, gb_u},
                     {(long)s.Hd * MD, s.Hd, s.D, 1},
                     {1, tiles_j * TL::WGN, tiles_n * TL::WGM, 1}};
}

static BwdPlanB plan_backward_b(const lmufft_shape& s) {
  BwdPlanB p;
  const long rows = (long)s.B * s.T;
  const int MD = s.M + s.D;
  const int tiles = cdivb(s.Hd, WgradTileB::BM) * cdivb(MD, WgradTileB::BN);
  // whole row ranges per XCD (32 CUs each), one workgroup per CU at most; every slab also costs a
  // write and a read of gW_h
  int S = 8 * std::max(1, 32 / tiles);
  int rps = cdivb(rows, S);
  rps = cdivb(rps, WgradTileB::BK) * WgradTileB::BK;
  p.S = cdivb(rows, rps);
  p.rps = rps;
  p.ntt = cdivb(s.T, ConvBwdTileB::BM);
  p.slab_floats = slab_stride(wgrad_outputs(s, nullptr, nullptr, nullptr, nullptr));
  size_t off = 0;
  p.off_gz = off;
  off += align256b(sizeof(bf16_t) * (size_t)rows * s.Hd);
  p.off_wslab = off;
  off += align256b(sizeof(float) * (size_t)p.S * (size_t)p.slab_floats);
  {  // second form of the adjoint convolution: P frame ranges per s tile
    const int tiles = cdivb(s.T, kCb2St) * cdivb(s.B, kCb2Bt);
    p.cb2_P = std::max(1, std::min(std::min(32, s.T), (768 + tiles - 1) / tiles));
    // ... and enough of them that a range's window of H^T (its frames + 63 rows of M + 8 bf16) fits the LDS beside the
    // kernel's 32 KB of partial sums: large batches of long sequences would otherwise ask for one or two ranges
    const long row_bytes = (long)(cdivb(s.M, 16) * 16 + 8) * sizeof(bf16_t);
    const long fit = (160L * 1024 - (long)sizeof(float) * 4 * kCb2Bt * kCb2St) / row_bytes - (kCb2St - 1);
    if (fit >= 1) p.cb2_P = std::max(p.cb2_P, cdivb(s.T, fit));
  }
  p.q_slots = guq_plan(s.T).nslot;
  p.q_rows = cdivb(s.T, 64) * 64;
  p.q_ld = up8(s.Hd);
  p.off_gupart = off;
  off += align256b(sizeof(float) * std::max(std::max((size_t)s.B * p.ntt * s.T, (size_t)p.cb2_P * s.B * s.T),
                                            (size_t)p.q_slots * s.B * s.T));
  p.off_ga = off;
  off += align256b(sizeof(float) * (size_t)rows);
  p.off_scan = off;  // scratch of the state-space route (slabs of its adjoint GEMM)
  p.scan_bytes = scan_workspace_bytes(s.B, s.T, s.M);
  off += align256b(p.scan_bytes);
  p.off_q = off;
  off += align256b(sizeof(bf16_t) * (size_t)p.q_rows * p.q_ld);
  p.total = off;
  return p;
}

size_t forward_workspace_bf16(const lmufft_shape& s) {
  return 256 + scan_forward_workspace_bytes(std::max(s.B, 1), std::max(s.T, 1), s.M);
}
size_t conv_workspace_bf16(const lmufft_shape& s0) {
  lmufft_shape s = s0;
  s.D = std::max(s.D, 1);
  s.Hd = std::max(s.Hd, 1);
  const BwdPlanB p = plan_backward_b(s);
  return p.off_ga - p.off_gupart;
}
size_t backward_workspace_bf16(const lmufft_shape& s) { return plan_backward_b(s).total; }

#define PROFB(name) ProfScope prof_scope_(name, st)
#define LAUNCH_CHECKB(name)                                                                       \
  do {                                                                                            \
    hipError_t e_ = hipGetLastError();                                                            \
    if (e_ != hipSuccess) return set_error("launch of %s failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

static int check_shape_b(const lmufft_shape* s) {
  if (!s) return set_error("shape is NULL");
  if (s->B <= 0 || s->T <= 0 || s->D <= 0 || s->M <= 0 || s->Hd <= 0)
    return set_error("shape: B, T, D, M, Hd must be positive (got B=%d T=%d D=%d M=%d Hd=%d)", s->B, s->T,
                     s->D, s->M, s->Hd);
  if (s->T > s->ldH)
    return set_error("sequence length %d exceeds the impulse response length (seq_len) %d", s->T, s->ldH);
  if ((long)s->B * s->T >= (1L << 31) / 4) return set_error("B*T too large");
  return 0;
}

}  // namespace lmufft

using namespace lmufft;

extern "C" {

size_t lmufft_hprep_bytes(const lmufft_shape* s) {
  if (!s || s->M <= 0 || s->ldH <= 0) return 0;
  return hprep_layout(s->M, s->ldH).total;
}

int lmufft_hprep_bf16(const lmufft_shape* s, const float* H, void* hprep, void* stream) {
  if (!s || s->M <= 0 || s->ldH <= 0) return set_error("hprep: bad shape");
  if (!H || !hprep) return set_error("hprep: NULL pointer");
  if (!al16(hprep)) return set_error("hprep: buffer must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const HprepLayout l = hprep_layout(s->M, s->ldH);
  bf16_t* Hb = (bf16_t*)((char*)hprep + l.off_Hb);
  bf16_t* HbT = (bf16_t*)((char*)hprep + l.off_HbT);
  const long n = (long)s->M * l.ldHb + (long)l.ldHb * l.ldHT;
  {
    PROFB("k_prep_H");
    hipLaunchKernelGGL(k_prep_H, dim3(min(cdivb(n, kThreads), 2048)), dim3(kThreads), 0, st, H, Hb, HbT, s->M,
                       s->ldH, l.ldHb, l.ldHT);
  }
  LAUNCH_CHECKB("k_prep_H");
  return 0;
}

size_t lmufft_wprep_bytes(const lmufft_shape* s) {
  if (check_shape_b(s)) return 0;
  return wprep_layout(*s).total;
}

static WeightPrep weight_prep(const lmufft_shape& s, const float* W_h, void* wprep, int* blocks) {
  const WprepLayout l = wprep_layout(s);
  const int MD = s.M + s.D;
  const long n = ((long)s.Hd * l.ldW + (long)MD * l.ldWT) / 8;  // eight elements per thread
  *blocks = min(cdivb(n, kThreads), 2048);
  return WeightPrep{W_h, (bf16_t*)((char*)wprep + l.off_Wb), (bf16_t*)((char*)wprep + l.off_WbT), s.Hd, MD, l.ldW,
                    l.ldWT};
}

int lmufft_wprep_bf16(const lmufft_shape* s, const float* W_h, void* wprep, void* stream) {
  if (int rc = check_shape_b(s)) return rc;
  if (!W_h || !wprep) return set_error("wprep: NULL pointer");
  if (!al16(wprep)) return set_error("wprep: buffer must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int blocks = 0;
  const WeightPrep q = weight_prep(*s, W_h, wprep, &blocks);
  {
    PROFB("k_prep_weights");
    hipLaunchKernelGGL(k_prep_weights, dim3(blocks), dim3(kThreads), 0, st, q);
  }
  LAUNCH_CHECKB("k_prep_weights");
  return 0;
}

}  // extern "C"

// m = causal convolution of u with H (bf16 out)
static int conv_forward_bf16_any(const lmufft_shape& sh, const float* u, const bf16_t* Hb, int ldHb, bf16_t* m,
                                 hipStream_t st) {
  const lmufft_shape* s = &sh;
  {
    using TL = ConvFwdTileB;
    const dim3 grid(cdivb(s->M, TL::BN), s->B, cdivb(s->T, TL::BM));
    const int kpad = cdivb(s->T, TL::BK) * TL::BK;
    const size_t lds = std::max(sizeof(bf16_t) * ((size_t)TL::BN * (TL::BK + 8) + (size_t)(kpad / 8 + TL::BM / 8) * 64) +
                                    sizeof(float) * (size_t)kpad,
                                sizeof(bf16_t) * (size_t)TL::BM * (TL::BN + 8));  // the epilogue's staging tile
    if (lds > 160 * 1024) return set_error("sequence too long for the Toeplitz forward kernel (T=%d)", s->T);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k_conv_fwd_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
      return set_error("cannot reserve %zu bytes of LDS for k_conv_fwd_bf16", lds);
    {
      PROFB("k_conv_fwd_bf16");
      hipLaunchKernelGGL(k_conv_fwd_bf16, grid, dim3(kThreads), lds, st, u, Hb, m, s->T, s->M, ldHb);
    }
    LAUNCH_CHECKB("k_conv_fwd_bf16");
  }
  return 0;
}

// The state-space route runs when it is asked for, and under AUTO whenever the caller hands in `scanprep`.
static int use_scan_route(int conv_algo, const void* scanprep, bool* scan) {
  if (conv_algo != LMUFFT_CONV_AUTO && conv_algo != LMUFFT_CONV_TOEPLITZ && conv_algo != LMUFFT_CONV_SCAN)
    return set_error("conv algorithm %d is not available for bf16", conv_algo);
  if (conv_algo == LMUFFT_CONV_SCAN && !scanprep) return set_error("conv algorithm SCAN needs scanprep");
  *scan = conv_algo == LMUFFT_CONV_SCAN || (conv_algo == LMUFFT_CONV_AUTO && scanprep);
  return 0;
}

// hmask[row][c] bit j = h[row][8 c + j] > 0 from a finished h (shapes the LDS-DMA projection kernel does not take)
__global__ __launch_bounds__(kThreads) void k_hmask_from_h(const bf16_t* __restrict__ h, unsigned char* __restrict__ hmask,
                                                           long chunks) {
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < chunks; i += (long)gridDim.x * kThreads) {
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) bits |= ((float)h[i * 8 + j] > 0.f ? 1u : 0u) << j;
    hmask[i] = (unsigned char)bits;
  }
}

// CUs of the current device (cached per device)
static int compute_units() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cached[dev]) {
    int n = 0;
    cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cached[dev];
}

static int forward_bf16_impl(const lmufft_shape* s, const void* x_, const float* w_u, const float* b_u, void* wprep,
                             const float* W_h, const float* b_h, const void* hprep, const void* scanprep, float* u,
                             void* m_, void* h_, void* h_n_, void* hmask, int conv_algo, void* workspace,
                             size_t workspace_bytes, void* stream, const LnEpilogue& ln) {
  if (int rc = check_shape_b(s)) return rc;
  if (!x_ || !w_u || !b_u || !wprep || !b_h || !hprep || !u || !m_ || (!h_ && !ln.y) || !h_n_)
    return set_error("forward: NULL pointer");
  bool scan = false;
  if (int rc = use_scan_route(conv_algo, scanprep, &scan)) return rc;
  if (!al16(wprep) || !al16(hprep)) return set_error("forward: wprep/hprep must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* x = (const bf16_t*)x_;
  bf16_t* m = (bf16_t*)m_;
  bf16_t* h = (bf16_t*)h_;
  bf16_t* h_n = (bf16_t*)h_n_;
  const long rows = (long)s->B * s->T;
  const WprepLayout wl = wprep_layout(*s);
  const int ldW = wl.ldW;
  const bf16_t* Wb = (const bf16_t*)((const char*)wprep + wl.off_Wb);
  const HprepLayout hl = hprep_layout(s->M, s->ldH);
  const bf16_t* Hb = (const bf16_t*)((const char*)hprep + hl.off_Hb);
  {
    const bool vec = (s->D % 8 == 0) && al16(x) && al16(w_u);
    int prep_blocks = 0;
    WeightPrep prep{nullptr, nullptr, nullptr, 0, 0, 0, 0};
    if (W_h) prep = weight_prep(*s, W_h, wprep, &prep_blocks);
    const dim3 grid(prep_blocks + min(cdivb(rows, vec ? 32 : 4), 4096));
    PROFB("k_encode_fwd_bf16");
    if (vec)
      hipLaunchKernelGGL(k_encode_fwd_bf16<true>, grid, dim3(kThreads), 0, st, x, w_u, b_u, u, (int)rows, s->D, prep,
                         prep_blocks);
    else
      hipLaunchKernelGGL(k_encode_fwd_bf16<false>, grid, dim3(kThreads), 0, st, x, w_u, b_u, u, (int)rows, s->D, prep,
                         prep_blocks);
  }
  LAUNCH_CHECKB("k_encode_fwd_bf16");
  const bool vec = (s->M % 8 == 0) && (s->D % 8 == 0) && (s->Hd % 8 == 0) && al16(m) && al16(x) && al16(h) &&
                   al16(h_n);  // (a NULL h counts as aligned)
  if (ln.y && !(vec && s->Hd <= ProjGldsTile::BN && al16(ln.y) && al16(ln.gamma) && al16(ln.beta)))
    return set_error("forward with layer norm: needs hidden_size <= %d, memory / input / hidden sizes that are "
                     "multiples of 8 and 16-byte aligned tensors; use lmufft_add_layernorm after the layer",
                     ProjGldsTile::BN);
  // AUTO: convolution and projection as one kernel where its tile shapes fit (TOEPLITZ asks for the separate
  // kernels; both give the same bits)
  // -- and where its workgroups (one per CU: 160 KB of LDS) run in about one round: the phases of a workgroup
  // do not overlap with anybody else's, so over several rounds the separate kernels, whose convolution runs at
  // full occupancy, are faster (batch 64 x 400 frames: 32 us against 12 + 25; batch 256: 115 against 23 + 84)
  const int fused_ntile = cdivb(s->T, ProjGldsTile::BM);
  if (!scan && conv_algo == LMUFFT_CONV_AUTO && vec && s->M % 64 == 0 && s->M <= FusedConvTile::BN &&
      s->Hd <= ProjGldsTile::BN && s->T <= kFusedMaxT && al16(Hb) && hl.ldHb % 8 == 0 &&
      (long)s->B * fused_ntile <= 3L * compute_units() / 2) {
    const int ntile = fused_ntile, FT = cdivb(s->T, ntile);
    if (hipFuncSetAttribute((const void*)k_conv_proj_fwd_bf16, hipFuncAttributeMaxDynamicSharedMemorySize,
                            kFusedLdsTotal) != hipSuccess)
      return set_error("cannot reserve %d bytes of LDS for k_conv_proj_fwd_bf16", kFusedLdsTotal);
    {
      PROFB("k_conv_proj_fwd_bf16");
      hipLaunchKernelGGL(k_conv_proj_fwd_bf16, dim3(s->B * ntile), dim3(ProjGldsTile::THREADS), kFusedLdsTotal, st, u, Hb,
                         hl.ldHb, x, Wb, b_h, m, h, h_n, (unsigned char*)hmask, s->T, s->M, s->D, s->Hd, ldW, ntile, FT,
                         ln);
    }
    LAUNCH_CHECKB("k_conv_proj_fwd_bf16");
    return 0;
  }
  if (scan) {
    if (int rc = scan_conv_forward(*s, u, scanprep, m, workspace, workspace_bytes, st)) return rc;
  } else if (int rc = conv_forward_bf16_any(*s, u, Hb, hl.ldHb, m, st)) {
    return rc;
  }
  {
    using TL = ProjTileB;
    if (vec) {  // LDS-DMA staging
      using TG = ProjGldsTile;
      const dim3 grid(cdivb(s->Hd, TG::BN), cdivb(rows, TG::BM));
      const size_t lds = (size_t)kProjStages * (TG::BM + TG::BN) * TG::BK * 2;
      static_assert(kProjStages * (TG::BM + TG::BN) * TG::BK * 2 >= TG::BM * (TG::BN + 8) * 2, "epilogue tile fits");
      if (hipFuncSetAttribute((const void*)k_proj_fwd_glds_bf16, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return set_error("cannot reserve %zu bytes of LDS for k_proj_fwd_glds_bf16", lds);
      {
        PROFB("k_proj_fwd_bf16");
        hipLaunchKernelGGL(k_proj_fwd_glds_bf16, grid, dim3(TG::THREADS), lds, st, m, x, Wb, b_h, h, h_n, (int)rows,
                           s->T, s->M, s->D, s->Hd, ldW, ln, (unsigned char*)hmask);
      }
    } else {
      const dim3 grid(cdivb(s->Hd, TL::BN), cdivb(rows, TL::BM));
      const size_t lds = sizeof(bf16_t) * (size_t)(TL::BM + TL::BN) * (TL::BK + 8);
      PROFB("k_proj_fwd_bf16");
      hipLaunchKernelGGL(k_proj_fwd_bf16<false>, grid, dim3(kThreads), lds, st, m, x, Wb, b_h, h, h_n, (int)rows,
                         s->T, s->M, s->D, s->Hd, ldW);
      if (hmask && s->Hd % 8 == 0) {
        const long chunks = rows * (s->Hd / 8);
        hipLaunchKernelGGL(k_hmask_from_h, dim3((int)std::min<long>((chunks + kThreads - 1) / kThreads, 4096)),
                           dim3(kThreads), 0, st, (const bf16_t*)h, (unsigned char*)hmask, chunks);
      }
    }
    LAUNCH_CHECKB("k_proj_fwd_bf16");
  }
  return 0;
}

// Adjoint convolution of gm.  Either leaves `ga` = gu * (u > 0) finished (u may be NULL: no mask),
// or -- when `defer` is set and the slab form ran -- leaves P partial slabs in `gupart` and describes
// them in *ga_parts for the input-gradient GEMM to sum.
static int conv_backward_bf16_any(const lmufft_shape& sh, const bf16_t* gm, const HprepLayout& hl, const bf16_t* HbT,
                                  const BwdPlanB& p, float* gupart, const float* u, float* ga, bool defer,
                                  GaParts* ga_parts, hipStream_t st) {
  const lmufft_shape* s = &sh;
  const long rows = (long)s->B * s->T;
  const bool dgrad_x_fast = defer;
  const int KS = cdivb(s->M, 16);
  if (s->T <= kGuQMaxT && s->M % GuQTile::BK == 0 && al16(gm) && al16(HbT) && hl.ldHT % 8 == 0 &&
      p.q_slots <= kGuQMaxSlots) {
    // Short sequences: the kernel that takes the encoder gradient straight from gz (k_gu_q_bf16) is this very
    // product with (gz, Q^T, Hd) = (gm, H^T, M) -- 128 frames x 256 lags per workgroup by LDS-DMA, diagonal sums in
    // registers, six partial slabs per utterance at 400 frames instead of twelve
    using TG = GuQTile;
    const size_t lds = (size_t)kGuQStages * (TG::BM + TG::BN) * TG::BK * 2;
    if (hipFuncSetAttribute((const void*)k_gu_q_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return set_error("cannot reserve %zu bytes of LDS for k_gu_q_bf16", lds);
    {
      PROFB("k_conv_bwd_bf16");
      hipLaunchKernelGGL(k_gu_q_bf16, dim3(s->B * p.q_slots), dim3(TG::THREADS), lds, st, gm, HbT, gupart, s->B, s->T,
                         s->M, hl.ldHT, hl.ldHb, guq_plan(s->T));
    }
    LAUNCH_CHECKB("k_conv_bwd_bf16");
    if (dgrad_x_fast && p.q_slots <= kGaMaxParts) {
      *ga_parts = GaParts{gupart, u, ga, rows, p.q_slots};
    } else {
      if (int rc = launch_sum_slabs_masked(gupart, u, ga, rows, p.q_slots, st)) return rc;
    }
  } else if (s->M % 8 == 0 && al16(gm) && (KS == 2 || KS == 4 || KS == 8 || KS == 12 || KS == 16)) {
    const int P = p.cb2_P;
    const dim3 grid(P, cdivb(s->T, kCb2St), cdivb(s->B, kCb2Bt));
    const int Lp_max = cdivb(s->T, P);
    const size_t lds = (size_t)(Lp_max + kCb2St - 1) * (KS * 16 + 8) * sizeof(bf16_t) +
                       sizeof(float) * 4 * kCb2Bt * kCb2St;
    if (lds > 160 * 1024) return set_error("adjoint convolution: LDS window of %zu bytes does not fit", lds);
#define LAUNCH_CB2(KSV)                                                                                       \
  do {                                                                                                        \
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)k_conv_bwd2_bf16<KSV>,                             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return set_error("cannot reserve %zu bytes of LDS for k_conv_bwd2_bf16", lds);                          \
    PROFB("k_conv_bwd_bf16");                                                                                 \
    hipLaunchKernelGGL(k_conv_bwd2_bf16<KSV>, grid, dim3(kThreads), lds, st, (const bf16_t*)gm, HbT, gupart,   \
                       s->B, s->T, s->M, hl.ldHT, hl.ldHb, P);                                                \
  } while (0)
    switch (KS) {
      case 2: LAUNCH_CB2(2); break;
      case 4: LAUNCH_CB2(4); break;
      case 8: LAUNCH_CB2(8); break;
      case 12: LAUNCH_CB2(12); break;
      default: LAUNCH_CB2(16); break;
    }
#undef LAUNCH_CB2
    LAUNCH_CHECKB("k_conv_bwd_bf16");
    if (dgrad_x_fast && P <= kGaMaxParts) {
      *ga_parts = GaParts{gupart, u, ga, rows, P};
    } else {
      if (int rc = launch_sum_slabs_masked(gupart, u, ga, rows, P, st)) return rc;
    }
  } else {
    using TL = ConvBwdTileB;
    const dim3 grid(1, s->B, p.ntt);
    const size_t lds = sizeof(bf16_t) * (size_t)(TL::BM + TL::BN) * (TL::BK + 8) +
                       sizeof(float) * ((size_t)p.ntt * TL::BM + (size_t)TL::BM * (TL::BN + 1));
    if (lds > 160 * 1024) return set_error("sequence too long for the Toeplitz backward kernel (T=%d)", s->T);
    const bool vec = (s->M % 8 == 0) && al16(gm);
    const void* fn = vec ? (const void*)k_conv_bwd_bf16<true> : (const void*)k_conv_bwd_bf16<false>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return set_error("cannot reserve %zu bytes of LDS for k_conv_bwd_bf16", lds);
    {
      PROFB("k_conv_bwd_bf16");
      if (vec)
        hipLaunchKernelGGL(k_conv_bwd_bf16<true>, grid, dim3(kThreads), lds, st, gm, HbT, gupart, s->T, s->M,
                           hl.ldHT, hl.ldHb);
      else
        hipLaunchKernelGGL(k_conv_bwd_bf16<false>, grid, dim3(kThreads), lds, st, gm, HbT, gupart, s->T, s->M,
                           hl.ldHT, hl.ldHb);
    }
    LAUNCH_CHECKB("k_conv_bwd_bf16");
    if (int rc = launch_gu_reduce(gupart, u, ga, s->B, s->T, p.ntt, TL::BM, st)) return rc;
  }
  return 0;
}

extern "C" {

int lmufft_forward_bf16(const lmufft_shape* s, const void* x, const float* w_u, const float* b_u, void* wprep,
                        const float* W_h, const float* b_h, const void* hprep, const void* scanprep, float* u, void* m,
                        void* h, void* h_n, void* hmask, int conv_algo, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (!h) return set_error("forward: NULL pointer");
  if (hmask && s && s->Hd % 8) return set_error("forward: hmask needs a hidden size that is a multiple of 8");
  return forward_bf16_impl(s, x, w_u, b_u, wprep, W_h, b_h, hprep, scanprep, u, m, h, h_n, hmask, conv_algo, workspace,
                           workspace_bytes, stream, LnEpilogue{nullptr, nullptr, 0.f, nullptr, 0});
}

int lmufft_forward_ln_bf16(const lmufft_shape* s, const void* x, const float* w_u, const float* b_u, void* wprep,
                           const float* W_h, const float* b_h, const void* hprep, const void* scanprep, float* u,
                           void* m, void* h, void* h_n, const float* gamma, const float* beta, float eps, void* y,
                           int y_dtype, int conv_algo, void* workspace, size_t workspace_bytes, void* stream) {
  if (!gamma || !beta || !y) return set_error("forward with layer norm: NULL pointer");
  if (y_dtype != LMUFFT_F32 && y_dtype != LMUFFT_BF16) return set_error("forward with layer norm: unknown dtype");
  return forward_bf16_impl(s, x, w_u, b_u, wprep, W_h, b_h, hprep, scanprep, u, m, h, h_n, nullptr, conv_algo, workspace,
                           workspace_bytes, stream, LnEpilogue{gamma, beta, eps, y, y_dtype == LMUFFT_BF16});
}

}  // extern "C"

// Q^T of one layer (rows and row stride as the backward's plan has them) from that layer's hprep and wprep
static int q_rows_of(const lmufft_shape& s) { return cdivb(s.T, 64) * 64; }
static int q_ld_of(const lmufft_shape& s) { return up8(s.Hd); }
static QPrep make_qprep(const lmufft_shape& s, const void* hprep, const void* wprep, void* qt) {
  const HprepLayout hl = hprep_layout(s.M, s.ldH);
  const WprepLayout wl = wprep_layout(s);
  return QPrep{(const bf16_t*)((const char*)hprep + hl.off_HbT), (const bf16_t*)((const char*)wprep + wl.off_Wb),
               (bf16_t*)qt, q_rows_of(s), hl.ldHb, s.Hd, s.M, hl.ldHT, wl.ldW, q_ld_of(s)};
}
static int qprep_blocks(const lmufft_shape& s) { return (q_rows_of(s) / QPrepTile::BM) * cdivb(s.Hd, QPrepTile::BN); }
static int check_q_items(const lmufft_q_item* items, int count) {
  if (count < 0 || (count > 0 && !items)) return set_error("Q preparation: bad item list");
  for (int i = 0; i < count; ++i) {
    if (int rc = check_shape_b(&items[i].shape)) return rc;
    if (!q_route_shape_ok(items[i].shape)) return set_error("Q preparation: item %d has no short-sequence route", i);
    if (!items[i].hprep || !items[i].wprep || !items[i].qt || !al16(items[i].hprep) || !al16(items[i].wprep) ||
        !al16(items[i].qt))
      return set_error("Q preparation: item %d needs 16-byte aligned hprep, wprep and qt", i);
  }
  return 0;
}
// the items as launches of their own, kQBatchMax layers each
static int launch_qprep_many(const lmufft_q_item* items, int count, hipStream_t st) {
  for (int i0 = 0; i0 < count; i0 += kQBatchMax) {
    QPrepBatch qb{};
    qb.count = std::min(count - i0, kQBatchMax);
    for (int i = 0; i < qb.count; ++i) {
      qb.q[i] = make_qprep(items[i0 + i].shape, items[i0 + i].hprep, items[i0 + i].wprep, items[i0 + i].qt);
      qb.first[i + 1] = qb.first[i] + qprep_blocks(items[i0 + i].shape);
    }
    {
      PROFB("k_relu_bwd_bf16");
      hipLaunchKernelGGL(k_qprep_bf16, dim3(qb.first[qb.count]), dim3(kThreads), 0, st, qb);
    }
    LAUNCH_CHECKB("k_qprep_bf16");
  }
  return 0;
}

// parts: 1 = everything up to gx (leaves gz and ga in the workspace), 2 = the parameter gradients from
// the workspace's gz and ga, 3 = both
static int backward_bf16_parts(int parts, const lmufft_shape* s, const void* x_, const float* w_u, const void* wprep,
                               const void* hprep, const void* scanprep, const float* u, const void* m_,
                               const void* h_, const void* hmask, const void* gh_, const void* gh_n_, void* gx_,
                               float* gw_u,
                               float* gb_u, float* gW_h, float* gb_h, void* gm_, int conv_algo, void* workspace,
                               size_t workspace_bytes, void* stream, unsigned flags = 0,
                               const lmufft_bwd_extras* extras = nullptr) {
  const void* x_mask = extras ? extras->x_mask : nullptr;
  const int n_extra_q = extras ? extras->q_count : 0;
  if (int rc = check_q_items(extras ? extras->q_items : nullptr, n_extra_q)) return rc;
  if (extras && extras->qt && !al16(extras->qt)) return set_error("backward: extras.qt must be 16-byte aligned");
  if (int rc = check_shape_b(s)) return rc;
  if (flags & ~(unsigned)(LMUFFT_BWD_GH_MASKED | LMUFFT_BWD_MASK_GX)) return set_error("backward: unknown flags %u", flags);
  if (flags && parts != 3 && parts != 7) return set_error("backward: flags need the whole backward in one call");
  if (!x_ || !m_) return set_error("backward: NULL pointer");
  if ((parts & 1) && (!w_u || !wprep || !hprep || !u || !h_)) return set_error("backward: NULL pointer");
  if ((parts & 2) && !(parts & 4) && (!gw_u || !gb_u || !gW_h || !gb_h)) return set_error("backward: NULL pointer");
  bool scan = false;
  if (parts & 1)
    if (int rc = use_scan_route(conv_algo, scanprep, &scan)) return rc;
  const BwdPlanB p = plan_backward_b(*s);
  if (!workspace || workspace_bytes < p.total)
    return set_error("backward: workspace of %zu bytes needed, %zu given", p.total, workspace_bytes);
  if (!al16(workspace) || ((parts & 1) && (!al16(hprep) || !al16(wprep))))
    return set_error("backward: workspace/hprep/wprep must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* x = (const bf16_t*)x_;
  const bf16_t* m = (const bf16_t*)m_;
  const bf16_t* h = (const bf16_t*)h_;
  const bf16_t* gh = (const bf16_t*)gh_;
  const bf16_t* gh_n = (const bf16_t*)gh_n_;
  bf16_t* gx = (bf16_t*)gx_;
  bf16_t* gm = (bf16_t*)gm_;
  char* ws = (char*)workspace;
  const WprepLayout wl = wprep_layout(*s);
  const bf16_t* WbT = wprep ? (const bf16_t*)((const char*)wprep + wl.off_WbT) : nullptr;
  float* wslab = (float*)(ws + p.off_wslab);
  float* gupart = (float*)(ws + p.off_gupart);
  float* ga = (float*)(ws + p.off_ga);
  const long rows = (long)s->B * s->T;
  const int MD = s->M + s->D;
  const HprepLayout hl = hprep_layout(s->M, s->ldH);
  const bf16_t* HbT = hprep ? (const bf16_t*)((const char*)hprep + hl.off_HbT) : nullptr;
  bf16_t* gz = (bf16_t*)(ws + p.off_gz);
  const bool gh_masked = flags & LMUFFT_BWD_GH_MASKED;
  if (gh_masked) {
    // the caller vouches that gh is already zero wherever h <= 0: gh IS gz, the masking pass is not run
    if (!gh || gh_n || !al16(gh)) return set_error("backward: LMUFFT_BWD_GH_MASKED needs a 16-byte aligned gh and no gh_n");
    gz = const_cast<bf16_t*>(gh);
  }
  const bool vec_h = (s->Hd % 8 == 0) && (s->Hd / 8 <= kThreads) && al16(h) && (!gh || al16(gh)) &&
                     (!gh_n || al16(gh_n));
  // gm == NULL selects the Q route: the encoder gradient straight from gz, gm is never formed
  const bool qroute = (parts & 1) && !gm && !scan && q_route_shape_ok(*s) && vec_h;
  if ((parts & 1) && !gm && !qroute)
    return set_error("backward: gm == NULL needs the Q route (no state-space route, T <= %d, memory size a multiple of 8, "
                     "hidden size a multiple of 32, 16-byte aligned h / gh); pass a gm buffer", kGuQMaxT);

  if (parts & 1) {
  // Q^T of this layer: given ready-made (extras.qt), or made here in the workspace; Q^T of other layers
  // (extras.q_items) rides on this call's first launch where that launch has workgroups to spare
  const bf16_t* QT = (qroute && extras && extras->qt) ? (const bf16_t*)extras->qt : (const bf16_t*)(ws + p.off_q);
  int extra_done = 0;
  if (qroute) {  // gz = relu'(h) * (gh + gh_n), and Q^T = H^T W_h[:, :M]^T on the pass's first workgroups
    QPrepBatch qb{};
    if (!(extras && extras->qt)) {
      qb.q[0] = make_qprep(*s, hprep, wprep, ws + p.off_q);
      qb.first[1] = qprep_blocks(*s);
      qb.count = 1;
    }
    if (!gh_masked)  // the masking pass hides them
      for (; extra_done < n_extra_q && qb.count < kQBatchMax; ++extra_done, ++qb.count) {
        const lmufft_q_item& it = extras->q_items[extra_done];
        qb.q[qb.count] = make_qprep(it.shape, it.hprep, it.wprep, it.qt);
        qb.first[qb.count + 1] = qb.first[qb.count] + qprep_blocks(it.shape);
      }
    const int qblocks = qb.first[qb.count];
    const long work = rows * (s->Hd / 8);
    const dim3 grid(qblocks + (int)std::min<long>((work + kThreads - 1) / kThreads, 8192));
    if (!gh_masked || qblocks > 0) {
      PROFB("k_relu_bwd_bf16");
      if (gh_masked)  // only Q^T is made
        hipLaunchKernelGGL(k_qprep_bf16, dim3(qblocks), dim3(kThreads), 0, st, qb);
      else
        hipLaunchKernelGGL(k_relu_bwd_q_bf16, grid, dim3(kThreads), 0, st, h, (const unsigned char*)hmask, gh, gh_n, gz,
                           (int)rows, s->T, s->Hd, qb);
    }
    LAUNCH_CHECKB("k_relu_bwd_q_bf16");
  }
  if (extra_done < n_extra_q)
    if (int rc = launch_qprep_many(extras->q_items + extra_done, n_extra_q - extra_done, st)) return rc;
  if (!qroute && !gh_masked) {  // gz = relu'(h) * (gh + gh_n)
    const long work = vec_h ? rows * (s->Hd / 8) : rows * s->Hd;
    const dim3 grid(std::min<long>((work + kThreads - 1) / kThreads, 8192));
    {
      PROFB("k_relu_bwd_bf16");
      if (vec_h)
        hipLaunchKernelGGL(k_relu_bwd_bf16<true>, grid, dim3(kThreads), 0, st, h, gh, gh_n, gz, (int)rows, s->T, s->Hd);
      else
        hipLaunchKernelGGL(k_relu_bwd_bf16<false>, grid, dim3(kThreads), 0, st, h, gh, gh_n, gz, (int)rows, s->T,
                           s->Hd);
    }
    LAUNCH_CHECKB("k_relu_bwd_bf16");
  }

  // out = gz W_h[:, j_begin : j_begin + ncols] (+ ga w_u), LDS-DMA kernel when everything is 16-byte friendly
  auto dgrad = [&](auto glds_tag, auto fallback_tag, const char* name, bf16_t* out, int j_begin, int ncols,
                   const float* ga_, const float* wu_, GaParts gparts, const bf16_t* relu_src = nullptr,
                   const unsigned char* relu_bits = nullptr) -> int {
    using TG = decltype(glds_tag);
    using TF = decltype(fallback_tag);
    const bool fast = (s->Hd % 8 == 0) && (ncols % 8 == 0) && al16(out);
    if (relu_src && !(fast && al16(relu_src)))
      return set_error("backward: LMUFFT_BWD_MASK_GX needs hidden and input sizes that are multiples of 8 and 16-byte "
                       "aligned x / gx");
    if (fast) {
      // rows per workgroup: the whole tile when that already gives a workgroup per CU, otherwise the row
      // ranges are shortened until the grid fills the chip (batch 64 x 400 frames: 200 rows, 256 workgroups)
      const int col_tiles = cdivb(ncols, TG::BN);
      int rpw = TG::BM;
      if ((long)cdivb(rows, TG::BM) * col_tiles < 256)
        rpw = std::min<long>(TG::BM, (cdivb(rows, std::max(1, 256 / col_tiles)) + 7) / 8 * 8);
      const dim3 grid(xcd_grid(cdivb(rows, rpw), col_tiles));
      const size_t lds = 2 * (size_t)(TG::BM + TG::BN) * kGldsRowBytes;
      static_assert(2 * (TG::BM + TG::BN) * kGldsRowBytes >= TG::BM * (TG::BN + 8) * 2, "epilogue tile fits");
      if (hipFuncSetAttribute((const void*)k_dgrad_glds_bf16<TG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return set_error("cannot reserve %zu bytes of LDS for k_dgrad_glds_bf16", lds);
      PROFB(name);
      hipLaunchKernelGGL(k_dgrad_glds_bf16<TG>, grid, dim3(TG::THREADS), lds, st, (const bf16_t*)gz, WbT, out, ga_,
                         wu_, (int)rows, s->Hd, wl.ldWT, j_begin, ncols, gparts, rpw, col_tiles, relu_src, relu_bits);
    } else {
      const dim3 grid(cdivb(ncols, TF::BN), cdivb(rows, TF::BM));
      const size_t lds = sizeof(bf16_t) * (size_t)(TF::BM + TF::BN) * (TF::BK + 8);
      PROFB(name);
      hipLaunchKernelGGL((k_dgrad_bf16<TF, false>), grid, dim3(kThreads), lds, st, (const bf16_t*)gz, WbT, out, ga_,
                         wu_, (int)rows, s->Hd, wl.ldWT, j_begin, ncols);
    }
    LAUNCH_CHECKB(name);
    return 0;
  };

  const GaParts no_parts{nullptr, nullptr, nullptr, 0, 0};
  // the input-gradient GEMM can sum the adjoint convolution's partial slabs itself (and write ga)
  const bool dgrad_x_fast = gx && (s->Hd % 8 == 0) && (s->D % 8 == 0) && al16(gx);
  GaParts ga_parts = no_parts;
  if (qroute) {  // gu slabs straight from gz
    using TG = GuQTile;
    const size_t lds = (size_t)kGuQStages * (TG::BM + TG::BN) * TG::BK * 2;
    static_assert(kGuQStages * (TG::BM + TG::BN) * TG::BK * 2 >= TG::NW * TG::TN * 128 * 4, "the waves' diagonal sums fit");
    if (hipFuncSetAttribute((const void*)k_gu_q_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return set_error("cannot reserve %zu bytes of LDS for k_gu_q_bf16", lds);
    {
      PROFB("k_gu_q_bf16");
      hipLaunchKernelGGL(k_gu_q_bf16, dim3(s->B * p.q_slots), dim3(TG::THREADS), lds, st, (const bf16_t*)gz,
                         QT, gupart, s->B, s->T, s->Hd, p.q_ld, p.q_rows, guq_plan(s->T));
    }
    LAUNCH_CHECKB("k_gu_q_bf16");
    if (dgrad_x_fast && p.q_slots <= kGaMaxParts) {
      ga_parts = GaParts{gupart, u, ga, rows, p.q_slots};
    } else if (int rc = launch_sum_slabs_masked(gupart, u, ga, rows, p.q_slots, st)) {
      return rc;
    }
  } else {
  // gm = gz W_h[:, :M]
  if (int rc = dgrad(DgradMTileB{}, DgradMTileB{}, "k_dgrad_m_bf16", gm, 0, s->M, nullptr, nullptr, no_parts)) return rc;
  // ga = (adjoint conv of gm) * (u > 0)
  if (scan) {
    if (int rc = scan_conv_backward(*s, gm, scanprep, u, ga, ws + p.off_scan, p.scan_bytes, st)) return rc;
  } else if (int rc = conv_backward_bf16_any(*s, gm, hl, HbT, p, gupart, u, ga, dgrad_x_fast, &ga_parts, st)) {
    return rc;
  }
  }
  // gx = gz W_h[:, M:] + ga w_u
  if (gx) {
    if (int rc = dgrad(DgradXGldsTile{}, DgradXTileB{}, "k_dgrad_x_bf16", gx, s->M, s->D, (const float*)ga, w_u, ga_parts,
                       (flags & LMUFFT_BWD_MASK_GX) ? x : nullptr,
                       (flags & LMUFFT_BWD_MASK_GX) ? (const unsigned char*)x_mask : nullptr))
      return rc;
  } else if (flags & LMUFFT_BWD_MASK_GX) {
    return set_error("backward: LMUFFT_BWD_MASK_GX without gx");
  } else if (ga_parts.part) {  // no input gradient wanted: nobody has summed the adjoint's slabs into ga yet
    if (int rc = launch_sum_slabs_masked(ga_parts.part, u, ga, rows, ga_parts.P, st)) return rc;
  }
  }  // parts & 1
  if (parts & 2) {  // gW_h, gb_h, gw_u, gb_u
    using TL = WgradTileB;
    const bool vec = (s->Hd % 8 == 0) && (s->M % 8 == 0) && (s->D % 8 == 0) && al16(m) && al16(x);
    const dim3 grid(xcd_grid(p.S, cdivb(MD, TL::BN) * cdivb(s->Hd, TL::BM)));
    const size_t lds = std::max(2 * (sizeof(bf16_t) * (size_t)TL::BK * ((TL::BM + 32) + (TL::BN + 32)) + sizeof(float) * TL::BK),
                                sizeof(float) * (size_t)TL::BM * (TL::BN + 4));
    const bool pair = vec;  // eight waves, two k-interleaved groups
    if (hipFuncSetAttribute(vec ? (const void*)k_wgrad_bf16<true> : (const void*)k_wgrad_bf16<false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return set_error("cannot reserve %zu bytes of LDS for k_wgrad_bf16", lds);
    if (pair && hipFuncSetAttribute((const void*)k_wgrad_pair_bf16, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kWgPairLds) != hipSuccess)
      return set_error("cannot reserve %zu bytes of LDS for k_wgrad_pair_bf16", kWgPairLds);
    {
      PROFB("k_wgrad_bf16");
      if (pair)
        hipLaunchKernelGGL(k_wgrad_pair_bf16, grid, dim3(kWgPairThreads), kWgPairLds, st, (const bf16_t*)gz, m, x,
                           (const float*)ga, wslab, (int)rows, p.rps, s->M, s->D, s->Hd, 1);
      else if (vec)
        hipLaunchKernelGGL(k_wgrad_bf16<true>, grid, dim3(kThreads), lds, st, (const bf16_t*)gz, m, x,
                           (const float*)ga, wslab, (int)rows, p.rps, s->M, s->D, s->Hd);
      else
        hipLaunchKernelGGL(k_wgrad_bf16<false>, grid, dim3(kThreads), lds, st, (const bf16_t*)gz, m, x,
                           (const float*)ga, wslab, (int)rows, p.rps, s->M, s->D, s->Hd);
    }
    LAUNCH_CHECKB("k_wgrad_bf16");
    if (!(parts & 4))  // (4: the slabs stay in the workspace for lmufft_backward_weights_sum_many_bf16)
      if (int rc = launch_reduce_slabs(wslab, p.S, wgrad_outputs(*s, gW_h, gb_h, gw_u, gb_u), st)) return rc;
  }
  return 0;
}

extern "C" {

int lmufft_backward_needs_gm_bf16(const lmufft_shape* s, int conv_algo, int has_scanprep) {
  if (!s) return 1;
  const bool scan = conv_algo == LMUFFT_CONV_SCAN || (conv_algo == LMUFFT_CONV_AUTO && has_scanprep);
  return (scan || !q_route_shape_ok(*s)) ? 1 : 0;
}

int lmufft_backward_bf16(const lmufft_shape* s, const void* x, const float* w_u, const void* wprep,
                         const void* hprep, const void* scanprep, const float* u, const void* m, const void* h,
                         const void* hmask, const void* gh, const void* gh_n, void* gx, float* gw_u, float* gb_u,
                         float* gW_h, float* gb_h, void* gm, int conv_algo, void* workspace, size_t workspace_bytes,
                         void* stream) {
  return backward_bf16_parts(3, s, x, w_u, wprep, hprep, scanprep, u, m, h, hmask, gh, gh_n, gx, gw_u, gb_u, gW_h, gb_h, gm,
                             conv_algo, workspace, workspace_bytes, stream);
}

int lmufft_backward_partial_bf16(const lmufft_shape* s, const void* x, const float* w_u, const void* wprep,
                                 const void* hprep, const void* scanprep, const float* u, const void* m, const void* h,
                                 const void* hmask, const void* gh, const void* gh_n, void* gx, void* gm, int conv_algo,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  return backward_bf16_parts(3 | 4, s, x, w_u, wprep, hprep, scanprep, u, m, h, hmask, gh, gh_n, gx, nullptr, nullptr,
                             nullptr, nullptr, gm, conv_algo, workspace, workspace_bytes, stream);
}

int lmufft_backward_flags_bf16(const lmufft_shape* s, const void* x, const float* w_u, const void* wprep,
                               const void* hprep, const void* scanprep, const float* u, const void* m, const void* h,
                               const void* hmask, const void* gh, const void* gh_n, void* gx, float* gw_u, float* gb_u,
                               float* gW_h, float* gb_h, void* gm, int conv_algo, void* workspace,
                               size_t workspace_bytes, void* stream, uint32_t flags,
                               const lmufft_bwd_extras* extras) {
  const uint32_t known = LMUFFT_BWD_GH_MASKED | LMUFFT_BWD_MASK_GX | LMUFFT_BWD_PARTIAL_SUMS;
  if (flags & ~known) return set_error("backward: unknown flags %u", flags);
  return backward_bf16_parts((flags & LMUFFT_BWD_PARTIAL_SUMS) ? 7 : 3, s, x, w_u, wprep, hprep, scanprep, u, m, h, hmask,
                             gh, gh_n, gx, gw_u, gb_u, gW_h, gb_h, gm, conv_algo, workspace, workspace_bytes, stream,
                             flags & ~(uint32_t)LMUFFT_BWD_PARTIAL_SUMS, extras);
}

size_t lmufft_qt_bytes(const lmufft_shape* s) {
  if (!s || check_shape_b(s) || !q_route_shape_ok(*s)) return 0;
  return align256b(sizeof(bf16_t) * (size_t)q_rows_of(*s) * q_ld_of(*s));
}

int lmufft_qprep_many_bf16(const lmufft_q_item* items, int32_t count, void* stream) {
  if (int rc = check_q_items(items, count)) return rc;
  return launch_qprep_many(items, count, (hipStream_t)stream);
}

int lmufft_backward_weights_sum_many_bf16(const lmufft_lmu_sum_item* items, int32_t count, void* stream) {
  if (count < 0 || (count > 0 && !items)) return set_error("backward weight sums: bad item list");
  std::vector<const float*> in(count);
  std::vector<int> S(count);
  std::vector<SlabOutputs> o(count);
  for (int i = 0; i < count; ++i) {
    const lmufft_lmu_sum_item& it = items[i];
    if (int rc = check_shape_b(&it.shape)) return rc;
    if (!it.gw_u || !it.gb_u || !it.gW_h || !it.gb_h || !it.workspace) return set_error("backward weight sums: NULL pointer");
    const BwdPlanB p = plan_backward_b(it.shape);
    if (it.workspace_bytes < p.total)
      return set_error("backward weight sums: workspace of %zu bytes needed, %zu given", p.total, it.workspace_bytes);
    in[i] = (const float*)((const char*)it.workspace + p.off_wslab);
    S[i] = p.S;
    o[i] = wgrad_outputs(it.shape, it.gW_h, it.gb_h, it.gw_u, it.gb_u);
  }
  return launch_reduce_slabs_many(in.data(), S.data(), o.data(), count, (hipStream_t)stream);
}

int lmufft_backward_data_bf16(const lmufft_shape* s, const void* x, const float* w_u, const void* wprep,
                              const void* hprep, const void* scanprep, const float* u, const void* m, const void* h,
                              const void* hmask, const void* gh, const void* gh_n, void* gx, void* gm, int conv_algo,
                              void* workspace, size_t workspace_bytes, void* stream) {
  return backward_bf16_parts(1, s, x, w_u, wprep, hprep, scanprep, u, m, h, hmask, gh, gh_n, gx, nullptr, nullptr, nullptr,
                             nullptr, gm, conv_algo, workspace, workspace_bytes, stream);
}

int lmufft_backward_weights_bf16(const lmufft_shape* s, const void* x, const void* m, float* gw_u, float* gb_u,
                                 float* gW_h, float* gb_h, void* workspace, size_t workspace_bytes, void* stream) {
  return backward_bf16_parts(2, s, x, nullptr, nullptr, nullptr, nullptr, nullptr, m, nullptr, nullptr, nullptr, nullptr,
                             nullptr, gw_u, gb_u, gW_h, gb_h, nullptr, LMUFFT_CONV_AUTO, workspace, workspace_bytes,
                             stream);
}

// Weight and bias gradient of a Linear layer, dW[out][in] = sum_rows dy[row][out] x[row][in], db[out] = sum_rows
// dy[row][out]: the layer's weight-gradient kernel with an empty memory block (M = 0) and no encoder gradient.
struct LinWgradPlan {
  int S, rps, tiles_j, tiles_n, tile_blocks;
  long slab_floats;
  size_t off_scratch, total;
};
static LinWgradPlan plan_linear_wgrad(long rows, int in, int out) {
  using TL = WgradTileB;
  LinWgradPlan p;
  p.tiles_j = cdivb(in, TL::BN);
  p.tiles_n = cdivb(out, TL::BM);
  const int tiles = p.tiles_j * p.tiles_n;
  // One round of workgroups: an XCD (32 CUs, one workgroup each) takes whole groups of `inner` tiles.  A row range
  // with more than 16 tiles is cut into tile_blocks groups, so that an XCD still holds several ranges' worth.
  p.tile_blocks = 1;
  while (tiles / p.tile_blocks > 16 || tiles % p.tile_blocks) ++p.tile_blocks;
  const int inner = tiles / p.tile_blocks;
  const int S = std::max(1, std::min(64, 8 * (32 / inner) / p.tile_blocks));
  int rps = cdivb(rows, S);
  rps = cdivb(rps, TL::BK) * TL::BK;
  p.S = cdivb(rows, rps);
  p.rps = rps;
  const SlabOutputs o{{nullptr, nullptr, nullptr, nullptr}, {(long)out * in, out, in, 1}, {1, p.tiles_j * TL::WGN, p.tiles_n * TL::WGM, 1}};
  p.slab_floats = slab_stride(o);
  p.off_scratch = align256b(sizeof(float) * (size_t)p.S * (size_t)p.slab_floats);
  p.total = p.off_scratch + align256b(sizeof(float) * ((size_t)in + (size_t)out + 8));
  return p;
}

size_t lmufft_linear_wgrad_workspace_bytes(int64_t rows, int32_t in, int32_t out) {
  if (rows <= 0 || in <= 0 || out <= 0) return 0;
  return plan_linear_wgrad(rows, in, out).total;
}

static int linear_wgrad_check(const void* x, const void* dy, int64_t rows, int32_t in, int32_t out, const void* workspace,
                              size_t workspace_bytes, bool slabs) {
  if (rows <= 0 || rows > 0x7fffffff || in <= 0 || out <= 0) return set_error("linear wgrad: bad shape (rows=%lld in=%d out=%d)", (long long)rows, in, out);
  if (in % 8 || out % 8) return set_error("linear wgrad: in and out must be multiples of 8 (got %d, %d)", in, out);
  if (!workspace || (slabs && (!x || !dy))) return set_error("linear wgrad: NULL pointer");
  if (!al16(workspace) || (slabs && (!al16(x) || !al16(dy)))) return set_error("linear wgrad: x, dy and the workspace must be 16-byte aligned");
  const LinWgradPlan p = plan_linear_wgrad(rows, in, out);
  if (workspace_bytes < p.total) return set_error("linear wgrad: workspace of %zu bytes needed, %zu given", p.total, workspace_bytes);
  return 0;
}

int lmufft_linear_wgrad_slabs_bf16(const void* x, const void* dy, int64_t rows, int32_t in, int32_t out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  if (int rc = linear_wgrad_check(x, dy, rows, in, out, workspace, workspace_bytes, true)) return rc;
  const LinWgradPlan p = plan_linear_wgrad(rows, in, out);
  hipStream_t st = (hipStream_t)stream;
  float* wslab = (float*)workspace;
  const dim3 grid(xcd_grid((long)p.S * p.tile_blocks, p.tiles_j * p.tiles_n / p.tile_blocks));
  if (hipFuncSetAttribute((const void*)k_wgrad_pair_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWgPairLds) !=
      hipSuccess)
    return set_error("cannot reserve %zu bytes of LDS for k_wgrad_pair_bf16", kWgPairLds);
  {
    PROFB("k_linear_wgrad_bf16");
    hipLaunchKernelGGL(k_wgrad_pair_bf16, grid, dim3(kWgPairThreads), kWgPairLds, st, (const bf16_t*)dy,
                       (const bf16_t*)nullptr, (const bf16_t*)x, (const float*)nullptr, wslab, (int)rows, p.rps, 0, in,
                       out, p.tile_blocks);
  }
  LAUNCH_CHECKB("k_linear_wgrad_bf16");
  return 0;
}

int lmufft_linear_wgrad_sum_bf16(int64_t rows, int32_t in, int32_t out, float* dW, float* db, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  using TL = WgradTileB;
  if (int rc = linear_wgrad_check(nullptr, nullptr, rows, in, out, workspace, workspace_bytes, false)) return rc;
  if (!dW) return set_error("linear wgrad: NULL pointer");
  const LinWgradPlan p = plan_linear_wgrad(rows, in, out);
  float* wslab = (float*)workspace;
  float* scratch = (float*)((char*)workspace + p.off_scratch);  // gw_u (in) | gb_u (1) | db when the layer has no bias
  const SlabOutputs o{{dW, db ? db : scratch + in + 4, scratch, scratch + in},
                      {(long)out * in, out, in, 1},
                      {1, p.tiles_j * TL::WGN, p.tiles_n * TL::WGM, 1}};
  return launch_reduce_slabs(wslab, p.S, o, (hipStream_t)stream);
}

int lmufft_linear_wgrad_sum_many_bf16(const lmufft_wgrad_sum_item* items, int32_t count, void* stream) {
  using TL = WgradTileB;
  if (count < 0 || (count > 0 && !items)) return set_error("linear wgrad: bad item list");
  std::vector<const float*> in(count);
  std::vector<int> S(count);
  std::vector<SlabOutputs> o(count);
  for (int i = 0; i < count; ++i) {
    const lmufft_wgrad_sum_item& it = items[i];
    if (int rc = linear_wgrad_check(nullptr, nullptr, it.rows, it.in, it.out, it.workspace, it.workspace_bytes, false)) return rc;
    if (!it.dW) return set_error("linear wgrad: NULL pointer");
    const LinWgradPlan p = plan_linear_wgrad(it.rows, it.in, it.out);
    float* scratch = (float*)((char*)it.workspace + p.off_scratch);
    in[i] = (const float*)it.workspace;
    S[i] = p.S;
    o[i] = SlabOutputs{{it.dW, it.db ? it.db : scratch + it.in + 4, scratch, scratch + it.in},
                       {(long)it.out * it.in, it.out, it.in, 1},
                       {1, p.tiles_j * TL::WGN, p.tiles_n * TL::WGM, 1}};
  }
  return launch_reduce_slabs_many(in.data(), S.data(), o.data(), count, (hipStream_t)stream);
}

int lmufft_linear_wgrad_bf16(const void* x, const void* dy, int64_t rows, int32_t in, int32_t out, float* dW, float* db,
                             void* workspace, size_t workspace_bytes, void* stream) {
  if (!dW) return set_error("linear wgrad: NULL pointer");
  if (int rc = lmufft_linear_wgrad_slabs_bf16(x, dy, rows, in, out, workspace, workspace_bytes, stream)) return rc;
  return lmufft_linear_wgrad_sum_bf16(rows, in, out, dW, db, workspace, workspace_bytes, stream);
}

int lmufft_conv_forward_bf16(const lmufft_shape* s, const float* u, const void* hprep, void* m, void* stream) {
  if (!s || s->B <= 0 || s->T <= 0 || s->M <= 0) return set_error("conv: bad shape");
  if (s->T > s->ldH) return set_error("sequence length %d exceeds the impulse response length %d", s->T, s->ldH);
  if (!u || !hprep || !m) return set_error("conv: NULL pointer");
  if (!al16(hprep)) return set_error("conv: hprep must be 16-byte aligned");
  const HprepLayout hl = hprep_layout(s->M, s->ldH);
  const bf16_t* Hb = (const bf16_t*)((const char*)hprep + hl.off_Hb);
  return conv_forward_bf16_any(*s, u, Hb, hl.ldHb, (bf16_t*)m, (hipStream_t)stream);
}

int lmufft_conv_backward_bf16(const lmufft_shape* s, const void* gm, const void* hprep, float* gu, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (!s || s->B <= 0 || s->T <= 0 || s->M <= 0) return set_error("conv: bad shape");
  if (s->T > s->ldH) return set_error("sequence length %d exceeds the impulse response length %d", s->T, s->ldH);
  if (!gm || !hprep || !gu) return set_error("conv: NULL pointer");
  if (!al16(hprep)) return set_error("conv: hprep must be 16-byte aligned");
  lmufft_shape sh = *s;
  sh.D = std::max(sh.D, 1);
  sh.Hd = std::max(sh.Hd, 1);
  const BwdPlanB p = plan_backward_b(sh);
  const size_t need = p.off_ga - p.off_gupart;
  if (!workspace || workspace_bytes < need || !al16(workspace))
    return set_error("conv backward: 16-byte aligned workspace of %zu bytes needed, %zu given", need, workspace_bytes);
  const HprepLayout hl = hprep_layout(s->M, s->ldH);
  const bf16_t* HbT = (const bf16_t*)((const char*)hprep + hl.off_HbT);
  GaParts unused{nullptr, nullptr, nullptr, 0, 0};
  return conv_backward_bf16_any(sh, (const bf16_t*)gm, hl, HbT, p, (float*)workspace, nullptr, gu, false, &unused,
                                (hipStream_t)stream);
}

}  // extern "C"

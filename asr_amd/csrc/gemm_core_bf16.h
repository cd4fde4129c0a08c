// Block-level bf16 GEMM building blocks for gfx950 (CDNA4): bf16 operands,
// fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// A 256-thread workgroup (4 waves, one per SIMD) owns a BM x BN output tile.
// Operand fragments of the MFMA: lane l (r = l & 31, hh = l >> 5) supplies
// A[row r][k = 8*hh + j] and B[k = 8*hh + j][col r], j = 0..7 -- eight
// consecutive k, i.e. one 16-byte LDS read when the LDS image is k-contiguous.
//
// Two kinds of LDS operand:
//   LdsOperandKC   source rows are k-contiguous ([m;x], W_h, H, gm ...).
//       LDS image [rows][BK + 8] bf16; the 144-byte row stride puts 16
//       consecutive rows on 16 distinct 16-byte slots: ds_read_b128 fragments
//       are bank-conflict-free.  Staging global -> VGPR -> LDS, the next tile's
//       global loads issued before the current tile's MFMAs.
//   LdsOperandTR   source rows are contiguous along the OUTPUT dimension and
//       strided along k (both operands of the weight gradient, whose k is the
//       frame index).  LDS image [BK][rows + pad] is a straight copy of the
//       global rows; the transposition happens in the read:
//       ds_read_b64_tr_b16 hands lane i of each 16-lane group column i of a
//       4(k) x 16(row) block.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_core_f32.h"  // kThreads, f32x16, WavePos

namespace lmufft {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int BM_, int BN_, int BK_, int WGM_, int WGN_>
struct TileB {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WGM = WGM_, WGN = WGN_;
  static constexpr int NW = WGM_ * WGN_;  // waves per workgroup
  static_assert(NW == 4 || NW == 8, "four or eight waves per workgroup");
  static constexpr int THREADS = 64 * NW;
  static constexpr int WM = BM / WGM, WN = BN / WGN;
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile is made of 32x32 MFMA tiles");
  static constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(BK % 16 == 0, "k-steps of 16");
};

__device__ __forceinline__ bf16x8 bzero8() {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (bf16_t)0.f;
  return v;
}

// eight consecutive bf16 of `base` starting at idx; elements whose position
// pos + j reaches `limit`, and the whole vector when !ok, read as zero.  With
// VEC the caller guarantees 16-byte alignment and pos % 8 == limit % 8 == 0;
// a masked vector is loaded from kZeroBytes (select on the address, not on
// the loaded value: see gemm_core_f32.h).
template <bool VEC>
__device__ __forceinline__ bf16x8 bload8(const bf16_t* __restrict__ base, long idx, int pos, int limit,
                                         bool ok = true) {
  if (VEC) {
    const bf16_t* p = (ok && pos < limit) ? base + idx : reinterpret_cast<const bf16_t*>(kZeroBytes);
    return *reinterpret_cast<const bf16x8*>(p);
  } else {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (ok && pos + j < limit) ? base[idx + j] : (bf16_t)0.f;
    return v;
  }
}

// Loader protocol as in gemm_core_f32.h: raw_t fetch(r, k) issues the global
// loads of one 16-byte staging vector, bf16x8 finish(raw) does the element-wise
// work and runs only when the tile is written to LDS.
template <class F>
struct PlainLoadB {
  F f;
  typedef bf16x8 raw_t;
  __device__ __forceinline__ raw_t fetch(int r, int k) const { return f(r, k); }
  __device__ __forceinline__ bf16x8 finish(const raw_t& v) const { return v; }
};
template <class F>
__device__ __forceinline__ PlainLoadB<F> plain_load_b(F f) {
  return PlainLoadB<F>{f};
}

// ---- k-contiguous operand -------------------------------------------------
template <int R, int BK, class Load>
struct LdsOperandKC {
  static constexpr int LDK = BK + 8;
  static constexpr int ELEMS = R * LDK;
  static_assert((R * BK / 8) % kThreads == 0, "each thread stages a whole number of 16-byte vectors");
  static constexpr int NV = R * BK / 8 / kThreads;
  bf16_t* S;
  const Load& load;
  typename Load::raw_t regs[NV];
  __device__ __forceinline__ LdsOperandKC(bf16_t* s, const Load& l) : S(s), load(l) {}
  // load.fetch(r, k): elements k..k+7 of tile row r
  __device__ __forceinline__ void fetch(int k0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      regs[i] = load.fetch(idx / (BK / 8), k0 + (idx % (BK / 8)) * 8);
    }
  }
  __device__ __forceinline__ void store() {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      *reinterpret_cast<bf16x8*>(&S[(idx / (BK / 8)) * LDK + (idx % (BK / 8)) * 8]) = load.finish(regs[i]);
    }
  }
  // fragment of tile row `row` for the k-step starting at tile-local k = kl (multiple of 16)
  __device__ __forceinline__ bf16x8 frag(int row, int /*k0*/, int kl, int hh) const {
    return *reinterpret_cast<const bf16x8*>(&S[row * LDK + kl + 8 * hh]);
  }
};

// ---- output-contiguous (k-strided) operand, transposed on read -------------
template <int R, int BK, int PAD, class Load>
struct LdsOperandTR {
  static constexpr int LD = R + PAD;  // elements per k row
  static constexpr int ELEMS = BK * LD;
  static_assert((LD * 2) % 256 == 64 || (LD * 2) % 256 == 192, "four k rows must land on distinct bank quarters");
  static_assert((R * BK / 8) % kThreads == 0, "each thread stages a whole number of 16-byte vectors");
  static constexpr int NV = R * BK / 8 / kThreads;
  bf16_t* S;
  const Load& load;
  typename Load::raw_t regs[NV];
  __device__ __forceinline__ LdsOperandTR(bf16_t* s, const Load& l) : S(s), load(l) {}
  // load.fetch(r, k): eight consecutive output-dimension elements r..r+7 at depth k
  __device__ __forceinline__ void fetch(int k0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      regs[i] = load.fetch((idx % (R / 8)) * 8, k0 + idx / (R / 8));
    }
  }
  // seen(v) is called with every finished vector (used for the bias-gradient column sums)
  template <class Seen>
  __device__ __forceinline__ void store(Seen&& seen) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      const bf16x8 v = load.finish(regs[i]);
      seen(v);
      *reinterpret_cast<bf16x8*>(&S[(idx / (R / 8)) * LD + (idx % (R / 8)) * 8]) = v;
    }
  }
  __device__ __forceinline__ void store() {
    store([](const bf16x8&) {});
  }
  // fragment for the 32 output rows starting at row0 and the k-step at tile-local kl.
  // In each 16-lane group, lane 4q+p addresses k row q, elements 4p..4p+3, and
  // receives the four k values of column (lane & 15).
  __device__ __forceinline__ bf16x8 frag32(int row0, int kl) const {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, g1 = (lane >> 4) & 1, hh = lane >> 5;
    const int q = i >> 2, p = i & 3;
    const bf16_t* a0 = &S[(kl + 8 * hh + q) * LD + row0 + 16 * g1 + 4 * p];
    typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(a0 + 4 * LD));
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = lo[j];
      v[4 + j] = hi[j];
    }
    return v;
  }
};

template <class T>
__device__ __forceinline__ void tileb_zero(f32x16 (&acc)[T::TM][T::TN]) {
#pragma unroll
  for (int a = 0; a < T::TM; ++a)
#pragma unroll
    for (int b = 0; b < T::TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
}

// acc += A[:, kbeg:kend] B[kbeg:kend, :].  OA / OB provide fetch/store/frag.
template <class T, class OA, class OB>
__device__ __forceinline__ void gemm_loop_kc(OA& a, OB& b, f32x16 (&acc)[T::TM][T::TN], int kbeg, int kend,
                                             const WavePos<T>& w) {
  if (kbeg < kend) {
    a.fetch(kbeg);
    b.fetch(kbeg);
  }
  for (int k0 = kbeg; k0 < kend; k0 += T::BK) {
    a.store();
    b.store();
    __syncthreads();
    if (k0 + T::BK < kend) {
      a.fetch(k0 + T::BK);
      b.fetch(k0 + T::BK);
    }
#pragma unroll
    for (int kl = 0; kl < T::BK; kl += 16) {
      bf16x8 af[T::TM], bfr[T::TN];
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm) af[tm] = a.frag(w.wm0 + tm * 32 + w.r, k0, kl, w.hh);
#pragma unroll
      for (int tn = 0; tn < T::TN; ++tn) bfr[tn] = b.frag(w.wn0 + tn * 32 + w.r, k0, kl, w.hh);
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
    }
    __syncthreads();
  }
}

template <class T, class F>
__device__ __forceinline__ void tileb_epilogue(const f32x16 (&acc)[T::TM][T::TN], const WavePos<T>& w, F&& f) {
  tile_epilogue<T>(acc, w, f);
}

}  // namespace lmufft

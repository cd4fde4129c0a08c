// Block-level f32 GEMM building block for gfx950 (CDNA4), shared by every
// GEMM-shaped kernel of the LMUFFT layer.
//
// One 256-thread workgroup (4 waves, one per SIMD) computes a BM x BN tile of
// C = A * B from fp32 operands with fp32 accumulation: every product as six bf16
// products of a three-way split of its factors on v_mfma_f32_32x32x16_bf16
// (tile_compute_split below: about one fp32 rounding of error per product), or,
// built with -DLMUFFT_F32_MFMA, with v_mfma_f32_32x32x2_f32, whose result is
// bit-for-bit a k-ordered fp32 fma chain.  Either way the fp32 path has the
// numerics class of the reference's fp32 GEMMs (measured against its goldens:
// 4e-7 against 3e-7 of the largest magnitude, tools/f32_error.py).
//
// Operands are described by loader functors, so that a kernel can generate an
// operand on the fly (Toeplitz matrix from u, ReLU-masked gradient, [m;x]
// concatenation) instead of materialising it in HBM:
//
//   layout KC ("k contiguous"): load(r, k) returns elements k..k+3 of tile row r.
//       LDS image [rows][BK+4]; fragments are read with ds_read_b128 (4 k per
//       lane).  Row stride 36 floats = 144 B puts 16 consecutive rows on 16
//       different 16-byte slots of the 256-byte bank row: conflict-free.
//   layout NC ("row contiguous"): load(r, k) returns rows r..r+3 at depth k.
//       LDS image [BK][rows]; fragments are read with ds_read_b32 (consecutive
//       lanes -> consecutive rows: conflict-free), i.e. the transposition that
//       a k-strided source needs costs nothing.
//
// Both layouts feed MFMA step j of k-group kk with k = 8*kk + 4*(lane>>5) + j,
// so a KC operand may be paired with an NC operand.
//
// Staging is global -> registers -> LDS with the next tile's global loads
// issued before the current tile's MFMAs (their latency hides under the
// 64-cycle f32 MFMAs).
#pragma once
#include <hip/hip_runtime.h>

namespace lmufft {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
enum Layout { KC = 0, NC = 1 };

template <int BM_, int BN_, int BK_, int WGM_, int WGN_, int ALAY_, int BLAY_>
struct Tile {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WGM = WGM_, WGN = WGN_;
  static constexpr int ALAY = ALAY_, BLAY = BLAY_;
  static_assert(WGM_ * WGN_ == 4, "four waves per workgroup");
  static constexpr int WM = BM / WGM, WN = BN / WGN;  // per-wave output
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile is made of 32x32 MFMA tiles");
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int LDK = BK + 4;  // KC row stride in floats
  static_assert(BK % 8 == 0, "k-groups of 8");
  static constexpr int A_FLOATS = ALAY == KC ? BM * LDK : BK * BM;
  static constexpr int B_FLOATS = BLAY == KC ? BN * LDK : BK * BN;
  static_assert((BM * BK / 4) % kThreads == 0 && (BN * BK / 4) % kThreads == 0,
                "each thread stages a whole number of float4");
  static constexpr int A_VECS = BM * BK / 4 / kThreads;
  static constexpr int B_VECS = BN * BK / 4 / kThreads;
  static constexpr int SMEM_FLOATS = A_FLOATS + B_FLOATS;
};

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// 32 bytes of zeros in device memory: masked-out vector loads are redirected
// here instead of being zeroed after the fact.  A select on a loaded VALUE
// makes the compiler wait for the load right where it was issued; a select on
// the ADDRESS lets every staging load of a tile stay in flight behind the MFMAs.
static __device__ __attribute__((aligned(32))) unsigned int kZeroBytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};

// *p when ok, else 0 -- as a select on the ADDRESS (masked elements read a line of zeros).  Written as
// `ok ? *p : 0.f` the load sits in a branch of its own with an s_waitcnt vmcnt(0) behind it, so a run
// of masked loads becomes as many serial memory round trips; this form leaves them all in flight.
__device__ __forceinline__ float load_or_zero(const float* p, bool ok) {
  return *(ok ? p : reinterpret_cast<const float*>(kZeroBytes));
}

// base[idx .. idx+3]; elements at or beyond `limit` (position pos + j) and the
// whole vector when !ok read as zero.  With VEC the caller guarantees
// idx % 4 == 0, limit % 4 == 0 and a 16-byte aligned base.
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* __restrict__ base, long idx, int pos, int limit,
                                       bool ok = true) {
  if (VEC) {
    const float* p = (ok && pos < limit) ? base + idx : reinterpret_cast<const float*>(kZeroBytes);
    return *reinterpret_cast<const f32x4*>(p);
  } else {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (ok && pos + j < limit) ? base[idx + j] : 0.f;
    return v;
  }
}

// Operand loaders have two halves so that nothing touches loaded data before
// the MFMAs of the current tile have been issued:
//   raw_t fetch(r, k)  -- issues the global loads of one staging vector;
//   f32x4 finish(raw)  -- element-wise work on the loaded values (masking,
//                         adds), run only when the tile is written to LDS.
// KC layout: fetch(r, k) covers elements k..k+3 of tile row r;
// NC layout: fetch(r, k) covers tile rows r..r+3 at depth k.
template <class F>
struct PlainLoad {
  F f;
  typedef f32x4 raw_t;
  __device__ __forceinline__ raw_t fetch(int r, int k) const { return f(r, k); }
  __device__ __forceinline__ f32x4 finish(const raw_t& v) const { return v; }
};
template <class F>
__device__ __forceinline__ PlainLoad<F> plain_load(F f) {
  return PlainLoad<F>{f};
}

template <int R, int BK, int LAY, int NV, class Load>
__device__ __forceinline__ void tile_fetch(typename Load::raw_t (&regs)[NV], const Load& load, int k0) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + i * kThreads;
    if (LAY == KC) {
      const int r = idx / (BK / 4), c = idx % (BK / 4);
      regs[i] = load.fetch(r, k0 + c * 4);
    } else {
      const int k = idx / (R / 4), r4 = idx % (R / 4);
      regs[i] = load.fetch(r4 * 4, k0 + k);
    }
  }
}

template <int R, int BK, int LAY, int NV, class Load>
__device__ __forceinline__ void tile_store(float* S, const typename Load::raw_t (&regs)[NV], const Load& load) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + i * kThreads;
    const f32x4 v = load.finish(regs[i]);
    if (LAY == KC) {
      const int r = idx / (BK / 4), c = idx % (BK / 4);
      *reinterpret_cast<f32x4*>(&S[r * (BK + 4) + c * 4]) = v;
    } else {
      const int k = idx / (R / 4), r4 = idx % (R / 4);
      *reinterpret_cast<f32x4*>(&S[k * R + r4 * 4]) = v;
    }
  }
}

// Workgroups are dealt to the chip's 8 XCDs round-robin by linear workgroup id, and every XCD
// has its own L2.  Workgroups that read the same operand panel are therefore given ids that are
// neighbours ON ONE XCD: with a 1-D grid of xcd_grid(groups, inner) workgroups, xcd_slot maps
// blockIdx.x -> (group, member) such that group % 8 == blockIdx.x % 8 and the `inner` members
// of a group follow each other in that XCD's dispatch order.  Slots with group >= groups are idle.
struct XcdSlot {
  int group, member;
};
__device__ __forceinline__ XcdSlot xcd_slot(int inner) {
  const int id = blockIdx.x, seq = id >> 3;
  return {(seq / inner) * 8 + (id & 7), seq % inner};
}
static inline int xcd_grid(long groups, int inner) { return (int)(8 * ((groups + 7) / 8) * inner); }

template <class T>
struct WavePos {
  int wm0, wn0, r, hh;
  __device__ __forceinline__ WavePos() {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    wm0 = (wave / T::WGN) * T::WM;
    wn0 = (wave % T::WGN) * T::WN;
    r = lane & 31;
    hh = lane >> 5;
  }
};

template <class T>
__device__ __forceinline__ void tile_zero(f32x16 (&acc)[T::TM][T::TN]) {
#pragma unroll
  for (int a = 0; a < T::TM; ++a)
#pragma unroll
    for (int b = 0; b < T::TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
}

// ---- fp32 products on the bf16 matrix cores -------------------------------------------------------
// x = hi + mid + lo with three bf16 (8 significant bits each: the 24 of an fp32, each difference exact),
// and a * b = sum of the six products whose weight is at least 2^-16 of the largest:
//   hi hi + (hi mid + mid hi) + (hi lo + lo hi + mid mid)      (mid lo, lo mid, lo lo: below 2^-24)
// bf16 x bf16 is exact in fp32 and the MFMA accumulates in fp32, so a product carries a relative error of about
// 2^-23 -- one fp32 rounding -- instead of being bit-for-bit the fma chain of v_mfma_f32_32x32x2_f32.  Six
// v_mfma_f32_32x32x16_bf16 (8 passes each) cover 16 k where the fp32 instruction needs eight of 16 passes:
// 2.7 times fewer matrix-core cycles.  The split costs ~5 VALU operations per operand element and k-step, which
// run beside the MFMAs.  -DLMUFFT_F32_MFMA restores the fp32 instruction.
typedef __bf16 split_bf16x8 __attribute__((ext_vector_type(8)));
struct Split3 {
  split_bf16x8 hi, mid, lo;
};
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
  Split3 s;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)x[j];
    const float r1 = x[j] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    s.hi[j] = h;
    s.mid[j] = m;
    s.lo[j] = (__bf16)r2;
  }
  return s;
}

template <class T>
__device__ __forceinline__ void tile_compute_split(const float* As, const float* Bs,
                                                   f32x16 (&acc)[T::TM][T::TN], const WavePos<T>& w) {
  static_assert(T::BK % 16 == 0, "k-steps of 16");
#pragma unroll
  for (int ks = 0; ks < T::BK / 16; ++ks) {
    Split3 a[T::TM], b[T::TN];
    const int kb = ks * 16 + w.hh * 8;  // this lane's eight k of the step
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm) {
      const int row = w.wm0 + tm * 32 + w.r;
      float x[8];
      if (T::ALAY == KC) {
        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(&As[row * T::LDK + kb]);
        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(&As[row * T::LDK + kb + 4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = lo4[j], x[4 + j] = hi4[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = As[(kb + j) * T::BM + row];
      }
      a[tm] = split3(x);
    }
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn) {
      const int col = w.wn0 + tn * 32 + w.r;
      float x[8];
      if (T::BLAY == KC) {
        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(&Bs[col * T::LDK + kb]);
        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(&Bs[col * T::LDK + kb + 4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = lo4[j], x[4 + j] = hi4[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = Bs[(kb + j) * T::BN + col];
      }
      b[tn] = split3(x);
    }
    // smallest terms first
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < T::TN; ++tn) {
        f32x16 c = acc[tm][tn];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm].mid, b[tn].mid, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm].lo, b[tn].hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm].hi, b[tn].lo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm].mid, b[tn].hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm].hi, b[tn].mid, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm].hi, b[tn].hi, c, 0, 0, 0);
        acc[tm][tn] = c;
      }
  }
}

template <class T>
__device__ __forceinline__ void tile_compute(const float* As, const float* Bs,
                                             f32x16 (&acc)[T::TM][T::TN], const WavePos<T>& w) {
#ifndef LMUFFT_F32_MFMA
  if constexpr (T::BK % 16 == 0) {
    tile_compute_split<T>(As, Bs, acc, w);
    return;
  }
#endif
#pragma unroll
  for (int kk = 0; kk < T::BK / 8; ++kk) {
    f32x4 a[T::TM], b[T::TN];
    const int kb = kk * 8 + w.hh * 4;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm) {
      const int row = w.wm0 + tm * 32 + w.r;
      if (T::ALAY == KC) {
        a[tm] = *reinterpret_cast<const f32x4*>(&As[row * T::LDK + kb]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[tm][j] = As[(kb + j) * T::BM + row];
      }
    }
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn) {
      const int col = w.wn0 + tn * 32 + w.r;
      if (T::BLAY == KC) {
        b[tn] = *reinterpret_cast<const f32x4*>(&Bs[col * T::LDK + kb]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[tn][j] = Bs[(kb + j) * T::BN + col];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
  }
}

// acc += A[:, kbeg:kend] * B[kbeg:kend, :] for this workgroup's tile.  Loaders
// must return zero outside the real operand.  Every thread of the workgroup
// must call it with the same kbeg/kend (it contains barriers).
template <class T, class LA, class LB>
__device__ __forceinline__ void gemm_tile(float* smem, f32x16 (&acc)[T::TM][T::TN], const LA& la, const LB& lb,
                                          int kbeg, int kend, const WavePos<T>& w) {
  float* As = smem;
  float* Bs = smem + T::A_FLOATS;
  typename LA::raw_t ra[T::A_VECS];
  typename LB::raw_t rb[T::B_VECS];
  if (kbeg < kend) {
    tile_fetch<T::BM, T::BK, T::ALAY>(ra, la, kbeg);
    tile_fetch<T::BN, T::BK, T::BLAY>(rb, lb, kbeg);
  }
  for (int k0 = kbeg; k0 < kend; k0 += T::BK) {
    tile_store<T::BM, T::BK, T::ALAY>(As, ra, la);
    tile_store<T::BN, T::BK, T::BLAY>(Bs, rb, lb);
    __syncthreads();
    if (k0 + T::BK < kend) {
      tile_fetch<T::BM, T::BK, T::ALAY>(ra, la, k0 + T::BK);
      tile_fetch<T::BN, T::BK, T::BLAY>(rb, lb, k0 + T::BK);
    }
    tile_compute<T>(As, Bs, acc, w);
    __syncthreads();
  }
}

// f(tm, tn, i, m, n, value) for every accumulator element; m, n are tile-local,
// (tm, tn, i) identify the MFMA tile and accumulator register.
// C/D map of the 32x32 MFMAs: column = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
template <class T, class F>
__device__ __forceinline__ void tile_epilogue_ix(const f32x16 (&acc)[T::TM][T::TN], const WavePos<T>& w, F&& f) {
#pragma unroll
  for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = w.wm0 + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * w.hh;
        const int n = w.wn0 + tn * 32 + w.r;
        f(tm, tn, i, m, n, acc[tm][tn][i]);
      }
}

// f(m, n, value) for every accumulator element.
template <class T, class F>
__device__ __forceinline__ void tile_epilogue(const f32x16 (&acc)[T::TM][T::TN], const WavePos<T>& w, F&& f) {
  tile_epilogue_ix<T>(acc, w, [&](int, int, int, int m, int n, float v) { f(m, n, v); });
}

// Per-column / per-row epilogue operands (bias, rank-1 terms) are loaded into
// registers in one batch BEFORE the epilogue; loading them per accumulator
// element makes 16*TM*TN dependent global loads, each waited for in turn.
template <class T, class G>
__device__ __forceinline__ void preload_cols(float (&c)[T::TN], const WavePos<T>& w, G&& g) {
#pragma unroll
  for (int tn = 0; tn < T::TN; ++tn) c[tn] = g(w.wn0 + tn * 32 + w.r);
}
template <class T, class G>
__device__ __forceinline__ void preload_rows(float (&r)[T::TM][16], const WavePos<T>& w, G&& g) {
#pragma unroll
  for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
    for (int i = 0; i < 16; ++i) r[tm][i] = g(w.wm0 + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * w.hh);
}

// Adds every element of a workgroup's accumulator tile P[mm][nn] to the diagonal
// s = s_origin + mm - nn it belongs to: the tile goes to LDS (`P`, BM * (BN + 1) floats), then one
// thread per diagonal sums it top to bottom -- a fixed order, unlike LDS atomics.  Rows mm > mm_last
// and diagonals s < 0 are left out.  All threads of the workgroup call it; `diag` is theirs alone
// between the two barriers it contains.
template <class T>
__device__ __forceinline__ void add_tile_diagonals(const f32x16 (&acc)[T::TM][T::TN], const WavePos<T>& w, float* P,
                                                   float* diag, int s_origin, int mm_last) {
  constexpr int LDP = T::BN + 1;
  static_assert(T::BM + T::BN - 1 <= kThreads, "one thread per diagonal");
  tile_epilogue<T>(acc, w, [&](int mm, int nn, float v) { P[mm * LDP + nn] = v; });
  __syncthreads();
  const int off = (int)threadIdx.x - (T::BN - 1);  // mm - nn of this thread's diagonal
  const int s = s_origin + off;
  if (off < T::BM && s >= 0) {
    const int lo = off > 0 ? off : 0;
    int hi = T::BN - 1 + off < T::BM - 1 ? T::BN - 1 + off : T::BM - 1;
    hi = hi < mm_last ? hi : mm_last;
    float sum = 0.f;
    for (int mm = lo; mm <= hi; ++mm) sum += P[mm * LDP + mm - off];
    if (lo <= hi) diag[s] += sum;
  }
  __syncthreads();
}

}  // namespace lmufft

// bf16 GEMM main loop with direct global->LDS staging (global_load_lds_dwordx4)
// for gfx950, used by the kernels whose operands are plain k-contiguous rows.
//
// Why: with register staging a k-tile costs 8 global loads + 8 ds_write per
// thread and 32 VGPRs, which caps the kernel at 2 waves per SIMD and leaves the
// loads' latency exposed.  The LDS-DMA form needs no staging registers and no
// ds_write; the loads of k-tile i+1 are in flight while tile i is multiplied.
//
// LDS image of one operand stage: [R rows][64 bf16] = R x 128 bytes, unpadded
// (a wave-instruction writes 64 lanes x 16 B = 1 KiB = 8 whole rows, so rows
// cannot be padded).  Bank conflicts are removed by an XOR swizzle instead:
// the 16-byte chunk d (0..7) of row r lives at chunk position d ^ ((r >> 1) & 7).
// The LDS-DMA destination is linear in the lane index, so the swizzle is
// applied to the per-lane SOURCE address (lane -> row r, position c, loads data
// chunk c ^ ((r >> 1) & 7)) and again on the fragment read.  For the 16 rows
// that one ds_read_b128 lane group touches, (r & 1, (r >> 1) & 7) takes 16
// distinct values: conflict-free.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_core_bf16.h"

namespace lmufft {

constexpr int kGldsBK = 64;           // bf16 elements per k-tile of the two-stage loop
constexpr int kGldsRowBytes = 128;    // = kGldsBK * 2

// One operand staged by LDS-DMA.  Everything that does not change from k-tile
// to k-tile is computed ONCE per lane: for each of its NI staging
// instructions the lane keeps a row pointer (two when the row is the
// concatenation of two tensors, as [m;x]) and its swizzled column offset.
// Issuing a k-tile then costs one 64-bit add and a select per instruction;
// recomputing rows, panels and bounds inside the loop took ~150 vector and
// ~100 scalar instructions per k-tile and was as expensive as the MFMAs.
//
// setup(rowfn, split, kmax): rowfn(r, p0, p1) -> bool fills, for tile row r,
//   p0 such that p0[k] is element k of the row for k < split and
//   p1 such that p1[k] is element k for split <= k < kmax; returns false for
//   rows outside the operand.  Elements at k >= kmax read as zero.
//
// BK = 64 (128-byte LDS rows, 8 rows per wave-instruction, swizzle (r >> 1) & 7) or
// BK = 32 ( 64-byte LDS rows, 16 rows per wave-instruction, swizzle (r >> 2) & 3):
// in both cases the 16 rows of a ds_read_b128 lane group fall on 16 distinct
// 16-byte slots of the 256-byte bank row.
template <int R, int NW, bool TWO_PIECES, int BK = 64>
struct GldsOperand {
  static_assert(BK == 64 || BK == 32, "k-tile of 64 or 32");
  static constexpr int ROWB = BK * 2;        // bytes per LDS row
  static constexpr int RPI = 1024 / ROWB;    // rows per wave-instruction
  static constexpr int CPR = ROWB / 16;      // 16-byte chunks per row
  static_assert(R % (RPI * NW) == 0, "every wave stages whole rows");
  static constexpr int STAGE_BYTES = R * ROWB;
  static constexpr int NI = R / (RPI * NW);  // wave-instructions per wave and stage
  __device__ static __forceinline__ int swz(int r) { return BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
  unsigned char* base;                     // LDS, two stages back to back
  const bf16_t* p0[NI];
  const bf16_t* p1[NI];
  unsigned okmask;
  int kof[NI];  // swizzled column offset (elements) of every staging instruction
  int split, kmax, wave_off;
  const bf16_t* zero;

  template <class RowFn>
  __device__ __forceinline__ GldsOperand(unsigned char* lds, RowFn&& rowfn, int split_, int kmax_)
      : base(lds), okmask(0), split(split_), kmax(kmax_) {
    zero = reinterpret_cast<const bf16_t*>(kZeroBytes);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    wave_off = wave * (R / NW) * ROWB;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = wave * (R / NW) + RPI * i + lane / CPR;
      const bf16_t *q0 = zero, *q1 = zero;
      const bool ok = rowfn(r, q0, q1);
      const int d = (lane % CPR) ^ swz(r);
      p0[i] = q0 + 8 * d;
      p1[i] = q1 + 8 * d;
      okmask |= (ok ? 1u : 0u) << i;
      kof[i] = 8 * d;
    }
  }

  __device__ __forceinline__ void issue(int stage, int k0) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = k0 + kof[i];
      const bf16_t* p = (TWO_PIECES && k >= split) ? p1[i] : p0[i];
      p = (((okmask >> i) & 1u) && k < kmax) ? p + k0 : zero;
      unsigned char* dst = base + stage * STAGE_BYTES + wave_off + i * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  }
  __device__ __forceinline__ bf16x8 frag(int stage, int row, int kl, int hh) const {
    const int d = (kl >> 3) + hh;
    const unsigned char* a = base + stage * STAGE_BYTES + row * ROWB + ((d ^ swz(row)) << 4);
    return *reinterpret_cast<const bf16x8*>(a);
  }
};

// acc += A[:, kbeg:kend] B[kbeg:kend, :]; kbeg a multiple of 8.  Two LDS stages.
// All threads of the workgroup must call it with the same bounds.
template <class T, class OA, class OB>
__device__ __forceinline__ void gemm_loop_glds(const OA& a, const OB& b, f32x16 (&acc)[T::TM][T::TN], int kbeg,
                                               int kend, const WavePos<T>& w) {
  static_assert(T::BK == kGldsBK, "k-tile of 64");
  if (kbeg < kend) {
    a.issue(0, kbeg);
    b.issue(0, kbeg);
  }
  int stage = 0;
  for (int k0 = kbeg; k0 < kend; k0 += kGldsBK, stage ^= 1) {
    // this wave's LDS-DMA writes have landed; after the barrier everybody's have,
    // and everybody has finished reading the other stage
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (k0 + kGldsBK < kend) {
      a.issue(stage ^ 1, k0 + kGldsBK);
      b.issue(stage ^ 1, k0 + kGldsBK);
    }
#pragma unroll
    for (int kl = 0; kl < kGldsBK; kl += 16) {
      bf16x8 af[T::TM], bfr[T::TN];
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm) af[tm] = a.frag(stage, w.wm0 + tm * 32 + w.r, kl, w.hh);
#pragma unroll
      for (int tn = 0; tn < T::TN; ++tn) bfr[tn] = b.frag(stage, w.wn0 + tn * 32 + w.r, kl, w.hh);
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
    }
  }
  __syncthreads();  // the stages may be reused (epilogue staging) after this
}

// Deeper form: NS LDS stages of BK = 32, NS - 1 k-tiles in flight at all times.
// The two-stage loop drains the load queue at every k-tile (vmcnt(0) before the
// barrier), so a CU has one tile in flight only while it multiplies the
// previous one; here the wait is COUNTED -- all but the newest NS - 2 tiles --
// and the barrier is a raw s_barrier, so loads stay in flight across it.
// Every iteration issues the same number of LDS-DMA instructions (tiles beyond
// kend are fetched from the zero line), which is what makes the count valid.
// All LDS of the kernel must live in the one dynamic array the stages are carved from.
template <class T, int NS, class OA, class OB>
__device__ __forceinline__ void gemm_loop_glds_deep(const OA& a, const OB& b, f32x16 (&acc)[T::TM][T::TN],
                                                    int kbeg, int kend, const WavePos<T>& w) {
  static_assert(T::BK == 32, "k-tile of 32");
  constexpr int G = OA::NI + OB::NI;          // LDS-DMA instructions per wave and k-tile
  constexpr int KEEP = G * (NS - 2);          // may stay outstanding when a tile is consumed
  static_assert(KEEP <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    a.issue(s, kbeg + s * T::BK);
    b.issue(s, kbeg + s * T::BK);
  }
  int stage = 0, pf = NS - 1;
  for (int k0 = kbeg; k0 < kend; k0 += T::BK) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
    __builtin_amdgcn_s_barrier();
    a.issue(pf, k0 + (NS - 1) * T::BK);
    b.issue(pf, k0 + (NS - 1) * T::BK);
#pragma unroll
    for (int kl = 0; kl < T::BK; kl += 16) {
      bf16x8 af[T::TM], bfr[T::TN];
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm) af[tm] = a.frag(stage, w.wm0 + tm * 32 + w.r, kl, w.hh);
#pragma unroll
      for (int tn = 0; tn < T::TN; ++tn) bfr[tn] = b.frag(stage, w.wn0 + tn * 32 + w.r, kl, w.hh);
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
    }
    stage = stage + 1 == NS ? 0 : stage + 1;
    pf = pf + 1 == NS ? 0 : pf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing zero-line tiles
  __syncthreads();  // the stages may be reused (epilogue staging) after this
}

// Writes the workgroup's accumulator tile as bf16 rows through LDS so that HBM
// sees 16-byte stores of whole row segments instead of 2-byte column stores.
//   value(tm, tn, i, m, n, v)  -> float to store for tile-local (m, n)
//   out[(row0 + m) * ld + col0 + n], 0 <= rows < row_limit (row0 may be negative), cols < col_limit,
//   col_limit % 8 == 0 and 16-byte aligned rows (the caller checks); out == NULL: only the LDS image;
//   row_first: rows below it are not stored either.
// `lds` needs BM * (BN + 8) * 2 bytes.  Returns with the tile still in LDS
// (image [BM][BN + 8]) so that callers can copy rows out of it again.
//   chunk(row, col, v) is called by the lane that stores the 8 elements v to out[row][col .. col + 8)
template <class T, class V, class C>
__device__ __forceinline__ void store_tile_bf16(const f32x16 (&acc)[T::TM][T::TN], const WavePos<T>& w,
                                                bf16_t* lds, bf16_t* __restrict__ out, long ld, long row0,
                                                int col0, long row_limit, int col_limit, V&& value, C&& chunk,
                                                long row_first = 0) {
  constexpr int LDT = T::BN + 8;
  tile_epilogue_ix<T>(acc, w, [&](int tm, int tn, int i, int m, int n, float v) {
    lds[m * LDT + n] = (bf16_t)value(tm, tn, i, m, n, v);
  });
  __syncthreads();
  constexpr int CH = T::BN / 8;  // 16-byte chunks per tile row
  for (int idx = threadIdx.x; idx < T::BM * CH; idx += T::THREADS) {
    const int m = idx / CH, c = idx % CH;
    const long row = row0 + m;
    const int col = col0 + c * 8;
    if (out && row >= row_first && row < row_limit && col < col_limit) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(&lds[m * LDT + c * 8]);
      *reinterpret_cast<bf16x8*>(out + row * ld + col) = v;
      chunk(row, col, v);
    }
  }
}

template <class T, class V>
__device__ __forceinline__ void store_tile_bf16(const f32x16 (&acc)[T::TM][T::TN], const WavePos<T>& w,
                                                bf16_t* lds, bf16_t* __restrict__ out, long ld, long row0,
                                                int col0, long row_limit, int col_limit, V&& value) {
  store_tile_bf16<T>(acc, w, lds, out, ld, row0, col0, row_limit, col_limit, value, [](long, int, const bf16x8&) {});
}

}  // namespace lmufft

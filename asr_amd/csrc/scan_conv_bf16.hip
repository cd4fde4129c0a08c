// Causal convolution with the Legendre impulse response for LONG sequences, bf16 activations, on gfx950:
// the state-space ("scan") route.
//
// Reference arithmetic: m[b,t,k] = sum_{s<=t} H[k,t-s] u[b,s] (src/lmu.py:303-312) with H[:,t] = A^t B
// (src/lmu.py:275-283), and its adjoint gu[b,s] = sum_{t>=s} sum_k H[k,t-s] gm[b,t,k].
//
// The Toeplitz contraction costs T^2 M / 2 multiply-adds per utterance.  Because H is the impulse
// response of x_t = A x_{t-1} + B u_t, everything a chunk of L frames needs from the frames before
// it is the M-vector X'_c = A x_{cL-1}:
//     m[cL+j] = A^j X'_c + sum_{i<=j} H[:,j-i] u[cL+i]                    (j = 0..L-1)
//     X'_{c+1} = A^L X'_c + sum_{i<L} (A^{L-i} B) u[cL+i]
// With rows (b,c), that is ONE dense GEMM  m_rows[(b,c)][(j,k)] = [X'_c | u_chunk] . Wc  against a
// constant matrix Wc [(M + L) x (L M)] (powers of A on top, the chunk's own Toeplitz block below):
// (M + L) multiply-adds per output instead of T/2 -- 4.7 times fewer at T = 3000, L = 128.  The
// adjoint is the transposed GEMM  [G_c | gu_intra] = gm_rows . Wc^T  followed by the reverse
// recurrence of the costate  Lambda_c = G_c + (A^L)^T Lambda_{c+1},
//     gu[cL+i] = gu_intra[i] + (A^{L-i} B) . Lambda_{c+1}.
// The recurrences over chunks (T/L steps of an M x M matrix-vector product in fp32) run one
// workgroup per utterance with the matrix row in registers.
//
// Kernels: k_scan_rows_fwd (per chunk: input of the recurrence, sample part of the GEMM rows),
//          k_scan_state_fwd (chunk states -> state part of the GEMM rows), k_scan_gemm_fwd (writes m);
//          k_scan_gemm_adj (split over (j,k) into fp32 slabs), k_scan_state_bwd (costates),
//          k_scan_rows_bwd (per chunk: gu).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gemm_glds_bf16.h"
#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

constexpr int kScanL = 128;  // frames per chunk
using ScanTile = TileB<128, 384, 64, 2, 4>;

struct ScanLayout {
  int M, Kp;
  long N;  // L * M: columns of Wc
  size_t off_WcT, off_Wc, off_AL, off_ALT, off_Hs, off_HsT, total;
};
static inline size_t scan_al256(size_t n) { return (n + 255) & ~(size_t)255; }
static ScanLayout scan_layout(int M) {
  ScanLayout l;
  l.M = M;
  l.Kp = (M + kScanL + 63) / 64 * 64;
  l.N = (long)kScanL * M;
  size_t off = 0;
  l.off_WcT = off;
  off += scan_al256(2 * (size_t)l.N * l.Kp);
  l.off_Wc = off;
  off += scan_al256(2 * (size_t)l.Kp * l.N);
  l.off_AL = off;
  off += scan_al256(4 * (size_t)M * M);
  l.off_ALT = off;
  off += scan_al256(4 * (size_t)M * M);
  l.off_Hs = off;
  off += scan_al256(4 * (size_t)kScanL * M);
  l.off_HsT = off;
  off += scan_al256(4 * (size_t)M * kScanL);
  l.total = off;
  return l;
}
bool scan_supported(int M) { return M >= 8 && M % 8 == 0 && M <= 256; }
static inline int scan_chunks(int T) { return (T + kScanL - 1) / kScanL; }

// ---------------------------------------------------------------------------------------------
// The parts of the state computation that are parallel over chunks run on the whole chip, one
// workgroup per (utterance, kScanCB chunks), four lanes per state component / frame; the recurrences
// themselves (k_scan_state_fwd / _bwd) run one workgroup per utterance.
//
// k_scan_rows_fwd: Z_c = sum_i Hs[i] u[cL+i] (fp32, row (b,c) of Zbuf; needed for full chunks only) and the
//   sample part of the GEMM operand rows: Arows[(b,c)][M ..] = [ bf16(u[cL .. cL+L-1]) | 0 ].
// ---------------------------------------------------------------------------------------------
constexpr int kScanCB = 8;  // chunks per workgroup of the row kernels

// dst[idx] = *src(idx) for idx < n, eight loads per thread in flight at a time (a plain loop waits for
// every load before it issues the next).  src(idx) returns an address that is always readable.
template <int THREADS, class Src>
__device__ __forceinline__ void scan_fill_lds(float* dst, int n, Src&& src) {
  for (int base = threadIdx.x; base < n; base += 8 * THREADS) {
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int idx = base + r * THREADS;
      v[r] = *src(idx < n ? idx : -1);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int idx = base + r * THREADS;
      if (idx < n) dst[idx] = v[r];
    }
  }
}

// Sum over the four lanes of a group by DPP quad permutes (no LDS round trip).
__device__ __forceinline__ float scan_quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // 1 0 3 2
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // 2 3 0 1
  return v;
}

template <int MP>
__global__ __launch_bounds__(4 * MP) void k_scan_rows_fwd(const float* __restrict__ u, const float* __restrict__ Hs,
                                                          bf16_t* __restrict__ Arows, float* __restrict__ Zbuf,
                                                          int T, int M, int Kp, int nc) {
  constexpr int L = kScanL, Q = 4, THREADS = 4 * MP, IQ = L / Q;
  __shared__ __attribute__((aligned(16))) float ublk[kScanCB * L];
  const int b = blockIdx.x, c0 = blockIdx.y * kScanCB, tid = threadIdx.x, k = tid >> 2, q = tid & 3;
  const float* ub = u + (long)b * T;
  const bool live = k < M;
  // masked elements are read from a line of zeros: a select on the ADDRESS keeps the loads unconditional
  // and all in flight together (a select on the value becomes a branch and a wait per load)
  const float* zero = reinterpret_cast<const float*>(kZeroBytes);
  float hreg[IQ];
#pragma unroll
  for (int ii = 0; ii < IQ; ++ii) hreg[ii] = *(live ? Hs + (q * IQ + ii) * M + k : zero);
  scan_fill_lds<THREADS>(ublk, kScanCB * L, [&](int idx) { return (idx >= 0 && c0 * L + idx < T) ? ub + c0 * L + idx : zero; });
  __syncthreads();
  const int ccn = min(kScanCB, nc - c0);
  {  // the chunks' own samples and the padding of their rows
    const int wcols = Kp - M;
    bf16_t* Ab = Arows + ((long)b * nc + c0) * Kp + M;
    for (int idx = tid; idx < ccn * wcols; idx += THREADS) {
      const int cc = idx / wcols, i = idx % wcols;
      Ab[(long)cc * Kp + i] = (bf16_t)(i < L ? ublk[cc * L + i] : 0.f);
    }
  }
  float* Zb = Zbuf + ((long)b * nc + c0) * MP;
  for (int cc = 0; cc < ccn; ++cc) {
    const f32x4* uv = reinterpret_cast<const f32x4*>(ublk + cc * L + q * IQ);
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int i4 = 0; i4 < IQ / 4; i4 += 2) {
      const f32x4 va = uv[i4], vb = uv[i4 + 1];
      v0 += hreg[4 * i4] * va[0] + hreg[4 * i4 + 1] * va[1] + hreg[4 * i4 + 2] * va[2] + hreg[4 * i4 + 3] * va[3];
      v1 += hreg[4 * i4 + 4] * vb[0] + hreg[4 * i4 + 5] * vb[1] + hreg[4 * i4 + 6] * vb[2] + hreg[4 * i4 + 7] * vb[3];
    }
    const float v = scan_quad_sum(v0 + v1);
    if (q == 0) Zb[(long)cc * MP + k] = v;
  }
}

// Barrier between the steps of a recurrence: only LDS traffic has to be complete (a __syncthreads()
// would also wait for every global store and load the wave has in flight, a memory round trip per step).
__device__ __forceinline__ void scan_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// One step of a recurrence: s = (row of the transition matrix held in a[]) . (vector in LDS), the
// lane's quarter of it, summed over the four lanes of the group.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PER>
__device__ __forceinline__ float scan_row_dot(const float (&a)[PER], const float* __restrict__ x_lds) {
  // two-wide accumulators: the products compile to packed fp32 FMAs (half the vector instructions;
  // the step is bound by them -- M * M multiply-adds on one CU)
  const f32x4* xv = reinterpret_cast<const f32x4*>(x_lds);
  f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
  for (int j4 = 0; j4 < PER / 4; ++j4) {
    const f32x4 v = xv[j4];
    s0 += f32x2{a[4 * j4], a[4 * j4 + 1]} * f32x2{v[0], v[1]};
    s1 += f32x2{a[4 * j4 + 2], a[4 * j4 + 3]} * f32x2{v[2], v[3]};
  }
  const f32x2 s = s0 + s1;
  return scan_quad_sum(s[0] + s[1]);
}

// ---------------------------------------------------------------------------------------------
// Chunk states of one utterance: X'_0 = 0, X'_{c+1} = A^L X'_c + Z_c; writes the state part of the GEMM
// operand rows, Arows[(b,c)][0 .. M) = bf16(X'_c).
// LDS: S [nc][MP] holds Z_c in row c + 1 until the recurrence replaces it by X'_{c+1} (a step reads
// row c whole and adds into its own element of row c + 1).
// ---------------------------------------------------------------------------------------------
template <int MP>
__global__ __launch_bounds__(4 * MP) void k_scan_state_fwd(const float* __restrict__ Zbuf, const float* __restrict__ ALT,
                                                           bf16_t* __restrict__ Arows, int M, int Kp, int nc) {
  constexpr int Q = 4, PER = MP / Q, THREADS = 4 * MP;
  extern __shared__ __attribute__((aligned(16))) float scan_sm[];
  float* S = scan_sm;  // [nc][MP]
  const int b = blockIdx.x, tid = threadIdx.x, k = tid >> 2, q = tid & 3;
  bf16_t* Ab = Arows + (long)b * nc * Kp;
  const bool live = k < M;
  const float* zero = reinterpret_cast<const float*>(kZeroBytes);  // masked loads: select on the address
  // row k of A^L, read from the transposed copy so that neighbouring lanes read neighbouring words
  float a[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int kk = q * PER + j;
    a[j] = *((live && kk < M) ? ALT + (long)kk * M + k : zero);
  }
  const float* Zb = Zbuf + (long)b * nc * MP;
  scan_fill_lds<THREADS>(S, nc * MP, [&](int idx) { return idx < MP ? zero : Zb + idx - MP; });  // row c + 1 <- Z_c
  __syncthreads();
  for (int c = 0; c + 1 < nc; ++c) {
    const float z = S[(c + 1) * MP + k];  // read ahead of the dot product
    const float s = scan_row_dot<PER>(a, S + c * MP + q * PER);
    if (q == 0) S[(c + 1) * MP + k] = z + s;
    scan_lds_barrier();
  }
  for (int idx = tid; idx < nc * MP; idx += THREADS) {
    const int c = idx / MP, kk = idx % MP;
    if (kk < M) Ab[(long)c * Kp + kk] = (bf16_t)S[idx];
  }
}

// ---------------------------------------------------------------------------------------------
// Costates of one utterance.  R [(b,c)][Kp] = [G_c | gu_intra | pad];
//   Lambda_nc = 0, Lambda_c = G_c + (A^L)^T Lambda_{c+1};  row (b,c) of Lbuf = Lambda_{c+1}.
// LDS: Ls [nc][MP], row c = Lambda_{c+1}; row c - 1 holds G_c until the recurrence adds to it.
// ---------------------------------------------------------------------------------------------
template <int MP>
__global__ __launch_bounds__(4 * MP) void k_scan_state_bwd(const float* __restrict__ R, const float* __restrict__ AL,
                                                           float* __restrict__ Lbuf, int M, int Kp, int nc) {
  constexpr int Q = 4, PER = MP / Q, THREADS = 4 * MP;
  extern __shared__ __attribute__((aligned(16))) float scan_sm[];
  float* Ls = scan_sm;  // [nc][MP]
  const int b = blockIdx.x, tid = threadIdx.x, k = tid >> 2, q = tid & 3;
  const bool live = k < M;
  const float* zero = reinterpret_cast<const float*>(kZeroBytes);  // masked loads: select on the address
  const float* Rb = R + (long)b * nc * Kp;
  float a[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int kk = q * PER + j;
    a[j] = *((live && kk < M) ? AL + (long)kk * M + k : zero);  // (A^L)^T [k][kk], neighbouring lanes neighbouring words
  }
  scan_fill_lds<THREADS>(Ls, nc * MP, [&](int idx) {
    const int c = idx / MP + 1, kk = idx % MP;  // row c - 1 <- G_c; row nc - 1 = Lambda_nc = 0
    return (idx >= 0 && c < nc && kk < M) ? Rb + (long)c * Kp + kk : zero;
  });
  __syncthreads();
  for (int c = nc - 1; c >= 1; --c) {
    const float g = Ls[(c - 1) * MP + k];  // read ahead of the dot product
    const float s = scan_row_dot<PER>(a, Ls + c * MP + q * PER);
    if (q == 0) Ls[(c - 1) * MP + k] = g + s;
    scan_lds_barrier();
  }
  float* Lb = Lbuf + (long)b * nc * MP;
  for (int idx = tid; idx < nc * MP; idx += THREADS) Lb[idx] = Ls[idx];
}

// k_scan_rows_bwd: gu[cL+i] = gu_intra[c][i] + sum_k HsT[k][i] Lambda_{c+1}[k]  (times u > 0 when u != NULL),
//   lane group (i, q) sums its quarter of the state components.
template <int MP>
__global__ __launch_bounds__(4 * MP) void k_scan_rows_bwd(const float* __restrict__ R, const float* __restrict__ Lbuf,
                                                          const float* __restrict__ HsT, const float* __restrict__ u,
                                                          float* __restrict__ gu, int T, int M, int Kp, int nc) {
  constexpr int L = kScanL, Q = 4, PER = MP / Q, THREADS = 4 * MP;
  __shared__ __attribute__((aligned(16))) float Ls[kScanCB * MP];
  const int b = blockIdx.x, c0 = blockIdx.y * kScanCB, tid = threadIdx.x, q = tid & 3;
  const int ccn = min(kScanCB, nc - c0);
  const float* zero = reinterpret_cast<const float*>(kZeroBytes);  // masked loads: select on the address
  const float* Lb = Lbuf + ((long)b * nc + c0) * MP;
  scan_fill_lds<THREADS>(Ls, kScanCB * MP, [&](int idx) { return (idx >= 0 && idx < ccn * MP) ? Lb + idx : zero; });
  const float* Rb = R + ((long)b * nc + c0) * Kp;
  __syncthreads();
  for (int i = tid >> 2; i < L; i += MP) {  // whole groups of four lanes stay together
    float h[PER];
#pragma unroll
    for (int jj = 0; jj < PER; ++jj) {
      const int kk = q * PER + jj;
      h[jj] = *(kk < M ? HsT + (long)kk * L + i : zero);
    }
    float gi[kScanCB];
#pragma unroll
    for (int cc = 0; cc < kScanCB; ++cc)
      gi[cc] = *((q == 0 && cc < ccn && (c0 + cc) * L + i < T) ? Rb + (long)cc * Kp + M + i : zero);
#pragma unroll 2
    for (int cc = 0; cc < kScanCB; ++cc) {
      const f32x4* lv = reinterpret_cast<const f32x4*>(Ls + cc * MP + q * PER);
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < PER / 4; j4 += 2) {
        const f32x4 va = lv[j4], vb = lv[j4 + 1];
        v0 += h[4 * j4] * va[0] + h[4 * j4 + 1] * va[1] + h[4 * j4 + 2] * va[2] + h[4 * j4 + 3] * va[3];
        v1 += h[4 * j4 + 4] * vb[0] + h[4 * j4 + 5] * vb[1] + h[4 * j4 + 6] * vb[2] + h[4 * j4 + 7] * vb[3];
      }
      const float v = scan_quad_sum(v0 + v1);
      const int t = (c0 + cc) * L + i;
      if (q == 0 && cc < ccn && t < T) {
        float gval = gi[cc] + v;
        if (u && !(u[(long)b * T + t] > 0.f)) gval = 0.f;
        gu[(long)b * T + t] = gval;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// m_rows = Arows . Wc: rows (b,c) of Kp bf16, WcT [(j,k)][Kp]; output row (b,c) is the L x M block
// of m that starts at frame cL of utterance b (the last chunk of an utterance may be short).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ScanTile::THREADS) void k_scan_gemm_fwd(const bf16_t* __restrict__ Arows,
                                                                     const bf16_t* __restrict__ WcT,
                                                                     bf16_t* __restrict__ m, int rows, int nc, int T,
                                                                     int M, int Kp, int N) {
  using TL = ScanTile;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n0 = blockIdx.x * TL::BN;
  const int m0 = blockIdx.y * TL::BM;
  typedef GldsOperand<TL::BM, TL::NW, false> OpA;
  typedef GldsOperand<TL::BN, TL::NW, false> OpB;
  OpA oa(
      smem_raw,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        const int row = m0 + r;
        if (row >= rows) return false;
        q0 = Arows + (long)row * Kp;
        return true;
      },
      Kp, Kp);
  OpB ob(
      smem_raw + 2 * OpA::STAGE_BYTES,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        const int n = n0 + r;
        if (n >= N) return false;
        q0 = WcT + (long)n * Kp;
        return true;
      },
      Kp, Kp);
  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  // the Toeplitz rows of Wc beyond the tile's last frame are zero
  const int j_last = min(N - 1, n0 + TL::BN - 1) / M;
  const int kend = min(Kp, (M + j_last + 1 + 63) & ~63);
  gemm_loop_glds<TL>(oa, ob, acc, 0, kend, w);
  bf16_t* tile = reinterpret_cast<bf16_t*>(smem_raw);
  store_tile_bf16<TL>(acc, w, tile, nullptr, 0, 0, 0, 0, 0, [](int, int, int, int, int, float v) { return v; });
  constexpr int LDT = TL::BN + 8, CH = TL::BN / 8;
  for (int idx = threadIdx.x; idx < TL::BM * CH; idx += TL::THREADS) {
    const int ml = idx / CH, c8 = idx % CH;
    const int row = m0 + ml, n = n0 + c8 * 8;
    if (row >= rows || n >= N) continue;
    const int b = row / nc, c = row % nc;
    const int j = n / M;
    if (c * kScanL + j >= T) continue;
    bf16_t* dst = m + ((long)b * T + (long)c * kScanL) * M + n;
    *reinterpret_cast<bf16x8*>(dst) = *reinterpret_cast<const bf16x8*>(&tile[ml * LDT + c8 * 8]);
  }
}

// LDS-DMA operand whose rows end at different k (the short last chunk of an utterance): as
// GldsOperand (gemm_glds_bf16.h, 64-wide k-tiles) with one limit per staging instruction.
template <int R, int NW>
struct GldsOperandRagged {
  static constexpr int ROWB = 128, RPI = 8, CPR = 8;
  static_assert(R % (RPI * NW) == 0, "every wave stages whole rows");
  static constexpr int STAGE_BYTES = R * ROWB;
  static constexpr int NI = R / (RPI * NW);
  unsigned char* base;
  const bf16_t* p0[NI];
  int kof[NI], kmaxv[NI];
  int wave_off;
  const bf16_t* zero;
  // rowfn(r, q0, kmax) -> false for rows outside the operand
  template <class RowFn>
  __device__ __forceinline__ GldsOperandRagged(unsigned char* lds, RowFn&& rowfn) : base(lds) {
    zero = reinterpret_cast<const bf16_t*>(kZeroBytes);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    wave_off = wave * (R / NW) * ROWB;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = wave * (R / NW) + RPI * i + lane / CPR;
      const bf16_t* q0 = zero;
      int km = 0;
      const bool ok = rowfn(r, q0, km);
      const int d = (lane % CPR) ^ ((r >> 1) & 7);
      p0[i] = q0 + 8 * d;
      kof[i] = 8 * d;
      kmaxv[i] = ok ? km : 0;
    }
  }
  __device__ __forceinline__ void issue(int stage, int k0) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bf16_t* p = (k0 + kof[i] < kmaxv[i]) ? p0[i] + k0 : zero;
      unsigned char* dst = base + stage * STAGE_BYTES + wave_off + i * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  }
  __device__ __forceinline__ bf16x8 frag(int stage, int row, int kl, int hh) const {
    const int d = (kl >> 3) + hh;
    const unsigned char* a = base + stage * STAGE_BYTES + row * ROWB + ((d ^ ((row >> 1) & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(a);
  }
};

// ---------------------------------------------------------------------------------------------
// slab[s][(b,c)][kk] = sum over (j,k) in the s-th range of gm_rows[(b,c)][(j,k)] Wc[kk][(j,k)].
// 1-D grid: the row tiles of one range are neighbours on one XCD (they share the range of Wc).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ScanTile::THREADS) void k_scan_gemm_adj(const bf16_t* __restrict__ gm,
                                                                     const bf16_t* __restrict__ Wc,
                                                                     float* __restrict__ slabs, int rows, int nc,
                                                                     int T, int M, int Kp, int N, int kper, int S,
                                                                     int row_tiles, int col_tiles) {
  using TL = ScanTile;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const XcdSlot slot = xcd_slot(row_tiles * col_tiles);
  if (slot.group >= S) return;
  const int s = slot.group;
  const int m0 = (slot.member % row_tiles) * TL::BM;
  const int n0 = (slot.member / row_tiles) * TL::BN;
  const int kbeg = s * kper, kend = min(N, kbeg + kper);
  typedef GldsOperandRagged<TL::BM, TL::NW> OpA;
  typedef GldsOperand<TL::BN, TL::NW, false> OpB;
  OpA oa(smem_raw, [&](int r, const bf16_t*& q0, int& km) {
    const int row = m0 + r;
    if (row >= rows) return false;
    const int b = row / nc, c = row % nc;
    q0 = gm + ((long)b * T + (long)c * kScanL) * M;
    km = min(kScanL, T - c * kScanL) * M;
    return true;
  });
  OpB ob(
      smem_raw + 2 * OpA::STAGE_BYTES,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        const int kk = n0 + r;
        if (kk >= Kp) return false;
        q0 = Wc + (long)kk * N;
        return true;
      },
      N, N);
  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  gemm_loop_glds<TL>(oa, ob, acc, kbeg, kend, w);
  float* slab = slabs + (long)s * rows * Kp;
  tile_epilogue<TL>(acc, w, [&](int mm, int nn, float v) {
    const int row = m0 + mm, kk = n0 + nn;
    if (row < rows && kk < Kp) slab[(long)row * Kp + kk] = v;
  });
}

struct ScanPlan {
  int nc, rows, row_tiles, col_tiles, S, kper;
  size_t off_arows, off_zbuf, off_slabs, off_R, total;  // zbuf: Z rows forward, costate rows backward
};
static ScanPlan scan_plan(int B, int T, int M) {
  const ScanLayout l = scan_layout(M);
  ScanPlan p;
  p.nc = scan_chunks(T);
  p.rows = B * p.nc;
  p.row_tiles = (p.rows + ScanTile::BM - 1) / ScanTile::BM;
  p.col_tiles = (l.Kp + ScanTile::BN - 1) / ScanTile::BN;
  const int ktiles = (int)((l.N + 63) / 64);
  int S = std::max(1, std::min(64, 256 / (p.row_tiles * p.col_tiles)));
  const int tiles_per = (ktiles + S - 1) / S;
  p.kper = tiles_per * 64;
  p.S = (ktiles + tiles_per - 1) / tiles_per;
  size_t off = 0;
  p.off_arows = off;
  off += scan_al256(2 * (size_t)p.rows * l.Kp);
  p.off_zbuf = off;
  off += scan_al256(4 * (size_t)p.rows * 256);
  p.off_slabs = off;
  off += scan_al256(4 * (size_t)p.S * p.rows * l.Kp);
  p.off_R = off;
  off += scan_al256(4 * (size_t)p.rows * l.Kp);
  p.total = off;
  return p;
}
size_t scan_workspace_bytes(int B, int T, int M) { return scan_supported(M) ? scan_plan(B, T, M).total : 0; }
size_t scan_forward_workspace_bytes(int B, int T, int M) { return scan_supported(M) ? scan_plan(B, T, M).off_slabs : 0; }

#define PROFS(name) ProfScope prof_scope_(name, st)
#define LAUNCH_CHECKS(name)                                                                          \
  do {                                                                                               \
    hipError_t e_ = hipGetLastError();                                                               \
    if (e_ != hipSuccess) return set_error("launch of %s failed: %s", name, hipGetErrorString(e_));  \
  } while (0)

static inline bool scan_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static int scan_check(const lmufft_shape& s, const void* scanprep, const void* ws, size_t ws_bytes, const ScanPlan& p,
                      size_t need) {
  if (!scan_supported(s.M))
    return set_error("scan convolution: memory_size %d is not supported (multiples of 8 up to 256)", s.M);
  if (!scanprep || !scan_al16(scanprep)) return set_error("scan convolution: scanprep is NULL or not 16-byte aligned");
  if (!ws || ws_bytes < need || !scan_al16(ws))
    return set_error("scan convolution: 16-byte aligned workspace of %zu bytes needed, %zu given", need, ws_bytes);
  if ((long)p.rows * scan_layout(s.M).Kp >= (1L << 31)) return set_error("scan convolution: B*T too large");
  return 0;
}

template <class F>
static int scan_dispatch_mp(int M, F&& f) {
  if (M <= 64) return f(std::integral_constant<int, 64>{});
  if (M <= 128) return f(std::integral_constant<int, 128>{});
  if (M <= 192) return f(std::integral_constant<int, 192>{});
  return f(std::integral_constant<int, 256>{});
}

int scan_conv_forward(const lmufft_shape& s, const float* u, const void* scanprep, bf16_t* m, void* ws,
                      size_t ws_bytes, hipStream_t st) {
  const ScanPlan p = scan_plan(s.B, s.T, s.M);
  if (int rc = scan_check(s, scanprep, ws, ws_bytes, p, p.off_slabs)) return rc;
  if (!scan_al16(m)) return set_error("scan convolution: m must be 16-byte aligned");
  const ScanLayout l = scan_layout(s.M);
  const char* sp = (const char*)scanprep;
  bf16_t* Arows = (bf16_t*)((char*)ws + p.off_arows);
  float* Zbuf = (float*)((char*)ws + p.off_zbuf);
  {
    const int rc = scan_dispatch_mp(s.M, [&](auto mp_c) -> int {
      constexpr int MP = decltype(mp_c)::value;
      const size_t lds = sizeof(float) * (size_t)p.nc * MP;
      if (lds > 160 * 1024) return set_error("scan convolution: sequence too long (T=%d)", s.T);
      if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)k_scan_state_fwd<MP>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return set_error("cannot reserve %zu bytes of LDS for k_scan_state_fwd", lds);
      {
        PROFS("k_scan_rows_fwd");
        hipLaunchKernelGGL(k_scan_rows_fwd<MP>, dim3(s.B, (p.nc + kScanCB - 1) / kScanCB), dim3(4 * MP), 0, st, u,
                           (const float*)(sp + l.off_Hs), Arows, Zbuf, s.T, s.M, l.Kp, p.nc);
      }
      PROFS("k_scan_state_fwd");
      hipLaunchKernelGGL(k_scan_state_fwd<MP>, dim3(s.B), dim3(4 * MP), lds, st, (const float*)Zbuf,
                         (const float*)(sp + l.off_ALT), Arows, s.M, l.Kp, p.nc);
      return 0;
    });
    if (rc) return rc;
    LAUNCH_CHECKS("k_scan_state_fwd");
  }
  {
    using TL = ScanTile;
    const dim3 grid((unsigned)((l.N + TL::BN - 1) / TL::BN), p.row_tiles);
    const size_t lds = 2 * (size_t)(TL::BM + TL::BN) * kGldsRowBytes;
    static_assert(2 * (TL::BM + TL::BN) * kGldsRowBytes >= TL::BM * (TL::BN + 8) * 2, "epilogue tile fits");
    if (hipFuncSetAttribute((const void*)k_scan_gemm_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return set_error("cannot reserve %zu bytes of LDS for k_scan_gemm_fwd", lds);
    {
      PROFS("k_scan_gemm_fwd");
      hipLaunchKernelGGL(k_scan_gemm_fwd, grid, dim3(TL::THREADS), lds, st, (const bf16_t*)Arows,
                         (const bf16_t*)(sp + l.off_WcT), m, p.rows, p.nc, s.T, s.M, l.Kp, (int)l.N);
    }
    LAUNCH_CHECKS("k_scan_gemm_fwd");
  }
  return 0;
}

// gu [B][T] = adjoint of gm, times (u > 0) when u != NULL
int scan_conv_backward(const lmufft_shape& s, const bf16_t* gm, const void* scanprep, const float* u, float* gu,
                       void* ws, size_t ws_bytes, hipStream_t st) {
  const ScanPlan p = scan_plan(s.B, s.T, s.M);
  if (int rc = scan_check(s, scanprep, ws, ws_bytes, p, p.total)) return rc;
  if (!scan_al16(gm)) return set_error("scan convolution: gm must be 16-byte aligned");
  const ScanLayout l = scan_layout(s.M);
  const char* sp = (const char*)scanprep;
  float* slabs = (float*)((char*)ws + p.off_slabs);
  float* R = (float*)((char*)ws + p.off_R);
  {
    using TL = ScanTile;
    const int inner = p.row_tiles * p.col_tiles;
    const dim3 grid(xcd_grid(p.S, inner));
    const size_t lds = 2 * (size_t)(TL::BM + TL::BN) * kGldsRowBytes;
    if (hipFuncSetAttribute((const void*)k_scan_gemm_adj, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return set_error("cannot reserve %zu bytes of LDS for k_scan_gemm_adj", lds);
    {
      PROFS("k_scan_gemm_adj");
      hipLaunchKernelGGL(k_scan_gemm_adj, grid, dim3(TL::THREADS), lds, st, gm, (const bf16_t*)(sp + l.off_Wc), slabs,
                         p.rows, p.nc, s.T, s.M, l.Kp, (int)l.N, p.kper, p.S, p.row_tiles, p.col_tiles);
    }
    LAUNCH_CHECKS("k_scan_gemm_adj");
  }
  const float* Rsum = slabs;
  if (p.S > 1) {
    if (int rc = launch_sum_slabs_masked(slabs, nullptr, R, (long)p.rows * l.Kp, p.S, st)) return rc;
    Rsum = R;
  }
  float* Lbuf = (float*)((char*)ws + p.off_zbuf);
  {
    const int rc = scan_dispatch_mp(s.M, [&](auto mp_c) -> int {
      constexpr int MP = decltype(mp_c)::value;
      const size_t lds = sizeof(float) * (size_t)p.nc * MP;
      if (lds > 160 * 1024) return set_error("scan convolution: sequence too long (T=%d)", s.T);
      if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)k_scan_state_bwd<MP>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return set_error("cannot reserve %zu bytes of LDS for k_scan_state_bwd", lds);
      {
        PROFS("k_scan_state_bwd");
        hipLaunchKernelGGL(k_scan_state_bwd<MP>, dim3(s.B), dim3(4 * MP), lds, st, Rsum,
                           (const float*)(sp + l.off_AL), Lbuf, s.M, l.Kp, p.nc);
      }
      PROFS("k_scan_rows_bwd");
      hipLaunchKernelGGL(k_scan_rows_bwd<MP>, dim3(s.B, (p.nc + kScanCB - 1) / kScanCB), dim3(4 * MP), 0, st, Rsum,
                         (const float*)Lbuf, (const float*)(sp + l.off_HsT), u, gu, s.T, s.M, l.Kp, p.nc);
      return 0;
    });
    if (rc) return rc;
    LAUNCH_CHECKS("k_scan_rows_bwd");
  }
  return 0;
}

static inline uint16_t scan_bf16_bits(double d) {
  const float f = (float)d;
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  x += 0x7fffu + ((x >> 16) & 1u);
  return (uint16_t)(x >> 16);
}

}  // namespace lmufft

using namespace lmufft;

extern "C" {

int32_t lmufft_scan_chunk(void) { return kScanL; }

size_t lmufft_scanprep_bytes(int32_t M) { return scan_supported(M) ? scan_layout(M).total : 0; }

int lmufft_scanprep_build_host(const float* A, const float* B, int32_t M, void* out) {
  if (!A || !B || !out) return set_error("scanprep: NULL pointer");
  if (!scan_supported(M)) return set_error("scanprep: memory_size %d is not supported (multiples of 8 up to 256)", M);
  const ScanLayout l = scan_layout(M);
  const int L = kScanL, Kp = l.Kp;
  const long N = l.N;
  char* o = (char*)out;
  memset(o, 0, l.total);
  uint16_t* WcT = (uint16_t*)(o + l.off_WcT);
  uint16_t* Wc = (uint16_t*)(o + l.off_Wc);
  float* AL = (float*)(o + l.off_AL);
  float* ALT = (float*)(o + l.off_ALT);
  float* Hs = (float*)(o + l.off_Hs);
  float* HsT = (float*)(o + l.off_HsT);
  std::vector<double> Ad((size_t)M * M), P((size_t)M * M, 0.0), Pn((size_t)M * M), h((size_t)(L + 1) * M);
  for (long i = 0; i < (long)M * M; ++i) Ad[i] = A[i];
  for (int k = 0; k < M; ++k) P[(size_t)k * M + k] = 1.0;
  for (int j = 0; j <= L; ++j) {  // P = A^j
    for (int k = 0; k < M; ++k) {
      double acc = 0.0;
      for (int q = 0; q < M; ++q) acc += P[(size_t)k * M + q] * (double)B[q];
      h[(size_t)j * M + k] = acc;  // H[k][j]
    }
    if (j == L) break;
    for (int k = 0; k < M; ++k)
      for (int kk = 0; kk < M; ++kk) {
        const uint16_t v = scan_bf16_bits(P[(size_t)k * M + kk]);
        WcT[((size_t)j * M + k) * Kp + kk] = v;
        Wc[(size_t)kk * N + (size_t)j * M + k] = v;
      }
    for (int k = 0; k < M; ++k) {
      double* dst = &Pn[(size_t)k * M];
      for (int kk = 0; kk < M; ++kk) dst[kk] = 0.0;
      for (int q = 0; q < M; ++q) {
        const double a = Ad[(size_t)k * M + q];
        const double* src = &P[(size_t)q * M];
        for (int kk = 0; kk < M; ++kk) dst[kk] += a * src[kk];
      }
    }
    P.swap(Pn);
  }
  for (int j = 0; j < L; ++j)
    for (int i = 0; i <= j; ++i)
      for (int k = 0; k < M; ++k) {
        const uint16_t v = scan_bf16_bits(h[(size_t)(j - i) * M + k]);
        WcT[((size_t)j * M + k) * Kp + M + i] = v;
        Wc[(size_t)(M + i) * N + (size_t)j * M + k] = v;
      }
  for (int k = 0; k < M; ++k)
    for (int kk = 0; kk < M; ++kk) {
      AL[(size_t)k * M + kk] = (float)P[(size_t)k * M + kk];
      ALT[(size_t)k * M + kk] = (float)P[(size_t)kk * M + k];
    }
  for (int i = 0; i < L; ++i)
    for (int k = 0; k < M; ++k) {
      const float v = (float)h[(size_t)(L - i) * M + k];
      Hs[(size_t)i * M + k] = v;
      HsT[(size_t)k * L + i] = v;
    }
  return 0;
}

size_t lmufft_conv_scan_workspace_bytes(const lmufft_shape* s) {
  if (!s || s->B <= 0 || s->T <= 0) return 0;
  return scan_workspace_bytes(s->B, s->T, s->M);
}

int lmufft_conv_forward_scan_bf16(const lmufft_shape* s, const float* u, const void* scanprep, void* m,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  if (!s || s->B <= 0 || s->T <= 0 || s->M <= 0) return set_error("conv: bad shape");
  if (!u || !m) return set_error("conv: NULL pointer");
  return scan_conv_forward(*s, u, scanprep, (bf16_t*)m, workspace, workspace_bytes, (hipStream_t)stream);
}

int lmufft_conv_backward_scan_bf16(const lmufft_shape* s, const void* gm, const void* scanprep, float* gu,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!s || s->B <= 0 || s->T <= 0 || s->M <= 0) return set_error("conv: bad shape");
  if (!gm || !gu) return set_error("conv: NULL pointer");
  return scan_conv_backward(*s, (const bf16_t*)gm, scanprep, nullptr, gu, workspace, workspace_bytes,
                            (hipStream_t)stream);
}

}  // extern "C"

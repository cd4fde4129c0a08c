// Linear layers of the encoder for bf16 inference on gfx950, with the residual add + layer norm
// that follows the attention block's output projection folded into the GEMM's epilogue.
//
// Reference arithmetic: torch.nn.Linear under bf16 autocast -- bf16 operands, fp32 accumulation,
//   bias added in fp32, one rounding to bf16 (src/models/lmu_encoder.py:40-43 w_q/w_k/w_v/w_o,
//   :184 input_projection, :229 output_projection; calls at :68-70, :94, :244, :278), and
//   `self.layer_norm(output + residual)` (src/models/lmu_encoder.py:97): the bf16 projection plus
//   the residual in fp32, torch.nn.LayerNorm statistics in fp32, biased variance.
//
// Two kernels: k_linear_ws_bf16 (weight-stationary, K <= 384: every Linear of the encoder; described at
//   its definition below) and the tiled k_linear_glds_bf16 for longer rows.
//
// k_linear_glds_bf16: out[rows][N] = x[rows][K] W[N][K]^T + bias.  Both operands are plain
//   k-contiguous rows, so they are staged by LDS-DMA (gemm_glds_bf16.h) with no transposition.
//   Tile 128 rows x 384 columns on eight waves: for N <= 384 every row of x passes through a CU's
//   load path once and W (L2-resident, N*K*2 bytes) once per 128 rows; wider outputs (the fused
//   q|k|v projection, N = 3 d_model) take ceil(N/384) column tiles whose workgroups are adjacent
//   in launch order, so the x rows they share are served by L2.
//   Layer-norm epilogue (N <= 384, one column tile): four lanes share a tile row, read the
//   bf16-rounded projection back from the LDS tile, add the residual, and keep the 96 sums of
//   their share of the row in registers across the mean / variance / scale passes -- the
//   projection's output and the sum never go to HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "gemm_glds_bf16.h"
#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

using LinearTile = TileB<128, 384, 64, 2, 4>;

struct LinearLn {  // y == NULL: none
  const void* res;  // [rows][N], may be NULL
  int res_bf16;
  const float *gamma, *beta;
  float eps;
  void* y;
  int y_bf16;
};

__global__ __launch_bounds__(LinearTile::THREADS) void k_linear_glds_bf16(const bf16_t* __restrict__ x,
                                                                          const bf16_t* __restrict__ W,
                                                                          const float* __restrict__ bias,
                                                                          bf16_t* __restrict__ out, long rows, int K,
                                                                          int N, LinearLn ln) {
  using TL = LinearTile;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n0 = blockIdx.x * TL::BN;
  const long m0 = (long)blockIdx.y * TL::BM;

  typedef GldsOperand<TL::BM, TL::NW, false> OpA;
  typedef GldsOperand<TL::BN, TL::NW, false> OpB;
  OpA oa(
      smem_raw,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        const long row = m0 + r;
        if (row >= rows) return false;
        q0 = x + row * K;
        return true;
      },
      K, K);
  OpB ob(
      smem_raw + 2 * OpA::STAGE_BYTES,
      [&](int r, const bf16_t*& q0, const bf16_t*&) {
        const int n = n0 + r;
        if (n >= N) return false;
        q0 = W + (long)n * K;
        return true;
      },
      K, K);

  WavePos<TL> w;
  f32x16 acc[TL::TM][TL::TN];
  tileb_zero<TL>(acc);
  gemm_loop_glds<TL>(oa, ob, acc, 0, K, w);

  float bcol[TL::TN];
  preload_cols<TL>(bcol, w, [&](int nn) { return load_or_zero(bias + n0 + nn, bias && n0 + nn < N); });
  bf16_t* tile = reinterpret_cast<bf16_t*>(smem_raw);
  store_tile_bf16<TL>(acc, w, tile, out, N, m0, n0, rows, N,
                      [&](int, int tn, int, int, int, float v) { return v + bcol[tn]; });

  if (ln.y) {  // the tile spans the whole row (the host checks N <= BN)
    static_assert(TL::THREADS == 4 * TL::BM, "four lanes per tile row");
    constexpr int LDT = TL::BN + 8;
    constexpr int PER = TL::BN / 8 / 4;  // 16-byte chunks per lane
    const int rl = threadIdx.x >> 2, q = threadIdx.x & 3;
    const long row = m0 + rl;
    const bool live = row < rows;
    const int chunks = N / 8;
    float v[PER][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = q + 4 * i;
      const bool on = c < chunks;
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(&tile[rl * LDT + c * 8]);
      float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (ln.res && on && live) {
        if (ln.res_bf16) {
          const bf16x8 rb = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(ln.res) + row * N + c * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] = (float)rb[j];
        } else {
          const float* rp = reinterpret_cast<const float*>(ln.res) + row * N + c * 8;
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp);
          const f32x4 r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            r[j] = r0[j];
            r[j + 4] = r1[j];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[i][j] = on ? (float)t[j] + r[j] : 0.f;
        s += v[i][j];
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    const float mean = s / (float)N;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const bool on = q + 4 * i < chunks;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = on ? v[i][j] - mean : 0.f;
        ss += d * d;
      }
    }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    const float rstd = rsqrtf(ss / (float)N + ln.eps);
    if (live) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int c = q + 4 * i;
        if (c < chunks) {
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(ln.gamma + c * 8);
          const f32x4 g1 = *reinterpret_cast<const f32x4*>(ln.gamma + c * 8 + 4);
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln.beta + c * 8);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(ln.beta + c * 8 + 4);
          float o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            o[j] = (v[i][j] - mean) * rstd * (j < 4 ? g0[j] : g1[j - 4]) + (j < 4 ? b0[j] : b1[j - 4]);
          if (ln.y_bf16) {
            bf16x8 ob8;
#pragma unroll
            for (int j = 0; j < 8; ++j) ob8[j] = (bf16_t)o[j];
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(ln.y) + row * N + c * 8) = ob8;
          } else {
            float* yp = reinterpret_cast<float*>(ln.y) + row * N + c * 8;
            *reinterpret_cast<f32x4*>(yp) = f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(yp + 4) = f32x4{o[4], o[5], o[6], o[7]};
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------
// Weight-stationary form for K <= 384.
//
// The tiled kernel above pulls, per 128 rows, 98 KB of x AND 295 KB of W through the CU's L1 --
// W, the operand that never changes, is three quarters of the fill traffic, and the fill rate of
// a CU's L1 is what bounds these GEMMs.  Here a persistent workgroup keeps its 384 columns of W
// in REGISTERS for its whole life -- eight waves x 48 columns x K: 144 registers per lane as A
// fragments of v_mfma_f32_16x16x32_bf16 -- and only x streams: 64-row tiles (48 KB) by LDS-DMA
// into two stages, the next tile in flight while this one is multiplied, one barrier pair per
// tile instead of one per k-tile.  The product is taken transposed, D = W x^T, so a lane ends up
// with four consecutive output columns of one row; the tile goes through an LDS image
// [64][384 + 8] to leave as 16-byte row segments (or through the layer-norm epilogue, sixteen
// lanes per row).
//
// LDS image of an x tile: [64 rows][48 chunks of 16 B], chunk c of row r at position
// (c & ~15) | ((c ^ r) & 15): the 16 rows a ds_read_b128 lane group reads fall on 16 distinct
// 16-byte slots of the 256-byte bank row.  The LDS-DMA destination is linear in the lane index,
// so the permutation is applied to the source address.
// ---------------------------------------------------------------------------
typedef float f32x4a __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kWsThreads = 512;
constexpr int kWsRows = 64;                      // x rows per tile
constexpr int kWsCols = 384;                     // output columns per workgroup: 8 waves x 48
constexpr int kWsKS = 12;                        // k-steps of 32: K <= 384
constexpr int kWsCPR = kWsKS * 4;                // 16-byte chunks per LDS row
constexpr int kWsPitch = kWsCPR * 16;            // bytes per LDS row
constexpr int kWsStage = kWsRows * kWsPitch;     // 48 KB
constexpr int kWsNI = kWsStage / 1024 / 8;       // LDS-DMA instructions per wave and tile
constexpr int kWsOutLd = kWsCols + 8;            // bf16 elements per row of the output image
constexpr int kWsLds = 2 * kWsStage + kWsRows * kWsOutLd * 2 + kWsCols * 4;  // x stages | output image | bias
constexpr int kWsAhead = 2;                      // k-steps between an x fragment's LDS read and its MFMAs
static_assert(kWsLds <= 160 * 1024, "fits the CU's LDS");

template <bool LN, int KSTEPS>
__global__ __launch_bounds__(kWsThreads) void k_linear_ws_bf16(const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ W,
                                                               const float* __restrict__ bias,
                                                               bf16_t* __restrict__ out, long rows, int K, int N,
                                                               LinearLn ln) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(kZeroBytes);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
  // blockIdx.y = column block, blockIdx.x = persistent slot: with a slot count that is a multiple
  // of eight, the workgroups that walk the same x tiles for different column blocks sit on the
  // same XCD (workgroup id % 8) and share those tiles through its L2
  const int n_blk = blockIdx.y * kWsCols;
  const int nw = n_blk + wave * 48;

  // this wave's 48 columns of W as A fragments: lane holds W[nw + 16 t + l15][32 ks + 8 q .. + 8]
  static_assert(KSTEPS <= kWsKS, "the LDS image holds 32 kWsKS elements per row");
  bf16x8 wf[3][KSTEPS];  // K <= 32 KSTEPS (the host picks the instantiation)
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int n = nw + t * 16 + l15, k = ks * 32 + 8 * q;
      wf[t][ks] = *reinterpret_cast<const bf16x8*>((n < N && k < K) ? W + (long)n * K + k : zero);
    }

  // LDS-DMA sources: slot p of the tile image holds chunk c of row r (recomputed per tile: the
  // registers are needed for W)
  auto issue = [&](int stage, long row0) {
#pragma unroll
    for (int i = 0; i < kWsNI; ++i) {
      const int p = (wave * kWsNI + i) * 64 + lane;
      const int r = p / kWsCPR, cp = p % kWsCPR;
      const int c = (cp & ~15) | ((cp ^ r) & 15);
      const bool ok = c * 8 < K && row0 + r < rows;
      const bf16_t* src = ok ? x + (row0 + r) * K + c * 8 : zero;
      unsigned char* dst = smem_raw + stage * kWsStage + (wave * kWsNI + i) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  int posl[4];  // position of chunk 4 (ks & 3) + q in this lane's rows, before the 16 (ks >> 2) part
#pragma unroll
  for (int j = 0; j < 4; ++j) posl[j] = ((4 * j + q) ^ l15) * 16;
  bf16_t* otile = reinterpret_cast<bf16_t*>(smem_raw + 2 * kWsStage);
  // the bias sits in LDS: a global load inside the round would make its s_waitcnt vmcnt wait for
  // the next tile's LDS-DMA as well
  float* bias_s = reinterpret_cast<float*>(smem_raw + 2 * kWsStage + kWsRows * kWsOutLd * 2);
  if (threadIdx.x < kWsCols)
    bias_s[threadIdx.x] = (bias && n_blk + (int)threadIdx.x < N) ? bias[n_blk + threadIdx.x] : 0.f;

  const long ntiles = (rows + kWsRows - 1) / kWsRows;
  long tile = blockIdx.x;
  int stage = 0;
  if (tile < ntiles) issue(0, tile * kWsRows);
  for (; tile < ntiles; tile += gridDim.x, stage ^= 1) {
    // this wave's LDS-DMA writes have landed; after the barrier everybody's have, and everybody has
    // finished with the other stage and with the output image of the previous tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tile + gridDim.x < ntiles) issue(stage ^ 1, (tile + gridDim.x) * kWsRows);

    const unsigned char* xs = smem_raw + stage * kWsStage + l15 * kWsPitch;
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // 32 rows at a time: 24 accumulator registers
      f32x4a acc[2][3];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[g][t] = f32x4a{0.f, 0.f, 0.f, 0.f};
      // x fragments are read kWsAhead k-steps before the MFMAs that use them; the scheduling barriers
      // keep that distance (left alone, the compiler sinks every read to its use and each k-step
      // waits out the LDS latency)
      bf16x8 xq[kWsAhead + 1][2];
      auto read_x = [&](int ks, bf16x8 (&d)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
          d[g] = *reinterpret_cast<const bf16x8*>(xs + (half * 2 + g) * 16 * kWsPitch + posl[ks & 3] + 256 * (ks >> 2));
      };
#pragma unroll
      for (int ks = 0; ks < kWsAhead && ks < KSTEPS; ++ks) read_x(ks, xq[ks % (kWsAhead + 1)]);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        if (ks + kWsAhead < KSTEPS) read_x(ks + kWsAhead, xq[(ks + kWsAhead) % (kWsAhead + 1)]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int t = 0; t < 3; ++t)
            acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][ks], xq[ks % (kWsAhead + 1)][g], acc[g][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // D[n = 4 q + v][row = l15]: four consecutive columns of one row per lane
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bias_s[wave * 48 + t * 16 + 4 * q]);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          bf16x4 o;
#pragma unroll
          for (int v = 0; v < 4; ++v) o[v] = (bf16_t)(acc[g][t][v] + b4[v]);
          *reinterpret_cast<bf16x4*>(
              &otile[((half * 2 + g) * 16 + l15) * kWsOutLd + wave * 48 + t * 16 + 4 * q]) = o;
        }
      }
    }
    __syncthreads();

    const long row0 = tile * kWsRows;
    if (out) {
#pragma unroll
      for (int i = 0; i < kWsRows * (kWsCols / 8) / kWsThreads; ++i) {
        const int idx = threadIdx.x + i * kWsThreads;
        const int m = idx / (kWsCols / 8), c = idx % (kWsCols / 8);
        const long row = row0 + m;
        const int col = n_blk + c * 8;
        if (row < rows && col < N)
          *reinterpret_cast<bf16x8*>(out + row * N + col) = *reinterpret_cast<const bf16x8*>(&otile[m * kWsOutLd + c * 8]);
      }
    }
    if (LN) {  // one column block holds whole rows (the host checks N <= 384)
      // sixteen lanes per row, 32 rows per pass: a lane keeps 24 sums, which leaves W in its registers
      static_assert(kWsThreads == 16 * (kWsRows / 2), "sixteen lanes per row, two passes");
      constexpr int PER = kWsCols / 8 / 16;
      const int sub = threadIdx.x & 15;
      const int chunks = N / 8;
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        const int rl = pass * (kWsRows / 2) + (threadIdx.x >> 4);
        const long row = row0 + rl;
        const bool live = row < rows;
        float v[PER][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int c = sub + 16 * i;
          const bool on = c < chunks;
          const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&otile[rl * kWsOutLd + c * 8]);
          float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (ln.res && on && live) {
            if (ln.res_bf16) {
              const bf16x8 rb =
                  *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(ln.res) + row * N + c * 8);
#pragma unroll
              for (int j = 0; j < 8; ++j) r[j] = (float)rb[j];
            } else {
              const float* rp = reinterpret_cast<const float*>(ln.res) + row * N + c * 8;
              const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp);
              const f32x4 r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                r[j] = r0[j];
                r[j + 4] = r1[j];
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[i][j] = on ? (float)t8[j] + r[j] : 0.f;
            s += v[i][j];
          }
        }
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) s += __shfl_xor(s, sh);
        const float mean = s / (float)N;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const bool on = sub + 16 * i < chunks;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = on ? v[i][j] - mean : 0.f;
            ss += d * d;
          }
        }
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) ss += __shfl_xor(ss, sh);
        const float rstd = rsqrtf(ss / (float)N + ln.eps);
        if (live) {
#pragma unroll
          for (int i = 0; i < PER; ++i) {
            const int c = sub + 16 * i;
            if (c < chunks) {
              const f32x4 g0 = *reinterpret_cast<const f32x4*>(ln.gamma + c * 8);
              const f32x4 g1 = *reinterpret_cast<const f32x4*>(ln.gamma + c * 8 + 4);
              const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln.beta + c * 8);
              const f32x4 b1 = *reinterpret_cast<const f32x4*>(ln.beta + c * 8 + 4);
              float o[8];
#pragma unroll
              for (int j = 0; j < 8; ++j)
                o[j] = (v[i][j] - mean) * rstd * (j < 4 ? g0[j] : g1[j - 4]) + (j < 4 ? b0[j] : b1[j - 4]);
              if (ln.y_bf16) {
                bf16x8 ob8;
#pragma unroll
                for (int j = 0; j < 8; ++j) ob8[j] = (bf16_t)o[j];
                *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(ln.y) + row * N + c * 8) = ob8;
              } else {
                float* yp = reinterpret_cast<float*>(ln.y) + row * N + c * 8;
                *reinterpret_cast<f32x4*>(yp) = f32x4{o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(yp + 4) = f32x4{o[4], o[5], o[6], o[7]};
              }
            }
          }
        }
      }
    }
  }
}

}  // namespace lmufft

using namespace lmufft;

namespace {
constexpr int kComputeUnits = 256;  // gfx950: persistent workgroups, one per CU
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int linear_impl(const void* x, const void* W, const float* bias, int64_t rows, int32_t K, int32_t N, void* out,
                const LinearLn& ln, void* stream) {
  using TL = LinearTile;
  if (rows < 0 || K <= 0 || N <= 0) return set_error("linear: bad shape (rows=%lld K=%d N=%d)", (long long)rows, K, N);
  if (K % 8 != 0 || N % 8 != 0) return set_error("linear: K=%d and N=%d must be multiples of 8", K, N);
  if (rows == 0) return 0;
  if (!x || !W || (!out && !ln.y)) return set_error("linear: NULL pointer");
  if (!al16(x) || !al16(W) || !al16(out) || (bias && (reinterpret_cast<uintptr_t>(bias) & 3)))
    return set_error("linear: x, W and out must be 16-byte aligned");
  if (ln.y) {
    if (N > TL::BN) return set_error("linear: the layer-norm epilogue needs N <= %d (got %d)", TL::BN, N);
    if (!ln.gamma || !ln.beta) return set_error("linear: NULL layer-norm parameters");
    if (!al16(ln.y) || !al16(ln.res) || !al16(ln.gamma) || !al16(ln.beta))
      return set_error("linear: y, res, gamma and beta must be 16-byte aligned");
  }
  hipStream_t st = (hipStream_t)stream;
  // One round of 128-row tiles on a single column tile (the training step's 384 -> 384 layers at batch 64 x 400
  // frames: 200 workgroups) is faster on the tiled kernel, 17.6 against 22.7 us: the persistent kernel's prologue
  // (295 KB of W into registers per workgroup) is paid for one or two 64-row tiles only.  More rows, the wide q|k|v
  // projection (38 against 47 us) and narrow outputs stay weight-stationary (tools/linear_shapes_bench.py).
  const bool one_round_tiled = K > 96 && N > 128 && N <= TL::BN && rows <= (int64_t)TL::BM * kComputeUnits;
  if (K <= kWsKS * 32 && !one_round_tiled) {  // weight-stationary persistent kernel
    const int col_blocks = (N + kWsCols - 1) / kWsCols;
    const long ntiles = (rows + kWsRows - 1) / kWsRows;
    long per_block = std::max(1L, (long)kComputeUnits / col_blocks);
    if (col_blocks > 1 && per_block >= 8) per_block &= ~7L;  // same-XCD placement of the column blocks of a tile
    const dim3 grid((unsigned)std::min(ntiles, per_block), col_blocks);
    const bool narrow = K <= 96;  // three k-steps: the input projection (80 mel bins)
    auto kernel = ln.y ? (narrow ? k_linear_ws_bf16<true, 3> : k_linear_ws_bf16<true, kWsKS>)
                       : (narrow ? k_linear_ws_bf16<false, 3> : k_linear_ws_bf16<false, kWsKS>);
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kWsLds) != hipSuccess)
      return set_error("cannot reserve %d bytes of LDS for k_linear_ws_bf16", kWsLds);
    {
      ProfScope ps("k_linear_ws_bf16", st);
      hipLaunchKernelGGL(kernel, grid, dim3(kWsThreads), kWsLds, st, (const bf16_t*)x, (const bf16_t*)W, bias,
                         (bf16_t*)out, (long)rows, K, N, ln);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error("k_linear_ws_bf16: %s", hipGetErrorString(e));
    return 0;
  }
  const long row_tiles = (rows + TL::BM - 1) / TL::BM;
  if (row_tiles > 65535) return set_error("linear: %lld rows exceed the launch grid", (long long)rows);
  const dim3 grid((N + TL::BN - 1) / TL::BN, (unsigned)row_tiles);
  const size_t lds = 2 * (size_t)(TL::BM + TL::BN) * kGldsRowBytes;
  static_assert(2 * (TL::BM + TL::BN) * kGldsRowBytes >= TL::BM * (TL::BN + 8) * 2, "epilogue tile fits");
  if (hipFuncSetAttribute((const void*)k_linear_glds_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
      hipSuccess)
    return set_error("cannot reserve %zu bytes of LDS for k_linear_glds_bf16", lds);
  {
    ProfScope ps("k_linear_glds_bf16", st);
    hipLaunchKernelGGL(k_linear_glds_bf16, grid, dim3(TL::THREADS), lds, st, (const bf16_t*)x, (const bf16_t*)W, bias,
                       (bf16_t*)out, (long)rows, K, N, ln);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("k_linear_glds_bf16: %s", hipGetErrorString(e));
  return 0;
}
}  // namespace

// ---------------------------------------------------------------------------
// Operands of a Linear layer's training step from its fp32 master weight in one launch: W in bf16 (forward), W^T
// in bf16 (the input gradient is the forward kernel on W^T) and the bias as autocast sees it (rounded to bf16,
// widened again).  A thread writes eight consecutive bf16: of one row for W, of one column j for W^T with
// consecutive threads on consecutive j, so that the strided reads of W coalesce across lanes.
// ---------------------------------------------------------------------------
namespace lmufft {
// blocks `block` of `nblocks` of one weight's conversion
__device__ __forceinline__ void wprep_body(const float* __restrict__ W, const float* __restrict__ bias, int N, int K,
                                           int Np, bf16_t* __restrict__ Wb, bf16_t* __restrict__ WbT, long ldT,
                                           float* __restrict__ bias_r, int block, int nblocks) {
  const int g1 = K / 8, g2 = Np / 8;
  const long n1 = (long)Np * g1, n2 = (long)K * g2, n3 = bias_r ? Np : 0;
  for (long i = block * 256L + threadIdx.x; i < n1 + n2 + n3; i += (long)nblocks * 256) {
    if (i < n1) {
      const int n = (int)(i / g1), k0 = (int)(i % g1) * 8;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)load_or_zero(W + (long)n * K + k0 + e, n < N);  // all eight in flight
      *reinterpret_cast<bf16x8*>(Wb + (long)n * K + k0) = o;
    } else if (i < n1 + n2) {
      const long r = i - n1;
      const int j = (int)(r % K), n0 = (int)(r / K) * 8;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)load_or_zero(W + (long)(n0 + e) * K + j, n0 + e < N);
      *reinterpret_cast<bf16x8*>(WbT + (long)j * ldT + n0) = o;
    } else {
      const int n = (int)(i - n1 - n2);
      bias_r[n] = (bias && n < N) ? (float)(bf16_t)bias[n] : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void k_linear_wprep(const float* __restrict__ W, const float* __restrict__ bias, int N,
                                                      int K, int Np, bf16_t* __restrict__ Wb, bf16_t* __restrict__ WbT,
                                                      long ldT, float* __restrict__ bias_r) {
  wprep_body(W, bias, N, K, Np, Wb, WbT, ldT, bias_r, (int)blockIdx.x, (int)gridDim.x);
}

// Several weights in one launch (a training step converts every Linear layer's weight: one launch instead of one per
// weight).  The table travels as a kernel argument; workgroup b belongs to the item whose block range holds it.
constexpr int kWprepMaxItems = 32;
struct WprepBatch {
  const float* W[kWprepMaxItems];
  const float* bias[kWprepMaxItems];
  bf16_t* Wb[kWprepMaxItems];
  bf16_t* WbT[kWprepMaxItems];
  float* bias_r[kWprepMaxItems];
  long ldT[kWprepMaxItems];
  int N[kWprepMaxItems], K[kWprepMaxItems], Np[kWprepMaxItems];
  int first_block[kWprepMaxItems + 1];
  int count;
};
__global__ __launch_bounds__(256) void k_linear_wprep_many(WprepBatch) {
  // The table is read where it lies, in the kernel-argument segment, through scalar loads at a run-time index.
  // (Indexing the by-value parameter itself at run time makes every thread keep a private copy of the 2 KB table,
  // and picking the item with compile-time indices and selects spills scalar registers: 68 and 43 us for the step's
  // fifteen weights.)
  typedef const __attribute__((address_space(4))) WprepBatch* table_ptr;
  table_ptr b = (table_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  int it = 0;
  for (int i = 1; i < b->count; ++i) it += (int)blockIdx.x >= b->first_block[i] ? 1 : 0;  // uniform
  const int first = b->first_block[it];
  wprep_body(b->W[it], b->bias[it], b->N[it], b->K[it], b->Np[it], b->Wb[it], b->WbT[it], b->ldT[it], b->bias_r[it],
             (int)blockIdx.x - first, b->first_block[it + 1] - first);
}
}  // namespace lmufft

static int wprep_check(const float* W, int32_t N, int32_t K, int32_t N_pad, const void* Wb, const void* WbT, int64_t ldT) {
  if (N <= 0 || K <= 0 || N_pad < N) return set_error("linear wprep: bad shape (N=%d K=%d N_pad=%d)", N, K, N_pad);
  if (K % 8 || N_pad % 8 || ldT % 8 || ldT < N_pad)
    return set_error("linear wprep: K, N_pad and ldT must be multiples of 8, ldT >= N_pad (got %d, %d, %lld)", K, N_pad, (long long)ldT);
  if (!W || !Wb || !WbT) return set_error("linear wprep: NULL pointer");
  if ((((uintptr_t)Wb | (uintptr_t)WbT) & 15) != 0) return set_error("linear wprep: outputs must be 16-byte aligned");
  return 0;
}
static int wprep_blocks(int32_t K, int32_t N_pad, bool with_bias, int cap) {
  const long items = (long)N_pad * (K / 8) + (long)K * (N_pad / 8) + (with_bias ? N_pad : 0);
  return (int)std::min<long>((items + 255) / 256, cap);
}

extern "C" int lmufft_linear_wprep_bf16(const float* W, const float* bias, int32_t N, int32_t K, int32_t N_pad, void* Wb,
                                        void* WbT, int64_t ldT, float* bias_rounded, void* stream) {
  if (int rc = wprep_check(W, N, K, N_pad, Wb, WbT, ldT)) return rc;
  const int blocks = wprep_blocks(K, N_pad, bias_rounded != nullptr, 2048);
  hipStream_t st = (hipStream_t)stream;
  {
    ProfScope ps("k_linear_wprep", st);
    hipLaunchKernelGGL(k_linear_wprep, dim3(blocks), dim3(256), 0, st, W, bias, N, K, N_pad, (bf16_t*)Wb, (bf16_t*)WbT,
                       (long)ldT, bias_rounded);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_linear_wprep failed: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int lmufft_linear_wprep_many_bf16(const lmufft_wprep_item* items, int32_t count, void* stream) {
  if (count < 0 || (count > 0 && !items)) return set_error("linear wprep: bad item list");
  for (int i = 0; i < count; ++i)
    if (int rc = wprep_check(items[i].W, items[i].N, items[i].K, items[i].N_pad, items[i].Wb, items[i].WbT, items[i].ldT))
      return rc;
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < count; i0 += kWprepMaxItems) {
    WprepBatch b;
    b.count = std::min(count - i0, kWprepMaxItems);
    b.first_block[0] = 0;
    for (int i = 0; i < b.count; ++i) {
      const lmufft_wprep_item& it = items[i0 + i];
      b.W[i] = it.W;
      b.bias[i] = it.bias;
      b.Wb[i] = (bf16_t*)it.Wb;
      b.WbT[i] = (bf16_t*)it.WbT;
      b.bias_r[i] = it.bias_rounded;
      b.ldT[i] = (long)it.ldT;
      b.N[i] = it.N;
      b.K[i] = it.K;
      b.Np[i] = it.N_pad;
      // 256 blocks per weight at most: a step's fourteen weights together fill the chip several times over
      b.first_block[i + 1] = b.first_block[i] + wprep_blocks(it.K, it.N_pad, it.bias_rounded != nullptr, 256);
    }
    {
      ProfScope ps("k_linear_wprep", st);
      hipLaunchKernelGGL(k_linear_wprep_many, dim3(b.first_block[b.count]), dim3(256), 0, st, b);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error("launch of k_linear_wprep_many failed: %s", hipGetErrorString(e));
  }
  return 0;
}

extern "C" int lmufft_linear_bf16(const void* x, const void* W, const float* bias, int64_t rows, int32_t K, int32_t N,
                                  void* out, void* stream) {
  if (rows > 0 && !out) return set_error("linear: NULL pointer");
  const LinearLn none{nullptr, 0, nullptr, nullptr, 0.f, nullptr, 0};
  return linear_impl(x, W, bias, rows, K, N, out, none, stream);
}

extern "C" int lmufft_linear_ln_bf16(const void* x, const void* W, const float* bias, int64_t rows, int32_t K,
                                     int32_t N, void* out, const void* res, int res_dtype, const float* gamma,
                                     const float* beta, float eps, void* y, int y_dtype, void* stream) {
  auto known = [](int d) { return d == LMUFFT_F32 || d == LMUFFT_BF16; };
  if (!known(y_dtype) || (res && !known(res_dtype))) return set_error("linear: unknown dtype");
  if (rows > 0 && !y) return set_error("linear: NULL pointer");
  if (y && (y == x || (res && y == res && y_dtype != res_dtype)))
    return set_error("linear: y may alias res only with the same dtype, and never x");
  const LinearLn ln{res, res_dtype == LMUFFT_BF16, gamma, beta, eps, y, y_dtype == LMUFFT_BF16};
  return linear_impl(x, W, bias, rows, K, N, out, ln, stream);
}

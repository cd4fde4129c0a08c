// CTC head for inference on gfx950: vocabulary projection + log-softmax + greedy decode,
// without the per-frame host round trips of the reference.
//
// Reference arithmetic: CTCDecoder.forward, src/models/asr_model.py:34-50
//   (log_probs = log_softmax(enc W^T + b)), and LMUASRModel.decode / _greedy_decode,
//   src/models/asr_model.py:158-213 (per utterance: argmax over the vocabulary for the first
//   input_lengths[i] frames, drop blanks (= vocab_size - 1) and consecutive repeats) -- there a
//   Python loop with one .item() per frame.
//
// k_ctc_logits: one wave per frame; lane v owns vocabulary entry v (V <= 64).  W^T sits in LDS as
//   [k][V + 1] (lanes hit distinct banks), the frame's encoder row in a wave-private LDS strip
//   (broadcast reads).  Max, sum-exp and the arg-max (first maximum, like torch.argmax on the
//   CPU) are xor-tree reductions over the wave.  HBM: reads enc once, writes log-probs once.
// k_ctc_collapse: one workgroup per utterance; keep-flags, block prefix sum, compaction.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>

#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

typedef __bf16 ctc_bf16;
constexpr int kCtcThreads = 256;
constexpr int kCtcMaxHd = 1024;

template <class TIn>
__global__ __launch_bounds__(kCtcThreads) void k_ctc_logits(const TIn* __restrict__ enc, const float* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ log_probs, int* __restrict__ best,
                                                            long frames, int Hd, int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ldw = V + 1;
  float* Wl = lds;                                  // [Hd][V + 1]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* er = lds + (size_t)Hd * ldw + (size_t)wave * Hd;  // this wave's encoder row
  for (int i = threadIdx.x; i < V * Hd; i += kCtcThreads) {
    const int v = i / Hd, k = i % Hd;
    Wl[k * ldw + v] = W[i];
  }
  __syncthreads();
  const float bv = lane < V ? bias[lane] : 0.f;
  for (long f = (long)blockIdx.x * 4 + wave; f < frames; f += (long)gridDim.x * 4) {
    for (int k = lane; k < Hd; k += 64) er[k] = (float)enc[f * Hd + k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private strip: program order suffices
    float acc = bv;
    if (lane < V) {
#pragma unroll 8
      for (int k = 0; k < Hd; ++k) acc = fmaf(er[k], Wl[k * ldw + lane], acc);
    }
    // log-softmax over the V lanes
    float mx = lane < V ? acc : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float ex = lane < V ? expf(acc - mx) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ex += __shfl_xor(ex, o);
    const float lp = acc - mx - logf(ex);
    // arg-max of the log-probabilities, first maximum on ties
    float bvv = lane < V ? lp : -INFINITY;
    int bi = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bvv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bvv || (ov == bvv && oi < bi)) {
        bvv = ov;
        bi = oi;
      }
    }
    if (log_probs && lane < V) log_probs[f * V + lane] = lp;
    if (lane == 0) best[f] = bi;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the strip is rewritten in the next round
  }
}

// ---------------------------------------------------------------------------
// The same arithmetic on the matrix pipe (fp32 products: v_mfma_f32_32x32x2f32), for encoder
// widths that are a multiple of 32 and fit the LDS together with W.
//
// A wave owns 32 frames and computes the TRANSPOSED logits tile D[v][frame] = W enc^T: the
// accumulator layout then gives every lane one frame (column lane & 31) and 16 vocabulary
// entries per 32-row tile (rows (i & 3) + 8 (i >> 2) + 4 (lane >> 5)), so max / sum-exp /
// arg-max run over the lane's own registers plus ONE exchange with lane ^ 32 -- no wave-wide
// reduction trees.  W sits in LDS once per workgroup as [32 VT][Hd + 4]; the wave's frames pass
// through a wave-private LDS strip of 32 x 32 floats (coalesced 16-byte global loads in, one
// ds_read_b128 per lane and 4 k-steps out; a wave's LDS operations complete in order, so the
// strip needs no workgroup barrier).  In one MFMA the two lane halves supply two k values; the
// halves take k = kb + 4 hh + j in step j, the same assignment for both operands, which is all
// the sum over k needs.  Log-probabilities leave through the strip as contiguous rows.
// ---------------------------------------------------------------------------
constexpr int kCtcMfmaWaves = 8;
constexpr int kCtcStripLd = 36;  // floats per strip row: 32 + 4, conflict-free ds_read_b128 over 16 rows

typedef float ctc_f32x4 __attribute__((ext_vector_type(4)));
typedef float ctc_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void ctc_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <class TIn, int VT>
__global__ __launch_bounds__(64 * kCtcMfmaWaves) void k_ctc_logits_mfma(const TIn* __restrict__ enc,
                                                                        const float* __restrict__ W,
                                                                        const float* __restrict__ bias,
                                                                        float* __restrict__ log_probs,
                                                                        int* __restrict__ best, long frames, int Hd,
                                                                        int V) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ldw = Hd + 4;
  float* Wl = lds;  // [32 VT][ldw], rows >= V are zero
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* strip = lds + (size_t)32 * VT * ldw + (size_t)wave * 32 * kCtcStripLd;
  for (int i = threadIdx.x; i < 32 * VT * (Hd / 4); i += 64 * kCtcMfmaWaves) {
    const int v = i / (Hd / 4), c = (i % (Hd / 4)) * 4;
    ctc_f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
    if (v < V) w4 = *reinterpret_cast<const ctc_f32x4*>(&W[(long)v * Hd + c]);
    *reinterpret_cast<ctc_f32x4*>(&Wl[v * ldw + c]) = w4;
  }
  __syncthreads();

  const int fr = lane & 31, hh = lane >> 5;
  // bias of the vocabulary entries this lane's accumulator registers hold; -inf marks padding
  float bl[VT][16];
#pragma unroll
  for (int t = 0; t < VT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int v = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
      bl[t][i] = v < V ? bias[v] : -INFINITY;
    }

  constexpr bool kBf16 = sizeof(TIn) == 2;
  constexpr int NLD = kBf16 ? 2 : 4;  // 16-byte global loads per lane and 32-wide k chunk
  const int nchunks = Hd / 32;
  const long ntiles = (frames + 31) / 32;
  const long tile_step = (long)gridDim.x * kCtcMfmaWaves;
  long tile = (long)blockIdx.x * kCtcMfmaWaves + wave;
  if (tile >= ntiles) return;  // (after the only workgroup barrier)

  ctc_f32x4 raw[NLD];
  auto fetch = [&](long tl, int chunk) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int row = kBf16 ? (lane >> 2) + 16 * i : (lane >> 3) + 8 * i;
      const int col = kBf16 ? 8 * (lane & 3) : 4 * (lane & 7);
      const long f = min(tl * 32 + row, frames - 1);  // rows past the end repeat the last frame; never stored
      raw[i] = *reinterpret_cast<const ctc_f32x4*>(
          reinterpret_cast<const char*>(enc) + ((f * Hd + chunk * 32 + col) * (long)sizeof(TIn)));
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (kBf16) {
        const int row = (lane >> 2) + 16 * i, col = 8 * (lane & 3);
        const unsigned* p = reinterpret_cast<const unsigned*>(&raw[i]);
        ctc_f32x4 lo, hi;
        lo[0] = __uint_as_float(p[0] << 16);
        lo[1] = __uint_as_float(p[0] & 0xffff0000u);
        lo[2] = __uint_as_float(p[1] << 16);
        lo[3] = __uint_as_float(p[1] & 0xffff0000u);
        hi[0] = __uint_as_float(p[2] << 16);
        hi[1] = __uint_as_float(p[2] & 0xffff0000u);
        hi[2] = __uint_as_float(p[3] << 16);
        hi[3] = __uint_as_float(p[3] & 0xffff0000u);
        *reinterpret_cast<ctc_f32x4*>(&strip[row * kCtcStripLd + col]) = lo;
        *reinterpret_cast<ctc_f32x4*>(&strip[row * kCtcStripLd + col + 4]) = hi;
      } else {
        const int row = (lane >> 3) + 8 * i, col = 4 * (lane & 7);
        *reinterpret_cast<ctc_f32x4*>(&strip[row * kCtcStripLd + col]) = raw[i];
      }
    }
  };

  fetch(tile, 0);
  stage();
  ctc_lds_fence();
  while (tile < ntiles) {
    ctc_f32x16 acc[VT];
#pragma unroll
    for (int t = 0; t < VT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const long next_tile = tile + tile_step;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      // the next chunk (or the first chunk of the wave's next tile) travels while this one is multiplied
      const bool last = chunk + 1 == nchunks;
      const bool more = !last || next_tile < ntiles;
      if (more) fetch(last ? next_tile : tile, last ? 0 : chunk + 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const ctc_f32x4 e = *reinterpret_cast<const ctc_f32x4*>(&strip[fr * kCtcStripLd + 8 * g + 4 * hh]);
        ctc_f32x4 wv[VT];
#pragma unroll
        for (int t = 0; t < VT; ++t)
          wv[t] = *reinterpret_cast<const ctc_f32x4*>(&Wl[(32 * t + fr) * ldw + chunk * 32 + 8 * g + 4 * hh]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < VT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t][j], e[j], acc[t], 0, 0, 0);
      }
      if (last) {
        // ---- log-softmax and arg-max of frame tile*32 + fr: own registers, then lane ^ 32 ----
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < VT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            acc[t][i] += bl[t][i];  // padding entries become -inf
            mx = fmaxf(mx, acc[t][i]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float se = 0.f;
#pragma unroll
        for (int t = 0; t < VT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) se += expf(acc[t][i] - mx);
        se += __shfl_xor(se, 32);
        const float lse = mx + logf(se);
        float bv = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int t = 0; t < VT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) {  // increasing v within the lane: strict > keeps the first maximum
            acc[t][i] -= lse;
            const int v = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (acc[t][i] > bv) {
              bv = acc[t][i];
              bi = v;
            }
          }
        {
          const float ov = __shfl_xor(bv, 32);
          const int oi = __shfl_xor(bi, 32);
          if (ov > bv || (ov == bv && oi < bi)) bi = oi;
        }
        const long f = tile * 32 + fr;
        if (hh == 0 && f < frames) best[f] = bi;
        if (log_probs) {
          // 16 frames at a time through the strip: [frame][V] rows leave as one contiguous block
          ctc_lds_fence();
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            if ((fr >> 4) == half) {
#pragma unroll
              for (int t = 0; t < VT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                  const int v = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
                  if (v < V) strip[(fr & 15) * V + v] = acc[t][i];
                }
            }
            ctc_lds_fence();
            const long fbase = tile * 32 + 16 * half;
            const long nrows = min(16L, frames - fbase);
            float* dst = log_probs + fbase * V;
            for (int idx = lane; idx < nrows * V; idx += 64) dst[idx] = strip[idx];
            ctc_lds_fence();
          }
        }
      }
      ctc_lds_fence();  // every read of the strip has returned
      if (more) stage();
      ctc_lds_fence();
    }
    tile = next_tile;
  }
}

// tokens[b][0 .. counts[b]) = collapsed arg-max path of utterance b; the rest of the row is -1
__global__ __launch_bounds__(kCtcThreads) void k_ctc_collapse(const int* __restrict__ best,
                                                              const int* __restrict__ lengths, int T, int blank,
                                                              int* __restrict__ tokens, int* __restrict__ counts) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const int* bp = best + (long)b * T;
  int* out = tokens + (long)b * T;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += kCtcThreads) {
    const int t = t0 + tid;
    const int tok = t < len ? bp[t] : blank;
    const int prev = (t > 0 && t < len) ? bp[t - 1] : -1;
    const int keep = (t < len && tok != blank && tok != prev) ? 1 : 0;
    // inclusive scan inside the wave, then across the four waves
    int s = keep;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(s, o);
      if (lane >= o) s += v;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int before = carry_s;
    for (int w2 = 0; w2 < wave; ++w2) before += wsum[w2];
    if (keep) out[before + s - 1] = tok;
    __syncthreads();
    if (tid == 0) carry_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  const int total = carry_s;
  for (int t = total + tid; t < T; t += kCtcThreads) out[t] = -1;
  if (tid == 0) counts[b] = total;
}

static int launch_collapse(const int* best, const int32_t* lengths, int B, int T, int blank, int32_t* tokens,
                           int32_t* counts, hipStream_t st) {
  {
    ProfScope ps("k_ctc_collapse", st);
    hipLaunchKernelGGL(k_ctc_collapse, dim3(B), dim3(kCtcThreads), 0, st, best, lengths, T, blank, tokens, counts);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_ctc_collapse failed: %s", hipGetErrorString(e));
  return 0;
}

// ---------------------------------------------------------------------------
// Target packing for the CTC loss: the padded [B][Lmax] label matrix -> the labels of all utterances
// back to back, as LMUASRModel.compute_loss builds them with one .tolist() per utterance
// (src/models/asr_model.py:145-151).  offsets[b] = first packed position of utterance b,
// offsets[B] = total.
// ---------------------------------------------------------------------------
constexpr int kPackThreads = 256;
__global__ __launch_bounds__(kPackThreads) void k_ctc_target_offsets(const int32_t* __restrict__ lengths, int B, int Lmax,
                                                                    int32_t* __restrict__ offsets) {
  __shared__ int part[kPackThreads];
  const int per = (B + kPackThreads - 1) / kPackThreads;
  const int b0 = threadIdx.x * per, b1 = min(B, b0 + per);
  int s = 0;
  for (int b = b0; b < b1; ++b) s += min(max(lengths[b], 0), Lmax);
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < kPackThreads; ++i) {
      const int v = part[i];
      part[i] = run;
      run += v;
    }
    offsets[B] = run;
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int b = b0; b < b1; ++b) {
    offsets[b] = run;
    run += min(max(lengths[b], 0), Lmax);
  }
}

__global__ __launch_bounds__(kPackThreads) void k_ctc_pack_targets(const int64_t* __restrict__ targets, long ld,
                                                                  const int32_t* __restrict__ offsets,
                                                                  int64_t* __restrict__ packed) {
  const int b = blockIdx.x;
  const int o = offsets[b], len = offsets[b + 1] - o;
  for (int i = threadIdx.x; i < len; i += kPackThreads) packed[o + i] = targets[(long)b * ld + i];
}

}  // namespace lmufft

using namespace lmufft;

extern "C" {

size_t lmufft_ctc_workspace_bytes(int32_t B, int32_t T) {
  if (B <= 0 || T <= 0) return 0;
  return (sizeof(int32_t) * (size_t)B * T + 255) & ~size_t(255);
}

int lmufft_ctc_greedy(const void* enc, int dtype, const float* W, const float* bias, const int32_t* lengths,
                      int32_t B, int32_t T, int32_t Hd, int32_t V, int32_t blank, float* log_probs, int32_t* tokens,
                      int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
  if (B <= 0 || T <= 0 || Hd <= 0 || V <= 0) return set_error("ctc: bad shape (B=%d T=%d Hd=%d V=%d)", B, T, Hd, V);
  if (V > 64) return set_error("ctc: vocabulary of %d entries; the kernel holds one entry per lane (<= 64)", V);
  if (Hd > kCtcMaxHd) return set_error("ctc: encoder width %d exceeds %d", Hd, kCtcMaxHd);
  if (blank < 0 || blank >= V) return set_error("ctc: blank index %d outside the vocabulary", blank);
  if (dtype != LMUFFT_F32 && dtype != LMUFFT_BF16) return set_error("ctc: unknown dtype %d", dtype);
  if (!enc || !W || !bias || !tokens || !counts) return set_error("ctc: NULL pointer");
  const size_t need = lmufft_ctc_workspace_bytes(B, T);
  if (!workspace || workspace_bytes < need)
    return set_error("ctc: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  int* best = (int*)workspace;
  const long frames = (long)B * T;
  // matrix-pipe kernel when its layout assumptions hold and W fits the LDS beside the strips
  {
    const int VT = V <= 32 ? 1 : 2;
    const size_t lds_m = sizeof(float) * ((size_t)32 * VT * (Hd + 4) + (size_t)kCtcMfmaWaves * 32 * kCtcStripLd);
    const bool aligned = ((uintptr_t)enc & 15) == 0 && ((uintptr_t)W & 15) == 0;
    static_assert(16 * 64 <= 32 * kCtcStripLd, "16 frames of log-probabilities fit the strip");
    if (Hd % 32 == 0 && aligned && lds_m <= 160 * 1024) {
      const void* fm = dtype == LMUFFT_F32
                           ? (VT == 1 ? (const void*)k_ctc_logits_mfma<float, 1> : (const void*)k_ctc_logits_mfma<float, 2>)
                           : (VT == 1 ? (const void*)k_ctc_logits_mfma<ctc_bf16, 1>
                                      : (const void*)k_ctc_logits_mfma<ctc_bf16, 2>);
      if (lds_m > 64 * 1024 &&
          hipFuncSetAttribute(fm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m) != hipSuccess)
        return set_error("cannot reserve %zu bytes of LDS for k_ctc_logits_mfma", lds_m);
      const long ntiles = (frames + 31) / 32;
      const int grid_m = (int)std::min<long>((ntiles + kCtcMfmaWaves - 1) / kCtcMfmaWaves, 256);
      {
        ProfScope ps("k_ctc_logits", st);
#define LAUNCH_CTC_MFMA(TIN, VTV)                                                                            \
  hipLaunchKernelGGL((k_ctc_logits_mfma<TIN, VTV>), dim3(grid_m), dim3(64 * kCtcMfmaWaves), lds_m, st,       \
                     (const TIN*)enc, W, bias, log_probs, best, frames, Hd, V)
        if (dtype == LMUFFT_F32) {
          if (VT == 1) LAUNCH_CTC_MFMA(float, 1); else LAUNCH_CTC_MFMA(float, 2);
        } else {
          if (VT == 1) LAUNCH_CTC_MFMA(ctc_bf16, 1); else LAUNCH_CTC_MFMA(ctc_bf16, 2);
        }
#undef LAUNCH_CTC_MFMA
      }
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return set_error("launch of k_ctc_logits_mfma failed: %s", hipGetErrorString(e));
      return launch_collapse(best, lengths, B, T, blank, tokens, counts, st);
    }
  }
  const size_t lds = sizeof(float) * ((size_t)Hd * (V + 1) + 4 * (size_t)Hd);
  if (lds > 160 * 1024) return set_error("ctc: %zu bytes of LDS needed", lds);
  const void* fn = dtype == LMUFFT_F32 ? (const void*)k_ctc_logits<float> : (const void*)k_ctc_logits<ctc_bf16>;
  if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return set_error("cannot reserve %zu bytes of LDS for k_ctc_logits", lds);
  const int grid = (int)std::min<long>((frames + 3) / 4, 1024);
  {
    ProfScope ps("k_ctc_logits", st);
    if (dtype == LMUFFT_F32)
      hipLaunchKernelGGL(k_ctc_logits<float>, dim3(grid), dim3(kCtcThreads), lds, st, (const float*)enc, W, bias,
                         log_probs, best, frames, Hd, V);
    else
      hipLaunchKernelGGL(k_ctc_logits<ctc_bf16>, dim3(grid), dim3(kCtcThreads), lds, st, (const ctc_bf16*)enc, W, bias,
                         log_probs, best, frames, Hd, V);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_ctc_logits failed: %s", hipGetErrorString(e));
  return launch_collapse(best, lengths, B, T, blank, tokens, counts, st);
}

int lmufft_ctc_pack_targets(const int64_t* targets, const int32_t* target_lengths, int32_t B, int32_t Lmax, int64_t ld,
                            int64_t* packed, int32_t* offsets, void* stream) {
  if (B <= 0 || Lmax <= 0 || ld < Lmax) return set_error("ctc pack: bad shape (B=%d, Lmax=%d, ld=%lld)", B, Lmax, (long long)ld);
  if ((long)B * Lmax > INT32_MAX) return set_error("ctc pack: %ld labels exceed the index range", (long)B * Lmax);
  if (!targets || !target_lengths || !packed || !offsets) return set_error("ctc pack: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  {
    ProfScope ps("k_ctc_pack_targets", st);
    hipLaunchKernelGGL(k_ctc_target_offsets, dim3(1), dim3(kPackThreads), 0, st, target_lengths, B, Lmax, offsets);
    hipLaunchKernelGGL(k_ctc_pack_targets, dim3(B), dim3(kPackThreads), 0, st, targets, (long)ld, offsets, packed);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_ctc_pack_targets failed: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"

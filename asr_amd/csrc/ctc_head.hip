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

// ---------------------------------------------------------------------------
// CTC loss and its gradient (training).  The reference uses nn.CTCLoss(blank = vocab_size - 1,
// reduction = 'mean', zero_infinity = True), src/models/asr_model.py:105, 130-156.  Per utterance
// (states 0..2L: even = blank, odd = label (s-1)/2; p_t = exp lp_t):
//   alpha_t(s) = p_t(c_s) (alpha_{t-1}(s) + alpha_{t-1}(s-1) + [c_s != c_{s-2}] alpha_{t-1}(s-2))
//   beta_t(s)  = p_t(c_s) (beta_{t+1}(s)  + beta_{t+1}(s+1)  + [c_s != c_{s+2}] beta_{t+1}(s+2))
//   nll = -log(alpha_{Tb-1}(2L) + alpha_{Tb-1}(2L-1))
//   grad_t(c) = w (p_t(c) - sum_{s: c_s = c} alpha_t(s) beta_t(s) / (p_t(c) exp(-nll)))   for t < Tb, else 0
// -- the gradient torch's ctc_loss returns for log_probs (it is the gradient through the log-softmax that produced
// them: its rows sum to zero), w = 1 / (B max(L, 1)) for the mean reduction.
//
// Two launches.  k_ctc_alpha_beta is the serial part: one workgroup per utterance, thread l owns blank state 2l
// and label state 2l+1, and the two recursions -- which do not depend on each other -- run at the same time in two
// halves of the workgroup (one after the other when 2 x 16 waves would not fit).  The recursions run in the LINEAR
// domain on doubles, rescaled every fourth frame by a power of two (exact): the state variables of frame t are stored
// as alpha_t 2^-E_t with the integer E_t beside them.  A step is then one exchange of the neighbour's value through
// double-buffered LDS, two additions and one multiplication -- the log-domain form (a maximum, three exponentials and
// a logarithm on the dependent chain of every step) took three times as long.  A shift is the exponent of the
// largest state of the frame before it, found by a DPP wave reduction.  A target of up to 63 labels is one wave per
// recursion: there the neighbour's value comes by a DPP lane shift and the LDS is not used at all.  exp lp_t and the log-probabilities themselves are fetched sixteen frames ahead, since nothing else
// hides a global load here.  k_ctc_grad is the parallel part: one wave per frame, lane c sums alpha beta over the
// list of class c's label positions (blank: a wave reduction) in double and applies the scales -- fixed orders, no
// atomics.
// ---------------------------------------------------------------------------

// other lanes' LDS stores feed this lane's next loads: the LDS serves a wave's instructions in order, the compiler
// only has to keep them in order
__device__ __forceinline__ void ctc_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// workgroup barrier for LDS traffic only: global loads and stores stay in flight across it
template <int NW>
__device__ __forceinline__ void ctc_sync() {
  if (NW == 1) {
    ctc_wave_sync();
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
}

__device__ const float kCtcNegInf = -INFINITY;  // what a thread without a state "loads" as its log-probability
constexpr int kCtcRescale = 4;  // frames between two rescalings (doubles hold 4 frames of any float probabilities)
constexpr int kCtcAhead = 16;  // frames of log-probabilities in flight per thread (loads queue behind the stores of the
                               // state variables: the distance has to cover a store's way to HBM)

__device__ __forceinline__ int ctc_exponent(double v) { return (__double2hiint(v) >> 20) & 0x7FF; }  // biased
// 2^-(e - 1023) for the biased exponent e of the largest state; 1 when every state is zero (or e is out of range)
__device__ __forceinline__ double ctc_unscale(int e, int& shift) {
  shift = (e >= 24 && e <= 2022) ? e - 1023 : 0;
  return __hiloint2double((1023 - shift) << 20, 0);
}
// Largest value of the wave (uniform result), without the LDS crossbar: butterflies inside the rows of 16 lanes by
// DPP, then the four rows through scalar registers.  All 64 lanes must be active.
__device__ __forceinline__ int ctc_wave_max(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false));  // row_mirror
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// the value of lane - 1 (lane 0: zero) / of lane + 1 (lane 63: zero), by DPP wave shifts
__device__ __forceinline__ double ctc_lane_below(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xF, 0xF, false);  // wave_shr:1
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ctc_lane_above(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xF, 0xF, false);  // wave_shl:1
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// One recursion over the frames of an utterance.  FWD: alpha, frames 0 .. Tb-1; otherwise beta, frames Tb-1 .. 0.
// xb / xl: [2][NT + 2] exchange buffers of the blank / label values, thread l at index l + 1, both ends 0;
// xe: [2][NW] the waves' largest exponents.  w: the state variables [T][2 NT], ex: the frames' cumulative shifts [T].
// Every thread of the group makes Tb - 1 calls of ctc_sync.  Returns the buffer the last step wrote; E: the last
// frame's cumulative shift.  When the recursion is one wave (NW = 1) the neighbours' values come by DPP lane shifts
// and the LDS is not used at all.
// exp(x) as a double for x <= 0 without the fp32 underflow at x < -87 (a log-probability of -100 is a tiny but
// nonzero probability; as an exact 0 it would make every path through that frame impossible and the utterance's
// loss infinite): x = e ln 2 + r with e an integer, exp(r) in fp32 (r within +-0.35: the accuracy of the fp32
// form), times 2^e applied to the double's exponent.  Below -700 (and for -inf) the result is 0.  (The recursions
// rescale every kCtcRescale-th frame, so the floor that matters is the product of that many frames: log-probabilities
// down to about -170 per frame on a needed path stay representable.)
__device__ __forceinline__ double ctc_exp(float x) {
  const float xc = fmaxf(x, -700.f);
  const float e = rintf(xc * 1.44269504f);
  float r = __builtin_fmaf(e, -0.693359375f, xc);  // Cody-Waite: ln 2 = 0.693359375 - 2.12194440e-4
  r = __builtin_fmaf(e, 2.12194440e-4f, r);
  const double p = (double)__expf(r) * __hiloint2double((1023 + (int)e) << 20, 0);
  return x < -700.f ? 0.0 : p;
}

template <int NW, bool FWD>
__device__ __forceinline__ int ctc_sweep(const float* __restrict__ lpb, int V, int blank, int my_label, int l, int L, int Tb,
                                         bool skip, double* __restrict__ w, int* __restrict__ ex,
                                         double (*xb)[64 * NW + 2], double (*xl)[64 * NW + 2], int (*xe)[NW],
                                         double& vb, double& vl, int& E) {
  constexpr int NT = 64 * NW;
  const int wave = l >> 6;
  const int t0 = FWD ? 0 : Tb - 1;
  {
    const float* row = lpb + (long)t0 * V;
    if (FWD) {
      vb = l == 0 ? ctc_exp(row[blank]) : 0.0;
      vl = (l == 0 && L > 0) ? ctc_exp(row[my_label]) : 0.0;
    } else {
      vb = l == L ? ctc_exp(row[blank]) : 0.0;
      vl = l == L - 1 ? ctc_exp(row[my_label]) : 0.0;
    }
  }
  E = 0;
  int cur = 0;
  if (NW > 1) {
    xb[0][l + 1] = vb;
    xl[0][l + 1] = vl;
  }
  if (l == 0) {
    if (NW > 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) xb[i][0] = xl[i][0] = xb[i][NT + 1] = xl[i][NT + 1] = 0.0;
    }
    ex[t0] = 0;
  }
  int emax = ctc_wave_max(max(ctc_exponent(vb), ctc_exponent(vl)));
  if (NW > 1 && (l & 63) == 0) xe[0][wave] = emax;
  *reinterpret_cast<double2*>(w + (long)t0 * 2 * NT + 2 * l) = make_double2(vb, vl);
  // A thread without a blank (l > L) or label (l >= L) state reads -inf from one fixed cell instead of its row:
  // its probability is 0 and so is its state, with no test inside the loop.
  const float* colb = l <= L ? lpb + blank : &kCtcNegInf;
  const float* coll = l < L ? lpb + my_label : &kCtcNegInf;
  const long strb = l <= L ? V : 0, strl = l < L ? V : 0;
  float qb[kCtcAhead], ql[kCtcAhead];
#pragma unroll
  for (int i = 0; i < kCtcAhead; ++i) {
    const int k = min(1 + i, Tb - 1);
    const long t = FWD ? k : Tb - 1 - k;
    qb[i] = colb[t * strb];
    ql[i] = coll[t * strl];
  }
  if (NW > 1) ctc_sync<NW>();
  static_assert(kCtcAhead % kCtcRescale == 0, "the unrolled block holds whole rescaling periods");
  for (int k0 = 1; k0 < Tb; k0 += kCtcAhead) {
    float nqb[kCtcAhead], nql[kCtcAhead];
#pragma unroll
    for (int i = 0; i < kCtcAhead; ++i) {
      const int k = min(k0 + kCtcAhead + i, Tb - 1);
      const long t = FWD ? k : Tb - 1 - k;
      nqb[i] = colb[t * strb];
      nql[i] = coll[t * strl];
    }
#pragma unroll
    for (int i = 0; i < kCtcAhead; ++i) {
      const int k = k0 + i;
      if (k < Tb) {  // uniform over the workgroup
        const int t = FWD ? k : Tb - 1 - k;
        double pb = ctc_exp(qb[i]), pl = ctc_exp(ql[i]);
        if (i % kCtcRescale == 0) {  // the shift found at the end of the last period
          if (NW > 1) {
#pragma unroll
            for (int j = 0; j < NW; ++j) emax = max(emax, xe[cur][j]);
          }
          int shift;
          const double un = ctc_unscale(emax, shift);
          E += shift;
          pb *= un;
          pl *= un;
        }
        double nb, nl;
        if (FWD) {
          const double prev = NW == 1 ? ctc_lane_below(vl) : xl[cur][l];  // label state of thread l - 1
          nb = (vb + prev) * pb;
          nl = (vl + vb + (skip ? prev : 0.0)) * pl;
        } else {
          const double nxb = NW == 1 ? ctc_lane_above(vb) : xb[cur][l + 2];  // states of thread l + 1
          const double nxl = NW == 1 ? ctc_lane_above(vl) : xl[cur][l + 2];
          nb = (vb + vl) * pb;
          nl = (vl + nxb + (skip ? nxl : 0.0)) * pl;
        }
        vb = nb;
        vl = nl;
        cur ^= 1;
        if (NW > 1) {
          if (!FWD) xb[cur][l + 1] = vb;
          xl[cur][l + 1] = vl;
        }
        if (i % kCtcRescale == kCtcRescale - 1) {  // the next step rescales: the largest state of this frame
          emax = ctc_wave_max(max(ctc_exponent(vb), ctc_exponent(vl)));
          if (NW > 1 && (l & 63) == 0) xe[cur][wave] = emax;
        }
        *reinterpret_cast<double2*>(w + (long)t * 2 * NT + 2 * l) = make_double2(vb, vl);
        if (wave == 0) ex[t] = E;  // every lane the same value to the same word
        if (NW > 1) ctc_sync<NW>();
      }
    }
#pragma unroll
    for (int i = 0; i < kCtcAhead; ++i) {
      qb[i] = nqb[i];
      ql[i] = nql[i];
    }
  }
  return cur;
}

// workspace of the two kernels: alpha [B][T][2 NT], beta [B][T][2 NT], nll [B] (doubles), then the cumulative
// shifts of alpha and of beta, [B][T] int32 each
template <int NW, bool BOTH>
__global__ __launch_bounds__(64 * NW * (BOTH ? 2 : 1)) void k_ctc_alpha_beta(
    const float* __restrict__ log_probs, const int64_t* __restrict__ targets, long ld, const int32_t* __restrict__ input_lengths,
    const int32_t* __restrict__ target_lengths, int B, int T, int V, int blank, float* __restrict__ nll_out,
    double* __restrict__ ws) {
  constexpr int NT = 64 * NW;
  constexpr int ND = BOTH ? 2 : 1;
  __shared__ double xb[ND][2][NT + 2], xl[ND][2][NT + 2];
  __shared__ int xe[ND][2][NW];
  __shared__ int lab[NT + 1];
  const int b = blockIdx.x;
  const int l = threadIdx.x & (NT - 1), half = BOTH ? threadIdx.x / NT : 0;
  const int Tb = min(max(input_lengths[b], 0), T);
  const int L = min(max(target_lengths[b], 0), NT - 1);
  const float* lpb = log_probs + (long)b * T * V;
  double* alpha = ws + (long)b * T * 2 * NT;
  double* beta = ws + ((long)B + b) * T * 2 * NT;
  double* nll_d = ws + (long)2 * B * T * 2 * NT;
  int* ex_alpha = reinterpret_cast<int*>(nll_d + B) + (long)b * T;
  int* ex_beta = ex_alpha + (long)B * T;
  const int my_label = l < L ? min(max((int)targets[(long)b * ld + l], 0), V - 1) : blank;
  if (half == 0) lab[l] = my_label;
  if (threadIdx.x == 0) lab[NT] = blank;
  __syncthreads();
  if (Tb == 0) {  // no frames: nothing aligns (torch returns inf here)
    if (threadIdx.x == 0) {
      nll_out[b] = INFINITY;
      nll_d[b] = INFINITY;
    }
    return;
  }
  const bool skip_prev = l >= 1 && l < L && lab[l - 1] != my_label;  // label l may be entered from label l-1
  const bool skip_next = l + 1 < L && lab[l + 1] != my_label;        // label l may move on to label l+1
  double vb, vl;
  int E;
  if (half == 0) {
    const int cur = ctc_sweep<NW, true>(lpb, V, blank, my_label, l, L, Tb, skip_prev, alpha, ex_alpha, xb[0], xl[0], xe[0],
                                        vb, vl, E);
    const double below = NW == 1 ? ctc_lane_below(vl) : xl[0][cur][l];  // label value of thread l-1
    if (l == L) {  // vb = alpha(2L), below = alpha(2L-1)
      const double ll = log(vb + below) + (double)E * 0.6931471805599453;
      nll_out[b] = (float)-ll;
      nll_d[b] = -ll;
    }
  }
  if (!BOTH) __syncthreads();  // the exchange buffers are used again
  if (half == ND - 1)
    ctc_sweep<NW, false>(lpb, V, blank, my_label, l, L, Tb, skip_next, beta, ex_beta, xb[ND - 1], xl[ND - 1], xe[ND - 1], vb, vl,
                         E);
}

constexpr int kCtcGradFrames = 16;  // frames per workgroup
// waves per workgroup, one frame per wave at a time (two for the longest targets: a wave's products take 16 KB of LDS)
constexpr int ctc_grad_waves(int NW) { return NW >= 16 ? 2 : 4; }

template <int NW>
__global__ __launch_bounds__(64 * ctc_grad_waves(NW)) void k_ctc_grad(
    const float* __restrict__ log_probs, const int64_t* __restrict__ targets, long ld, const int32_t* __restrict__ input_lengths,
    const int32_t* __restrict__ target_lengths, int B, int T, int V, int blank, int mean, int zero_infinity,
    float* __restrict__ grad, const double* __restrict__ ws) {
  constexpr int NT = 64 * NW;
  constexpr int kCtcGradWaves = ctc_grad_waves(NW);
  __shared__ double x[kCtcGradWaves][2 * NT];
  __shared__ int lab[NT];
  __shared__ int cls_start[65], cls_pos[NT];
  const int b = blockIdx.y, t_first = blockIdx.x * kCtcGradFrames;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int Tb = min(max(input_lengths[b], 0), T);
  const int L = min(max(target_lengths[b], 0), NT - 1);
  const double* nll_d = ws + (long)2 * B * T * 2 * NT;
  const int* ex_alpha = reinterpret_cast<const int*>(nll_d + B) + (long)b * T;
  const int* ex_beta = ex_alpha + (long)B * T;
  const double nll = nll_d[b];
  const bool dead = !(nll < (double)INFINITY);  // no valid alignment (or NaN)
  float* gb = grad + (long)b * T * V;
  const int t_end = min(t_first + kCtcGradFrames, T);
  if (t_first >= Tb || (dead && zero_infinity)) {  // frames past the utterance's end, or a zeroed utterance
    for (int i = threadIdx.x; i < (t_end - t_first) * V; i += 64 * kCtcGradWaves) gb[(long)t_first * V + i] = 0.f;
    return;
  }
  // label positions grouped by class, ascending inside a class: thread c counts, a scan places, thread c fills
  for (int i = threadIdx.x; i < L; i += 64 * kCtcGradWaves) lab[i] = min(max((int)targets[(long)b * ld + i], 0), V - 1);
  __syncthreads();
  if (wave == 0) {
    int n = 0;
    if (lane < V)
      for (int i = 0; i < L; ++i) n += lab[i] == lane;
    int incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    int at = incl - n;
    cls_start[lane] = at;
    if (lane == 63) cls_start[64] = incl;
    if (lane < V)
      for (int i = 0; i < L; ++i)
        if (lab[i] == lane) cls_pos[at++] = i;
  }
  __syncthreads();
  const float w = mean ? 1.f / ((float)B * (float)max(L, 1)) : 1.f;
  const int S = 2 * L + 1;
  double* xw = x[wave];
  for (int t = t_first + wave; t < t_end; t += kCtcGradWaves) {
    float g = 0.f;
    if (t < Tb) {
      const float lpc = lane < V ? log_probs[((long)b * T + t) * V + lane] : 0.f;
      if (dead) {  // (zero_infinity off) no path through any state: the posterior term vanishes
        g = w * __expf(lpc);
      } else {
        const double* a = ws + ((long)b * T + t) * 2 * NT;
        const double* bt = ws + (((long)B + b) * T + t) * 2 * NT;
        // (path mass through the state) x p_t(class), scaled.  alpha and beta may each lie hundreds of binary orders
        // below the largest state of their frame, and a plain product of two such doubles underflows: the products
        // are formed on the mantissas and scaled by their exponents RELATIVE to the frame's largest product, 2^pmax.
        int pmax = 0;
        for (int s = lane; s < S; s += 64) {
          const int ea = ctc_exponent(a[s]), eb = ctc_exponent(bt[s]);
          pmax = max(pmax, (ea > 0 && eb > 0) ? ea + eb : 0);
        }
        pmax = ctc_wave_max(pmax);  // biased twice: true exponent pmax - 2046; 0: every product is zero
        for (int s = lane; s < S; s += 64) {
          const double av = a[s], bv = bt[s];
          const int ea = ctc_exponent(av), eb = ctc_exponent(bv), rel = ea + eb - pmax;  // <= 0
          const double ma = __hiloint2double((__double2hiint(av) & 0x800FFFFF) | (1023 << 20), __double2loint(av));
          const double mb = __hiloint2double((__double2hiint(bv) & 0x800FFFFF) | (1023 << 20), __double2loint(bv));
          xw[s] = (ea > 0 && eb > 0 && rel > -1000) ? ma * mb * __hiloint2double((1023 + rel) << 20, 0) : 0.0;
        }
        ctc_wave_sync();
        double sum = 0.0;
        for (int i = lane; i <= L; i += 64) sum += xw[2 * i];  // blank class: states 0, 2, .., 2L over the lanes
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane < V && lane != blank) {
          sum = 0.0;
          for (int i = cls_start[lane]; i < cls_start[lane + 1]; ++i) sum += xw[2 * cls_pos[i] + 1];
        }
        // occupancy = sum 2^(E_t + F_t) exp(nll) / p_t(c): the mantissa as it is, everything else as one exponent --
        // a few units below zero, so fp32 from here
        float occ = 0.f;
        if (sum > 0.0) {
          const int e2 = ctc_exponent(sum) - 1022;  // sum = m 2^e2, m in [0.5, 1)
          const double m = __hiloint2double((__double2hiint(sum) & 0x800FFFFF) | (1022 << 20), __double2loint(sum));
          const double arg = (double)(ex_alpha[t] + ex_beta[t] + e2 + pmax - 2046) * 0.6931471805599453 + nll - (double)lpc;
          occ = (float)m * __expf((float)arg);
        }
        g = w * (__expf(lpc) - occ);
        ctc_wave_sync();  // the next frame overwrites xw
      }
    }
    if (lane < V) gb[(long)t * V + lane] = g;
  }
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

}  // extern "C" (reopened below)

namespace lmufft {
// loss = mean_b( nll[b] / max(target_lengths[b], 1) ), infinite terms zeroed when asked: the tail of
// nn.CTCLoss(reduction='mean', zero_infinity) (src/models/asr_model.py:105) as one launch instead of seven
// elementwise ones.  One workgroup; a fixed summation order.
__global__ __launch_bounds__(256) void k_ctc_mean_loss(const float* __restrict__ nll, const int32_t* __restrict__ target_lengths,
                                                       int B, int zero_infinity, float* __restrict__ loss) {
  __shared__ float red[256];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const int len = target_lengths[b];
    const float per = nll[b] / (float)(len < 1 ? 1 : len);
    s += (zero_infinity && isinf(per)) ? 0.f : per;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] / (float)B;
}
}  // namespace lmufft

extern "C" {

int lmufft_ctc_mean_loss(const float* nll, const int32_t* target_lengths, int32_t B, int zero_infinity, float* loss,
                         void* stream) {
  if (B <= 0) return set_error("ctc mean loss: bad batch size %d", B);
  if (!nll || !target_lengths || !loss) return set_error("ctc mean loss: NULL pointer");
  hipLaunchKernelGGL(lmufft::k_ctc_mean_loss, dim3(1), dim3(256), 0, (hipStream_t)stream, nll, target_lengths, B,
                     zero_infinity, loss);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_ctc_mean_loss failed: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C" (reopened below)

namespace lmufft {
// Row-wise log-softmax over a vocabulary of at most 64 entries and its backward: one wave per row, lane v holds
// entry v.  The logits are rows of `ld` elements (the Linear kernel pads the vocabulary to a multiple of 8), bf16 or
// f32; the log-probabilities and their gradient are dense f32 [rows][V], as the CTC kernel takes and returns them.
//   forward:  lp = x - max - log(sum exp(x - max))            (torch.log_softmax(logits.float(), -1))
//   backward: dx = dlp - exp(lp) * sum_v dlp, columns V .. ld-1 zero
template <bool BF16>
__global__ __launch_bounds__(256) void k_log_softmax_rows(const void* __restrict__ logits, long ld, long rows, int V,
                                                          float* __restrict__ lp) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int v = threadIdx.x & 63;
  if (row >= rows) return;
  float x = -INFINITY;
  if (v < V)
    x = BF16 ? (float)reinterpret_cast<const __bf16*>(logits)[row * ld + v] : reinterpret_cast<const float*>(logits)[row * ld + v];
  float m = x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float e = v < V ? expf(x - m) : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
  if (v < V) lp[row * V + v] = x - m - logf(e);
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_log_softmax_rows_bwd(const float* __restrict__ dlp, const float* __restrict__ lp,
                                                              long rows, int V, void* __restrict__ dlogits, long ld) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int v = threadIdx.x & 63;
  if (row >= rows) return;
  const float g = v < V ? dlp[row * V + v] : 0.f;
  float s = g;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float d = v < V ? g - expf(lp[row * V + v]) * s : 0.f;
  if (v < ld) {
    if (BF16) reinterpret_cast<__bf16*>(dlogits)[row * ld + v] = (__bf16)d;
    else reinterpret_cast<float*>(dlogits)[row * ld + v] = d;
  }
}
}  // namespace lmufft

extern "C" {

int lmufft_log_softmax_rows(const void* logits, int dtype, int64_t ld, int64_t rows, int32_t V, float* log_probs,
                            void* stream) {
  if (rows < 0 || V <= 0 || V > 64 || ld < V || ld > 64) return set_error("log softmax: bad shape (rows=%lld V=%d ld=%lld; V <= ld <= 64)", (long long)rows, V, (long long)ld);
  if (dtype != LMUFFT_F32 && dtype != LMUFFT_BF16) return set_error("log softmax: unknown dtype %d", dtype);
  if (rows == 0) return 0;
  if (!logits || !log_probs) return set_error("log softmax: NULL pointer");
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LMUFFT_BF16)
    hipLaunchKernelGGL(lmufft::k_log_softmax_rows<true>, grid, dim3(256), 0, st, logits, (long)ld, (long)rows, V, log_probs);
  else
    hipLaunchKernelGGL(lmufft::k_log_softmax_rows<false>, grid, dim3(256), 0, st, logits, (long)ld, (long)rows, V, log_probs);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_log_softmax_rows failed: %s", hipGetErrorString(e));
  return 0;
}

int lmufft_log_softmax_rows_bwd(const float* dlog_probs, const float* log_probs, int64_t rows, int32_t V, void* dlogits,
                                int dtype, int64_t ld, void* stream) {
  if (rows < 0 || V <= 0 || V > 64 || ld < V || ld > 64) return set_error("log softmax backward: bad shape (rows=%lld V=%d ld=%lld; V <= ld <= 64)", (long long)rows, V, (long long)ld);
  if (dtype != LMUFFT_F32 && dtype != LMUFFT_BF16) return set_error("log softmax backward: unknown dtype %d", dtype);
  if (rows == 0) return 0;
  if (!dlog_probs || !log_probs || !dlogits) return set_error("log softmax backward: NULL pointer");
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LMUFFT_BF16)
    hipLaunchKernelGGL(lmufft::k_log_softmax_rows_bwd<true>, grid, dim3(256), 0, st, dlog_probs, log_probs, (long)rows, V, dlogits, (long)ld);
  else
    hipLaunchKernelGGL(lmufft::k_log_softmax_rows_bwd<false>, grid, dim3(256), 0, st, dlog_probs, log_probs, (long)rows, V, dlogits, (long)ld);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_log_softmax_rows_bwd failed: %s", hipGetErrorString(e));
  return 0;
}

// waves per workgroup: one thread per label position and one more, rounded up to 1, 2, 4, 8 or 16 waves
static int ctc_loss_waves(int Lmax) {
  const int need = (Lmax + 1 + 63) / 64;
  int nw = 1;
  while (nw < need) nw *= 2;
  return nw;
}

size_t lmufft_ctc_loss_workspace_bytes(int32_t B, int32_t T, int32_t Lmax) {
  if (B <= 0 || T <= 0 || Lmax < 0 || Lmax + 1 > 1024) return 0;
  return sizeof(double) * ((size_t)2 * B * T * 2 * 64 * (size_t)ctc_loss_waves(Lmax) + (size_t)B) +
         sizeof(int32_t) * (size_t)2 * B * T;
}

int lmufft_ctc_loss(const float* log_probs, const int64_t* targets, int64_t ld, const int32_t* input_lengths,
                    const int32_t* target_lengths, int32_t B, int32_t T, int32_t V, int32_t Lmax, int32_t blank,
                    int reduction_mean, int zero_infinity, float* nll, float* grad, void* workspace, size_t workspace_bytes,
                    void* stream) {
  if (B <= 0 || T <= 0 || V <= 0 || Lmax < 0 || ld < Lmax) return set_error("ctc loss: bad shape (B=%d T=%d V=%d Lmax=%d)", B, T, V, Lmax);
  if (V > 64) return set_error("ctc loss: vocabulary of %d entries (at most 64)", V);
  if (blank < 0 || blank >= V) return set_error("ctc loss: blank index %d outside the vocabulary", blank);
  if (Lmax + 1 > 1024) return set_error("ctc loss: targets of up to %d labels (at most 1023)", Lmax);
  if (!log_probs || (!targets && Lmax > 0) || !input_lengths || !target_lengths || !nll || !grad || !workspace)
    return set_error("ctc loss: NULL pointer");
  if (workspace_bytes < lmufft_ctc_loss_workspace_bytes(B, T, Lmax))
    return set_error("ctc loss: workspace of %zu bytes needed, %zu given", lmufft_ctc_loss_workspace_bytes(B, T, Lmax), workspace_bytes);
  if (((uintptr_t)workspace & 15) != 0) return set_error("ctc loss: the workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int nw = ctc_loss_waves(Lmax);
  const dim3 ggrid((T + kCtcGradFrames - 1) / kCtcGradFrames, B);
  {
    ProfScope ps("k_ctc_loss", st);
#define CTC_LOSS(NWV, BOTH)                                                                                             \
  do {                                                                                                                  \
    hipLaunchKernelGGL((k_ctc_alpha_beta<NWV, BOTH>), dim3(B), dim3(64 * NWV * (BOTH ? 2 : 1)), 0, st, log_probs,       \
                       targets, (long)ld, input_lengths, target_lengths, B, T, V, blank, nll, (double*)workspace);      \
    hipLaunchKernelGGL(k_ctc_grad<NWV>, ggrid, dim3(64 * ctc_grad_waves(NWV)), 0, st, log_probs, targets, (long)ld,     \
                       input_lengths, target_lengths, B, T, V, blank, reduction_mean, zero_infinity, grad,              \
                       (const double*)workspace);                                                                       \
  } while (0)
    if (nw <= 1) CTC_LOSS(1, true);
    else if (nw <= 2) CTC_LOSS(2, true);
    else if (nw <= 4) CTC_LOSS(4, true);
    else if (nw <= 8) CTC_LOSS(8, true);
    else CTC_LOSS(16, false);
#undef CTC_LOSS
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_ctc_loss failed: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"

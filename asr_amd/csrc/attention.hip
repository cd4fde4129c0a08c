// Masked multi-head self-attention core on gfx950 (inference kernels; training forward with dropout and
// the backward further down):
//   ctx[b, i, h*dk + d] = sum_j softmax_j( q_i . k_j / sqrt(dk), masked )[j] * v[b, j, h*dk + d]
// without materialising the [B, H, T, T] score and weight tensors.
//
// Reference arithmetic: MultiHeadAttention.forward, src/models/lmu_encoder.py:68-91 (scores =
// Q K^T / sqrt(d_k); scores.masked_fill(mask == 0, -1e9) with mask[b, i, j] = (i < len_b) and
// (j < len_b), built at src/models/lmu_encoder.py:247-252; softmax over j; dropout (identity in
// eval mode); context = weights V, heads concatenated).  Masked scores are SET to -1e9, so a query
// row past the utterance's end attends uniformly to all T positions -- reproduced here.
//
// fp32 tensors: fp32 arithmetic on the exact-fp32 MFMA (v_mfma_f32_32x32x2f32); bf16 tensors: the
// bf16 MFMA variant further down.  One wave owns 32 queries and
// walks the keys in tiles of 32.  It computes the TRANSPOSED score tile S^T[key][query] = K Q^T:
// in the MFMA output layout a lane then holds one query's scores (16 registers x 2 lane halves),
// so the online-softmax row statistics cost one cross-half shuffle, and the probability
// registers are, as they stand, the B operand of O^T[d][query] += V^T[d][key] P^T[key][query].
// Nothing goes through LDS between the two contractions.  K and V tiles are staged in LDS once
// per workgroup (4 waves = 128 queries of one head), next tile prefetched into registers.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stdint.h>

#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

typedef float attn_f16v __attribute__((ext_vector_type(16)));
typedef float attn_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 attn_bf16;

constexpr int kAttnKeys = 32;       // keys per tile
// what masked loads read (the address is selected, not the value: the load stays unconditional)
static __device__ __attribute__((aligned(16))) unsigned int attn_zero_line[4] = {0, 0, 0, 0};

template <class T>
__device__ __forceinline__ float attn_ld(const T* p) { return (float)*p; }

// Dropout of the attention weights (src/models/lmu_encoder.py:45, 86) without a stored mask: a row is one
// (b, h, query), its columns the keys; the hash (HashDropout) serves two neighbouring keys of a query.  Kernels
// whose lanes own a query hash once per key pair; the key-owning sweep shares a hash between the two lanes of a
// key pair.  The backward kernels evaluate the same function.
typedef HashDropout AttnDropout;

// TRAIN: dropout on the weights, and the row statistics (running maximum m, 1 / sum exp(s - m)) are saved for
// the backward as stats[b][h][query][2] (the pair, not m + log l: a fully masked row has m = -1e9, where
// fp32 cannot hold the logarithm beside it).
template <int DK, class TIO, int NW, bool TRAIN = false>
__global__ __launch_bounds__(64 * NW, DK <= 64 ? (TRAIN ? 2 : 4) : 1) void k_attention_fwd(const TIO* __restrict__ q, const TIO* __restrict__ k,
                                                                const TIO* __restrict__ v,
                                                                const int* __restrict__ lengths,
                                                                TIO* __restrict__ ctx, int T, int H, int ld,
                                                                float scale, AttnDropout dr = AttnDropout{0, 0, 0, 1.f},
                                                                float* __restrict__ stats = nullptr) {
  hash_resolve_epoch(dr);
  static_assert(DK % 32 == 0 && DK <= 128, "head width: 32, 64, 96 or 128");
  constexpr int KP = DK + 1;   // K tile pitch: lane = key, banks (key + d) distinct
  constexpr int VP = DK + 8;   // V tile pitch: lane = d, the two lane halves 4 rows apart land 32 banks apart
  constexpr int DT = DK / 32;  // 32-row tiles of O^T
  constexpr int NS = DK / 2;   // MFMA steps of the score contraction (2 values of d per step)
  __shared__ float Ks[kAttnKeys * KP];
  __shared__ float Vs[kAttnKeys * VP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = H * DK;
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const long base = (long)b * T * C + (long)h * DK;    // context rows are dense
  const long ibase = (long)b * T * ld + (long)h * DK;  // q, k, v rows are ld elements apart
  const int qi = blockIdx.x * (32 * NW) + wave * 32 + r;  // this lane's query
  const bool q_in = qi < T, q_valid = qi < len;

  // Q[qi][d] for d = s + hh*NS, pre-scaled by 1/sqrt(dk): the B operand of every score step
  float qreg[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) qreg[s] = q_in ? attn_ld(q + ibase + (long)qi * ld + s + hh * NS) * scale : 0.f;

  attn_f16v o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging: thread -> (row, 4-float chunk) of the K and V tiles
  constexpr int CH = DK / 4;                       // chunks per row
  constexpr int NV = (kAttnKeys * CH + 64 * NW - 1) / (64 * NW);  // chunks per thread and tensor
  attn_f4 kst[NV], vst[NV];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW), row = idx / CH, c = (idx % CH) * 4;
      if (idx >= kAttnKeys * CH) continue;
      const int j = min(j0 + row, T - 1);  // rows past T are masked below; keep the address valid
      const TIO* kp = k + ibase + (long)j * ld + c;
      const TIO* vp = v + ibase + (long)j * ld + c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kst[n][e] = attn_ld(kp + e);
        vst[n][e] = attn_ld(vp + e);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW), row = idx / CH, c = (idx % CH) * 4;
      if (idx >= kAttnKeys * CH) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Ks[row * KP + c + e] = kst[n][e];
        Vs[row * VP + c + e] = vst[n][e];
      }
    }
  };

  fetch(0);
  for (int j0 = 0; j0 < T; j0 += kAttnKeys) {
    __syncthreads();  // everybody has finished with the previous tile
    stage();
    __syncthreads();
    if (j0 + kAttnKeys < T) fetch(j0 + kAttnKeys);

    // S^T[key r'][query] : lane supplies K[key = r][d = s + hh*NS] and its own Q value
    attn_f16v st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[r * KP + s + hh * NS], qreg[s], st, 0, 0, 0);

    // register i of this lane: key j0 + rho_i + 4*hh, query qi
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int j = j0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      float sv = (q_valid && j < len) ? st[i] : -1e9f;
      sv = j < T ? sv : -INFINITY;
      st[i] = sv;
      mx = fmaxf(mx, sv);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);           // finite: every tile holds at least one key < T
    const float alpha = __expf(m_run - m_new);      // 0 on the first tile (m_run = -inf)
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      st[i] = __expf(st[i] - m_new);
      psum += st[i];
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (TRAIN && dr.threshold) {  // the normaliser keeps every weight; the context only the surviving ones
      const HashRow hrow = hash_row(dr, ((unsigned long long)b * H + h) * T + qi, (unsigned)(T + 1) >> 1);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int j = j0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        st[i] = hash_keep(dr, hrow, j) ? st[i] * dr.inv_keep : 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
    // O^T[d][query] += V^T[d][key] P^T[key][query]: A = V[key rho_i + 4hh][d = 32t + r], B = st[i]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
#pragma unroll
      for (int t = 0; t < DT; ++t)
        o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[row * VP + 32 * t + r], st[i], o[t], 0, 0, 0);
    }
  }

  // O^T / l -> ctx rows (through LDS: lanes hold queries, rows of ctx want lanes along d)
  __syncthreads();
  const float inv = 1.0f / l_run;
  if (TRAIN && hh == 0 && q_in) {
    float* sp = stats + 2 * ((((long)b * H + h) * T) + qi);
    sp[0] = m_run;
    sp[1] = inv;
  }
  // each wave owns a strip of 8 queries x KP floats of the (now free) K tile and empties its 32
  // queries in four passes
  constexpr int STRIP = 8 * KP;  // 4 waves x 8 x KP = the K tile; waves 4..7 use the V tile (VP > KP) likewise
  static_assert(NW <= 8, "two tiles hold eight strips");
  float* ws = (wave < 4 ? Ks : Vs) + (wave & 3) * STRIP;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    // queries 8*pass .. 8*pass+7 of this wave
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int d = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if ((r >> 3) == pass) ws[(r & 7) * KP + d] = o[t][i] * inv;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private strip: program order suffices
    for (int idx = lane; idx < 8 * DK; idx += 64) {
      const int ql = idx / DK, d = idx % DK;
      const int qq = blockIdx.x * (32 * NW) + wave * 32 + pass * 8 + ql;
      if (qq < T) ctx[base + (long)qq * C + d] = (TIO)ws[ql * KP + d];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// ---------------------------------------------------------------------------
// Backward of the attention core (training).  With P = softmax(S), S = Q K^T / sqrt(dk) where it is not
// masked, W = dropout(P), O = W V:
//   dV = W^T dO;  dW = dO V^T;  dP = dropout'(dW);  dS = P * (dP - D),  D_i = sum_j P_ij dP_ij = dO_i . O_i;
//   dS = 0 where the score was overwritten by the mask value (masked_fill passes no gradient);
//   dQ = dS K / sqrt(dk);  dK = dS^T Q / sqrt(dk).
// Two kernels, no atomics, each a transposed-tile sweep like the forward: k_attention_bwd_dq gives a lane
// one QUERY (sweeping key tiles: S^T and dW^T tiles by MFMA, P from the saved row statistics, dQ^T accumulated
// like the forward's O^T) and also writes D; k_attention_bwd_dkv gives a lane one KEY (sweeping query tiles:
// S and dW tiles with the lane's K and V rows as B operands, dV^T and dK^T accumulated).  fp32 arithmetic on
// the exact-fp32 MFMA for fp32 and bf16 tensors alike.  The dropout mask is recomputed from the element
// index (HashDropout), nothing [T x T] is ever stored.
// ---------------------------------------------------------------------------
template <int DK, class TIO, int NW>
__global__ __launch_bounds__(64 * NW) void k_attention_bwd_dq(const TIO* __restrict__ q, const TIO* __restrict__ k,
                                                              const TIO* __restrict__ v, const TIO* __restrict__ ctx,
                                                              const TIO* __restrict__ dctx, const float* __restrict__ stats,
                                                              const int* __restrict__ lengths, TIO* __restrict__ dq,
                                                              float* __restrict__ dsum, int T, int H, int ld, int ldg,
                                                              float scale, AttnDropout dr) {
  hash_resolve_epoch(dr);
  constexpr int KP = DK + 1, DT = DK / 32, NS = DK / 2;
  __shared__ float Ks[kAttnKeys * KP];
  __shared__ float Vs[kAttnKeys * KP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = H * DK;
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const long base = (long)b * T * C + (long)h * DK;    // ctx, dctx rows are dense
  const long ibase = (long)b * T * ld + (long)h * DK;  // q, k, v rows are ld elements apart
  const long gbase = (long)b * T * ldg + (long)h * DK; // dq rows are ldg elements apart
  const long bh = (long)b * H + h;
  const int qi = blockIdx.x * (32 * NW) + wave * 32 + r;
  const bool q_in = qi < T, q_valid = qi < len;

  float qreg[NS], doreg[NS], D = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    qreg[s] = q_in ? attn_ld(q + ibase + (long)qi * ld + s + hh * NS) * scale : 0.f;
    doreg[s] = q_in ? attn_ld(dctx + base + (long)qi * C + s + hh * NS) : 0.f;
    D += q_in ? doreg[s] * attn_ld(ctx + base + (long)qi * C + s + hh * NS) : 0.f;
  }
  D += __shfl_xor(D, 32);
  if (hh == 0 && q_in) dsum[bh * T + qi] = D;
  const float* sp = q_in ? stats + 2 * (bh * T + qi) : reinterpret_cast<const float*>(attn_zero_line);
  const float m = sp[0], invl = sp[1];

  attn_f16v dqa[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) dqa[t][i] = 0.f;

  constexpr int CH = DK / 4;
  constexpr int NV = (kAttnKeys * CH + 64 * NW - 1) / (64 * NW);
  attn_f4 kst[NV], vst[NV];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW), row = idx / CH, c = (idx % CH) * 4;
      if (idx >= kAttnKeys * CH) continue;
      const int j = min(j0 + row, T - 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kst[n][e] = attn_ld(k + ibase + (long)j * ld + c + e);
        vst[n][e] = attn_ld(v + ibase + (long)j * ld + c + e);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW), row = idx / CH, c = (idx % CH) * 4;
      if (idx >= kAttnKeys * CH) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Ks[row * KP + c + e] = kst[n][e];
        Vs[row * KP + c + e] = vst[n][e];
      }
    }
  };
  const HashRow hrow = hash_row(dr, (unsigned long long)bh * T + qi, (unsigned)(T + 1) >> 1);

  fetch(0);
  for (int j0 = 0; j0 < T; j0 += kAttnKeys) {
    __syncthreads();
    stage();
    __syncthreads();
    if (j0 + kAttnKeys < T) fetch(j0 + kAttnKeys);

    attn_f16v st, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f, dp[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[r * KP + s + hh * NS], qreg[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[r * KP + s + hh * NS], doreg[s], dp, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int j = j0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      const bool inside = q_valid && j < len;
      const float pv = j < T ? __expf((inside ? st[i] : -1e9f) - m) * invl : 0.f;
      float dw = dp[i];
      if (dr.threshold) dw = hash_keep(dr, hrow, j) ? dw * dr.inv_keep : 0.f;
      st[i] = inside ? pv * (dw - D) * scale : 0.f;
    }
    // dQ^T[d][query] += K^T[d][key] dS^T[key][query]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
#pragma unroll
      for (int t = 0; t < DT; ++t)
        dqa[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[row * KP + 32 * t + r], st[i], dqa[t], 0, 0, 0);
    }
  }

  __syncthreads();
  constexpr int STRIP = 8 * KP;
  static_assert(NW <= 8, "two tiles hold eight strips");
  float* ws = (wave < 4 ? Ks : Vs) + (wave & 3) * STRIP;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int d = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if ((r >> 3) == pass) ws[(r & 7) * KP + d] = dqa[t][i];
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private strip: program order suffices
    for (int idx = lane; idx < 8 * DK; idx += 64) {
      const int ql = idx / DK, d = idx % DK;
      const int qq = blockIdx.x * (32 * NW) + wave * 32 + pass * 8 + ql;
      if (qq < T) dq[gbase + (long)qq * ldg + d] = (TIO)ws[ql * KP + d];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int DK, class TIO, int NW>
__global__ __launch_bounds__(64 * NW) void k_attention_bwd_dkv(const TIO* __restrict__ q, const TIO* __restrict__ k,
                                                               const TIO* __restrict__ v, const TIO* __restrict__ dctx,
                                                               const float* __restrict__ stats, const float* __restrict__ dsum,
                                                               const int* __restrict__ lengths, TIO* __restrict__ dk,
                                                               TIO* __restrict__ dv, int T, int H, int ld, int ldg,
                                                               float scale, AttnDropout dr) {
  hash_resolve_epoch(dr);
  constexpr int KP = DK + 1, DT = DK / 32, NS = DK / 2;
  __shared__ float Qs[kAttnKeys * KP];
  __shared__ float Gs[kAttnKeys * KP];  // dO tile
  __shared__ float Ms[kAttnKeys], Ls[kAttnKeys], Ds[kAttnKeys];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int C = H * DK;
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const long base = (long)b * T * C + (long)h * DK;
  const long ibase = (long)b * T * ld + (long)h * DK;
  const long gbase = (long)b * T * ldg + (long)h * DK;
  const long bh = (long)b * H + h;
  const int kj = blockIdx.x * (32 * NW) + wave * 32 + r;  // this lane's key
  const bool k_in = kj < T, k_valid = kj < len;

  float kreg[NS], vreg[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    kreg[s] = k_in ? attn_ld(k + ibase + (long)kj * ld + s + hh * NS) : 0.f;
    vreg[s] = k_in ? attn_ld(v + ibase + (long)kj * ld + s + hh * NS) : 0.f;
  }
  attn_f16v dka[DT], dva[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) dka[t][i] = 0.f, dva[t][i] = 0.f;

  constexpr int CH = DK / 4;
  constexpr int NV = (kAttnKeys * CH + 64 * NW - 1) / (64 * NW);
  attn_f4 qst[NV], gst[NV];
  float mst = 0.f, lst = 0.f, dst = 0.f;
  auto fetch = [&](int i0) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW), row = idx / CH, c = (idx % CH) * 4;
      if (idx >= kAttnKeys * CH) continue;
      const int qq = min(i0 + row, T - 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qst[n][e] = attn_ld(q + ibase + (long)qq * ld + c + e);
        gst[n][e] = attn_ld(dctx + base + (long)qq * C + c + e);
      }
    }
    if (tid < kAttnKeys) {  // rows past T get weight 0 through 1/l = 0
      const int qq = i0 + tid;
      // rows past T read a line of zeros: with the select on the address the three loads are unconditional and
      // stay in flight behind the tile's own prefetch (as selects on the value each sat in a branch with a wait
      // for EVERY outstanding load behind it, once per query tile)
      const float* zp = reinterpret_cast<const float*>(attn_zero_line);
      const float* sp = qq < T ? stats + 2 * (bh * T + qq) : zp;
      mst = sp[0];
      lst = sp[1];
      dst = *(qq < T ? dsum + bh * T + qq : zp);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW), row = idx / CH, c = (idx % CH) * 4;
      if (idx >= kAttnKeys * CH) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Qs[row * KP + c + e] = qst[n][e];
        Gs[row * KP + c + e] = gst[n][e];
      }
    }
    if (tid < kAttnKeys) {
      Ms[tid] = mst;
      Ls[tid] = lst;
      Ds[tid] = dst;
    }
  };

  fetch(0);
  for (int i0 = 0; i0 < T; i0 += kAttnKeys) {
    __syncthreads();
    stage();
    __syncthreads();
    if (i0 + kAttnKeys < T) fetch(i0 + kAttnKeys);

    // S[query r'][key] and dW[query r'][key]: lane supplies Q / dO of query r and its own K / V values
    attn_f16v st, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f, dp[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[r * KP + s + hh * NS], kreg[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[r * KP + s + hh * NS], vreg[s], dp, 0, 0, 0);
    }
    // register i of this lane: query i0 + rho_i + 4*hh, key kj
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh, qq = i0 + row;
      const bool inside = qq < len && k_valid;
      const float pv = __expf((inside ? st[i] * scale : -1e9f) - Ms[row]) * Ls[row];
      const bool keep = dr.threshold ? hash_keep(dr, hash_row(dr, (unsigned long long)bh * T + qq, (unsigned)(T + 1) >> 1), kj) : true;
      const float dw = keep ? dp[i] * dr.inv_keep : 0.f;
      st[i] = keep ? pv * dr.inv_keep : 0.f;                 // W[query][key]
      dp[i] = inside ? pv * (dw - Ds[row]) * scale : 0.f;    // dS[query][key] / sqrt(dk)
    }
    // dV^T[d][key] += dO^T[d][query] W[query][key];  dK^T[d][key] += Q^T[d][query] dS[query][key]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        dva[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[row * KP + 32 * t + r], st[i], dva[t], 0, 0, 0);
        dka[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[row * KP + 32 * t + r], dp[i], dka[t], 0, 0, 0);
      }
    }
  }

  __syncthreads();
  constexpr int STRIP = 8 * KP;
  static_assert(NW <= 8, "two tiles hold eight strips");
  float* ws = (wave < 4 ? Qs : Gs) + (wave & 3) * STRIP;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    TIO* out = which ? dv : dk;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int d = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if ((r >> 3) == pass) ws[(r & 7) * KP + d] = which ? dva[t][i] : dka[t][i];
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private strip: program order suffices
      for (int idx = lane; idx < 8 * DK; idx += 64) {
        const int kl = idx / DK, d = idx % DK;
        const int kk = blockIdx.x * (32 * NW) + wave * 32 + pass * 8 + kl;
        if (kk < T) out[gbase + (long)kk * ldg + d] = (TIO)ws[kl * KP + d];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// ---------------------------------------------------------------------------
// bf16 inputs/outputs: the same structure on v_mfma_f32_32x32x16_bf16 (fp32 scores, softmax
// statistics and accumulation; probabilities rounded to bf16 for the value contraction).
// A probability register i of lane half hh is key 16*(i>>3) + 4*hh + (i&3) + 8*((i>>2)&1) of the
// tile; the eight registers 8*half .. 8*half+7 of a lane form one B fragment as they stand, if the
// A fragment V^T[d][.] lists the keys in that order.  V is staged as it lies in HBM ([key][d], one
// 16-byte LDS write per lane) and the fragment is gathered by two ds_read_b64_tr_b16, whose lanes
// name the key rows they want (an earlier form scattered V^T into LDS with eight 2-byte writes per
// lane between the two barriers of every round).
// ---------------------------------------------------------------------------
typedef __bf16 attn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 attn_bf16x4 __attribute__((ext_vector_type(4)));

// TRAIN: dropout on the weights and saved row statistics as in k_attention_fwd, the maximum in the base-2
// domain this kernel works in (stats[..][0] = max of score * log2 e).
// Which (sequence tile x, head, utterance) a workgroup of the bf16 kernels takes.  The grid is (tiles, heads, batch)
// and workgroups are dispatched in the order of their linear index; read as tile-major, every FULL tile of every
// (utterance, head) pair is dispatched before the ragged last tiles (T = 400 on 256-wide tiles: 256 + 144), so the
// short workgroups are the ones that fill the last, partial round of the chip (the two backward kernels, which take
// one and a half rounds at batch 64 x 6 heads x 400 frames: 198 -> 192.6 us per layer; the forward's workgroups are
// all resident at once there and keep the grid's own order, which keeps a pair's two workgroups next to each other).
struct AttnBlock {
  int x, h, b;
};
template <bool TILE_MAJOR>
__device__ __forceinline__ AttnBlock attn_block() {
#ifdef LMUFFT_ATTN_PLAIN_ORDER  // (measurement switch: every kernel on the grid's own order)
  return AttnBlock{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
#endif
  if (!TILE_MAJOR) return AttnBlock{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned pairs = gridDim.y * gridDim.z;
  const unsigned x = lin / pairs, bh = lin - x * pairs;
  return AttnBlock{(int)x, (int)(bh % gridDim.y), (int)(bh / gridDim.y)};
}

template <int DK, int NW, bool TRAIN = false>
__global__ __launch_bounds__(64 * NW, DK <= 64 ? (TRAIN ? 3 : 4) : 1) void k_attention_fwd_bf16(const attn_bf16* __restrict__ q,
                                                                     const attn_bf16* __restrict__ k,
                                                                     const attn_bf16* __restrict__ v,
                                                                     const int* __restrict__ lengths,
                                                                     attn_bf16* __restrict__ ctx, int T, int H,
                                                                     int ld, float scale,
                                                                     AttnDropout dr = AttnDropout{0, 0, 0, 1.f},
                                                                     float* __restrict__ stats = nullptr) {
  hash_resolve_epoch(dr);
  static_assert(DK % 32 == 0 && DK <= 128, "head width: 32, 64, 96 or 128");
  constexpr int KP = DK + 8;          // K tile [key][d], 16-byte padded rows: conflict-free b128 fragments
  constexpr int NKT = 1;              // 32-key score tiles per round (2 was measured slower: 0.31 against 0.26 ms at
                                      // B=256, T=400 -- more registers and LDS per workgroup, fewer resident waves)
  constexpr int KEYS = 32 * NKT;
  constexpr int VP = DK <= 96 ? 96 : 160;  // V tile [key][d]; row pitch 192 or 320 bytes: the four key rows of a
                                           // transposing read fall on distinct bank quarters
  constexpr int DT = DK / 32, NS = DK / 16;
  constexpr int OP = DK + 1;          // fp32 output strip pitch
  __shared__ __attribute__((aligned(16))) attn_bf16 Ks[KEYS * KP];
  __shared__ __attribute__((aligned(16))) attn_bf16 Vs[KEYS * VP];
  __shared__ float Os[NW * 8 * OP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const AttnBlock blk = attn_block<false>();
  const int b = blk.b, h = blk.h, bx = blk.x;
  const int C = H * DK;
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const long base = (long)b * T * C + (long)h * DK;    // context rows are dense
  const long ibase = (long)b * T * ld + (long)h * DK;  // q, k, v rows are ld elements apart
  const int qi = bx * (32 * NW) + wave * 32 + r;
  const bool q_valid = qi < len;
  const int qc = min(qi, T - 1);
  const bool wave_live = bx * (32 * NW) + wave * 32 < T;  // wave-uniform

  attn_bf16x8 qf[NS];  // B fragments of the score contraction: Q[query r][16s + 8hh .. +7]
#pragma unroll
  for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const attn_bf16x8*>(q + ibase + (long)qc * ld + 16 * s + 8 * hh);

  attn_f16v o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float scale2 = scale * 1.4426950408889634f;
  constexpr float kMasked2 = -1e9f * 1.4426950408889634f;  // the reference's mask value, base-2 domain
  const bool all_q_valid = __all(q_valid);
  // transposing-read addressing: in each 16-lane group lane 4q+p names key row q, d columns 4p..4p+3
  const int tq = (lane & 15) >> 2, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  constexpr int CH = DK / 8;                         // 16-byte chunks per key row
  constexpr int NV = (KEYS * CH + (64 * NW) - 1) / (64 * NW);
  attn_bf16x8 kst[NV], vst[NV];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW);
      if (idx < KEYS * CH) {
        const int row = idx / CH, c = (idx % CH) * 8;
        const int j = min(j0 + row, T - 1);
        kst[n] = *reinterpret_cast<const attn_bf16x8*>(k + ibase + (long)j * ld + c);
        vst[n] = *reinterpret_cast<const attn_bf16x8*>(v + ibase + (long)j * ld + c);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW);
      if (idx < KEYS * CH) {
        const int row = idx / CH, c = (idx % CH) * 8;
        *reinterpret_cast<attn_bf16x8*>(&Ks[row * KP + c]) = kst[n];
        *reinterpret_cast<attn_bf16x8*>(&Vs[row * VP + c]) = vst[n];
      }
    }
  };

  [[maybe_unused]] const HashRow hrow = hash_row(dr, ((unsigned long long)b * H + h) * T + qi, (unsigned)(T + 1) >> 1);

  fetch(0);
  for (int j0 = 0; j0 < T; j0 += KEYS) {
    __syncthreads();
    stage();
    __syncthreads();
    if (j0 + KEYS < T) fetch(j0 + KEYS);
    if (!wave_live) continue;  // a wave whose 32 queries all lie past T only helps staging the tiles

    attn_f16v st[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) st[kt][i] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const attn_bf16x8 kf = *reinterpret_cast<const attn_bf16x8*>(&Ks[(32 * kt + r) * KP + 16 * s + 8 * hh]);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kt], 0, 0, 0);
      }
    }
    // Scores live in the base-2 domain (scale2 = scale * log2 e, the mask value likewise), so a
    // probability is one v_exp_f32.  The per-element mask tests run only on tiles that touch the
    // end of the utterance (or waves with queries past it): both conditions are wave-uniform.
    float mx = -INFINITY;
    if (j0 + KEYS <= len && all_q_valid) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          st[kt][i] *= scale2;
          mx = fmaxf(mx, st[kt][i]);
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int j = j0 + 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * hh;
          float sv = (q_valid && j < len) ? st[kt][i] * scale2 : kMasked2;
          sv = j < T ? sv : -INFINITY;
          st[kt][i] = sv;
          mx = fmaxf(mx, sv);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
    attn_bf16x8 pf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {  // registers i, i + 1: keys j, j + 1 (j even), one hash for the pair
        float pv0 = __builtin_amdgcn_exp2f(st[kt][i] - m_new);
        float pv1 = __builtin_amdgcn_exp2f(st[kt][i + 1] - m_new);
        psum += pv0 + pv1;
        if (TRAIN && dr.threshold) {  // the normaliser keeps every weight; the context only the surviving ones
          const int j = j0 + 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * hh;
          const uint32_t x = hash_pair(hrow, (unsigned)j >> 1);  // (the 1 / (1 - p) is applied with 1 / l at the end)
          pv0 = hash_keep_lo(dr, x) ? pv0 : 0.f;
          pv1 = hash_keep_hi(dr, x) ? pv1 : 0.f;
        }
        pf[kt][i >> 3][i & 7] = (attn_bf16)pv0;
        pf[kt][i >> 3][(i & 7) + 1] = (attn_bf16)pv1;
      }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    if (__any(m_new != m_run)) {  // otherwise alpha == 1 in every lane: nothing to rescale
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
    }
    m_run = m_new;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          // A fragment V^T[d = 32 t + r][.] in the key order of the probability registers: slots 0..3 are
          // keys 16 half + 4 hh + (0..3), slots 4..7 the same keys + 8 -- two transposing reads
          const int key0 = 32 * kt + 16 * half + 4 * hh + tq;
          typedef attn_bf16x4 __attribute__((address_space(3))) * lds4;
          const attn_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)&Vs[key0 * VP + 32 * t + tcol]);
          const attn_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)&Vs[(key0 + 8) * VP + 32 * t + tcol]);
          attn_bf16x8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vf[j] = lo[j];
            vf[4 + j] = hi[j];
          }
          o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][half], o[t], 0, 0, 0);
        }
  }

  float inv = 1.0f / l_run;
  if (TRAIN && hh == 0 && qi < T) {
    float* sp = stats + 2 * ((((long)b * H + h) * T) + qi);
    sp[0] = m_run;
    sp[1] = inv;
  }
  if (TRAIN) inv *= dr.inv_keep;
  float* ws = Os + wave * 8 * OP;  // wave-private strip of 8 queries
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int d = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if ((r >> 3) == pass) ws[(r & 7) * OP + d] = o[t][i] * inv;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int idx = lane; idx < 8 * DK; idx += 64) {
      const int ql = idx / DK, d = idx % DK;
      const int qq = bx * (32 * NW) + wave * 32 + pass * 8 + ql;
      if (qq < T) ctx[base + (long)qq * C + d] = (attn_bf16)ws[ql * OP + d];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// ---------------------------------------------------------------------------
// bf16 backward: the two sweeps of k_attention_bwd_dq / k_attention_bwd_dkv on v_mfma_f32_32x32x16_bf16
// (fp32 scores, statistics and accumulation; W and dS rounded to bf16 for the second contractions, as the
// forward rounds its probabilities).  A tile that serves both as the A operand of a score-type product
// (16-byte fragments along d) and, transposed, of an accumulation-type product (ds_read_b64_tr_b16) is
// kept in LDS in both pitches.
// ---------------------------------------------------------------------------
template <int DK>
__device__ __forceinline__ attn_bf16x8 attn_tr_frag(const attn_bf16* tile, int VP, int key0, int col) {
  typedef attn_bf16x4 __attribute__((address_space(3))) * lds4;
  const attn_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)&tile[key0 * VP + col]);
  const attn_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)&tile[(key0 + 8) * VP + col]);
  attn_bf16x8 f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[j] = lo[j];
    f[4 + j] = hi[j];
  }
  return f;
}

template <int DK, int NW>
__global__ __launch_bounds__(64 * NW, DK <= 64 ? 2 : 1) void k_attention_bwd_dq_bf16(const attn_bf16* __restrict__ q, const attn_bf16* __restrict__ k,
                                                                   const attn_bf16* __restrict__ v, const attn_bf16* __restrict__ ctx,
                                                                   const attn_bf16* __restrict__ dctx, const float* __restrict__ stats,
                                                                   const int* __restrict__ lengths, attn_bf16* __restrict__ dq,
                                                                   float* __restrict__ dsum, int T, int H, int ld, int ldg,
                                                                   float scale, AttnDropout dr) {
  hash_resolve_epoch(dr);
  constexpr int KP = DK + 8, VP = DK <= 96 ? 96 : 160, DT = DK / 32, NS = DK / 16, OP = DK + 1;
  __shared__ __attribute__((aligned(16))) attn_bf16 Ks[32 * KP];
  __shared__ __attribute__((aligned(16))) attn_bf16 Vs[32 * KP];
  __shared__ __attribute__((aligned(16))) attn_bf16 Kt[32 * VP];
  __shared__ float Os[NW * 8 * OP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const AttnBlock blk = attn_block<true>();
  const int b = blk.b, h = blk.h, bx = blk.x;
  const int C = H * DK;
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const long base = (long)b * T * C + (long)h * DK;
  const long ibase = (long)b * T * ld + (long)h * DK;
  const long gbase = (long)b * T * ldg + (long)h * DK;
  const long bh = (long)b * H + h;
  const int qi = bx * (32 * NW) + wave * 32 + r;
  const bool q_valid = qi < len;
  const int qc = min(qi, T - 1);
  const bool wave_live = bx * (32 * NW) + wave * 32 < T;  // wave-uniform

  attn_bf16x8 qf[NS], dof[NS];
  float D = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    qf[s] = *reinterpret_cast<const attn_bf16x8*>(q + ibase + (long)qc * ld + 16 * s + 8 * hh);
    dof[s] = *reinterpret_cast<const attn_bf16x8*>(dctx + base + (long)qc * C + 16 * s + 8 * hh);
    const attn_bf16x8 of = *reinterpret_cast<const attn_bf16x8*>(ctx + base + (long)qc * C + 16 * s + 8 * hh);
#pragma unroll
    for (int e = 0; e < 8; ++e) D += (float)dof[s][e] * (float)of[e];
  }
  D += __shfl_xor(D, 32);
  if (hh == 0 && qi < T) dsum[bh * T + qi] = D;
  // With W = keep p / (1 - p):  dS = p (dW - D) / sqrt(dk) = [p / (1 - p)] ([keep ? dP : 0] - (1 - p) D) / sqrt(dk):
  // the 1 / (1 - p) rides on 1 / l, D is scaled once, and 1 / sqrt(dk) is applied to the accumulated dQ.
  const float m2 = stats[2 * (bh * T + qc)], invl = stats[2 * (bh * T + qc) + 1] * dr.inv_keep;
  const float Dk = D / dr.inv_keep;
  const float scale2 = scale * 1.4426950408889634f;
  const int tq = (lane & 15) >> 2, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const bool all_q_valid = __all(q_valid);

  attn_f16v dqa[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) dqa[t][i] = 0.f;

  constexpr int CH = DK / 8;
  constexpr int NV = (32 * CH + (64 * NW) - 1) / (64 * NW);
  attn_bf16x8 kst[NV], vst[NV];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW);
      if (idx < 32 * CH) {
        const int row = idx / CH, c = (idx % CH) * 8;
        const int j = min(j0 + row, T - 1);
        kst[n] = *reinterpret_cast<const attn_bf16x8*>(k + ibase + (long)j * ld + c);
        vst[n] = *reinterpret_cast<const attn_bf16x8*>(v + ibase + (long)j * ld + c);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW);
      if (idx < 32 * CH) {
        const int row = idx / CH, c = (idx % CH) * 8;
        *reinterpret_cast<attn_bf16x8*>(&Ks[row * KP + c]) = kst[n];
        *reinterpret_cast<attn_bf16x8*>(&Vs[row * KP + c]) = vst[n];
        *reinterpret_cast<attn_bf16x8*>(&Kt[row * VP + c]) = kst[n];
      }
    }
  };
  const HashRow hrow = hash_row(dr, (unsigned long long)bh * T + qi, (unsigned)(T + 1) >> 1);

  fetch(0);
  for (int j0 = 0; j0 < T; j0 += 32) {
    __syncthreads();
    stage();
    __syncthreads();
    if (j0 + 32 < T) fetch(j0 + 32);
    if (!wave_live) continue;  // a wave whose 32 queries all lie past T only helps staging the tiles

    attn_f16v st, dp;
    [[maybe_unused]] uint32_t xpair = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f, dp[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const attn_bf16x8 kf = *reinterpret_cast<const attn_bf16x8*>(&Ks[r * KP + 16 * s + 8 * hh]);
      const attn_bf16x8 vf = *reinterpret_cast<const attn_bf16x8*>(&Vs[r * KP + 16 * s + 8 * hh]);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s], dp, 0, 0, 0);
    }
    attn_bf16x8 dsf[2];
    const bool interior = all_q_valid && j0 + 32 <= len;  // wave-uniform: no mask tests on such tiles
    // The sixteen elements of a lane, as straight-line code: the two wave-uniform switches (mask tests, dropout)
    // are compile-time flags of this block, chosen once per tile -- tested per element they cut the block into
    // some 150 basic blocks whose exp2 / hash chains ran one after the other.
    auto elements = [&](auto interior_c, auto drop_c) __attribute__((always_inline)) {
      constexpr bool INTERIOR = decltype(interior_c)::value, DROP = decltype(drop_c)::value;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int j = j0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        float dw = dp[i];
        if (DROP) {  // registers i, i + 1 (i even) are the two keys of one hash
          uint32_t x = 0;
          if ((i & 1) == 0) x = xpair = hash_pair(hrow, (unsigned)j >> 1);
          else x = xpair;
          dw = ((i & 1) ? hash_keep_hi(dr, x) : hash_keep_lo(dr, x)) ? dw : 0.f;
        }
        float ds;
        if (INTERIOR) {
          ds = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], scale2, -m2)) * invl * (dw - Dk);
        } else {
          const bool inside = q_valid && j < len;
          const float e = __builtin_amdgcn_exp2f(st[i] * scale2 - m2) * invl * (dw - Dk);
          ds = inside ? e : 0.f;
        }
        dsf[i >> 3][i & 7] = (attn_bf16)ds;
      }
    };
    if (interior) {
      if (dr.threshold) elements(std::true_type{}, std::true_type{});
      else elements(std::true_type{}, std::false_type{});
    } else {
      if (dr.threshold) elements(std::false_type{}, std::true_type{});
      else elements(std::false_type{}, std::false_type{});
    }
    // dQ^T[d][query] += K^T[d][key] dS^T[key][query]
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int t = 0; t < DT; ++t)
        dqa[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(attn_tr_frag<DK>(Kt, VP, 16 * half + 4 * hh + tq, 32 * t + tcol),
                                                         dsf[half], dqa[t], 0, 0, 0);
  }

  float* ws = Os + wave * 8 * OP;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int d = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if ((r >> 3) == pass) ws[(r & 7) * OP + d] = dqa[t][i] * scale;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int idx = lane; idx < 8 * DK; idx += 64) {
      const int ql = idx / DK, d = idx % DK;
      const int qq = bx * (32 * NW) + wave * 32 + pass * 8 + ql;
      if (qq < T) dq[gbase + (long)qq * ldg + d] = (attn_bf16)ws[ql * OP + d];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int DK, int NW>
__global__ __launch_bounds__(64 * NW, DK <= 64 ? 2 : 1) void k_attention_bwd_dkv_bf16(const attn_bf16* __restrict__ q, const attn_bf16* __restrict__ k,
                                                                    const attn_bf16* __restrict__ v, const attn_bf16* __restrict__ dctx,
                                                                    const float* __restrict__ stats, const float* __restrict__ dsum,
                                                                    const int* __restrict__ lengths, attn_bf16* __restrict__ dk,
                                                                    attn_bf16* __restrict__ dv, int T, int H, int ld, int ldg,
                                                                    float scale, AttnDropout dr) {
  hash_resolve_epoch(dr);
  constexpr int KP = DK + 8, VP = DK <= 96 ? 96 : 160, DT = DK / 32, NS = DK / 16, OP = DK + 1;
  __shared__ __attribute__((aligned(16))) attn_bf16 Qs[32 * KP];
  __shared__ __attribute__((aligned(16))) attn_bf16 Gs[32 * KP];
  __shared__ __attribute__((aligned(16))) attn_bf16 Qt[32 * VP];
  __shared__ __attribute__((aligned(16))) attn_bf16 Gt[32 * VP];
  __shared__ __attribute__((aligned(16))) float Ms[32], Ls[32], Ds[32];
  __shared__ __attribute__((aligned(8))) uint2 Hs[32];  // the dropout hash's per-query-row part, made once per tile
  __shared__ float Os[NW * 8 * OP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const AttnBlock blk = attn_block<true>();
  const int b = blk.b, h = blk.h, bx = blk.x;
  const int C = H * DK;
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  const long base = (long)b * T * C + (long)h * DK;
  const long ibase = (long)b * T * ld + (long)h * DK;
  const long gbase = (long)b * T * ldg + (long)h * DK;
  const long bh = (long)b * H + h;
  const int kj = bx * (32 * NW) + wave * 32 + r;
  const bool k_valid = kj < len;
  const bool wave_live = bx * (32 * NW) + wave * 32 < T;  // wave-uniform
  const int kc = min(kj, T - 1);

  attn_bf16x8 kf[NS], vf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    kf[s] = *reinterpret_cast<const attn_bf16x8*>(k + ibase + (long)kc * ld + 16 * s + 8 * hh);
    vf[s] = *reinterpret_cast<const attn_bf16x8*>(v + ibase + (long)kc * ld + 16 * s + 8 * hh);
  }
  const float scale2 = scale * 1.4426950408889634f;
  constexpr float kMasked2 = -1e9f * 1.4426950408889634f;
  const int tq = (lane & 15) >> 2, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  attn_f16v dka[DT], dva[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) dka[t][i] = 0.f, dva[t][i] = 0.f;

  constexpr int CH = DK / 8;
  constexpr int NV = (32 * CH + (64 * NW) - 1) / (64 * NW);
  attn_bf16x8 qst[NV], gst[NV];
  float mst = 0.f, lst = 0.f, dst = 0.f;
  auto fetch = [&](int i0) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW);
      if (idx < 32 * CH) {
        const int row = idx / CH, c = (idx % CH) * 8;
        const int qq = min(i0 + row, T - 1);
        qst[n] = *reinterpret_cast<const attn_bf16x8*>(q + ibase + (long)qq * ld + c);
        gst[n] = *reinterpret_cast<const attn_bf16x8*>(dctx + base + (long)qq * C + c);
      }
    }
    if (tid < 32) {  // rows past T get weight 0 through 1/l = 0
      const int qq = i0 + tid;
      // rows past T read a line of zeros: with the select on the address the three loads are unconditional and
      // stay in flight behind the tile's own prefetch (as selects on the value each sat in a branch with a wait
      // for EVERY outstanding load behind it, once per query tile)
      const float* zp = reinterpret_cast<const float*>(attn_zero_line);
      const float* sp = qq < T ? stats + 2 * (bh * T + qq) : zp;
      mst = sp[0];
      lst = sp[1];
      dst = *(qq < T ? dsum + bh * T + qq : zp);
    }
  };
  auto stage = [&](int i0) {
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int idx = tid + n * (64 * NW);
      if (idx < 32 * CH) {
        const int row = idx / CH, c = (idx % CH) * 8;
        *reinterpret_cast<attn_bf16x8*>(&Qs[row * KP + c]) = qst[n];
        *reinterpret_cast<attn_bf16x8*>(&Gs[row * KP + c]) = gst[n];
        *reinterpret_cast<attn_bf16x8*>(&Qt[row * VP + c]) = qst[n];
        *reinterpret_cast<attn_bf16x8*>(&Gt[row * VP + c]) = gst[n];
      }
    }
    if (tid < 32) {  // 1 / (1 - p) rides on 1 / l, D is scaled once (see k_attention_bwd_dq_bf16)
      Ms[tid] = mst;
      Ls[tid] = lst * dr.inv_keep;
      Ds[tid] = dst / dr.inv_keep;
      if (dr.threshold) {
        const HashRow hr = hash_row(dr, (unsigned long long)bh * T + (i0 + tid), (unsigned)(T + 1) >> 1);
        Hs[tid] = make_uint2(hr.lo, hr.c);
      }
    }
  };

  fetch(0);
  for (int i0 = 0; i0 < T; i0 += 32) {
    __syncthreads();
    stage(i0);
    __syncthreads();
    if (i0 + 32 < T) fetch(i0 + 32);
    if (!wave_live) continue;  // a wave whose 32 keys all lie past T only helps staging the tiles

    attn_f16v st, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f, dp[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const attn_bf16x8 qa = *reinterpret_cast<const attn_bf16x8*>(&Qs[r * KP + 16 * s + 8 * hh]);
      const attn_bf16x8 ga = *reinterpret_cast<const attn_bf16x8*>(&Gs[r * KP + 16 * s + 8 * hh]);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[s], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, vf[s], dp, 0, 0, 0);
    }
    attn_bf16x8 wf[2], dsf[2];
    // Dropout decisions: lanes r and r ^ 1 own the two keys of a pair and need the same 16 hashes (one per query
    // register); the even lane computes those of the even registers, the odd lane those of the odd ones, and a
    // lane swap within the pair delivers the rest.
    uint32_t xq[16];
    if (dr.threshold) {
      const int odd = r & 1;
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) {
        const int i = 2 * ii;  // this lane hashes register i + odd: query row rho(i) + odd
        const int row = (i & 3) + 8 * (i >> 2) + 4 * hh + odd;
        const uint2 hq = Hs[row];
        const uint32_t mine = hash_pair(HashRow{hq.x, hq.y}, (unsigned)kj >> 1);
        const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, false);  // lane ^ 1
        xq[i] = odd ? other : mine;
        xq[i + 1] = odd ? mine : other;
      }
    }
    // Straight-line code for the sixteen elements: dropout on or off is a compile-time flag of the block, chosen
    // once per tile (see k_attention_bwd_dq_bf16); the mask tests are two compares and selects on values for every
    // tile -- a second pair of specialised blocks for interior tiles cost more in registers (spills) than the
    // tests cost in instructions.
    auto elements = [&](auto drop_c) __attribute__((always_inline)) {
      constexpr bool DROP = decltype(drop_c)::value;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // registers 4g .. 4g + 3 are four consecutive query rows: their statistics come as three 16-byte reads
        const int row0 = 8 * g + 4 * hh;
        const attn_f4 m4 = *reinterpret_cast<const attn_f4*>(&Ms[row0]);
        const attn_f4 l4 = *reinterpret_cast<const attn_f4*>(&Ls[row0]);
        const attn_f4 d4 = *reinterpret_cast<const attn_f4*>(&Ds[row0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g + e;
          const bool keep = DROP ? ((kj & 1) ? hash_keep_hi(dr, xq[i]) : hash_keep_lo(dr, xq[i])) : true;
          const bool inside = i0 + row0 + e < len && k_valid;
          const float pv = __builtin_amdgcn_exp2f((inside ? st[i] * scale2 : kMasked2) - m4[e]) * l4[e];
          const float dw = keep ? dp[i] : 0.f;
          wf[i >> 3][i & 7] = (attn_bf16)(keep ? pv : 0.f);
          dsf[i >> 3][i & 7] = (attn_bf16)(inside ? pv * (dw - d4[e]) : 0.f);
        }
      }
    };
    if (dr.threshold) elements(std::true_type{});
    else elements(std::false_type{});
    // dV^T[d][key] += dO^T[d][query] W[query][key];  dK^T[d][key] += Q^T[d][query] dS[query][key]
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const int row0 = 16 * half + 4 * hh + tq, col = 32 * t + tcol;
        dva[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(attn_tr_frag<DK>(Gt, VP, row0, col), wf[half], dva[t], 0, 0, 0);
        dka[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(attn_tr_frag<DK>(Qt, VP, row0, col), dsf[half], dka[t], 0, 0, 0);
      }
  }

  float* ws = Os + wave * 8 * OP;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    attn_bf16* out = which ? dv : dk;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int d = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if ((r >> 3) == pass) ws[(r & 7) * OP + d] = which ? dva[t][i] : dka[t][i] * scale;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int idx = lane; idx < 8 * DK; idx += 64) {
        const int kl = idx / DK, d = idx % DK;
        const int kk = bx * (32 * NW) + wave * 32 + pass * 8 + kl;
        if (kk < T) out[gbase + (long)kk * ldg + d] = (attn_bf16)ws[kl * OP + d];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

}  // namespace lmufft

using namespace lmufft;

extern "C" {

int lmufft_attention_fwd(const void* q, const void* k, const void* v, int dtype, const int32_t* lengths, int32_t B,
                         int32_t T, int32_t H, int32_t DK, void* ctx, void* stream) {
  return lmufft_attention_fwd_ld(q, k, v, dtype, (int64_t)H * DK, lengths, B, T, H, DK, ctx, stream);
}

int lmufft_attention_fwd_ld(const void* q, const void* k, const void* v, int dtype, int64_t ld, const int32_t* lengths,
                            int32_t B, int32_t T, int32_t H, int32_t DK, void* ctx, void* stream) {
  if (B <= 0 || T <= 0 || H <= 0) return set_error("attention: bad shape (B=%d T=%d H=%d)", B, T, H);
  if (ld < (int64_t)H * DK || ld > 0x7fffffff) return set_error("attention: row stride %lld", (long long)ld);
  if (dtype == LMUFFT_BF16 && ld % 8 != 0) return set_error("attention: bf16 row stride must be a multiple of 8");
  if (DK != 32 && DK != 64 && DK != 128) return set_error("attention: head width %d; supported: 32, 64, 128", DK);
  if (dtype != LMUFFT_F32 && dtype != LMUFFT_BF16) return set_error("attention: unknown dtype %d", dtype);
  if (!q || !k || !v || !ctx) return set_error("attention: NULL pointer");
  if (H > 65535 || B > 65535) return set_error("attention: H and B must fit a grid dimension");
  // the bf16 kernel loads 16 bytes at a time (a head's columns start on multiples of 64 bytes)
  if (dtype == LMUFFT_BF16 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) != 0)
    return set_error("attention: bf16 tensors must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  // Waves (32 queries each) per workgroup: every workgroup sweeps all keys, so the sweeps should be
  // few and full.  T = 400 is 12.5 wave tiles: four waves per workgroup need four sweeps (the last
  // for 16 queries), eight need two -- bf16 0.20 -> 0.17 ms at B = 256, and 0.70 -> 0.63 ms at
  // B = 32, T = 3000.  Five or seven waves (fewer idle wave slots on paper) load the four SIMDs
  // unevenly and were slower than either.
  const int nw = (T + 31) / 32 <= 4 ? 4 : 8;
  const dim3 grid((T + 32 * nw - 1) / (32 * nw), H, B);
  const float scale = 1.0f / sqrtf((float)DK);
#define ATTN_LAUNCH_W(DKV, TY, NWV)                                                                              \
  hipLaunchKernelGGL((k_attention_fwd<DKV, TY, NWV>), grid, dim3(64 * NWV), 0, st, (const TY*)q, (const TY*)k,   \
                     (const TY*)v, lengths, (TY*)ctx, T, H, (int)ld, scale)
#define ATTN_LAUNCH(DKV, TY)                   \
  do {                                         \
    if (nw == 8) ATTN_LAUNCH_W(DKV, TY, 8);    \
    else ATTN_LAUNCH_W(DKV, TY, 4);            \
  } while (0)
  {
    ProfScope ps("k_attention_fwd", st);
    if (dtype == LMUFFT_F32) {
      if (DK == 32) ATTN_LAUNCH(32, float);
      else if (DK == 64) ATTN_LAUNCH(64, float);
      else ATTN_LAUNCH(128, float);
    } else {
#define ATTN_LAUNCH_BW(DKV, NWV)                                                                                  \
  hipLaunchKernelGGL((k_attention_fwd_bf16<DKV, NWV>), grid, dim3(64 * NWV), 0, st, (const attn_bf16*)q,         \
                     (const attn_bf16*)k, (const attn_bf16*)v, lengths, (attn_bf16*)ctx, T, H, (int)ld, scale)
#define ATTN_LAUNCH_B(DKV)                    \
  do {                                        \
    if (nw == 8) ATTN_LAUNCH_BW(DKV, 8);      \
    else ATTN_LAUNCH_BW(DKV, 4);              \
  } while (0)
      if (DK == 32) ATTN_LAUNCH_B(32);
      else if (DK == 64) ATTN_LAUNCH_B(64);
      else ATTN_LAUNCH_B(128);
#undef ATTN_LAUNCH_BW
#undef ATTN_LAUNCH_B
    }
  }
#undef ATTN_LAUNCH
#undef ATTN_LAUNCH_W
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_attention_fwd failed: %s", hipGetErrorString(e));
  return 0;
}

static AttnDropout attn_dropout(float p_drop, uint32_t seed0, uint32_t seed1) {
  return make_hash_dropout(p_drop, seed0, seed1);
}
// the bf16 MFMA kernels load 16 bytes at a time: row strides in multiples of 8 elements, 16-byte aligned tensors
static bool attn_bf16_fast(int64_t ld, int64_t ld2, const void* a, const void* b, const void* c, const void* d,
                           const void* e, const void* f) {
  const uintptr_t bits = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e | (uintptr_t)f;
  return ld % 8 == 0 && ld2 % 8 == 0 && (bits & 15) == 0;
}
static int attn_train_check(const char* what, int64_t ld, int64_t ld_grad, int32_t B, int32_t T, int32_t H, int32_t DK,
                            int dtype, float p_drop) {
  if (B <= 0 || T <= 0 || H <= 0) return set_error("%s: bad shape (B=%d T=%d H=%d)", what, B, T, H);
  if (ld < (int64_t)H * DK || ld > 0x7fffffff || ld_grad < (int64_t)H * DK || ld_grad > 0x7fffffff)
    return set_error("%s: row strides %lld / %lld", what, (long long)ld, (long long)ld_grad);
  if (DK != 32 && DK != 64 && DK != 128) return set_error("%s: head width %d; supported: 32, 64, 128", what, DK);
  if (dtype != LMUFFT_F32 && dtype != LMUFFT_BF16) return set_error("%s: unknown dtype %d", what, dtype);
  if (H > 65535 || B > 65535) return set_error("%s: H and B must fit a grid dimension", what);
  if (!(p_drop >= 0.f && p_drop < 1.f)) return set_error("%s: dropout probability %g outside [0, 1)", what, (double)p_drop);
  return 0;
}

int lmufft_attention_fwd_train(const void* q, const void* k, const void* v, int dtype, int64_t ld, const int32_t* lengths,
                               int32_t B, int32_t T, int32_t H, int32_t DK, float p_drop, uint32_t seed0, uint32_t seed1,
                               void* ctx, float* stats, void* stream) {
  if (int rc = attn_train_check("attention forward (training)", ld, ld, B, T, H, DK, dtype, p_drop)) return rc;
  if (!q || !k || !v || !ctx || !stats) return set_error("attention forward (training): NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  const AttnDropout dr = attn_dropout(p_drop, seed0, seed1);
  constexpr int NWV = 4;
  const dim3 grid((T + 32 * NWV - 1) / (32 * NWV), H, B);
  const dim3 grid8((T + 32 * 8 - 1) / (32 * 8), H, B);
  // eight waves per workgroup when the sequence is more than four wave tiles long, as in the inference forward
  // (batch 64 x 400 frames, bf16: 66.8 -> 62.5 us per layer)
  const bool fwd8 = (T + 31) / 32 > 4;
  const float scale = 1.0f / sqrtf((float)DK);
#define ATTN_TRAIN_FWD(DKV, TY)                                                                                   \
  hipLaunchKernelGGL((k_attention_fwd<DKV, TY, NWV, true>), grid, dim3(64 * NWV), 0, st, (const TY*)q, (const TY*)k, \
                     (const TY*)v, lengths, (TY*)ctx, T, H, (int)ld, scale, dr, stats)
  {
    ProfScope ps("k_attention_fwd_train", st);
    if (dtype == LMUFFT_F32) {
      if (DK == 32) ATTN_TRAIN_FWD(32, float);
      else if (DK == 64) ATTN_TRAIN_FWD(64, float);
      else ATTN_TRAIN_FWD(128, float);
    } else if (attn_bf16_fast(ld, ld, q, k, v, ctx, nullptr, nullptr)) {
#define ATTN_TRAIN_FWD_B(DKV)                                                                                      \
  if (fwd8)                                                                                                        \
    hipLaunchKernelGGL((k_attention_fwd_bf16<DKV, 8, true>), grid8, dim3(64 * 8), 0, st, (const attn_bf16*)q,      \
                       (const attn_bf16*)k, (const attn_bf16*)v, lengths, (attn_bf16*)ctx, T, H, (int)ld, scale, dr, stats); \
  else                                                                                                             \
  hipLaunchKernelGGL((k_attention_fwd_bf16<DKV, NWV, true>), grid, dim3(64 * NWV), 0, st, (const attn_bf16*)q,     \
                     (const attn_bf16*)k, (const attn_bf16*)v, lengths, (attn_bf16*)ctx, T, H, (int)ld, scale, dr, stats)
      if (DK == 32) ATTN_TRAIN_FWD_B(32);
      else if (DK == 64) ATTN_TRAIN_FWD_B(64);
      else ATTN_TRAIN_FWD_B(128);
#undef ATTN_TRAIN_FWD_B
    } else {
      if (DK == 32) ATTN_TRAIN_FWD(32, attn_bf16);
      else if (DK == 64) ATTN_TRAIN_FWD(64, attn_bf16);
      else ATTN_TRAIN_FWD(128, attn_bf16);
    }
  }
#undef ATTN_TRAIN_FWD
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_attention_fwd (training) failed: %s", hipGetErrorString(e));
  return 0;
}

int lmufft_attention_bwd(const void* q, const void* k, const void* v, const void* ctx, const void* dctx, const float* stats,
                         int dtype, int64_t ld, const int32_t* lengths, int32_t B, int32_t T, int32_t H, int32_t DK,
                         float p_drop, uint32_t seed0, uint32_t seed1, void* dq, void* dk, void* dv, int64_t ld_grad,
                         float* dsum, void* stream) {
  if (int rc = attn_train_check("attention backward", ld, ld_grad, B, T, H, DK, dtype, p_drop)) return rc;
  if (!q || !k || !v || !ctx || !dctx || !stats || !dq || !dk || !dv || !dsum) return set_error("attention backward: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  const AttnDropout dr = attn_dropout(p_drop, seed0, seed1);
  constexpr int NWV = 4;
  const dim3 grid((T + 32 * NWV - 1) / (32 * NWV), H, B);
  // The bf16 key-owning sweep runs eight waves per workgroup: every workgroup stages all query tiles, so twice the
  // keys per workgroup halve the staging and barriers per key (batch 64 x 400 frames: backward 222 -> 212 us per
  // layer, 32 x 3000: 3.53 -> 3.27 ms); the query-owning sweep measured slower that way (212 -> 223 us) and stays at four.
  constexpr int NWK = 8;
  const dim3 gridk((T + 32 * NWK - 1) / (32 * NWK), H, B);
  const float scale = 1.0f / sqrtf((float)DK);
#define ATTN_BWD(DKV, TY)                                                                                           \
  do {                                                                                                              \
    hipLaunchKernelGGL((k_attention_bwd_dq<DKV, TY, NWV>), grid, dim3(64 * NWV), 0, st, (const TY*)q, (const TY*)k,  \
                       (const TY*)v, (const TY*)ctx, (const TY*)dctx, stats, lengths, (TY*)dq, dsum, T, H, (int)ld,  \
                       (int)ld_grad, scale, dr);                                                                    \
    hipLaunchKernelGGL((k_attention_bwd_dkv<DKV, TY, NWV>), grid, dim3(64 * NWV), 0, st, (const TY*)q, (const TY*)k, \
                       (const TY*)v, (const TY*)dctx, stats, (const float*)dsum, lengths, (TY*)dk, (TY*)dv, T, H,    \
                       (int)ld, (int)ld_grad, scale, dr);                                                           \
  } while (0)
  {
    ProfScope ps("k_attention_bwd", st);
    if (dtype == LMUFFT_F32) {
      if (DK == 32) ATTN_BWD(32, float);
      else if (DK == 64) ATTN_BWD(64, float);
      else ATTN_BWD(128, float);
    } else if (attn_bf16_fast(ld, ld_grad, q, k, v, ctx, dctx, dq) && (((uintptr_t)dk | (uintptr_t)dv) & 15) == 0) {
#define ATTN_BWD_B(DKV)                                                                                              \
  do {                                                                                                               \
    typedef const attn_bf16* cb;                                                                                     \
    hipLaunchKernelGGL((k_attention_bwd_dq_bf16<DKV, NWV>), grid, dim3(64 * NWV), 0, st, (cb)q, (cb)k, (cb)v, (cb)ctx, \
                       (cb)dctx, stats, lengths, (attn_bf16*)dq, dsum, T, H, (int)ld, (int)ld_grad, scale, dr);      \
    hipLaunchKernelGGL((k_attention_bwd_dkv_bf16<DKV, NWK>), gridk, dim3(64 * NWK), 0, st, (cb)q, (cb)k, (cb)v,       \
                       (cb)dctx, stats, (const float*)dsum, lengths, (attn_bf16*)dk, (attn_bf16*)dv, T, H, (int)ld,   \
                       (int)ld_grad, scale, dr);                                                                     \
  } while (0)
      if (DK == 32) ATTN_BWD_B(32);
      else if (DK == 64) ATTN_BWD_B(64);
      else ATTN_BWD_B(128);
#undef ATTN_BWD_B
    } else {
      if (DK == 32) ATTN_BWD(32, attn_bf16);
      else if (DK == 64) ATTN_BWD(64, attn_bf16);
      else ATTN_BWD(128, attn_bf16);
    }
  }
#undef ATTN_BWD
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_attention_bwd failed: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"

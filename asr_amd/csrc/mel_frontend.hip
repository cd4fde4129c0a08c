// 80-bin log-mel frontend on gfx950: raw 16 kHz samples in HBM -> [B][T][80]
// features, without a round trip through the host.
//
// Reference arithmetic: AudioPreprocessor.extract_mel_features and
// _normalize_features, src/data/preprocessing.py:76-117, built on
// torchaudio.transforms.MelSpectrogram(16000, n_fft=400, hop_length=160,
// win_length=400, n_mels=80, power=2.0) and AmplitudeToDB() (:41-51):
//   frames of 400 samples every 160, centred (reflect padding of 200), periodic
//   Hann window, |DFT_400|^2 (201 bins), triangular HTK mel filters over
//   0..8000 Hz without normalisation, 10 log10(max(., 1e-10)), then per mel bin
//   (x - mean_t) / std_t with the unbiased std, std == 0 replaced by 1.
//
// k_mel_power: a 400-point real FFT per frame, eight lanes to a frame, in registers and LDS.  The
// op is HBM-bound by its definition (640 bytes of new samples in, 320 bytes out, ~10 kflop per
// frame); a dense DFT -- even on the matrix cores -- makes it compute-bound instead.
//   z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1], n = 0..199, is the windowed frame packed as 200 complex
//   numbers, Z = FFT_200(z), and the real transform's bins follow from pairs of Z:
//     E = (Z[k] + conj Z[200-k]) / 2, O = -i (Z[k] - conj Z[200-k]) / 2,
//     X[k] = E + e^{-2 pi i k / 400} O,  X[200-k] = conj(E - e^{-2 pi i k / 400} O).
//   With n = r + 8 m and k = k1 + 25 k2:
//     Z[k1 + 25 k2] = sum_r W_8^{r k2} ( W_200^{r k1} sum_m z[r + 8 m] W_25^{m k1} ).
//   1. lane r of a frame loads exactly the samples n = r (mod 8), windows them and takes their
//      25-point DFT in registers (two rounds of radix 5, constants for twiddles);
//   2. times W_200^{r k1}, out to LDS, the only exchange of the transform;
//   3. a lane then owns the columns k1 = u and 25 - u: two 8-point DFTs give Z[u + 25 k2] and
//      Z[(25 - u) + 25 k2], that is, both halves of eight pairs (k, 200 - k), so sixteen bins of
//      power come out of registers.  Column 0 pairs with itself shifted by one (Z[25 k2] with
//      Z[25 (8 - k2)]): its lane feeds the second DFT the same column times W_8^r, which shifts the
//      output by one bin.  Thirteen such units over eight lanes: two rounds.
//   4. power into the frame's LDS buffer; every mel filter is a short run of consecutive bins, lane
//      l takes the filters m = l (mod 8), 10 log10 of the weighted sum goes straight to HBM.
// The eight lanes of a frame sit in one wave and a wave's LDS operations execute in order, so the
// stages need no hardware barrier; the compiler, which reasons about aliasing one thread at a time,
// must still be told that other lanes' stores feed the next stage's loads (wave_sync()).
// One persistent workgroup of twelve waves per CU (the exchange buffers of its 96 frames fill the
// LDS); a wave takes eight consecutive frames of one utterance at a time.
//
// k_mel_norm: per (utterance, mel bin) mean / unbiased std over the valid frames, then the
// normalisation in place; utterances of up to 458 frames are held in LDS in between (one read,
// one write of HBM), longer ones take three passes over HBM.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <mutex>
#include <vector>

#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

constexpr int kMelFft = 400, kMelHop = 160, kMelBins = 201, kMelN = 80, kMelPad = 200;
constexpr int kMelHalf = kMelFft / 2;      // points of the packed complex transform
constexpr int kMelWaves = 12;              // waves per workgroup, eight frames each
constexpr int kMelThreads = 64 * kMelWaves;
constexpr int kMelFrames = 8 * kMelWaves;  // frames in flight per workgroup
constexpr int kMelZPitch = 2 * kMelHalf;   // floats per frame buffer
// Filters 8 i .. 8 i + 7 are evaluated together over kMelRun[i] bins, the longest support among them
// (shorter ones are padded with zero weights).  The supports follow from the fixed configuration
// (16 kHz, 400-point DFT, 80 HTK filters up to 8 kHz) and are checked when the tables are built.
constexpr int kMelRun[kMelN / 8] = {2, 2, 2, 3, 4, 5, 6, 8, 11, 14};
constexpr int kMelRunTotal = 57;

// constants in HBM (lmufft_melprep), copied to LDS by every workgroup
constexpr size_t kPrepWin = 0;                                          // float2[200]: half the periodic Hann window, in pairs
constexpr size_t kPrepTw = kPrepWin + sizeof(float) * kMelFft;          // float2[8][25]: e^{-2 pi i r k1 / 200}
constexpr size_t kPrepTab = kPrepTw + sizeof(float) * 2 * kMelHalf;     // float2[200]: e^{-2 pi i k / 400}
constexpr size_t kPrepFw = kPrepTab + sizeof(float) * 2 * kMelHalf;     // float[57][8]: filter weights, run position major
constexpr size_t kPrepFs = kPrepFw + sizeof(float) * kMelRunTotal * 8;  // int[80]: first bin of every filter
constexpr size_t kPrepBytes = kPrepFs + sizeof(int) * kMelN;
constexpr int kPrepFloats = (int)(kPrepBytes / 4);
constexpr size_t kMelLds = kPrepBytes + sizeof(float) * (size_t)kMelFrames * kMelZPitch;
static_assert(kMelLds <= 160 * 1024, "LDS budget");

// a complex number as a two-lane vector, so that sums and products map onto the packed fp32 instructions
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf mk(float re, float im) {
  cf v;
  v.x = re;
  v.y = im;
  return v;
}
__device__ __forceinline__ cf cmul(cf a, cf b) { return a.xx * b + a.yy * mk(-b.y, b.x); }
__device__ __forceinline__ cf mulmi(cf a) { return mk(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ cf conj(cf a) { return mk(a.x, -a.y); }
__device__ __forceinline__ cf scale(cf a, float s) { return a * s; }
__device__ __forceinline__ float norm2(cf a) {
  const cf q = a * a;
  return q.x + q.y;
}

__device__ __forceinline__ void dft5(cf (&x)[5]) {
  constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;  // cos(2 pi / 5), cos(4 pi / 5)
  constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;   // sin(2 pi / 5), sin(4 pi / 5)
  const cf t1 = x[1] + x[4], t2 = x[2] + x[3], t3 = x[1] - x[4], t4 = x[2] - x[3];
  const cf m1 = x[0] + scale(t1, c1) + scale(t2, c2), m2 = x[0] + scale(t1, c2) + scale(t2, c1);
  const cf q1 = mulmi(scale(t3, s1) + scale(t4, s2)), q2 = mulmi(scale(t3, s2) - scale(t4, s1));
  x[0] = x[0] + t1 + t2;
  x[1] = m1 + q1;
  x[2] = m2 + q2;
  x[3] = m2 - q2;
  x[4] = m1 - q1;
}

__device__ __forceinline__ void dft8(cf (&x)[8]) {
  constexpr float h = 0.70710678118654752f;
  const cf a0 = x[0] + x[4], a1 = x[0] - x[4], a2 = x[2] + x[6], a3 = mulmi(x[2] - x[6]);
  const cf a4 = x[1] + x[5], a5 = x[1] - x[5], a6 = x[3] + x[7], a7 = mulmi(x[3] - x[7]);
  const cf b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3, b3 = a1 - a3;
  const cf b4 = a4 + a6, b6 = mulmi(a4 - a6);
  const cf b5 = cmul(a5 + a7, mk(h, -h)), b7 = cmul(a5 - a7, mk(-h, -h));
  x[0] = b0 + b4;
  x[1] = b1 + b5;
  x[2] = b2 + b6;
  x[3] = b3 + b7;
  x[4] = b0 - b4;
  x[5] = b1 - b5;
  x[6] = b2 - b6;
  x[7] = b3 - b7;
}

// cos, sin of 2 pi e / 25
__device__ constexpr float kCos25[17] = {1.0f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899655f, 0.30901699437494745f, 0.062790519529313527f, -0.1873813145857246f, -0.42577929156507272f, -0.63742398974868975f, -0.80901699437494734f, -0.92977648588825135f, -0.99211470131447776f, -0.99211470131447788f, -0.92977648588825146f, -0.80901699437494778f, -0.63742398974868952f};
__device__ constexpr float kSin25[17] = {0.0f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868862f, 0.84432792550201508f, 0.95105651629515353f, 0.99802672842827156f, 0.98228725072868872f, 0.90482705246601947f, 0.77051324277578925f, 0.58778525229247325f, 0.36812455268467814f, 0.12533323356430454f, -0.12533323356430429f, -0.36812455268467792f, -0.58778525229247269f, -0.77051324277578936f};

// X[q0 + 5 q1] = sum over m = 5 m1 + m0 of x[m] W_25^{m (q0 + 5 q1)}, in place
__device__ __forceinline__ void dft25(cf (&x)[25]) {
#pragma unroll
  for (int m0 = 0; m0 < 5; ++m0) {
    cf y[5] = {x[m0], x[m0 + 5], x[m0 + 10], x[m0 + 15], x[m0 + 20]};
    dft5(y);
#pragma unroll
    for (int q0 = 0; q0 < 5; ++q0)
      x[m0 + 5 * q0] = m0 * q0 ? cmul(y[q0], mk(kCos25[m0 * q0], -kSin25[m0 * q0])) : y[q0];  // slot [q0][m0]
  }
#pragma unroll
  for (int q0 = 0; q0 < 5; ++q0) {
    cf y[5] = {x[5 * q0], x[5 * q0 + 1], x[5 * q0 + 2], x[5 * q0 + 3], x[5 * q0 + 4]};
    dft5(y);
#pragma unroll
    for (int q1 = 0; q1 < 5; ++q1) x[5 * q0 + q1] = y[q1];  // slot [q0][q1] holds X[q0 + 5 q1]
  }
}
// position of X[k1] in the array dft25() leaves behind
__device__ __host__ constexpr int dft25_slot(int k1) { return 5 * (k1 % 5) + k1 / 5; }

// Columns a (k1 = u) and b (k1 = 25 - u, or column 0 times W_8^r for u = 0) of one frame ->
// power of the bins u + 25 k2 (pp) and 200 - u - 25 k2 (pm).  w[k2] = e^{-2 pi i (u + 25 k2) / 400}.
__device__ __forceinline__ void bins16(cf (&a)[8], cf (&b)[8], const cf (&w)[8], float (&pp)[8], float (&pm)[8]) {
  dft8(a);
  dft8(b);
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) {
    const cf bc = conj(b[7 - k2]);
    const cf e = a[k2] + bc, o = cmul(mulmi(a[k2] - bc), w[k2]);
    const cf xp = e + o, xm = e - o;
    pp[k2] = norm2(xp);
    pm[k2] = norm2(xm);
  }
}

// orders this wave's LDS traffic for the compiler; emits no instruction
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The 25 pairs of samples n = l8 (mod 8) of frame t of utterance b (centred, reflect-padded; frames
// past the utterance's end read clamped data and are masked by the caller), all loads in flight at once.
__device__ __forceinline__ void mel_load(cf (&x)[25], const float* __restrict__ wb, int len, int t0, int t, int l8,
                                         int pairs_aligned) {
  const int s00 = t * kMelHop - kMelPad + 2 * l8;
  if (t0 * kMelHop - kMelPad >= 0 && (t0 + 7) * kMelHop + kMelPad <= len) {  // no edge in any of the wave's eight frames
    if (pairs_aligned) {
      const float2* p2 = reinterpret_cast<const float2*>(wb + s00);
#pragma unroll
      for (int i = 0; i < 25; ++i) {
        const float2 v = p2[8 * i];
        x[i] = mk(v.x, v.y);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 25; ++i) x[i] = mk(wb[s00 + 16 * i], wb[s00 + 16 * i + 1]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 25; ++i) {
      int sa = s00 + 16 * i, sb = sa + 1;
      sa = sa < 0 ? -sa : sa;
      sb = sb < 0 ? -sb : sb;
      sa = sa >= len ? 2 * (len - 1) - sa : sa;
      sb = sb >= len ? 2 * (len - 1) - sb : sb;
      x[i] = mk(wb[min(max(sa, 0), len - 1)], wb[min(max(sb, 0), len - 1)]);
    }
  }
}

__global__ __launch_bounds__(kMelThreads) void k_mel_power(const float* __restrict__ wave, const int* __restrict__ lengths,
                                                           long ld_wave, int L, float* __restrict__ mel, int Tmax,
                                                           const float* __restrict__ prep, int groups_per_utt, int ngroups,
                                                           int pairs_aligned) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const cf* win2 = reinterpret_cast<const cf*>(lds + kPrepWin / 4);
  const cf* tw = reinterpret_cast<const cf*>(lds + kPrepTw / 4);
  const cf* tab = reinterpret_cast<const cf*>(lds + kPrepTab / 4);
  const float* fw = lds + kPrepFw / 4;
  const int* fstart = reinterpret_cast<const int*>(lds + kPrepFs / 4);
  const int tid = threadIdx.x, l8 = tid & 7, fl = tid >> 3;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  cf* z = reinterpret_cast<cf*>(lds + kPrepFloats + fl * kMelZPitch);
  // workgroups that share an XCD (blockIdx mod 8) take neighbouring groups of frames, whose samples overlap
  const int nb = gridDim.x, bx = blockIdx.x;
  const int per = nb >> 3, rem = nb & 7, xcd = bx & 7, idx = bx >> 3;
  const int wg = xcd * per + min(xcd, rem) + idx;
  for (int i = tid; i < kPrepFloats; i += kMelThreads) lds[i] = prep[i];
  __syncthreads();
  for (int g = wg + nb * wv; g < ngroups; g += nb * kMelWaves) {
    const int b = g / groups_per_utt, t0 = (g - b * groups_per_utt) * 8;
    const int t = t0 + (fl & 7);
    const int len = min(max(lengths ? lengths[b] : L, 1), L), Tb = 1 + len / kMelHop;
    float* out = mel + ((long)b * Tmax + t) * kMelN;
    cf x[25];
    mel_load(x, wave + (long)b * ld_wave, len, t0, t, l8, pairs_aligned);
    // 1. window, 25-point transform of the lane's residue class
#pragma unroll
    for (int i = 0; i < 25; ++i) {
      x[i] = x[i] * win2[l8 + 8 * i];
    }
    dft25(x);
    // 2. twiddle and exchange: lane r leaves W_200^{r k1} X_r[k1] at z[25 r + k1]
    wave_sync();  // the previous group's filters have read the buffer
#pragma unroll
    for (int k1 = 0; k1 < 25; ++k1) z[25 * l8 + k1] = k1 ? cmul(x[dft25_slot(k1)], tw[25 * l8 + k1]) : x[0];
    wave_sync();
    // 3. thirteen units (column 0; columns u and 25 - u) over eight lanes
    float pw[2][16];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int u = min(l8 + 8 * it, 12);
      const int cb = u ? 25 - u : 0;
      cf a[8], bb[8], w[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        a[r] = z[25 * r + u];
        bb[r] = z[25 * r + cb];
        w[r] = tab[u + 25 * r];
      }
      if (it == 0) {  // unit 0 sits on lane 0 of round 0
        constexpr float h = 0.70710678118654752f;
        constexpr float c8[8] = {1.f, h, 0.f, -h, -1.f, -h, 0.f, h}, s8[8] = {0.f, h, 1.f, h, 0.f, -h, -1.f, -h};
#pragma unroll
        for (int r = 1; r < 8; ++r) bb[r] = cmul(bb[r], mk(u ? 1.f : c8[r], u ? 0.f : -s8[r]));
      }
      float pp[8], pm[8];
      bins16(a, bb, w, pp, pm);
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        pw[it][k2] = pp[k2];
        pw[it][8 + k2] = pm[k2];
      }
    }
    wave_sync();  // every lane has read its columns
    float* P = reinterpret_cast<float*>(z);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int u = l8 + 8 * it;
      if (u < 13) {
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
          P[u + 25 * k2] = pw[it][k2];
          P[200 - u - 25 * k2] = pw[it][8 + k2];
        }
      }
    }
    wave_sync();
    // 4. filters m = l8 + 8 i: 10 log10 of the weighted sum of a short run of bins
    if (t < Tmax) {
      int run0 = 0;
#pragma unroll
      for (int i = 0; i < kMelN / 8; ++i) {
        const int m = l8 + 8 * i;
        const float* Pm = P + fstart[m];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < kMelRun[i]; ++c) s += fw[(run0 + c) * 8 + l8] * Pm[c];
        run0 += kMelRun[i];
        out[m] = t < Tb ? 3.01029995663981195f * __log2f(fmaxf(s, 1e-10f)) : 0.f;
      }
    }
  }
}

// grid (5, B): 16 mel bins per block; thread = (bin c, row group g of kMelNormGroups): 1024 threads, so that a
// 30 s utterance (3000 frames) is 47 rows per thread and pass rather than 188
constexpr int kMelNormGroups = 64;
__global__ __launch_bounds__(16 * kMelNormGroups) void k_mel_norm(float* __restrict__ mel, const int* __restrict__ lengths, int L,
                                                  int Tmax) {
  __shared__ float red[kMelNormGroups][17];
  __shared__ float stat[2][16];
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int b = blockIdx.y, c0 = blockIdx.x * 16;
  int len = lengths ? lengths[b] : L;
  len = min(max(len, 1), L);
  const int Tb = min(1 + len / kMelHop, Tmax);
  float* mb = mel + (long)b * Tmax * kMelN + c0 + c;
  float s = 0.f;
  for (int t = g; t < Tb; t += kMelNormGroups) s += mb[(long)t * kMelN];
  red[g][c] = s;
  __syncthreads();
  if (g == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < kMelNormGroups; ++i) tot += red[i][c];
    stat[0][c] = tot / (float)Tb;
  }
  __syncthreads();
  const float mean = stat[0][c];
  float q = 0.f;
  for (int t = g; t < Tb; t += kMelNormGroups) {
    const float d = mb[(long)t * kMelN] - mean;
    q += d * d;
  }
  __syncthreads();
  red[g][c] = q;
  __syncthreads();
  if (g == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < kMelNormGroups; ++i) tot += red[i][c];
    float sd = sqrtf(tot / (float)(Tb - 1));  // unbiased, as torch.std (NaN for a single frame, as there)
    stat[1][c] = sd == 0.f ? 1.f : sd;
  }
  __syncthreads();
  const float sd = stat[1][c];
  for (int t = g; t < Tb; t += kMelNormGroups) mb[(long)t * kMelN] = (mb[(long)t * kMelN] - mean) / sd;
}

// One workgroup per utterance, its valid frames held in LDS between the passes: thread = (group of four
// bins q, row group rg), 51 row groups.  Sums run over rows rg, rg + 51, ... and then over the row groups
// in a fixed order.
constexpr int kMelNormRg = 51;
constexpr int kMelNormLdsRows = 458;  // 458 * 320 + 51 * 320 + 640 bytes <= 160 KB
__global__ __launch_bounds__(1024) void k_mel_norm_lds(float* __restrict__ mel, const int* __restrict__ lengths, int L, int Tmax) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x, tid = threadIdx.x;
  int len = lengths ? lengths[b] : L;
  len = min(max(len, 1), L);
  const int Tb = min(1 + len / kMelHop, Tmax);
  float4* tile = reinterpret_cast<float4*>(lds);          // [Tb][20]
  float4* red = tile + (size_t)Tb * 20;                   // [51][20]
  float* stat = reinterpret_cast<float*>(red + kMelNormRg * 20);  // mean[80], std[80]
  float4* mb = reinterpret_cast<float4*>(mel + (long)b * Tmax * kMelN);
  const int q = tid % 20, rg = tid / 20;
  const bool on = rg < kMelNormRg;
  float4 s = {0.f, 0.f, 0.f, 0.f};
  if (on)
    for (int r = rg; r < Tb; r += kMelNormRg) {
      const float4 v = mb[r * 20 + q];
      tile[r * 20 + q] = v;
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  if (on) red[rg * 20 + q] = s;
  __syncthreads();
  if (tid < kMelN) {
    const float* rf = reinterpret_cast<const float*>(red);
    float tot = 0.f;
    for (int i = 0; i < kMelNormRg; ++i) tot += rf[i * kMelN + tid];
    stat[tid] = tot / (float)Tb;
  }
  __syncthreads();
  const float4 mean = on ? reinterpret_cast<const float4*>(stat)[q] : s;
  s = {0.f, 0.f, 0.f, 0.f};
  if (on) {
    for (int r = rg; r < Tb; r += kMelNormRg) {
      const float4 v = tile[r * 20 + q];
      const float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
      s.x += dx * dx, s.y += dy * dy, s.z += dz * dz, s.w += dw * dw;
    }
    red[rg * 20 + q] = s;
  }
  __syncthreads();
  if (tid < kMelN) {
    const float* rf = reinterpret_cast<const float*>(red);
    float tot = 0.f;
    for (int i = 0; i < kMelNormRg; ++i) tot += rf[i * kMelN + tid];
    const float sd = sqrtf(tot / (float)(Tb - 1));  // unbiased, as torch.std (NaN for a single frame, as there)
    stat[kMelN + tid] = sd == 0.f ? 1.f : sd;
  }
  __syncthreads();
  if (on) {
    const float4 sd = reinterpret_cast<const float4*>(stat + kMelN)[q];
    for (int r = rg; r < Tb; r += kMelNormRg) {
      const float4 v = tile[r * 20 + q];
      mb[r * 20 + q] = float4{(v.x - mean.x) / sd.x, (v.y - mean.y) / sd.y, (v.z - mean.z) / sd.z, (v.w - mean.w) / sd.w};
    }
  }
}

// ---------------------------------------------------------------------------
// constants, computed once on the host in double precision
// ---------------------------------------------------------------------------
struct MelConstants {
  std::vector<char> bytes;
  bool ok = false;  // the filters' supports fit kMelRun
};

static const MelConstants& mel_constants() {
  static MelConstants mc;
  static std::once_flag once;
  std::call_once(once, [] {
    mc.bytes.assign(kPrepBytes, 0);
    float* win = reinterpret_cast<float*>(mc.bytes.data() + kPrepWin);  // half the periodic Hann window
    for (int n = 0; n < kMelFft; ++n) win[n] = (float)(0.5 * (0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)kMelFft)));
    float* tw = reinterpret_cast<float*>(mc.bytes.data() + kPrepTw);
    for (int r = 0; r < 8; ++r)
      for (int k1 = 0; k1 < 25; ++k1) {
        const double a = 2.0 * M_PI * (double)(r * k1) / (double)kMelHalf;
        tw[2 * (25 * r + k1)] = (float)cos(a);
        tw[2 * (25 * r + k1) + 1] = (float)-sin(a);
      }
    float* tab = reinterpret_cast<float*>(mc.bytes.data() + kPrepTab);
    for (int k = 0; k < kMelHalf; ++k) {
      const double a = 2.0 * M_PI * (double)k / (double)kMelFft;
      tab[2 * k] = (float)cos(a);
      tab[2 * k + 1] = (float)-sin(a);
    }
    // HTK mel filters, 82 points equally spaced in mel over 0..8000 Hz, bin centres k * 40 Hz
    auto hz2mel = [](double f) { return 2595.0 * log10(1.0 + f / 700.0); };
    auto mel2hz = [](double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); };
    const double mmin = hz2mel(0.0), mmax = hz2mel(8000.0);
    std::vector<double> fpts(kMelN + 2);
    for (int i = 0; i < kMelN + 2; ++i) fpts[i] = mel2hz(mmin + (mmax - mmin) * (double)i / (double)(kMelN + 1));
    auto fb = [&](int k, int m) -> double {
      const double f = 8000.0 * (double)k / (double)(kMelBins - 1);
      const double down = (f - fpts[m]) / (fpts[m + 1] - fpts[m]);
      const double up = (fpts[m + 2] - f) / (fpts[m + 2] - fpts[m + 1]);
      const double v = down < up ? down : up;
      return v > 0.0 ? v : 0.0;
    };
    int* fs = reinterpret_cast<int*>(mc.bytes.data() + kPrepFs);
    float* fwt = reinterpret_cast<float*>(mc.bytes.data() + kPrepFw);
    int run0 = 0;
    for (int i = 0; i < kMelN / 8; ++i) {
      for (int l = 0; l < 8; ++l) {
        const int m = 8 * i + l;
        int first = -1, last = -1;
        for (int k = 0; k < kMelBins; ++k)
          if ((float)fb(k, m) != 0.f) {
            if (first < 0) first = k;
            last = k;
          }
        if (first < 0) first = last = 0;  // a filter narrower than the bin spacing: weight 0 on one bin
        if (last - first + 1 > kMelRun[i]) {
          mc.ok = false;
          return;
        }
        fs[m] = first;
        for (int c = 0; c < kMelRun[i]; ++c) fwt[(run0 + c) * 8 + l] = first + c <= last ? (float)fb(first + c, m) : 0.f;
      }
      run0 += kMelRun[i];
    }
    mc.ok = run0 == kMelRunTotal;
  });
  return mc;
}

}  // namespace lmufft

using namespace lmufft;

extern "C" {

size_t lmufft_melprep_bytes(void) { return kPrepBytes; }

int lmufft_melprep(void* melprep, void* stream) {
  if (!melprep) return set_error("melprep: NULL pointer");
  const MelConstants& mc = mel_constants();
  if (!mc.ok) return set_error("melprep: the mel filters' supports do not fit the compiled run lengths");
  const hipError_t e = hipMemcpyAsync(melprep, mc.bytes.data(), kPrepBytes, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return set_error("melprep: copy failed: %s", hipGetErrorString(e));
  return 0;
}

int lmufft_mel_frames(int32_t samples) { return samples < 0 ? 0 : 1 + samples / kMelHop; }

int lmufft_mel_forward_f32(const float* wave, const int32_t* lengths, int32_t B, int64_t ld_wave, int32_t L,
                           float* mel, int32_t Tmax, int normalize, const void* melprep, void* stream) {
  if (B <= 0 || L <= 0 || Tmax <= 0 || ld_wave < L) return set_error("mel: bad shape (B=%d, L=%d, Tmax=%d)", B, L, Tmax);
  if (L <= kMelPad) return set_error("mel: reflect padding needs more than %d samples, got %d", kMelPad, L);
  if (Tmax < 1 + L / kMelHop)
    return set_error("mel: %d samples give %d frames but the output holds %d", L, 1 + L / kMelHop, Tmax);
  if (!wave || !mel || !melprep) return set_error("mel: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      cus <= 0)
    return set_error("mel: cannot query the device");
  if (hipFuncSetAttribute((const void*)k_mel_power, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMelLds) != hipSuccess)
    return set_error("cannot reserve %zu bytes of LDS for k_mel_power", kMelLds);
  const int groups_per_utt = (Tmax + 7) / 8;
  const long ngroups = (long)groups_per_utt * B;
  if (ngroups > INT32_MAX) return set_error("mel: %ld groups of frames exceed the index range", ngroups);
  const int grid = (int)(ngroups < cus ? ngroups : cus);
  const int pairs_aligned = ((uintptr_t)wave % 8 == 0) && (ld_wave % 2 == 0);
  {
    ProfScope ps("k_mel_power", st);
    hipLaunchKernelGGL(k_mel_power, dim3(grid), dim3(kMelThreads), kMelLds, st, wave, lengths, (long)ld_wave, L, mel, Tmax,
                       (const float*)melprep, groups_per_utt, (int)ngroups, pairs_aligned);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_mel_power failed: %s", hipGetErrorString(e));
  if (normalize) {
    // one workgroup per utterance pays off once there are enough utterances to fill a good part of the chip
    if (Tmax <= kMelNormLdsRows && (uintptr_t)mel % 16 == 0 && 4 * B >= cus) {
      const size_t nlds = sizeof(float) * ((size_t)(Tmax + kMelNormRg) * kMelN + 2 * kMelN);
      if (hipFuncSetAttribute((const void*)k_mel_norm_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nlds) != hipSuccess)
        return set_error("cannot reserve %zu bytes of LDS for k_mel_norm", nlds);
      ProfScope ps("k_mel_norm", st);
      hipLaunchKernelGGL(k_mel_norm_lds, dim3(B), dim3(1024), nlds, st, mel, lengths, L, Tmax);
    } else {
      ProfScope ps("k_mel_norm", st);
      hipLaunchKernelGGL(k_mel_norm, dim3(kMelN / 16, B), dim3(16 * kMelNormGroups), 0, st, mel, lengths, L, Tmax);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return set_error("launch of k_mel_norm failed: %s", hipGetErrorString(e));
  }
  return 0;
}

}  // extern "C"

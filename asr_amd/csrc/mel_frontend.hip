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
// k_mel_power: the windowed 400-point real DFT as fp32 MFMA contractions with
// both symmetries of the real transform folded in.  With y = window * frame,
//   e[n] = y[n] + y[400-n], o[n] = y[n] - y[400-n]   (n = 0..200),
//   Re X[k]      =  Ee[k] + Eo[k],   Im X[k]      = -(Oe[k] + Oo[k]),
//   Re X[200-k]  =  Ee[k] - Eo[k],   Im X[200-k]  =   Oe[k] - Oo[k],
// where Ee / Eo sum e[n] cos(2 pi k n / 400) over even / odd n and Oe / Oo sum
// o[n] sin(.) likewise: four products with K ~ 100 and 101 output bins instead
// of two with K = 400 and 201 bins.  The MFMA output layout (lane = frame,
// registers = bins) is exactly the B-operand layout of the next contraction, so
// the power spectrum goes into the mel filterbank product straight from
// registers.  One wave owns 32 frames; the samples of a block's 128 frames sit
// in LDS once (hop chunks with an odd pitch: lanes = frames hit distinct banks).
//
// k_mel_norm: per (utterance, mel bin) mean / unbiased std over the valid
// frames in two passes, then the normalisation in place.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <mutex>
#include <vector>

#include "lmufft.h"
#include "lmufft_internal.h"

namespace lmufft {

typedef float f32x16m __attribute__((ext_vector_type(16)));

constexpr int kMelFft = 400, kMelHop = 160, kMelBins = 201, kMelN = 80, kMelPad = 200;
constexpr int kMelFrames = 128;                 // frames per block, 32 per wave
constexpr int kMelChunks = kMelFrames + 2;      // hop chunks of samples held in LDS
constexpr int kMelPitch = kMelHop + 1;          // LDS pitch of a chunk (odd)
constexpr int kMelSamples = kMelChunks * kMelPitch;
constexpr int kMelW = 208;                      // window table entries (n = 0..207)
constexpr int kMelFbRows = 96;                  // mel rows padded to three 32-row tiles
constexpr int kMelFbPitch = 65;
constexpr int kMelTilePitch = kMelN + 1;
constexpr int kMelBinTiles = 4;                 // bins 0..127 in tiles of 32 (0..100 used)

// constants in HBM (lmufft_melprep): twiddles | window | filterbank per bin tile
constexpr size_t kPrepTab = 0;                                   // float2[400] = (cos, sin)(2 pi i / 400)
constexpr size_t kPrepWin = kPrepTab + sizeof(float) * 2 * kMelFft;           // float[208]
constexpr size_t kPrepFb = 4096;                                 // float[4][96][64]: [tile][mel][lo 32 | hi 32]
constexpr size_t kPrepBytes = kPrepFb + sizeof(float) * kMelBinTiles * kMelFbRows * 64;

// LDS carve in floats
constexpr int kLdsS = 0;
constexpr int kLdsTab = (kMelSamples + 1) & ~1;
constexpr int kLdsWin = kLdsTab + 2 * kMelFft;
constexpr int kLdsFb = kLdsWin + kMelW;
constexpr int kLdsTotal = kLdsFb + 2 * kMelFbRows * kMelFbPitch;  // one filterbank tile per wave group
static_assert(2 * kMelFrames * kMelTilePitch <= kMelSamples, "output tile and the exchange tile reuse the sample area");
constexpr int kMelThreads = 512;  // two groups of four waves: group g takes bin tiles 2g, 2g+1 of the same 128 frames

__device__ __forceinline__ int sample_addr(int base, int n) {  // frame-relative sample n in [0, 400]
  return base + n + (n >= kMelHop ? 1 : 0) + (n >= 2 * kMelHop ? 1 : 0);
}

__global__ __launch_bounds__(kMelThreads) void k_mel_power(const float* __restrict__ wave, const int* __restrict__ lengths,
                                                   long ld_wave, int L, float* __restrict__ mel, int Tmax,
                                                   const char* __restrict__ prep, uint32_t fbmask) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* S = lds + kLdsS;
  float2* tab = reinterpret_cast<float2*>(lds + kLdsTab);
  float* win = lds + kLdsWin;
  float* fbS = lds + kLdsFb;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int wv = (tid >> 6) & 3;  // which 32 frames of the block
  const int grp = tid >> 8;       // which half of the bins (wave-uniform)
  const int gtid = tid & 255;     // thread index inside the group
  const int b = blockIdx.y, f0 = blockIdx.x * kMelFrames;
  int len = lengths ? lengths[b] : L;
  len = min(max(len, 1), L);
  const int Tb = 1 + len / kMelHop;
  const float* wb = wave + (long)b * ld_wave;

  // samples f0*160 - 200 ... of the centred, reflect-padded utterance
  for (int q = tid; q < kMelChunks * kMelHop; q += kMelThreads) {
    int s = f0 * kMelHop - kMelPad + q;
    if (s < 0) s = -s;
    if (s >= len) s = 2 * (len - 1) - s;
    s = min(max(s, 0), len - 1);  // frames past the utterance's end are computed on clamped data and masked
    S[q + q / kMelHop] = wb[s];
  }
  {
    const float* ptab = reinterpret_cast<const float*>(prep + kPrepTab);
    for (int i = tid; i < 2 * kMelFft; i += kMelThreads) lds[kLdsTab + i] = ptab[i];
    const float* pwin = reinterpret_cast<const float*>(prep + kPrepWin);
    for (int i = tid; i < kMelW; i += kMelThreads) win[i] = pwin[i];
  }
  __syncthreads();

  const int base = (32 * wv + r) * kMelPitch;  // this lane's frame starts at chunk 32*wv + r
  // a wave whose 32 frames all lie past the utterance's last frame skips the contractions (its output
  // rows are written as zeros below); it still takes part in the barriers and the staging.  With
  // 400 frames per utterance the fourth block holds 16 frames: three of its waves have none.
  const bool active = f0 + 32 * wv < min(Tb, Tmax);
  f32x16m melacc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) melacc[t][i] = 0.f;

  // Each group of four waves takes one half of the bins (bin tiles 2 grp, 2 grp + 1) for all 128
  // frames: two waves per SIMD instead of one hide each other's LDS latency; the two partial mel
  // sums are added through LDS below.
  float* fbG = fbS + grp * kMelFbRows * kMelFbPitch;
  {
    const int pass = grp;
    f32x16m acc[2][4];
    int idx_e[2], idx_o[2], step[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k = 32 * (2 * pass + t) + r;
      step[t] = (4 * k) % kMelFft;
      idx_e[t] = (k * 2 * hh) % kMelFft;
      idx_o[t] = (k * (2 * hh + 1)) % kMelFft;
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][p][i] = 0.f;
    }
    // MFMA k-step: lanes 0..31 carry n = 2j, 2j+1; lanes 32..63 carry n = 2j+2, 2j+3
#pragma unroll 2
    for (int j = 0; active && j < 102; j += 2) {
      const int n0 = 2 * (j + hh);
      const float xa0 = S[sample_addr(base, n0)], xa1 = S[sample_addr(base, n0 + 1)];
      const float xb0 = S[sample_addr(base, kMelFft - n0)], xb1 = S[sample_addr(base, kMelFft - 1 - n0)];
      const float w0 = win[n0], w1 = win[n0 + 1];
      const float ee = w0 * (xa0 + xb0), oe = w0 * (xa0 - xb0);
      const float eo = w1 * (xa1 + xb1), oo = w1 * (xa1 - xb1);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float2 ce = tab[idx_e[t]], co = tab[idx_o[t]];
        acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ce.x, ee, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(co.x, eo, acc[t][1], 0, 0, 0);
        acc[t][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ce.y, oe, acc[t][2], 0, 0, 0);
        acc[t][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(co.y, oo, acc[t][3], 0, 0, 0);
        idx_e[t] += step[t];
        idx_e[t] -= idx_e[t] >= kMelFft ? kMelFft : 0;
        idx_o[t] += step[t];
        idx_o[t] -= idx_o[t] >= kMelFft ? kMelFft : 0;
      }
    }
    // power of bins k (lo) and 200 - k (hi) -> mel filterbank, straight from registers
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int bt = 2 * pass + t;
      __syncthreads();
      const float* pfb = reinterpret_cast<const float*>(prep + kPrepFb) + (size_t)bt * kMelFbRows * 64;
      for (int i = gtid; i < kMelFbRows * 64; i += 256) fbG[(i >> 6) * kMelFbPitch + (i & 63)] = pfb[i];
      __syncthreads();
      f32x16m plo, phi;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float re = acc[t][0][i] + acc[t][1][i], im = acc[t][2][i] + acc[t][3][i];
        const float rh = acc[t][0][i] - acc[t][1][i], ih = acc[t][2][i] - acc[t][3][i];
        plo[i] = re * re + im * im;
        phi[i] = rh * rh + ih * ih;
      }
#pragma unroll
      for (int mt = 0; active && mt < 3; ++mt) {
        const float* frow = fbG + (32 * mt + r) * kMelFbPitch + 4 * hh;
        if ((fbmask >> (bt * 6 + mt * 2)) & 1u) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            melacc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(frow[(i & 3) + 8 * (i >> 2)], plo[i], melacc[mt], 0, 0, 0);
        }
        if ((fbmask >> (bt * 6 + mt * 2 + 1)) & 1u) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            melacc[mt] =
                __builtin_amdgcn_mfma_f32_32x32x2f32(frow[32 + (i & 3) + 8 * (i >> 2)], phi[i], melacc[mt], 0, 0, 0);
        }
      }
    }
  }

  // group 1 hands its partial mel sums to group 0 through LDS (the sample area is free now); group 0
  // adds them, takes 10 log10 and lays the tile out as rows of 80 floats
  __syncthreads();
  float* tile = S;
  float* xch = S + kMelFrames * kMelTilePitch;
  if (grp == 1) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (c < kMelN) xch[(32 * wv + r) * kMelTilePitch + c] = melacc[mt][i];
      }
  }
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (c < kMelN) {
          const int at = (32 * wv + r) * kMelTilePitch + c;
          tile[at] = 10.0f * log10f(fmaxf(melacc[mt][i] + xch[at], 1e-10f));
        }
      }
  }
  __syncthreads();
  float* mb = mel + (long)b * Tmax * kMelN;
  for (int idx = tid; idx < kMelFrames * kMelN; idx += kMelThreads) {
    const int f = idx / kMelN, c = idx % kMelN, t = f0 + f;
    if (t < Tmax) mb[(long)t * kMelN + c] = t < Tb ? tile[f * kMelTilePitch + c] : 0.f;
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

// ---------------------------------------------------------------------------
// constants, computed once on the host in double precision
// ---------------------------------------------------------------------------
struct MelConstants {
  std::vector<char> bytes;
  uint32_t fbmask = 0;
};

static const MelConstants& mel_constants() {
  static MelConstants mc;
  static std::once_flag once;
  std::call_once(once, [] {
    mc.bytes.assign(kPrepBytes, 0);
    float* tab = reinterpret_cast<float*>(mc.bytes.data() + kPrepTab);
    for (int i = 0; i < kMelFft; ++i) {
      const double a = 2.0 * M_PI * (double)i / (double)kMelFft;
      tab[2 * i] = (float)cos(a);
      tab[2 * i + 1] = (float)sin(a);
    }
    // periodic Hann; the self-paired sample n = 200 enters e[n] = y[n] + y[400-n] twice, hence the half
    float* win = reinterpret_cast<float*>(mc.bytes.data() + kPrepWin);
    for (int n = 0; n < kMelW; ++n) {
      const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)kMelFft);
      win[n] = n < 200 ? (float)w : (n == 200 ? (float)(0.5 * w) : 0.f);
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
    float* pfb = reinterpret_cast<float*>(mc.bytes.data() + kPrepFb);
    for (int bt = 0; bt < kMelBinTiles; ++bt)
      for (int m = 0; m < kMelFbRows; ++m)
        for (int slot = 0; slot < 32; ++slot) {
          const int k = 32 * bt + slot;
          // lo: bin k (k <= 100); hi: bin 200 - k (k < 100, so that bin 100 is counted once)
          const double lo = (m < kMelN && k <= 100) ? fb(k, m) : 0.0;
          const double hi = (m < kMelN && k < 100) ? fb(200 - k, m) : 0.0;
          pfb[((size_t)bt * kMelFbRows + m) * 64 + slot] = (float)lo;
          pfb[((size_t)bt * kMelFbRows + m) * 64 + 32 + slot] = (float)hi;
          if (lo != 0.0) mc.fbmask |= 1u << (bt * 6 + (m / 32) * 2);
          if (hi != 0.0) mc.fbmask |= 1u << (bt * 6 + (m / 32) * 2 + 1);
        }
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
  const MelConstants& mc = mel_constants();
  const size_t lds = sizeof(float) * (size_t)kLdsTotal;
  static_assert(sizeof(float) * (size_t)kLdsTotal <= 160 * 1024, "LDS budget");
  if (hipFuncSetAttribute((const void*)k_mel_power, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return set_error("cannot reserve %zu bytes of LDS for k_mel_power", lds);
  {
    ProfScope ps("k_mel_power", st);
    hipLaunchKernelGGL(k_mel_power, dim3((Tmax + kMelFrames - 1) / kMelFrames, B), dim3(kMelThreads), lds, st, wave, lengths,
                       (long)ld_wave, L, mel, Tmax, (const char*)melprep, mc.fbmask);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("launch of k_mel_power failed: %s", hipGetErrorString(e));
  if (normalize) {
    {
      ProfScope ps("k_mel_norm", st);
      hipLaunchKernelGGL(k_mel_norm, dim3(kMelN / 16, B), dim3(16 * kMelNormGroups), 0, st, mel, lengths, L, Tmax);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return set_error("launch of k_mel_norm failed: %s", hipGetErrorString(e));
  }
  return 0;
}

}  // extern "C"

// Helpers shared by the translation units of liblmufft_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "lmufft.h"

namespace lmufft {

inline char* last_error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

// Records the message for lmufft_last_error() and returns the error code 1.
inline int set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buffer(), 512, fmt, ap);
  va_end(ap);
  return 1;
}

// Partial results of split reductions are written as S "slabs" of `stride` floats; a slab holds up
// to four segments (e.g. gW_h | gb_h | gw_u | gb_u).  Segment q is reps[q] partial copies of n[q]
// floats, every copy starting on a multiple of four floats.
struct SlabOutputs {
  float* out[4];  // destination of every segment (NULL / n = 0 for unused ones)
  long n[4];      // elements per segment
  int reps[4];    // partial copies per slab (0 is read as 1)
};
inline long slab_seg_pad(long n) { return (n + 3) & ~3L; }
inline int slab_reps(const SlabOutputs& o, int q) { return o.reps[q] > 0 ? o.reps[q] : 1; }
inline long slab_seg_offset(const SlabOutputs& o, int q) {
  long off = 0;
  for (int i = 0; i < q; ++i) off += slab_reps(o, i) * slab_seg_pad(o.n[i]);
  return off;
}
inline long slab_stride(const SlabOutputs& o) { return slab_seg_offset(o, 4); }
// out[q][i] = sum over slabs and copies, added in a fixed order (deterministic); defined in lmufft_f32.hip
int launch_reduce_slabs(const float* in, int S, const SlabOutputs& o, hipStream_t st);
// `count` such sums in one launch (each item's slabs, slab count and outputs)
int launch_reduce_slabs_many(const float* const* in, const int* S, const SlabOutputs* o, int count, hipStream_t st);
int launch_gu_reduce(const float* gu_part, const float* u, float* out, int B, int T, int ntt, int tile_t,
                     hipStream_t st);
// workspace sizes of the bf16 path (defined in lmufft_bf16.hip)
size_t forward_workspace_bf16(const lmufft_shape& s);
size_t backward_workspace_bf16(const lmufft_shape& s);
size_t conv_workspace_bf16(const lmufft_shape& s);  // scratch of lmufft_conv_backward_bf16

// FFT route of the causal convolution (defined in fft_conv_f32.hip)
int fft_length(int T);                      // 0 when one transform cannot hold the sequence (2T - 1 > 1024)
bool fft_route_available(int T);            // one transform, or the partitioned kernels for long sequences
size_t fftprep_bytes(int M, int T);
int fft_prepare(const lmufft_shape& s, const float* H, void* prep, hipStream_t st);
size_t fft_conv_workspace_bytes(const lmufft_shape& s);  // scratch of either direction
int fft_conv_forward(const lmufft_shape& s, const float* u, const void* prep, float* m, void* ws, hipStream_t st);
int fft_conv_groups(int M);
// gu[b][s], times (u[b][s] > 0) when u != NULL
int fft_conv_backward(const lmufft_shape& s, const float* gm, const void* prep, const float* u, float* gu, void* ws,
                      hipStream_t st);
// State-space ("scan") route of the bf16 causal convolution for long sequences (defined in scan_conv_bf16.hip).
// `scanprep`: lmufft_scanprep_bytes(M) bytes built by lmufft_scanprep_build_host and copied to the device.
bool scan_supported(int M);
size_t scan_forward_workspace_bytes(int B, int T, int M);  // 0 when the route does not support M
size_t scan_workspace_bytes(int B, int T, int M);          // either direction
int scan_conv_forward(const lmufft_shape& s, const float* u, const void* scanprep, __bf16* m, void* ws, size_t ws_bytes,
                      hipStream_t st);
// gu[b][s], times (u[b][s] > 0) when u != NULL
int scan_conv_backward(const lmufft_shape& s, const __bf16* gm, const void* scanprep, const float* u, float* gu,
                       void* ws, size_t ws_bytes, hipStream_t st);
// out[i] = sum_p part[p*n + i], masked by u[i] > 0 when u != NULL (defined in lmufft_f32.hip)
int launch_sum_slabs_masked(const float* part, const float* u, float* out, long n, int P, hipStream_t st);

// Dropout without a stored mask.  The elements of a call form rows of `width` columns; columns 2c and 2c + 1 of a
// row share one 32-bit hash and read its low / high 16 bits:
//   pairs_per_row = (width + 1) / 2,   g = row * pairs_per_row  (64 bits),
//   x = fmix32((lo32(g) + col / 2) ^ seed0 ^ ((hi32(g) + seed1) * 0x9E3779B1)),   fmix32 = MurmurHash3's finaliser,
//   keep <=> ((x >> 16 (col & 1)) & 0xFFFF) >= threshold,   threshold = round(p * 65536) in [1, 65535].
// The row part (HashRow) is computed once per row, a hash costs an add, a xor and the finaliser, and serves two
// elements.  The forward and the backward of a call recompute the same decisions.  threshold = 0: no dropout.
// `epoch` (may be NULL) points at one 32-bit counter in device memory that is added to seed1 when the kernel RUNS:
// a training step replayed from a HIP graph has its seeds frozen at capture, and draws fresh masks only because
// the graph itself increments that counter (lmufft_dropout_epoch).
struct HashDropout {
  uint32_t seed0, seed1, threshold;
  float inv_keep;  // 1 / (1 - p)
  const uint32_t* epoch;
};
struct HashRow {
  uint32_t lo, c;  // lo32(g);  seed0 ^ ((hi32(g) + seed1) * 0x9E3779B1)
};
__device__ __forceinline__ uint32_t hash_fmix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
// First statement of every kernel that takes a HashDropout by value: folds the device-side epoch into seed1.
__device__ __forceinline__ void hash_resolve_epoch(HashDropout& dr) {
  if (dr.epoch) dr.seed1 += *dr.epoch;
  dr.epoch = nullptr;
}
__device__ __forceinline__ HashRow hash_row(const HashDropout& dr, unsigned long long row, unsigned pairs_per_row) {
  const unsigned long long g = row * pairs_per_row;
  return HashRow{(uint32_t)g, dr.seed0 ^ (((uint32_t)(g >> 32) + dr.seed1) * 0x9E3779B1u)};
}
__device__ __forceinline__ uint32_t hash_pair(const HashRow& hr, unsigned pair) { return hash_fmix32((hr.lo + pair) ^ hr.c); }
__device__ __forceinline__ bool hash_keep_lo(const HashDropout& dr, uint32_t x) { return (x & 0xFFFFu) >= dr.threshold; }
__device__ __forceinline__ bool hash_keep_hi(const HashDropout& dr, uint32_t x) { return (x >> 16) >= dr.threshold; }
__device__ __forceinline__ bool hash_keep(const HashDropout& dr, const HashRow& hr, unsigned col) {
  const uint32_t x = hash_pair(hr, col >> 1);
  return (col & 1) ? hash_keep_hi(dr, x) : hash_keep_lo(dr, x);
}
const uint32_t* dropout_epoch_ptr();  // the current device's counter, or NULL (lmufft_dropout_epoch; layernorm.hip)
inline HashDropout make_hash_dropout(float p_drop, uint32_t seed0, uint32_t seed1) {
  HashDropout dr;
  dr.seed0 = seed0;
  dr.seed1 = seed1;
  dr.epoch = p_drop > 0.f ? dropout_epoch_ptr() : nullptr;
  const double t = (double)p_drop * 65536.0 + 0.5;
  dr.threshold = p_drop > 0.f ? (uint32_t)(t < 1.0 ? 1.0 : (t > 65535.0 ? 65535.0 : t)) : 0u;
  dr.inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  return dr;
}

// Brackets one kernel launch with HIP events when profiling is enabled
// (lmufft_profile_enable); a no-op otherwise.
bool profiling_enabled();
class ProfScope {
 public:
  ProfScope(const char* name, hipStream_t st);
  ~ProfScope();

 private:
  bool active_;
  hipStream_t st_;
  int id_ = -1;
  hipEvent_t start_ = nullptr, stop_ = nullptr;
};

}  // namespace lmufft

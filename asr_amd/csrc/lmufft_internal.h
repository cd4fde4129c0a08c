// Helpers shared by the translation units of liblmufft_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
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

// shared launchers (defined in lmufft_f32.hip)
int launch_reduce_slabs(const float* in, int S, long stride, float* out0, long n0, float* out1, long n1,
                        hipStream_t st);
int launch_gu_reduce(const float* gu_part, const float* u, float* out, int B, int T, int ntt, int tile_t,
                     hipStream_t st);
// workspace sizes of the bf16 path (defined in lmufft_bf16.hip)
size_t forward_workspace_bf16(const lmufft_shape& s);
size_t backward_workspace_bf16(const lmufft_shape& s);

// FFT route of the causal convolution (defined in fft_conv_f32.hip)
int fft_length(int T);                      // 0 when the LDS-resident kernels cannot take T
size_t fftprep_bytes(int M, int T);
int fft_prepare(const lmufft_shape& s, const float* H, void* prep, hipStream_t st);
int fft_conv_forward(const lmufft_shape& s, const float* u, const void* prep, float* m, hipStream_t st);
int fft_conv_groups(int M);
int fft_conv_backward(const lmufft_shape& s, const float* gm, const void* prep, float* part, hipStream_t st);
// out[i] = sum_p part[p*n + i], masked by u[i] > 0 when u != NULL (defined in lmufft_f32.hip)
int launch_sum_slabs_masked(const float* part, const float* u, float* out, long n, int P, hipStream_t st);

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

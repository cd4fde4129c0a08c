// How fast can every CU pull an L2-resident matrix (the weight operand of the projection GEMMs)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every workgroup reads the same `bytes` of w `reps` times, plus its own private slice of a (streamed once)
template <int UNROLL>
__global__ __launch_bounds__(512) void k_pull(const u32x4* __restrict__ w, long n16, int reps, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r)
    for (long i = threadIdx.x; i + (UNROLL - 1) * 512 < n16; i += UNROLL * 512) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int q = 0; q < UNROLL; ++q) v[q] = __builtin_nontemporal_load(&w[i + q * 512]);
#pragma unroll
      for (int q = 0; q < UNROLL; ++q) acc ^= v[q];
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) out[0] = 1;
}

template <int UNROLL>
__global__ __launch_bounds__(512) void k_pull_lds(const u32x4* __restrict__ w, long n16, int reps, unsigned* out) {
  extern __shared__ u32x4 lds[];
  for (int r = 0; r < reps; ++r)
    for (long i = threadIdx.x; i + (UNROLL - 1) * 512 < n16; i += UNROLL * 512) {
#pragma unroll
      for (int q = 0; q < UNROLL; ++q) {
        typedef __attribute__((address_space(1))) const void* gp;
        typedef __attribute__((address_space(3))) void* lp;
        __builtin_amdgcn_global_load_lds((gp)(&w[i + q * 512]), (lp)(lds + (threadIdx.x >> 6) * 64 + q * 512), 16, 0, 0);
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x].x == 0x12345) out[0] = 1;
}

int main() {
  const long bytes = 442368;  // W_h in bf16
  void* w; unsigned* out;
  hipMalloc(&w, bytes); hipMemset(w, 1, bytes); hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 20;
  for (int blocks : {1, 32, 256}) {
    for (int mode = 0; mode < 2; ++mode) {
      for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k_pull<8>, dim3(blocks), dim3(512), 0, 0, (const u32x4*)w, bytes / 16, reps, out);
        else hipLaunchKernelGGL(k_pull_lds<8>, dim3(blocks), dim3(512), 65536, 0, (const u32x4*)w, bytes / 16, reps, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double per_cu = (double)bytes * reps / (ms * 1e-3);
      printf("%s, %3d workgroups of 8 waves: %.1f GB/s per CU (%.1f B/clk at 2.4 GHz), %.2f TB/s in total\n",
             mode ? "global_load_lds   " : "global_load_dwordx4", blocks, per_cu / 1e9, per_cu / 2.4e9, per_cu * blocks / 1e12);
    }
  }
  return 0;
}

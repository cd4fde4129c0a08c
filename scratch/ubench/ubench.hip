// Micro-benchmarks of the per-CU rates the GEMM kernels depend on (one workgroup of 4 waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_mfma(long long* out, int iters, int nwaves_active) {
  f32x16 acc[6];
  for (int a = 0; a < 6; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  bf16x8 x, y;
  for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(float)(threadIdx.x & 3); y[j] = (__bf16)1.f; }
  __syncthreads();
  long long t0 = clock64(), w0 = wall_clock64();
  if ((threadIdx.x >> 6) < nwaves_active)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int a = 0; a < 6; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int a = 0; a < 6; ++a) s += acc[a][0];
  if (threadIdx.x % 64 == 0) { out[(threadIdx.x >> 6) * 4] = t1 - t0; out[(threadIdx.x >> 6) * 4 + 1] = w1 - w0; out[(threadIdx.x >> 6) * 4 + 2] = (long long)s; }
}

template <int MODE>  // 0: ds_read_b64_tr_b16, 1: ds_read_b128, 2: ds_read_b64
__global__ void k_lds(long long* out, int iters, int ld_elems) {
  extern __shared__ __bf16 S[];
  for (int i = threadIdx.x; i < 64 * ld_elems; i += blockDim.x) S[i] = (__bf16)(float)(i & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, g1 = (lane >> 4) & 1, hh = lane >> 5, q = i >> 2, p = i & 3;
  float sum = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const int kl = ((r + it) & 3) * 16, row0 = (r >> 2) * 32;  // loop-variant address: nothing can be hoisted
      if (MODE == 0) {
        typedef bf16x4 __attribute__((address_space(3))) * lp;
        const __bf16* a0 = &S[(kl + 8 * hh + q) * ld_elems + row0 + 16 * g1 + 4 * p];
        bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)a0);
        sum += (float)v[0];
      } else if (MODE == 1) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(&S[(row0 + (lane & 31)) * 72 + kl + 8 * hh]);
        sum += (float)v[0];
      } else {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(&S[(row0 + (lane & 31)) * 68 + kl + 4 * hh]);
        sum += (float)v[0];
      }
    }
  }
  long long t1 = clock64();
  if (lane == 0) { out[(threadIdx.x >> 6) * 4] = t1 - t0; out[(threadIdx.x >> 6) * 4 + 2] = (long long)sum; }
}

int main() {
  long long *d, h[16];
  hipMalloc(&d, sizeof(h));
  const int iters = 2000;
  for (int nw = 1; nw <= 4; nw += 3) {
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(256), 0, 0, d, iters, nw);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mfma 32x32x16 bf16, %d waves active: %.1f clk per MFMA (wave 0), shader clk / 100MHz tick = %.2f\n", nw,
           (double)h[0] / (iters * 6.0), (double)h[0] / (double)h[1]);
  }
  hipFuncSetAttribute((const void*)k_lds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100000);
  for (int ld : {160, 224, 128}) {
    hipLaunchKernelGGL(k_lds<0>, dim3(1), dim3(256), 64 * ld * 2, 0, d, iters, ld);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16 LD=%d, 4 waves: %.1f clk per instruction per wave\n", ld, (double)h[0] / (iters * 10.0));
  }
  hipLaunchKernelGGL(k_lds<1>, dim3(1), dim3(256), 96 * 72 * 2, 0, d, iters, 108);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("ds_read_b128 (144-byte rows), 4 waves: %.1f clk per instruction per wave\n", (double)h[0] / (iters * 10.0));
  hipLaunchKernelGGL(k_lds<2>, dim3(1), dim3(256), 96 * 72 * 2, 0, d, iters, 108);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("ds_read_b64 (136-byte rows), 4 waves: %.1f clk per instruction per wave\n", (double)h[0] / (iters * 10.0));
  hipLaunchKernelGGL(k_lds<0>, dim3(1), dim3(64), 64 * 160 * 2, 0, d, iters, 160);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("ds_read_b64_tr_b16 LD=160, 1 wave: %.1f clk per instruction\n", (double)h[0] / (iters * 10.0));
  return 0;
}

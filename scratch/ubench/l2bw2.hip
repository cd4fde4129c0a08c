// Per-CU pull rate from an L2-resident row-major matrix when a wave-instruction gathers SEG-byte
// pieces of different rows (the k-tile of a GEMM operand) instead of 1 KB of one row.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// matrix [rows][row_bytes]; a k-tile = SEG bytes of every row; thread t reads 16 B: row = t / (SEG/16)
template <int SEG, bool DMA>
__global__ __launch_bounds__(512) void k_tile(const char* __restrict__ w, int rows, int row_bytes, int reps,
                                              unsigned* out) {
  extern __shared__ u32x4 lds[];
  constexpr int LPR = SEG / 16;            // lanes per row
  const int ktiles = row_bytes / SEG;
  u32x4 acc = {0, 0, 0, 0};
  const int r0 = threadIdx.x / LPR, c = (threadIdx.x % LPR) * 16;
  const int rows_per_pass = 512 / LPR;
  for (int rep = 0; rep < reps; ++rep)
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll 4
      for (int r = r0; r < rows; r += rows_per_pass) {
        const char* p = w + (long)r * row_bytes + kt * SEG + c;
        if (DMA) {
          typedef __attribute__((address_space(1))) const void* gp;
          typedef __attribute__((address_space(3))) void* lp;
          __builtin_amdgcn_global_load_lds((gp)p, (lp)(lds + (threadIdx.x >> 6) * 64 + ((r / rows_per_pass) & 7) * 512), 16, 0, 0);
        } else {
          acc ^= *reinterpret_cast<const u32x4*>(p);
        }
      }
    }
  if (DMA) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc = lds[threadIdx.x]; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) out[0] = 1;
}

template <int SEG, bool DMA>
void run(const void* w, unsigned* out, hipEvent_t e0, hipEvent_t e1) {
  const int rows = 384, row_bytes = 1152, reps = 20;
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_tile<SEG, DMA>), dim3(256), dim3(512), 65536, 0, (const char*)w, rows, row_bytes, reps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_cu = (double)rows * row_bytes * reps / (ms * 1e-3);
  printf("%s pieces of %3d B per row: %.1f GB/s per CU (%.1f B/clk), 256 workgroups\n", DMA ? "global_load_lds    " : "global_load_dwordx4",
         SEG, per_cu / 1e9, per_cu / 2.4e9);
}

int main() {
  void* w; unsigned* out;
  hipMalloc(&w, 384 * 1152); hipMemset(w, 1, 384 * 1152); hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  run<64, false>(w, out, e0, e1);  run<64, true>(w, out, e0, e1);
  run<128, false>(w, out, e0, e1); run<128, true>(w, out, e0, e1);
  run<384, false>(w, out, e0, e1); run<384, true>(w, out, e0, e1);
  return 0;
}

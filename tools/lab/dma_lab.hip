// Microbenchmark: per-CU LDS-DMA (global_load_lds_dwordx4) throughput for the two GEMM staging shapes:
//   shape 0: 16 rows x 64 B per wave instruction (BK=32 K-contiguous tile)
//   shape 1:  8 rows x 128 B per wave instruction (BK=64 K-contiguous tile, full cache lines)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/dma_lab tools/lab/dma_lab.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int SHAPE>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ base, long ld_bytes, int rows, int ksteps, int row_blocks) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // block reads a 256-row panel (like the R operand) + another 256-row panel (S operand) per k-step: 32 KB
  const int panel = (blockIdx.x % row_blocks) * 256;
  const int panel2 = ((blockIdx.x * 7 + 3) % row_blocks) * 256;
  uint32_t voff[4];
  for (int i = 0; i < 4; ++i) {
    int row, col;
    if (SHAPE == 0) { const int inst = (i & 1) * 8 + wave; row = inst * 16 + (lane >> 2); col = (lane & 3) * 16; }
    else { const int inst = (i & 1) * 8 + wave; row = inst * 8 + (lane >> 3); col = (lane & 7) * 16; }   // 128 rows x 128 B per operand per k64 step
    const int p = (i < 2) ? panel : panel2;
    voff[i] = 0; (void)p;
    voff[i] = (uint32_t)((long)((p + row) % rows) * ld_bytes + col);
  }
  const int kbytes = (SHAPE == 0) ? 64 : 128;
  for (int k = 0; k < ksteps; ++k) {
    const char* ub = base + (long)k * kbytes;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)(ub + voff[i]), (lds_void*)(smem + ((k & 3) * 32 + i * 8 + wave) * 1024), 16, 0, 0);
    if ((k & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (smem[threadIdx.x] == 123 && ksteps < 0) printf("x");
}

int main(int argc, char** argv) {
  const int rows = 131072; const long ld = 2048;      // bytes per row (K = 1024 bf16)
  char* d; hipMalloc(&d, (long)rows * ld);
  hipMemset(d, 1, (long)rows * ld);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape = 0; shape < 2; ++shape)
    for (int row_blocks : {8, 64, 512}) {
      const int ksteps = (shape == 0) ? 32 : 16;       // one pass over K = 1024
      const int reps = 24;                              // blocks per CU in sequence
      auto run = [&]() {
        if (shape == 0) hipLaunchKernelGGL(dma_kernel<0>, dim3(256 * reps), dim3(512), 131072, 0, d, ld, rows, ksteps, row_blocks);
        else hipLaunchKernelGGL(dma_kernel<1>, dim3(256 * reps), dim3(512), 131072, 0, d, ld, rows, ksteps, row_blocks);
      };
      hipFuncSetAttribute((const void*)dma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute((const void*)dma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      run(); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) run();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double bytes = 256.0 * reps * ksteps * ((shape == 0) ? 32768.0 : 32768.0);
      printf("shape %d (%s) distinct panels %4d: %.3f ms  %.2f TB/s  = %.1f B/clk/CU @2.4GHz\n", shape,
             shape == 0 ? "16 rows x 64 B" : "8 rows x 128 B", row_blocks, ms, bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9);
    }
  return 0;
}

// Lab: does the relative placement of a stream that is READ and a stream that is WRITTEN at the same row offsets matter?  (The SwiGLU-backward
// epilogue reads gate|up[r] and writes dgate|dup[r]; rms_bwd reads dout[r], x[r] and writes dx[r]: tensors from a 2 MiB-granular allocator sit
// at the same offset modulo every interleave period.)  dst = src-shaped buffer + `shift` bytes.
// build: hipcc --offload-arch=gfx950 -O3 tools/lab/copy_lab.hip -o tools/lab/libs/copy_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16) {
  // a workgroup walks contiguous 64 KiB blocks (256 threads x 16 loads x 16 B)
  for (long blk = blockIdx.x; blk * 4096 < n16; blk += gridDim.x) {
    u32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(src + blk * 4096 + i * 256 + threadIdx.x);
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i][0] += 1; dst[blk * 4096 + i * 256 + threadIdx.x] = v[i]; }
  }
}

int main() {
  const size_t bytes = (size_t)1610612736;      // 1.5 GiB each way
  char *a, *b;
  (void)hipMalloc(&a, bytes + (64 << 20)); (void)hipMalloc(&b, bytes + (64 << 20));
  (void)hipMemset(a, 1, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep)
    for (long shift : {0L, 128L, 1024L, 4096L + 128, 65536L + 128, (1L << 20) + 4096 + 128, (3L << 20) + 8192 + 256}) {
      copy_kernel<<<2048, 256>>>((const u32x4*)a, (u32x4*)(b + shift), bytes / 16);
      (void)hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) copy_kernel<<<2048, 256>>>((const u32x4*)a, (u32x4*)(b + shift), bytes / 16);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("shift %9ld B: %.3f ms  %.0f GB/s (read + write)\n", shift, ms / 5, 2.0 * bytes / (ms / 5) / 1e6);
    }
  return 0;
}

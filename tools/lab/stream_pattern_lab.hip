// Lab: does the GB/s of a token-major [M, W] bf16 stream depend on how many contiguous bytes of a row one visit takes?
// Every wave owns 16 rows (the MFMA column-operand layout of the LoRA kernels: lane = (row lane&15, 16-byte piece
// lane>>4), one instruction = 16 rows x 64 B) and walks the columns in chunks of CH: all loads of a chunk are issued,
// then consumed (a dependent reduction), then the next chunk.  CH = 128 is what lora_project does today (256 B of a
// row per visit); CH = 512 / 1024 / 3072 take 1 / 2 / 6 KiB of a row per visit.  ROWWISE: one wave per row, 1 KiB per
// instruction (the norm / SwiGLU kernels' pattern) for reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int CH>
__global__ __launch_bounds__(256) void tok16_kernel(const uint16_t* __restrict__ X, long ld, int M, int W, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l15 = lane & 15;
  const long row = ((long)blockIdx.x * 4 + wave) * 16 + l15;
  if (row >= M) return;
  const uint16_t* xr = X + row * ld;
  uint32_t acc = 0;
  for (int c = 0; c < W; c += CH) {
    uint4 v[CH / 32];
#pragma unroll
    for (int s = 0; s < CH / 32; ++s) v[s] = *reinterpret_cast<const uint4*>(xr + c + 32 * s + 8 * g);
#pragma unroll
    for (int s = 0; s < CH / 32; ++s) acc ^= v[s].x + v[s].y + v[s].z + v[s].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void rowwise_kernel(const uint16_t* __restrict__ X, long ld, int M, int W, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= M) return;
  const uint16_t* xr = X + row * ld;
  uint32_t acc = 0;
  for (int c = 0; c < W; c += 512) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + c + 8 * lane);
    acc ^= v.x + v.y + v.z + v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// lora_bgrad's tiling: a wave owns RB x 16 rows, walks 64-column chunks (2 loads per row block and chunk), 4 waves per
// workgroup, `lds` bytes of dynamic LDS to cap the workgroups per CU the way the kernel's 80 KiB tile does.
template <int RB>
__global__ __launch_bounds__(256) void tokrb_kernel(const uint16_t* __restrict__ X, long ld, int M, int W, uint32_t* out) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l15 = lane & 15;
  const long row0 = ((long)blockIdx.x * 4 + wave) * (16 * RB) + l15;
  if (row0 >= M) return;
  uint32_t acc = 0;
  for (int c = 0; c < W; c += 64) {
    uint4 v[RB][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) v[rb][s] = *reinterpret_cast<const uint4*>(X + (row0 + 16 * rb) * ld + c + 32 * s + 8 * g);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc ^= v[rb][s].x + v[rb][s].y + v[rb][s].z + v[rb][s].w;
  }
  if (acc == 0x12345678u) { out[0] = acc; smem[0] = 1; }
}
template <typename F> float timeit(F f, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < iters; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / iters;
}
int main() {
  const int M = 131072;
  for (int W : {1024, 3072}) {
    uint16_t* X; uint32_t* out;
    hipMalloc(&X, (size_t)M * W * 2); hipMalloc(&out, 4);
    hipMemset(X, 1, (size_t)M * W * 2);
    const double gb = (double)M * W * 2 / 1e9;
    const int nb = M / 64;
    float t;
    t = timeit([&] { hipLaunchKernelGGL(tok16_kernel<128>, dim3(nb), dim3(256), 0, 0, X, (long)W, M, W, out); }, 10);
    printf("W=%d tok16 CH=128 : %.1f us  %.0f GB/s\n", W, t * 1e3, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL(tok16_kernel<256>, dim3(nb), dim3(256), 0, 0, X, (long)W, M, W, out); }, 10);
    printf("W=%d tok16 CH=256 : %.1f us  %.0f GB/s\n", W, t * 1e3, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL(tok16_kernel<512>, dim3(nb), dim3(256), 0, 0, X, (long)W, M, W, out); }, 10);
    printf("W=%d tok16 CH=512 : %.1f us  %.0f GB/s\n", W, t * 1e3, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL(tok16_kernel<1024>, dim3(nb), dim3(256), 0, 0, X, (long)W, M, W, out); }, 10);
    printf("W=%d tok16 CH=1024: %.1f us  %.0f GB/s\n", W, t * 1e3, gb / t * 1e3);
    t = timeit([&] { hipLaunchKernelGGL(rowwise_kernel, dim3(M / 4), dim3(256), 0, 0, X, (long)W, M, W, out); }, 10);
    printf("W=%d rowwise        : %.1f us  %.0f GB/s\n", W, t * 1e3, gb / t * 1e3);
    for (int lds : {0, 81920}) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&tokrb_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
      hipFuncSetAttribute(reinterpret_cast<const void*>(&tokrb_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
      t = timeit([&] { hipLaunchKernelGGL(tokrb_kernel<8>, dim3(M / 512), dim3(256), lds, 0, X, (long)W, M, W, out); }, 10);
      printf("W=%d tokrb RB=8 (128 rows per wave, %d WGs) lds=%d: %.1f us  %.0f GB/s\n", W, M / 512, lds, t * 1e3, gb / t * 1e3);
      t = timeit([&] { hipLaunchKernelGGL(tokrb_kernel<2>, dim3(M / 128), dim3(256), lds, 0, X, (long)W, M, W, out); }, 10);
      printf("W=%d tokrb RB=2 (32 rows per wave, %d WGs) lds=%d: %.1f us  %.0f GB/s\n", W, M / 128, lds, t * 1e3, gb / t * 1e3);
    }
    hipFree(X); hipFree(out);
  }
  return 0;
}

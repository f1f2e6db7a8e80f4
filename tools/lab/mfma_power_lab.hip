// Lab: sustained matrix-pipe throughput and shader clock of v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 streams
// with nothing else running (register operands, random bf16 data, 2 waves per SIMD on every CU, ~40 ms of back-to-back launches):
// is the 2.5 PFLOP/s datasheet peak reachable at all under sustained load, and does the tile shape (operand register reads per
// flop: 32x32x16 reads half as many) change the clock the chip settles at?
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power_lab tools/lab/mfma_power_lab.hip && /tmp/mfma_power_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ bf16x8 rnd8(uint32_t s) {
  bf16x8 v;
  for (int e = 0; e < 8; ++e) { s = s * 1664525u + 1013904223u; v[e] = (short)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 3) & 0x8000)); }   // +-[0.0078, 0.0156)-ish bf16, random mantissa/sign
  return v;
}

template <int SHAPE, int ZERO>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(long long* cyc, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd8(threadIdx.x * 9781u + i * 77u + 1u); b[i] = rnd8(threadIdx.x * 6151u + i * 131u + 7u); }
  if (ZERO) for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = 0; b[i][e] = 0; }
  const long long t0 = __builtin_readcyclecounter();
  float r = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * rep], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][15];
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && (threadIdx.x >> 6) == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int SHAPE, int ZERO>
void run(const char* name, long long* d_cyc, float* d_sink, int iters, int launches) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, ZERO>), dim3(256), dim3(512), 0, 0, d_cyc, d_sink, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k<SHAPE, ZERO>), dim3(256), dim3(512), 0, 0, d_cyc, d_sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c[256]; hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost);
  double cm = 0; for (int i = 0; i < 256; ++i) cm += (double)c[i]; cm /= 256;
  // per wave per iteration: SHAPE 16: 32 MFMAs x 16384 flop; SHAPE 32: 16 MFMAs x 32768 flop  -> 524288 flop either way
  const double flop = 256.0 * 8 * (double)iters * 524288.0 * launches;
  const double mf_per_wave = (SHAPE == 16 ? 32.0 : 16.0) * iters;
  printf("%-44s %8.2f ms  %7.1f TFLOP/s  | %6.2f cyc/MFMA/wave (2 waves/SIMD)  counter-clock %5.0f MHz\n", name, ms, flop / ms / 1e9, cm / mf_per_wave,
         cm * launches / (ms * 1e3));
}

int main() {
  long long* d_cyc; float* d_sink;
  hipMalloc(&d_cyc, 256 * sizeof(long long)); hipMalloc(&d_sink, 256 * 512 * sizeof(float));
  const int iters = 4096, launches = 40;     // ~1.4 ms per launch at peak
  for (int round = 0; round < 2; ++round) {
    run<16, 0>("16x16x32 bf16, random operands", d_cyc, d_sink, iters, launches);
    run<32, 0>("32x32x16 bf16, random operands", d_cyc, d_sink, iters, launches);
    run<16, 1>("16x16x32 bf16, zero operands", d_cyc, d_sink, iters, launches);
    run<32, 1>("32x32x16 bf16, zero operands", d_cyc, d_sink, iters, launches);
  }
  return 0;
}

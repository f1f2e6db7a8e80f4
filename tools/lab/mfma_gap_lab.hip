// Lab: what hides in the gap of a v_mfma_f32_32x32x16_bf16 stream issued by ONE wave per SIMD (the dK/dV kernel's regime)?
// One workgroup of 4 waves per CU, 1024 MFMAs per wave, accumulators in AccVGPRs (inline asm), per MFMA a configurable filler
// set: NV x (v_mul, v_exp, v_mul) independent vector ops, NT x ds_read_b64_tr_b16 pairs (ring of 7, counted lgkmcnt), NOP wait
// states in front of the MFMA.  Prints shader cycles per MFMA (s_memtime-class counter) for each variant.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_gap_lab tools/lab/mfma_gap_lab.hip && /tmp/mfma_gap_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NV, int NT, int NOP, int MODE, int ORDER = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(long long* out, float* sink, float c2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i * 1e-3f;
  __syncthreads();
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a0 = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 2, 2, 3, 3, 4, 4};
  float x[16], y[16];
  for (int r = 0; r < 16; ++r) { x[r] = 0.001f * (lane + r); y[r] = 1.0f + r; }
  const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (lane & 15) * 264 + (lane >> 4) * 8;
  bf16x4 tl[8], th[8];
  for (int j = 0; j < 8; ++j) { tl[j] = {0, 0, 0, 0}; th[j] = {0, 0, 0, 0}; }
  if (NT) {
#pragma unroll
    for (int j = 0; j < 7; ++j) asm volatile("ds_read_b64_tr_b16 %0, %2 offset:0\n\tds_read_b64_tr_b16 %1, %2 offset:4224" : "=&v"(tl[j]), "=&v"(th[j]) : "v"(la));
  }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 128; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      auto issue = [&]() { asm volatile("ds_read_b64_tr_b16 %0, %2 offset:2048\n\tds_read_b64_tr_b16 %1, %2 offset:6272" : "=&v"(tl[(j + 7) & 7]), "=&v"(th[(j + 7) & 7]) : "v"(la)); };
      auto wait14 = [&]() { asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(tl[j]), "+v"(th[j])); };
      auto wait12 = [&]() { asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(tl[j]), "+v"(th[j])); };
      auto mfma = [&]() {
        bf16x8 a = a0;
        if (NT) { a[0] = tl[j][0]; a[1] = tl[j][1]; a[2] = tl[j][2]; a[3] = tl[j][3]; a[4] = th[j][0]; a[5] = th[j][1]; a[6] = th[j][2]; a[7] = th[j][3]; }
        if (MODE == 0) {
          if (NOP == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
          else asm volatile("s_nop %3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b), "n"(NOP - 1));
        } else {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        }
      };
      auto valu = [&]() {
#pragma unroll
        for (int v = 0; v < NV; ++v) {      // staged: three independent ops on three different elements (asm: no SLP packing, order pinned)
          const int e = (j * NV + v) & 15;
          asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[e]) : "s"(c2));
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[(e + 5) & 15]));
          asm volatile("v_mul_f32 %0, %1, %0" : "+v"(y[(e + 9) & 15]) : "v"(x[(e + 9) & 15]));
        }
      };
      if (ORDER == 0) { if (NT) { issue(); wait14(); } mfma(); valu(); }
      if (ORDER == 1) { if (NT) wait12(); mfma(); if (NT) issue(); valu(); }
      if (ORDER == 2) { if (NT) wait12(); mfma(); valu(); if (NT) issue(); }
      if (ORDER == 3) { if (NT) issue(); valu(); if (NT) wait14(); mfma(); }
      if (ORDER == 4) { valu(); if (NT) wait12(); mfma(); if (NT) issue(); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int r = 0; r < 16; ++r) s += x[r] + y[r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NV, int NT, int NOP, int MODE, int ORDER = 0>
void run(const char* name, long long* d_out, float* d_sink) {
  hipFuncSetAttribute((const void*)k<NV, NT, NOP, MODE, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NV, NT, NOP, MODE, ORDER>), dim3(256), dim3(256), 65536, 0, d_out, d_sink, 1.0001f);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += (double)h[i];
  printf("%-58s %7.1f cycles per MFMA\n", name, m / 256 / 1024.0);
}

int main() {
  long long* d_out; float* d_sink;
  hipMalloc(&d_out, 256 * sizeof(long long)); hipMalloc(&d_sink, 256 * 256 * sizeof(float));
  run<0, 0, 0, 0>("MFMA only", d_out, d_sink);
  run<1, 0, 0, 0>("MFMA + 1 x (mul, exp2, mul)", d_out, d_sink);
  run<2, 0, 0, 0>("MFMA + 2 x (mul, exp2, mul)", d_out, d_sink);
  run<0, 1, 0, 0, 0>("order 0 [TR TR wait MFMA V]: reads only", d_out, d_sink);
  run<1, 1, 0, 0, 0>("order 0 [TR TR wait MFMA V]: reads + 1 x V", d_out, d_sink);
  run<1, 1, 1, 0, 0>("order 0 + s_nop 0: reads + 1 x V", d_out, d_sink);
  run<1, 1, 2, 0, 0>("order 0 + s_nop 1: reads + 1 x V", d_out, d_sink);
  run<1, 1, 0, 0, 1>("order 1 [wait MFMA TR TR V]: reads + 1 x V", d_out, d_sink);
  run<1, 1, 0, 0, 2>("order 2 [wait MFMA V TR TR]: reads + 1 x V", d_out, d_sink);
  run<1, 1, 0, 0, 3>("order 3 [TR TR V wait MFMA]: reads + 1 x V", d_out, d_sink);
  run<1, 1, 0, 0, 4>("order 4 [V wait MFMA TR TR]: reads + 1 x V", d_out, d_sink);
  run<2, 1, 0, 0, 1>("order 1: reads + 2 x V", d_out, d_sink);
  run<2, 1, 0, 0, 4>("order 4: reads + 2 x V", d_out, d_sink);
  return 0;
}

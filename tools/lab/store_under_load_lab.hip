// Lab: how fast does ONE CU drain a 128 KiB store burst (a 256x256 bf16 C tile, 16 stores of 16 B per lane and wave) while
// every CU of the chip streams operand tiles L2 -> LDS at the persistent GEMM's rate (64 KiB per ~2400 cycles per CU)?
// One 512-thread workgroup per CU, 128 KiB of LDS (1 workgroup per CU), de-phased starts.  Every workgroup loops over
// "K tiles": 8 LDS-DMA pieces per wave from a PANEL-bytes window (re-read by everyone: L2 / Infinity Cache hits), paced with
// s_sleep; every 16th K tile it issues the store burst and stamps first issue -> last issue -> vmcnt(0).
// Variants: store policy (0 plain, 1 non-temporal, 2 sc1, 3 sc0 sc1), LOADS on/off, rows x bytes shape of a store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int POLICY>
__device__ __forceinline__ void st16(char* p, u32x4 v) {
  if (POLICY == 0) *reinterpret_cast<u32x4*>(p) = v;
  else if (POLICY == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

template <int POLICY, int ROWS, int LOADS, int DRAIN, int SHARE>
__global__ __launch_bounds__(512, 2) void k(const char* panel, long panel_bytes, char* C, long ldc_bytes, long long* out, int ktiles, int pace) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {   // de-phase
    const long long until = (long long)__builtin_readcyclecounter() + (long long)((blockIdx.x >> 3) & 15) * 2400;
    while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(8);
  }
  constexpr int BPR = 1024 / ROWS;
  const int r = lane / (BPR / 16), c = lane % (BPR / 16);
  long long t_issue = 0, t_done = 0; int nb = 0;
  // SHARE = 1: the 32 workgroups of an XCD (ids equal mod 8) stream the SAME bytes, a few K tiles apart at most: L2 hits, as the
  // GEMM's operand panels; SHARE = 0: every workgroup its own bytes (fabric / HBM traffic: 256 x 64 KiB per K tile)
  long off = SHARE ? ((long)(blockIdx.x & 7) * (panel_bytes / 8)) % panel_bytes : ((long)blockIdx.x * 65536) % panel_bytes;
  for (int kt = 0; kt < ktiles; ++kt) {
    const long long t0 = __builtin_readcyclecounter();
    if (LOADS == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const char* src = panel + (off + (long)(i * 8 + wave) * 1024 + lane * 16) % panel_bytes;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + ((kt & 1) * 65536 + (i * 8 + wave) * 1024)), 16, 0, 0);
      }
      off = (off + 65536) % panel_bytes;
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    if (LOADS == 2) {       // the same bytes as ordinary register loads (asm: hipcc must not wait for them), 8 in flight across the burst
      u32x4 r0, r1, r2, r3, r4, r5, r6, r7;
      const char* b0 = panel + (off + (long)wave * 1024 + lane * 16) % panel_bytes;
      asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:1024\n\tglobal_load_dwordx4 %2, %8, off offset:2048\n\t"
                   "global_load_dwordx4 %3, %8, off offset:3072\n\tglobal_load_dwordx4 %4, %9, off\n\tglobal_load_dwordx4 %5, %9, off offset:1024\n\t"
                   "global_load_dwordx4 %6, %9, off offset:2048\n\tglobal_load_dwordx4 %7, %9, off offset:3072\n\ts_waitcnt vmcnt(8)"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(b0), "v"(b0 + 32768) : "memory");
      asm volatile("" :: "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7));
      off = (off + 65536) % panel_bytes;
    }
    if ((kt & 15) == 15) {
      if (DRAIN == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (DRAIN == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      const long long s0 = __builtin_readcyclecounter();
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int piece = wave * 16 + s;
        const int prow = (piece * 1024) / (512 * ROWS) * ROWS;
        const int pcol = ((piece * 1024) % (512 * ROWS)) / ROWS;
        char* p = C + ((long)blockIdx.x * 256 + prow + r) * ldc_bytes + pcol + c * 16;
        const u32x4 v = {(unsigned)lane, (unsigned)s, (unsigned)kt, 7u};
        st16<POLICY>(p, v);
      }
      const long long s1 = __builtin_readcyclecounter();
      t_issue += s1 - s0; ++nb;
      if (LOADS == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_done += __builtin_readcyclecounter() - s0; }
    }
    while ((long long)__builtin_readcyclecounter() - t0 < pace) __builtin_amdgcn_s_sleep(2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = nb ? t_issue / nb : 0; out[(blockIdx.x * 8 + wave) * 2 + 1] = nb ? t_done / nb : 0; }
}

template <int POLICY, int ROWS, int LOADS, int DRAIN, int SHARE>
void run(const char* panel, long pb, char* C, long ldc, long long* dout, int nwg, const char* name) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<POLICY, ROWS, LOADS, DRAIN, SHARE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<POLICY, ROWS, LOADS, DRAIN, SHARE><<<nwg, 512, 131072>>>(panel, pb, C, ldc, dout, 128, 2400);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(nwg * 16);
  hipMemcpy(h.data(), dout, sizeof(long long) * nwg * 16, hipMemcpyDeviceToHost);
  std::vector<long long> a;
  for (int i = 0; i < nwg * 8; ++i) a.push_back(h[2 * i]);
  std::sort(a.begin(), a.end());
  printf("  %-34s loads %d drain %d share %d: burst issue (16 stores per wave, 8 waves) median %6lld cycles, p90 %6lld  -> %.1f B/clk/CU   kernel %.3f ms\n", name, (int)LOADS, DRAIN, SHARE,
         a[a.size() / 2], a[a.size() * 9 / 10], 131072.0 / a[a.size() / 2], ms);
}
int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256;
  const long pb = (argc > 2 ? atol(argv[2]) : 64) << 20;
  const long ldc = 8192;
  char *C, *panel; long long* dout;
  hipMalloc(&C, (size_t)nwg * 256 * ldc); hipMalloc(&panel, pb + 65536); hipMalloc(&dout, sizeof(long long) * nwg * 16);
  hipMemset(panel, 1, pb + 65536);
  printf("%d workgroups, operand window %ld MiB, 128 K tiles of 64 KiB per workgroup paced at 2400 cycles, a 128 KiB store burst every 16th\n", nwg, pb >> 20);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 16, 1, 0, 1>(panel, pb, C, ldc, dout, nwg, "plain 16x64, LDS-DMA shared panels");
    run<1, 16, 1, 0, 1>(panel, pb, C, ldc, dout, nwg, "nt    16x64, LDS-DMA shared panels");
    run<0, 2, 1, 0, 1>(panel, pb, C, ldc, dout, nwg, "plain 2x512, LDS-DMA shared panels");
    run<0, 16, 1, 0, 0>(panel, pb, C, ldc, dout, nwg, "plain 16x64, LDS-DMA private bytes");
    run<0, 16, 0, 0, 1>(panel, pb, C, ldc, dout, nwg, "plain 16x64, no loads");
  }
  return 0;
}

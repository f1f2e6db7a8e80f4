// Lab: what does one 16-byte-per-lane global store instruction cost a wave, by the shape of the 1 KiB it writes?
// One 512-thread workgroup per CU (as the persistent GEMM), every wave issues NST stores back to back (the epilogue of a
// 256x256 bf16 tile = 16 per wave), shapes: rows x bytes per instruction = 16x64, 8x128, 4x256, 2x512, 1x1024; row stride LDC.
// Prints cycles (s_memtime) from first issue to last issue, and to completion (vmcnt(0)), median over workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NST = 16;
template <int ROWS, bool NT>
__global__ __launch_bounds__(512) void k(char* C, long ldc_bytes, long long* out, int rounds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int BPR = 1024 / ROWS;            // bytes per row per instruction
  const int r = lane / (BPR / 16), c = lane % (BPR / 16);
  // tile of this workgroup: 256 rows x 512 bytes; wave w, store s -> disjoint pieces
  char* base = C + (long)blockIdx.x * 256 * ldc_bytes;
  long long t_issue = 0, t_done = 0;
  for (int it = 0; it < rounds; ++it) {
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      const int piece = wave * NST + s;                       // 128 pieces of 1 KiB = the 128 KiB tile
      const int prow = (piece * 1024) / (512 * ROWS) * ROWS;  // first row of the piece when the tile is cut into ROWS-row slabs of 512 B rows
      const int pcol = ((piece * 1024) % (512 * ROWS)) / ROWS; // byte column of the piece inside the 512-byte row
      char* p = base + (long)(prow + r) * ldc_bytes + pcol + c * 16 + (long)it * 0;
      const u32x4 v = {(unsigned)lane, (unsigned)s, (unsigned)it, 7u};
      if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
      else *reinterpret_cast<u32x4*>(p) = v;
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = __builtin_readcyclecounter();
    t_issue += t1 - t0; t_done += t2 - t0;
  }
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t_issue / rounds; out[(blockIdx.x * 8 + wave) * 2 + 1] = t_done / rounds; }
}
template <int ROWS, bool NT> void run(char* C, long ldc, long long* dout, int nwg) {
  const int rounds = 8;
  k<ROWS, NT><<<nwg, 512>>>(C, ldc, dout, rounds);
  hipDeviceSynchronize();
  std::vector<long long> h(nwg * 16);
  hipMemcpy(h.data(), dout, sizeof(long long) * nwg * 16, hipMemcpyDeviceToHost);
  std::vector<long long> a, b;
  for (int i = 0; i < nwg * 8; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  printf("  %2d rows x %4d B per store%s: issue %6lld cycles (p90 %6lld), complete %6lld (p90 %6lld)  -> %.1f B/clk/CU by completion\n", ROWS, 1024 / ROWS, NT ? " nt" : "   ",
         a[a.size() / 2], a[a.size() * 9 / 10], b[b.size() / 2], b[b.size() * 9 / 10], 131072.0 / b[b.size() / 2]);
}
int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256;
  const long ldc = 8192;                     // bytes per C row (N = 4096 bf16)
  char* C; long long* dout;
  hipMalloc(&C, (size_t)nwg * 256 * ldc); hipMalloc(&dout, sizeof(long long) * nwg * 16);
  printf("%d workgroups x 8 waves x %d stores of 1 KiB (128 KiB per workgroup), C row stride %ld B\n", nwg, NST, ldc);
  for (int rep = 0; rep < 2; ++rep) {
    run<16, true>(C, ldc, dout, nwg); run<8, true>(C, ldc, dout, nwg); run<4, true>(C, ldc, dout, nwg); run<2, true>(C, ldc, dout, nwg);
    run<16, false>(C, ldc, dout, nwg); run<2, false>(C, ldc, dout, nwg);
  }
  return 0;
}

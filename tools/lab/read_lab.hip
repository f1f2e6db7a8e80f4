// Lab: HBM read rate of a row-major [M, W] bf16 matrix streamed by blocks of 256 token rows in COLUMN-CHUNK order (the LoRA ring kernels'
// order: all 256 rows' piece of chunk c, then chunk c + 1, ...), as a function of the chunk width: 128 B (one line per row and stage,
// the ring kernels' shape), 256, 512, 1024 B, and whole rows.  Loads are global_load_dwordx4 into registers (8 per lane in flight, 4 waves per
// workgroup, 2-4 workgroups per CU), summed so that nothing is dropped.
// build: hipcc --offload-arch=gfx950 -O3 tools/lab/read_lab.hip -o tools/lab/libs/read_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// CW = chunk width in bytes; a wave instruction covers (1024 / CW') rows x CW' bytes with CW' = min(CW, 1024)
template <int CW>
__global__ __launch_bounds__(256) void read_kernel(const char* X, long ldb, int M, int Wb, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int CWI = CW > 1024 ? 1024 : CW;           // bytes per row per instruction
  constexpr int RPI = 1024 / CWI;                      // rows per instruction
  const int lrow = lane / (CWI / 16), lcol = (lane % (CWI / 16)) * 16;
  unsigned acc = 0;
  for (long blk = blockIdx.x; blk * 256 < M; blk += gridDim.x) {
    const char* base = X + (blk * 256 + wave * 64) * ldb;       // the wave's 64 rows
    for (int c = 0; c < Wb; c += CW) {
      // the wave's [64 rows x CW bytes] piece of this chunk: 64 * CW / 1024 instructions
      constexpr int NI = 64 * CW / 1024;
      u32x4 v[NI > 16 ? 16 : NI];
#pragma unroll
      for (int i0 = 0; i0 < NI; i0 += 16) {
#pragma unroll
        for (int i = 0; i < (NI > 16 ? 16 : NI); ++i) {
          const int ii = i0 + i;
          const int sub = (ii * 1024) / (64 * CWI);     // which CWI-wide slice of the chunk (CW > 1024)
          const int r = (ii % (64 / RPI)) * RPI + lrow;
          v[i] = __builtin_nontemporal_load((const u32x4*)(base + (long)r * ldb + c + sub * CWI + lcol));
        }
#pragma unroll
        for (int i = 0; i < (NI > 16 ? 16 : NI); ++i) acc += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int CW>
float run(const char* X, int M, int W, unsigned* out, int grid) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  read_kernel<CW><<<grid, 256>>>(X, (long)W * 2, M, W * 2, out);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) read_kernel<CW><<<grid, 256>>>(X, (long)W * 2, M, W * 2, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  unsigned* out; (void)hipMalloc(&out, 64);
  // (row strides: 2 KiB, 2 KiB + 128 B, 4 KiB, 4 KiB + 128 B, 6 KiB, 8 KiB, 12 KiB; M scaled so that every matrix is ~1.6 GB, far beyond the 256 MiB Infinity Cache)
  for (int W : {1024, 1088, 2048, 2112, 3072, 4096, 6144}) {
    const int M = (int)(((size_t)131072 * 6144 / W) / 256 * 256);
    char* X; (void)hipMalloc(&X, (size_t)M * W * 2); (void)hipMemset(X, 1, (size_t)M * W * 2);
    const double gb = (double)M * W * 2 / 1e9;
    for (int grid : {1024}) {
      float a = run<128>(X, M, W, out, grid), b = run<256>(X, M, W, out, grid), c = run<512>(X, M, W, out, grid), d = run<1024>(X, M, W, out, grid), e = W % 1024 == 0 ? run<2048>(X, M, W, out, grid) : 1e9f;
      printf("W=%5d (%.2f GB) grid %4d: chunk 128 B %.0f GB/s | 256 B %.0f | 512 B %.0f | 1024 B %.0f | 2048 B %.0f\n", W, gb, grid, gb / a * 1e3, gb / b * 1e3, gb / c * 1e3, gb / d * 1e3, gb / e * 1e3);
    }
    (void)hipFree(X);
  }
  return 0;
}

// Lab: the store rate of a GEMM register epilogue's access shapes (bf16 C tile of 256 x 256 per workgroup, 8 waves, persistent over tiles).
//   shape 0: per wave instruction 16 rows x 64 B (gemm_pers.hip: lane = (row l15, 16-byte piece g4) of the wave's 32 columns)
//   shape 1: per wave instruction 8 rows x 128 B (lane = (row lane >> 3, piece lane & 7) of 64 contiguous columns)
//   shape 2: per wave instruction 32 rows x 32 B (attn.hip store_T: lane & 31 = row, lane >> 5 = 16-byte piece), a wave = 32 rows x 256 B
//   shape 3: the same 32 x 256 B wave tile as 4 rows x 256 B per instruction (lane >> 4 = row, lane & 15 = piece)
// build: hipcc --offload-arch=gfx950 -O3 tools/lab/store_lab.hip -o tools/lab/libs/store_lab ; run: tools/lab/libs/store_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool NT>
__global__ __launch_bounds__(512, 2) void store_kernel(unsigned short* C, long ldc, int gm, int gn, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int bm = t / gn, bn = t - bm * gn;
    const long m0 = (long)bm * 256, n0 = (long)bn * 256;
    u32x4 v = {(unsigned)t, (unsigned)lane, 3u, 4u};
    if (SHAPE == 0) {
      const int l15 = lane & 15, g4 = lane >> 4;
      const int cs = (g4 & 1) * 16 + (g4 >> 1) * 8;
#pragma unroll
      for (int sh = 0; sh < 2; ++sh)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          unsigned short* p = C + (m0 + (j >> 2) * 128 + wr * 64 + (j & 3) * 16 + l15) * ldc + n0 + sh * 128 + wc * 32 + cs;
          if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
        }
    } else if (SHAPE == 2 || SHAPE == 3) {
      // wave = rows (wave * 32 ..) x 128 columns (256 B) of each 128-column block cb of the tile: 2 column blocks x 8 waves x 32 rows = the 256 x 256 tile
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          unsigned short* p = SHAPE == 2 ? C + (m0 + wave * 32 + (lane & 31)) * ldc + n0 + cb * 128 + i * 16 + (lane >> 5) * 8
                                         : C + (m0 + wave * 32 + i * 4 + (lane >> 4)) * ldc + n0 + cb * 128 + (lane & 15) * 8;
          if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
        }
    } else {
      // the wave's 64 contiguous columns wc * 64 .., rows wr * 128 + 8 i + (lane >> 3)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        unsigned short* p = C + (m0 + wr * 128 + i * 8 + (lane >> 3)) * ldc + n0 + wc * 64 + (lane & 7) * 8;
        if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
      }
    }
  }
}

template <int SHAPE, bool NT>
float run(unsigned short* C, int M, int N, int iters) {
  const int gm = M / 256, gn = N / 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  store_kernel<SHAPE, NT><<<256, 512>>>(C, N, gm, gn, gm * gn);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) store_kernel<SHAPE, NT><<<256, 512>>>(C, N, gm, gn, gm * gn);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int M = 131072;
  for (int N : {1024, 4096, 6144}) {
    unsigned short* C; hipMalloc(&C, (size_t)M * N * 2);
    const double gb = (double)M * N * 2 / 1e9;
    for (int rep = 0; rep < 2; ++rep) {
      float a = run<0, true>(C, M, N, 5), b = run<1, true>(C, M, N, 5), c = run<0, false>(C, M, N, 5), d = run<1, false>(C, M, N, 5);
      float e = run<2, false>(C, M, N, 5), f = run<3, false>(C, M, N, 5);
      printf("N=%5d           : 32 rows x 32 B plain %.3f ms (%.0f GB/s) | 4 rows x 256 B plain %.3f ms (%.0f GB/s)\n", N, e, gb / e * 1e3, f, gb / f * 1e3);
      printf("N=%5d (%.2f GB): 16 rows x 64 B nt %.3f ms (%.0f GB/s) | 8 rows x 128 B nt %.3f ms (%.0f GB/s) | 16x64 plain %.3f (%.0f) | 8x128 plain %.3f (%.0f)\n",
             N, gb, a, gb / a * 1e3, b, gb / b * 1e3, c, gb / c * 1e3, d, gb / d * 1e3);
    }
    hipFree(C);
  }
  return 0;
}

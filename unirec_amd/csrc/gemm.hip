// bf16 MFMA GEMM for every Linear on the hot path (forward, dX and dW), gfx950 only.
//
//   C[m][n] = alpha * ( sum_k R(m,k) * S(n,k)  +  sum_k2 R2(m,k2) * S2(n,k2) )   (+ epilogue)
//
// R gives C's rows, S gives C's columns.  Each operand is either "K-contiguous" (stored [rows][K],
// the nn.Linear [out,in] weight layout and the activation layout) or "K-strided" (stored [K][rows]),
// so all three Linear products run on tensors exactly as they lie in HBM -- no transposed copies:
//   forward  Y  = X W^T      : R = X  [M,K]  kc ;  S = W  [N,K]  kc
//   dX       dX = dY W       : R = dY [M,N]  kc ;  S = W  [N(red),K'] k-strided
//   dW       dW = dY^T X     : R = dY [M(red),N] k-strided ; S = X [M(red),K'] k-strided
// The second (R2,S2,K2) range is the LoRA low-rank term: R2 = x A^T (rank r), S2 = B, so the
// B-product is accumulated in the same MFMA accumulators as the base GEMM (no extra pass over Y).
//
// Tiling: 256x256x64 block tile with 8 waves (2x4, each wave 128x64 = 8x4 v_mfma_f32_16x16x32_bf16
// tiles) for the large projections, 128x128x64 with 4 waves (2x2) for small / edge shapes.  The weight-side operand S is the MFMA "A" (row) operand and the
// token-side operand R the "B" (column) operand, so each lane ends up with 4 CONSECUTIVE n for one
// m: packed 8-byte (bf16) / 16-byte (f32) stores into row-major C.
// Staging: LDS-DMA (global_load_lds_dwordx4) straight into a 2-slot ring of XOR-swizzled 64-deep LDS images
// (no staging VGPRs, no ds_write): every wave instruction moves 8 full 128-byte lines; tile t+1 lands and
// tile t+2 is issued under tile t's 64 MFMAs per wave (one barrier per tile, between its two k-halves).
// K-contiguous tiles are read with ds_read_b128, K-strided tiles with ds_read_b64_tr_b16 (hardware
// transpose); both images are bank-conflict-free.
#include <cstdlib>
#include <type_traits>
#include "common.hip.h"
#include "unirec_hip.h"
#include "gemm_common.hip.h"

using urgemm::GemmP;
using urgemm::GemmGroups;
using urgemm::GemmGroupSlot;
using urgemm::UR_GEMM_MAX_GROUPS;
using urgemm::uniform_ptr;

namespace {

#ifndef UR_GEMM_NO_PH8
#define UR_GEMM_NO_PH8 0          // lab builds only: 1 = keep the grouped 2-slot loop for the 256x256 tile (A/B against the 8-phase loop)
#endif
#ifndef UR_GEMM_STAMPS
#define UR_GEMM_STAMPS 0          // lab builds only: 1 = thread 0 of every workgroup logs s_memtime at 8 points (ur_lab_gemm_stamps)
#endif
#if UR_GEMM_STAMPS
__device__ long long g_gemm_stamps[8192 * 8];
#define UR_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192 && blockIdx.z == 0) g_gemm_stamps[blockIdx.x * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define UR_STAMP(k) do { } while (0)
#endif
#ifndef UR_GEMM_ABLATE
#define UR_GEMM_ABLATE 0          // lab builds only (tools/lab): 1 = no LDS-DMA in the steady state, 2 = no MFMAs, 3 = no barrier
#endif
constexpr int BK = 64;                     // K depth of one LDS stage = two v_mfma_f32_16x16x32_bf16 k-steps ("halves")
constexpr int NSTAGE = 2;                  // LDS ring: tile t is consumed while tile t+1 lands and tile t+2 is issued
constexpr int KC_ROWB = BK * 2;            // K-contiguous tile row bytes: one full 128-byte cache line per row
// LDS image of one operand tile of T rows (T = 128 or 256), no padding (LDS-DMA writes linearly):
//   K-contiguous  [T][64 k] : 128-B rows, 16-B chunk c of row r stored at chunk c ^ g(r), g(r) = (r >> 1) & 7
//                             -> conflict-free ds_read_b128 fragment reads (16 rows x 4 chunks per read), and one
//                                LDS-DMA wave instruction moves 8 rows x 128 B = 8 FULL cache lines (a 32-deep
//                                tile moves 16 half lines per instruction: 36 vs 49 B/clk/CU from L2, tools/lab)
//   K-strided     [64 k][T] : (T = 256: TWO images of 128 columns each, one after the other -- a half tile is then one contiguous
//                             16 KiB region, as in the K-contiguous image, which is what the 8-phase loop refills and reads)
//                             2T-byte rows, 32-B segment s of k-row r stored at segment s ^ f(r),
//                             f(r) = (r & 3) | (((r >> 3) & 1) << 2)
//                             -> the 8 (k-row, 32-B) pieces one half-wave ds_read_b64_tr_b16 touches
//                                land on 8 different 32-B bank groups: conflict-free transposed reads
template <int T> struct Tile {
  static constexpr int KC_BYTES = T * KC_ROWB;
  static constexpr int KS_T = T == 256 ? 128 : T;      // columns of one K-strided image
  static constexpr int KS_ROWB = KS_T * 2;
  static constexpr int KS_IMG = BK * KS_ROWB;          // bytes of one image
  static constexpr int KS_BYTES = BK * T * 2;
};
__device__ __forceinline__ int ks_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
__device__ __forceinline__ int kc_g(int r) { return (r >> 1) & 7; }

// ---- LDS-DMA staging of a FULL 64-deep tile: 1 KiB per wave instruction, swizzle on the source ---
// rows/cols past the matrix edge are clamped (they only feed output rows/cols that are never stored)
template <bool KC, int T, int NT>
__device__ __forceinline__ void dma_tile(char* tile, const bf16_t* __restrict__ base, long ld, int rows_total, int row0,
                                         int k0, int tid) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
  constexpr int PIECES = T * BK * 2 / 1024;     // 1 KiB pieces per tile
  constexpr int PER_WAVE = PIECES / (NT / 64);
  static_assert(PER_WAVE >= 1, "tile too small for the workgroup");
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int inst = i * (NT / 64) + wave;
    const bf16_t* src;
    if (KC) {
      const int row = inst * 8 + (lane >> 3), pos = lane & 7;
      const int g = min(row0 + row, rows_total - 1);
      src = base + (long)g * ld + k0 + ((pos ^ kc_g(row)) << 3);
    } else {
      constexpr int TH = Tile<T>::KS_T;
      const int c = inst * 64 + lane;
      const int img = c / (TH * BK / 8), cc = c % (TH * BK / 8);
      const int kr = cc / (TH / 8), ch = cc % (TH / 8);
      const int col = min(row0 + img * TH + ((ch ^ (ks_f(kr) << 1)) << 3), rows_total - 8);
      src = base + (long)(k0 + kr) * ld + col;
    }
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(tile + inst * 1024), 16, 0, 0);
  }
}

// Steady-state LDS-DMA: the lane's source offset inside the block's operand panel does not change from
// tile to tile (only the uniform K position does), so it is computed once (dma_offsets) and a full tile
// costs no vector arithmetic: every piece is  uniform base + 32-bit lane offset.
template <bool KC, int T, int NT>
__device__ __forceinline__ void dma_offsets(uint32_t (&voff)[(T * BK * 2 / 1024) / (NT / 64)], long ld, int rows_total, int row0, int tid) {
  constexpr int PER_WAVE = (T * BK * 2 / 1024) / (NT / 64);
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int inst = i * (NT / 64) + wave;
    long e;                                     // element offset from base + (KC ? row0 * ld : row0)
    if (KC) {
      const int row = inst * 8 + (lane >> 3), pos = lane & 7;
      const int g = min(row0 + row, rows_total - 1) - row0;
      e = (long)g * ld + ((pos ^ kc_g(row)) << 3);
    } else {
      constexpr int TH = Tile<T>::KS_T;
      const int c = inst * 64 + lane;
      const int img = c / (TH * BK / 8), cc = c % (TH * BK / 8);
      const int kr = cc / (TH / 8), ch = cc % (TH / 8);
      const int col = min(row0 + img * TH + ((ch ^ (ks_f(kr) << 1)) << 3), rows_total - 8) - row0;
      e = (long)kr * ld + col;
    }
    voff[i] = (uint32_t)(e * 2);
  }
}

// ---- register staging: partial K tiles (zero-fill past kend) ------------------------------------
template <bool KC, int T, int NT>
__device__ __forceinline__ void reg_tile(char* tile, const bf16_t* __restrict__ base, long ld, int rows_total, int row0,
                                         int k0, int kend, int tid) {
  constexpr int CHUNKS = T * BK / 8;
#pragma unroll
  for (int i = 0; i < (CHUNKS + NT - 1) / NT; ++i) {
    const int c = tid + i * NT;
    if (c >= CHUNKS) break;
    uint4 z = make_uint4(0, 0, 0, 0);
    int off;
    if (KC) {
      const int row = c >> 3, kc = c & 7;
      const int grow = min(row0 + row, rows_total - 1), gk = k0 + kc * 8;
      if (gk < kend) z = *reinterpret_cast<const uint4*>(base + (long)grow * ld + gk);
      off = row * KC_ROWB + ((kc ^ kc_g(row)) << 4);
    } else {
      const int kr = c / (T / 8), ch = c % (T / 8);
      const int gk = k0 + kr, gcol = min(row0 + ch * 8, rows_total - 8);
      if (gk < kend) z = *reinterpret_cast<const uint4*>(base + (long)gk * ld + gcol);
      constexpr int TH8 = Tile<T>::KS_T / 8;
      off = (ch / TH8) * Tile<T>::KS_IMG + kr * Tile<T>::KS_ROWB + (((ch % TH8) ^ (ks_f(kr) << 1)) << 4);
    }
    *reinterpret_cast<uint4*>(tile + off) = z;
  }
}

// ---- LDS -> MFMA fragments of k-half h (k = 32h .. 32h+31 of the stage):
//      lane holds [idx = idx0 + 16*i + (lane&15)][k = 32h + 8*(lane>>4) + 0..7] ---------------------------------
template <bool KC, int T, int N>
__device__ __forceinline__ void lds_frags(bf16x8* f, const char* tile, int idx0, int h, int lane) {
  if (KC) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int idx = idx0 + 16 * i + (lane & 15);
      f[i] = *reinterpret_cast<const bf16x8*>(tile + idx * KC_ROWB + (((4 * h + (lane >> 4)) ^ kc_g(idx)) << 4));
    }
  } else {
    static_assert(N == 1 || (N % 2) == 0, "transposed fragment reads go in groups of 4 or 2");
    if (T == 256 && idx0 >= 128) { tile += Tile<T>::KS_IMG; idx0 -= 128; }      // the second image (a group of fragments never straddles)
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int ka = 32 * h + 8 * g + q;                  // k-rows ka (elements 0..3) and ka+4 (elements 4..7)
    const uint32_t ra = lds_off(tile) + ka * Tile<T>::KS_ROWB + pp * 8, rb = ra + 4 * Tile<T>::KS_ROWB;
    const int fa = ks_f(ka), fb = ks_f(ka + 4);
#pragma unroll
    for (int i0 = 0; i0 + 4 <= N; i0 += 4) {
      uint32_t a[4], b[4];
      bf16x8 t4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int seg = (idx0 >> 4) + i0 + i;           // 32-byte segment of this 16-column block
        a[i] = ra + ((seg ^ fa) << 5);
        b[i] = rb + ((seg ^ fb) << 5);
      }
      tr_read(t4, a, b);
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i0 + i] = t4[i];
    }
    if ((N % 4) == 2) {
      constexpr int i0 = N - 2;
      uint32_t a[2], b[2];
      bf16x8 t2[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int seg = (idx0 >> 4) + i0 + i;
        a[i] = ra + ((seg ^ fa) << 5);
        b[i] = rb + ((seg ^ fb) << 5);
      }
      tr_read(t2, a, b);
      f[i0] = t2[0]; f[i0 + 1] = t2[1];
    }
  }
}

// BM x BN block tile, NWM x NWN waves; each wave owns (BM/NWM) rows x (BN/NWN) columns of C.
// EPI: 1 = SwiGLU backward epilogue (ur_gemm_args.swiglu_gu), 2 = SwiGLU forward epilogue (ur_gemm_args.swiglu_gate), each its
// own instantiation, so the ordinary kernels' code and register allocation do not change with them.
// MFMA with the accumulator tile in AccVGPRs (inline asm) for the 256x256 kernel: under -amdgpu-mfma-vgpr-form hipcc keeps all
// 128 accumulator registers of a wave in arch VGPRs, which leaves the 8-phase loop exactly at the 256-register limit (and made
// every attempt to wrap the body in a tile loop spill).  With "+a" they live in the other half of the unified file.
#ifndef UR_GEMM_ACC_AGPR
#define UR_GEMM_ACC_AGPR 0      // lab: at two waves per SIMD the unified file gives a wave 256 registers in TOTAL, so 128 AccVGPRs leave 128 arch VGPRs and the loop spills (468 B scratch): off
#endif
template <bool ACC_A>
__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (ACC_A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

template <bool RK, bool SK, bool OUTF32, int BM, int BN, int NWM, int NWN, int EPI = 0>
__global__ __launch_bounds__(NWM * NWN * 64, 2) void gemm_kernel(GemmP p) {
#define UR_GEMM_BID ((int)blockIdx.x)
#include "gemm_body.hip.h"
#undef UR_GEMM_BID
}

// Grouped launch (ur_gemm_grouped): up to UR_GEMM_MAX_GROUPS independent products of the SAME kind (operand layouts, output type, K, split)
// and different M / N / pointers in ONE grid -- the weight-gradient token reductions of one Q-Former layer (five to seven [out, in] matrices
// of 9-36 big tiles each: alone none of them fills 256 CUs without slicing K into pieces too short to pay for a tile's prologue).  Group g
// owns the workgroups wg0[g] .. wg0[g + 1] (runs padded to multiples of 8: blockIdx.x % 8 stays the XCD of the tile order); the padding
// workgroups leave at once.  Everything the body reads per group is patched into a copy of the launch's parameters (scalar registers).
template <bool RK, bool SK, bool OUTF32, int BM, int BN, int NWM, int NWN>
__global__ __launch_bounds__(NWM * NWN * 64, 2) void gemm_grouped_kernel(GemmP pk, GemmGroups gs) {
  int g = 0;
  for (int i = 1; i < gs.n; ++i)
    if ((int)blockIdx.x >= gs.g[i].wg0) g = i;
  const GemmGroupSlot& s = gs.g[g];
  GemmP p = pk;
  p.R = s.R; p.S = s.S; p.C = s.C; p.ldr = s.ldr; p.lds = s.lds; p.ldc = s.ldc; p.slab_stride = s.slab_stride;
  p.M = s.M; p.N = s.N; p.gm = s.gm; p.gn = s.gn;
  const int bid = (int)blockIdx.x - s.wg0;
  if (bid >= p.gm * p.gn) return;                    // (uniform: before any barrier)
  constexpr int EPI = 0;
#define UR_GEMM_BID bid
#include "gemm_body.hip.h"
#undef UR_GEMM_BID
}

// deterministic split-K combine: C[m][n] = sum_z slab[z][m][n]  (f32, vectorised)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, long total4, long slab4,
                                     int splits) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const float4* w4 = reinterpret_cast<const float4*>(ws);
  float4* c4 = reinterpret_cast<float4*>(C);
  for (; i < total4; i += stride) {
    float4 a = w4[i];
    for (int z = 1; z < splits; ++z) {
      float4 b = w4[i + (long)z * slab4];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    c4[i] = a;
  }
}

template <bool RK, bool SK, bool OUTF32, int BM, int BN, int NWM, int NWN, int EPI = 0>
int launch_cfg(GemmP p, int splits, hipStream_t st) {
  constexpr int S_BYTES = SK ? Tile<BN>::KC_BYTES : Tile<BN>::KS_BYTES;
  constexpr int R_BYTES = RK ? Tile<BM>::KC_BYTES : Tile<BM>::KS_BYTES;
  constexpr int RING = NSTAGE * (S_BYTES + R_BYTES), CTILE = BM * (BN * 2 + 16);      // bf16 tile == f32 half tile rows
  constexpr int SMEM = OUTF32 ? RING : (RING > CTILE ? RING : CTILE);
  static std::atomic<uint64_t> attr_set{0};   // per device
  UR_ONCE_PER_DEVICE(attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<RK, SK, OUTF32, BM, BN, NWM, NWN, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) UR_FAIL((int)e, "ur_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  }
  p.gm = ur_cdiv(p.M, BM); p.gn = ur_cdiv(p.N, BN);
  {
    // UR_GEMM_CW = n (lab): walk each XCD's tiles in column chunks of n tiles.  With K = 1024 a tile's S panel is 512 KiB: an
    // XCD's 32 concurrent tiles over 12 column tiles keep 6 MiB of S panels in play against a 4 MiB L2 and every tile
    // re-fetches its panel from the Infinity Cache; chunks of 4 halve the fabric reads of the N = 3072 launches (FETCH_SIZE
    // 2.24 -> 1.20 GB for gate_proj, profiles/r1_gemm_pmc.json) -- and cost 1 % of the joint step in alternating same-box
    // runs (118.8 vs 117.5 seq/s), so the plain row-major runs stay the default.
    // Round 2, merged launches (q|k|v: 16 column tiles, gate|up: 24; profiles/r2_gemm_pmc.json): the fabric reads reach
    // 2.4 / 5.9 GB per launch (x8.7 / x21 of A + W, ~4 TB/s) and chunks of 4 are 2.1 / 3.6 % faster in isolation
    // (tools/kernel_bench.py gemm_step), while launches of <= 12 column tiles still lose 1-3 %.  Inside the joint step,
    // alternating same-box runs: chunks of 4 on the two wide launches only 516.6 vs 517.0 ms (nothing), chunks of 4 wherever
    // they divide 551.8 vs 545.8 ms (+1.1 %) -- the plain order stays the default (UR_GEMM_CW = n: lab).
    static const int env_cw = ur_lab_int("UR_GEMM_CW", -1);
    p.gcw = 0;
    // default: chunks of 4 on launches of >= 16 column tiles (the merged q|k|v and gate|up forwards) -- time-neutral inside the
    // step, but the fabric reads of those launches drop from x8.7 / x21 of A + W to what profiles/r2_gemm_pmc.json lists
    const int cw = env_cw >= 0 ? env_cw : (p.gn >= 16 ? 4 : 0);
    if (cw > 0 && BM == 256 && (p.gm % 8) == 0 && p.gn > cw && (p.gn % cw) == 0) p.gcw = cw;
  }
  {
    static const int env_st = ur_lab_int("UR_GEMM_STAGGER", 0);     // lab
    p.stagger = (BM == 256 && p.gm * p.gn >= 1024) ? env_st : 0;
  }
  dim3 grid(p.gm * p.gn, 1, splits);
  hipLaunchKernelGGL((gemm_kernel<RK, SK, OUTF32, BM, BN, NWM, NWN, EPI>), grid, dim3(NWM * NWN * 64), SMEM, st, p);
  UR_CHECK_LAUNCH("ur_gemm");
  return 0;
}

// Tile choice: 256x256 (8 waves, 130 FLOP per byte of L2 traffic) once the grid still fills the
// chip (>= 256 workgroups); otherwise the 128x128 tile (4 waves, 2 workgroups per CU).
template <bool RK, bool SK, bool OUTF32>
int launch(const GemmP& p, int splits, hipStream_t st) {
  const long big_wgs = (long)ur_cdiv(p.M, 256) * ur_cdiv(p.N, 256) * splits;
  static const bool force128 = ur_lab_int("UR_GEMM_FORCE128", 0) == 1;      // lab: 128x128 tiles (2 workgroups per CU) everywhere
  if constexpr (RK && SK && !OUTF32) {
    if (urgemm::gemm_pers_eligible(p, splits, RK, SK, OUTF32)) return urgemm::gemm_pers_launch(p, st);    // gemm_pers.hip
    if (p.sw_gu && p.sw_mode == 1) {       // SwiGLU backward epilogue: K-contiguous bf16 launches only (ur_gemm checks)
      if (p.M >= 256 && p.N >= 256 && big_wgs >= 256 && !force128) return launch_cfg<true, true, false, 256, 256, 2, 4, 1>(p, splits, st);
      return launch_cfg<true, true, false, 128, 128, 2, 2, 1>(p, splits, st);
    }
    if (p.sw_gu && p.sw_mode == 2) {       // SwiGLU forward epilogue
      if (p.M >= 256 && p.N >= 256 && big_wgs >= 256) return launch_cfg<true, true, false, 256, 256, 2, 4, 2>(p, splits, st);
      return launch_cfg<true, true, false, 128, 128, 2, 2, 2>(p, splits, st);
    }
  }
  // (token-reduction launches, both operands K-strided: one round of 224+ big tiles already beats the small tile -- the host picks
  // such splits, qformer.py:_split_k_for)
  const long big_min = (!RK && !SK) ? 224 : 256;
  if (p.M >= 256 && p.N >= 256 && big_wgs >= big_min && !force128) return launch_cfg<RK, SK, OUTF32, 256, 256, 2, 4>(p, splits, st);
  return launch_cfg<RK, SK, OUTF32, 128, 128, 2, 2>(p, splits, st);
}

// split-K combine of a grouped launch: blockIdx.y = group
struct ReduceGroups { int n, pad_; struct { const float* ws; float* C; long total4; } g[UR_GEMM_MAX_GROUPS]; };
__global__ void splitk_reduce_grouped_kernel(ReduceGroups rg, int splits) {
  const float4* w4 = reinterpret_cast<const float4*>(rg.g[blockIdx.y].ws);
  float4* c4 = reinterpret_cast<float4*>(rg.g[blockIdx.y].C);
  const long total4 = rg.g[blockIdx.y].total4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    float4 a = w4[i];
    for (int z = 1; z < splits; ++z) {
      const float4 b = w4[i + (long)z * total4];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    c4[i] = a;
  }
}

// token-major operands, f32 output (the weight-gradient products): the only kind ur_gemm_grouped takes
template <int BM, int BN, int NWM, int NWN>
int launch_grouped_cfg(const GemmP& p, const GemmGroups& gs, int nwg, int splits, hipStream_t st) {
  constexpr int SMEM = NSTAGE * (Tile<BN>::KS_BYTES + Tile<BM>::KS_BYTES);
  static std::atomic<uint64_t> attr_set{0};   // per device
  UR_ONCE_PER_DEVICE(attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_grouped_kernel<false, false, true, BM, BN, NWM, NWN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) UR_FAIL((int)e, "ur_gemm_grouped: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL((gemm_grouped_kernel<false, false, true, BM, BN, NWM, NWN>), dim3(nwg, 1, splits), dim3(NWM * NWN * 64), SMEM, st, p, gs);
  UR_CHECK_LAUNCH("ur_gemm_grouped");
  return 0;
}

}  // namespace

static inline bool splits_ok(const ur_gemm_args* a) { return a->split_k <= 1; }

#if UR_GEMM_STAMPS
extern "C" int ur_lab_gemm_stamps(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_stamps), sizeof(long long) * n);
}
#endif

static void fill_params(const ur_gemm_args* a, GemmP& p) {
  p.R = (const bf16_t*)a->R; p.S = (const bf16_t*)a->S; p.ldr = a->ldr; p.lds = a->lds; p.K = a->K;
  p.R2 = (const bf16_t*)a->R2; p.S2 = (const bf16_t*)a->S2; p.ldr2 = a->ldr2; p.lds2 = a->lds2; p.K2 = a->K2;
  p.C = a->C; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.alpha = a->alpha;
  p.bias = a->bias; p.res = (const bf16_t*)a->residual; p.ldres = a->ldres;
  p.gelu_out = (bf16_t*)a->gelu_out; p.ldg = a->ldg; p.aux = (const bf16_t*)a->gelu_grad_aux; p.ldaux = a->ldaux;
  p.gm = 0; p.gn = 0;   // set by launch_cfg for the chosen tile
  p.drop_bits = (const uint8_t*)a->drop_bits; p.drop_bits_ld = a->drop_bits_ld; p.drop_bits_stride = a->drop_bits_stride;
  p.drop_rank = a->drop_rank;
  p.drop_inv_keep = a->drop_bits ? 1.0f / (1.0f - a->drop_p) : 1.0f;
  p.sw_gu = (const bf16_t*)a->swiglu_gu; p.sw_ldgu = a->swiglu_ldgu; p.sw_dgu = (bf16_t*)a->swiglu_dgu; p.sw_lddgu = a->swiglu_lddgu;
  p.sw_I = a->swiglu_I; p.sw_mode = a->swiglu_gu ? 1 : 0;
  if (a->swiglu_gate) {
    p.sw_gu = (const bf16_t*)a->swiglu_gate; p.sw_ldgu = a->swiglu_ldgate; p.sw_dgu = (bf16_t*)a->swiglu_act; p.sw_lddgu = a->swiglu_ldact;
    p.sw_I = 0; p.sw_mode = 2;
  }
  p.qk_q = (bf16_t*)a->qkr_q; p.qk_ldq = a->qkr_ldq; p.qk_k = (bf16_t*)a->qkr_k; p.qk_ldk = a->qkr_ldk; p.qk_v = (bf16_t*)a->qkr_v; p.qk_ldv = a->qkr_ldv;
  p.qk_rstd = a->qkr_rstd; p.qk_qw = a->qkr_qw; p.qk_kw = a->qkr_kw; p.qk_cos = a->qkr_cos; p.qk_sin = a->qkr_sin;
  p.qk_S = a->qkr_S; p.qk_nq = a->qkr_nq_cols; p.qk_nk = a->qkr_nk_cols; p.qk_eps = a->qkr_eps;
  p.sp_act = (bf16_t*)a->swp_act; p.sp_ldact = a->swp_ldact; p.sp_I = a->swp_I;
  p.ksplit_len = 0; p.slab_stride = 0; p.gcw = 0; p.stagger = 0;
}

static bool swiglu_paired_ok(const ur_gemm_args* a, const GemmP& p) {
  if (!a->swp_act || a->swp_I <= 0 || a->N != 2 * a->swp_I || (a->swp_I % 128)) return false;
  if (!a->r_kcontig || !a->s_kcontig || a->c_f32 || a->split_k > 1) return false;
  if (a->bias || a->residual || a->gelu_out || a->gelu_grad_aux || a->drop_bits || a->swiglu_gu || a->swiglu_gate || a->qkr_q || a->alpha != 1.0f) return false;
  if ((a->swp_ldact & 7) || a->swp_ldact < a->swp_I || !UR_ALIGNED16(a->swp_act) || a->ldc < a->N) return false;
  return urgemm::gemm_pers_eligible(p, 1, true, true, false);
}

extern "C" int ur_gemm_swiglu_paired_supported(const ur_gemm_args* a) {
  if (!a || a->M <= 0 || a->N <= 0 || !a->R || !a->S) return 0;
  GemmP p;
  fill_params(a, p);
  return swiglu_paired_ok(a, p) ? 1 : 0;
}

// the q/k-norm + RoPE epilogue exists on the persistent kernel only: everything ur_gemm_args.qkr_* promises, checked once
static bool qkrope_ok(const ur_gemm_args* a, const GemmP& p) {
  if (!a->qkr_q || !a->qkr_k || !a->qkr_v || !a->qkr_rstd || !a->qkr_qw || !a->qkr_kw || !a->qkr_cos || !a->qkr_sin) return false;
  if (!a->r_kcontig || !a->s_kcontig || a->c_f32 || a->split_k > 1) return false;
  if (a->bias || a->residual || a->gelu_out || a->gelu_grad_aux || a->drop_bits || a->swiglu_gu || a->swiglu_gate || a->alpha != 1.0f) return false;
  if (a->qkr_S < 256 || (a->qkr_S % 256) || (a->qkr_nq_cols % 256) || (a->qkr_nk_cols % 256) || a->qkr_nq_cols < 0 || a->qkr_nk_cols < 0 ||
      a->qkr_nq_cols + a->qkr_nk_cols > a->N || ((a->N - a->qkr_nq_cols - a->qkr_nk_cols) % 256)) return false;
  if ((a->qkr_ldq & 7) || (a->qkr_ldk & 7) || (a->qkr_ldv & 7) || !UR_ALIGNED16(a->qkr_q) || !UR_ALIGNED16(a->qkr_k) || !UR_ALIGNED16(a->qkr_v) ||
      !UR_ALIGNED16(a->qkr_qw) || !UR_ALIGNED16(a->qkr_kw) || !UR_ALIGNED16(a->qkr_cos) || !UR_ALIGNED16(a->qkr_sin)) return false;
  if (a->qkr_ldq < a->qkr_nq_cols || a->qkr_ldk < a->qkr_nk_cols || a->qkr_ldv < a->N - a->qkr_nq_cols - a->qkr_nk_cols) return false;
  return urgemm::gemm_pers_eligible(p, 1, true, true, false);
}

extern "C" int ur_qkrope_perm(int c) { return ((c >> 4) & 1) * 64 + ((c >> 5) & 3) * 16 + (c & 15); }

extern "C" int ur_gemm_qkrope_supported(const ur_gemm_args* a) {
  if (!a || a->M <= 0 || a->N <= 0 || !a->R || !a->S) return 0;
  GemmP p;
  fill_params(a, p);
  return qkrope_ok(a, p) ? 1 : 0;
}

extern "C" int64_t ur_gemm_workspace_bytes(const ur_gemm_args* a) {
  if (!a || a->split_k <= 1) return 0;
  return (int64_t)a->split_k * a->M * a->ldc * (int64_t)sizeof(float);
}

extern "C" int ur_gemm(const ur_gemm_args* a, void* workspace, int64_t workspace_bytes, void* stream) {
  UR_REQUIRE(a != nullptr, "ur_gemm: null args");
  UR_REQUIRE(a->M >= 0 && a->N >= 0 && a->K >= 0 && a->K2 >= 0, "ur_gemm: negative dimension");
  if (a->M == 0 || a->N == 0) return 0;
  UR_REQUIRE(a->R && a->S && a->C, "ur_gemm: null operand");
  // a K-contiguous operand is staged in 16-byte chunks along K; K-strided operands put K on rows
  UR_REQUIRE((!a->r_kcontig && !a->s_kcontig) || ((a->K % 8) == 0 && (a->K2 % 8) == 0),
             "ur_gemm: K (%d) and K2 (%d) must be multiples of 8 for K-contiguous operands", a->K, a->K2);
  UR_REQUIRE((a->N % 4) == 0 && (a->ldc % 4) == 0, "ur_gemm: N (%d) and ldc (%ld) must be multiples of 4", a->N, (long)a->ldc);
  UR_REQUIRE((a->ldr % 8) == 0 && (a->lds % 8) == 0, "ur_gemm: ldr/lds must be multiples of 8 elements");
  UR_REQUIRE(a->r_kcontig || (a->M % 8) == 0, "ur_gemm: K-strided R needs M %% 8 == 0 (M=%d)", a->M);
  UR_REQUIRE(a->s_kcontig || (a->N % 8) == 0, "ur_gemm: K-strided S needs N %% 8 == 0 (N=%d)", a->N);
  UR_REQUIRE(UR_ALIGNED16(a->R) && UR_ALIGNED16(a->S) && UR_ALIGNED16(a->C), "ur_gemm: operands must be 16-byte aligned");
  UR_REQUIRE(a->ldr >= (a->r_kcontig ? a->K : a->M) && a->lds >= (a->s_kcontig ? a->K : a->N) && (a->ldc >= a->N || a->qkr_q),
             "ur_gemm: leading dimension smaller than row length");
  if (a->K2 > 0) {
    UR_REQUIRE(a->R2 && a->S2 && UR_ALIGNED16(a->R2) && UR_ALIGNED16(a->S2), "ur_gemm: bad second operand pair");
    UR_REQUIRE((a->ldr2 % 8) == 0 && (a->lds2 % 8) == 0, "ur_gemm: ldr2/lds2 must be multiples of 8");
    UR_REQUIRE(a->ldr2 >= (a->r_kcontig ? a->K2 : a->M) && a->lds2 >= (a->s_kcontig ? a->K2 : a->N),
               "ur_gemm: second-pair leading dimension too small");
  }
  const int splits = a->split_k > 1 ? a->split_k : 1;
  if (splits > 1) {
    UR_REQUIRE(a->c_f32 && !a->bias && !a->residual && !a->gelu_out && !a->gelu_grad_aux && a->K2 == 0,
               "ur_gemm: split_k needs f32 output and no epilogue / second pair");
    UR_REQUIRE(workspace && workspace_bytes >= ur_gemm_workspace_bytes(a) && UR_ALIGNED16(workspace),
               "ur_gemm: split_k workspace too small (%lld < %lld)", (long long)workspace_bytes,
               (long long)ur_gemm_workspace_bytes(a));
  }
  if (a->c_f32) {
    UR_REQUIRE(!a->residual && !a->gelu_out && !a->gelu_grad_aux, "ur_gemm: f32 output supports alpha/bias only");
  } else {
    UR_REQUIRE(!a->residual || ((a->ldres % 4) == 0 && (((uintptr_t)a->residual) & 7) == 0), "ur_gemm: residual misaligned");
    UR_REQUIRE(!a->gelu_out || ((a->ldg % 4) == 0 && (((uintptr_t)a->gelu_out) & 7) == 0), "ur_gemm: gelu_out misaligned");
    UR_REQUIRE(!a->gelu_grad_aux || ((a->ldaux % 4) == 0 && (((uintptr_t)a->gelu_grad_aux) & 7) == 0), "ur_gemm: aux misaligned");
  }
  UR_REQUIRE(!a->bias || UR_ALIGNED16(a->bias), "ur_gemm: bias must be 16-byte aligned");
  if (a->swiglu_gate) {
    UR_REQUIRE(!a->swiglu_gu && !a->c_f32 && splits <= 1 && !a->residual && !a->gelu_out && !a->gelu_grad_aux && a->r_kcontig && a->s_kcontig,
               "ur_gemm: the SwiGLU forward epilogue needs K-contiguous operands, bf16 output, no split_k / residual / gelu modes");
    UR_REQUIRE(a->swiglu_act && (a->swiglu_ldgate % 4) == 0 && (a->swiglu_ldact % 4) == 0 && a->swiglu_ldgate >= a->N && a->swiglu_ldact >= a->N &&
               (((uintptr_t)a->swiglu_gate) & 7) == 0 && (((uintptr_t)a->swiglu_act) & 7) == 0,
               "ur_gemm: SwiGLU forward epilogue: gate / act rows of >= N elements, 8-byte aligned");
  }
  if (a->swiglu_gu) {
    UR_REQUIRE(!a->c_f32 && splits <= 1 && !a->residual && !a->gelu_out && !a->gelu_grad_aux && !a->bias && a->r_kcontig && a->s_kcontig,
               "ur_gemm: the SwiGLU backward epilogue needs K-contiguous operands, bf16 output, no split_k / bias / residual / gelu modes");
    UR_REQUIRE(a->swiglu_dgu && a->swiglu_I == a->N && (a->swiglu_I % 4) == 0 && (a->swiglu_ldgu % 4) == 0 && (a->swiglu_lddgu % 4) == 0 &&
               a->swiglu_ldgu >= 2 * (int64_t)a->swiglu_I && a->swiglu_lddgu >= 2 * (int64_t)a->swiglu_I &&
               (((uintptr_t)a->swiglu_gu) & 7) == 0 && (((uintptr_t)a->swiglu_dgu) & 7) == 0,
               "ur_gemm: SwiGLU backward epilogue: N must equal swiglu_I, gu / dgu rows of >= 2 I elements, 8-byte aligned");
  }
  if (a->drop_bits) {
    UR_REQUIRE(a->K2 > 0 && a->drop_rank >= 8 && a->drop_rank <= 32 && (a->drop_rank % 8) == 0 && (a->K2 % a->drop_rank) == 0 &&
               a->K2 / a->drop_rank <= 4 && a->r_kcontig && a->s_kcontig && splits_ok(a) && !a->c_f32,
               "ur_gemm: the masked LoRA epilogue needs K-contiguous operands, bf16 output, no split_k, rank in {8,16,24,32} and at most 4 adapters");
    UR_REQUIRE(a->drop_p >= 0.f && a->drop_p < 1.f && (a->drop_bits_ld % 16) == 0 && a->drop_bits_ld * 8 >= a->N && (a->drop_bits_stride % 8) == 0 &&
               ((uintptr_t)a->drop_bits & 15) == 0, "ur_gemm: bad LoRA dropout bit planes (rows of ur_lora_bits_ld(N) bytes, 16-byte aligned)");
  }
  GemmP p;
  fill_params(a, p);
  p.slab_stride = 0;
  if (splits > 1) {
    int tiles = ur_cdiv(a->K, BK);
    p.ksplit_len = ur_cdiv(tiles, splits) * BK;
    p.slab_stride = (long)a->M * a->ldc;
    p.C = workspace;
  } else {
    p.ksplit_len = a->K > 0 ? ur_cdiv(a->K, BK) * BK : BK;
  }
  hipStream_t st = (hipStream_t)stream;
  if (a->swp_act) {
    UR_REQUIRE(swiglu_paired_ok(a, p), "ur_gemm: the paired SwiGLU forward epilogue is not available for these arguments (ur_gemm_swiglu_paired_supported)");
    return urgemm::gemm_pers_launch(p, st);
  }
  if (a->qkr_q) {
    UR_REQUIRE(qkrope_ok(a, p), "ur_gemm: the q/k-norm + RoPE epilogue is not available for these arguments (ur_gemm_qkrope_supported)");
    return urgemm::gemm_pers_launch(p, st);
  }
  int rc;
  const bool rk = a->r_kcontig != 0, sk = a->s_kcontig != 0, f32 = a->c_f32 != 0;
  if (rk && sk) rc = f32 ? launch<true, true, true>(p, splits, st) : launch<true, true, false>(p, splits, st);
  else if (rk && !sk) rc = f32 ? launch<true, false, true>(p, splits, st) : launch<true, false, false>(p, splits, st);
  else if (!rk && !sk) rc = f32 ? launch<false, false, true>(p, splits, st) : launch<false, false, false>(p, splits, st);
  else rc = f32 ? launch<false, true, true>(p, splits, st) : launch<false, true, false>(p, splits, st);
  if (rc) return rc;
  if (splits > 1) {
    long total = (long)a->M * a->ldc;
    UR_REQUIRE((total % 4) == 0, "ur_gemm: split_k output size must be a multiple of 4");
    long total4 = total / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, (float*)a->C,
                       total4, total4, splits);
    UR_CHECK_LAUNCH("ur_gemm(splitk_reduce)");
  }
  return 0;
}

// ---- grouped launch: the token reductions of one layer's weight gradients in one grid ---------------------------------------
static int grouped_check(const ur_gemm_args* a, int32_t count) {
  UR_REQUIRE(a != nullptr && count >= 1 && count <= UR_GEMM_MAX_GROUPS, "ur_gemm_grouped: 1 .. %d products per launch (got %d)", UR_GEMM_MAX_GROUPS, (int)count);
  for (int i = 0; i < count; ++i) {
    const ur_gemm_args* g = a + i;
    UR_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0 && g->R && g->S && g->C, "ur_gemm_grouped: product %d: empty or null", i);
    UR_REQUIRE(!g->r_kcontig && !g->s_kcontig && g->c_f32, "ur_gemm_grouped: product %d: token-major (K-strided) bf16 operands and f32 output only", i);
    UR_REQUIRE(!g->bias && !g->residual && !g->gelu_out && !g->gelu_grad_aux && g->K2 == 0 && !g->drop_bits && !g->swiglu_gu && !g->swiglu_gate && !g->qkr_q && !g->swp_act,
               "ur_gemm_grouped: product %d: no epilogue, no second operand pair", i);
    UR_REQUIRE(g->K == a->K && g->split_k == a->split_k && g->alpha == a->alpha, "ur_gemm_grouped: product %d: K, split_k and alpha must equal product 0's", i);
    UR_REQUIRE((g->M % 8) == 0 && (g->N % 8) == 0 && (g->ldr % 8) == 0 && (g->lds % 8) == 0 && (g->ldc % 4) == 0 && g->ldr >= g->M && g->lds >= g->N && g->ldc >= g->N,
               "ur_gemm_grouped: product %d: M, N, ldr, lds multiples of 8, ldc of 4, leading dimensions >= row lengths", i);
    UR_REQUIRE(UR_ALIGNED16(g->R) && UR_ALIGNED16(g->S) && UR_ALIGNED16(g->C), "ur_gemm_grouped: product %d: operands must be 16-byte aligned", i);
  }
  return 0;
}

extern "C" int64_t ur_gemm_grouped_workspace_bytes(const ur_gemm_args* a, int32_t count) {
  if (!a || count < 1 || count > UR_GEMM_MAX_GROUPS || a->split_k <= 1) return 0;
  int64_t t = 0;
  for (int i = 0; i < count; ++i) t += (int64_t)a->split_k * a[i].M * a[i].ldc * (int64_t)sizeof(float);
  return t;
}

extern "C" int ur_gemm_grouped(const ur_gemm_args* a, int32_t count, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = grouped_check(a, count)) return rc;
  const int splits = a->split_k > 1 ? a->split_k : 1;
  if (splits > 1)
    UR_REQUIRE(workspace && UR_ALIGNED16(workspace) && workspace_bytes >= ur_gemm_grouped_workspace_bytes(a, count),
               "ur_gemm_grouped: split_k workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)ur_gemm_grouped_workspace_bytes(a, count));
  // tile: 256 x 256 when every product has at least one whole big tile in both directions and the grid still fills half the chip
  bool big = true;
  long t256 = 0;
  for (int i = 0; i < count; ++i) {
    big = big && a[i].M >= 256 && a[i].N >= 256;
    t256 += (long)ur_cdiv(a[i].M, 256) * ur_cdiv(a[i].N, 256);
  }
  big = big && t256 * splits >= 128;
  const int BT = big ? 256 : 128;
  GemmP p;
  fill_params(a, p);
  {
    const int tiles = ur_cdiv(a->K, BK);
    p.ksplit_len = ur_cdiv(tiles, splits) * BK;
  }
  GemmGroups gs = {};
  ReduceGroups rg = {};
  gs.n = rg.n = count; gs.pad_ = rg.pad_ = 0;
  int nwg = 0;
  long maxtot4 = 0;
  float* ws = reinterpret_cast<float*>(workspace);
  for (int i = 0; i < count; ++i) {
    GemmGroupSlot& g = gs.g[i];
    g.R = (const bf16_t*)a[i].R; g.S = (const bf16_t*)a[i].S; g.ldr = a[i].ldr; g.lds = a[i].lds; g.ldc = a[i].ldc;
    g.M = a[i].M; g.N = a[i].N; g.gm = ur_cdiv(a[i].M, BT); g.gn = ur_cdiv(a[i].N, BT);
    g.wg0 = nwg; g.pad_ = 0;
    nwg += (g.gm * g.gn + 7) & ~7;
    const long tot = (long)a[i].M * a[i].ldc;
    if (splits > 1) {
      UR_REQUIRE((tot % 4) == 0, "ur_gemm_grouped: product %d: M * ldc must be a multiple of 4", i);
      g.C = ws; g.slab_stride = tot;
      rg.g[i].ws = ws; rg.g[i].C = reinterpret_cast<float*>(a[i].C); rg.g[i].total4 = tot / 4;
      if (tot / 4 > maxtot4) maxtot4 = tot / 4;
      ws += (long)splits * tot;
    } else {
      g.C = a[i].C; g.slab_stride = 0;
    }
  }
  for (int i = count; i < UR_GEMM_MAX_GROUPS; ++i) { gs.g[i] = gs.g[count - 1]; rg.g[i] = rg.g[count - 1]; }
  hipStream_t st = (hipStream_t)stream;
  int rc = big ? launch_grouped_cfg<256, 256, 2, 4>(p, gs, nwg, splits, st) : launch_grouped_cfg<128, 128, 2, 2>(p, gs, nwg, splits, st);
  if (rc) return rc;
  if (splits > 1) {
    int blocks = (int)((maxtot4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_grouped_kernel, dim3(blocks, count), dim3(256), 0, st, rg, splits);
    UR_CHECK_LAUNCH("ur_gemm_grouped(splitk_reduce)");
  }
  return 0;
}

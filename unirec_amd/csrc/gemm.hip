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
  constexpr bool ACC_A = UR_GEMM_ACC_AGPR && BM == 256 && BN == 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NWM * NWN * 64;
  constexpr int S_BYTES = SK ? Tile<BN>::KC_BYTES : Tile<BN>::KS_BYTES;
  constexpr int R_BYTES = RK ? Tile<BM>::KC_BYTES : Tile<BM>::KS_BYTES;
  constexpr int STAGE = S_BYTES + R_BYTES;
  constexpr int WM = BM / NWM, WN = BN / NWN;       // wave tile
  constexpr int MI = WM / 16, NI = WN / 16;         // 16x16 MFMA tiles per wave
  static_assert((MI % 4) == 0 && (NI % 4) == 0, "fragment reads go in groups of 4");
  // A wave owns TWO row groups of each operand tile, one in each half of the tile: R rows rh*BM/2 + wr*WM/2 + [0, WM/2)
  // and S rows sh*BN/2 + wc*WN/2 + [0, WN/2) (rh, sh = 0, 1).  Each half tile (128 rows of a 256-row tile) is then one
  // contiguous 16 KiB LDS region that all waves stop reading at the same phase of the 8-phase loop below, and
  // that two LDS-DMA pieces per wave refill.
  constexpr int HM = WM / 2, HN = WN / 2, MH = MI / 2, NH = NI / 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / NWN, wc = wave % NWN;
  auto s_row = [&](int i) { return (i / (NI / 2)) * (BN / 2) + wc * (BN / NWN / 2) + (i % (NI / 2)) * 16; };     // tile row of S block i
  auto r_row = [&](int j) { return (j / (MI / 2)) * (BM / 2) + wr * (BM / NWM / 2) + (j % (MI / 2)) * 16; };     // tile row of R block j

  // XCD-aware tile order: blocks sharing (id % 8) sit on one XCD (speed only); give each XCD a
  // contiguous run of tiles, column-tile fastest, so an R panel is re-read from that XCD's L2.
  const int nwg = p.gm * p.gn;
  int id = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, x = id & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
  }
  int bm = id / p.gn, bn = id - bm * p.gn;
  if (p.gcw > 0) {
    // each XCD owns gm/8 whole tile rows (host checks divisibility): walk them in column chunks of gcw tiles, so a
    // chunk's S panels (gcw * BN * K * 2 bytes) are what the XCD's L2 has to hold while the R panels stream past
    const int run = nwg >> 3, rows_x = run / p.gn, x = blockIdx.x & 7;
    const int j = id - x * run, per = rows_x * p.gcw;
    const int ch = j / per, rem = j - ch * per;
    bm = x * rows_x + rem / p.gcw;
    bn = ch * p.gcw + rem % p.gcw;
  }
  const int m0 = bm * BM, n0 = bn * BN;
  const int z = blockIdx.z;
  // De-phase the CUs: every tile of a launch takes the same time, so the 256 workgroups of a round reach their epilogues
  // together and 32 MiB of C leave for HBM at once (an epilogue of ~10 k cycles, most of it write back-pressure) while HBM
  // idles during the main loops.  The launch's FIRST wave of workgroups starts in 8 groups `stagger` cycles apart; the
  // offsets then persist from round to round.  (Lab builds only; measured neutral: DESIGN / docs/lab_notes.md.)
#if UR_LAB
  if (p.stagger > 0 && blockIdx.x < 256 && blockIdx.z == 0) {
    const long long until = (long long)__builtin_readcyclecounter() + (long long)((blockIdx.x >> 3) & 7) * p.stagger;
    while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(16);
  }
#endif
  UR_STAMP(0);

  int kbeg = z * p.ksplit_len;
  int kend = min(p.K, kbeg + p.ksplit_len);
  const int nt1 = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
  const int nt2 = (p.K2 > 0 && !p.drop_bits) ? (p.K2 + BK - 1) / BK : 0;
  const int nt = nt1 + nt2;
  const int nfull1 = (kend > kbeg) ? (kend - kbeg) / BK : 0;      // leading full tiles of the first K range

  f32x4 acc[NI][MI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  bool acc_zeroed = false;

  // k extent of tile t (how many of its two 32-deep halves carry data)
  auto tile_k = [&](int t, int& k0, int& ke) {
    if (t < nt1) { k0 = kbeg + t * BK; ke = kend; } else { k0 = (t - nt1) * BK; ke = p.K2; }
  };
  // stage tile t into ring slot `buf`: LDS-DMA for full tiles, register path (zero-fill) for K tails
  auto stage = [&](int t, char* buf) {
    const bf16_t* S; const bf16_t* R; long lds_, ldr_; int k0, ke;
    tile_k(t, k0, ke);
    if (t < nt1) { S = p.S; R = p.R; lds_ = p.lds; ldr_ = p.ldr; } else { S = p.S2; R = p.R2; lds_ = p.lds2; ldr_ = p.ldr2; }
    const bool full = k0 + BK <= ke;
    if (full) dma_tile<SK, BN, NT>(buf, S, lds_, p.N, n0, k0, tid);
    else reg_tile<SK, BN, NT>(buf, S, lds_, p.N, n0, k0, ke, tid);
    if (full) dma_tile<RK, BM, NT>(buf + S_BYTES, R, ldr_, p.M, m0, k0, tid);
    else reg_tile<RK, BM, NT>(buf + S_BYTES, R, ldr_, p.M, m0, k0, ke, tid);
  };
  auto read_frags = [&](bf16x8 (&sf)[NI], bf16x8 (&rf)[MI], int t, int h) {
    const char* sb = smem + (t & 1) * STAGE;
    lds_frags<SK, BN, NH>(sf, sb, wc * HN, h, lane);
    lds_frags<SK, BN, NH>(sf + NH, sb, BN / 2 + wc * HN, h, lane);
    lds_frags<RK, BM, MH>(rf, sb + S_BYTES, wr * HM, h, lane);
    lds_frags<RK, BM, MH>(rf + MH, sb + S_BYTES, BM / 2 + wr * HM, h, lane);
  };
  auto mfmas = [&](const bf16x8 (&sf)[NI], const bf16x8 (&rf)[MI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        mfma16<ACC_A>(acc[i][j], sf[i], rf[j]);
  };

  // Software pipeline, ONE barrier per 64-deep tile, placed between its two halves.  Fragment register
  // sets: A = half 0, B = half 1.  Tile t lives in ring slot t & 1.
  //   P0(t):  MFMAs(t, half 0) from A   ||  fragment reads (t, half 1) -> B
  //   mid(t): lgkmcnt(0) (every read of slot t&1 by this wave is done), vmcnt(0) (this wave's pieces of
  //           tile t+1, issued one tile ago, have landed), barrier  => slot t&1 is free, tile t+1 is complete
  //   P1(t):  LDS-DMA of tile t+2 -> slot t&1  ||  MFMAs(t, half 1) from B  ||  fragment reads (t+1, half 0) -> A
  // In the steady state (tile t+2 is a full tile of the first K range) P0 and P1 are four hard-fenced groups
  // each: 8 (4) MFMAs + a quarter of the fragment reads (+ two DMA pieces in P1), so the issue cost of the
  // DMA pieces and LDS reads hides under the matrix pipe instead of preceding it.
  constexpr int SPW = (BN * BK * 2 / 1024) / (NT / 64), RPW = (BM * BK * 2 / 1024) / (NT / 64);
  constexpr bool GROUPED = SK && RK && NI <= MI && ((SPW + RPW) % MI) == 0;
  // Interior blocks (no edge clamping): piece i of an operand is piece 0 shifted by a uniform number of rows
  // (K-contiguous: 8 * NT/64 rows; the swizzle term does not depend on i), so ONE lane offset per operand
  // serves all pieces and the per-piece shift goes into the scalar base.
  uint32_t svoff0, rvoff0;
  {
    uint32_t sv[SPW], rv[RPW];
    dma_offsets<SK, BN, NT>(sv, p.lds, p.N, n0, tid);
    dma_offsets<RK, BM, NT>(rv, p.ldr, p.M, m0, tid);
    svoff0 = sv[0]; rvoff0 = rv[0];
  }
  const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  const long spiece = (SK ? (long)(8 * (NT / 64)) * p.lds : (long)((NT / 64) * 512 / (BN / 8)) * p.lds) * 2;   // bytes between pieces
  const long rpiece = (RK ? (long)(8 * (NT / 64)) * p.ldr : (long)((NT / 64) * 512 / (BM / 8)) * p.ldr) * 2;
  const char* const sbase = reinterpret_cast<const char*>(p.S + (SK ? (long)n0 * p.lds : (long)n0));
  const char* const rbase = reinterpret_cast<const char*>(p.R + (RK ? (long)m0 * p.ldr : (long)m0));
  const long skstep = (SK ? 1 : p.lds) * 2, rkstep = (RK ? 1 : p.ldr) * 2;      // bytes per unit of k
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;

  auto mid = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0), as a builtin so the compiler's wait tracking sees it
#if UR_GEMM_ABLATE != 3
    __builtin_amdgcn_s_barrier();
#endif
  };
  // One half-step (32 deep) of the steady state = MI groups, group j = the NI MFMAs of row block j.
  // The R fragment of the NEXT half-step for block j-1 is read right after block j-1's last use, so it can
  // take over that register (the next set costs 16 registers for S instead of 48 for S and R), the S
  // fragments of the next half are spread over the first NI groups, and `extra(j)` issues the group's
  // share of the LDS-DMA pieces.  Hard fences keep loads and MFMAs of a group in the chosen order.
  auto half_step = [&](auto loads_first, const bf16x8 (&csf)[NI], bf16x8 (&crf)[MI], bf16x8 (&nsf)[NI],
                       const char* nb, int nh, auto extra) {
    constexpr bool LF = decltype(loads_first)::value;
    auto loads = [&](int j) {
      bf16x8 one[1];
      if (j >= 1) { lds_frags<true, BM, 1>(one, nb + S_BYTES, r_row(j - 1), nh, lane); crf[j - 1] = one[0]; }
      if (j < NI) { lds_frags<true, BN, 1>(one, nb, s_row(j), nh, lane); nsf[j] = one[0]; }
      extra(j);
    };
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      if (LF) { loads(j); __builtin_amdgcn_sched_barrier(0); }
      bf16x8 rj = crf[j];
#pragma unroll
#if UR_GEMM_ABLATE == 2
      for (int i = 0; i < NI; ++i) acc[i][j][0] += (float)(csf[i][0] ^ rj[0]);       // lab build: no MFMAs
#else
      for (int i = 0; i < NI; ++i) mfma16<ACC_A>(acc[i][j], csf[i], rj);
#endif
      __builtin_amdgcn_sched_barrier(0);
      if (!LF) { loads(j); __builtin_amdgcn_sched_barrier(0); }
    }
    { bf16x8 one[1]; lds_frags<true, BM, 1>(one, nb + S_BYTES, r_row(MI - 1), nh, lane); crf[MI - 1] = one[0]; }
  };
  bf16x8 sfA[NI], rfA[MI], sfB[NI], rfB[MI];
  int t = 0;

  // ==== 8-phase ping-pong steady state (256x256 tile, both operands K-contiguous, interior blocks) ==================
  // One K tile = 4 phases, one C quadrant (64 m x 32 n per wave, 16 MFMAs over the tile's 64 k) each:
  //   phase   LDS reads (load segment)        MFMAs (matrix segment)      LDS-DMA issued (tile t+2, same ring slot)
  //   1       S half 0 of tile t   (4)        Q(s0, r0)                   R half 0   (free since phase 4 of tile t-1)
  //   2       S half 1 of tile t   (4)        Q(s1, r0)                   S half 0   (free since phase 1)
  //   3       R half 1 of tile t   (8)        Q(s1, r1)                   S half 1   (free since phase 2)
  //   4       R half 0 of tile t+1 (8)        Q(s0, r1)                   R half 1   (free since phase 3)
  // Every phase is  [reads | 2 DMA pieces | vmcnt(12) | lgkmcnt(0)] barrier [16 MFMAs at priority 1] barrier.  The waves
  // with wr = 1 run one barrier interval behind the waves with wr = 0, so on every SIMD (one wave of each group) one
  // wave feeds the matrix pipe while the other reads LDS and issues DMA.
  // Hazards: a half tile is read 7 phases after its DMA was issued; the counted wait that retires it (this wave's two
  // pieces; 12 = the two pieces of each of the six phases issued after them) sits in the load segment of the phase
  // BEFORE the read, and a barrier follows it in both wave groups before either group reads.  A half tile's buffer is
  // re-filled one phase after its last read; lgkmcnt(0) before the load segment's closing barrier makes those reads
  // complete before any wave can issue the refill.
  // The same loop serves the token reductions (dW = dY^T X: BOTH operands K-strided): a half tile is one 16 KiB image there too
  // ([64 k][128 columns]), filled by two LDS-DMA pieces per wave (4 k-rows x 256 B each) and read with transposed LDS reads into
  // the same fragment registers; phases, counted waits and hazards are unchanged.
  constexpr bool PH8 = (RK == SK) && BM == 256 && BN == 256 && NWM == 2 && NWN == 4 && (UR_GEMM_ABLATE == 0) && !UR_GEMM_NO_PH8;
  if constexpr (PH8) {
    if (interior && nfull1 >= 3) {
      constexpr bool KC = RK;
      // K-strided fragment (16 columns c, k-half h) = two transposed 4-row reads at k-rows 32 h + 8 g + q (+ 4): the swizzle term
      // f = q | (g & 1) << 2 does not depend on h, so a lane needs ONE offset per fragment -- the k-half, the k-row + 4 and the half
      // tile are immediates of the read (lds_frags spends ~30 vector instructions and a full wait per group on the same addresses)
      uint32_t toffR[4], toffS[2];
      {
        const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3, tfz = tq | ((tg & 1) << 2);
        const uint32_t l0 = (uint32_t)((8 * tg + tq) * 256 + tp * 8);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) toffR[jj] = l0 + (uint32_t)(((((wr * 4) ^ (tfz & 4)) | (jj ^ (tfz & 3)))) << 5);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) toffS[ii] = l0 + (uint32_t)(((((wc * 2) ^ (tfz & 6)) | (ii ^ (tfz & 1)))) << 5);
      }
      const int l15 = lane & 15, g4 = lane >> 4;
      const uint32_t lo0 = l15 * 128 + (((g4) ^ ((l15 >> 1) & 7)) << 4), lo1 = l15 * 128 + (((4 + g4) ^ ((l15 >> 1) & 7)) << 4);
      bf16x8 R0[4][2], R1[4][2], S0[2][2], S1[2][2];
      auto rdR = [&](bf16x8 (&F)[4][2], const char* slot, int rh) {
        if constexpr (KC) {
          const char* b = slot + S_BYTES + (rh * 128 + wr * 64) * 128;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            F[jj][0] = *reinterpret_cast<const bf16x8*>(b + jj * 2048 + lo0);
            F[jj][1] = *reinterpret_cast<const bf16x8*>(b + jj * 2048 + lo1);
          }
        } else {
          // (inline asm, one wait for the sixteen reads: as a builtin the read is a tracked LDS load, and hipcc drains the LDS-DMA
          // queue -- s_waitcnt vmcnt(0) -- in front of every group because the pieces in flight may alias it)
          const uint32_t b = lds_off(slot) + S_BYTES + rh * 16384;
          const uint32_t a0 = b + toffR[0], a1 = b + toffR[1], a2 = b + toffR[2], a3 = b + toffR[3];
          bf16x4 q00, q01, q02, q03, q10, q11, q12, q13, q20, q21, q22, q23, q30, q31, q32, q33;
          asm volatile(
              "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:1024\n\t"
              "ds_read_b64_tr_b16 %2, %16 offset:8192\n\tds_read_b64_tr_b16 %3, %16 offset:9216\n\t"
              "ds_read_b64_tr_b16 %4, %17\n\tds_read_b64_tr_b16 %5, %17 offset:1024\n\t"
              "ds_read_b64_tr_b16 %6, %17 offset:8192\n\tds_read_b64_tr_b16 %7, %17 offset:9216\n\t"
              "ds_read_b64_tr_b16 %8, %18\n\tds_read_b64_tr_b16 %9, %18 offset:1024\n\t"
              "ds_read_b64_tr_b16 %10, %18 offset:8192\n\tds_read_b64_tr_b16 %11, %18 offset:9216\n\t"
              "ds_read_b64_tr_b16 %12, %19\n\tds_read_b64_tr_b16 %13, %19 offset:1024\n\t"
              "ds_read_b64_tr_b16 %14, %19 offset:8192\n\tds_read_b64_tr_b16 %15, %19 offset:9216\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(q00), "=&v"(q01), "=&v"(q02), "=&v"(q03), "=&v"(q10), "=&v"(q11), "=&v"(q12), "=&v"(q13),
                "=&v"(q20), "=&v"(q21), "=&v"(q22), "=&v"(q23), "=&v"(q30), "=&v"(q31), "=&v"(q32), "=&v"(q33)
              : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
          F[0][0] = cat4(q00, q01); F[0][1] = cat4(q02, q03); F[1][0] = cat4(q10, q11); F[1][1] = cat4(q12, q13);
          F[2][0] = cat4(q20, q21); F[2][1] = cat4(q22, q23); F[3][0] = cat4(q30, q31); F[3][1] = cat4(q32, q33);
        }
      };
      auto rdS = [&](bf16x8 (&F)[2][2], const char* slot, int sh) {
        if constexpr (KC) {
          const char* b = slot + (sh * 128 + wc * 32) * 128;
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            F[ii][0] = *reinterpret_cast<const bf16x8*>(b + ii * 2048 + lo0);
            F[ii][1] = *reinterpret_cast<const bf16x8*>(b + ii * 2048 + lo1);
          }
        } else {
          const uint32_t b = lds_off(slot) + sh * 16384;
          const uint32_t a0 = b + toffS[0], a1 = b + toffS[1];
          bf16x4 q00, q01, q02, q03, q10, q11, q12, q13;
          asm volatile(
              "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
              "ds_read_b64_tr_b16 %2, %8 offset:8192\n\tds_read_b64_tr_b16 %3, %8 offset:9216\n\t"
              "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:1024\n\t"
              "ds_read_b64_tr_b16 %6, %9 offset:8192\n\tds_read_b64_tr_b16 %7, %9 offset:9216\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(q00), "=&v"(q01), "=&v"(q02), "=&v"(q03), "=&v"(q10), "=&v"(q11), "=&v"(q12), "=&v"(q13)
              : "v"(a0), "v"(a1));
          F[0][0] = cat4(q00, q01); F[0][1] = cat4(q02, q03); F[1][0] = cat4(q10, q11); F[1][1] = cat4(q12, q13);
        }
      };
      // K-strided pieces: piece d of half hf = k-rows 32 d + 4 wave + (lane >> 4), 16-byte chunk lane & 15 of the half's 128 columns;
      // the swizzle term f = ks_f(k-row) depends on the lane and on (wave >> 1) & 1 only: one lane offset per operand
      const int tf = ((lane >> 4) & 3) | (((uwave >> 1) & 1) << 2);
      const uint32_t tsv = (uint32_t)(((long)(lane >> 4) * p.lds + (((lane & 15) ^ (tf << 1)) << 3)) * 2);
      const uint32_t trv = (uint32_t)(((long)(lane >> 4) * p.ldr + (((lane & 15) ^ (tf << 1)) << 3)) * 2);
      const long tsw = (long)uwave * 4 * p.lds * 2, trw = (long)uwave * 4 * p.ldr * 2;          // the wave's first k-row
      const long tsd = (long)32 * p.lds * 2, trd = (long)32 * p.ldr * 2;                        // piece 0 -> piece 1
      auto quad = [&](const bf16x8 (&S)[2][2], const bf16x8 (&R)[4][2], auto shc, auto rhc) {
        constexpr int sh = decltype(shc)::value, rh = decltype(rhc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              mfma16<ACC_A>(acc[2 * sh + ii][4 * rh + jj], S[ii][h], R[jj][h]);
        __builtin_amdgcn_s_setprio(0);
      };
      // the two LDS-DMA pieces of this wave for half `hf` (rows 128 hf ..) of an operand tile whose k position is in `ub`
      auto dma_half = [&](char* slot, auto is_s, int hf, const char* ub) {
        constexpr bool IS_S = decltype(is_s)::value;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int li = 2 * hf + d;
          const char* src;
          if constexpr (KC) src = ub + li * (IS_S ? spiece : rpiece) + (IS_S ? svoff0 : rvoff0);
          else src = ub + hf * 256 + (IS_S ? tsw + d * tsd : trw + d * trd) + (IS_S ? tsv : trv);
          char* dst = slot + (IS_S ? 0 : S_BYTES) + (li * (NT / 64) + uwave) * 1024;
          __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
        }
      };
      auto seg_end = [&]() {            // end of a load segment
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      auto mat_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      const std::integral_constant<int, 0> c0;
      const std::integral_constant<int, 1> c1;
      // prologue: both tiles' half tiles in consumption order R0 S0 S1 R1 (the loop's issue order shifted back 8 phases)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const long kn = kbeg + (long)tt * BK;
        const char* const ubs = uniform_ptr(sbase + kn * skstep);
        const char* const ubr = uniform_ptr(rbase + kn * rkstep);
        char* slot = smem + tt * STAGE;
        // the counted waits rely on this issue order: keep the scheduler from re-ordering the pieces
        dma_half(slot, std::false_type{}, 0, ubr); __builtin_amdgcn_sched_barrier(0);
        dma_half(slot, std::true_type{}, 0, ubs);  __builtin_amdgcn_sched_barrier(0);
        dma_half(slot, std::true_type{}, 1, ubs);  __builtin_amdgcn_sched_barrier(0);
        dma_half(slot, std::false_type{}, 1, ubr); __builtin_amdgcn_sched_barrier(0);
      }
      UR_STAMP(1);
      zero_acc();                                              // under the first pieces' flight
      acc_zeroed = true;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");       // R half 0 and S half 0 of tile 0 have landed (this wave's pieces)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      rdR(R0, smem, 0);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      UR_STAMP(2);
      if (wr == 1) __builtin_amdgcn_s_barrier();               // stagger: this group now runs one interval behind
      __builtin_amdgcn_sched_barrier(0);
      for (; t + 2 < nfull1; ++t) {
        char* slot = smem + (t & 1) * STAGE;
        const char* nslot = smem + ((t + 1) & 1) * STAGE;
        const long kn = kbeg + (long)(t + 2) * BK;
        const char* const ubs = uniform_ptr(sbase + kn * skstep);
        const char* const ubr = uniform_ptr(rbase + kn * rkstep);
        // phase 1
        rdS(S0, slot, 0);
        dma_half(slot, std::false_type{}, 0, ubr);
        seg_end();
        quad(S0, R0, c0, c0);
        mat_end();
        // phase 2
        rdS(S1, slot, 1);
        dma_half(slot, std::true_type{}, 0, ubs);
        seg_end();
        quad(S1, R0, c1, c0);
        mat_end();
        // phase 3
        rdR(R1, slot, 1);
        dma_half(slot, std::true_type{}, 1, ubs);
        seg_end();
        quad(S1, R1, c1, c1);
        mat_end();
        // phase 4
        rdR(R0, nslot, 0);
        dma_half(slot, std::false_type{}, 1, ubr);
        seg_end();
        quad(S0, R1, c0, c1);
        mat_end();
      }
      // hand over to the generic loop: tiles t and t+1 are issued (t+1 possibly still in flight); re-join the groups
      if (wr == 0) __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_frags(sfA, rfA, t, 0);
      UR_STAMP(3);
    }
  }

  if (!acc_zeroed) zero_acc();
  // prologue: tiles 0 and 1 issued, tile 0 landed, its half-0 fragments in A
  if (t == 0 && nt > 0) stage(0, smem);
  if (t == 0 && nt > 1) stage(1, smem + STAGE);
  if (t == 0 && nt > 0) {
    // a K-tail tile staged through registers issues no DMA; its own (compiler-waited) loads are older
    if (nt > 1 && nfull1 >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(SPW + RPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    read_frags(sfA, rfA, 0, 0);
  }
  if (!PH8 && GROUPED && interior) {
    // ---- steady state: tiles t, t+1 and t+2 are full tiles of the first K range ----
    auto steady = [&](auto order) {
      for (; t + 2 < nfull1; ++t) {
        const char* cur = smem + (t & 1) * STAGE;
        char* slot = smem + (t & 1) * STAGE;
        const char* nxt = smem + ((t + 1) & 1) * STAGE;
        // P0: half 0 from (sfA, rf); the next half's fragments come from the same tile
        half_step(order, sfA, rfA, sfB, cur, 1, [](int) {});
        mid();
        const long kn = kbeg + (long)(t + 2) * BK;
        // uniform tile bases pinned to SGPRs (scalar per-piece shifts, one lane offset per operand)
        const char* const ubs = uniform_ptr(sbase + kn * skstep);
        const char* const ubr = uniform_ptr(rbase + kn * rkstep);
        auto dma = [&](int j) {
          constexpr int PPG = (SPW + RPW) / MI;             // pieces per group
#pragma unroll
          for (int d = 0; d < PPG; ++d) {
            const int pi = j * PPG + d;                     // piece index: S pieces first, then R pieces
            const bool is_s = pi < SPW;
            const int li = is_s ? pi : pi - SPW;
            const char* ub = is_s ? ubs + li * spiece : ubr + li * rpiece;      // scalar
            const uint32_t vo = is_s ? svoff0 : rvoff0;
            char* dst = slot + (is_s ? 0 : S_BYTES) + (li * (NT / 64) + uwave) * 1024;
#if UR_GEMM_ABLATE != 1
            __builtin_amdgcn_global_load_lds((gbl_void*)(ub + vo), (lds_void*)dst, 16, 0, 0);
#endif
          }
        };
        // P1: half 1 from (sfB, rf); the next half's fragments come from tile t+1
        half_step(order, sfB, rfA, sfA, nxt, 0, dma);
      }
    };
    steady(std::false_type{});       // MFMAs, then the group's loads (loads-first is the same stream shifted by one group)
  }
  // ---- generic tiles: K tails, the second (LoRA) K range, the last two tiles, K-strided operands ----
  for (; t < nt; ++t) {
    char* slot = smem + (t & 1) * STAGE;
    int k0, ke;
    tile_k(t, k0, ke);
    const bool two = ke - k0 > 32;                      // the tile's second half carries data
    if (two) read_frags(sfB, rfB, t, 1);
    mfmas(sfA, rfA);
    mid();
    if (t + 2 < nt) stage(t + 2, slot);
    if (t + 1 < nt) read_frags(sfA, rfA, t + 1, 0);
    if (two) mfmas(sfB, rfB);
  }

  UR_STAMP(4);
  // the asm MFMAs' results are read below by instructions hipcc schedules without knowing an MFMA wrote them: let the
  // last one retire (16x16x32: 8 passes) before anything touches the accumulators
  if constexpr (ACC_A) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  // ---- LoRA dropout, backward to the adapter input: C(m,n) += sum_a keep_a(m,n)/(1-p) * tb_a(m,:) . A_a(:,n).
  // Each adapter's rank-r product of a 16x16 sub-tile is ONE MFMA (k = r <= 32, zero-padded) into a scratch
  // accumulator; the keep flags come from the adapters' dropped-flag bit planes (lora.hip: 8 bytes cover the
  // 64 columns this wave owns of one row).
  if (p.drop_bits && p.K2 > 0) {
    const int nad = p.K2 / p.drop_rank, kq = 8 * (lane >> 4);
    const bool kin = kq < p.drop_rank;
    const int g4 = lane >> 4;
    static_assert(HN == 32, "the masked LoRA epilogue reads one 4-byte flag word per row and column half of the wave tile");
    const long boff0 = min((long)((n0 + wc * HN) >> 3), p.drop_bits_ld - 4);
    const long boff1 = min((long)((n0 + BN / 2 + wc * HN) >> 3), p.drop_bits_ld - 4);
    for (int a = 0; a < nad; ++a) {
      bf16x8 s2[NI], r2[MI];
      const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int n = min(n0 + s_row(i) + (lane & 15), p.N - 1);
        s2[i] = kin ? *reinterpret_cast<const bf16x8*>(p.S2 + (long)n * p.lds2 + a * p.drop_rank + kq) : zero8;
      }
      uint2 fl[MI];
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int m = min(m0 + r_row(j) + (lane & 15), p.M - 1);
        r2[j] = kin ? *reinterpret_cast<const bf16x8*>(p.R2 + (long)m * p.ldr2 + a * p.drop_rank + kq) : zero8;
        const uint8_t* brow = p.drop_bits + (long)a * p.drop_bits_stride + (long)m * p.drop_bits_ld;
        fl[j] = make_uint2(*reinterpret_cast<const uint32_t*>(brow + boff0), *reinterpret_cast<const uint32_t*>(brow + boff1));
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s2[i], r2[j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          // columns 16 (i % 2) + 4 g4 .. + 3 of column half i / 2: byte 2 (i % 2) + (g4 >> 1) of that half's word,
          // pair-interleaved flag order (lora.hip): element 4 q + e -> bit 2 q + (e >> 1) + 4 (e & 1), q = g4 & 1
          const uint32_t wsel = (i >= NH) ? fl[j].y : fl[j].x;
          const uint32_t f = (wsel >> (16 * (i % NH) + 8 * (g4 >> 1) + 2 * (g4 & 1))) & 0x33u;
          if (!(f & 0x01u)) acc[i][j][0] += d[0] * p.drop_inv_keep;
          if (!(f & 0x10u)) acc[i][j][1] += d[1] * p.drop_inv_keep;
          if (!(f & 0x02u)) acc[i][j][2] += d[2] * p.drop_inv_keep;
          if (!(f & 0x20u)) acc[i][j][3] += d[3] * p.drop_inv_keep;
        }
    }
  }

  // ---- epilogue: lane holds n = n0 + s_row(i) + (lane>>4)*4 + 0..3, m = m0 + r_row(j) + (lane&15).
  // Measured with in-kernel stamps (tools/lab/gemm_stamps.py): a per-element epilogue in the MFMA layout (32 (i, j)
  // sub-tiles, each with its own predicates, scalar-pointer checks and 8-byte residual / aux loads waited one by one)
  // cost 27k cycles per 256x256 tile, as much as 11 K tiles of the main loop.  So:
  //   f32 output        : float4 stores from the accumulators, predicates only on edge tiles.
  //   bf16, plain       : (no residual / aux / gelu_out) alpha * acc + bias -> bf16 -> LDS tile -> whole rows.
  //   bf16, rich        : the tile goes through LDS in f32, one column half at a time; bias, residual, gelu' and the
  //                       GELU second output are applied on the way out, where every access is a coalesced 16-byte
  //                       piece of a row (and the f32 sum is rounded once, as before).
  const int nq = (lane >> 4) * 4, ml = lane & 15;
  if (OUTF32) {
    float* Cf = reinterpret_cast<float*>(p.C) + (long)z * p.slab_stride;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = n0 + s_row(i) + nq;
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && n < p.N) bb = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int m = m0 + r_row(j) + ml;
        const f32x4 a = acc[i][j];
        const float4 v = make_float4(a[0] * p.alpha + bb.x, a[1] * p.alpha + bb.y, a[2] * p.alpha + bb.z, a[3] * p.alpha + bb.w);
        if (interior || (n < p.N && m < p.M)) *reinterpret_cast<float4*>(Cf + (long)m * p.ldc + n) = v;
      }
    }
  } else {
    bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C);
    const bool rich = EPI != 0 || p.res || p.aux || p.gelu_out;        // uniform (a bias alone stays on the plain path)
    __syncthreads();                                                   // every wave is done with the ring
    if (!rich) {
      constexpr int CROWB = BN * 2 + 16;            // padded LDS row of the bf16 C tile
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);                  // the lane's four columns of this block
        if (p.bias && n0 + s_row(i) + nq < p.N) bb = *reinterpret_cast<const float4*>(p.bias + n0 + s_row(i) + nq);
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          const f32x4 a = acc[i][j];
          *reinterpret_cast<uint2*>(smem + (r_row(j) + ml) * CROWB + (s_row(i) + nq) * 2) =
              make_uint2(pack_bf2(fmaf(a[0], p.alpha, bb.x), fmaf(a[1], p.alpha, bb.y)), pack_bf2(fmaf(a[2], p.alpha, bb.z), fmaf(a[3], p.alpha, bb.w)));
        }
      }
      __syncthreads();
      const bool wide = ((p.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
      constexpr int CPR = BN / 8;                   // 16-byte chunks per tile row
#pragma unroll 4
      for (int c = tid; c < BM * CPR; c += NT) {
        const int row = c / CPR, ch = c % CPR;
        const int m = m0 + row, n = n0 + ch * 8;
        if (interior || (m < p.M && n < p.N)) {
          const uint4 val = *reinterpret_cast<const uint4*>(smem + row * CROWB + ch * 16);
          bf16_t* dst = Cb + (long)m * p.ldc + n;
          if (wide && (interior || n + 8 <= p.N)) {
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t tv = {val.x, val.y, val.z, val.w};
            __builtin_nontemporal_store(tv, reinterpret_cast<u32x4_t*>(dst));
          } else {
            *reinterpret_cast<uint2*>(dst) = make_uint2(val.x, val.y);                       // N % 4 == 0: first half always fits
            if (n + 8 <= p.N) *reinterpret_cast<uint2*>(dst + 4) = make_uint2(val.z, val.w);
          }
        }
      }
    } else {
      constexpr int FROWB = (BN / 2) * 4 + 16;      // padded LDS row of one f32 column half of the C tile
      const bool wide8 = ((p.N & 7) == 0) && ((p.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                         (!p.res || (((p.ldres & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0))) &&
                         (!p.aux || (((p.ldaux & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.aux) & 15) == 0))) &&
                         (!p.gelu_out || (((p.ldg & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.gelu_out) & 15) == 0))) &&
                         (EPI == 0 || (((p.sw_ldgu & 7) == 0) && ((p.sw_lddgu & 7) == 0) && ((p.sw_I & 7) == 0) &&
                                       ((reinterpret_cast<uintptr_t>(p.sw_gu) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.sw_dgu) & 15) == 0)));
      // pieces of 8 columns (16 bytes) when everything is 16-byte addressable, else of 4 columns (N, ld % 4 == 0 always)
      const int cw = wide8 ? 8 : 4, cpr = (BN / 2) / cw, ch = tid % cpr, rstep = NT / cpr;
      auto ldp = [&](const bf16_t* q, uint32_t (&w)[4]) {
        if (wide8) { const uint4 t4 = *reinterpret_cast<const uint4*>(q); w[0] = t4.x; w[1] = t4.y; w[2] = t4.z; w[3] = t4.w; }
        else { const uint2 t2 = *reinterpret_cast<const uint2*>(q); w[0] = t2.x; w[1] = t2.y; w[2] = w[3] = 0; }
      };
      auto stp = [&](bf16_t* q, const uint32_t (&w)[4]) {
        if (wide8) *reinterpret_cast<uint4*>(q) = make_uint4(w[0], w[1], w[2], w[3]);
        else *reinterpret_cast<uint2*>(q) = make_uint2(w[0], w[1]);
      };
#pragma unroll 1
      for (int sh = 0; sh < 2; ++sh) {
        if (sh) __syncthreads();                    // the copy-out of half 0 has read the tile
        if (sh == 0) {
#pragma unroll
          for (int ii = 0; ii < NH; ++ii)
#pragma unroll
            for (int j = 0; j < MI; ++j) {
              const f32x4 a = acc[ii][j];
              *reinterpret_cast<float4*>(smem + (r_row(j) + ml) * FROWB + (wc * HN + ii * 16 + nq) * 4) =
                  make_float4(a[0] * p.alpha, a[1] * p.alpha, a[2] * p.alpha, a[3] * p.alpha);
            }
        } else {
#pragma unroll
          for (int ii = 0; ii < NH; ++ii)
#pragma unroll
            for (int j = 0; j < MI; ++j) {
              const f32x4 a = acc[NH + ii][j];
              *reinterpret_cast<float4*>(smem + (r_row(j) + ml) * FROWB + (wc * HN + ii * 16 + nq) * 4) =
                  make_float4(a[0] * p.alpha, a[1] * p.alpha, a[2] * p.alpha, a[3] * p.alpha);
            }
        }
        __syncthreads();
        const int n = n0 + sh * (BN / 2) + ch * cw;
        const bool nok = interior || n < p.N;
        float bsv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias && nok) {
          const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
          bsv[0] = b0.x; bsv[1] = b0.y; bsv[2] = b0.z; bsv[3] = b0.w;
          if (wide8) {
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bsv[4] = b1.x; bsv[5] = b1.y; bsv[6] = b1.z; bsv[7] = b1.w;
          }
        }
        // four rows per trip: their residual / aux pieces go out together (one wait instead of one per row)
#pragma unroll 1
        for (int r0 = tid / cpr; r0 < BM; r0 += 4 * rstep) {
          uint32_t rw[4][4], aw[4][4], gw[4][4], uw[4][4];
          bool ok[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int row = r0 + k * rstep, m = m0 + row;
            ok[k] = nok && row < BM && (interior || m < p.M);
            rw[k][0] = rw[k][1] = rw[k][2] = rw[k][3] = 0;
            aw[k][0] = aw[k][1] = aw[k][2] = aw[k][3] = 0;
            if (p.res && ok[k]) ldp(p.res + (long)m * p.ldres + n, rw[k]);
            if (p.aux && ok[k]) ldp(p.aux + (long)m * p.ldaux + n, aw[k]);
            gw[k][0] = gw[k][1] = gw[k][2] = gw[k][3] = 0;
            uw[k][0] = uw[k][1] = uw[k][2] = uw[k][3] = 0;
            if (EPI == 1 && ok[k]) {
              ldp(p.sw_gu + (long)m * p.sw_ldgu + n, gw[k]);
              ldp(p.sw_gu + (long)m * p.sw_ldgu + p.sw_I + n, uw[k]);
            }
            if (EPI == 2 && ok[k]) ldp(p.sw_gu + (long)m * p.sw_ldgu + n, gw[k]);       // gate
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int row = r0 + k * rstep, m = m0 + row;
            if (!ok[k]) continue;
            const char* lrow = smem + row * FROWB + ch * cw * 4;
            const float4 f0 = *reinterpret_cast<const float4*>(lrow);
            float4 f1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wide8) f1 = *reinterpret_cast<const float4*>(lrow + 16);
            float v[8] = {f0.x + bsv[0], f0.y + bsv[1], f0.z + bsv[2], f0.w + bsv[3], f1.x + bsv[4], f1.y + bsv[5], f1.z + bsv[6], f1.w + bsv[7]};
            if (p.res) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[2 * e] += bf_lo(rw[k][e]); v[2 * e + 1] += bf_hi(rw[k][e]); }
            }
            if (p.aux) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[2 * e] *= gelu_erf_grad_f(bf_lo(aw[k][e])); v[2 * e + 1] *= gelu_erf_grad_f(bf_hi(aw[k][e])); }
            }
            if (EPI == 1) {
              // d(act) = v (f32, unrounded): dgate = v u silu'(g), dup = v silu(g)   (elementwise.hip: swiglu_bwd_kernel)
              uint32_t og[4], ou[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float dgv[2], duv[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                  const float gg = hh ? bf_hi(gw[k][e]) : bf_lo(gw[k][e]), uu = hh ? bf_hi(uw[k][e]) : bf_lo(uw[k][e]);
                  const float d = v[2 * e + hh];
                  const float sg = sigmoid_f(gg);
                  duv[hh] = d * (gg * sg);
                  dgv[hh] = d * uu * (sg * (1.0f + gg * (1.0f - sg)));
                }
                og[e] = pack_bf2(dgv[0], dgv[1]); ou[e] = pack_bf2(duv[0], duv[1]);
              }
              stp(p.sw_dgu + (long)m * p.sw_lddgu + n, og);
              stp(p.sw_dgu + (long)m * p.sw_lddgu + p.sw_I + n, ou);
              continue;
            }
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
            stp(Cb + (long)m * p.ldc + n, o);
            if (EPI == 2) {
              // act = silu(gate) * up, from the bf16-ROUNDED up the backward will read (elementwise.hip: swiglu_fwd_kernel)
              uint32_t oa[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) oa[e] = pack_bf2(silu_f(bf_lo(gw[k][e])) * bf_lo(o[e]), silu_f(bf_hi(gw[k][e])) * bf_hi(o[e]));
              stp(p.sw_dgu + (long)m * p.sw_lddgu + n, oa);
            }
            if (p.gelu_out) {
              // GELU of the bf16-ROUNDED pre-activation, so backward's gelu'(u) sees the same u
              uint32_t gq[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) gq[e] = pack_bf2(gelu_erf_f(bf_lo(o[e])), gelu_erf_f(bf_hi(o[e])));
              stp(p.gelu_out + (long)m * p.ldg + n, gq);
            }
          }
        }
      }
    }
  }
  UR_STAMP(5);
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

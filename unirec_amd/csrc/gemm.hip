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
// Staging: LDS-DMA (global_load_lds_dwordx4) straight into a 4-stage ring of XOR-swizzled LDS images
// (no staging VGPRs, no ds_write); three 32-deep tiles stay in flight under the current tile's MFMAs
// (counted s_waitcnt vmcnt + raw s_barrier, one barrier per tile).  K-contiguous tiles are read with ds_read_b128,
// K-strided tiles with ds_read_b64_tr_b16 (hardware transpose); both images are bank-conflict-free.
#include "common.cuh"
#include "unirec_hip.h"

namespace {

constexpr int BK = 32;                     // K depth of one ring stage = one v_mfma_f32_16x16x32_bf16 k-step
constexpr int NSTAGE = 4;                  // LDS ring: 3 tiles of LDS-DMA in flight under the current tile's MFMAs
constexpr int KC_ROWB = BK * 2;            // K-contiguous tile row bytes
// LDS image of one operand tile of T rows (T = 128 or 256), no padding (LDS-DMA writes linearly):
//   K-contiguous  [T][32 k] : 64-B rows, 16-B chunk c of row r stored at chunk c ^ g(r),
//                             g(r) = (-(r >> 2)) & 3  -> conflict-free ds_read_b128 fragment reads
//   K-strided     [32 k][T] : 2T-byte rows, 32-B segment s of k-row r stored at segment s ^ f(r),
//                             f(r) = (r & 3) | (((r >> 3) & 1) << 2)
//                             -> the 8 (k-row, 32-B) pieces one half-wave ds_read_b64_tr_b16 touches
//                                land on 8 different 32-B bank groups: conflict-free transposed reads
template <int T> struct Tile {
  static constexpr int KC_BYTES = T * KC_ROWB;
  static constexpr int KS_ROWB = T * 2;
  static constexpr int KS_BYTES = BK * KS_ROWB;
};
struct GemmP {
  const bf16_t* R; const bf16_t* S; long ldr, lds; int K;
  const bf16_t* R2; const bf16_t* S2; long ldr2, lds2; int K2;
  void* C; long ldc; int M, N; float alpha;
  const float* bias; const bf16_t* res; long ldres;
  bf16_t* gelu_out; long ldg; const bf16_t* aux; long ldaux;
  int ksplit_len; long slab_stride;
  int gm, gn;
};

__device__ __forceinline__ int ks_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
__device__ __forceinline__ int kc_g(int r) { return (-(r >> 2)) & 3; }

// ---- LDS-DMA staging of a FULL 32-deep tile: 1 KiB per wave instruction, swizzle on the source ---
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
      const int row = inst * 16 + (lane >> 2), pos = lane & 3;
      const int g = min(row0 + row, rows_total - 1);
      src = base + (long)g * ld + k0 + ((pos ^ kc_g(row)) << 3);
    } else {
      const int c = inst * 64 + lane;
      const int kr = c / (T / 8), ch = c % (T / 8);
      const int col = min(row0 + ((ch ^ (ks_f(kr) << 1)) << 3), rows_total - 8);
      src = base + (long)(k0 + kr) * ld + col;
    }
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(tile + inst * 1024), 16, 0, 0);
  }
}

// Steady-state LDS-DMA: the lane's source offset inside the block's operand panel does not change from
// tile to tile (only the uniform K position does), so it is computed once (dma_offsets) and a full tile
// costs no vector arithmetic: every piece is  uniform base (SGPRs) + 32-bit lane offset.
template <bool KC, int T, int NT>
__device__ __forceinline__ void dma_offsets(uint32_t (&voff)[(T * BK * 2 / 1024) / (NT / 64)], long ld, int rows_total, int row0, int tid) {
  constexpr int PER_WAVE = (T * BK * 2 / 1024) / (NT / 64);
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int inst = i * (NT / 64) + wave;
    long e;                                     // element offset from base + (KC ? row0 * ld : row0)
    if (KC) {
      const int row = inst * 16 + (lane >> 2), pos = lane & 3;
      const int g = min(row0 + row, rows_total - 1) - row0;
      e = (long)g * ld + ((pos ^ kc_g(row)) << 3);
    } else {
      const int c = inst * 64 + lane;
      const int kr = c / (T / 8), ch = c % (T / 8);
      const int col = min(row0 + ((ch ^ (ks_f(kr) << 1)) << 3), rows_total - 8) - row0;
      e = (long)kr * ld + col;
    }
    voff[i] = (uint32_t)(e * 2);
  }
}
template <int T, int NT>
__device__ __forceinline__ void dma_tile_fast(char* tile, const char* ubase, const uint32_t (&voff)[(T * BK * 2 / 1024) / (NT / 64)], int wave) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
  constexpr int PER_WAVE = (T * BK * 2 / 1024) / (NT / 64);
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i)
    __builtin_amdgcn_global_load_lds((gbl_void*)(ubase + voff[i]), (lds_void*)(tile + (i * (NT / 64) + wave) * 1024), 16, 0, 0);
}

// ---- register staging (partial K tiles only: zero-fill past kend) ------------------------------
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
      const int row = c >> 2, kc = c & 3;
      const int grow = min(row0 + row, rows_total - 1), gk = k0 + kc * 8;
      if (gk < kend) z = *reinterpret_cast<const uint4*>(base + (long)grow * ld + gk);
      off = row * KC_ROWB + ((kc ^ kc_g(row)) << 4);
    } else {
      const int kr = c / (T / 8), ch = c % (T / 8);
      const int gk = k0 + kr, gcol = min(row0 + ch * 8, rows_total - 8);
      if (gk < kend) z = *reinterpret_cast<const uint4*>(base + (long)gk * ld + gcol);
      off = kr * Tile<T>::KS_ROWB + ((ch ^ (ks_f(kr) << 1)) << 4);
    }
    *reinterpret_cast<uint4*>(tile + off) = z;
  }
}

// ---- LDS -> MFMA fragments: lane holds [idx = idx0 + 16*i + (lane&15)][k = 8*(lane>>4) + 0..7] ---
template <bool KC, int T, int N>
__device__ __forceinline__ void lds_frags(bf16x8 (&f)[N], const char* tile, int idx0, int lane) {
  if (KC) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int idx = idx0 + 16 * i + (lane & 15);
      f[i] = *reinterpret_cast<const bf16x8*>(tile + idx * KC_ROWB + (((lane >> 4) ^ kc_g(idx)) << 4));
    }
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int ka = 8 * g + q;                           // k-rows ka (elements 0..3) and ka+4 (elements 4..7)
    const uint32_t ra = lds_off(tile) + ka * Tile<T>::KS_ROWB + pp * 8, rb = ra + 4 * Tile<T>::KS_ROWB;
    const int fa = ks_f(ka), fb = ks_f(ka + 4);
#pragma unroll
    for (int i0 = 0; i0 < N; i0 += 4) {
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
  }
}

// BM x BN block tile, NWM x NWN waves; each wave owns (BM/NWM) rows x (BN/NWN) columns of C.
template <bool RK, bool SK, bool OUTF32, int BM, int BN, int NWM, int NWN>
__global__ __launch_bounds__(NWM * NWN * 64, 2) void gemm_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NWM * NWN * 64;
  constexpr int S_BYTES = SK ? Tile<BN>::KC_BYTES : Tile<BN>::KS_BYTES;
  constexpr int R_BYTES = RK ? Tile<BM>::KC_BYTES : Tile<BM>::KS_BYTES;
  constexpr int STAGE = S_BYTES + R_BYTES;
  constexpr int WM = BM / NWM, WN = BN / NWN;       // wave tile
  constexpr int MI = WM / 16, NI = WN / 16;         // 16x16 MFMA tiles per wave

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / NWN, wc = wave % NWN;

  // XCD-aware tile order: blocks sharing (id % 8) sit on one XCD (speed only); give each XCD a
  // contiguous run of tiles, column-tile fastest, so an R panel is re-read from that XCD's L2.
  const int nwg = p.gm * p.gn;
  int id = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, x = id & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
  }
  const int bm = id / p.gn, bn = id - bm * p.gn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int z = blockIdx.z;

  int kbeg = z * p.ksplit_len;
  int kend = min(p.K, kbeg + p.ksplit_len);
  const int nt1 = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
  const int nt2 = (p.K2 > 0) ? (p.K2 + BK - 1) / BK : 0;
  const int nt = nt1 + nt2;

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // stage tile t into ring slot `buf`: LDS-DMA for full tiles, register path (zero-fill) for K tails
  auto stage = [&](int t, char* buf) {
    const bf16_t* S; const bf16_t* R; long lds_, ldr_; int k0, ke;
    if (t < nt1) { S = p.S; R = p.R; lds_ = p.lds; ldr_ = p.ldr; k0 = kbeg + t * BK; ke = kend; }
    else { S = p.S2; R = p.R2; lds_ = p.lds2; ldr_ = p.ldr2; k0 = (t - nt1) * BK; ke = p.K2; }
    if (k0 + BK <= ke) {
      dma_tile<SK, BN, NT>(buf, S, lds_, p.N, n0, k0, tid);
      dma_tile<RK, BM, NT>(buf + S_BYTES, R, ldr_, p.M, m0, k0, tid);
    } else {
      reg_tile<SK, BN, NT>(buf, S, lds_, p.N, n0, k0, ke, tid);
      reg_tile<RK, BM, NT>(buf + S_BYTES, R, ldr_, p.M, m0, k0, ke, tid);
    }
  };
  // LDS-DMA instructions one wave issues per full tile (both operands): the unit of the counted waits
  constexpr int DMA_PER_TILE = (BN * BK * 2 / 1024) / (NT / 64) + (BM * BK * 2 / 1024) / (NT / 64);

  // Software pipeline (one barrier per 32-deep tile).  Invariant at the top of step(t): tile t's
  // fragments are in registers, tiles t+1 .. t+NSTAGE-1 have been ISSUED (tile t+1 must have landed,
  // the others may be in flight).  step(t):
  //   wait(tile t+1 landed) -> barrier -> refill the ring slot of tile t (its fragments are in registers
  //   of every wave) with tile t+NSTAGE -> issue the LDS fragment reads of tile t+1 -> MFMAs of tile t.
  // The fragment reads of the next tile and NSTAGE-2 tiles of LDS-DMA run under the MFMAs.
  // vmcnt is in-order: "at most n*DMA_PER_TILE outstanding" == "all but the n youngest tiles landed".
  // A K-tail tile staged through registers issues no DMA, but its own (compiler-waited) global loads
  // are younger than every DMA before it, so it only makes these waits more conservative.
  auto wait_all_but = [&](int n_tiles) {
    if (n_tiles >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * DMA_PER_TILE) : "memory");
    else if (n_tiles == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * DMA_PER_TILE) : "memory");
    else if (n_tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto read_frags = [&](bf16x8 (&sf)[NI], bf16x8 (&rf)[MI], int t) {
    const char* sb = smem + (t % NSTAGE) * STAGE;
    lds_frags<SK, BN, NI>(sf, sb, wc * WN, lane);
    lds_frags<RK, BM, MI>(rf, sb + S_BYTES, wr * WM, lane);
  };
  auto mfmas = [&](const bf16x8 (&sf)[NI], const bf16x8 (&rf)[MI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf[i], rf[j], acc[i][j], 0, 0, 0);
  };
  // one pipeline step: `cur` holds tile t's fragments, `nxt` receives tile t+1's
  auto step = [&](int t, const bf16x8 (&csf)[NI], const bf16x8 (&crf)[MI], bf16x8 (&nsf)[NI], bf16x8 (&nrf)[MI]) {
    // issued so far: tiles <= min(nt-1, t+NSTAGE-1); tile t+1 must land, younger issued tiles may fly
    if (t + 1 < nt) wait_all_but(min(nt - 1, t + NSTAGE - 1) - (t + 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // tile t's fragment reads (and K-tail ds_writes) are done
    __builtin_amdgcn_s_barrier();                           // => slot of tile t is free, tile t+1 is complete
    if (t + NSTAGE < nt) stage(t + NSTAGE, smem + (t % NSTAGE) * STAGE);
    if (t + 1 < nt) read_frags(nsf, nrf, t + 1);
    mfmas(csf, crf);
  };

  // Steady state (tile t+NSTAGE is a full tile of the first K range, so every tile in flight is a DMA
  // tile and the counted wait is exact): ONE basic block per step -- the 4 LDS-DMA pieces of the tile
  // NSTAGE ahead and the 12 fragment reads of tile t+1 are spread between the 32 MFMAs of tile t, so
  // their issue cost (60-180 cycles per DMA piece) hides under the matrix pipe instead of preceding it.
  constexpr int SPW = (BN * BK * 2 / 1024) / (NT / 64), RPW = (BM * BK * 2 / 1024) / (NT / 64);
  uint32_t svoff[SPW], rvoff[RPW];
  dma_offsets<SK, BN, NT>(svoff, p.lds, p.N, n0, tid);
  dma_offsets<RK, BM, NT>(rvoff, p.ldr, p.M, m0, tid);
  const char* const sbase = reinterpret_cast<const char*>(p.S + (SK ? (long)n0 * p.lds : (long)n0));
  const char* const rbase = reinterpret_cast<const char*>(p.R + (RK ? (long)m0 * p.ldr : (long)m0));
  const long skstep = (SK ? 1 : p.lds) * 2, rkstep = (RK ? 1 : p.ldr) * 2;      // bytes per unit of k
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const int nfull1 = (kend > kbeg) ? (kend - kbeg) / BK : 0;
  auto fast_step = [&](int t, const bf16x8 (&csf)[NI], const bf16x8 (&crf)[MI], bf16x8 (&nsf)[NI], bf16x8 (&nrf)[MI]) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 2) * DMA_PER_TILE) : "memory");   // tile t+1 landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* buf = smem + (t % NSTAGE) * STAGE;
    const long k0 = kbeg + (long)(t + NSTAGE) * BK;
    const char* sb = smem + ((t + 1) % NSTAGE) * STAGE;
    if (SK && RK && NI == 4 && (MI % 4) == 0 && SPW == 2 && RPW == 2) {
      // four hard-fenced groups: 8 MFMAs + one 1-KiB DMA piece + a quarter of tile t+1's fragment reads
      typedef __attribute__((address_space(3))) void lds_void;
      typedef const __attribute__((address_space(1))) void gbl_void;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const char* ub = (g < 2) ? sbase + k0 * skstep : rbase + k0 * rkstep;
        const uint32_t vo = (g < 2) ? svoff[g & 1] : rvoff[g & 1];
        char* dst = buf + (g < 2 ? 0 : S_BYTES) + ((g & 1) * (NT / 64) + uwave) * 1024;
        __builtin_amdgcn_global_load_lds((gbl_void*)(ub + vo), (lds_void*)dst, 16, 0, 0);
        {
          bf16x8 one[1];
          lds_frags<true, BN, 1>(one, sb, wc * WN + 16 * g, lane); nsf[g] = one[0];
#pragma unroll
          for (int j = 0; j < MI / 4; ++j) {
            lds_frags<true, BM, 1>(one, sb + S_BYTES, wr * WM + 16 * (g * (MI / 4) + j), lane);
            nrf[g * (MI / 4) + j] = one[0];
          }
        }
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(csf[g], crf[j], acc[g][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      dma_tile_fast<BN, NT>(buf, sbase + k0 * skstep, svoff, uwave);
      dma_tile_fast<BM, NT>(buf + S_BYTES, rbase + k0 * rkstep, rvoff, uwave);
      read_frags(nsf, nrf, t + 1);
      mfmas(csf, crf);
    }
  };

  // prologue: NSTAGE tiles issued, tile 0 landed and in registers
#pragma unroll
  for (int t = 0; t < NSTAGE; ++t)
    if (t < nt) stage(t, smem + t * STAGE);
  bf16x8 sfA[NI], rfA[MI], sfB[NI], rfB[MI];
  if (nt > 0) {
    wait_all_but(min(nt - 1, NSTAGE - 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(sfA, rfA, 0);
  }
  int t = 0;
  for (; t + 1 + NSTAGE < nfull1; t += 2) {
    fast_step(t, sfA, rfA, sfB, rfB);
    fast_step(t + 1, sfB, rfB, sfA, rfA);
  }
  for (; t < nt; t += 2) {
    step(t, sfA, rfA, sfB, rfB);
    if (t + 1 < nt) step(t + 1, sfB, rfB, sfA, rfA);
  }

  // ---- epilogue: lane holds n = n0 + wc*WN + i*16 + (lane>>4)*4 + 0..3, m = m0 + wr*WM + j*16 + (lane&15)
  // bf16 output: the finished tile goes through LDS (the ring is free now) and leaves as whole rows,
  // 16 B per lane = 512 contiguous bytes per row of a 256-wide tile; direct 8-byte stores from the MFMA
  // layout (16 rows x 4 pieces per instruction) ran at ~1.8 TB/s and cost ~30 % of a K=1024 GEMM.
  constexpr int CROWB = BN * 2 + 16;              // padded LDS row of the C tile
  const int nq = (lane >> 4) * 4, ml = lane & 15;
  if (!OUTF32) __syncthreads();                   // every wave is done with the ring
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int nl = wc * WN + i * 16 + nq;
    const int n = n0 + nl;
    const bool nok = n < p.N;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && nok) {
      const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
      b4[0] = bb.x; b4[1] = bb.y; b4[2] = bb.z; b4[3] = bb.w;
    }
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int mloc = wr * WM + j * 16 + ml;
      const int m = m0 + mloc;
      const bool ok = nok && m < p.M;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * p.alpha + b4[e];
      if (OUTF32) {
        if (ok) {
          float* c = reinterpret_cast<float*>(p.C) + (long)z * p.slab_stride + (long)m * p.ldc + n;
          *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
        if (ok && p.res) {
          const uint2 r = *reinterpret_cast<const uint2*>(p.res + (long)m * p.ldres + n);
          v[0] += bf_lo(r.x); v[1] += bf_hi(r.x); v[2] += bf_lo(r.y); v[3] += bf_hi(r.y);
        }
        if (ok && p.aux) {
          const uint2 a = *reinterpret_cast<const uint2*>(p.aux + (long)m * p.ldaux + n);
          v[0] *= gelu_erf_grad_f(bf_lo(a.x)); v[1] *= gelu_erf_grad_f(bf_hi(a.x));
          v[2] *= gelu_erf_grad_f(bf_lo(a.y)); v[3] *= gelu_erf_grad_f(bf_hi(a.y));
        }
        *reinterpret_cast<uint2*>(smem + mloc * CROWB + nl * 2) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        if (ok && p.gelu_out) {
          // GELU of the bf16-ROUNDED pre-activation, so backward's gelu'(u) sees the same u
          float u0 = bf2f(f2bf(v[0])), u1 = bf2f(f2bf(v[1])), u2 = bf2f(f2bf(v[2])), u3 = bf2f(f2bf(v[3]));
          bf16_t* g = p.gelu_out + (long)m * p.ldg + n;
          *reinterpret_cast<uint2*>(g) = make_uint2(pack_bf2(gelu_erf_f(u0), gelu_erf_f(u1)),
                                                     pack_bf2(gelu_erf_f(u2), gelu_erf_f(u3)));
        }
      }
    }
  }
  if (!OUTF32) {
    __syncthreads();
    bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C);
    const bool wide = ((p.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
    constexpr int CPR = BN / 8;                   // 16-byte chunks per tile row
#pragma unroll 4
    for (int c = tid; c < BM * CPR; c += NT) {
      const int row = c / CPR, ch = c % CPR;
      const int m = m0 + row, n = n0 + ch * 8;
      if (m < p.M && n < p.N) {
        const uint4 val = *reinterpret_cast<const uint4*>(smem + row * CROWB + ch * 16);
        bf16_t* dst = Cb + (long)m * p.ldc + n;
        if (wide && n + 8 <= p.N) {
          *reinterpret_cast<uint4*>(dst) = val;
        } else {
          *reinterpret_cast<uint2*>(dst) = make_uint2(val.x, val.y);                       // N % 4 == 0: first half always fits
          if (n + 8 <= p.N) *reinterpret_cast<uint2*>(dst + 4) = make_uint2(val.z, val.w);
        }
      }
    }
  }
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

template <bool RK, bool SK, bool OUTF32, int BM, int BN, int NWM, int NWN>
int launch_cfg(GemmP p, int splits, hipStream_t st) {
  constexpr int S_BYTES = SK ? Tile<BN>::KC_BYTES : Tile<BN>::KS_BYTES;
  constexpr int R_BYTES = RK ? Tile<BM>::KC_BYTES : Tile<BM>::KS_BYTES;
  constexpr int RING = NSTAGE * (S_BYTES + R_BYTES), CTILE = BM * (BN * 2 + 16);
  constexpr int SMEM = OUTF32 ? RING : (RING > CTILE ? RING : CTILE);
  static bool attr_set = false;   // idempotent; a race only repeats the call
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<RK, SK, OUTF32, BM, BN, NWM, NWN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) UR_FAIL((int)e, "ur_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    attr_set = true;
  }
  p.gm = ur_cdiv(p.M, BM); p.gn = ur_cdiv(p.N, BN);
  dim3 grid(p.gm * p.gn, 1, splits);
  hipLaunchKernelGGL((gemm_kernel<RK, SK, OUTF32, BM, BN, NWM, NWN>), grid, dim3(NWM * NWN * 64), SMEM, st, p);
  UR_CHECK_LAUNCH("ur_gemm");
  return 0;
}

// Tile choice: 256x256 (8 waves, 130 FLOP per byte of L2 traffic) once the grid still fills the
// chip (>= 256 workgroups); otherwise the 128x128 tile (4 waves, 2 workgroups per CU).
template <bool RK, bool SK, bool OUTF32>
int launch(const GemmP& p, int splits, hipStream_t st) {
  const long big_wgs = (long)ur_cdiv(p.M, 256) * ur_cdiv(p.N, 256) * splits;
  if (p.M >= 256 && p.N >= 256 && big_wgs >= 256) return launch_cfg<RK, SK, OUTF32, 256, 256, 2, 4>(p, splits, st);
  return launch_cfg<RK, SK, OUTF32, 128, 128, 2, 2>(p, splits, st);
}

}  // namespace

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
  UR_REQUIRE(a->ldr >= (a->r_kcontig ? a->K : a->M) && a->lds >= (a->s_kcontig ? a->K : a->N) && a->ldc >= a->N,
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

  GemmP p;
  p.R = (const bf16_t*)a->R; p.S = (const bf16_t*)a->S; p.ldr = a->ldr; p.lds = a->lds; p.K = a->K;
  p.R2 = (const bf16_t*)a->R2; p.S2 = (const bf16_t*)a->S2; p.ldr2 = a->ldr2; p.lds2 = a->lds2; p.K2 = a->K2;
  p.C = a->C; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.alpha = a->alpha;
  p.bias = a->bias; p.res = (const bf16_t*)a->residual; p.ldres = a->ldres;
  p.gelu_out = (bf16_t*)a->gelu_out; p.ldg = a->ldg; p.aux = (const bf16_t*)a->gelu_grad_aux; p.ldaux = a->ldaux;
  p.gm = 0; p.gn = 0;   // set by launch_cfg for the chosen tile
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

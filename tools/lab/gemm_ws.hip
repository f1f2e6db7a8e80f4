// Wave-specialised persistent bf16 projection GEMM for gfx950: 128 x 256 output tiles, the epilogue of tile n runs BESIDE the K loop
// of tile n + 1 on the same CU (C = R S^T: R = activations [M,K], S = nn.Linear weight [N,K]; replaces nn.Linear at
// modeling_qwen3.py:81-83,227-238).  LAB KERNEL (ur_gemm_persistent_mode(2)): correct and tested, but NOT faster than gemm_pers.hip --
// see "Measured" below; no default path uses it.
//
// Why it was built: gemm_pers.hip's epilogues run with the MFMA pipe idle (plain: ~4.8 us beside a 23 us K loop at K = 1024; SwiGLU
// backward: 25 us beside 23), and their rate is a per-CU limit: de-phasing workgroups or XCDs does not help
// (profiles/r5_gemm_grid_sweep.txt).  Here a workgroup of 8 waves has three roles (gfx950 has ONE barrier per workgroup, so all
// of them execute the same barrier sequence, one barrier per phase of 16 MFMAs):
//   * waves 0-3, one per SIMD: MFMA only.  128 x 256 accumulator tile (128 registers per lane, gemm_pers.hip's wave tile); the fragments
//     of phase n + 1 are read from LDS under the MFMAs of phase n (a wave alone on its SIMD has nobody to hide behind); no vector-memory
//     instruction.  At the end of an output tile they round the accumulators to bf16 into a 64 KiB LDS stash (16-byte chunk c of row
//     r at chunk c ^ (r & 15): conflict-free for the MFMA layout's writes and for row-contiguous reads) and start the next K loop;
//   * waves 4-5: all LDS-DMA (2 ring slots of S 256 x 64 + R 128 x 64 = 96 KiB; the half tile read in phase n is refilled in phase
//     n + 1 with the K tile two ahead; counted vmcnt publishes the half tile the next phase reads);
//   * waves 6-7: take the stash apart in 32 steps of [8 rows x 128 bytes per wave instruction] spread over the next K loop; global
//     loads of the step 8 ahead are in flight under hand-counted waits, so their latency never meets a barrier.
// Scope: M % 128 == 0, N % 256 == 0, K % 512 == 0, K >= 1024, bf16 output, no LoRA terms; epilogues: plain (alpha), SwiGLU backward.
//
// Measured (one MI355X, M = 131072, tools/lab/gemm_ws_ab.py; docs/lab_notes.md 13.6): bit-identical to the persistent kernel (plain)
// and to ur_gemm + ur_swiglu_bwd (SwiGLU backward), and a FLAT ~1.2 PFLOP/s on every shape (persistent: 1.2-1.3 at K = 1024, 1.37-1.46
// at K >= 3072; SwiGLU backward 1.13 ms against 1.07).  Ablations: without its LDS-DMA the K loop runs 1.56-1.83 PFLOP/s, the DMA
// alone (no MFMA) moves the K = 1024 shapes' bytes at the equivalent of 1.93 and the K = 6144 shape's at 1.28; with the DMA issued by
// the MFMA waves themselves (first version) the numbers were the same.  A 128 x 256 tile needs 1.5x the L2 -> LDS bytes per FLOP of the
// 256 x 256 one, and the 64 KiB stash leaves 96 KiB of ring = two K tiles in flight: at the ~1.8 us an LDS-DMA piece takes to land beside
// running MFMAs that is 55 GB/s per CU = 1.2 PFLOP/s.  What it would take: a third ring slot (144 KiB) with the stash shrunk to a
// 16 KiB quarter tile that the epilogue waves pull into registers at once.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "gemm_common.hip.h"
#include "unirec_hip.h"

namespace {
using namespace urgemm;

constexpr int BK = 64, BM = 128, BN = 256;
constexpr int S_BYTES = BN * 128, R_BYTES = BM * 128;                                  // one ring slot = 32 + 16 KiB
constexpr int STASH = BM * BN * 2;                                                     // 64 KiB
// LDS map: S slots at 0 and 32 KiB, R slots at 64 and 80 KiB, the stash at 96 KiB.  (Slot-major would put the second slot's R tile
// beyond the 64 KiB reach of a ds_read offset from the first slot's base: hipcc then keeps ~24 address registers for the fragment
// reads; this way every S read is base + immediate from ONE register per k half, every R read from one more.)
constexpr int R_RING = 2 * S_BYTES, STASH_OFF = 2 * S_BYTES + 2 * R_BYTES;
constexpr int SMEM = STASH_OFF + STASH;                                                // 160 KiB: all of a CU's LDS
constexpr int NGROUP = 32;             // epilogue steps per output tile and wave

#ifndef UR_WS_ABLATE
#define UR_WS_ABLATE 0                 // lab builds only (WRONG results), bits: 1 = the epilogue waves only count barriers, 2 = no barriers at all (the epilogue
#endif                                 // waves leave at once), 4 = no LDS-DMA in the K loop, 8 = no fragment reads in the K loop, 16 = no MFMAs

template <int VM> __device__ __forceinline__ void phase_end() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): this phase's fragment reads are in their registers
#if !(UR_WS_ABLATE & 2)
  __builtin_amdgcn_s_barrier();
#endif
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void count_barrier() {
  __builtin_amdgcn_sched_barrier(0);
#if !(UR_WS_ABLATE & 2)
  __builtin_amdgcn_s_barrier();
#endif
  __builtin_amdgcn_sched_barrier(0);
}

// ================================================== the four MFMA waves ==================================================
// No vector-memory instruction at all: an LDS-DMA piece costs the issuing wave 60-185 cycles of its in-order instruction stream
// (MI355X_MICROARCH.md, per-instruction constants), and a wave alone on its SIMD has nobody to cover for it.
template <int EPI>
__device__ __forceinline__ void mfma_role(const GemmP& p, int ntiles, char* smem, int wc, int lane) {
  const int l15 = lane & 15, g4 = lane >> 4;
  // fragment reads: lane holds [row = 16 i + l15][k = 32 h + 8 g4 .. + 7], 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7)
  const uint32_t lo0 = l15 * 128 + (((g4) ^ ((l15 >> 1) & 7)) << 4), lo1 = l15 * 128 + (((4 + g4) ^ ((l15 >> 1) & 7)) << 4);
  // the four address registers of the fragment reads (opaque: every read is one of them + an immediate below 64 KiB)
  uint32_t as0 = lds_off(smem) + wc * 4096 + lo0, as1 = lds_off(smem) + wc * 4096 + lo1, ar0 = lds_off(smem) + R_RING + lo0, ar1 = lds_off(smem) + R_RING + lo1;
  asm volatile("" : "+v"(as0), "+v"(as1), "+v"(ar0), "+v"(ar1));
  typedef const __attribute__((address_space(3))) bf16x8* frag_ptr;
  const int nk = p.K / BK;
  const int gstride = gridDim.x;
  int tiles_left = (ntiles - (int)blockIdx.x + gstride - 1) / gstride;      // >= 1

  f32x4 acc[4][8];                       // [2 sh + ii][4 rh + jj]: columns sh*128 + wc*32 + ii*16 .., rows rh*64 + jj*16 ..
  bf16x8 R0[4][2], R1[4][2], SA[2][2], SB[2][2];
  auto rdR = [&](bf16x8 (&F)[4][2], int par, int rh) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      F[jj][0] = *(frag_ptr)(uintptr_t)(ar0 + (uint32_t)(par * R_BYTES + rh * 8192 + jj * 2048));
      F[jj][1] = *(frag_ptr)(uintptr_t)(ar1 + (uint32_t)(par * R_BYTES + rh * 8192 + jj * 2048));
    }
  };
  auto rdS = [&](bf16x8 (&F)[2][2], int par, int sh) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      F[ii][0] = *(frag_ptr)(uintptr_t)(as0 + (uint32_t)(par * S_BYTES + sh * 16384 + ii * 2048));
      F[ii][1] = *(frag_ptr)(uintptr_t)(as1 + (uint32_t)(par * S_BYTES + sh * 16384 + ii * 2048));
    }
  };
  auto quad = [&](const bf16x8 (&S)[2][2], const bf16x8 (&R)[4][2], auto shc, auto rhc) {
    constexpr int sh = decltype(shc)::value, rh = decltype(rhc)::value;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          f32x4& a = acc[2 * sh + ii][4 * rh + jj];
#if UR_WS_ABLATE & 16
          asm volatile("" : "+v"(a) : "v"(S[ii][h]), "v"(R[jj][h]));
          continue;
#endif
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S[ii][h], R[jj][h], a, 0, 0, 0);
        }
  };
  // the phase's instruction order: the next phase's fragment reads between the first MFMAs
  auto interleave = [&](auto ndsc) {
    constexpr int NDS = decltype(ndsc)::value;
#pragma unroll
    for (int i = 0; i < NDS; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x008, 16 - NDS, 0);
  };
  auto phase_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): the next phase's fragments are in their registers
    count_barrier();
  };
  const std::integral_constant<int, 0> c0;
  const std::integral_constant<int, 1> c1;
  const std::integral_constant<int, 4> c4;
  const std::integral_constant<int, 8> c8;
#if UR_WS_ABLATE & 8
#define UR_WS_RD(x) do { } while (0)
#else
#define UR_WS_RD(x) x
#endif
  // One K tile = 4 phases of 16 MFMAs; K tile kt lives in ring slot PAR = kt & 1 (nk is even).  Its S0 fragments arrive in Sa (read in
  // the previous K tile's phase 4), S1 goes to Sb, and the NEXT K tile's S0 to Sb again once phase 3 has used it: the two register sets
  // swap roles every K tile.  The fragments of phase n + 1 are read under the MFMAs of phase n.
  auto ktile = [&](auto parc) {
    constexpr int PAR = decltype(parc)::value;
    constexpr int slot = PAR, nslot = 1 - PAR;
    bf16x8 (&Sa)[2][2] = PAR ? SB : SA;
    bf16x8 (&Sb)[2][2] = PAR ? SA : SB;
    UR_WS_RD(rdS(Sb, slot, 1));  quad(Sa, R0, c0, c0); interleave(c4); phase_end();
    UR_WS_RD(rdR(R1, slot, 1));  quad(Sb, R0, c1, c0); interleave(c8); phase_end();
    UR_WS_RD(rdR(R0, nslot, 0)); quad(Sb, R1, c1, c1); interleave(c8); phase_end();
    UR_WS_RD(rdS(Sb, nslot, 0)); quad(Sa, R1, c0, c1); interleave(c4); phase_end();
  };

  count_barrier();                                         // K tiles 0 and 1 have landed (loader waves)
  rdR(R0, 0, 0);
  rdS(SA, 0, 0);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  count_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
  for (;;) {
#pragma unroll 1
    for (int kt = 0; kt < nk; kt += 2) {
      ktile(c0);
      ktile(c1);
    }
    // ---- hand the tile over: bf16, 8 consecutive columns per lane (pack + 16-lane swap), chunk c of row r at c ^ (r & 15) ----
    {
      // (lane constants from an opaque copy of the lane id: as loop invariants they would stay live across the K loop's 256 registers)
      int elane = lane;
      asm volatile("" : "+v"(elane));
      const int el15 = elane & 15, eg4 = elane >> 4;
      const float alpha = p.alpha;
      const int cs8 = (eg4 & 1) * 2 + (eg4 >> 1);            // this lane's 16-byte chunk within the wave's 32 columns
      const uint32_t sbase = lds_off(smem) + STASH_OFF + el15 * 512;
      typedef __attribute__((address_space(3))) u32x4_t* stash_ptr;
#pragma unroll
      for (int sh = 0; sh < 2; ++sh)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 a = acc[2 * sh][j], b = acc[2 * sh + 1][j];
          uint32_t a0 = pack_bf2(a[0] * alpha, a[1] * alpha), a1 = pack_bf2(a[2] * alpha, a[3] * alpha);
          uint32_t b0 = pack_bf2(b[0] * alpha, b[1] * alpha), b1 = pack_bf2(b[2] * alpha, b[3] * alpha);
          swap16(a0, b0); swap16(a1, b1);
          const int c16 = sh * 16 + wc * 4 + cs8;
          const u32x4_t v = {a0, a1, b0, b1};
          *(stash_ptr)(uintptr_t)(sbase + (uint32_t)(((j >> 2) * 64 + (j & 3) * 16) * 512) + (uint32_t)((c16 ^ el15) << 4)) = v;
          // (cleared here rather than by a zero C operand in the next tile's first K tile: a second copy of the K tile body with its own
          // register assignment made hipcc spill accumulators)
          acc[2 * sh][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[2 * sh + 1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      count_barrier();
    }
    tiles_left -= 1;
    if (tiles_left == 0) break;
  }
}

// ================================================== the two loader waves ==================================================
// All LDS-DMA of the workgroup: K tile kt + 2 of the stream (R half 0: kt + 3) behind the barrier that frees its half tile, the counted
// vmcnt that publishes the half tile the next phase reads, the same barrier sequence as everybody else.
__device__ __forceinline__ void loader_role(const GemmP& p, const TileOrder& ord, int ntiles, char* smem, int ld, int lane) {
  // wave instruction `inst` = li * 2 + ld moves rows inst * 8 .. + 7 of an operand tile (128 B each); this lane fetches row (lane >> 3)
  // of them, k chunk kch (the swizzle -- chunk c of row r at c ^ ((r >> 1) & 7) -- lives on the source address)
  const int kch = (lane & 7) ^ (((lane >> 4) + 4 * (ld & 1)) & 7);
  const int vo_s = (int)((lane >> 3) * p.lds * 2 + kch * 16), vo_r = (int)((lane >> 3) * p.ldr * 2 + kch * 16);
  const int nk = p.K / BK;
  const int gstride = gridDim.x;
  int vid = blockIdx.x, m0, n0;
  { int bm, bn; tile_coords(ord, vid, bm, bn); m0 = bm * BM; n0 = bn * BN; }
  int tiles_left = (ntiles - (int)blockIdx.x + gstride - 1) / gstride;
  // sources as raw buffers: the lane offset is ONE 32-bit register per operand, everything else (output tile, K tile, piece) rides in
  // the scalar offset.  S: one descriptor over the wave's rows of the whole weight (N * lds * 2 < 2^31: host); R: one per output tile.
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.S + (long)(ld * 8) * p.lds), 0, 0x7fffffff, 0x00020000);
  auto r_base = [&](int m) { return reinterpret_cast<const char*>(p.R + (long)(m + ld * 8) * p.ldr); };
  const int lds2 = (int)(p.lds * 2);
  int cs = n0 * lds2, ns = cs;                             // scalar byte offset of K = 0 of the current / next output tile's S rows
  const char* cr = uniform_ptr(r_base(m0));
  const char* nr = cr;
  const int sp = 16 * lds2, rp = 16 * (int)(p.ldr * 2);    // bytes between the wave's pieces (16 rows)
  // the wave's pieces of half `hf` of an S tile (8 pieces: rows 128 hf ..) / R tile (4 pieces: rows 64 hf ..)
  auto dma_s = [&](int par, int hf, int soff) {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int li = 8 * hf + d;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lds_void*)(smem + par * S_BYTES + (li * 2 + ld) * 1024), 16, vo_s, soff + li * sp, 0, 0);
    }
  };
  auto dma_r = [&](int par, int hf, const char* base, int soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int li = 4 * hf + d;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + R_RING + par * R_BYTES + (li * 2 + ld) * 1024), 16, vo_r, soff + li * rp, 0, 0);
    }
  };
  auto src_s = [&](int kt, int ahead) { const int k = kt + ahead; return k < nk ? cs + k * (BK * 2) : ns + (k - nk) * (BK * 2); };
  auto rb_of = [&](int kt, int ahead) { return (kt + ahead) < nk ? cr : nr; };
  auto ro_of = [&](int kt, int ahead) { const int k = kt + ahead; return (k < nk ? k : k - nk) * (BK * 2); };
#if UR_WS_ABLATE & 4
#define UR_WS_DMA(x) do { } while (0)
#else
#define UR_WS_DMA(x) x
#endif
  // vmcnt at the end of a phase = the pieces issued after the half tile that the NEXT phase reads (8, 8, 4, 4 per phase and wave)
  auto ktile = [&](auto parc, int kt) {
    constexpr int slot = decltype(parc)::value, nslot = 1 - slot;
    UR_WS_DMA(dma_s(slot, 0, src_s(kt, 2)));                  phase_end<36>();
    UR_WS_DMA(dma_s(slot, 1, src_s(kt, 2)));                  phase_end<40>();
    UR_WS_DMA(dma_r(slot, 1, rb_of(kt, 2), ro_of(kt, 2)));    phase_end<36>();
    UR_WS_DMA(dma_r(nslot, 0, rb_of(kt, 3), ro_of(kt, 3)));   phase_end<32>();
  };
  // prologue: K tiles 0 and 1 whole; once the MFMA waves hold the first fragments, R half 0 of K tile 2 (what phase 4 of "K tile -1" would have issued)
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    dma_s(tt, 0, cs + tt * (BK * 2)); dma_s(tt, 1, cs + tt * (BK * 2)); dma_r(tt, 0, cr, tt * (BK * 2)); dma_r(tt, 1, cr, tt * (BK * 2));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  count_barrier();
  count_barrier();
  dma_r(0, 0, cr, 2 * (BK * 2));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (;;) {
    int nm0 = m0, nn0 = n0;
    if (tiles_left > 1) {
      vid += gstride;
      int bm, bn; tile_coords(ord, vid, bm, bn); nm0 = bm * BM; nn0 = bn * BN;
    }
    ns = nn0 * lds2; nr = uniform_ptr(r_base(nm0));
#pragma unroll 1
    for (int kt = 0; kt < nk; kt += 2) {
      ktile(std::integral_constant<int, 0>{}, kt);
      ktile(std::integral_constant<int, 1>{}, kt + 1);
    }
    count_barrier();                                          // hand-over
    tiles_left -= 1;
    if (tiles_left == 0) break;
    m0 = nm0; n0 = nn0; cs = ns; cr = nr;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the never-consumed tail of the stream has landed before the wave leaves
}

// ================================================== the two epilogue waves ==================================================
// step g (0 .. 31) of a tile: rows ew*64 + (g >> 3)*16 + {0..3, 8..11} + 4*((g >> 2) & 1), columns (g & 3)*64 .. + 63:
// lane = (row (lane >> 4) + 8 ((lane >> 3) & 1), chunk lane & 7): 8 rows x 128 contiguous bytes per wave instruction
struct EpiLane { int row; int c16; uint32_t lds_off; };
__device__ __forceinline__ EpiLane epi_lane(int ew, int lane, int g) {
  EpiLane e;
  e.row = ew * 64 + (g >> 3) * 16 + ((lane >> 3) & 1) * 8 + (lane >> 4) + 4 * ((g >> 2) & 1);
  e.c16 = (g & 3) * 8 + (lane & 7);
  e.lds_off = (uint32_t)(e.row * 512 + ((e.c16 ^ (e.row & 15)) << 4));
  return e;
}

template <int EPI>
__device__ __forceinline__ void epi_role(const GemmP& p, const TileOrder& ord, int ntiles, char* smem, int ew, int lane) {
  const int nk = p.K / BK, bstep = nk * 4 / NGROUP;        // barriers (phases) per epilogue step: >= 2 (K % 512 == 0: host)
  const int gstride = gridDim.x;
  const char* stash = smem + STASH_OFF;
  int vid = blockIdx.x, m0, n0;
  { int bm, bn; tile_coords(ord, vid, bm, bn); m0 = bm * BM; n0 = bn * BN; }
  int tiles_left = (ntiles - (int)blockIdx.x + gstride - 1) / gstride;
  count_barrier();
  count_barrier();

  // SwiGLU backward: gate / up pieces of the step 8 ahead in flight (two 16-byte pieces per lane and step).
  // The loads are inline asm and every wait is written by hand: hipcc's wait-count pass merges the states of the tile loop's back edge
  // conservatively (vmcnt(15) where 28 operations may stay in flight: the wave then waits for stores issued long before, arrives
  // late at the barrier and stalls the MFMA waves).  In steady state step s waits for the pieces issued at step s - 8; behind them
  // the queue holds 7 x {2 stores, 2 loads}.
  u32x4_t gq[8], uq[8];
  const long up_g = (long)p.sw_I * 2;
  auto load_gu = [&](int tm0, int tn0, int g, u32x4_t& gv, u32x4_t& uv) {
    const EpiLane e = epi_lane(ew, lane, g);
    const char* gb = reinterpret_cast<const char*>(p.sw_gu + (long)(tm0 + e.row) * p.sw_ldgu + tn0 + e.c16 * 8);
    const char* ub = gb + up_g;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off" : "=&v"(gv), "=&v"(uv) : "v"(gb), "v"(ub) : "memory");
  };
  // one step; BAR: two of the K loop's barriers are counted inside it (the SwiGLU arithmetic in two halves)
  auto step = [&](auto barc, auto vmc, int tm0, int tn0, int g, u32x4_t& gv, u32x4_t& uv) {
    constexpr bool BAR = decltype(barc)::value;
    constexpr int VM = decltype(vmc)::value;
    if constexpr (EPI == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(gv), "+v"(uv) : "n"(VM) : "memory");
    const EpiLane e = epi_lane(ew, lane, g);
    const u32x4_t d = *reinterpret_cast<const u32x4_t*>(stash + e.lds_off);
    if constexpr (EPI == 0) {
      char* cb = reinterpret_cast<char*>(p.C) + ((long)(tm0 + e.row) * p.ldc + tn0 + e.c16 * 8) * 2;
      __builtin_nontemporal_store(d, (__attribute__((address_space(1))) u32x4_t*)(cb));
      if constexpr (BAR) { count_barrier(); count_barrier(); }
    } else {
      // d(act) = the bf16-rounded product (what the unfused pair ur_gemm + ur_swiglu_bwd computes): dgate = d u silu'(g), dup = d silu(g)
      const uint32_t gw[4] = {gv[0], gv[1], gv[2], gv[3]}, uw[4] = {uv[0], uv[1], uv[2], uv[3]}, dw[4] = {d[0], d[1], d[2], d[3]};
      uint32_t og[4], ou[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float dgv[2], duv[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const float gg = hh ? bf_hi(gw[i]) : bf_lo(gw[i]), uu = hh ? bf_hi(uw[i]) : bf_lo(uw[i]), dd = hh ? bf_hi(dw[i]) : bf_lo(dw[i]);
          const float sg = sigmoid_f(gg);
          duv[hh] = dd * (gg * sg);
          dgv[hh] = dd * uu * (sg * (1.0f + gg * (1.0f - sg)));
        }
        og[i] = pack_bf2(dgv[0], dgv[1]); ou[i] = pack_bf2(duv[0], duv[1]);
        if constexpr (BAR) { if (i == 1) count_barrier(); }
      }
      char* db = reinterpret_cast<char*>(p.sw_dgu + (long)(tm0 + e.row) * p.sw_lddgu + tn0 + e.c16 * 8);
      const u32x4_t vg = {og[0], og[1], og[2], og[3]}, vu = {ou[0], ou[1], ou[2], ou[3]};
      st_g<u32x4_t>(db, vg);
      st_g<u32x4_t>(db + up_g, vu);
    }
  };

  if constexpr (EPI == 1) {
#pragma unroll
    for (int g = 0; g < 8; ++g) load_gu(m0, n0, g, gq[g], uq[g]);
  }
  // the K loop of tile 0: nothing to take apart yet
#pragma unroll 1
  for (int i = 0; i < nk * 4; ++i) count_barrier();
  count_barrier();                                            // hand-over: tile 0 is in the stash
  if constexpr (EPI == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the first eight steps' pieces (issued a K loop ago): from here on the queue is in steady state
  int pm0 = m0, pn0 = n0;
  tiles_left -= 1;
#pragma unroll 1
  while (tiles_left > 0) {
    vid += gstride;
    { int bm, bn; tile_coords(ord, vid, bm, bn); m0 = bm * BM; n0 = bn * BN; }
#pragma unroll
    for (int g = 0; g < NGROUP; ++g) {
#if !(UR_WS_ABLATE & 1)
      step(std::true_type{}, std::integral_constant<int, 28>{}, pm0, pn0, g, gq[g & 7], uq[g & 7]);
      if constexpr (EPI == 1) {
        if (g < NGROUP - 8) load_gu(pm0, pn0, g + 8, gq[g & 7], uq[g & 7]);
        else load_gu(m0, n0, g + 8 - NGROUP, gq[g & 7], uq[g & 7]);
        count_barrier();
      }
#else
      count_barrier(); count_barrier();
#endif
#pragma unroll 1
      for (int i = 2; i < bstep; ++i) count_barrier();
    }
    count_barrier();                                          // hand-over
    pm0 = m0; pn0 = n0;
    tiles_left -= 1;
  }
  // the last tile: the MFMA waves have left (its last eight steps issue no loads: two operations fewer behind the awaited pieces per step)
#if !(UR_WS_ABLATE & 1)
#pragma unroll
  for (int g = 0; g < NGROUP; ++g) {
    if (g <= NGROUP - 8) step(std::false_type{}, std::integral_constant<int, 28>{}, pm0, pn0, g, gq[g & 7], uq[g & 7]);
    else step(std::false_type{}, std::integral_constant<int, 0>{}, pm0, pn0, g, gq[g & 7], uq[g & 7]);
    if constexpr (EPI == 1) { if (g < NGROUP - 8) load_gu(pm0, pn0, g + 8, gq[g & 7], uq[g & 7]); }
  }
#endif
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_ws_kernel(GemmP p, TileOrder ord, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int uwave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (uwave < 4) mfma_role<EPI>(p, ntiles, smem, uwave, lane);
  else if (uwave < 6) loader_role(p, ord, ntiles, smem, uwave - 4, lane);
#if !(UR_WS_ABLATE & 2)
  else epi_role<EPI>(p, ord, ntiles, smem, uwave - 6, lane);
#endif
}

template <int EPI>
int launch_ws(const GemmP& p, hipStream_t st) {
  static std::atomic<uint64_t> attr_set{0};   // per device
  UR_ONCE_PER_DEVICE(attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) UR_FAIL((int)e, "ur_gemm(wave-specialised): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  }
  int ncu = ur_device_cu_count();
  ncu -= ncu % 8;
  if (ncu <= 0) ncu = 8;
  const int ntiles = p.gm * p.gn;
  const int grid = ntiles < ncu ? ntiles : ncu;
  auto magic = [](int d) { return (uint32_t)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
  TileOrder o;
  o.nwg = ntiles; o.gn = p.gn; o.gcw = p.gcw;
  o.rows_x = (ntiles >> 3) / p.gn; o.per = o.rows_x * (p.gcw > 0 ? p.gcw : 1);
  o.m_gn = magic(p.gn); o.m_per = magic(o.per > 0 ? o.per : 1); o.m_gcw = magic(p.gcw > 0 ? p.gcw : 1);
  hipLaunchKernelGGL((gemm_ws_kernel<EPI>), dim3(grid), dim3(512), SMEM, st, p, o, ntiles);
  UR_CHECK_LAUNCH("ur_gemm(wave-specialised)");
  return 0;
}

}  // namespace

namespace urgemm {

// (called for launches gemm_pers_eligible has accepted: alignment, 32-bit offsets, tile counts below the fdiv limits for 256-row tiles)
bool gemm_ws_eligible(const GemmP& p) {
  if ((p.M % BM) || (p.N % BN) || (p.K % (BK * NGROUP / 4)) || p.K < BK * NGROUP / 2) return false;      // an epilogue step every >= 2 phases
  if (p.K2 > 0 || p.bias || p.res || p.gelu_out || p.aux || p.qk_q || p.sp_act || p.sw_mode == 2) return false;
  if ((long)(p.M / BM) * (p.N / BN) >= (1L << 20) || p.M / BM >= (1 << 15)) return false;
  if ((long)p.N * p.lds * 2 >= (1L << 31) || p.ldr * 64 >= (1L << 31)) return false;      // 32-bit scalar / lane offsets of the loader waves' buffer descriptors
  return true;
}

int gemm_ws_launch(GemmP p, hipStream_t st) {
  p.gm = p.M / BM; p.gn = p.N / BN;
  p.gcw = 0;
  {
    static const int env_cw = ur_lab_int("UR_WS_CW", -1);
    const int cw = env_cw >= 0 ? env_cw : ((p.gn % 6) == 0 && p.gn >= 24 ? 6 : 4);
    if (cw > 0 && p.gn >= 16 && (p.gm % 8) == 0 && p.gn > cw && (p.gn % cw) == 0) p.gcw = cw;
  }
  if (p.sw_mode == 1) return launch_ws<1>(p, st);
  return launch_ws<0>(p, st);
}

}  // namespace urgemm

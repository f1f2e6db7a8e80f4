// Persistent 256x256 bf16 projection GEMM for gfx950: the K-contiguous forward / frozen-weight dX launches of the Qwen3 decoder
// (C = R S^T: R = activations [M,K], S = nn.Linear weight [N,K]; replaces nn.Linear at modeling_qwen3.py:81-83,227-238).
//
// Why a second kernel: gemm.hip's 256x256 kernel pays a fixed ~11.6 us per output tile (first-DMA latency, LDS-staged epilogue,
// the gap between a workgroup's exit and its successor's entry on the CU) -- a third of a tile's life at K = 1024.  Here ONE
// workgroup per CU walks its output tiles and the LDS-DMA ring never drains:
//   * the K tiles of ALL of a workgroup's output tiles form one stream; K tile q+2 (which may belong to the NEXT output tile)
//     is issued while K tile q is consumed, exactly as inside a tile, so a new tile starts with its first two K tiles landed;
//   * the epilogue runs from the accumulator registers with no LDS staging and no barrier: v_permlane16_swap turns the MFMA
//     layout (4 consecutive columns per lane) into 8 consecutive columns per lane = one 16-byte store per lane and 16x16
//     sub-tile pair, 16 rows x 64 contiguous bytes per store instruction; stores stay in flight under the next tile's K loop;
//   * the first K tile of a tile multiplies into a zero C operand (no accumulator clearing);
//   * the LoRA second K range (t B^T, K2 <= 64) is one more K tile of the stream: lanes whose k chunk lies beyond K2 fetch
//     from a 16-byte zero word, so no padded copies of t / B are needed.
// The steady state is gemm.hip's 8-phase ping-pong (two wave groups one barrier interval apart, half tiles refilled one phase
// after their last read, counted vmcnt), see the comments there; the hazards below refer to it.
// Scope: M % 256 == 0, N % 256 == 0, K % 64 == 0, K >= 256, bf16 output, epilogues: plain, bias, residual, masked LoRA
// (dropout), SwiGLU backward.  Everything else stays on gemm.hip.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "gemm_common.hip.h"
#include "unirec_hip.h"

namespace {
using namespace urgemm;

#ifndef UR_PERS_ABLATE
#define UR_PERS_ABLATE 0          // lab builds only (WRONG results): 1 = epilogue without its stores, 2 = no epilogue, 3 = vmcnt(12) in every K tile, 4 = plain (not non-temporal) stores
#endif
#ifndef UR_PERS_JOIN
#define UR_PERS_JOIN 0            // lab: 1 = the two wave groups re-join for EVERY epilogue (measured neutral for the plain one: the CU's store rate under load, ~17 B/clk, is the limit either way)
#endif
#ifndef UR_PERS_EPI_WAIT
#define UR_PERS_EPI_WAIT 0        // lab: 1 = the epilogue starts with s_waitcnt vmcnt(0): the next tile's two prefetched K tiles have landed before the first C store
#endif
#ifndef UR_PERS_NT
#define UR_PERS_NT 0              // lab: 1 = the epilogues' read-once operands (gate | up of the SwiGLU backward, residual / GELU aux) are loaded non-temporally
#endif
#if UR_PERS_NT
#define UR_LD_STREAM ld_g_nt
#else
#define UR_LD_STREAM ld_g
#endif
#ifndef UR_PERS_REMAP
#define UR_PERS_REMAP 1           // 0 = the 128-apart column groups and 16 rows x 64 B stores everywhere (A/B builds)
#endif
#ifndef UR_PERS_STAMPS
#define UR_PERS_STAMPS 0          // lab builds only: n > 0 = waves 0 and 4 of every workgroup log s_memtime around their n-th output tile (ur_lab_pers_stamps)
#endif
#if UR_PERS_STAMPS
__device__ long long g_pers_stamps[256 * 2 * 8];
#define UR_PSTAMP(k) do { if (tile_ord == UR_PERS_STAMPS && lane == 0 && (uwave & 3) == 0 && blockIdx.x < 256) g_pers_stamps[(blockIdx.x * 2 + (uwave >> 2)) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define UR_PSTAMP(k) do { } while (0)
#endif
constexpr int BK = 64, BM = 256, BN = 256;
constexpr int S_BYTES = BN * 128, R_BYTES = BM * 128, STAGE = S_BYTES + R_BYTES;      // one ring slot = 64 KiB

// EPI: 0 = plain (alpha only), 2 = residual (+ bias), 5 = bias only, 1 = SwiGLU backward (result is d(act); dgate | dup leave instead of C),
//      3 = q/k-norm + RoPE of the q|k|v projection (ur_gemm_args.qkr_*; q_r, k_r, v and the row constants leave instead of C),
//      4 = SwiGLU forward of the merged gate|up projection with 128-row interleaved weights (ur_gemm_args.swp_*).
// MODE: 0 = no second K range, 1 = the LoRA second K range rides in the K stream (one zero-padded K tile per output tile),
//       2 = masked rank-16 LoRA epilogue (dX under LoRA dropout; the second operand pair is read by the epilogue only).
template <int EPI, int MODE>
__global__ __launch_bounds__(512, 2) void gemm_pers_kernel(GemmP p, TileOrder ord, int ntiles) {
  constexpr bool DROP = MODE == 2, K2S = MODE == 1;
  // REMAP: a wave's two 32-column groups are ADJACENT output columns (n0 + wc*64 + sh*32 ..) instead of 128 apart, so that one store
  // instruction can cover 8 rows x 128 contiguous bytes -- whole lines -- where the old shape wrote 16 rows x 64 bytes (store-only
  // kernels: 5.8-6.0 TB/s against 3.3-4.0, tools/lab/store_lab.hip).  The LDS image does not change: LDS row sh*128 + wc*32 + x of the
  // S tile is FILLED from weight row n0 + wc*64 + sh*32 + x (a different scalar offset per LDS-DMA piece, nothing else).
  constexpr bool REMAP = UR_PERS_REMAP && EPI != 3 && EPI != 4;      // (EPI 3 / 4: the tile's 128-column halves are two heads / gate | up)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = uwave >> 2, wc = uwave & 3;
  const int l15 = lane & 15, g4 = lane >> 4;
  // fragment reads: lane holds [row = 16 i + l15][k = 32 h + 8 g4 .. + 7], 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7)
  const uint32_t lo0 = l15 * 128 + (((g4) ^ ((l15 >> 1) & 7)) << 4), lo1 = l15 * 128 + (((4 + g4) ^ ((l15 >> 1) & 7)) << 4);
  // LDS-DMA: wave instruction `inst` = li * 8 + wave moves rows inst * 8 .. + 7 of an operand tile (128 B each); this lane
  // fetches row (lane >> 3) of them, k chunk kch (the swizzle lives on the source address)
  const int kch = (lane & 7) ^ (((lane >> 4) + 4 * (uwave & 1)) & 7);
  const uint32_t vo_s = (uint32_t)((lane >> 3) * p.lds * 2 + kch * 16), vo_r = (uint32_t)((lane >> 3) * p.ldr * 2 + kch * 16);
  const int nk1 = p.K / BK;                                // >= 4 (host)
  const int nkt = nk1 + (K2S ? 1 : 0);
  const int gstride = gridDim.x;

  // Lab switch (UR_PERS_STAGGER = cycles per step, default 0 = off): workgroup w starts (w / 8) % 16 steps late.  Built to test
  // whether the store bursts of the epilogues (all CUs reach them together) are what makes them slow: they are not -- de-phased
  // epilogues take as long (in-kernel stamps: 3.85 k cycles per wave group either way), and the de-phasing costs the operand
  // panels their L2 sharing: CUs that read the same A / W panel a few K tiles apart no longer hit the lines their neighbours
  // fetched (rocprofv3 FETCH_SIZE of the N = 1024 launches x1.5 of A + W in lockstep, x2.9-4.3 staggered).  See DESIGN.md 11.
#if UR_LAB
  if (p.stagger != 0) {
    // p.stagger < 0 (UR_PERS_STAGGER_XCD): whole XCDs are de-phased instead (ids equal mod 8 share an XCD and its L2): the 32
    // workgroups of an XCD stay in lockstep on their shared panels, the eight XCDs reach their epilogues 1/8 period apart
    const int steps = p.stagger < 0 ? (int)(blockIdx.x & 7) : (int)((blockIdx.x >> 3) & 15);
    const long long until = (long long)__builtin_readcyclecounter() + (long long)steps * (p.stagger < 0 ? -p.stagger : p.stagger);
    while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(8);
  }
#endif
  // ---- producer: where the K tile that is fetched next comes from (scalars + one lane offset per operand) ----
  int vid = blockIdx.x, m0, n0;                            // current output tile
  { int bm, bn; tile_coords(ord, vid, bm, bn); m0 = bm * BM; n0 = bn * BN; }
  int tiles_left = (ntiles - (int)blockIdx.x + gstride - 1) / gstride;      // >= 1: the grid never exceeds ntiles
  auto s_base = [&](int n) { return reinterpret_cast<const char*>(p.S + (long)(n + (REMAP ? (uwave & 3) * 8 + (uwave >> 2) * 64 : uwave * 8)) * p.lds); };
  auto r_base = [&](int m) { return reinterpret_cast<const char*>(p.R + (long)(m + uwave * 8) * p.ldr); };
  const char* ubs = uniform_ptr(s_base(n0));               // K tile being fetched: uniform bases of this wave's first piece
  const char* ubr = uniform_ptr(r_base(m0));
  const long sp_main = 64 * p.lds * 2, rp_main = 64 * p.ldr * 2;            // bytes between the wave's pieces (64 rows)
  // the two LDS-DMA pieces of this wave for half `hf` (rows 128 hf ..) of the S or R tile at (ubs, ubr)
  auto dma_half = [&](char* slot, auto is_s, int hf, auto zc) {
    constexpr bool IS_S = decltype(is_s)::value, Z = decltype(zc)::value;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int li = 2 * hf + d;
      char* dst = slot + (IS_S ? 0 : S_BYTES) + (li * 8 + uwave) * 1024;
      if constexpr (!Z) {
        // (REMAP: piece li = 2 hf + d of the wave holds LDS rows 64 li + 8 wave ..: 32-row group (li >> 1) * 4 + (li & 1) * 2 + (wave >> 2)
        // = (sh, wc') -> weight rows (wc' * 2 + sh) * 32 ..)
        const long s_off = REMAP ? (long)(d * 128 + hf * 32) * p.lds * 2 : li * sp_main;
        const char* src = (IS_S ? ubs + s_off : ubr + li * rp_main) + (IS_S ? vo_s : vo_r);
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
      } else {
        // the K2 tile: rows of lds2 / ldr2 elements; k chunks at or beyond K2 must read as zero.  A raw buffer descriptor over the
        // wave's 256-row window does that in hardware: such lanes get an offset past num_records and an out-of-range buffer
        // load returns 0 -- one v_cndmask per piece and 32-bit offsets (per-lane 64-bit pointers with a select against a zero
        // word cost this K tile body ~30 more registers, and hipcc spilled inside it)
        const long ld2 = IS_S ? p.lds2 : p.ldr2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(IS_S ? ubs : ubr), 0, (int)(256 * ld2 * 2), 0x00020000);
        // ONE lane offset per operand (the piece's 64-row step rides in the scalar offset): eight precomputed per-piece offsets
        // did not fit beside the loop's registers -- hipcc spilled five and reloaded each behind an s_waitcnt vmcnt(0)
        // (derived from an opaque copy of the lane id at every use: as a loop invariant hipcc keeps it across the K loop, spills it,
        // and waits vmcnt(0) on the scratch reload in front of the DMA)
        int zl = lane;
        asm volatile("" : "+v"(zl));
        const int zk = (zl & 7) ^ (((zl >> 4) + 4 * (uwave & 1)) & 7);
        const uint32_t vo = (zk * 8 < p.K2) ? (uint32_t)((zl >> 3) * ld2 * 2 + zk * 16) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, vo, (int)((IS_S && REMAP ? d * 128 + hf * 32 : li * 64) * ld2 * 2), 0, 0);
      }
    }
  };

  // ---- consumer ----
  f32x4 acc[4][8];                       // [2 sh + ii][4 rh + jj]: columns sh*128 + wc*32 + ii*16 .., rows rh*128 + wr*64 + jj*16 ..
  bf16x8 R0[4][2], R1[4][2], S0[2][2], S1[2][2];
  auto rdR = [&](bf16x8 (&F)[4][2], const char* slot, int rh) {
    const char* b = slot + S_BYTES + (rh * 128 + wr * 64) * 128;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      F[jj][0] = *reinterpret_cast<const bf16x8*>(b + jj * 2048 + lo0);
      F[jj][1] = *reinterpret_cast<const bf16x8*>(b + jj * 2048 + lo1);
    }
  };
  auto rdS = [&](bf16x8 (&F)[2][2], const char* slot, int sh) {
    const char* b = slot + (sh * 128 + wc * 32) * 128;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      F[ii][0] = *reinterpret_cast<const bf16x8*>(b + ii * 2048 + lo0);
      F[ii][1] = *reinterpret_cast<const bf16x8*>(b + ii * 2048 + lo1);
    }
  };
  auto quad = [&](const bf16x8 (&S)[2][2], const bf16x8 (&R)[4][2], auto shc, auto rhc, auto firstc) {
    constexpr int sh = decltype(shc)::value, rh = decltype(rhc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          f32x4& a = acc[2 * sh + ii][4 * rh + jj];
          if (FIRST && h == 0) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S[ii][h], R[jj][h], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(S[ii][h], R[jj][h], a, 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  auto mat_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  const std::integral_constant<int, 0> c0;
  const std::integral_constant<int, 1> c1;
  // One K tile = 4 phases (gemm.hip).  VM = the counted wait of a load segment: 12 = the two pieces of each of the six phases
  // issued after the half tile that the NEXT phase reads; in the first K tile after an epilogue the epilogue's >= 16 stores
  // (issued after the pieces of the previous K tile, before this one's) sit in the queue as well: 28.
  auto ktile = [&](char* slot, const char* nslot, auto firstc, auto vmc, auto zc) {
    constexpr int VM = decltype(vmc)::value;
    auto seg_end = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    // phase 1
    rdS(S0, slot, 0);
    dma_half(slot, std::false_type{}, 0, zc);
    seg_end();
    quad(S0, R0, c0, c0, firstc);
    mat_end();
    // phase 2
    rdS(S1, slot, 1);
    dma_half(slot, std::true_type{}, 0, zc);
    seg_end();
    quad(S1, R0, c1, c0, firstc);
    mat_end();
    // phase 3
    rdR(R1, slot, 1);
    dma_half(slot, std::true_type{}, 1, zc);
    seg_end();
    quad(S1, R1, c1, c1, firstc);
    mat_end();
    // phase 4
    rdR(R0, nslot, 0);
    dma_half(slot, std::false_type{}, 1, zc);
    seg_end();
    quad(S0, R1, c0, c1, firstc);
    mat_end();
  };

  // prologue: the first two K tiles in consumption order R0 S0 S1 R1 (the loop's issue order shifted back 8 phases)
#pragma unroll 1
  for (int tt = 0; tt < 2; ++tt) {
    char* slot = smem + tt * STAGE;
    dma_half(slot, std::false_type{}, 0, std::false_type{}); __builtin_amdgcn_sched_barrier(0);
    dma_half(slot, std::true_type{}, 0, std::false_type{});  __builtin_amdgcn_sched_barrier(0);
    dma_half(slot, std::true_type{}, 1, std::false_type{});  __builtin_amdgcn_sched_barrier(0);
    dma_half(slot, std::false_type{}, 1, std::false_type{}); __builtin_amdgcn_sched_barrier(0);
    ubs = uniform_ptr(ubs + BK * 2); ubr = uniform_ptr(ubr + BK * 2);
  }
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");       // R half 0 and S half 0 of K tile 0 have landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  rdR(R0, smem, 0);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();               // stagger: this group now runs one barrier interval behind
  __builtin_amdgcn_sched_barrier(0);

  int q = 0;                                               // K tiles consumed so far: K tile q lives in ring slot q & 1
  bool first_tile = true;
#if UR_PERS_STAMPS
  int tile_ord = 0;
#endif
  // consumer K tile kt fetches K tile kt + 2 of the stream: the main K tiles of this output tile up to kt = nk1 - 3, then
  // (MODE 1) its K2 tile, then K tiles 0 and 1 of the NEXT output tile (past the last one: of this one again, never consumed)
  const int sw = nkt - 2;
#pragma unroll 1
  for (;;) {
#if UR_PERS_STAMPS
    ++tile_ord;
#endif
    UR_PSTAMP(0);
    int nm0 = m0, nn0 = n0;
    if (tiles_left > 1) {
      vid += gstride;
      int bm, bn; tile_coords(ord, vid, bm, bn); nm0 = bm * BM; nn0 = bn * BN;
    }
    {
      char* slot = smem + (q & 1) * STAGE;
      const char* nslot = smem + ((q + 1) & 1) * STAGE;
      if (first_tile) ktile(slot, nslot, std::true_type{}, std::integral_constant<int, 12>{}, std::false_type{});
      else ktile(slot, nslot, std::true_type{}, std::integral_constant<int, (UR_PERS_ABLATE == 3 || UR_PERS_ABLATE == 1 || UR_PERS_ABLATE == 2) ? 12 : 28>{}, std::false_type{});
      ++q;
      first_tile = false;
    }
    UR_PSTAMP(1);
    if constexpr (!K2S) {
#pragma unroll 1
      for (int kt = 1; kt < nkt; ++kt, ++q) {
        char* slot = smem + (q & 1) * STAGE;
        const char* nslot = smem + ((q + 1) & 1) * STAGE;
        if (kt == sw) { ubs = uniform_ptr(s_base(nn0)); ubr = uniform_ptr(r_base(nm0)); }
        else { ubs = uniform_ptr(ubs + BK * 2); ubr = uniform_ptr(ubr + BK * 2); }
        ktile(slot, nslot, std::false_type{}, std::integral_constant<int, 12>{}, std::false_type{});
#if UR_PERS_STAMPS
        if (kt <= 4) UR_PSTAMP(1 + kt);
#endif
      }
    } else {
      // (two different K tile bodies inside ONE loop make hipcc spill hundreds of registers: straight-line sequence instead)
#pragma unroll 1
      for (int kt = 1; kt < nk1 - 2; ++kt, ++q) {
        char* slot = smem + (q & 1) * STAGE;
        const char* nslot = smem + ((q + 1) & 1) * STAGE;
        ubs = uniform_ptr(ubs + BK * 2); ubr = uniform_ptr(ubr + BK * 2);
        ktile(slot, nslot, std::false_type{}, std::integral_constant<int, 12>{}, std::false_type{});
      }
      {
        char* slot = smem + (q & 1) * STAGE;
        const char* nslot = smem + ((q + 1) & 1) * STAGE;
        ubs = uniform_ptr(reinterpret_cast<const char*>(p.S2 + (long)(n0 + (REMAP ? (uwave & 3) * 8 + (uwave >> 2) * 64 : uwave * 8)) * p.lds2));
        ubr = uniform_ptr(reinterpret_cast<const char*>(p.R2 + (long)(m0 + uwave * 8) * p.ldr2));
        ktile(slot, nslot, std::false_type{}, std::integral_constant<int, 12>{}, std::true_type{});
        ++q;
      }
      ubs = uniform_ptr(s_base(nn0)); ubr = uniform_ptr(r_base(nm0));
#pragma unroll 1
      for (int kt = 0; kt < 2; ++kt, ++q) {
        char* slot = smem + (q & 1) * STAGE;
        const char* nslot = smem + ((q + 1) & 1) * STAGE;
        ktile(slot, nslot, std::false_type{}, std::integral_constant<int, 12>{}, std::false_type{});
        ubs = uniform_ptr(ubs + BK * 2); ubr = uniform_ptr(ubr + BK * 2);
      }
    }
    // the next tile's first K tile fetches ITS K tile 2
    ubs = uniform_ptr(s_base(nn0) + 2 * BK * 2); ubr = uniform_ptr(r_base(nm0) + 2 * BK * 2);
#if UR_PERS_ABLATE == 2
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(acc[i][j]));
    if (true) { tiles_left -= 1; if (tiles_left == 0) break; m0 = nm0; n0 = nn0; continue; }
#endif
    // The masked LoRA epilogue is a chain of load -> wait -> MFMA -> add: one barrier interval apart the two groups would run
    // it one after the other (each waiting at the other's barrier); re-joined they run it together.
    constexpr bool JOIN = UR_PERS_JOIN || DROP || EPI == 3;      // (EPI 3: its row sums cross the waves through LDS behind a workgroup barrier)
    if (JOIN && wr == 0) __builtin_amdgcn_s_barrier();
    UR_PSTAMP(6);
#if UR_PERS_EPI_WAIT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    // ================= epilogue, from the accumulators (no LDS, no barrier) =================
    __builtin_amdgcn_sched_barrier(0);      // nothing of the epilogue (its loads!) is scheduled up into the last K tile's phases
    // Lane constants of the epilogue are derived from an opaque copy of the lane id EVERY tile: hoisted out of the tile loop they
    // would stay live across the 256-register K loop (hipcc then spills inside it, and a scratch reload's vmcnt(0) drains the DMA ring)
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int el15 = elane & 15, eg4 = elane >> 4;
    // lane: rows m0 + rh*128 + wr*64 + jj*16 + el15; MFMA layout columns n0 + sh*128 + wc*32 + ii*16 + eg4*4 + 0..3
    if constexpr (DROP) {
      // C(m,n) += sum_a keep_a(m,n)/(1-p) * tb_a(m,:) . A_a(:,n): one rank-r MFMA per adapter and 16x16 sub-tile into a scratch
      // accumulator; keep flags from the adapters' dropped-flag bit planes (lora.hip: pair-interleaved byte order)
      // (rank 16 exactly: v_mfma_f32_16x16x16_bf16, lane holds k = 4 eg4 .. + 3 of its row -- half the fragment registers of the
      // zero-padded 16x16x32 form, which matters here: R0 of the next tile stays live across the epilogue)
      const int nad = p.K2 >> 4, kq = 4 * eg4;
      // every address = uniform base (scalar registers) + ONE 32-bit lane offset per tensor
      const uint32_t lo_s2 = (uint32_t)((el15 * p.lds2 + kq) * 2), lo_r2 = (uint32_t)((el15 * p.ldr2 + kq) * 2), lo_fl = (uint32_t)(el15 * p.drop_bits_ld);
      for (int a = 0; a < nad; ++a) {
        bf16x4 s2[4], r2[8];
        uint2 fl[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const char* ub = uniform_ptr(reinterpret_cast<const char*>(p.S2 + (long)(n0 + (REMAP ? wc * 64 + (i >> 1) * 32 : (i >> 1) * 128 + wc * 32) + (i & 1) * 16) * p.lds2 + a * 16));
          s2[i] = ld_g<bf16x4>(ub + lo_s2);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long mb = m0 + (j >> 2) * 128 + wr * 64 + (j & 3) * 16;
          const char* ub = uniform_ptr(reinterpret_cast<const char*>(p.R2 + mb * p.ldr2 + a * 16));
          r2[j] = ld_g<bf16x4>(ub + lo_r2);
          const char* fb = uniform_ptr(reinterpret_cast<const char*>(p.drop_bits + (long)a * p.drop_bits_stride + mb * p.drop_bits_ld + ((n0 + (REMAP ? wc * 64 : wc * 32)) >> 3)));
          fl[j] = make_uint2(ld_g<uint32_t>(fb + lo_fl), ld_g<uint32_t>(fb + lo_fl + (REMAP ? 4 : 16)));      // the flag words of the wave's two 32-column groups
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(s2[i], r2[j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const uint32_t wsel = (i >= 2) ? fl[j].y : fl[j].x;
            const uint32_t f = (wsel >> (16 * (i & 1) + 8 * (eg4 >> 1) + 2 * (eg4 & 1))) & 0x33u;
            f32x4& c = acc[i][j];
            if (!(f & 0x01u)) c[0] += d[0] * p.drop_inv_keep;
            if (!(f & 0x10u)) c[1] += d[1] * p.drop_inv_keep;
            if (!(f & 0x02u)) c[2] += d[2] * p.drop_inv_keep;
            if (!(f & 0x20u)) c[3] += d[3] * p.drop_inv_keep;
          }
      }
    }
    if constexpr (EPI == 4) {
      // ---- SwiGLU forward (modeling_qwen3.py:81-91): the tile's first 128 columns are gate features 128 bn .., the second 128 the
      // up projections of the SAME features (interleaved weight rows), so after the usual pack + 16-lane swap a lane holds gate
      // and up of the same 8 features: act = silu(gate) * up from the bf16-rounded values (bit-identical to swiglu_fwd_kernel);
      // gate | up leave in the standard [M, 2 I] order for the backward.
      const int cs = (eg4 & 1) * 16 + (eg4 >> 1) * 8;
      const int f0 = (n0 >> 1) + wc * 32 + cs;                               // first of this lane's 8 features
      bf16_t* Cb = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long m = m0 + (j >> 2) * 128 + wr * 64 + (j & 3) * 16 + el15;
        uint32_t gq[4], uq[4];
        {
          const f32x4 a = acc[0][j], b = acc[1][j];
          gq[0] = pack_bf2(a[0], a[1]); gq[1] = pack_bf2(a[2], a[3]); gq[2] = pack_bf2(b[0], b[1]); gq[3] = pack_bf2(b[2], b[3]);
          swap16(gq[0], gq[2]); swap16(gq[1], gq[3]);
        }
        {
          const f32x4 a = acc[2][j], b = acc[3][j];
          uq[0] = pack_bf2(a[0], a[1]); uq[1] = pack_bf2(a[2], a[3]); uq[2] = pack_bf2(b[0], b[1]); uq[3] = pack_bf2(b[2], b[3]);
          swap16(uq[0], uq[2]); swap16(uq[1], uq[3]);
        }
        // (swap16 pairs (x, y): afterwards the lane's 8 consecutive columns are {x regs, y regs} = words 0, 1, 2, 3 in order)
        uint32_t aq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) aq[e] = pack_bf2(silu_f(bf_lo(gq[e])) * bf_lo(uq[e]), silu_f(bf_hi(gq[e])) * bf_hi(uq[e]));
        const u32x4_t vg = {gq[0], gq[1], gq[2], gq[3]}, vu = {uq[0], uq[1], uq[2], uq[3]}, va = {aq[0], aq[1], aq[2], aq[3]};
        st_g<u32x4_t>(Cb + m * p.ldc + f0, vg);
        st_g<u32x4_t>(Cb + m * p.ldc + p.sp_I + f0, vu);
        st_g<u32x4_t>(p.sp_act + m * p.sp_ldact + f0, va);
      }
    } else
    if constexpr (EPI == 3) {
      // ---- q/k-norm + RoPE (modeling_qwen3.py:59-64,107-137,227-245) from the accumulators.  The tile is two whole heads (its two
      // 128-column halves sh); the head's rows of W were stored in the paired order, so acc[2 sh][j][e] is feature
      // d = wc*16 + 4*g4 + e and acc[2 sh + 1][j][e] its rotate-half partner d + 64: RoPE is register-local.  Only the row sums
      // of squares cross lanes (the 4 g4 lanes of a row: two shuffles) and waves (the 4 wc waves of a head: 8 KiB of LDS behind
      // one workgroup barrier; both wave groups are joined here).
      const int qkc = p.qk_nq + p.qk_nk;
      if (n0 < qkc) {           // uniform: a q / k tile
        float* part = reinterpret_cast<float*>(smem + 2 * STAGE);          // [2 sh][256 rows][4 wc]
        const bool isq = n0 < p.qk_nq;
        const float* wgt = isq ? p.qk_qw : p.qk_kw;
        const int dq0 = wc * 16 + 4 * eg4;                                  // this lane's features dq0 .. + 3 (and + 64)
        const int pos0 = m0 % p.qk_S;                                       // (scalar) S % 256 == 0: the tile's 256 rows are positions pos0 .. pos0 + 255
        // Every global load of the epilogue is issued HERE, ahead of the row-sum barrier that hides its latency: a load between
        // the stores below would make hipcc wait vmcnt(0) -- i.e. for the previous rows' stores as well -- once per row block
        // (measured: ~6 us per tile).  cos / sin of the lane's 8 rows come from ONE table row (the first: position
        // pos0 + wr*64 + l15) advanced by the angle-addition theorem: + 16 positions per row block, + 80 across the two row halves
        // (table rows 16 and 80 are cos / sin of exactly those steps).
        const int prow0 = pos0 + wr * 64 + el15;
        const float4 w0 = ld_g<float4>(wgt + dq0), w1 = ld_g<float4>(wgt + 64 + dq0);
        const float4 c_0 = ld_g<float4>(p.qk_cos + (long)prow0 * 64 + dq0), s_0 = ld_g<float4>(p.qk_sin + (long)prow0 * 64 + dq0);
        const float4 c16 = ld_g<float4>(p.qk_cos + 16 * 64 + dq0), s16 = ld_g<float4>(p.qk_sin + 16 * 64 + dq0);
        const float4 c80 = ld_g<float4>(p.qk_cos + 80 * 64 + dq0), s80 = ld_g<float4>(p.qk_sin + 80 * 64 + dq0);
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const f32x4 a = acc[2 * sh][j], b = acc[2 * sh + 1][j];
            float ss = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const int row = (j >> 2) * 128 + wr * 64 + (j & 3) * 16 + el15;
            if (eg4 == 0) part[(sh * 256 + row) * 4 + wc] = ss;
          }
        __syncthreads();
        bf16_t* outb = isq ? p.qk_q : p.qk_k;
        const long ldo = isq ? p.qk_ldq : p.qk_ldk;
        const int c0 = isq ? n0 : n0 - p.qk_nq;                             // first column of the tile inside q_out / k_out
        const int nheads = qkc >> 7, head0 = n0 >> 7;
        // after the 16-lane swap: 16-lane row rho even -> the 8 features (rho >> 1) * 8 .. of block d, odd -> of block d + 64
        const int ocol = (eg4 & 1) * 64 + wc * 16 + (eg4 >> 1) * 8;
        float cv[4] = {c_0.x, c_0.y, c_0.z, c_0.w}, sv[4] = {s_0.x, s_0.y, s_0.z, s_0.w};
        const float wa[4] = {w0.x, w0.y, w0.z, w0.w}, wb[4] = {w1.x, w1.y, w1.z, w1.w};
        const float dc16[4] = {c16.x, c16.y, c16.z, c16.w}, ds16[4] = {s16.x, s16.y, s16.z, s16.w};
        const float dc80[4] = {c80.x, c80.y, c80.z, c80.w}, ds80[4] = {s80.x, s80.y, s80.z, s80.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int row = (j >> 2) * 128 + wr * 64 + (j & 3) * 16 + el15;
#pragma unroll
          for (int sh = 0; sh < 2; ++sh) {
            const float4 pr = *reinterpret_cast<const float4*>(part + (sh * 256 + row) * 4);
            const float rs = rsqrtf((pr.x + pr.y + pr.z + pr.w) * (1.0f / 128.0f) + p.qk_eps);
            const f32x4 a = acc[2 * sh][j], b = acc[2 * sh + 1][j];
            float o0[4], o1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xa = a[e] * rs * wa[e], xb = b[e] * rs * wb[e];
              o0[e] = xa * cv[e] - xb * sv[e];
              o1[e] = xb * cv[e] + xa * sv[e];
            }
            uint32_t a0 = pack_bf2(o0[0], o0[1]), a1 = pack_bf2(o0[2], o0[3]), b0 = pack_bf2(o1[0], o1[1]), b1 = pack_bf2(o1[2], o1[3]);
            swap16(a0, b0); swap16(a1, b1);
            const long m = m0 + row;
            const u32x4_t v = {a0, a1, b0, b1};
            st_g<u32x4_t>(outb + m * ldo + c0 + sh * 128 + ocol, v);
            if (wc == 0 && eg4 == 0) p.qk_rstd[m * nheads + head0 + sh] = rs;
          }
          // the next row block's angles: + 16 positions, or + 80 from the last block of the first row half to the first of the second
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dc = (j == 3) ? dc80[e] : dc16[e], dsn = (j == 3) ? ds80[e] : ds16[e];
            const float cn = cv[e] * dc - sv[e] * dsn, sn2 = sv[e] * dc + cv[e] * dsn;
            cv[e] = cn; sv[e] = sn2;
          }
        }
      } else {                  // a v tile: the plain projection into v_out
        const int cs = (eg4 & 1) * 16 + (eg4 >> 1) * 8;
        const uint32_t loff = (uint32_t)((el15 * p.qk_ldv + cs) * 2);
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const f32x4 a = acc[2 * sh][j], b = acc[2 * sh + 1][j];
            uint32_t a0 = pack_bf2(a[0], a[1]), a1 = pack_bf2(a[2], a[3]), b0 = pack_bf2(b[0], b[1]), b1 = pack_bf2(b[2], b[3]);
            swap16(a0, b0); swap16(a1, b1);
            char* base = uniform_wptr(reinterpret_cast<char*>(p.qk_v) +
                                      ((long)(m0 + (j >> 2) * 128 + wr * 64 + (j & 3) * 16) * p.qk_ldv + (n0 - qkc) + sh * 128 + wc * 32) * 2);
            const u32x4_t v = {a0, a1, b0, b1};
            st_g<u32x4_t>(base + loff, v);
          }
      }
    } else
    {
      // after the 16-lane swap a lane holds 8 consecutive columns of row m: cs = (eg4 & 1) * 16 + (eg4 >> 1) * 8 within the wave's 32
      const int cs = (eg4 & 1) * 16 + (eg4 >> 1) * 8;
      const float alpha = p.alpha;
      // lanes l15 < 8 keep their own group-0 piece in store 1 and take the group-1 piece of the lane 8 rows further into store 2; lanes >= 8 the other way round
      [[maybe_unused]] const int r7 = el15 & 7;
      [[maybe_unused]] const bool lo8 = el15 < 8;
      [[maybe_unused]] auto line_offs = [&](long ld, uint32_t& o1, uint32_t& o2) {
        o1 = (uint32_t)((r7 * ld + (lo8 ? 0 : 32) + cs) * 2); o2 = (uint32_t)(((8 + r7) * ld + (lo8 ? 32 : 0) + cs) * 2);
      };
      [[maybe_unused]] auto merge_lines = [&](const uint32_t (&xa)[4], const uint32_t (&xb)[4], u32x4_t& z1, u32x4_t& z2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          z1[e] = (uint32_t)__builtin_amdgcn_update_dpp((int)xa[e], (int)xb[e], 0x128, 0xf, 0xc, false);     // row_ror:8 into lanes 8..15 of each 16-lane row
          z2[e] = (uint32_t)__builtin_amdgcn_update_dpp((int)xa[e], (int)xb[e], 0x128, 0xf, 0x3, false);     // ... into lanes 0..7
        }
      };
      if constexpr (EPI == 0 && REMAP) {
        // whole-line stores: XA = the lane's 8 columns of group sh 0, XB = of group sh 1 (32 columns further); one DPP row rotation by 8
        // lanes hands XB to the lane 8 rows away, merged under a bank mask: store 1 = rows 0..7 x 128 bytes, store 2 = rows 8..15
        uint32_t loff1, loff2;
        line_offs(p.ldc, loff1, loff2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          uint32_t xa[4], xb[4];
          {
            const f32x4 a = acc[0][j], b = acc[1][j];
            xa[0] = pack_bf2(a[0] * alpha, a[1] * alpha); xa[1] = pack_bf2(a[2] * alpha, a[3] * alpha);
            xa[2] = pack_bf2(b[0] * alpha, b[1] * alpha); xa[3] = pack_bf2(b[2] * alpha, b[3] * alpha);
            swap16(xa[0], xa[2]); swap16(xa[1], xa[3]);
          }
          {
            const f32x4 a = acc[2][j], b = acc[3][j];
            xb[0] = pack_bf2(a[0] * alpha, a[1] * alpha); xb[1] = pack_bf2(a[2] * alpha, a[3] * alpha);
            xb[2] = pack_bf2(b[0] * alpha, b[1] * alpha); xb[3] = pack_bf2(b[2] * alpha, b[3] * alpha);
            swap16(xb[0], xb[2]); swap16(xb[1], xb[3]);
          }
          u32x4_t z1, z2;
          merge_lines(xa, xb, z1, z2);
          char* base = uniform_wptr(reinterpret_cast<char*>(p.C) + ((long)(m0 + (j >> 2) * 128 + wr * 64 + (j & 3) * 16) * p.ldc + n0 + wc * 64) * 2);
          __builtin_nontemporal_store(z1, (__attribute__((address_space(1))) u32x4_t*)(base + loff1));
          __builtin_nontemporal_store(z2, (__attribute__((address_space(1))) u32x4_t*)(base + loff2));
        }
      } else
      if constexpr (EPI == 0) {
        const uint32_t loff = (uint32_t)((el15 * p.ldc + cs) * 2);
#pragma unroll
        for (int sh = 0; sh < 2; ++sh)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const f32x4 a = acc[2 * sh][j], b = acc[2 * sh + 1][j];
            uint32_t a0 = pack_bf2(a[0] * alpha, a[1] * alpha), a1 = pack_bf2(a[2] * alpha, a[3] * alpha);
            uint32_t b0 = pack_bf2(b[0] * alpha, b[1] * alpha), b1 = pack_bf2(b[2] * alpha, b[3] * alpha);
            swap16(a0, b0); swap16(a1, b1);
            char* base = uniform_wptr(reinterpret_cast<char*>(p.C) +
                                      ((long)(m0 + (j >> 2) * 128 + wr * 64 + (j & 3) * 16) * p.ldc + n0 + sh * 128 + wc * 32) * 2);
            const u32x4_t v = {a0, a1, b0, b1};
#if UR_PERS_ABLATE == 1
            asm volatile("" :: "v"(v), "v"(base + loff));
#elif UR_PERS_ABLATE == 4
            st_g<u32x4_t>(base + loff, v);
#else
            __builtin_nontemporal_store(v, (__attribute__((address_space(1))) u32x4_t*)(base + loff));
#endif
          }
      } else if constexpr ((EPI == 2 || EPI == 5 || EPI == 6 || EPI == 7) && REMAP) {
        // bias / residual / GELU with whole-line stores.  Bias and residual are added in the MFMA layout (8-byte residual pieces, as below);
        // quarter q = (row half q >> 1, row blocks 2 (q & 1) .. + 1) x BOTH column groups, so that every row block has its two pieces
        // at hand for the line merge; the residual pieces of quarter q + 1 are issued before quarter q's stores.
        constexpr bool RES = EPI == 2 || EPI == 7, GRAD = EPI == 7, GOUT = EPI == 6;
        const int nq4 = eg4 * 4;
        const bf16_t* const rsrc = GRAD ? p.aux : p.res;
        const long ldrs = GRAD ? p.ldaux : p.ldres;
        uint32_t loc1, loc2, log1 = 0, log2 = 0;
        line_offs(p.ldc, loc1, loc2);
        if constexpr (GOUT) line_offs(p.ldg, log1, log2);
        const uint32_t loff_r = (uint32_t)((el15 * ldrs + nq4) * 2);
        uint2 ra[2][2][2], rb[2][2][2];          // [buffer][row block of the quarter][column group]
        float bsa[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, bsb[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (p.bias) {
#pragma unroll
          for (int sh = 0; sh < 2; ++sh) {
            const float4 b0 = ld_g<float4>(p.bias + n0 + wc * 64 + sh * 32 + nq4), b1 = ld_g<float4>(p.bias + n0 + wc * 64 + sh * 32 + 16 + nq4);
            bsa[sh][0] = b0.x; bsa[sh][1] = b0.y; bsa[sh][2] = b0.z; bsa[sh][3] = b0.w; bsb[sh][0] = b1.x; bsb[sh][1] = b1.y; bsb[sh][2] = b1.z; bsb[sh][3] = b1.w;
          }
        }
        auto load_quarter = [&](int q) {
          if constexpr (RES) {
            const int rh = q >> 1;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int sh = 0; sh < 2; ++sh) {
                const char* rbase = uniform_ptr(reinterpret_cast<const char*>(rsrc + (long)(m0 + rh * 128 + wr * 64 + (2 * (q & 1) + t) * 16) * ldrs + n0 + wc * 64 + sh * 32));
                ra[q & 1][t][sh] = UR_LD_STREAM<uint2>(rbase + loff_r);
                rb[q & 1][t][sh] = UR_LD_STREAM<uint2>(rbase + loff_r + 32);
              }
          }
        };
        load_quarter(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q + 1 < 4) load_quarter(q + 1);
          const int rh = q >> 1;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int jj = 2 * (q & 1) + t, j = 4 * rh + jj;
            uint32_t x[2][4];
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
              const f32x4 a = acc[2 * sh][j], b = acc[2 * sh + 1][j];
              float va[4], vb[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) { va[e] = a[e] * alpha + bsa[sh][e]; vb[e] = b[e] * alpha + bsb[sh][e]; }
              if constexpr (RES && !GRAD) {
                const uint2 xa = ra[q & 1][t][sh], xb = rb[q & 1][t][sh];
                va[0] += bf_lo(xa.x); va[1] += bf_hi(xa.x); va[2] += bf_lo(xa.y); va[3] += bf_hi(xa.y);
                vb[0] += bf_lo(xb.x); vb[1] += bf_hi(xb.x); vb[2] += bf_lo(xb.y); vb[3] += bf_hi(xb.y);
              }
              if constexpr (GRAD) {
                const uint2 xa = ra[q & 1][t][sh], xb = rb[q & 1][t][sh];
                va[0] *= gelu_erf_grad_f(bf_lo(xa.x)); va[1] *= gelu_erf_grad_f(bf_hi(xa.x)); va[2] *= gelu_erf_grad_f(bf_lo(xa.y)); va[3] *= gelu_erf_grad_f(bf_hi(xa.y));
                vb[0] *= gelu_erf_grad_f(bf_lo(xb.x)); vb[1] *= gelu_erf_grad_f(bf_hi(xb.x)); vb[2] *= gelu_erf_grad_f(bf_lo(xb.y)); vb[3] *= gelu_erf_grad_f(bf_hi(xb.y));
              }
              x[sh][0] = pack_bf2(va[0], va[1]); x[sh][1] = pack_bf2(va[2], va[3]); x[sh][2] = pack_bf2(vb[0], vb[1]); x[sh][3] = pack_bf2(vb[2], vb[3]);
              swap16(x[sh][0], x[sh][2]); swap16(x[sh][1], x[sh][3]);
            }
            u32x4_t z1, z2;
            merge_lines(x[0], x[1], z1, z2);
            char* cb = uniform_wptr(reinterpret_cast<char*>(p.C) + ((long)(m0 + rh * 128 + wr * 64 + jj * 16) * p.ldc + n0 + wc * 64) * 2);
            st_g<u32x4_t>(cb + loc1, z1);
            st_g<u32x4_t>(cb + loc2, z2);
            if constexpr (GOUT) {
              char* gb = uniform_wptr(reinterpret_cast<char*>(p.gelu_out) + ((long)(m0 + rh * 128 + wr * 64 + jj * 16) * p.ldg + n0 + wc * 64) * 2);
              u32x4_t g1, g2;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                g1[e] = pack_bf2(gelu_erf_f(bf_lo(z1[e])), gelu_erf_f(bf_hi(z1[e])));
                g2[e] = pack_bf2(gelu_erf_f(bf_lo(z2[e])), gelu_erf_f(bf_hi(z2[e])));
              }
              st_g<u32x4_t>(gb + log1, g1);
              st_g<u32x4_t>(gb + log2, g2);
            }
          }
        }
      } else if constexpr (EPI == 2 || EPI == 5 || EPI == 6 || EPI == 7) {
        // EPI 5: bias only (a RUNTIME residual switch made hipcc spill 16-25 registers); EPI 6: bias, then GELU of the bf16-ROUNDED
        // pre-activation into a second output (gemm.hip's rich epilogue: the backward's gelu'(u) sees the same u); EPI 7: the second
        // operand is that saved pre-activation and gelu'(u) is multiplied in instead of a residual added
        constexpr bool RES = EPI == 2 || EPI == 7, GRAD = EPI == 7, GOUT = EPI == 6;
        // bias / residual in the MFMA layout (lane: row m, 4 consecutive columns of each 16x16 sub-tile: 8-byte residual pieces),
        // then the plain path's pack + 16-lane swap + 16-byte store.  (Adding after an f32 swap -- 16-byte residual pieces --
        // costs 8 more live registers per row block; hipcc spilled 16-25 registers around it, some inside the K tiles.)
        const int nq4 = eg4 * 4;
        const bf16_t* const rsrc = GRAD ? p.aux : p.res;
        const long ldrs = GRAD ? p.ldaux : p.ldres;
        const uint32_t loff_c = (uint32_t)((el15 * p.ldc + cs) * 2), loff_r = (uint32_t)((el15 * ldrs + nq4) * 2);
        const uint32_t loff_g = GOUT ? (uint32_t)((el15 * p.ldg + cs) * 2) : 0u;
        // Software pipeline over the four quarter tiles q = 2 sh + rh (see the SwiGLU-backward branch below): the residual
        // pieces (and, per column half, the bias) of quarter q + 1 are issued BEFORE quarter q's stores -- a load issued after
        // stores cannot be waited for without their acknowledgements (vmcnt counts both, in order).
        uint2 ra[2][4], rb[2][4];
        float bsa[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, bsb[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        auto load_quarter = [&](int q) {
          const int sh = q >> 1, rh = q & 1;
          if ((q & 1) == 0 && p.bias) {
            const float4 b0 = ld_g<float4>(p.bias + n0 + sh * 128 + wc * 32 + nq4), b1 = ld_g<float4>(p.bias + n0 + sh * 128 + wc * 32 + 16 + nq4);
            bsa[sh][0] = b0.x; bsa[sh][1] = b0.y; bsa[sh][2] = b0.z; bsa[sh][3] = b0.w; bsb[sh][0] = b1.x; bsb[sh][1] = b1.y; bsb[sh][2] = b1.z; bsb[sh][3] = b1.w;
          }
          if constexpr (RES) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const char* rbase = uniform_ptr(reinterpret_cast<const char*>(rsrc + (long)(m0 + rh * 128 + wr * 64 + jj * 16) * ldrs + n0 + sh * 128 + wc * 32));
              ra[q & 1][jj] = ld_g<uint2>(rbase + loff_r);
              rb[q & 1][jj] = ld_g<uint2>(rbase + loff_r + 32);
            }
          }
        };
        load_quarter(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q + 1 < 4) load_quarter(q + 1);
          const int sh = q >> 1, rh = q & 1;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * rh + jj;
            const f32x4 a = acc[2 * sh][j], b = acc[2 * sh + 1][j];
            float va[4], vb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = a[e] * alpha + bsa[sh][e]; vb[e] = b[e] * alpha + bsb[sh][e]; }
            if constexpr (RES && !GRAD) {
              const uint2 xa = ra[q & 1][jj], xb = rb[q & 1][jj];
              va[0] += bf_lo(xa.x); va[1] += bf_hi(xa.x); va[2] += bf_lo(xa.y); va[3] += bf_hi(xa.y);
              vb[0] += bf_lo(xb.x); vb[1] += bf_hi(xb.x); vb[2] += bf_lo(xb.y); vb[3] += bf_hi(xb.y);
            }
            if constexpr (GRAD) {
              const uint2 xa = ra[q & 1][jj], xb = rb[q & 1][jj];
              va[0] *= gelu_erf_grad_f(bf_lo(xa.x)); va[1] *= gelu_erf_grad_f(bf_hi(xa.x)); va[2] *= gelu_erf_grad_f(bf_lo(xa.y)); va[3] *= gelu_erf_grad_f(bf_hi(xa.y));
              vb[0] *= gelu_erf_grad_f(bf_lo(xb.x)); vb[1] *= gelu_erf_grad_f(bf_hi(xb.x)); vb[2] *= gelu_erf_grad_f(bf_lo(xb.y)); vb[3] *= gelu_erf_grad_f(bf_hi(xb.y));
            }
            uint32_t a0 = pack_bf2(va[0], va[1]), a1 = pack_bf2(va[2], va[3]), b0 = pack_bf2(vb[0], vb[1]), b1 = pack_bf2(vb[2], vb[3]);
            swap16(a0, b0); swap16(a1, b1);
            char* cb = uniform_wptr(reinterpret_cast<char*>(p.C) + ((long)(m0 + rh * 128 + wr * 64 + jj * 16) * p.ldc + n0 + sh * 128 + wc * 32) * 2);
            const u32x4_t o = {a0, a1, b0, b1};
            st_g<u32x4_t>(cb + loff_c, o);
            if constexpr (GOUT) {
              char* gb = uniform_wptr(reinterpret_cast<char*>(p.gelu_out) + ((long)(m0 + rh * 128 + wr * 64 + jj * 16) * p.ldg + n0 + sh * 128 + wc * 32) * 2);
              const u32x4_t og = {pack_bf2(gelu_erf_f(bf_lo(a0)), gelu_erf_f(bf_hi(a0))), pack_bf2(gelu_erf_f(bf_lo(a1)), gelu_erf_f(bf_hi(a1))),
                                  pack_bf2(gelu_erf_f(bf_lo(b0)), gelu_erf_f(bf_hi(b0))), pack_bf2(gelu_erf_f(bf_lo(b1)), gelu_erf_f(bf_hi(b1)))};
              st_g<u32x4_t>(gb + loff_g, og);
            }
          }
        }
      } else if constexpr (EPI == 1 && REMAP) {
        // ---- SwiGLU backward with whole-line loads and stores: the f32 products change lanes first (the line merge on 8 registers per
        // column group), then gate / up arrive and dgate / dup leave in the merged layout: 8 rows x 128 contiguous bytes per
        // instruction.  Same software pipeline as below: quarter q = (row half, two row blocks) x both column groups.
        __builtin_amdgcn_sched_barrier(0);
        int em0 = __builtin_amdgcn_readfirstlane(m0), en0 = __builtin_amdgcn_readfirstlane(n0);
        asm volatile("" : "+s"(em0), "+s"(en0));
        // three buffers of one row block each (its gate / up pieces at the two store positions: 16 registers), two row blocks ahead:
        // the same 8 loads per wave in flight as two quarter-tile buffers, in 48 registers instead of 64 (the merged f32 products need 16)
        uint4 gw[3][2], uw[3][2];
        uint32_t lg1, lg2, ld1, ld2;
        line_offs(p.sw_ldgu, lg1, lg2);
        line_offs(p.sw_lddgu, ld1, ld2);
        const long up_g = (long)p.sw_I * 2;
        auto load_block = [&](int j, uint4 (&g)[2], uint4 (&u)[2]) {
          const char* gb = uniform_ptr(reinterpret_cast<const char*>(p.sw_gu + (long)(em0 + (j >> 2) * 128 + wr * 64 + (j & 3) * 16) * p.sw_ldgu + en0 + wc * 64));
          g[0] = UR_LD_STREAM<uint4>(gb + lg1); g[1] = UR_LD_STREAM<uint4>(gb + lg2);
          u[0] = UR_LD_STREAM<uint4>(gb + up_g + lg1); u[1] = UR_LD_STREAM<uint4>(gb + up_g + lg2);
        };
        load_block(0, gw[0], uw[0]);
        load_block(1, gw[1], uw[1]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (j + 2 < 8) load_block(j + 2, gw[(j + 2) % 3], uw[(j + 2) % 3]);
          char* db = uniform_wptr(reinterpret_cast<char*>(p.sw_dgu + (long)(em0 + (j >> 2) * 128 + wr * 64 + (j & 3) * 16) * p.sw_lddgu + en0 + wc * 64));
          float x[2][8];
#pragma unroll
          for (int sh = 0; sh < 2; ++sh) {
            f32x4 a = acc[2 * sh][j], b = acc[2 * sh + 1][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) { float xx = a[e], yy = b[e]; swap16f(xx, yy); a[e] = xx; b[e] = yy; }
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[sh][e] = a[e] * alpha; x[sh][4 + e] = b[e] * alpha; }
          }
#pragma unroll
          for (int pos = 0; pos < 2; ++pos) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
              v[e] = pos ? __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x[0][e]), __float_as_int(x[1][e]), 0x128, 0xf, 0x3, false))
                         : __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x[0][e]), __float_as_int(x[1][e]), 0x128, 0xf, 0xc, false));
            // d(act) = v (f32, unrounded): dgate = v u silu'(g), dup = v silu(g)   (elementwise.hip: swiglu_bwd_kernel)
            const uint4 gq4 = gw[j % 3][pos], uq4 = uw[j % 3][pos];
            const uint32_t gq[4] = {gq4.x, gq4.y, gq4.z, gq4.w}, uq[4] = {uq4.x, uq4.y, uq4.z, uq4.w};
            uint32_t og[4], ou[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float dgv[2], duv[2];
#pragma unroll
              for (int hh = 0; hh < 2; ++hh) {
                const float gg = hh ? bf_hi(gq[e]) : bf_lo(gq[e]), uu = hh ? bf_hi(uq[e]) : bf_lo(uq[e]);
                const float d = v[2 * e + hh];
                const float sg = sigmoid_f(gg);
                duv[hh] = d * (gg * sg);
                dgv[hh] = d * uu * (sg * (1.0f + gg * (1.0f - sg)));
              }
              og[e] = pack_bf2(dgv[0], dgv[1]); ou[e] = pack_bf2(duv[0], duv[1]);
            }
            const u32x4_t vg = {og[0], og[1], og[2], og[3]}, vu = {ou[0], ou[1], ou[2], ou[3]};
            st_g<u32x4_t>(db + (pos ? ld2 : ld1), vg);
            st_g<u32x4_t>(db + up_g + (pos ? ld2 : ld1), vu);
          }
        }
      } else {
        // ---- SwiGLU backward (EPI 1; no bias / residual: gemm_pers_eligible).  vmcnt counts loads and stores in issue order, so a
        // load issued AFTER a batch of stores cannot be waited for without those stores' acknowledgements: four quarter-tile
        // batches of {8 loads, wait, math, 8 stores} exposed one load latency + one store round trip each (~28 us per tile
        // beside a 23 us K loop at K = 1024).  Software pipeline instead: quarter q + 1's gate / up pieces are issued BEFORE
        // quarter q's stores, into the second of two 32-register buffers; the wait for quarter q's pieces then leaves the 8
        // stores of q - 1 and the 8 loads of q + 1 in flight (hipcc counts them: vmcnt(16)).
        static_assert(EPI == 1 && !REMAP, "the remaining epilogue is the SwiGLU backward in the 128-apart layout");
        __builtin_amdgcn_sched_barrier(0);       // (behind the masked LoRA epilogue: its uniform bases are dead before these are made)
        int em0 = __builtin_amdgcn_readfirstlane(m0), en0 = __builtin_amdgcn_readfirstlane(n0);
        asm volatile("" : "+s"(em0), "+s"(en0));
        uint4 gw[2][4], uw[2][4];
        // every address = uniform base (scalar registers) + ONE 32-bit lane offset per tensor
        const uint32_t loff_g = (uint32_t)((el15 * p.sw_ldgu + cs) * 2), loff_d = (uint32_t)((el15 * p.sw_lddgu + cs) * 2);
        const long up_g = (long)p.sw_I * 2;                                  // bytes from a row's gate half to its up half
        auto load_quarter = [&](int q, uint4 (&g)[4], uint4 (&u)[4]) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const char* gb = uniform_ptr(reinterpret_cast<const char*>(p.sw_gu + (long)(em0 + (q & 1) * 128 + wr * 64 + jj * 16) * p.sw_ldgu + en0 + (q >> 1) * 128 + wc * 32));
            g[jj] = ld_g<uint4>(gb + loff_g);
            u[jj] = ld_g<uint4>(gb + up_g + loff_g);
          }
        };
        load_quarter(0, gw[0], uw[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q + 1 < 4) load_quarter(q + 1, gw[(q + 1) & 1], uw[(q + 1) & 1]);
          const int sh = q >> 1, rh = q & 1;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * rh + jj;
            char* db = uniform_wptr(reinterpret_cast<char*>(p.sw_dgu + (long)(em0 + rh * 128 + wr * 64 + jj * 16) * p.sw_lddgu + en0 + sh * 128 + wc * 32));
            f32x4 a = acc[2 * sh][j], b = acc[2 * sh + 1][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) { float x = a[e], y = b[e]; swap16f(x, y); a[e] = x; b[e] = y; }
            const float v[8] = {a[0] * alpha, a[1] * alpha, a[2] * alpha, a[3] * alpha, b[0] * alpha, b[1] * alpha, b[2] * alpha, b[3] * alpha};
            // d(act) = v (f32, unrounded): dgate = v u silu'(g), dup = v silu(g)   (elementwise.hip: swiglu_bwd_kernel)
            const uint4 gq4 = gw[q & 1][jj], uq4 = uw[q & 1][jj];
            const uint32_t gq[4] = {gq4.x, gq4.y, gq4.z, gq4.w}, uq[4] = {uq4.x, uq4.y, uq4.z, uq4.w};
            uint32_t og[4], ou[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float dgv[2], duv[2];
#pragma unroll
              for (int hh = 0; hh < 2; ++hh) {
                const float gg = hh ? bf_hi(gq[e]) : bf_lo(gq[e]), uu = hh ? bf_hi(uq[e]) : bf_lo(uq[e]);
                const float d = v[2 * e + hh];
                const float sg = sigmoid_f(gg);
                duv[hh] = d * (gg * sg);
                dgv[hh] = d * uu * (sg * (1.0f + gg * (1.0f - sg)));
              }
              og[e] = pack_bf2(dgv[0], dgv[1]); ou[e] = pack_bf2(duv[0], duv[1]);
            }
            const u32x4_t vg = {og[0], og[1], og[2], og[3]}, vu = {ou[0], ou[1], ou[2], ou[3]};
            st_g<u32x4_t>(db + loff_d, vg);
            st_g<u32x4_t>(db + up_g + loff_d, vu);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    UR_PSTAMP(7);
    if (JOIN && wr == 1 && tiles_left > 1) __builtin_amdgcn_s_barrier();
    tiles_left -= 1;
    if (tiles_left == 0) break;
    m0 = nm0; n0 = nn0;
  }
  // drain: re-join the wave groups, let the never-consumed tail of the stream land before the workgroup's LDS is released
  if (!(UR_PERS_JOIN || DROP || EPI == 3) && wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI, int MODE>
int launch_pers(const GemmP& p, hipStream_t st) {
  static std::atomic<uint64_t> attr_set{0};   // per device
  constexpr int SMEM = 2 * STAGE + (EPI == 3 ? 2 * 256 * 4 * (int)sizeof(float) : 0);
  UR_ONCE_PER_DEVICE(attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pers_kernel<EPI, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) UR_FAIL((int)e, "ur_gemm(persistent): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  }
  int ncu = ur_device_cu_count();
  ncu -= ncu % 8;                  // ids equal mod 8 share an XCD: the stride of the tile walk must keep that
  if (ncu <= 0) ncu = 8;
  const int ntiles = p.gm * p.gn;
  static const int env_grid = ur_lab_int("UR_PERS_GRID", 0);      // lab: fewer workgroups (a multiple of 8)
  const int cap = (env_grid >= 8 && env_grid < ncu) ? env_grid - env_grid % 8 : ncu;
  const int grid = ntiles < cap ? ntiles : cap;
  auto magic = [](int d) { return (uint32_t)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
  TileOrder o;
  o.nwg = ntiles; o.gn = p.gn; o.gcw = p.gcw;
  o.rows_x = (ntiles >> 3) / p.gn; o.per = o.rows_x * (p.gcw > 0 ? p.gcw : 1);
  o.m_gn = magic(p.gn); o.m_per = magic(o.per > 0 ? o.per : 1); o.m_gcw = magic(p.gcw > 0 ? p.gcw : 1);
  hipLaunchKernelGGL((gemm_pers_kernel<EPI, MODE>), dim3(grid), dim3(512), SMEM, st, p, o, ntiles);
  UR_CHECK_LAUNCH("ur_gemm(persistent)");
  return 0;
}

}  // namespace

namespace urgemm {

static std::atomic<int> g_pers_mode{-1};     // -1 = not set: the default (1); 0 = generic kernel only (A/B runs, bit-identity tests)

bool gemm_pers_eligible(const GemmP& p, int splits, bool rk, bool sk, bool outf32) {
  const int set = g_pers_mode.load(std::memory_order_relaxed);
  const int mode = set >= 0 ? set : 1;
  if (!mode || !rk || !sk || outf32 || splits > 1) return false;
  if ((p.M % BM) || (p.N % BN) || (p.K % BK) || p.K < 4 * BK) return false;
  static const int min_tiles = ur_lab_int("UR_PERS_MIN_TILES", 128);      // lab; default 128: half a round already gains from the register epilogue (C2 item stage 21.66 -> 21.05 ms; 256 and 512 equal within noise, user stage unchanged)
  if ((long)(p.M / BM) * (p.N / BN) < min_tiles) return false;
  if ((long)(p.M / BM) * (p.N / BN) >= (1L << 20) || p.N / BN >= (1 << 12) || p.M / BM >= (1 << 15)) return false;      // fdiv: n * d < 2^32
  if (p.sw_mode == 2) return false;
  if (p.gelu_out || p.aux) {                 // the GELU epilogues (EPI 6 / 7): alone on the first K range, never with a residual
    if ((p.gelu_out && p.aux) || p.res || p.K2 > 0 || p.sw_mode || p.qk_q || p.sp_act) return false;
    if (p.gelu_out && ((p.ldg & 7) || (reinterpret_cast<uintptr_t>(p.gelu_out) & 15))) return false;
    if (p.aux && ((p.ldaux & 7) || (reinterpret_cast<uintptr_t>(p.aux) & 15))) return false;
  }
  if (p.K2 > 0 && !p.drop_bits && p.K2 > BK) return false;
  if (p.drop_bits && p.K2 > 0) {
    if (p.drop_rank != 16 || (p.K2 & 15)) return false;                        // the masked epilogue here is rank 16 only
    // measured (tools/lab/gemm_pers_ab.py, C4 shapes): with the masked LoRA epilogue the persistent kernel wins at K = 1024 (+5 %, +17 %
    // with the SwiGLU backward epilogue) and was even at K = 4096 / lost 1.5 % at K = 6144 in round 3 (profiles/r3_gemm_pers_ab.txt: 8 long
    // tiles per CU, nothing to hide, and the joined epilogue costs two barrier intervals); with the whole-line epilogue stores of round 5
    // it wins there too (K = 4096 +8 %, K = 6144 +2-3 %, same box), so the limit only keeps untested lengths off it
    static const int kmax = ur_lab_int("UR_PERS_DROP_KMAX", 8192);
    if (p.K > kmax) return false;
  }
  if (p.sw_mode == 1 && (p.bias || p.res)) return false;
  // 16-byte pieces everywhere
  if ((p.ldc & 7) || (reinterpret_cast<uintptr_t>(p.C) & 15)) return false;
  if (p.res && ((p.ldres & 7) || (reinterpret_cast<uintptr_t>(p.res) & 15))) return false;
  if (p.sw_mode == 1 && ((p.sw_ldgu & 7) || (p.sw_lddgu & 7) || (p.sw_I & 7) || (reinterpret_cast<uintptr_t>(p.sw_gu) & 15) ||
                         (reinterpret_cast<uintptr_t>(p.sw_dgu) & 15))) return false;
  // 32-bit lane offsets
  if (p.lds * 16 >= (1L << 31) || p.ldr * 16 >= (1L << 31) || p.ldc * 32 >= (1L << 31)) return false;
  // (every epilogue operand is addressed as uniform base + a 32-bit lane offset of at most 16 rows)
  auto wide = [](long ld) { return ld * 32 >= (1L << 31); };
  if ((p.res && wide(p.ldres)) || (p.sw_mode == 1 && (wide(p.sw_ldgu) || wide(p.sw_lddgu))) || (p.qk_q && (wide(p.qk_ldq) || wide(p.qk_ldk) || wide(p.qk_ldv))) ||
      (p.sp_act && wide(p.sp_ldact)) || (p.gelu_out && wide(p.ldg)) || (p.aux && wide(p.ldaux)) || (p.drop_bits && (wide(p.drop_bits_ld) || wide(p.lds2) || wide(p.ldr2))))
    return false;
  return true;
}

int gemm_pers_launch(GemmP p, hipStream_t st) {
  {
#if UR_LAB      // lab builds only (tools/lab/gemm_ws.hip linked in by tools/lab/gemm_ws_build.sh): the wave-specialised 128 x 256 kernel of round 5
    const int set = g_pers_mode.load(std::memory_order_relaxed);
    if (set == 2 && gemm_ws_eligible(p)) return gemm_ws_launch(p, st);
#endif
  }
  p.gm = p.M / BM; p.gn = p.N / BN;
  p.gcw = 0;
  {
    // column chunks on wide launches (gemm.hip): an XCD's 32 concurrent tiles cover (32 / cw) tile rows x cw tile columns and
    // every A row panel crosses the fabric gn / cw times.  UR_PERS_CW = n (lab) overrides the width where it divides gn.
    // Measured (gate|up, gn = 24, K = 1024; same process, interleaved): cw 6 1.448 ms, 12 1.482, 2 1.496, 4 1.507, row-major
    // 1.528, 8 1.535 -- the order moves the launch by +-3 % although the fabric reads differ x2.5 between them; q|k|v (gn 16):
    // 4 and 8 equal, 2 and row-major 1-2 % slower.
    static const int env_cw = ur_lab_int("UR_PERS_CW", -1);
    const int cw = env_cw >= 0 ? env_cw : ((p.gn % 6) == 0 && p.gn >= 24 ? 6 : 4);
    if (cw > 0 && p.gn >= 16 && (p.gm % 8) == 0 && p.gn > cw && (p.gn % cw) == 0) p.gcw = cw;
  }
  const bool drop = p.drop_bits != nullptr && p.K2 > 0;
  {
    static const int env_st = ur_lab_int("UR_PERS_STAGGER", 0);      // lab: cycles per start step; 0 = off (default)
    static const int env_sx = ur_lab_int("UR_PERS_STAGGER_XCD", 0);  // lab: cycles per XCD step
    p.stagger = env_sx > 0 ? -env_sx : (env_st > 0 ? env_st : 0);
  }
  const int mode = drop ? 2 : (p.K2 > 0 ? 1 : 0);
  const int epi = p.sp_act ? 4 : (p.qk_q ? 3 : (p.sw_mode == 1 ? 1 : (p.gelu_out ? 6 : (p.aux ? 7 : (p.res ? 2 : (p.bias ? 5 : 0))))));
#define UR_PERS_CASE(E, MD) if (epi == E && mode == MD) return launch_pers<E, MD>(p, st)
  UR_PERS_CASE(0, 0); UR_PERS_CASE(0, 1); UR_PERS_CASE(0, 2);
  UR_PERS_CASE(1, 0); UR_PERS_CASE(1, 1); UR_PERS_CASE(1, 2);
  UR_PERS_CASE(2, 0); UR_PERS_CASE(2, 1); UR_PERS_CASE(2, 2);
  UR_PERS_CASE(3, 0); UR_PERS_CASE(3, 1);
  UR_PERS_CASE(4, 0); UR_PERS_CASE(4, 1);
  UR_PERS_CASE(5, 0); UR_PERS_CASE(5, 1); UR_PERS_CASE(5, 2);
  UR_PERS_CASE(6, 0); UR_PERS_CASE(7, 0);
#undef UR_PERS_CASE
  UR_FAIL(-1, "ur_gemm(persistent): no kernel for epilogue %d, mode %d", epi, mode);
}

}  // namespace urgemm

#if UR_PERS_STAMPS
extern "C" int ur_lab_pers_stamps(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pers_stamps), sizeof(long long) * n);
}
#endif

extern "C" int ur_gemm_persistent_mode(int mode) {
  const int prev = urgemm::g_pers_mode.exchange(mode < 0 ? -1 : (mode > 2 ? 1 : mode));
  return prev;
}

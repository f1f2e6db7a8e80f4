// Fused MFMA attention (forward + two-kernel backward) for BOTH attention shapes on the path:
//   * Q-Former learned-query attention: small Q (<= 64 queries), K/V length T = Q (self), F (item
//     cross) or hist*Q_item (user cross, up to 3200), head_dim 64, additive key mask with the
//     reference's finfo.min semantics (a fully masked row is a UNIFORM softmax, not NaN) and
//     attention-probability dropout.      models/qformer.py:169-275
//   * Qwen3 causal GQA attention, head_dim 128, causal AND key-padding mask with SDPA semantics
//     (a query row with no allowed key outputs 0).   transformers modeling_qwen3.py:185-208,244-280
//
// Structure (per wave = one 32-query block; a workgroup = NW waves sharing LDS K/V tiles):
//   S^T = K Q^T with v_mfma_f32_32x32x16_bf16, KEYS on the MFMA rows -> every lane owns ONE query
//   column, so the online-softmax row reductions are 16 in-register ops + one cross-half shuffle,
//   and the P accumulators are directly the B operand of O^T += V^T P (no LDS round trip, no lane
//   movement).  V^T (and K^T / Q^T / dO^T in the backward) fragments come from the plain row-major
//   LDS tile through ds_read_b64_tr_b16.
//   Staging: 64-key tiles, double-buffered in LDS, next tile's global loads issued into registers
//   before the current tile's MFMAs and written to LDS after them (one barrier per tile).
//   head_dim 128 tiles use 256-B rows with a 16-B chunk XOR swizzle that is conflict-free for both the
//   ds_read_b128 row reads and the transposed reads; head_dim 64 tiles use 16-B padded rows.
//   Interior tiles (all keys valid, below the causal diagonal, no dropout) take a mask-free softmax:
//   p = exp2(fma(s, scale*log2e, -m*log2e)); O is rescaled only when some row's max moved.
// Backward = dQ kernel (same decomposition as forward) + dK/dV kernel (one 32-key block per wave,
// looping over the query heads of its GQA group): no atomics, bitwise reproducible.
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include "common.hip.h"
#include "unirec_hip.h"
#ifndef UR_ATTN_FWD_C128_HDR
#define UR_ATTN_FWD_C128_HDR "gen/attn_fwd_c128_asm.h"      // lab builds point this at an ablated variant (tools/lab/c128_variants.sh)
#endif
#include UR_ATTN_FWD_C128_HDR
#ifndef UR_ATTN_DQ_C128_HDR
#define UR_ATTN_DQ_C128_HDR "gen/attn_dq_c128_asm.h"
#endif
#include UR_ATTN_DQ_C128_HDR
#ifndef UR_ATTN_DKV_C128_HDR
#define UR_ATTN_DKV_C128_HDR "gen/attn_dkv_c128_asm.h"
#endif
#include UR_ATTN_DKV_C128_HDR

namespace {

constexpr float NEG_INF = -__builtin_huge_valf();
constexpr float F32_MIN = -3.4028234663852886e38f;   // torch.finfo(torch.float32).min
constexpr float LOG2E = 1.4426950408889634f;
constexpr int KT = 64;                                // keys (or queries, in dK/dV) staged per LDS tile
#ifndef UR_FWD_PRIO
#define UR_FWD_PRIO 1                                 // s_setprio(1) around the forward kernel's MFMA bursts (S chains, P V), 0 around the softmax: of the two waves of a
                                                      // SIMD the one in its matrix segment issues first (1801 -> 1764 us dense causal B 64 S 2048; the same hint in
                                                      // the dQ kernel is 1.5 % slower and is not applied there)
#endif
#ifndef UR_FWD_ABLATE
#define UR_FWD_ABLATE 0                               // lab (tools/lab/dkv2_ablate.sh <tag> "<n...>" UR_FWD_ABLATE; results WRONG when != 0): 1 no max/exp2, 2 no LDS fragment reads, 3 no staging of the next tile, 4 no P V MFMAs, 5 = 3 + no barrier
#endif
#ifndef UR_ATTN_DEFER_MAX
#define UR_ATTN_DEFER_MAX 1                           // lab: 0 = rescale O at every 32-key sub-tile
#endif
constexpr float DEFER_NAT = 6.0f * 0.6931471805599453f;   // defer the running-max update while it grows by < 2^6 (natural-log units)

template <int HD> struct Cfg {
  static constexpr int ROWB = (HD == 128) ? 256 : 144;
  static constexpr int NS = HD / 16;                  // MFMA k-steps contracting over head_dim
  static constexpr int NDT = HD / 32;                 // 32-wide head_dim tiles of a transposed accumulator
  static constexpr int CH = HD / 8;                   // 16-byte chunks per row
  static constexpr int TILE = KT * ROWB;
  // byte offset of 16-byte chunk `ch` of row `row`
  static __device__ __forceinline__ int off(int row, int ch) {
    if (HD == 128) return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    return 144 * row + 16 * ch;
  }
};

struct AttnP {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o; float* stats;
  long ldq, ldk, ldv, ldo;
  const uint8_t* kmask;
  int B, Sq, Sk, nq, nkv, rep;
  float scale; uint32_t drop_thr; float drop_inv; uint64_t seed;      // drop_thr: p * 65536 (16-bit decision fields, common.hip.h: ur_attn_pair_word); 0 = no dropout
  uint64_t drow0;       // dropout ROW index of (batch row 0, head 0, query 0) in the GLOBAL minibatch: drop_batch0 * nq * Sq
  uint32_t* rowkeys;    // backward: the two 32-bit dropout keys of every query row, planes [2][B*nq*Sq] behind the row constants of `delta`
                        // (written by the dQ kernel, whose lanes own query rows; read by the dK/dV kernels, whose lanes own keys)
  // backward
  const bf16_t* dout; bf16_t* dq; bf16_t* dk; bf16_t* dv; const float* delta;
  long lddo, lddq, lddk, lddv;
  // q-norm + RoPE backward fused into the dQ kernel's store (head_dim 128, ur_attn_bwd_args.rope_*): dq leaves as the
  // gradient of the RAW q projection
  const bf16_t* rp_raw; long rp_ldraw; const float* rp_w; const float* rp_cos; const float* rp_sin; float rp_eps;
  bf16_t* rp_draw; long rp_lddraw;
  const float* rp_rstd; long rp_rstd_ld; int rp_rstd_h0;      // non-null: rp_raw is the ROPED, normed q and 1 / rms comes from the forward (ur_attn_bwd_args.rope_rstd)
  // the k heads' q/k-norm + RoPE backward in the dK/dV kernel's store (with rp_rstd): roped k, its norm weight, first k column of rstd, raw-gradient output
  const bf16_t* rk_src; long rk_ld; const float* rk_w; int rk_rstd_h0; bf16_t* rk_dst; long rk_lddst;
  float* colsum_part;   // few-query dK/dV kernel: per (batch) partial column sums of dK | dV over the keys, [B][2][nq][hd] f32 (ur_attn_bwd_args.kv_colsum); NULL = off
  unsigned int* queue;  // work queues of the persistent dK/dV kernel: 8 words (one per XCD lane) in the CALLER's workspace, behind the two row-constant
                        // planes of `delta` (ur_attn_bwd_workspace_floats); zeroed by the dQ kernel of the same call
  // hand-scheduled causal head_dim-128 backward (both kernels or neither): plane 1 of `delta` holds -LSE * log2(e) instead of -LSE / scale
  int lse_log2;
};

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
// accumulator register r of lane-half h  <->  row index inside the 32-row tile
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// Pin a value inside the branch that uses it: without this hipcc speculates the masked path's
// per-element compare/select arithmetic (~110 VALU per sub-tile) into the mask-free interior path.
__device__ __forceinline__ uint32_t opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// Tile loader: rows [row0, row0+KT) x HD of a [S][ld] bf16 matrix -> LDS tile.
//   head_dim 128: LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write.  One wave
//     instruction writes 1 KiB linearly (4 rows of 256 B), so the XOR swizzle is applied to the SOURCE
//     chunk (pos ^ swz(row)); rows past S are clamped to S-1 (finite data, masked by the key state).
//     issue() starts the copy into the tile that will be read NEXT iteration; the __syncthreads() that
//     ends the iteration drains it (vmcnt(0) + barrier).
//   head_dim 64 (Q-Former): global -> registers at issue(), registers -> padded LDS rows at commit().
template <int HD, int NT>
struct Loader {
  static constexpr bool DMA = (HD == 128);
  static constexpr int N = (KT * Cfg<HD>::CH) / NT;
  // 256-thread workgroups: piece i of a tile covers rows 16i + 4*wave + (lane>>4), and the swizzled source
  // chunk does not depend on i, so the lane's source offset is loop invariant (voff) and a tile costs
  // no vector arithmetic: uniform row base (scalar) + voff.
  static constexpr bool HOIST = DMA && NT == 256;
  uint4 r[DMA ? 1 : N];
  uint32_t voff;
  __device__ __forceinline__ void init(long ld, int tid) {
    if (HOIST) {
      const int lane = tid & 63, wave = tid >> 6;
      const int row = 4 * wave + (lane >> 4), pos = lane & 15;
      voff = (uint32_t)(row * ld + (pos ^ (((row & 3) << 2) | ((row >> 2) & 3))) * 8) * 2u;
    }
  }
  __device__ __forceinline__ void issue(char* tile, const bf16_t* __restrict__ base, long ld, int row0, int S, int tid) {
    if (DMA) {
      typedef __attribute__((address_space(3))) void lds_void;
      typedef const __attribute__((address_space(1))) void gbl_void;
      const int lane = tid & 63;
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      if (HOIST && row0 + KT <= S) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const char* ub = reinterpret_cast<const char*>(base + (long)(row0 + 16 * i) * ld);     // wave-uniform
          __builtin_amdgcn_global_load_lds((gbl_void*)(ub + voff), (lds_void*)(tile + (i * 4 + wave) * 1024), 16, 0, 0);
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int inst = i * (NT / 64) + wave;                 // wave-uniform: 1 KiB piece index
        const int c = inst * 64 + lane;
        const int row = c >> 4, pos = c & 15;
        const int g = min(row0 + row, S - 1);
        const int src_chunk = pos ^ (((row & 3) << 2) | ((row >> 2) & 3));
        const bf16_t* src = base + (long)g * ld + src_chunk * 8;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(tile + inst * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int c = tid + i * NT;
        const int row = c / Cfg<HD>::CH, cc = c % Cfg<HD>::CH;
        const int g = row0 + row;
        r[i] = (g < S) ? *reinterpret_cast<const uint4*>(base + (long)g * ld + cc * 8) : make_uint4(0, 0, 0, 0);
      }
    }
  }
  __device__ __forceinline__ void commit(char* tile, int tid) const {
    if (!DMA) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int c = tid + i * NT;
        const int row = c / Cfg<HD>::CH, cc = c % Cfg<HD>::CH;
        *reinterpret_cast<uint4*>(tile + Cfg<HD>::off(row, cc)) = r[i];
      }
    }
  }
};

// row fragment: X[r0 + (lane&31)][16*s + 8*(lane>>5) + 0..7]   (MFMA A or B operand, k = head_dim)
template <int HD>
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int r0, int s, int lane) {
  return *reinterpret_cast<const bf16x8*>(tile + Cfg<HD>::off(r0 + (lane & 31), 2 * s + (lane >> 5)));
}
// transposed fragments: A[m = 32*dt + (lane&31)][k-element j] = X[r0 + 8*(j>>2) + 4*(lane>>5) + (j&3)][m]
// for dt = 0..NDT-1 (16 rows r0..r0+15 of X; k order matches an accumulator tile used as the B operand).
// Inline-asm ds_read_b64_tr_b16 batches (common.hip.h: the builtin would drain the LDS-DMA prefetch).
template <int HD>
__device__ __forceinline__ void tr_frags(bf16x8 (&f)[Cfg<HD>::NDT], const char* tile, int r0, int lane) {
  const int h = lane >> 5, g16 = (lane >> 4) & 1, i = lane & 15;
  const int row = r0 + 4 * h + (i >> 2), sub8 = 8 * (i & 1);
  const uint32_t base = lds_off(tile) + sub8;
  uint32_t a[Cfg<HD>::NDT], b[Cfg<HD>::NDT];
#pragma unroll
  for (int dt = 0; dt < Cfg<HD>::NDT; ++dt) {
    const int ch = 4 * dt + 2 * g16 + ((i & 3) >> 1);
    a[dt] = base + Cfg<HD>::off(row, ch);
    b[dt] = base + Cfg<HD>::off(row + 8, ch);
  }
  tr_read(f, a, b);
}
// split form of tr_frags: issue the 8 reads of one 4-fragment batch now, wait for them later (tr_landed<N>: at most N younger LDS
// reads still outstanding); between the two the registers hold no data yet
__device__ __forceinline__ void tr_issue4(bf16x4 (&lo)[4], bf16x4 (&hi)[4], const uint32_t (&a)[4], const uint32_t (&b)[4]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\t"
      "ds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
      "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\t"
      "ds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15"
      : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3])
      : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]));
}
template <int N>
__device__ __forceinline__ void tr_landed(bf16x4 (&lo)[4], bf16x4 (&hi)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3])
               : "n"(N));
}

// head_dim 128 with the lane-constant part of the swizzled offsets hoisted out of the tile loop (LaneOff, 16 registers):
// a row fragment then costs one add (tile base + offset) and a transposed batch four, against ~5 / ~24 vector
// instructions of swizzle arithmetic per call (the swizzle repeats every 16 rows, so a 32- or 16-row step is an immediate)
struct LaneOff128 {
  uint32_t rf[8], ta[4], tb[4];
  __device__ __forceinline__ void init(int lane) {
    using C = Cfg<128>;
    const int h = lane >> 5;
#pragma unroll
    for (int st = 0; st < 8; ++st) rf[st] = C::off(lane & 31, 2 * st + h);
    const int g16 = (lane >> 4) & 1, i = lane & 15;
    const int row = 4 * h + (i >> 2), sub8 = 8 * (i & 1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int ch = 4 * dt + 2 * g16 + ((i & 3) >> 1);
      ta[dt] = C::off(row, ch) + sub8;
      tb[dt] = C::off(row + 8, ch) + sub8;
    }
  }
};
template <int R0>
__device__ __forceinline__ bf16x8 row_frag_h(const char* tile, const LaneOff128& lo, int s) {
  return *reinterpret_cast<const bf16x8*>(tile + lo.rf[s] + 256 * R0);
}
template <int R0>
__device__ __forceinline__ void tr_frags_h(bf16x8 (&f)[4], const char* tile, const LaneOff128& lo) {
  const uint32_t base = lds_off(tile) + 256 * R0;
  uint32_t a[4], b[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { a[dt] = base + lo.ta[dt]; b[dt] = base + lo.tb[dt]; }
  tr_read(f, a, b);
}

// accumulator registers 8*s2 .. 8*s2+7 -> bf16 B-operand fragment of k-step s2
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s2) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t w = pack_bf2(a[8 * s2 + 2 * j], a[8 * s2 + 2 * j + 1]);
    r[2 * j] = (short)(w & 0xffffu);
    r[2 * j + 1] = (short)(w >> 16);
  }
  return r;
}
// global row fragment (same element map as row_frag), zero when !ok
__device__ __forceinline__ bf16x8 g_frag(const bf16_t* rowp, int s, int lane, bool ok) {
  bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  if (!ok) return z;
  return *reinterpret_cast<const bf16x8*>(rowp + 16 * s + 8 * (lane >> 5));
}
// store a transposed accumulator set (lane = row of the output matrix, regs = head_dim) as bf16.  The two lane halves
// of a row hold alternating 4-column groups, i.e. 8-byte pieces: one v_permlane32_swap per dword trades the upper
// half's group k against the lower half's group k+1, after which every lane owns 16 contiguous bytes -- 8 dwordx4
// stores per row instead of 16 dwordx2 (the store tail is issue-bound).  Called by ALL lanes of the wave (the swaps
// need the full EXEC mask); `ok` predicates the stores only.
template <int HD>
__device__ __forceinline__ void store_T(bf16_t* rowp, const f32x16 (&acc)[Cfg<HD>::NDT], float mul, int lane, bool ok) {
  const int h = lane >> 5;
  const bool wide = __all(!ok || ((reinterpret_cast<uintptr_t>(rowp) & 15) == 0));       // wave-uniform
#pragma unroll
  for (int dt = 0; dt < Cfg<HD>::NDT; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; rq += 2) {
      uint2 a = make_uint2(pack_bf2(acc[dt][4 * rq] * mul, acc[dt][4 * rq + 1] * mul), pack_bf2(acc[dt][4 * rq + 2] * mul, acc[dt][4 * rq + 3] * mul));
      uint2 b = make_uint2(pack_bf2(acc[dt][4 * rq + 4] * mul, acc[dt][4 * rq + 5] * mul), pack_bf2(acc[dt][4 * rq + 6] * mul, acc[dt][4 * rq + 7] * mul));
      if (wide) {
        auto r0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
        // lanes 0-31: [own group k | upper half's group k] = columns 8k .. 8k+7; lanes 32-63: the next eight
        if (ok) *reinterpret_cast<uint4*>(rowp + 32 * dt + 8 * rq + 8 * h) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
      } else if (ok) {
        *reinterpret_cast<uint2*>(rowp + 32 * dt + 8 * rq + 4 * h) = a;
        *reinterpret_cast<uint2*>(rowp + 32 * dt + 8 * rq + 8 + 4 * h) = b;
      }
    }
}

// Per-tile key state of the 64 keys k0..k0+63 as wave-uniform bit masks (bit i = key k0+i):
// `valid` = inside Sk and allowed by key_mask, `inr` = inside Sk.
struct KeyBits { unsigned long long valid, inr; };
__device__ __forceinline__ KeyBits key_bits(const uint8_t* km, int k0, int Sk, int lane) {
  const int key = k0 + lane;
  const bool in = key < Sk;
  const bool ok = in && (km == nullptr || km[key] != 0);
  KeyBits kb; kb.valid = __ballot(ok); kb.inr = __ballot(in);
  return kb;
}
// All tiles' key-state words are computed ONCE per workgroup into LDS before the first LDS-DMA is
// issued: an ordinary global load inside the DMA loop would make hipcc drain the prefetch (vmcnt(0)).
constexpr int MAX_KTILES = 128;                      // Sk <= 8192
__device__ __forceinline__ void fill_key_words(unsigned long long* kw, const uint8_t* km, int ntiles, int Sk, int tid, int nthreads) {
  const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
  for (int t = wave; t < ntiles; t += nw) {
    const KeyBits kb = key_bits(km, t * KT, Sk, lane);
    if (lane == 0) { kw[2 * t] = kb.valid; kw[2 * t + 1] = kb.inr; }
  }
}

// Causal (SDPA) semantics only: a key tile without a single valid key contributes exactly nothing (its scores are
// -inf for every query), so the leading run of such tiles -- the left padding of a prompt, which every later query
// block of the sequence would otherwise sweep -- is skipped together with its K/V loads.  The Q-Former's additive
// finfo.min mask must NOT take this path: a fully masked row is a uniform softmax over all keys there.
__device__ __forceinline__ int first_valid_tile(const unsigned long long* kw, int ntiles) {
  int t = 0;
  while (t < ntiles && kw[2 * t] == 0ull) ++t;
  return t;
}

// Additive-mask (Q-Former) semantics: a masked key's score is finfo.min, so its probability is exp(finfo.min - m) = 0 EXACTLY as
// soon as the row has one allowed key (m is then a real score); only a row without any allowed key is the uniform softmax over all
// keys.  Every query of a sample sees the same key mask, so when the sample has a valid key the trailing run of key tiles without
// one -- the right padding of a ragged history, 25 % of the keys of the C3 batch -- contributes nothing to O, l, dQ (and gets
// dK = dV = 0): the sweep ends at the last tile that holds a valid key, K / V loads included.  Bit-identical to the full sweep.
__device__ __forceinline__ int trim_masked_tail(const unsigned long long* kw, int ntiles) {
  int t = ntiles;
  while (t > 0 && kw[2 * (t - 1)] == 0ull) --t;
  return t == 0 ? ntiles : t;          // no valid key at all: the uniform row needs every key
}

// Workgroup -> (x block, head, batch).  The grid is 1-D.  Consecutive workgroup ids are dealt round-robin to
// the 8 XCDs, each with its own L2, so the ids are re-read as (xcd = id % 8, slot = id / 8): every (batch,
// kv head) group -- whose blocks stream the same K/V (forward, dQ) or the same Q/dO (dK/dV) -- gets all its
// blocks on ONE XCD, adjacent in time, and under the causal mask the heaviest block of the group first.
struct BlockMap { int x, head, b; };
template <bool HEAVY_LAST>     // HEAVY_LAST: work grows with x (forward, dQ); else it shrinks (dK/dV)
__device__ __forceinline__ BlockMap block_map_id(int id, int nx, int heads_per_group, int ngroups_per_batch, int B) {
  const int gsz = nx * heads_per_group, ngroups = ngroups_per_batch * B;
  int grp, j;
  if ((ngroups & 7) == 0) { const int slot = id >> 3; grp = (slot / gsz) * 8 + (id & 7); j = slot % gsz; }
  else { grp = id / gsz; j = id % gsz; }
  BlockMap m;
  const int xi = j / heads_per_group;
  m.x = HEAVY_LAST ? nx - 1 - xi : xi;
  m.head = (grp % ngroups_per_batch) * heads_per_group + j % heads_per_group;
  m.b = grp / ngroups_per_batch;
  return m;
}

template <bool HEAVY_LAST>
__device__ __forceinline__ BlockMap block_map(int nx, int heads_per_group, int ngroups_per_batch, int B) {
  return block_map_id<HEAVY_LAST>((int)blockIdx.x, nx, heads_per_group, ngroups_per_batch, B);
}

// masked, scaled score (natural-log domain).  CAUSAL: SDPA semantics (-inf); else the Q-Former's
// additive finfo.min (the sum collapses to exactly finfo.min in f32).
template <bool CAUSAL>
__device__ __forceinline__ float mask_score(float raw, float scale, bool valid, bool inr, int key, int qpos) {
  if (CAUSAL) return (valid && key <= qpos) ? raw * scale : NEG_INF;
  return !inr ? NEG_INF : (valid ? raw * scale : F32_MIN);
}

// ================================================================================================
// GQ2 (lab, UR_FWD_GQ2=1): NW = 8 waves = the SAME 128 queries of the two query heads of one kv head (GQA 2:1): one staged K / V
// tile serves both heads (half the LDS-DMA pieces per wave and tile, half the L2 -> LDS bytes per flop).  Measured: 1.82-1.84 ms
// against 1.79-1.80 ms for the one-head workgroups (dense causal B 64 S 2048): the two waves of a SIMD now wait at the SAME
// barrier, which costs more than the staging it saves.  Not the default.
// (NW == 1: the few-query launches of the Q-Formers -- one wave per (sample, head), thousands of them, each a short latency-bound
// chain: two waves per SIMD instead of the one that 276 registers allowed, and LDS for ONE K | V stage when Sk fits one tile)
template <int HD, bool CAUSAL, int NW, bool GQ2 = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 1 ? 2 : 1))) void attn_fwd_kernel(AttnP p) {
  constexpr int NWQ = GQ2 ? NW / 2 : NW;              // waves along the query axis
  using C = Cfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const BlockMap bm = block_map<true>((p.Sq + 32 * NWQ - 1) / (32 * NWQ), GQ2 ? p.rep / 2 : p.rep, p.nkv, p.B);
  const int wq = GQ2 ? wave % NWQ : wave;
  const int hq = GQ2 ? bm.head * 2 + wave / NWQ : bm.head, b = bm.b, kvh = hq / p.rep;
  const int qblk = bm.x * (32 * NWQ) + wq * 32;
  const int q = qblk + (lane & 31);
  const bool qok = q < p.Sq;

  const bf16_t* qrow = p.q + ((long)b * p.Sq + (qok ? q : 0)) * p.ldq + (long)hq * HD;
  bf16x8 qf[C::NS];
#pragma unroll
  for (int s = 0; s < C::NS; ++s) qf[s] = g_frag(qrow, s, lane, qok);

  [[maybe_unused]] LaneOff128 lo;
  if constexpr (HD == 128) lo.init(lane);
  auto kfrag = [&](const char* tile, auto R0c, int st) {
    constexpr int R0 = decltype(R0c)::value;
    if constexpr (HD == 128) return row_frag_h<R0>(tile, lo, st);
    else return row_frag<HD>(tile, R0, st, lane);
  };
  f32x16 o[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt) o[dt] = zero16();
  float m = NEG_INF, l = 0.f;          // running max (natural-log domain, scaled scores) and row sum
  const float c2 = p.scale * LOG2E;

  int kend = p.Sk;
  if (CAUSAL) kend = min(p.Sk, (bm.x + 1) * (32 * NWQ));
  int ntiles = (kend + KT - 1) / KT;
  const bf16_t* kb = p.k + (long)b * p.Sk * p.ldk + (long)kvh * HD;
  const bf16_t* vb = p.v + (long)b * p.Sk * p.ldv + (long)kvh * HD;
  const uint8_t* km = p.kmask ? p.kmask + (long)b * p.Sk : nullptr;
  const bool dropping = (!CAUSAL) && p.drop_thr != 0;
  ur_rowkey rk = {0u, 0u};            // this lane's query row: two keys for the whole sweep
  if (dropping) rk = ur_attn_row_key(p.seed, p.drow0 + ((uint64_t)((long)b * p.nq + hq) * p.Sq + (uint64_t)q));

  unsigned long long* kwords = reinterpret_cast<unsigned long long*>(smem + (p.Sk <= KT ? 2 : 4) * C::TILE);      // (ur_attn_fwd's LDS size)
  fill_key_words(kwords, km, ntiles, p.Sk, tid, NW * 64);
  int t_first = 0;
  if (km != nullptr) {
    __syncthreads();
    if (CAUSAL) t_first = first_valid_tile(kwords, ntiles);
    else ntiles = trim_masked_tail(kwords, ntiles);
  }
  Loader<HD, NW * 64> ks, vs;
  ks.init(p.ldk, tid); vs.init(p.ldv, tid);
  if (t_first < ntiles) {
    char* first = smem + (t_first & 1) * 2 * C::TILE;
    ks.issue(first, kb, p.ldk, t_first * KT, p.Sk, tid);
    vs.issue(first + C::TILE, vb, p.ldv, t_first * KT, p.Sk, tid);
    ks.commit(first, tid);
    vs.commit(first + C::TILE, tid);
  }
  __syncthreads();

  for (int t = t_first; t < ntiles; ++t) {
    const int k0 = t * KT;
    const char* ktile = smem + (t & 1) * 2 * C::TILE;
    const char* vtile = ktile + C::TILE;
    char* nk = smem + ((t + 1) & 1) * 2 * C::TILE;
    if (t + 1 < ntiles && UR_FWD_ABLATE != 3 && UR_FWD_ABLATE != 5) {
      ks.issue(nk, kb, p.ldk, k0 + KT, p.Sk, tid);
      vs.issue(nk + C::TILE, vb, p.ldv, k0 + KT, p.Sk, tid);
    }
    KeyBits kbits; kbits.valid = kwords[2 * t]; kbits.inr = kwords[2 * t + 1];
    if (qblk < p.Sq) {
      // both 32-key sub-tiles' S = K Q^T chains are issued before the first softmax: the second chain runs on the
      // matrix pipe while the vector units do the first sub-tile's maximum / exp2 / row sum
      // causal: a 32-key sub-tile without a valid key adds nothing (see first_valid_tile)
      const bool act0 = k0 < kend && !(CAUSAL && (k0 > qblk + 31 || (uint32_t)kbits.valid == 0u));
      const bool act1 = (k0 + 32) < kend && !(CAUSAL && (k0 + 32 > qblk + 31 || (uint32_t)(kbits.valid >> 32) == 0u));
      f32x16 sA = zero16(), sB = zero16();
#if UR_FWD_PRIO
      __builtin_amdgcn_s_setprio(1);
#endif
      if (act0) {
#pragma unroll
        for (int st = 0; st < C::NS; ++st)
          sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(UR_FWD_ABLATE == 2 ? qf[(st + 1) % C::NS] : kfrag(ktile, std::integral_constant<int, 0>{}, st), qf[st], sA, 0, 0, 0);
      }
      if (act1) {
#pragma unroll
        for (int st = 0; st < C::NS; ++st)
          sB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(UR_FWD_ABLATE == 2 ? qf[(st + 2) % C::NS] : kfrag(ktile, std::integral_constant<int, 32>{}, st), qf[st], sB, 0, 0, 0);
      }
#if UR_FWD_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      auto soft_pv = [&](const int sub, f32x16& s) {
        const int kbase = k0 + 32 * sub;
        const uint32_t v32 = (uint32_t)(kbits.valid >> (32 * sub)), i32 = (uint32_t)(kbits.inr >> (32 * sub));
        // interior sub-tile (every key valid and below the diagonal), or -- `diag` -- every key valid ON the causal diagonal:
        // the same mask-free softmax after one position compare per element (the general path below costs ~7 vector
        // instructions per element for the key-state bits; every 32-query block crosses the diagonal once)
        const bool allv = (v32 == 0xffffffffu);          // (with dropout too: the keep factors are applied to the finished probabilities below)
        const bool diag = CAUSAL && allv && kbase + 31 > qblk;
        const bool fast = allv;
        float mx = NEG_INF;
        if (fast) {
          if (diag) {
            const int dqk = opaque(q - kbase - 4 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = ((r & 3) + 8 * (r >> 2) <= dqk) ? s[r] : NEG_INF;
          }
#if UR_FWD_ABLATE == 1
          mx = s[0];
#else
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
#endif
          mx *= p.scale;
        } else {
          const uint32_t vh = opaque(v32 >> (4 * h)), ih = opaque(i32 >> (4 * h));
          const int qq = opaque(q), kb0 = opaque(kbase + 4 * h);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int bit = (r & 3) + 8 * (r >> 2);
            s[r] = mask_score<CAUSAL>(s[r], p.scale, (vh >> bit) & 1u, (ih >> bit) & 1u, kb0 + bit, qq);
            mx = fmaxf(mx, s[r]);
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // branch-free rescale (alpha == 1 exactly when the max did not move): keeps the O accumulators
        // in place across the loop (a conditional rescale made hipcc copy all 64 registers per sub-tile)
#if UR_ATTN_DEFER_MAX
        // deferred running maximum: O and l are rescaled only when some row's maximum grew by more than 2^DEFER_LOG2
        // (wave-uniform branch); until then p = exp2(c*s - m_old) may exceed 1 by at most that factor, which f32 sums
        // and the bf16 P fragments carry at unchanged relative precision.  m = -inf (first tile) always takes the branch.
        if (__builtin_amdgcn_ballot_w64(!(mx <= m + DEFER_NAT)) != 0ull) {
          const float mnew = fmaxf(m, mx);
          const float mu = (mnew == NEG_INF) ? 0.f : mnew;
          const float alpha = fast_exp2((m - mu) * LOG2E);
          l *= alpha;
#pragma unroll
          for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
          m = mnew;
        }
#else
        {
          const float mnew = fmaxf(m, mx);
          const float mu = (mnew == NEG_INF) ? 0.f : mnew;
          const float alpha = fast_exp2((m - mu) * LOG2E);
          l *= alpha;
#pragma unroll
          for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
          m = mnew;
        }
#endif
        const float muse = (m == NEG_INF) ? 0.f : m;
        const float mc = muse * LOG2E;
        float rs = 0.f;
        if (fast) {
#pragma unroll
#if UR_FWD_ABLATE == 1
          for (int r = 0; r < 16; ++r) { rs += s[r]; }
#else
          for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], c2, -mc)); rs += s[r]; }
#endif
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[r] = fast_exp2((s[r] - muse) * LOG2E); rs += s[r]; }
        }
        if (dropping) {
          // registers 4 g .. 4 g + 3 hold keys kbase + 8 g + 4 h + 0 .. 3: two pair words per group (models/qformer.py:258: the row sum
          // above is taken BEFORE dropout, as softmax-then-dropout requires)
          const uint32_t kp0 = (uint32_t)(kbase + 4 * h) >> 1;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint32_t w0 = ur_attn_pair_word(rk.k1, rk.k2, kp0 + 4 * g), w1 = ur_attn_pair_word(rk.k1, rk.k2, kp0 + 4 * g + 1);
            s[4 * g] *= ur_attn_keep_scale(w0, 0u, p.drop_thr, p.drop_inv);
            s[4 * g + 1] *= ur_attn_keep_scale(w0, 1u, p.drop_thr, p.drop_inv);
            s[4 * g + 2] *= ur_attn_keep_scale(w1, 0u, p.drop_thr, p.drop_inv);
            s[4 * g + 3] *= ur_attn_keep_scale(w1, 1u, p.drop_thr, p.drop_inv);
          }
        }
        rs += __shfl_xor(rs, 32, 64);
        l += rs;
#if UR_FWD_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_frag(s, s2);
          bf16x8 vt[C::NDT];
          if constexpr (UR_FWD_ABLATE == 2) {
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt) vt[dt] = qf[dt];
          } else if constexpr (HD == 128) {
            if (sub == 0) { if (s2 == 0) tr_frags_h<0>(vt, vtile, lo); else tr_frags_h<16>(vt, vtile, lo); }
            else { if (s2 == 0) tr_frags_h<32>(vt, vtile, lo); else tr_frags_h<48>(vt, vtile, lo); }
          } else {
            tr_frags<HD>(vt, vtile, 32 * sub + 16 * s2, lane);
          }
#pragma unroll
#if UR_FWD_ABLATE == 4
          for (int dt = 0; dt < C::NDT; ++dt) o[dt][s2] += (float)(vt[dt][0] ^ pf[dt & 7]);
#else
          for (int dt = 0; dt < C::NDT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt[dt], pf, o[dt], 0, 0, 0);
#endif
        }
      };
      if (act0) soft_pv(0, sA);
#if UR_FWD_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      if (act1) soft_pv(1, sB);
#if UR_FWD_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    if (t + 1 < ntiles && UR_FWD_ABLATE != 3 && UR_FWD_ABLATE != 5) {
      ks.commit(nk, tid);
      vs.commit(nk + C::TILE, tid);
    }
#if UR_FWD_ABLATE != 5
    __syncthreads();
#endif
  }
  {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    store_T<HD>(p.o + ((long)b * p.Sq + (qok ? q : 0)) * p.ldo + (long)hq * HD, o, inv, lane, qok);
  }
  if (qok) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (h == 0) {
      float* st = p.stats + (((long)b * p.nq + hq) * p.Sq + q) * 2;
      st[0] = (m == NEG_INF) ? 0.f : m;
      st[1] = inv;
    }
  }
}

// q/k-norm + RoPE backward of one row from the ROPED, normed forward output (qknorm_rope_bwd_roped_kernel's arithmetic in the transposed
// accumulator layout: the lane holds d = 32 dt + acc_row(r, h) of its row, the rotate-half partner d +- 64 is the same register of tile
// dt +- 2).  x^ = R^T(o) / w, g = R^T(d) w, d_raw = rstd (g - x^ mean(g x^)).  d: gradient of the roped row in, of the raw projection out.
template <int HD>
__device__ __forceinline__ void rope_bwd_from_roped(f32x16 (&d)[Cfg<HD>::NDT], const bf16_t* orow, const float* w, const float* cr, const float* sr, float rs, int h) {
  uint2 xp[4][4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) xp[dt][rq] = *reinterpret_cast<const uint2*>(orow + 32 * dt + 8 * rq + 4 * h);
  float xh[4][16];
  float t = 0.f;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int j = 32 * dt + 8 * rq + 4 * h;
      const float4 c4 = *reinterpret_cast<const float4*>(cr + j), s4 = *reinterpret_cast<const float4*>(sr + j);
      const float4 wa = *reinterpret_cast<const float4*>(w + j), wb = *reinterpret_cast<const float4*>(w + j + 64);
      const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
      const float wA[4] = {wa.x, wa.y, wa.z, wa.w}, wB[4] = {wb.x, wb.y, wb.z, wb.w};
      const float oA[4] = {bf_lo(xp[dt][rq].x), bf_hi(xp[dt][rq].x), bf_lo(xp[dt][rq].y), bf_hi(xp[dt][rq].y)};
      const float oB[4] = {bf_lo(xp[dt + 2][rq].x), bf_hi(xp[dt + 2][rq].x), bf_lo(xp[dt + 2][rq].y), bf_hi(xp[dt + 2][rq].y)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * rq + e;
        const float hA = (oA[e] * cc[e] + oB[e] * sn[e]) * __builtin_amdgcn_rcpf(wA[e]);      // forward: o = xn c -+ partner(xn) s
        const float hB = (oB[e] * cc[e] - oA[e] * sn[e]) * __builtin_amdgcn_rcpf(wB[e]);
        const float dA = bf2f(f2bf(d[dt][r])), dB = bf2f(f2bf(d[dt + 2][r]));                 // (the standalone kernel reads the gradient back as bf16)
        const float gA = (dA * cc[e] + dB * sn[e]) * wA[e];
        const float gB = (dB * cc[e] - dA * sn[e]) * wB[e];
        t += gA * hA + gB * hB;
        d[dt][r] = gA; d[dt + 2][r] = gB;
        xh[dt][r] = hA; xh[dt + 2][r] = hB;
      }
    }
  t += __shfl_xor(t, 32, 64);
  t /= (float)HD;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) d[dt][r] = rs * (d[dt][r] - xh[dt][r] * t);
}

// Store of a 32-query block's dQ^T accumulators (lane = query row, registers = head_dim), shared by both dQ kernels: plain bf16 rows, or
// -- head_dim 128 with ur_attn_bwd_args.rope_* -- the q-norm + RoPE backward applied in registers first (dq leaves as the gradient of
// the RAW q projection).
template <int HD>
__device__ __forceinline__ void dq_store_block(const AttnP& p, f32x16 (&dq)[Cfg<HD>::NDT], int b, int hq, int q, bool qok, int lane) {
  const int h = lane >> 5;
  if (HD == 128 && p.rp_raw != nullptr) {
    // Qwen3Attention: q = rope(q_norm(q_raw)) (modeling_qwen3.py:59-64,107-170,244-252).  dq above is the gradient of the
    // ROTATED, NORMED q; the lane holds half of its row (d = 32 dt + acc_row(r, h)), and the rotate-half partner d +- 64 is
    // the same register of tile dt +- 2, so the whole chain back to the raw projection -- un-rotate, norm weight, RMS-norm
    // backward with its two row sums (registers + one cross-half shuffle each) -- is lane-local.  Same arithmetic as
    // qknorm_rope_kernel<128, true>; the 2 + 2 activation passes of writing dq and reading it back are gone.
    const long qrow = (long)b * p.Sq + (qok ? q : 0);
    const bf16_t* xr = p.rp_raw + qrow * p.rp_ldraw + (long)hq * HD;
    const int pos = qok ? q : 0;
    const float* cr = p.rp_cos + (long)pos * (HD / 2);
    const float* sr = p.rp_sin + (long)pos * (HD / 2);
    uint2 xp[4][4];
    if (p.rp_rstd != nullptr) {
      // the forward ran as the q|k|v GEMM's epilogue: rp_raw holds the ROPED, normed q
      rope_bwd_from_roped<HD>(dq, xr, p.rp_w, cr, sr, p.rp_rstd[qrow * p.rp_rstd_ld + p.rp_rstd_h0 + hq], h);
      store_T<HD>(p.rp_draw + qrow * p.rp_lddraw + (long)hq * HD, dq, 1.0f, lane, qok);
      return;
    }
    float ss = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        xp[dt][rq] = *reinterpret_cast<const uint2*>(xr + 32 * dt + 8 * rq + 4 * h);
        const float x0 = bf_lo(xp[dt][rq].x), x1 = bf_hi(xp[dt][rq].x), x2 = bf_lo(xp[dt][rq].y), x3 = bf_hi(xp[dt][rq].y);
        ss += x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
      }
    ss += __shfl_xor(ss, 32, 64);
    const float rs = rsqrtf(ss / (float)HD + p.rp_eps);
    float t = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int j = 32 * dt + 8 * rq + 4 * h;              // d of the first-half element; its partner is d + 64
        const float4 c4 = *reinterpret_cast<const float4*>(cr + j), s4 = *reinterpret_cast<const float4*>(sr + j);
        const float4 wa = *reinterpret_cast<const float4*>(p.rp_w + j), wb = *reinterpret_cast<const float4*>(p.rp_w + j + 64);
        const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
        const float wA[4] = {wa.x, wa.y, wa.z, wa.w}, wB[4] = {wb.x, wb.y, wb.z, wb.w};
        const float xA[4] = {bf_lo(xp[dt][rq].x), bf_hi(xp[dt][rq].x), bf_lo(xp[dt][rq].y), bf_hi(xp[dt][rq].y)};
        const float xB[4] = {bf_lo(xp[dt + 2][rq].x), bf_hi(xp[dt + 2][rq].x), bf_lo(xp[dt + 2][rq].y), bf_hi(xp[dt + 2][rq].y)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * rq + e;
          // the standalone kernel reads dq back as bf16: round here too, so both paths see the same upstream gradient
          const float dA = bf2f(f2bf(dq[dt][r])), dB = bf2f(f2bf(dq[dt + 2][r]));
          const float gA = (dA * cc[e] + dB * sn[e]) * wA[e];            // d <  64: dy c + dy[d+64] s
          const float gB = (dB * cc[e] - dA * sn[e]) * wB[e];            // d >= 64: dy c - dy[d-64] s
          t += gA * (xA[e] * rs) + gB * (xB[e] * rs);
          dq[dt][r] = gA; dq[dt + 2][r] = gB;
        }
      }
    t += __shfl_xor(t, 32, 64);
    t /= (float)HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float xv[4] = {bf_lo(xp[dt][rq].x), bf_hi(xp[dt][rq].x), bf_lo(xp[dt][rq].y), bf_hi(xp[dt][rq].y)};
#pragma unroll
        for (int e = 0; e < 4; ++e) dq[dt][4 * rq + e] = rs * (dq[dt][4 * rq + e] - xv[e] * rs * t);
      }
    store_T<HD>(p.rp_draw + qrow * p.rp_lddraw + (long)hq * HD, dq, 1.0f, lane, qok);
    return;
  }
  store_T<HD>(p.dq + ((long)b * p.Sq + (qok ? q : 0)) * p.lddq + (long)hq * HD, dq, 1.0f, lane, qok);
}

// ================================================================================================
// dQ: same decomposition as forward.  dQ^T[d][q] += K^T[d][key] * dS^T[key][q]
template <int HD, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(AttnP p) {
  using C = Cfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const BlockMap bm = block_map<true>((p.Sq + 32 * NW - 1) / (32 * NW), p.rep, p.nkv, p.B);
  const int hq = bm.head, b = bm.b, kvh = hq / p.rep;
  const int qblk = bm.x * (32 * NW) + wave * 32;
  const int q = qblk + (lane & 31);
  const bool qok = q < p.Sq;
  const long qtok = (long)b * p.Sq + (qok ? q : 0);

  bf16x8 qf[C::NS], dof[C::NS];
#pragma unroll
  for (int s = 0; s < C::NS; ++s) {
    qf[s] = g_frag(p.q + qtok * p.ldq + (long)hq * HD, s, lane, qok);
    dof[s] = g_frag(p.dout + qtok * p.lddo + (long)hq * HD, s, lane, qok);
  }
  const long srow = ((long)b * p.nq + hq) * p.Sq + (qok ? q : 0);
  const float m = p.stats[srow * 2], inv = qok ? p.stats[srow * 2 + 1] : 0.f;
  // Row constants of the backward, computed here for this block's query rows (the two lane halves of a row hold
  // complementary halves of head_dim) and published for the dK/dV kernel that runs next on the stream:
  //   plane 0: nd[row] = -sum_d dO[q][d] * O[q][d]        plane 1: ns[row] = -(m + ln l) / scale = -LSE / scale
  // (a row without any allowed key, l = 0 under SDPA semantics, gets ns = -inf: P = 0)
  float dlt = 0.f;
#pragma unroll
  for (int s = 0; s < C::NS; ++s) {
    const bf16x8 of = g_frag(p.o + qtok * p.ldo + (long)hq * HD, s, lane, qok);
#pragma unroll
    for (int e = 0; e < 8; ++e) dlt = fmaf(bf2f((bf16_t)dof[s][e]), bf2f((bf16_t)of[e]), dlt);
  }
  dlt += __shfl_xor(dlt, 32, 64);
  if (qok && h == 0) {
    float* ws = const_cast<float*>(p.delta);
    const long nrows = (long)p.B * p.nq * p.Sq;
    ws[srow] = -dlt;
    ws[nrows + srow] = (inv > 0.f) ? -(m - __logf(inv)) / p.scale : NEG_INF;
  }
  const float c2 = p.scale * LOG2E, mc = m * LOG2E;
  const float invs = inv * p.scale;    // fold the 1/sqrt(d) of dS into the normaliser

  f32x16 dq[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt) dq[dt] = zero16();

  int kend = p.Sk;
  if (CAUSAL) kend = min(p.Sk, (bm.x + 1) * (32 * NW));
  int ntiles = (kend + KT - 1) / KT;
  const bf16_t* kb = p.k + (long)b * p.Sk * p.ldk + (long)kvh * HD;
  const bf16_t* vb = p.v + (long)b * p.Sk * p.ldv + (long)kvh * HD;
  const uint8_t* km = p.kmask ? p.kmask + (long)b * p.Sk : nullptr;
  const bool dropping = (!CAUSAL) && p.drop_thr != 0;
  ur_rowkey rk = {0u, 0u};
  if (dropping) {
    // this lane's query row: its two dropout keys, also published for the dK/dV kernel (whose lanes own keys, not rows)
    rk = ur_attn_row_key(p.seed, p.drow0 + ((uint64_t)((long)b * p.nq + hq) * p.Sq + (uint64_t)q));
    if (qok && h == 0) {
      const long nrows = (long)p.B * p.nq * p.Sq;
      p.rowkeys[srow] = rk.k1;
      p.rowkeys[nrows + srow] = rk.k2;
    }
  }

  unsigned long long* kwords = reinterpret_cast<unsigned long long*>(smem + 4 * C::TILE);
  fill_key_words(kwords, km, ntiles, p.Sk, tid, NW * 64);
  int t_first = 0;
  if (km != nullptr) {
    __syncthreads();
    if (CAUSAL) t_first = first_valid_tile(kwords, ntiles);
    else ntiles = trim_masked_tail(kwords, ntiles);          // trailing key tiles without a valid key: dS = 0 exactly (see trim_masked_tail)
  }
  Loader<HD, NW * 64> ks, vs;
  ks.init(p.ldk, tid); vs.init(p.ldv, tid);
  if (t_first < ntiles) {
    char* first = smem + (t_first & 1) * 2 * C::TILE;
    ks.issue(first, kb, p.ldk, t_first * KT, p.Sk, tid);
    vs.issue(first + C::TILE, vb, p.ldv, t_first * KT, p.Sk, tid);
    ks.commit(first, tid);
    vs.commit(first + C::TILE, tid);
  }
  __syncthreads();

  for (int t = t_first; t < ntiles; ++t) {
    const int k0 = t * KT;
    const char* ktile = smem + (t & 1) * 2 * C::TILE;
    const char* vtile = ktile + C::TILE;
    char* nk = smem + ((t + 1) & 1) * 2 * C::TILE;
    if (t + 1 < ntiles) {
      ks.issue(nk, kb, p.ldk, k0 + KT, p.Sk, tid);
      vs.issue(nk + C::TILE, vb, p.ldv, k0 + KT, p.Sk, tid);
    }
    KeyBits kbits; kbits.valid = kwords[2 * t]; kbits.inr = kwords[2 * t + 1];
    if (qblk < p.Sq) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const int kbase = k0 + 32 * sub;
        if (kbase >= kend || (CAUSAL && kbase > qblk + 31)) break;
        if (CAUSAL && (uint32_t)(kbits.valid >> (32 * sub)) == 0u) continue;      // no valid key: dS = 0
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int st = 0; st < C::NS; ++st) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(ktile, 32 * sub, st, lane), qf[st], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(vtile, 32 * sub, st, lane), dof[st], dp, 0, 0, 0);
        }
        const uint32_t v32 = (uint32_t)(kbits.valid >> (32 * sub)), i32 = (uint32_t)(kbits.inr >> (32 * sub));
        const bool fast = (v32 == 0xffffffffu);          // incl. the causal diagonal (one compare per element) and dropout (keep factors on dP)
        if (dropping) {
          // dS = P (keep / (1 - p) * dP - delta): the keep factors of this lane's 16 keys (two pair words per group of four)
          const uint32_t kp0 = (uint32_t)(kbase + 4 * h) >> 1;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint32_t w0 = ur_attn_pair_word(rk.k1, rk.k2, kp0 + 4 * g), w1 = ur_attn_pair_word(rk.k1, rk.k2, kp0 + 4 * g + 1);
            dp[4 * g] *= ur_attn_keep_scale(w0, 0u, p.drop_thr, p.drop_inv);
            dp[4 * g + 1] *= ur_attn_keep_scale(w0, 1u, p.drop_thr, p.drop_inv);
            dp[4 * g + 2] *= ur_attn_keep_scale(w1, 0u, p.drop_thr, p.drop_inv);
            dp[4 * g + 3] *= ur_attn_keep_scale(w1, 1u, p.drop_thr, p.drop_inv);
          }
        }
        if (fast) {
          if (CAUSAL && kbase + 31 > qblk) {
            const int dqk = opaque(q - kbase - 4 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = ((r & 3) + 8 * (r >> 2) <= dqk) ? s[r] : NEG_INF;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = fast_exp2(fmaf(s[r], c2, -mc)) * invs * (dp[r] - dlt);
        } else {
          const uint32_t vh = opaque(v32 >> (4 * h)), ih = opaque(i32 >> (4 * h));
          const int qq = opaque(q), kb0 = opaque(kbase + 4 * h);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int bit = (r & 3) + 8 * (r >> 2);
            const int kr = kb0 + bit;
            const float sc = mask_score<CAUSAL>(s[r], p.scale, (vh >> bit) & 1u, (ih >> bit) & 1u, kr, qq);
            const float pr = (sc == NEG_INF) ? 0.f : fast_exp2((sc - m) * LOG2E) * invs;
            s[r] = pr * (dp[r] - dlt);
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 df = acc_frag(s, s2);
          bf16x8 kt[C::NDT];
          tr_frags<HD>(kt, ktile, 32 * sub + 16 * s2, lane);
#pragma unroll
          for (int dt = 0; dt < C::NDT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[dt], df, dq[dt], 0, 0, 0);
        }
      }
    }
    if (t + 1 < ntiles) {
      ks.commit(nk, tid);
      vs.commit(nk + C::TILE, tid);
    }
    __syncthreads();
  }
  dq_store_block<HD>(p, dq, b, hq, q, qok, lane);
}

// ================================================================================================
// dK/dV: one 32-key block per wave (lane = key column).  Per 32-query sub-tile:
//   S[q][key] = Q K^T, dP[q][key] = dO V^T (A = Q / dO rows from LDS, B = K / V fragments in registers)
//   dV^T[d][key] += dO^T[d][q] * (P.drop)[q][key];   dK^T[d][key] += Q^T[d][q] * dS[q][key]
// LDS per buffer: Q tile | dO tile | row stats ((m, 1/l) pairs [64][2], delta [64], the rows' dropout keys k1 [64], k2 [64]).
template <int HD, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dkv_kernel(AttnP p) {
  using C = Cfg<HD>;
  constexpr int STG = 2 * C::TILE + 5 * KT * (int)sizeof(float);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const BlockMap bm = block_map<false>((p.Sk + 32 * NW - 1) / (32 * NW), 1, p.nkv, p.B);
  const int kvh = bm.head, b = bm.b;
  const int kblk = bm.x * (32 * NW) + wave * 32;
  const int key = kblk + (lane & 31);
  const bool kok = key < p.Sk;
  const long ktok = (long)b * p.Sk + (kok ? key : 0);

  bf16x8 kf[C::NS], vf[C::NS];
#pragma unroll
  for (int s = 0; s < C::NS; ++s) {
    kf[s] = g_frag(p.k + ktok * p.ldk + (long)kvh * HD, s, lane, kok);
    vf[s] = g_frag(p.v + ktok * p.ldv + (long)kvh * HD, s, lane, kok);
  }
  const bool kvalid = kok && (p.kmask == nullptr || p.kmask[(long)b * p.Sk + key] != 0);
  const bool all_valid = __all(kvalid);
  const float c2 = p.scale * LOG2E;
  const bool dropping = (!CAUSAL) && p.drop_thr != 0;

  f32x16 dk[C::NDT], dv[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt) { dk[dt] = zero16(); dv[dt] = zero16(); }

  // causal (SDPA) semantics: a key block without a single valid key (the left padding of a prompt: the blocks with
  // the MOST query tiles to sweep) has P = 0 everywhere: dK = dV = 0 without reading Q or dO
  if (CAUSAL && p.kmask != nullptr) {
    if (!__syncthreads_or(kvalid ? 1 : 0)) {
      store_T<HD>(p.dk + ktok * p.lddk + (long)kvh * HD, dk, 0.f, lane, kok);
      store_T<HD>(p.dv + ktok * p.lddv + (long)kvh * HD, dv, 0.f, lane, kok);
      return;
    }
  }

  const int qstart = CAUSAL ? ((bm.x * (32 * NW)) / KT) * KT : 0;
  const int ntq = (p.Sq - qstart + KT - 1) / KT;
  const int ntot = ntq * p.rep;                    // tiles over (query head of the group, query tile)

  Loader<HD, NW * 64> qs, dos;
  qs.init(p.ldq, tid); dos.init(p.lddo, tid);
  auto tile_ptrs = [&](int it, const bf16_t*& qb, const bf16_t*& dob, long& sbase, int& q0) {
    const int hr = it / ntq, tq = it - hr * ntq;
    const int hq = kvh * p.rep + hr;
    qb = p.q + (long)b * p.Sq * p.ldq + (long)hq * HD;
    dob = p.dout + (long)b * p.Sq * p.lddo + (long)hq * HD;
    sbase = ((long)b * p.nq + hq) * p.Sq;
    q0 = qstart + tq * KT;
  };
  // row stats travel by LDS-DMA too (dword per lane): [128 floats (m, 1/l) pairs][64 floats delta];
  // rows past Sq are clamped copies and are masked by position in the general path
  auto load_tile = [&](int it, char* buf) {
    const bf16_t* qb; const bf16_t* dob; long sbase; int q0;
    tile_ptrs(it, qb, dob, sbase, q0);
    qs.issue(buf, qb, p.ldq, q0, p.Sq, tid);
    dos.issue(buf + C::TILE, dob, p.lddo, q0, p.Sq, tid);
    if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
      typedef __attribute__((address_space(3))) void lds_void;
      typedef const __attribute__((address_space(1))) void gbl_void;
      char* fb = buf + 2 * C::TILE;
      const int r0 = min(q0 + (lane >> 1), p.Sq - 1), r1 = min(q0 + 32 + (lane >> 1), p.Sq - 1), rd = min(q0 + lane, p.Sq - 1);
      __builtin_amdgcn_global_load_lds((gbl_void*)(p.stats + (sbase + r0) * 2 + (lane & 1)), (lds_void*)fb, 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)(p.stats + (sbase + r1) * 2 + (lane & 1)), (lds_void*)(fb + 256), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)(p.delta + sbase + rd), (lds_void*)(fb + 512), 4, 0, 0);
      if (dropping) {      // (uniform) the rows' dropout keys, published by the dQ kernel of this call
        const long nrows = (long)p.B * p.nq * p.Sq;
        __builtin_amdgcn_global_load_lds((gbl_void*)(p.rowkeys + sbase + rd), (lds_void*)(fb + 768), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(p.rowkeys + nrows + sbase + rd), (lds_void*)(fb + 1024), 4, 0, 0);
      }
    }
  };
  auto store_tile = [&](char* buf) {
    qs.commit(buf, tid);
    dos.commit(buf + C::TILE, tid);
  };

  if (ntot > 0) { load_tile(0, smem); store_tile(smem); }
  __syncthreads();

  for (int it = 0; it < ntot; ++it) {
    const char* qtile = smem + (it & 1) * STG;
    const char* dotile = qtile + C::TILE;
    const float* fst = reinterpret_cast<const float*>(qtile + 2 * C::TILE);
    if (it + 1 < ntot) load_tile(it + 1, smem + ((it + 1) & 1) * STG);
    const bf16_t* qb_; const bf16_t* dob_; long sbase_; int q0;
    tile_ptrs(it, qb_, dob_, sbase_, q0);
    [[maybe_unused]] const int hq = kvh * p.rep + it / ntq;
    // Interior tile (all keys valid, both 32-query sub-tiles in range and past the causal diagonal, no
    // dropout): software-pipelined -- sub-tile 1's S/dP MFMAs are independent of sub-tile 0's softmax
    // VALU and sub-tile 0's dV/dK MFMAs of sub-tile 1's softmax, so one wave keeps both pipes busy.
    const bool tile_fast = all_valid && !dropping && (q0 + KT <= p.Sq) && (!CAUSAL || q0 >= kblk + 31);
    if (kblk < p.Sk && tile_fast) {
      f32x16 s0 = zero16(), dp0 = zero16(), s1 = zero16(), dp1 = zero16();
      // phase 1: every row fragment of both sub-tiles goes in flight before the first MFMA (one wave per
      // SIMD: nobody else hides the LDS latency, so a read-wait-MFMA chain would stall the matrix pipe)
      bf16x8 qa0[C::NS], da0[C::NS], qa1[C::NS], da1[C::NS];
#pragma unroll
      for (int st = 0; st < C::NS; ++st) { qa0[st] = row_frag<HD>(qtile, 0, st, lane); da0[st] = row_frag<HD>(dotile, 0, st, lane); }
#pragma unroll
      for (int st = 0; st < C::NS; ++st) { qa1[st] = row_frag<HD>(qtile, 32, st, lane); da1[st] = row_frag<HD>(dotile, 32, st, lane); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int st = 0; st < C::NS; ++st) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa0[st], kf[st], s0, 0, 0, 0);
        dp0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da0[st], vf[st], dp0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // transposed fragments for sub-tile 0's dV/dK products: in flight under sub-tile 1's S/dP MFMAs
      bf16x8 tdo0[2][C::NDT], tq0[2][C::NDT];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) { tr_frags<HD>(tdo0[s2], dotile, 16 * s2, lane); tr_frags<HD>(tq0[s2], qtile, 16 * s2, lane); }
#pragma unroll
      for (int st = 0; st < C::NS; ++st) {
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa1[st], kf[st], s1, 0, 0, 0);
        dp1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da1[st], vf[st], dp1, 0, 0, 0);
      }
      auto soft = [&](f32x16& sv, f32x16& dpv, int sub) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int qr = 32 * sub + 8 * rq + 4 * h;
          const float4 a = *reinterpret_cast<const float4*>(fst + 2 * qr);          // (m, 1/l) of rows qr, qr+1
          const float4 bq = *reinterpret_cast<const float4*>(fst + 2 * qr + 4);     // rows qr+2, qr+3
          const float4 cq = *reinterpret_cast<const float4*>(fst + 2 * KT + qr);    // delta of rows qr..qr+3
          const float ma[4] = {a.x, a.z, bq.x, bq.z}, iv[4] = {a.y, a.w, bq.y, bq.w}, dl[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            const float ps = fast_exp2(fmaf(sv[r], c2, -ma[e] * LOG2E)) * iv[e];      // p (dS scale applied at the store)
            sv[r] = ps;
            dpv[r] = ps * (dpv[r] + dl[e]);        // dl = -delta
          }
        }
      };
      soft(s0, dp0, 0);                 // VALU, independent of the s1/dp1 MFMAs above
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 tdo1[2][C::NDT], tq1[2][C::NDT];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) { tr_frags<HD>(tdo1[s2], dotile, 32 + 16 * s2, lane); tr_frags<HD>(tq1[s2], qtile, 32 + 16 * s2, lane); }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_frag(s0, s2), df = acc_frag(dp0, s2);
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tdo0[s2][dt], pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq0[s2][dt], df, dk[dt], 0, 0, 0);
        }
      }
      soft(s1, dp1, 1);                 // VALU under sub-tile 0's dV/dK MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_frag(s1, s2), df = acc_frag(dp1, s2);
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tdo1[s2][dt], pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq1[s2][dt], df, dk[dt], 0, 0, 0);
        }
      }
    } else
    if (kblk < p.Sk) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const int qbase = q0 + 32 * sub;
        if (qbase >= p.Sq) break;
        if (CAUSAL && qbase + 31 < kblk) continue;        // every query of this sub-tile precedes every key
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int st = 0; st < C::NS; ++st) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(qtile, 32 * sub, st, lane), kf[st], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(dotile, 32 * sub, st, lane), vf[st], dp, 0, 0, 0);
        }
        const bool fast = all_valid && (!CAUSAL || qbase >= kblk + 31) && !dropping && (qbase + 32 <= p.Sq);   // rows past Sq are clamped copies
        // row stats (m, scale/l, delta) of this lane's query rows: 4 consecutive rows per 16-byte LDS read,
        // consumed group by group so only 12 of them are live at a time
        if (fast) {
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int qr = 32 * sub + 8 * rq + 4 * h;
            const float4 a = *reinterpret_cast<const float4*>(fst + 2 * qr);
            const float4 bq = *reinterpret_cast<const float4*>(fst + 2 * qr + 4);
            const float4 cq = *reinterpret_cast<const float4*>(fst + 2 * KT + qr);
            const float ma[4] = {a.x, a.z, bq.x, bq.z}, iv[4] = {a.y, a.w, bq.y, bq.w}, dl[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * rq + e;
              const float ps = fast_exp2(fmaf(s[r], c2, -ma[e] * LOG2E)) * iv[e];      // p * scale
              s[r] = ps;
              dp[r] = ps * (dp[r] + dl[e]);          // dl = -delta
            }
          }
        } else {
          const int qb0 = opaque(qbase + 4 * h), keyo = opaque(key);
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int qr = 32 * sub + 8 * rq + 4 * h;
            const float4 a = *reinterpret_cast<const float4*>(fst + 2 * qr);
            const float4 bq = *reinterpret_cast<const float4*>(fst + 2 * qr + 4);
            const float4 cq = *reinterpret_cast<const float4*>(fst + 2 * KT + qr);
            const float ma[4] = {a.x, a.z, bq.x, bq.z}, iv[4] = {a.y, a.w, bq.y, bq.w}, dl[4] = {cq.x, cq.y, cq.z, cq.w};
            uint4 k1q = make_uint4(0, 0, 0, 0), k2q = make_uint4(0, 0, 0, 0);          // dropout keys of rows qr .. qr + 3
            if (dropping) {
              k1q = *reinterpret_cast<const uint4*>(fst + 3 * KT + qr);
              k2q = *reinterpret_cast<const uint4*>(fst + 4 * KT + qr);
            }
            const uint32_t k1[4] = {k1q.x, k1q.y, k1q.z, k1q.w}, k2[4] = {k2q.x, k2q.y, k2q.z, k2q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * rq + e;
              const int qpos = qb0 + e + 8 * rq;
              const float sc = mask_score<CAUSAL>(s[r], p.scale, kvalid, kok, keyo, qpos);       // natural-log domain
              const float ps = (sc == NEG_INF || qpos >= p.Sq) ? 0.f : fast_exp2((sc - ma[e]) * LOG2E) * iv[e];
              float g = dp[r], pd = ps;
              if (dropping) {
                const float dsc = ur_attn_keep_scale(ur_attn_pair_word(k1[e], k2[e], (uint32_t)keyo >> 1), (uint32_t)keyo, p.drop_thr, p.drop_inv);
                g *= dsc; pd *= dsc;
              }
              s[r] = pd;
              dp[r] = ps * (g + dl[e]);
            }
          }
        }
        // s = P*scale (with dropout) -> dV needs P: undo the scale on the dV side with one multiply per output
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_frag(s, s2), df = acc_frag(dp, s2);
          bf16x8 tdo[C::NDT], tq[C::NDT];
          tr_frags<HD>(tdo, dotile, 32 * sub + 16 * s2, lane);
          tr_frags<HD>(tq, qtile, 32 * sub + 16 * s2, lane);
#pragma unroll
          for (int dt = 0; dt < C::NDT; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tdo[dt], pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[dt], df, dk[dt], 0, 0, 0);
          }
        }
      }
    }
    if (it + 1 < ntot) store_tile(smem + ((it + 1) & 1) * STG);
    __syncthreads();
  }
  store_T<HD>(p.dk + ktok * p.lddk + (long)kvh * HD, dk, p.scale, lane, kok);     // dS was kept unscaled
  store_T<HD>(p.dv + ktok * p.lddv + (long)kvh * HD, dv, 1.0f, lane, kok);
}

// ================================================================================================
// dK/dV for FEW queries against MANY keys: the user Q-Former's cross-attention (models/qformer.py:169-275 called from
// training/user_qformer_training.py:47-68 -- 64 learned queries, T = hist*Q_item = 1600 keys, B*heads = 8192 pairs).
// attn_bwd_dkv_kernel gives every 128-key workgroup its own copy of the (single) Q / dO tile and runs at one wave per
// SIMD: 106 496 workgroups that each load, wait, do 32 MFMAs per wave and leave -- latency end to end (4.1 ms per
// layer at C3 for 6.8 GB of K/V/dK/dV traffic).  Here a workgroup keeps the Q / dO tile and the row constants of ONE
// (batch, head) pair in LDS and its 4 waves walk that pair's 32-key blocks (wave w takes blocks w, w+4, ...), two waves
// per SIMD covering each other's K / V load latency (an explicit register prefetch of the next block measured 3 % slower:
// 256 VGPRs + spills).  No barrier after the prologue.  Same arithmetic in the same order as
// attn_bwd_dkv_kernel -> bit-identical results (tests/test_gpu_attention.py).  Non-causal, rep == 1, Sq <= 64.
// Sum over the 32 lanes of a lane half of each of 32 registers, transposed: lane j (of its half) returns the sum of register j.
// Five halving steps; at a step the lane keeps the register half its bit selects and adds the partner's copy of it: lanes 16 apart
// by v_permlane16_swap, then row_mirror (j ^ 15), row_half_mirror (j ^ 7), quad reversal (j ^ 3), quad swap (j ^ 1) -- each partner has
// the selecting bit flipped, and together the masks generate every lane of the row.  31 exchanges + adds instead of 32 x 5 shuffles.
template <int CTRL>
__device__ __forceinline__ float dppx_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lanes_transpose_sum32(const float (&v)[32], int lane) {
  float a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[r]), __float_as_uint(v[r + 16]), false, false);
    a[r] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // rows 0 / 2: register r over lanes j, j + 16; rows 1 / 3: register r + 16
  }
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
  float b[8], c[4], d[2];
#pragma unroll
  for (int r = 0; r < 8; ++r) b[r] = (b3 ? a[r + 8] : a[r]) + dppx_f32<0x140>(b3 ? a[r] : a[r + 8]);        // row_mirror
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = (b2 ? b[r + 4] : b[r]) + dppx_f32<0x141>(b2 ? b[r] : b[r + 4]);        // row_half_mirror
#pragma unroll
  for (int r = 0; r < 2; ++r) d[r] = (b1 ? c[r + 2] : c[r]) + dppx_f32<0x1B>(b1 ? c[r] : c[r + 2]);         // quad_perm [3, 2, 1, 0]
  return (b0 ? d[1] : d[0]) + dppx_f32<0xB1>(b0 ? d[0] : d[1]);                                              // quad_perm [1, 0, 3, 2]
}

#ifndef UR_FEWQ_PREFETCH
#define UR_FEWQ_PREFETCH 0      // lab: 2 = the next live key block's K / V fragments are requested in the MIDDLE of the current block, into the registers the
                                // last S / dP product has just released -- hipcc keeps both sets live instead (256 VGPRs + 96 B of scratch): 2.28 -> 2.83 ms
                                // for dQ + dK/dV of a C3 layer; round 3's whole-block register prefetch likewise (256 VGPRs + 5 spills, 3 % slower).  0 = at the top
#endif
#ifndef UR_FEWQ_ABLATE
#define UR_FEWQ_ABLATE 0        // lab builds only (WRONG results): 1 = no dK / dV stores, 2 = no K / V loads (zero fragments)
#endif
#ifndef UR_FEWQ_WAVES
#define UR_FEWQ_WAVES 2         // lab: waves per SIMD the kernel is compiled for
#endif
template <int HD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(UR_FEWQ_WAVES, UR_FEWQ_WAVES))) void attn_bwd_dkv_fewq_kernel(AttnP p, int nchunk, int bpc) {
  using C = Cfg<HD>;
  static_assert(HD == 64, "register-staged Q / dO tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int chunk = blockIdx.x % nchunk, pair = blockIdx.x / nchunk;
  const int hq = pair % p.nq, b = pair / p.nq;
  char* qtile = smem;
  char* dotile = smem + C::TILE;
  float* fst = reinterpret_cast<float*>(smem + 2 * C::TILE);      // [128] (m, 1/l) pairs, [64] -delta, [64] + [64] the rows' dropout keys
  int* flag = reinterpret_cast<int*>(fst + 5 * KT);               // [0]: the sample has a valid key
  const long sbase = ((long)b * p.nq + hq) * p.Sq;
  const bool dropping = p.drop_thr != 0;
  const uint8_t* kmrow = p.kmask ? p.kmask + (long)b * p.Sk : nullptr;
  {
    Loader<HD, 256> qs, dos;
    qs.issue(qtile, p.q + (long)b * p.Sq * p.ldq + (long)hq * HD, p.ldq, 0, p.Sq, tid);
    dos.issue(dotile, p.dout + (long)b * p.Sq * p.lddo + (long)hq * HD, p.lddo, 0, p.Sq, tid);
    if (tid == 0) flag[0] = 0;
    if (tid < 2 * KT) fst[tid] = p.stats[(sbase + min(tid >> 1, p.Sq - 1)) * 2 + (tid & 1)];      // rows past Sq: clamped copies,
    else if (tid < 3 * KT) fst[tid] = p.delta[sbase + min(tid - 2 * KT, p.Sq - 1)];             // masked by position below
    if (dropping && tid < 2 * KT) {
      const long nrows = (long)p.B * p.nq * p.Sq;
      reinterpret_cast<uint32_t*>(fst)[3 * KT + tid] = p.rowkeys[(tid >> 6) * nrows + sbase + min(tid & 63, p.Sq - 1)];
    }
    qs.commit(qtile, tid);
    dos.commit(dotile, tid);
  }
  __syncthreads();
  // which 32-key blocks hold a valid key, and does the sample have one at all?  (then blocks without one have dK = dV = 0 exactly:
  // trim_masked_tail's argument.)  One byte per block in LDS: the loop decides without a mask load in its way.
  unsigned char* live = reinterpret_cast<unsigned char*>(flag + 4 + 4 * 2 * 64);          // [<= 256 blocks], behind the waves' column sums
  if (kmrow != nullptr) {
    int any = 0;
    for (int j = tid; j < ((p.Sk + 31) >> 5); j += 256) {
      int lv = 0;
      const int i0 = 32 * j;
      if (i0 + 32 <= p.Sk && ((reinterpret_cast<uintptr_t>(kmrow + i0) & 15) == 0)) {
        const uint4 w0 = *reinterpret_cast<const uint4*>(kmrow + i0), w1 = *reinterpret_cast<const uint4*>(kmrow + i0 + 16);
        lv = (w0.x | w0.y | w0.z | w0.w | w1.x | w1.y | w1.z | w1.w) != 0u;
      } else {
        for (int i = i0; i < min(i0 + 32, p.Sk); ++i) lv |= kmrow[i] != 0;
      }
      live[j] = (unsigned char)lv;
      any |= lv;
    }
    if (any) flag[0] = 1;
    __syncthreads();
  }
  const bool sample_has_key = kmrow != nullptr && flag[0] != 0;

  const int nblk = (p.Sk + 31) >> 5;
  const int blk_hi = min(nblk, (chunk + 1) * bpc);
  const float c2 = p.scale * LOG2E;
  const bf16_t* kbase = p.k + (long)b * p.Sk * p.ldk + (long)hq * HD;
  const bf16_t* vbase = p.v + (long)b * p.Sk * p.ldv + (long)hq * HD;

  auto load_kv = [&](int blk, bf16x8 (&kf)[C::NS], bf16x8 (&vf)[C::NS], uint32_t& state) {
    const int key = blk * 32 + (lane & 31);
    const bool kok = blk < blk_hi && key < p.Sk;
    const long krow = kok ? key : 0;
#pragma unroll
    for (int st = 0; st < C::NS; ++st) {
#if UR_FEWQ_ABLATE == 2
      kf[st] = g_frag(kbase + krow * p.ldk, st, lane, false);
      vf[st] = g_frag(vbase + krow * p.ldv, st, lane, false);
#else
      kf[st] = g_frag(kbase + krow * p.ldk, st, lane, kok);
      vf[st] = g_frag(vbase + krow * p.ldv, st, lane, kok);
#endif
    }
    uint32_t m = 1;
    if (kok && kmrow) m = kmrow[key];
    state = (kok ? 1u : 0u) | ((kok && m != 0) ? 2u : 0u);          // bit 0: inside Sk, bit 1: allowed by the key mask
  };

  bf16x8 kf[C::NS], vf[C::NS];
  uint32_t state = 0;
  float csk = 0.f, csv = 0.f;          // this lane's feature of the dK / dV column sums over the wave's key blocks
  int blk = chunk * bpc + wave;
  bool have = false;                   // kf / vf / state already hold block `blk` (requested in the middle of the previous block)
#if UR_FEWQ_PREFETCH == 2
  const int last_sub = p.Sq > 32 ? 1 : 0;
#endif
  for (; blk < blk_hi; blk += 4) {
    if (sample_has_key && live[blk] == 0) {
      // a key block without one valid key while the sample has some: every probability is exactly 0 -> dK = dV = 0, nothing is read
      const int key0 = blk * 32 + (lane & 31);
      const bool kok0 = key0 < p.Sk;
      const long ktok0 = (long)b * p.Sk + (kok0 ? key0 : 0);
      f32x16 z[C::NDT];
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt) z[dt] = zero16();
      store_T<HD>(p.dk + ktok0 * p.lddk + (long)hq * HD, z, 0.f, lane, kok0);
      store_T<HD>(p.dv + ktok0 * p.lddv + (long)hq * HD, z, 0.f, lane, kok0);
      continue;
    }
    if (!have) load_kv(blk, kf, vf, state);
    const int kblk = blk * 32, key = kblk + (lane & 31);
    const bool kok = (state & 1u) != 0, kvalid = (state & 2u) != 0;
    const bool all_valid = __all(kvalid);
    f32x16 dk[C::NDT], dv[C::NDT];
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) { dk[dt] = zero16(); dv[dt] = zero16(); }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int qbase = 32 * sub;
      if (qbase >= p.Sq) break;
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int st = 0; st < C::NS; ++st) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(qtile, 32 * sub, st, lane), kf[st], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(dotile, 32 * sub, st, lane), vf[st], dp, 0, 0, 0);
      }
#if UR_FEWQ_PREFETCH == 2
      if (sub == last_sub) {
        // the K / V fragments have had their last use: the wave's NEXT live block is requested into the same registers here and lands
        // under this block's softmax, dV / dK products and stores (the kernel is bound by this per-block chain, not by bytes)
        const int nb = blk + 4;
        have = nb < blk_hi && !(sample_has_key && live[nb] == 0);
        if (have) load_kv(nb, kf, vf, state);
      }
#endif
      const bool fast = all_valid && !dropping && (qbase + 32 <= p.Sq);
      if (fast) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int qr = 32 * sub + 8 * rq + 4 * h;
          const float4 a = *reinterpret_cast<const float4*>(fst + 2 * qr);
          const float4 bq = *reinterpret_cast<const float4*>(fst + 2 * qr + 4);
          const float4 cq = *reinterpret_cast<const float4*>(fst + 2 * KT + qr);
          const float ma[4] = {a.x, a.z, bq.x, bq.z}, iv[4] = {a.y, a.w, bq.y, bq.w}, dl[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            const float ps = fast_exp2(fmaf(s[r], c2, -ma[e] * LOG2E)) * iv[e];
            s[r] = ps;
            dp[r] = ps * (dp[r] + dl[e]);          // dl = -delta
          }
        }
      } else {
        const int qb0 = opaque(qbase + 4 * h), keyo = opaque(key);
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int qr = 32 * sub + 8 * rq + 4 * h;
          const float4 a = *reinterpret_cast<const float4*>(fst + 2 * qr);
          const float4 bq = *reinterpret_cast<const float4*>(fst + 2 * qr + 4);
          const float4 cq = *reinterpret_cast<const float4*>(fst + 2 * KT + qr);
          const float ma[4] = {a.x, a.z, bq.x, bq.z}, iv[4] = {a.y, a.w, bq.y, bq.w}, dl[4] = {cq.x, cq.y, cq.z, cq.w};
          uint4 k1q = make_uint4(0, 0, 0, 0), k2q = make_uint4(0, 0, 0, 0);          // dropout keys of rows qr .. qr + 3
          if (dropping) {
            k1q = *reinterpret_cast<const uint4*>(fst + 3 * KT + qr);
            k2q = *reinterpret_cast<const uint4*>(fst + 4 * KT + qr);
          }
          const uint32_t k1[4] = {k1q.x, k1q.y, k1q.z, k1q.w}, k2[4] = {k2q.x, k2q.y, k2q.z, k2q.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            const int qpos = qb0 + e + 8 * rq;
            const float sc = mask_score<false>(s[r], p.scale, kvalid, kok, keyo, qpos);       // natural-log domain
            const float ps = (sc == NEG_INF || qpos >= p.Sq) ? 0.f : fast_exp2((sc - ma[e]) * LOG2E) * iv[e];
            float g = dp[r], pd = ps;
            if (dropping) {
              const float dsc = ur_attn_keep_scale(ur_attn_pair_word(k1[e], k2[e], (uint32_t)keyo >> 1), (uint32_t)keyo, p.drop_thr, p.drop_inv);
              g *= dsc; pd *= dsc;
            }
            s[r] = pd;
            dp[r] = ps * (g + dl[e]);
          }
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = acc_frag(s, s2), df = acc_frag(dp, s2);
        bf16x8 tdo[C::NDT], tq[C::NDT];
        tr_frags<HD>(tdo, dotile, 32 * sub + 16 * s2, lane);
        tr_frags<HD>(tq, qtile, 32 * sub + 16 * s2, lane);
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tdo[dt], pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[dt], df, dk[dt], 0, 0, 0);
        }
      }
    }
    if (p.colsum_part != nullptr) {
      // column sums of dK | dV over this block's 32 keys (the K | V projections' bias gradients, models/qformer.py:186-188: the caller no
      // longer re-reads the 13 GB of dK | dV of a C3 step for them): register 16 dt + i of the lane half = feature 32 dt + acc_row(i, h)
      float t[32];
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) t[16 * dt + i] = dk[dt][i];
      csk += lanes_transpose_sum32(t, lane);
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) t[16 * dt + i] = dv[dt][i];
      csv += lanes_transpose_sum32(t, lane);
    }
    const long ktok = (long)b * p.Sk + (kok ? key : 0);
#if UR_FEWQ_ABLATE == 1          // lab (timing only, results wrong): no dK / dV stores
    asm volatile("" :: "v"(dk[0][0]), "v"(dv[0][0]), "v"(dk[1][15]), "v"(dv[1][15]));
#else
    store_T<HD>(p.dk + ktok * p.lddk + (long)hq * HD, dk, p.scale, lane, kok);     // dS was kept unscaled
    store_T<HD>(p.dv + ktok * p.lddv + (long)hq * HD, dv, 1.0f, lane, kok);
#endif
  }
  if (p.colsum_part != nullptr) {
    // the four waves' sums -> one partial per (batch, head): [b][dK | dV][head][feature] (the launcher gives such calls ONE workgroup per pair)
    float* red = reinterpret_cast<float*>(flag + 4);          // [4 waves][2][64]
    red[(wave * 2 + 0) * 64 + lane] = csk;
    red[(wave * 2 + 1) * 64 + lane] = csv;
    __syncthreads();
    if (tid < 128) {
      const int t = tid >> 6, l = tid & 63, j = l & 31, hh = l >> 5;
      const float sum = red[(0 * 2 + t) * 64 + l] + red[(1 * 2 + t) * 64 + l] + red[(2 * 2 + t) * 64 + l] + red[(3 * 2 + t) * 64 + l];
      const int d = 32 * (j >> 4) + acc_row(j & 15, hh);
      p.colsum_part[(((long)b * 2 + t) * p.nq + hq) * HD + d] = (t == 0) ? sum * p.scale : sum;       // (dS was kept unscaled)
    }
  }
}

// out[i] = sum_b part[b][i]: the per-batch partial column sums of the few-query dK/dV kernel -> [dK sums | dV sums] (fixed order: reproducible).
// 64 columns x 4 batch quarters per workgroup (one thread per column walking all B partials serially took 60 us for B = 512, n = 2048)
__global__ __launch_bounds__(256) void kv_colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int n) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  const int b0 = (int)((long)B * s / 4), b1 = (int)((long)B * (s + 1) / 4);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    int bq = b0;
    for (; bq + 4 <= b1; bq += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += part[(long)(bq + u) * n + i];
    }
    for (; bq < b1; ++bq) acc[0] += part[(long)bq * n + i];
  }
  red[s][c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (s == 0 && i < n) out[i] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ================================================================================================
// dK/dV for head_dim 128, 4 waves (the Qwen3 shape): same decomposition as attn_bwd_dkv_kernel, one wave per
// SIMD with the whole register file, but built for instruction ISSUE, which bounded the first version
// (10 vector instructions per MFMA, 175 accumulator moves per tile):
//   * interior tiles load -LSE/scale and -delta (published by the dQ kernel) as the INITIAL accumulators of the S and
//     dP chains: P = exp2(c * S'), dS = P * dP' -- three vector instructions per element, no row constants
//     held in registers;
//   * lane-constant LDS offsets (row fragments, transposed fragments) are computed once per workgroup;
//   * four hard-fenced phases per 64-query tile: [S,dP](a) | [S,dP](b) + softmax(a) | [dV,dK](a) + softmax(b)
//     | [dV,dK](b): every phase pairs 16 MFMAs with independent vector work of the other half.
// LDS per buffer: Q tile | dO tile | ns[64] | nd[64] | (m, 1/l)[64] (general path only).
// Split transposed reads for the one-wave-per-SIMD dK/dV kernel: tr_issue4 starts the 8 reads of one batch
// (4 fragments), tr_landed<N> waits until at most N younger LDS reads are outstanding and ties the batch's
// registers to that wait.  Between the two the registers hold no data yet: this is only sound while the
// allocator leaves them alone (it parked them in AGPRs at 426 registers; at <= 380 it does not -- the kernel's
// parity tests in tests/test_gpu_attention.py are what catches a build where it does).

// ---- hand-ordered LDS streams of the dK/dV fast path (one wave per SIMD: nobody else hides an LDS latency, so every read
// is issued up to 14 LDS operations ahead of its use and every use waits with a COUNTED lgkmcnt for exactly its own data;
// the asm forms are invisible to hipcc's waitcnt pass, which would otherwise drain the in-flight LDS-DMA of the next tile
// in front of the first LDS load that follows it).  Between issue and landed the destination registers hold no data yet.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int OQ, int ODO>
__device__ __forceinline__ void rf_issue(bf16x8& qa, bf16x8& da, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(qa), "=&v"(da) : "v"(addr), "n"(OQ), "n"(ODO));
}
template <int N>
__device__ __forceinline__ void rf_landed(bf16x8& qa, bf16x8& da) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(qa), "+v"(da) : "n"(N));
}
template <int OFF>
__device__ __forceinline__ void tr_issue1(bf16x4& lo, bf16x4& hi, uint32_t a, uint32_t b) {
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void tr_landed1(bf16x4& lo, bf16x4& hi) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(N));
}

// single-instruction f32 multiply: hipcc's SLP vectoriser packs neighbouring f32 multiplies of the softmax into v_pk_mul_f32,
// which does not hide in an MFMA gap at one wave per SIMD (tools/lab/mfma_gap_lab.hip: +17 cycles per MFMA)
__device__ __forceinline__ float mul1(float a, float b) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// pin: the MFMA intrinsics are pure values, which hipcc is free to sink past sched_barriers (it bunched 12 of stream 2's
// MFMAs behind the softmax fillers); an empty volatile asm that "rewrites" the accumulator just produced keeps every
// MFMA between the two volatile asm statements (landed / next issue) it was written between
__device__ __forceinline__ void pin(f32x16& a) { asm volatile("" : "+v"(a)); }
// ... and a converted bf16 fragment: materialised HERE (hipcc otherwise sinks the conversions to just in front of the asm
// MFMA that reads them, where nothing pads the vector-write -> MFMA-read hazard: the first MFMA of a batch read stale registers)
__device__ __forceinline__ void pin(bf16x8& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin(f32x16& a, f32x16& b) { asm volatile("" : "+v"(a), "+v"(b)); }

// MFMA with the accumulator held in AccVGPRs for the whole kernel (inline asm: under -amdgpu-mfma-vgpr-form hipcc keeps every
// accumulator in arch VGPRs and, with dK^T / dV^T (128 registers) + K / V fragments (64) + S / dP of two halves (64) alone at the
// 256-register limit, parked dK / dV in AccVGPRs anyway and copied 16 registers in and 16 out around MFMAs).  Volatile: keeps its
// place between the counted LDS waits.  The leading s_nop (two wait states) covers what hipcc would put between a vector write of an
// operand (a landed LDS fragment, a packed bf16 conversion) and the MFMA that reads it.
__device__ __forceinline__ void mfma_acc_nop(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// the same without the wait states: ONLY where both operands were written long before (LDS fragments behind a counted
// lgkmcnt, bf16 conversions at least one MFMA earlier) -- measured (tools/lab/mfma_gap_lab.hip): an s_nop between the
// lgkmcnt wait and the MFMA costs 4-8 cycles per MFMA at one wave per SIMD
__device__ __forceinline__ void mfma_acc(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

#ifndef UR_DKV2_REGSTAGE
#define UR_DKV2_REGSTAGE 0   // 1 (lab): register-staged next tile instead of LDS-DMA pieces -- 33 more live registers: 144 B of scratch, kernel 2x slower
#endif
#ifndef UR_DKV2_V2
#define UR_DKV2_V2 1         // 0 (lab): the four hard-fenced phases of round 1 instead of the three counted streams
#endif
#ifndef UR_DKV2_STAMPS
#define UR_DKV2_STAMPS 0     // lab builds only: lane 0 of wave 0 of the first 256 workgroups logs the cycle counter at 8 points of tiles 4..11 (ur_lab_attn_stamps)
#endif
#if UR_DKV2_STAMPS
__device__ long long g_attn_stamps[256 * 8 * 8];
#ifndef UR_STAMP_BLK0
#define UR_STAMP_BLK0 0      // first stamped workgroup (0: the launch's first wave of workgroups; 4096: mid-kernel, clocks settled)
#endif
#define UR_STAMP_ON (threadIdx.x == 0 && blockIdx.x >= UR_STAMP_BLK0 && blockIdx.x < UR_STAMP_BLK0 + 256 && it >= 4 && it < 12)
#define UR_ASTAMP(k) do { if (UR_STAMP_ON) g_attn_stamps[((blockIdx.x - UR_STAMP_BLK0) * 8 + (it - 4)) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define UR_ASTAMP(k) do { } while (0)
#endif
#ifndef UR_DKV2_ABLATE
#define UR_DKV2_ABLATE 0     // lab (tools/lab/dkv2_ablate.sh; timing only, results wrong): 1 no tile reload, 2 no softmax, 3 no dV/dK phases, 4 no S/dP phases, 5 diagonal tiles on the fast path; three-stream path: 6 no MFMAs, 7 no LDS reads, 8 no fillers, 9 no LDS-DMA
#endif
template <bool CAUSAL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dkv2_kernel(AttnP p) {
  constexpr int HD = 128, NW = 4;
  using C = Cfg<HD>;
  constexpr int STG = 2 * C::TILE + 4 * KT * (int)sizeof(float);
  // NB tile buffers, the LDS-DMA runs AH = NB - 1 tiles ahead.  Three buffers (tile t+2 issued at the top of tile t) measured
  // the same as two (stamps: the ~400 cycles at the barrier are arrival skew, not the DMA): two stay.
  constexpr int NB = 2, AH = NB - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const BlockMap bm = block_map<false>((p.Sk + 32 * NW - 1) / (32 * NW), 1, p.nkv, p.B);
  const int kvh = bm.head, b = bm.b;
  const int kblk = bm.x * (32 * NW) + wave * 32;
  const int key = kblk + (lane & 31);
  const bool kok = key < p.Sk;
  const long ktok = (long)b * p.Sk + (kok ? key : 0);

  bf16x8 kf[C::NS], vf[C::NS];
#pragma unroll
  for (int s = 0; s < C::NS; ++s) {
    kf[s] = g_frag(p.k + ktok * p.ldk + (long)kvh * HD, s, lane, kok);
    vf[s] = g_frag(p.v + ktok * p.ldv + (long)kvh * HD, s, lane, kok);
  }
  const float c2 = p.scale * LOG2E;
  const long nrows = (long)p.B * p.nq * p.Sq;

  f32x16 dk[C::NDT], dv[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt) { dk[dt] = zero16(); dv[dt] = zero16(); }

  const int qstart = CAUSAL ? ((bm.x * (32 * NW)) / KT) * KT : 0;
  const int ntq = (p.Sq - qstart + KT - 1) / KT;
  const int ntot = ntq * p.rep;                    // tiles over (query head of the group, query tile)

  Loader<HD, NW * 64> qs, dos;
  qs.init(p.ldq, tid); dos.init(p.lddo, tid);
  auto tile_ptrs2 = [&](int hr, int tq, const bf16_t*& qb, const bf16_t*& dob, long& sbase, int& q0) {
    const int hq = kvh * p.rep + hr;
    qb = p.q + (long)b * p.Sq * p.ldq + (long)hq * HD;
    dob = p.dout + (long)b * p.Sq * p.lddo + (long)hq * HD;
    sbase = ((long)b * p.nq + hq) * p.Sq;
    q0 = qstart + tq * KT;
  };
  auto tile_ptrs = [&](int it, const bf16_t*& qb, const bf16_t*& dob, long& sbase, int& q0) {
    const int hr = it / ntq, tq = it - hr * ntq;
    const int hq = kvh * p.rep + hr;
    qb = p.q + (long)b * p.Sq * p.ldq + (long)hq * HD;
    dob = p.dout + (long)b * p.Sq * p.lddo + (long)hq * HD;
    sbase = ((long)b * p.nq + hq) * p.Sq;
    q0 = qstart + tq * KT;
  };
  // row constants travel by LDS-DMA too (one dword per lane): ns[64] | nd[64] | (m, 1/l)[64]; rows past Sq are
  // clamped copies (masked by position in the general path; an interior tile has none)
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  // row constants: four 256-byte pieces, one per wave (ns | nd | (m, 1/l) of rows 0-31 | of rows 32-63)
  auto load_consts = [&](char* buf, long sbase, int q0) {
    char* fb = buf + 2 * C::TILE + 256 * wave_u;
    const int rd = min(q0 + lane, p.Sq - 1);
    const int rs = min(q0 + 32 * (wave_u & 1) + (lane >> 1), p.Sq - 1);
    const float* src = wave_u == 0 ? p.delta + nrows + sbase + rd : (wave_u == 1 ? p.delta + sbase + rd : p.stats + (sbase + rs) * 2 + (lane & 1));
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)fb, 4, 0, 0);
  };
  auto load_tile = [&](int it, char* buf) {
    const bf16_t* qb; const bf16_t* dob; long sbase; int q0;
    tile_ptrs(it, qb, dob, sbase, q0);
    qs.issue(buf, qb, p.ldq, q0, p.Sq, tid);
    dos.issue(buf + C::TILE, dob, p.lddo, q0, p.Sq, tid);
    load_consts(buf, sbase, q0);
  };

  // lane-constant LDS byte offsets (relative to a tile): row fragments of rows (lane&31) for the 8 k-steps, and the
  // transposed-read offsets of the 4 head_dim blocks; 32-row / 16-row steps are immediates (the swizzle repeats)
  uint32_t rfo[C::NS], tao[C::NDT], tbo[C::NDT];
#pragma unroll
  for (int st = 0; st < C::NS; ++st) rfo[st] = C::off(lane & 31, 2 * st + h);
  {
    const int g16 = (lane >> 4) & 1, i = lane & 15;
    const int row = 4 * h + (i >> 2), sub8 = 8 * (i & 1);
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) {
      const int ch = 4 * dt + 2 * g16 + ((i & 3) >> 1);
      tao[dt] = C::off(row, ch) + sub8;
      tbo[dt] = C::off(row + 8, ch) + sub8;
    }
  }

  if (ntot > 0) load_tile(0, smem);
  if (AH == 2 && ntot > 1) load_tile(1, smem + STG);
  // the key-state byte is read AFTER the first tile's LDS-DMA is on its way: the wait for it (vmcnt is in order) then covers
  // the K / V fragments, the mask byte and the first tile in ONE memory latency instead of two per workgroup
  const bool kvalid = kok && (p.kmask == nullptr || p.kmask[(long)b * p.Sk + key] != 0);
  const bool all_valid = __all(kvalid);
  // causal (SDPA) semantics: a key block without a single valid key (the left padding of a prompt: the blocks with
  // the MOST query tiles to sweep) has P = 0 everywhere: dK = dV = 0 without reading Q or dO (the barrier inside
  // __syncthreads_or drains this wave's LDS-DMA pieces before the workgroup gives its LDS back)
  if (CAUSAL && p.kmask != nullptr) {
    if (!__syncthreads_or(kvalid ? 1 : 0)) {
      store_T<HD>(p.dk + ktok * p.lddk + (long)kvh * HD, dk, 0.f, lane, kok);
      store_T<HD>(p.dv + ktok * p.lddv + (long)kvh * HD, dv, 0.f, lane, kok);
      return;
    }
  }
  __syncthreads();

  // (head of the group, query tile) of tile `it` and of the tile whose DMA it issues, advanced incrementally: the two
  // scalar divisions per tile were ~100 instructions of the loop top
  int hr_c = 0, tq_c = 0, hr_n = ntq > 0 ? AH / ntq : 0, tq_n = ntq > 0 ? AH % ntq : 0, buf_c = 0;
  for (int it = 0; it < ntot; ++it) {
    const char* qtile = smem + buf_c * STG;
    const char* dotile = qtile + C::TILE;
    const float* fst = reinterpret_cast<const float*>(qtile + 2 * C::TILE);
    // The next tile's LDS-DMA is issued only AFTER the last compiler-visible LDS load of this tile (the row
    // fragments and row constants of phases 1-2): hipcc puts s_waitcnt vmcnt(0) in front of the first plain LDS
    // load that follows an LDS-DMA (it cannot tell the two buffers apart), which would expose the whole DMA
    // latency on every tile.  Phases 3-4 only read LDS through the inline-asm transposed reads.
    const int q0 = qstart + tq_c * KT;
    bool tile_fast = all_valid && (q0 + KT <= p.Sq) && (!CAUSAL || q0 >= kblk + 31 || UR_DKV2_ABLATE == 5);
#if UR_DKV2_V2
    // the fast path issues the next tile's LDS-DMA itself, as whole-row pieces: the next tile must be full too.  Tiles on
    // the causal diagonal take the same three streams with the mask applied to P (two more vector instructions per
    // element) instead of the general path (position compares, row constants from LDS: ~2x a fast tile)
    const bf16_t* nqb; const bf16_t* ndob; long nsb; int nq0;
    // the tile whose DMA this iteration issues (none left: this tile again, into the idle buffer)
    if (it + AH < ntot) tile_ptrs2(hr_n, tq_n, nqb, ndob, nsb, nq0); else tile_ptrs2(hr_c, tq_c, nqb, ndob, nsb, nq0);
    const bool tile_diag = CAUSAL && q0 < kblk + 31;
    tile_fast = all_valid && (q0 + KT <= p.Sq) && (nq0 + KT <= p.Sq);
#endif
    UR_ASTAMP(0);
#if UR_DKV2_STAMPS
    if (UR_STAMP_ON) g_attn_stamps[((blockIdx.x - UR_STAMP_BLK0) * 8 + (it - 4)) * 8 + 1] = (long long)wall_clock64();   // 100 MHz constant clock
#endif
#if UR_DKV2_V2
    if (kblk < p.Sk && tile_fast) {
     auto fast_body = [&](auto DIAG_) {
      constexpr bool DIAG = decltype(DIAG_)::value;
      // ---------------- fast path, three counted streams (64 MFMAs per 64-query tile) ----------------
      const uint32_t bufb = lds_off(qtile);
      // causal diagonal: P[query row e of half x][key] = 0 where key > q0 + 32 x + acc_row(e, h)
      const int dkh = key - q0 - 4 * h;
      auto cmask = [&](float v, int e, int half) { return (dkh - 32 * half <= (e & 3) + 8 * (e >> 2)) ? v : 0.f; };
      // initial accumulators (plain LDS loads, BEFORE this tile issues any LDS-DMA): rows 8g + 4h + 0..3 of the 32-row
      // half -> registers 4g .. 4g+3
      auto init16 = [&](f32x16& a, const float* src) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *reinterpret_cast<const float4*>(src + 8 * g + 4 * h);
          a[4 * g] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
        }
      };
      f32x16 sa, dpa, sb, dpb;
      const float c2v = c2;       // one VGPR copy: mul1's operands are vector registers
      init16(sa, fst); init16(dpa, fst + 64); init16(sb, fst + 32); init16(dpb, fst + 96);
      __builtin_amdgcn_sched_barrier(0);
      // next tile: wave-uniform row bases; its 8 + 1 LDS-DMA pieces ride in the MFMA gaps of stream 1
      // (no next tile: the same tile is copied again into the idle buffer -- unconditional pieces keep stream 1 one
      // straight-line block; a ragged next tile never gets here, see tile_fast)
      char* nbuf = smem + ((buf_c + AH) % NB) * STG;
      // per-lane source pointers of piece 0, formed once per tile; a piece then costs one 64-bit add of a scalar step
      const char* qrow = reinterpret_cast<const char*>(nqb + (long)nq0 * p.ldq) + qs.voff;
      const char* dorow = reinterpret_cast<const char*>(ndob + (long)nq0 * p.lddo) + dos.voff;
      const long qstep = 32L * p.ldq, dostep = 32L * p.lddo;           // bytes per 16 rows
      char* nbw = nbuf + wave_u * 1024;
#if UR_DKV2_REGSTAGE
      // register-staged next tile: 8 global_load_dwordx4 + 1 dword per lane issued in stream 1's gaps (a few cycles of issue
      // each, against ~60 for an LDS-DMA piece), written to LDS behind stream 3.  Same lane <-> (row, chunk) map as the DMA:
      // the swizzle sits on the source address, the LDS side is linear.
      uint4 stg[8]; float stc = 0.f;
      auto dma_piece = [&](auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < 4) stg[j] = *reinterpret_cast<const uint4*>(qrow + j * qstep);
        else if constexpr (j < 8) stg[j] = *reinterpret_cast<const uint4*>(dorow + (j - 4) * dostep);
        else {
          const float* cb = wave_u == 0 ? p.delta + nrows + nsb + nq0 : (wave_u == 1 ? p.delta + nsb + nq0 : p.stats + (nsb + nq0 + 32 * (wave_u & 1)) * 2);
          stc = cb[lane];
        }
      };
      auto stage_commit = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          *reinterpret_cast<uint4*>(nbw + j * 4096 + lane * 16) = stg[j];
          *reinterpret_cast<uint4*>(nbw + C::TILE + j * 4096 + lane * 16) = stg[4 + j];
        }
        *reinterpret_cast<float*>(nbuf + 2 * C::TILE + 256 * wave_u + lane * 4) = stc;
      };
#else
      auto stage_commit = [&]() {};
      auto dma_piece = [&](auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < 4) {
          __builtin_amdgcn_global_load_lds((gbl_void*)(qrow + j * qstep), (lds_void*)(nbw + j * 4096), 16, 0, 0);
        } else if constexpr (j < 8) {
          __builtin_amdgcn_global_load_lds((gbl_void*)(dorow + (j - 4) * dostep), (lds_void*)(nbw + C::TILE + (j - 4) * 4096), 16, 0, 0);
        } else {
          const float* cb = wave_u == 0 ? p.delta + nrows + nsb + nq0 : (wave_u == 1 ? p.delta + nsb + nq0 : p.stats + (nsb + nq0 + 32 * (wave_u & 1)) * 2);
          __builtin_amdgcn_global_load_lds((gbl_void*)(cb + lane), (lds_void*)(nbuf + 2 * C::TILE + 256 * wave_u), 4, 0, 0);
        }
      };
#endif
      // ---- streams 1 + 2: S', dP' of the two 32-query halves = 16 k-steps of {2 row fragments, 2 MFMAs}, fragments
      //      issued RD k-steps ahead; softmax of half a rides under half b's MFMAs
      constexpr int RD = 6;
      bf16x8 qa[RD + 1], da[RD + 1];
      uint32_t ra[C::NS];
#pragma unroll
      for (int st = 0; st < C::NS; ++st) ra[st] = bufb + rfo[st];
      auto rissue = [&](auto K) {
        constexpr int k = decltype(K)::value;
        if (UR_DKV2_ABLATE != 7 && UR_DKV2_ABLATE != 10) rf_issue<8192 * (k >> 3), C::TILE + 8192 * (k >> 3)>(qa[k % (RD + 1)], da[k % (RD + 1)], ra[k & 7]);
      };
      auto soft1 = [&](f32x16& sv, f32x16& dpv, int r) {
        const float pr = fast_exp2(sv[r] * c2);
        sv[r] = pr;
        dpv[r] = pr * dpv[r];
      };
      static_for<0, RD>([&](auto K) { rissue(K); });
      bf16x8 p0a, d0a, p1a, d1a, p0b, d0b, p1b, d1b;
      static_for<0, 16>([&](auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr (k + RD < 16) rissue(std::integral_constant<int, k + RD>{});
        constexpr int younger = (k + RD < 16) ? RD : (15 - k);
        if (UR_DKV2_ABLATE != 7 && UR_DKV2_ABLATE != 10) rf_landed<2 * younger>(qa[k % (RD + 1)], da[k % (RD + 1)]);
        if constexpr (UR_DKV2_ABLATE == 6) {
          if constexpr (k < 8) { dma_piece(std::integral_constant<int, k>{}); if constexpr (k == 7) dma_piece(std::integral_constant<int, 8>{}); }
          else { soft1(sa, dpa, 2 * (k - 8)); soft1(sa, dpa, 2 * (k - 8) + 1); if constexpr (k == 12) { p0a = acc_frag(sa, 0); d0a = acc_frag(dpa, 0); } }
        } else if constexpr (k < 8) {
          sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[k % (RD + 1)], kf[k & 7], sa, 0, 0, 0);
          dpa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[k % (RD + 1)], vf[k & 7], dpa, 0, 0, 0);
          pin(sa, dpa);
          if constexpr (UR_DKV2_ABLATE != 9) { dma_piece(std::integral_constant<int, k>{}); if constexpr (k == 7) dma_piece(std::integral_constant<int, 8>{}); }
        } else {
          sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[k % (RD + 1)], kf[k & 7], sb, 0, 0, 0);
          dpb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[k % (RD + 1)], vf[k & 7], dpb, 0, 0, 0);
          pin(sb, dpb);
          if constexpr (UR_DKV2_ABLATE != 8 && UR_DKV2_ABLATE != 10) {      // softmax of half a, staged: scale (k) | exp2 (k+1) | dS (k+2)
            constexpr int e = 2 * (k - 8);
            sa[e] = mul1(sa[e], c2v); sa[e + 1] = mul1(sa[e + 1], c2v);
            if constexpr (k >= 9) {
              sa[e - 2] = fast_exp2(sa[e - 2]); sa[e - 1] = fast_exp2(sa[e - 1]);
              if constexpr (DIAG) { sa[e - 2] = cmask(sa[e - 2], e - 2, 0); sa[e - 1] = cmask(sa[e - 1], e - 1, 0); }
            }
            if constexpr (k >= 10) { dpa[e - 4] = mul1(sa[e - 4], dpa[e - 4]); dpa[e - 3] = mul1(sa[e - 3], dpa[e - 3]); }
          }
          if constexpr (k == 14) { p0a = acc_frag(sa, 0); d0a = acc_frag(dpa, 0); pin(p0a); pin(d0a); }       // elements 0..7 are complete after k = 13
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      UR_ASTAMP(2);
      // ---- stream 3: dV^T += dO^T P, dK^T += Q^T dS = 32 units of {one transposed fragment (2 reads), 1 MFMA}, fragments
      //      issued RT units ahead; softmax / bf16 conversion of half b ride in the first half of the stream
      constexpr int RT = 7;
      bf16x4 tl[RT + 1], th[RT + 1];
      uint32_t ta[4], tb[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { ta[dt] = bufb + tao[dt]; tb[dt] = bufb + tbo[dt]; }
      auto tissue = [&](auto U) {
        constexpr int u = decltype(U)::value;
        constexpr int bt = u >> 2, dt = u & 3;
        constexpr int off = ((bt & 1) ? 0 : C::TILE) + 256 * (16 * (bt >> 1));      // even batches: dO tile, odd: Q tile; 16 query rows per pair
        if (UR_DKV2_ABLATE != 7 && UR_DKV2_ABLATE != 10) tr_issue1<off>(tl[u % (RT + 1)], th[u % (RT + 1)], ta[dt], tb[dt]);
      };
      static_for<0, RT>([&](auto U) { tissue(U); });
      // Per unit: wait for its own fragment (6 younger units stay in flight) | MFMA | issue unit u + RT | vector fillers.
      // Issue budget of a 32x32x16 gap at one wave per SIMD (mfma_gap_lab): 32 cycles = MFMA 8 + two transposed reads ~14 +
      // ~10 of vector work -- so the 72 vector instructions of half b's softmax + the bf16 conversions are spread over the
      // whole stream, at most three per unit, each a different stage of a different element (no instruction waits on the
      // one in front of it): element e is scaled in unit e + 3, exponentiated in unit e + 4, multiplied into dS in unit e + 5
      // (e < 8), and three units later for e >= 8 (units 11 .. 21); conversions follow in units 10-13 and 22-25.
      static_for<0, 32>([&](auto U) {
        constexpr int u = decltype(U)::value;
        constexpr int bt = u >> 2, dt = u & 3;
        constexpr int younger = (31 - u < RT - 1) ? (31 - u) : (RT - 1);
        if (UR_DKV2_ABLATE != 7 && UR_DKV2_ABLATE != 10) tr_landed1<2 * younger>(tl[u % (RT + 1)], th[u % (RT + 1)]);
        const bf16x8 af = cat4(tl[u % (RT + 1)], th[u % (RT + 1)]);
        if constexpr (UR_DKV2_ABLATE == 6) { sb[u & 15] += __builtin_bit_cast(float, (int)af[0] | ((int)af[4] << 16)); }
        else {
          auto mm = [&](f32x16& acc, const bf16x8& bfrag) {
            if constexpr (dt == 0) mfma_acc_nop(acc, af, bfrag); else mfma_acc(acc, af, bfrag);     // first read of a fragment
          };
          if constexpr (bt == 0) mm(dv[dt], p0a);
          else if constexpr (bt == 1) mm(dk[dt], d0a);
          else if constexpr (bt == 2) mm(dv[dt], p1a);
          else if constexpr (bt == 3) mm(dk[dt], d1a);
          else if constexpr (bt == 4) mm(dv[dt], p0b);
          else if constexpr (bt == 5) mm(dk[dt], d0b);
          else if constexpr (bt == 6) mm(dv[dt], p1b);
          else mm(dk[dt], d1b);
        }
        if constexpr (u + RT < 32) tissue(std::integral_constant<int, u + RT>{});
        if constexpr (UR_DKV2_ABLATE != 8 && UR_DKV2_ABLATE != 10) {
          // tail of half a (its elements 12..15), then half b
          if constexpr (u == 0) {
            sa[14] = fast_exp2(sa[14]); sa[15] = fast_exp2(sa[15]);
            if constexpr (DIAG) { sa[14] = cmask(sa[14], 14, 0); sa[15] = cmask(sa[15], 15, 0); }
          }
          if constexpr (u == 1) { dpa[12] = mul1(sa[12], dpa[12]); dpa[13] = mul1(sa[13], dpa[13]); }
          if constexpr (u == 2) { dpa[14] = mul1(sa[14], dpa[14]); dpa[15] = mul1(sa[15], dpa[15]); }
          constexpr int sh = 3;                                         // half b starts in unit 3
          if constexpr (u >= sh && u < sh + 16) sb[u - sh] = mul1(sb[u - sh], c2v);
          if constexpr (u >= sh + 1 && u < sh + 17) {
            sb[u - sh - 1] = fast_exp2(sb[u - sh - 1]);
            if constexpr (DIAG) sb[u - sh - 1] = cmask(sb[u - sh - 1], u - sh - 1, 1);
          }
          if constexpr (u >= sh + 2 && u < sh + 18) dpb[u - sh - 2] = mul1(sb[u - sh - 2], dpb[u - sh - 2]);
        }
        if constexpr (u == 3) { p1a = acc_frag(sa, 1); pin(p1a); }
        if constexpr (u == 4) { d1a = acc_frag(dpa, 1); pin(d1a); }
        if constexpr (u == 13) { p0b = acc_frag(sb, 0); pin(p0b); }                  // elements 0..7 of half b are complete after unit 12
        if constexpr (u == 14) { d0b = acc_frag(dpb, 0); pin(d0b); }
        if constexpr (u == 21) { p1b = acc_frag(sb, 1); pin(p1b); }                  // elements 8..15 after unit 20
        if constexpr (u == 22) { d1b = acc_frag(dpb, 1); pin(d1b); }
        __builtin_amdgcn_sched_barrier(0);
      });
      stage_commit();
      UR_ASTAMP(6);
     };
     if (tile_diag) fast_body(std::true_type{}); else fast_body(std::false_type{});
    } else
#else
    if (kblk < p.Sk && tile_fast) {
      const uint32_t qbase = lds_off(qtile), dobase = lds_off(dotile);
      // initial accumulators: rows 8g + 4h + 0..3 of the 32-row half -> registers 4g .. 4g+3
      auto init16 = [&](f32x16& a, const float* src) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *reinterpret_cast<const float4*>(src + 8 * g + 4 * h);
          a[4 * g] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
        }
      };
      // S' and dP' of one 32-query half: row fragments are read two k-steps ahead of the MFMAs that use them
      // (one wave per SIMD: nobody else hides the LDS latency), fenced per k-step so the order stays as written
      auto s_dp = [&](f32x16& sv, f32x16& dpv, int half) {
        init16(sv, fst + 32 * half);
        init16(dpv, fst + 64 + 32 * half);
        bf16x8 qa[3], da[3];
        auto rd = [&](int st) {
          qa[st % 3] = *reinterpret_cast<const bf16x8*>(qtile + rfo[st] + 32 * 256 * half);
          da[st % 3] = *reinterpret_cast<const bf16x8*>(dotile + rfo[st] + 32 * 256 * half);
        };
        rd(0); rd(1);
#pragma unroll
        for (int st = 0; st < C::NS; ++st) {
          if (st + 2 < C::NS) rd(st + 2);
          __builtin_amdgcn_sched_barrier(0);
          sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[st % 3], kf[st], sv, 0, 0, 0);
          dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[st % 3], vf[st], dpv, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto soft = [&](f32x16& sv, f32x16& dpv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pr = fast_exp2(sv[r] * c2);
          sv[r] = pr;
          dpv[r] = pr * dpv[r];
        }
      };
      f32x16 sa, dpa, sb, dpb;
      // dV^T += dO^T P, dK^T += Q^T dS of one 32-query half: four batches of transposed fragments (dO^T and Q^T of
      // the two 16-query steps), each issued one batch ahead of the 4 MFMAs that consume the previous one
      auto dvdk_half = [&](const f32x16& pv, const f32x16& dsv, int row0) {
        const bf16x8 p0 = acc_frag(pv, 0), d0 = acc_frag(dsv, 0), p1 = acc_frag(pv, 1), d1 = acc_frag(dsv, 1);
        bf16x4 l0[4], h0[4], l1[4], h1[4];
        auto issue = [&](bf16x4 (&lo)[4], bf16x4 (&hi)[4], uint32_t base, int r0) {
          uint32_t a[4], bb[4];
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) { a[dt] = base + tao[dt] + 256 * r0; bb[dt] = base + tbo[dt] + 256 * r0; }
          tr_issue4(lo, hi, a, bb);
        };
        issue(l0, h0, dobase, row0);            // batch 0: dO^T, queries row0 .. row0+15
        issue(l1, h1, qbase, row0);             // batch 1: Q^T
        tr_landed<8>(l0, h0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat4(l0[dt], h0[dt]), p0, dv[dt], 0, 0, 0);
        issue(l0, h0, dobase, row0 + 16);       // batch 2: dO^T, queries row0+16 .. row0+31
        tr_landed<8>(l1, h1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat4(l1[dt], h1[dt]), d0, dk[dt], 0, 0, 0);
        issue(l1, h1, qbase, row0 + 16);        // batch 3: Q^T
        tr_landed<8>(l0, h0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat4(l0[dt], h0[dt]), p1, dv[dt], 0, 0, 0);
        tr_landed<0>(l1, h1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat4(l1[dt], h1[dt]), d1, dk[dt], 0, 0, 0);
      };
#if UR_DKV2_ABLATE == 4
      sa = dk[0]; dpa = dv[0]; sb = dk[1]; dpb = dv[1];
#else
      // phase 1: S', dP' of half a
      s_dp(sa, dpa, 0);
      __builtin_amdgcn_sched_barrier(0);
      UR_ASTAMP(1);
      // phase 2: S', dP' of half b; softmax of half a
      s_dp(sb, dpb, 1);
#endif
      UR_ASTAMP(2);
#if UR_DKV2_ABLATE != 2
      soft(sa, dpa);
#endif
      __builtin_amdgcn_sched_barrier(0);
      UR_ASTAMP(3);
#if UR_DKV2_ABLATE != 1
      if (it + 1 < ntot) load_tile(it + 1, smem + ((it + 1) & 1) * STG);
#endif
      __builtin_amdgcn_sched_barrier(0);
      UR_ASTAMP(4);
#if UR_DKV2_ABLATE == 3
      dk[0][0] += sa[0] + dpa[1] + sb[2] + dpb[3];
#else
      // phase 3: dV, dK of half a; softmax of half b
      dvdk_half(sa, dpa, 0);
#if UR_DKV2_ABLATE != 2
      soft(sb, dpb);
#endif
      __builtin_amdgcn_sched_barrier(0);
      UR_ASTAMP(5);
      // phase 4: dV, dK of half b
      dvdk_half(sb, dpb, 32);
#endif
      UR_ASTAMP(6);
    } else
#endif
    if (kblk < p.Sk) {

#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const int qbase = q0 + 32 * sub;
        if (qbase >= p.Sq) break;
        if (CAUSAL && qbase + 31 < kblk) continue;        // every query of this sub-tile precedes every key
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int st = 0; st < C::NS; ++st) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(qtile, 32 * sub, st, lane), kf[st], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(dotile, 32 * sub, st, lane), vf[st], dp, 0, 0, 0);
        }
        const int qb0 = opaque(qbase + 4 * h), keyo = opaque(key);
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int qr = 32 * sub + 8 * rq + 4 * h;
          const float4 a = *reinterpret_cast<const float4*>(fst + 128 + 2 * qr);          // (m, 1/l) of rows qr, qr+1
          const float4 bq = *reinterpret_cast<const float4*>(fst + 128 + 2 * qr + 4);     // rows qr+2, qr+3
          const float4 cq = *reinterpret_cast<const float4*>(fst + 64 + qr);              // -delta of rows qr..qr+3
          const float ma[4] = {a.x, a.z, bq.x, bq.z}, iv[4] = {a.y, a.w, bq.y, bq.w}, dl[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            const int qpos = qb0 + e + 8 * rq;
            const float sc = mask_score<CAUSAL>(s[r], p.scale, kvalid, kok, keyo, qpos);       // natural-log domain
            const float ps = (sc == NEG_INF || qpos >= p.Sq) ? 0.f : fast_exp2((sc - ma[e]) * LOG2E) * iv[e];
            s[r] = ps;
            dp[r] = ps * (dp[r] + dl[e]);
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = acc_frag(s, s2), df = acc_frag(dp, s2);
          bf16x8 tdo[C::NDT], tq[C::NDT];
          tr_frags<HD>(tdo, dotile, 32 * sub + 16 * s2, lane);
          tr_frags<HD>(tq, qtile, 32 * sub + 16 * s2, lane);
#pragma unroll
          for (int dt = 0; dt < C::NDT; ++dt) {
#if UR_DKV2_V2
            mfma_acc_nop(dv[dt], tdo[dt], pf);      // dK^T / dV^T stay in AccVGPRs on every path (a builtin here would copy 128 registers in and out)
            mfma_acc_nop(dk[dt], tq[dt], df);
#else
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tdo[dt], pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[dt], df, dk[dt], 0, 0, 0);
#endif
          }
        }
      }
      if (it + AH < ntot) load_tile(it + AH, smem + ((it + AH) % NB) * STG);
    } else {
      if (it + AH < ntot) load_tile(it + AH, smem + ((it + AH) % NB) * STG);     // waves past Sk still take part in the staging
    }
    __syncthreads();
    UR_ASTAMP(7);
    if (++tq_c == ntq) { tq_c = 0; ++hr_c; }
    if (++tq_n == ntq) { tq_n = 0; ++hr_n; }
    buf_c = (buf_c + 1 == NB) ? 0 : buf_c + 1;
  }
  store_T<HD>(p.dk + ktok * p.lddk + (long)kvh * HD, dk, p.scale, lane, kok);     // dS was kept unscaled
  store_T<HD>(p.dv + ktok * p.lddv + (long)kvh * HD, dv, 1.0f, lane, kok);
}

// ================================================================================================
// Tiny attention: <= 4 learned queries against <= 16 keys, head_dim 64 -- the item Q-Former inside the joint step
// (models/qformer.py:169-275 with Q_item = 2: self-attention 2 x 2, cross-attention 2 x 14 fields, over
// 64 x 50 items x 16 heads = 51 200 (item, head) pairs per launch) and BASELINE config C1 (4 x 4, 4 x 8).
// The MFMA kernels above give every pair a 32-query x 64-key tile of its own (a wave, an LDS tile, a barrier) and use
// 2 of its 32 query rows: 246 / 304 / 280 us per forward / dQ / dK-dV launch for ~200 MB of traffic.  Here a 16-lane
// DPP row owns one pair (4 pairs per wave): lane sl holds elements 4 sl .. 4 sl + 3 of every q / k / v / dO row (8-byte
// loads, a 128-byte row per DPP row), a score is 4 FMAs + a 4-step DPP row reduction that leaves the total in all 16
// lanes, so softmax, the mask semantics (additive finfo.min: a fully masked row is uniform) and dropout run on
// row-uniform registers in f32; the dropout decision of key j is hashed once, by lane j, with the same counter as the
// MFMA kernels (identical masks) and shared through a ballot.  The backward is ONE kernel (dQ, dK, dV: a pair's
// gradients never leave its row -- no atomics, no row constants to publish).
constexpr int TK = 16;
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, total in every lane: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row_sum16(float x) {
  x += dpp_mov<0xB1>(x);
  x += dpp_mov<0x4E>(x);
  x += dpp_mov<0x141>(x);
  x += dpp_mov<0x140>(x);
  return x;
}
__device__ __forceinline__ float dot4(uint2 a, uint2 b) {
  return fmaf(bf_lo(a.x), bf_lo(b.x), fmaf(bf_hi(a.x), bf_hi(b.x), fmaf(bf_lo(a.y), bf_lo(b.y), bf_hi(a.y) * bf_hi(b.y))));
}
__device__ __forceinline__ void axpy4(float (&acc)[4], float w, uint2 x) {
  acc[0] = fmaf(w, bf_lo(x.x), acc[0]); acc[1] = fmaf(w, bf_hi(x.x), acc[1]);
  acc[2] = fmaf(w, bf_lo(x.y), acc[2]); acc[3] = fmaf(w, bf_hi(x.y), acc[3]);
}
struct TinyPair { int b, hq, sl, g; bool ok; long qoff, koff; };
__device__ __forceinline__ TinyPair tiny_pair(const AttnP& p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  TinyPair t;
  t.g = lane >> 4; t.sl = lane & 15;
  const long npairs = (long)p.B * p.nq;
  const long pr = ((long)blockIdx.x * 4 + wave) * 4 + t.g;
  t.ok = pr < npairs;
  const long pc = t.ok ? pr : npairs - 1;          // tail rows compute on a clamped pair (DPP / ballots need every lane)
  t.b = (int)(pc / p.nq); t.hq = (int)(pc % p.nq);
  return t;
}
// scores -> probabilities of one pair, row-uniform: P[qi][kj] = exp(sc - m) (before dropout and 1/l), m, 1/l and the
// keep bits (bit kj of keep[qi]; all ones without dropout)
template <int TQ>
__device__ __forceinline__ void tiny_probs(const AttnP& p, const TinyPair& t, const uint2 (&qv)[TQ], const uint2 (&kv)[TK],
                                           float (&P)[TQ][TK], float (&mrow)[TQ], float (&inv)[TQ], uint32_t (&keep)[TQ]) {
  const bool kval = t.sl < p.Sk && (p.kmask == nullptr || p.kmask[(long)t.b * p.Sk + t.sl] != 0);
  const uint32_t vbits = (uint32_t)(__ballot(kval) >> (16 * t.g)) & 0xffffu;
#pragma unroll
  for (int qi = 0; qi < TQ; ++qi) {
    mrow[qi] = 0.f; inv[qi] = 0.f; keep[qi] = 0xffffu;
    if (qi >= p.Sq) continue;
    float mx = NEG_INF;
#pragma unroll
    for (int kj = 0; kj < TK; ++kj) {
      P[qi][kj] = NEG_INF;
      if (kj >= p.Sk) continue;
      const float raw = row_sum16(dot4(qv[qi], kv[kj]));
      const float sc = ((vbits >> kj) & 1u) ? raw * p.scale : F32_MIN;
      P[qi][kj] = sc;
      mx = fmaxf(mx, sc);
    }
    float l = 0.f;
#pragma unroll
    for (int kj = 0; kj < TK; ++kj) {
      const float e = (kj < p.Sk) ? fast_exp2((P[qi][kj] - mx) * LOG2E) : 0.f;
      P[qi][kj] = e;
      l += e;
    }
    mrow[qi] = mx; inv[qi] = 1.0f / l;             // l >= 1: the maximum itself contributes exp(0)
    if (p.drop_thr != 0) {
      // lane = key t.sl of query row qi: the row's two keys, then the pair word of this key (the same words the MFMA kernels draw)
      const ur_rowkey rk = ur_attn_row_key(p.seed, p.drow0 + ((uint64_t)((long)t.b * p.nq + t.hq) * p.Sq + (uint64_t)qi));
      const bool kp = ur_attn_keep_scale(ur_attn_pair_word(rk.k1, rk.k2, (uint32_t)t.sl >> 1), (uint32_t)t.sl, p.drop_thr, 1.0f) != 0.f;
      keep[qi] = (uint32_t)(__ballot(kp) >> (16 * t.g)) & 0xffffu;
    }
  }
}
template <int TQ>
__device__ __forceinline__ void tiny_load(const AttnP& p, const TinyPair& t, uint2 (&qv)[TQ], uint2 (&kv)[TK], uint2 (&vv)[TK]) {
  const long e0 = (long)t.hq * 64 + 4 * t.sl;
#pragma unroll
  for (int qi = 0; qi < TQ; ++qi)
    qv[qi] = (qi < p.Sq) ? *reinterpret_cast<const uint2*>(p.q + ((long)t.b * p.Sq + qi) * p.ldq + e0) : make_uint2(0, 0);
#pragma unroll
  for (int kj = 0; kj < TK; ++kj) {
    const bool in = kj < p.Sk;
    kv[kj] = in ? *reinterpret_cast<const uint2*>(p.k + ((long)t.b * p.Sk + kj) * p.ldk + e0) : make_uint2(0, 0);
    vv[kj] = in ? *reinterpret_cast<const uint2*>(p.v + ((long)t.b * p.Sk + kj) * p.ldv + e0) : make_uint2(0, 0);
  }
}

template <int TQ>
__global__ __launch_bounds__(256) void attn_tiny_fwd_kernel(AttnP p) {
  const TinyPair t = tiny_pair(p);
  uint2 qv[TQ], kv[TK], vv[TK];
  tiny_load<TQ>(p, t, qv, kv, vv);
  float P[TQ][TK], mrow[TQ], inv[TQ];
  uint32_t keep[TQ];
  tiny_probs<TQ>(p, t, qv, kv, P, mrow, inv, keep);
  const long e0 = (long)t.hq * 64 + 4 * t.sl;
#pragma unroll
  for (int qi = 0; qi < TQ; ++qi) {
    if (qi >= p.Sq) continue;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kj = 0; kj < TK; ++kj) {
      if (kj >= p.Sk) continue;
      const float w = ((keep[qi] >> kj) & 1u) ? P[qi][kj] * p.drop_inv : 0.f;
      axpy4(acc, w, vv[kj]);
    }
    if (t.ok) {
      const float s = inv[qi];
      *reinterpret_cast<uint2*>(p.o + ((long)t.b * p.Sq + qi) * p.ldo + e0) = make_uint2(pack_bf2(acc[0] * s, acc[1] * s), pack_bf2(acc[2] * s, acc[3] * s));
      if (t.sl == 0) {
        float* st = p.stats + (((long)t.b * p.nq + t.hq) * p.Sq + qi) * 2;
        st[0] = mrow[qi]; st[1] = s;
      }
    }
  }
}

template <int TQ>
__global__ __launch_bounds__(256) void attn_tiny_bwd_kernel(AttnP p) {
  const TinyPair t = tiny_pair(p);
  uint2 qv[TQ], kv[TK], vv[TK], dov[TQ];
  tiny_load<TQ>(p, t, qv, kv, vv);
  const long e0 = (long)t.hq * 64 + 4 * t.sl;
#pragma unroll
  for (int qi = 0; qi < TQ; ++qi)
    dov[qi] = (qi < p.Sq) ? *reinterpret_cast<const uint2*>(p.dout + ((long)t.b * p.Sq + qi) * p.lddo + e0) : make_uint2(0, 0);
  float P[TQ][TK], mrow[TQ], inv[TQ];
  uint32_t keep[TQ];
  tiny_probs<TQ>(p, t, qv, kv, P, mrow, inv, keep);
  // normalised probabilities with dropout, and delta[qi] = sum_d dO * O
  float delta[TQ];
#pragma unroll
  for (int qi = 0; qi < TQ; ++qi) {
    delta[qi] = 0.f;
    if (qi >= p.Sq) continue;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kj = 0; kj < TK; ++kj) {
      if (kj >= p.Sk) continue;
      P[qi][kj] *= inv[qi];                                              // p / l
      const float w = ((keep[qi] >> kj) & 1u) ? P[qi][kj] * p.drop_inv : 0.f;
      axpy4(acc, w, vv[kj]);
    }
    const uint2 d = dov[qi];
    delta[qi] = row_sum16(fmaf(bf_lo(d.x), acc[0], fmaf(bf_hi(d.x), acc[1], fmaf(bf_lo(d.y), acc[2], bf_hi(d.y) * acc[3]))));
  }
  float dq[TQ][4];
#pragma unroll
  for (int qi = 0; qi < TQ; ++qi) { dq[qi][0] = dq[qi][1] = dq[qi][2] = dq[qi][3] = 0.f; }
#pragma unroll
  for (int kj = 0; kj < TK; ++kj) {
    if (kj >= p.Sk) continue;
    float dk[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qi = 0; qi < TQ; ++qi) {
      if (qi >= p.Sq) continue;
      const float dsc = ((keep[qi] >> kj) & 1u) ? p.drop_inv : 0.f;
      const float dP = row_sum16(dot4(dov[qi], vv[kj]));
      const float pd = P[qi][kj] * dsc;
      const float dS = P[qi][kj] * (dP * dsc - delta[qi]) * p.scale;
      axpy4(dv, pd, dov[qi]);
      axpy4(dk, dS, qv[qi]);
      axpy4(dq[qi], dS, kv[kj]);
    }
    if (t.ok) {
      *reinterpret_cast<uint2*>(p.dk + ((long)t.b * p.Sk + kj) * p.lddk + e0) = make_uint2(pack_bf2(dk[0], dk[1]), pack_bf2(dk[2], dk[3]));
      *reinterpret_cast<uint2*>(p.dv + ((long)t.b * p.Sk + kj) * p.lddv + e0) = make_uint2(pack_bf2(dv[0], dv[1]), pack_bf2(dv[2], dv[3]));
    }
  }
#pragma unroll
  for (int qi = 0; qi < TQ; ++qi)
    if (qi < p.Sq && t.ok)
      *reinterpret_cast<uint2*>(p.dq + ((long)t.b * p.Sq + qi) * p.lddq + e0) = make_uint2(pack_bf2(dq[qi][0], dq[qi][1]), pack_bf2(dq[qi][2], dq[qi][3]));
}

// Kernel-selection switches of the attention entry points: process-wide words set through the C ABI (ur_attn_mode, include/unirec_hip.h) --
// the library reads no environment variable.  Every alternative is a complete, tested path (bit-identity / oracle tests flip them).
std::atomic<int> g_attn_mode[UR_ATTN_MODE_COUNT] = {{3}, {1}, {1}, {1}};      // TINY: bit 0 forward, bit 1 backward; C128, DKV_PERSIST, FEWQ: 0 / 1
inline int attn_mode(int key) { return g_attn_mode[key].load(std::memory_order_relaxed); }
// UR_ATTN_MODE_TINY = 0 keeps tiny shapes on the MFMA kernels (1 / 2: tiny forward / backward only)
inline bool tiny_enabled(bool bwd) { return (attn_mode(UR_ATTN_MODE_TINY) & (bwd ? 2 : 1)) != 0; }
inline bool tiny_shape(const AttnP& p, int hd, bool causal, bool bwd) {
  return hd == 64 && !causal && p.rep == 1 && p.Sq <= 4 && p.Sk <= TK && tiny_enabled(bwd);
}
int launch_tiny(const AttnP& p, bool bwd, hipStream_t st) {
  const long npairs = (long)p.B * p.nq;
  dim3 grid((unsigned)((npairs + 15) / 16));
  if (!bwd) {
    if (p.Sq <= 2) hipLaunchKernelGGL((attn_tiny_fwd_kernel<2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_tiny_fwd_kernel<4>), grid, dim3(256), 0, st, p);
    UR_CHECK_LAUNCH("ur_attn_fwd(tiny)");
  } else {
    if (p.Sq <= 2) hipLaunchKernelGGL((attn_tiny_bwd_kernel<2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_tiny_bwd_kernel<4>), grid, dim3(256), 0, st, p);
    UR_CHECK_LAUNCH("ur_attn_bwd(tiny)");
  }
  return 0;
}

// ================================================================================================
// Causal head_dim-128 forward, hand-scheduled: the key-tile loop is ONE generated inline-asm block (tools/asmgen/attn_fwd.py ->
// gen/attn_fwd_c128_asm.h; emulated and hazard-checked on the CPU by tests/test_asmgen_attn_fwd.py).  A workgroup = 4 waves = 256
// query rows of one (batch, query head); a wave owns 64 rows and the whole register file of its SIMD (one wave per SIMD), so every
// K / V fragment read from LDS feeds two MFMAs; q is pre-scaled by scale*log2e here and the running maximum enters the S chain as its
// C operand, so a score costs one v_exp_f32, one v_add_f32, half a v_max3 and half a v_cvt_pk.  Replaces the SDPA call of
// transformers modeling_qwen3.py:185-208 behind /root/reference/training/train_item_individual_token_joint.py:173-177 for Sq == Sk,
// Sk % 64 == 0, Sk <= 4096; other shapes keep attn_fwd_kernel.
// LDS: K ring 4 x 16 KiB | V ring 4 x 16 KiB | key bias f32[Sk] (0 / -inf) | key-state words.
namespace c128 {
constexpr int VBASE_LDS = 65536, BIAS_LDS = 131072, WORDS_LDS = BIAS_LDS + 16384, LDS_BYTES = WORDS_LDS + 64 * 8;
constexpr int MAX_SK = 4096;
typedef __attribute__((ext_vector_type(32))) float f32x32;
typedef __attribute__((ext_vector_type(32))) int i32x32;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
}  // namespace c128
#ifndef UR_C128_STAMPS
#define UR_C128_STAMPS 0      // lab builds only (tools/lab/c128_variants.sh stamps): per-wave cycle accumulators of the generated loop + four stamps of the C++ part
#endif
#if UR_C128_STAMPS
__device__ unsigned int g_c128_stamps[8192 * 4 * 32];
#endif

// Persistent: the grid is one workgroup per CU; a work item = one (batch, query head) x one PAIR of query blocks (the heaviest
// remaining and the lightest remaining: every item sweeps 4 nx + 4 key tiles, so a static round robin balances).  Item ids are dealt
// so that the rep x nch items of a (batch, kv head) group run on ONE XCD at the same time (its K / V stay in that L2).
struct C128Div { uint32_t m_pg, m_nch, m_nkv; };      // ceil(2^32 / d) of the item-decode divisors
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_fwd_c128_kernel(AttnP p, int nitems, int nch, C128Div dv) {
  using namespace c128;
  using C = Cfg<128>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, l31 = lane & 31;
  const int nx = (p.Sq + 255) / 256, ntiles = p.Sk / KT;
  const uint32_t lds0 = lds_off(smem);
  float* bias = reinterpret_cast<float*>(smem + BIAS_LDS);
  unsigned long long* words = reinterpret_cast<unsigned long long*>(smem + WORDS_LDS);

  // lane-constant LDS addresses (tile-relative swizzled offsets; the ring slot is an immediate in the generated code)
  i32x8 ka, tatb;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) ka[ks] = (int)(lds0 + C::off(l31, 2 * ks + h));
  {
    const int g16 = (lane >> 4) & 1, i = lane & 15;
    const int row = 4 * h + (i >> 2), sub8 = 8 * (i & 1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int ch = 4 * dt + 2 * g16 + ((i & 3) >> 1);
      tatb[dt] = (int)(lds0 + VBASE_LDS + C::off(row, ch) + sub8);
      tatb[4 + dt] = (int)(lds0 + VBASE_LDS + C::off(row + 8, ch) + sub8);
    }
  }
  i32x2 voff, bd;
  {
    const int row = 4 * wave + (lane >> 4), pos = lane & 15;
    const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
    voff[0] = (int)((uint32_t)(row * p.ldk + (pos ^ sw) * 8) * 2u);
    voff[1] = (int)((uint32_t)(row * p.ldv + (pos ^ sw) * 8) * 2u);
  }
  bd[0] = (int)(lds0 + BIAS_LDS + 16 * h);
  bd[1] = l31 - 4 * h;
  const int k16b = __builtin_amdgcn_readfirstlane((int)(p.ldk * 32)), v16b = __builtin_amdgcn_readfirstlane((int)(p.ldv * 32));
  const uint32_t waveb = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 1024u);
  const float c = p.scale * LOG2E;
  const int ngroups = p.nkv * p.B;
  const uint32_t per_group = (uint32_t)(p.rep * nch);

  // One query block = one pass of the generated loop.  The passes of this workgroup are software-pipelined: as soon as every wave
  // has left the loop of block n, the q rows and the first K / V tiles of block n + 1 are requested, and their latencies run
  // under the normalisation and the stores of block n.
  struct Blk { int item, half, x, hq, b, tend, tfirst; uint32_t mb_lo, mb_hi, kb_lo, kb_hi, vb_lo, vb_hi; bool valid; };
  // n / d for the small launch constants d (n * d < 2^32): one s_mul_hi_u32 against ceil(2^32 / d) instead of hipcc's ~30-instruction
  // expansion of a 32-bit division -- the decode below ran 2200 cycles per block with plain '/' and '%'
  auto udiv = [](uint32_t n, uint32_t magic, uint32_t d) { return d == 1u ? n : __umulhi(n, magic); };
  auto decode = [&](int item, int half, Blk& d) {
    d.valid = item < nitems;
    d.item = item; d.half = half;
    if (!d.valid) return;
    uint32_t g, j;
    if ((ngroups & 7) == 0) { const uint32_t slot = (uint32_t)item >> 3, gq = udiv(slot, dv.m_pg, per_group); g = gq * 8u + ((uint32_t)item & 7u); j = slot - gq * per_group; }
    else { g = udiv((uint32_t)item, dv.m_pg, per_group); j = (uint32_t)item - g * per_group; }
    const uint32_t hr = udiv(j, dv.m_nch, nch), ch = j - hr * nch;
    const uint32_t bb = udiv(g, dv.m_nkv, p.nkv), kvh = g - bb * p.nkv;
    d.hq = (int)(kvh * p.rep + hr); d.b = (int)bb;
    d.x = half == 0 ? nx - 1 - (int)ch : (int)ch;
    d.tend = __builtin_amdgcn_readfirstlane(min(ntiles, 4 * d.x + 4));
    const bf16_t* kb = p.k + (long)d.b * p.Sk * p.ldk + (long)kvh * 128;
    const bf16_t* vb = p.v + (long)d.b * p.Sk * p.ldv + (long)kvh * 128;
    d.kb_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)kb); d.kb_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)kb >> 32));
    d.vb_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)vb); d.vb_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)vb >> 32));
  };
  // the block after (item, half): the second block of the pair (unless the pair is the single middle block), else the next item
  auto advance = [&](const Blk& c, Blk& n) {
    const int chc = c.half == 0 ? nx - 1 - c.x : c.x;
    if (c.half == 0 && nx - 1 - chc != chc) decode(c.item, 1, n); else decode(c.item + (int)gridDim.x, 0, n);
  };
  // key state of a sample: bias table (0 / -inf per key) + one word per tile (only with a key mask).  Caller: every wave has left the loop.
  auto key_state = [&](Blk& d) {
    d.tfirst = 0; d.mb_lo = 0u; d.mb_hi = 0u;
    const uint8_t* km = p.kmask ? p.kmask + (long)d.b * p.Sk : nullptr;
    if (km != nullptr) {
      for (int t = wave; t < ntiles; t += 4) {
        const int key = t * KT + lane;
        const bool ok = km[key] != 0;
        bias[key] = ok ? 0.f : NEG_INF;
        const unsigned long long wv = __ballot(ok);
        if (lane == 0) words[t] = wv;
      }
      __syncthreads();
      const unsigned long long wv = lane < ntiles ? words[lane] : ~0ull;
      const unsigned long long anym = __ballot(lane < ntiles && wv != 0ull), partm = __ballot(lane < ntiles && wv != ~0ull);
      d.tfirst = anym ? __builtin_ctzll(anym) : ntiles;
      d.mb_lo = (uint32_t)partm; d.mb_hi = (uint32_t)(partm >> 32);
    }
    d.tfirst = __builtin_amdgcn_readfirstlane(d.tfirst);
    d.mb_lo = __builtin_amdgcn_readfirstlane(d.mb_lo); d.mb_hi = __builtin_amdgcn_readfirstlane(d.mb_hi);
  };
  bf16x8 qf[2][8];
  // q rows of this wave (two 32-row blocks) and the first tiles' LDS-DMA of block d.  Caller: the K / V rings are free.
  auto request = [&](const Blk& d) {
    // (qf is assigned on every path: a conditional assignment would keep its 64 registers alive across the generated loop)
    const bool live = d.valid && d.tfirst < d.tend;
    const int q0 = live ? 256 * d.x + 64 * wave : 0;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int row = q0 + 32 * qb + l31;
      const bool ok = live && row < p.Sq;
      const bf16_t* qrow = p.q + ((long)(live ? d.b : 0) * p.Sq + (ok ? row : 0)) * p.ldq + (long)(live ? d.hq : 0) * 128;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qf[qb][ks] = g_frag(qrow, ks, lane, ok);
    }
    if (!live) return;
    asm volatile(
        "s_mov_b32 s36, %[kbl]\n\ts_mov_b32 s37, %[kbh]\n\ts_mov_b32 s38, %[vbl]\n\ts_mov_b32 s39, %[vbh]\n\t"
        "s_mov_b32 s40, %[k16]\n\ts_mov_b32 s41, %[v16]\n\ts_mov_b32 s43, %[tend]\n\ts_mov_b32 s45, %[tfirst]\n\ts_mov_b32 s57, %[waveb]\n\t"
        UR_ATTN_FWD_C128_DMA_ASM
        :
        : "{v[204:205]}"(voff), [kbl] "s"(d.kb_lo), [kbh] "s"(d.kb_hi), [vbl] "s"(d.vb_lo), [vbh] "s"(d.vb_hi), [k16] "s"(k16b), [v16] "s"(v16b),
          [tend] "s"(d.tend), [tfirst] "s"(d.tfirst), [waveb] "s"(waveb)
        : UR_ATTN_FWD_C128_DMA_CLOBBERS);
  };

  Blk cur, nxt;
  decode((int)blockIdx.x, 0, cur);
  if (cur.valid) { key_state(cur); __syncthreads(); request(cur); }
  while (cur.valid) {
    const int x = cur.x, hq = cur.hq, b = cur.b;
    const int q0 = 256 * x + 64 * wave;
#if UR_C128_STAMPS
    const unsigned long long st0 = __builtin_readcyclecounter();
    unsigned long long st1 = st0, st2 = st0;
    unsigned int* dbg = g_c128_stamps + ((size_t)((cur.item * 2 + cur.half) & 8191) * 4 + wave) * 32;
    unsigned long long sa = 0, sb = 0, sc = 0, sd = 0;
#endif
    f32x32 o0, o1, o2, o3, lsum;       // lsum: [qb][16], every register of a query block's tile = the row sum
    c128::f32x2 mrow;
    advance(cur, nxt);                 // scalar work: its results wait in scalar registers / spill lanes while the loop runs
    if (cur.tfirst < cur.tend) {
      // element j of lane half h of k-step ks = q[row][16 ks + 8 h + j] * scale * log2(e), rounded to bf16 once more
      i32x32 qv0, qv1;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          qv0[4 * ks + jj] = (int)pack_bf2(bf2f((bf16_t)qf[0][ks][2 * jj]) * c, bf2f((bf16_t)qf[0][ks][2 * jj + 1]) * c);
          qv1[4 * ks + jj] = (int)pack_bf2(bf2f((bf16_t)qf[1][ks][2 * jj]) * c, bf2f((bf16_t)qf[1][ks][2 * jj + 1]) * c);
        }
      const int tlast = __builtin_amdgcn_readfirstlane(q0 < p.Sq ? min(4 * x + wave, ntiles - 1) : -1);
#if UR_C128_STAMPS
      const uint32_t db_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dbg), db_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)dbg >> 32));
      st1 = __builtin_readcyclecounter();
#endif
      asm volatile(
          "s_mov_b32 s36, %[kbl]\n\ts_mov_b32 s37, %[kbh]\n\ts_mov_b32 s38, %[vbl]\n\ts_mov_b32 s39, %[vbh]\n\t"
          "s_mov_b32 s40, %[k16]\n\ts_mov_b32 s41, %[v16]\n\ts_mov_b32 s43, %[tend]\n\ts_mov_b32 s44, %[tlast]\n\t"
          "s_mov_b32 s45, %[tfirst]\n\ts_mov_b32 s46, %[mbl]\n\ts_mov_b32 s47, %[mbh]\n\ts_mov_b32 s57, %[waveb]\n\t"
#if UR_C128_STAMPS
          "s_mov_b32 s72, %[dbl]\n\ts_mov_b32 s73, %[dbh]\n\t"
#endif
          UR_ATTN_FWD_C128_ASM
          : "=&{a[0:31]}"(o0), "=&{a[32:63]}"(o1), "=&{a[64:95]}"(o2), "=&{a[96:127]}"(o3), "=&{a[192:223]}"(lsum), "=&{v[192:193]}"(mrow)
          : "{a[128:159]}"(qv0), "{a[160:191]}"(qv1), "{v[176:183]}"(ka), "{v[184:191]}"(tatb), "{v[204:205]}"(voff), "{v[208:209]}"(bd),
            [kbl] "s"(cur.kb_lo), [kbh] "s"(cur.kb_hi), [vbl] "s"(cur.vb_lo), [vbh] "s"(cur.vb_hi), [k16] "s"(k16b), [v16] "s"(v16b), [tend] "s"(cur.tend),
            [tlast] "s"(tlast), [tfirst] "s"(cur.tfirst), [mbl] "s"(cur.mb_lo), [mbh] "s"(cur.mb_hi), [waveb] "s"(waveb)
#if UR_C128_STAMPS
            , [dbl] "s"(db_lo), [dbh] "s"(db_hi)
#endif
          : UR_ATTN_FWD_C128_CLOBBERS);
#if UR_C128_STAMPS
      st2 = __builtin_readcyclecounter();
#endif
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) { o0[i] = 0.f; o1[i] = 0.f; o2[i] = 0.f; o3[i] = 0.f; lsum[i] = 0.f; }
      mrow = c128::f32x2{0.f, 0.f};
    }
    // the next block (decoded before the loop): once every wave has left the loop, its key state is built; its q rows and first tiles
    // are requested between the two halves of this block's epilogue (after the first half has freed its registers) and fly under the second half
    // (the first half of the epilogue runs BEFORE the barrier: the waves with fewer diagonal tiles leave the loop up to ~4 k cycles
    // early and spend that wait on their own stores)
    // epilogue (the row sums came off the matrix pipe: complete in every lane)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int row = q0 + 32 * qb + l31;
      const bool ok = row < p.Sq;
      const float l = lsum[16 * qb];
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      f32x16 acc[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int idx = 16 * (dt & 1) + r;
          acc[dt][r] = qb == 0 ? (dt < 2 ? o0[idx] : o1[idx]) : (dt < 2 ? o2[idx] : o3[idx]);
        }
      store_T<128>(p.o + ((long)b * p.Sq + (ok ? row : 0)) * p.ldo + (long)hq * 128, acc, inv, lane, ok);
      if (ok && h == 0) {
        float* st = p.stats + (((long)b * p.nq + hq) * p.Sq + row) * 2;
        st[0] = mrow[qb] * (1.0f / LOG2E);        // running maximum in natural-log units of the scaled scores (the backward's convention)
        st[1] = inv;
      }
      if (qb == 0) {
        asm volatile("" ::: "memory");
#if UR_C128_STAMPS
        sa = __builtin_readcyclecounter();
#endif
        __syncthreads();                     // every wave has left the loop: the key state of the next sample and the rings are free
#if UR_C128_STAMPS
        sb = __builtin_readcyclecounter();
#endif
        if (nxt.valid) {
          if (nxt.item != cur.item) { key_state(nxt); if (p.kmask) __syncthreads(); }
          else { nxt.tfirst = cur.tfirst; nxt.mb_lo = cur.mb_lo; nxt.mb_hi = cur.mb_hi; }
        }
#if UR_C128_STAMPS
        sc = __builtin_readcyclecounter();
#endif
        request(nxt);
#if UR_C128_STAMPS
        sd = __builtin_readcyclecounter();
#endif
      }
    }
#if UR_C128_STAMPS
    if (lane == 0) {
      const unsigned long long st3 = __builtin_readcyclecounter();
      dbg[12] = (unsigned int)(st1 - st0); dbg[13] = (unsigned int)(st2 - st1); dbg[14] = (unsigned int)(st3 - st2); dbg[15] = (unsigned int)x;
      dbg[16] = (unsigned int)(sa - st2); dbg[17] = (unsigned int)(sb - sa); dbg[18] = (unsigned int)(sc - sb); dbg[19] = (unsigned int)(sd - sc); dbg[20] = (unsigned int)(st3 - sd);
    }
#endif
    cur = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA pieces of the last loop iterations: landed before the workgroup gives its LDS back
}

// ================================================================================================
// Causal head_dim-128 backward dQ, hand-scheduled (tools/asmgen/attn_dq.py -> gen/attn_dq_c128_asm.h; CPU-emulated by
// tests/test_asmgen_attn_dq.py).  Same decomposition and persistent item loop as attn_fwd_c128_kernel; per 64-key tile and wave
// 96 MFMAs (S, dP, dQ) with every K / V fragment feeding two of them.  The C++ part loads q (pre-scaled by scale*log2e), dO and O of
// the wave's 64 rows, forms the row constants delta = sum_d dO O and LSE2 = (m + ln l) log2e from the forward's statistics, and
// publishes -delta and -LSE/scale for the dK/dV kernel exactly as attn_bwd_dq_kernel does.
// LDS: K ring 4 x 16 KiB | V ring 4 x 16 KiB | key-state words.
// Work queues of the persistent dK/dV kernel: p.queue[XCD lane] = next key-block item of that lane.  The eight words live in the
// caller's workspace (the tail of `delta`): this call's dQ kernel zeroes them, its dK/dV kernel -- stream-ordered behind it -- draws
// from them; two calls on two streams never share a word (the library owns no mutable device state).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
namespace c128 { constexpr int DQ_WORDS_LDS = 131072, DQ_LDS_BYTES = DQ_WORDS_LDS + 64 * 8; }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dq_c128_kernel(AttnP p, int nitems, int nch, C128Div dv) {
  using namespace c128;
  using C = Cfg<128>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, l31 = lane & 31;
  const int nx = (p.Sq + 255) / 256, ntiles = p.Sk / KT;
  if (blockIdx.x == 0 && tid < 8) p.queue[tid] = 0u;      // (the dK/dV kernel of this call starts after this kernel has finished)
  const uint32_t lds0 = lds_off(smem);
  unsigned long long* words = reinterpret_cast<unsigned long long*>(smem + DQ_WORDS_LDS);
  i32x2 kava, dw, voff;
  i32x8 tatb;
  kava[0] = (int)(lds0 + C::off(l31, h));
  kava[1] = kava[0] + VBASE_LDS;
  {
    const int g16 = (lane >> 4) & 1, i = lane & 15;
    const int row = 4 * h + (i >> 2), sub8 = 8 * (i & 1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int ch = 4 * dt + 2 * g16 + ((i & 3) >> 1);
      tatb[dt] = (int)(lds0 + C::off(row, ch) + sub8);
      tatb[4 + dt] = (int)(lds0 + C::off(row + 8, ch) + sub8);
    }
  }
  {
    const int row = 4 * wave + (lane >> 4), pos = lane & 15;
    const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
    voff[0] = (int)((uint32_t)(row * p.ldk + (pos ^ sw) * 8) * 2u);
    voff[1] = (int)((uint32_t)(row * p.ldv + (pos ^ sw) * 8) * 2u);
  }
  dw[0] = l31 - 4 * h;
  dw[1] = (int)(lds0 + DQ_WORDS_LDS);
  const int k16b = __builtin_amdgcn_readfirstlane((int)(p.ldk * 32)), v16b = __builtin_amdgcn_readfirstlane((int)(p.ldv * 32));
  const uint32_t waveb = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 1024u);
  const float c = p.scale * LOG2E;
  const int ngroups = p.nkv * p.B;
  const uint32_t per_group = (uint32_t)(p.rep * nch);
  const long nrows = (long)p.B * p.nq * p.Sq;
  auto udiv = [](uint32_t n, uint32_t magic, uint32_t d) { return d == 1u ? n : __umulhi(n, magic); };

  // One query block = one pass of the generated loop, software-pipelined as in the forward: as soon as every wave has left the loop
  // of block n, the first K / V tiles of block n + 1 are requested, and their latency runs under block n's epilogue (dQ stores) and
  // block n + 1's row loads.
  struct Blk { int item, half, x, hq, b, tend, tfirst; uint32_t mb_lo, mb_hi, kb_lo, kb_hi, vb_lo, vb_hi; bool valid; };
  auto decode = [&](int item, int half, Blk& d) {
    d.valid = item < nitems;
    d.item = item; d.half = half;
    if (!d.valid) return;
    uint32_t g, j;
    if ((ngroups & 7) == 0) { const uint32_t slot = (uint32_t)item >> 3, gq = udiv(slot, dv.m_pg, per_group); g = gq * 8u + ((uint32_t)item & 7u); j = slot - gq * per_group; }
    else { g = udiv((uint32_t)item, dv.m_pg, per_group); j = (uint32_t)item - g * per_group; }
    const uint32_t hr = udiv(j, dv.m_nch, nch), ch = j - hr * nch;
    const uint32_t bb = udiv(g, dv.m_nkv, p.nkv), kvh = g - bb * p.nkv;
    d.hq = (int)(kvh * p.rep + hr); d.b = (int)bb;
    d.x = half == 0 ? nx - 1 - (int)ch : (int)ch;
    d.tend = __builtin_amdgcn_readfirstlane(min(ntiles, 4 * d.x + 4));
    const bf16_t* kb = p.k + (long)d.b * p.Sk * p.ldk + (long)kvh * 128;
    const bf16_t* vb = p.v + (long)d.b * p.Sk * p.ldv + (long)kvh * 128;
    d.kb_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)kb); d.kb_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)kb >> 32));
    d.vb_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)vb); d.vb_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)vb >> 32));
  };
  auto advance = [&](const Blk& c, Blk& n) {
    const int chc = c.half == 0 ? nx - 1 - c.x : c.x;
    if (c.half == 0 && nx - 1 - chc != chc) decode(c.item, 1, n); else decode(c.item + (int)gridDim.x, 0, n);
  };
  // key state of a sample: one word per tile (only with a key mask).  Caller: every wave has left the loop.
  auto key_state = [&](Blk& d) {
    d.tfirst = 0; d.mb_lo = 0u; d.mb_hi = 0u;
    const uint8_t* km = p.kmask ? p.kmask + (long)d.b * p.Sk : nullptr;
    if (km != nullptr) {
      for (int t = wave; t < ntiles; t += 4) {
        const unsigned long long wv = __ballot(km[t * KT + lane] != 0);
        if (lane == 0) words[t] = wv;
      }
      __syncthreads();
      const unsigned long long wv = lane < ntiles ? words[lane] : ~0ull;
      const unsigned long long anym = __ballot(lane < ntiles && wv != 0ull), partm = __ballot(lane < ntiles && wv != ~0ull);
      d.tfirst = anym ? __builtin_ctzll(anym) : ntiles;
      d.mb_lo = (uint32_t)partm; d.mb_hi = (uint32_t)(partm >> 32);
    }
    d.tfirst = __builtin_amdgcn_readfirstlane(d.tfirst);
    d.mb_lo = __builtin_amdgcn_readfirstlane(d.mb_lo); d.mb_hi = __builtin_amdgcn_readfirstlane(d.mb_hi);
  };
  // the first tiles' LDS-DMA of block d.  Caller: the K / V rings are free.
  auto request = [&](const Blk& d) {
    if (!(d.valid && d.tfirst < d.tend)) return;
    asm volatile(
        "s_mov_b32 s36, %[kbl]\n\ts_mov_b32 s37, %[kbh]\n\ts_mov_b32 s38, %[vbl]\n\ts_mov_b32 s39, %[vbh]\n\t"
        "s_mov_b32 s40, %[k16]\n\ts_mov_b32 s41, %[v16]\n\ts_mov_b32 s43, %[tend]\n\ts_mov_b32 s45, %[tfirst]\n\ts_mov_b32 s57, %[waveb]\n\t"
        UR_ATTN_DQ_C128_DMA_ASM
        :
        : "{v[12:13]}"(voff), [kbl] "s"(d.kb_lo), [kbh] "s"(d.kb_hi), [vbl] "s"(d.vb_lo), [vbh] "s"(d.vb_hi), [k16] "s"(k16b), [v16] "s"(v16b),
          [tend] "s"(d.tend), [tfirst] "s"(d.tfirst), [waveb] "s"(waveb)
        : UR_ATTN_DQ_C128_DMA_CLOBBERS);
  };

  Blk cur, nxt;
  decode((int)blockIdx.x, 0, cur);
  if (cur.valid) { key_state(cur); __syncthreads(); request(cur); }
  while (cur.valid) {
    const int x = cur.x, hq = cur.hq, b = cur.b, tend = cur.tend, tfirst = cur.tfirst;
    // Lane-derived values of the C++ parts are rebuilt from an opaque copy of the lane id EVERY block: hoisted out of this loop they
    // would have to survive the generated statement, which leaves the compiler 8 vector registers -- it spilled ~100 of them to
    // scratch and reloaded each behind an s_waitcnt vmcnt(0) in the middle of the row loads and the dQ stores
    int lane_b = lane;
    asm volatile("" : "+v"(lane_b));
    const int lane = lane_b, h = lane >> 5, l31 = lane & 31;
#if UR_C128_STAMPS
    const unsigned long long st0 = __builtin_readcyclecounter();
    unsigned long long sa = 0, sb = 0, sc = 0;
#endif
    {
      const int q0 = 256 * x + 64 * wave;
      // Rows of this wave: q, dO, O; delta = sum_d dO O and the forward's statistics.  The MFMA fragment layout (lane = row) would
      // fetch 32 rows x 32 bytes per load instruction -- the texture path then spends ~24 k cycles per block on the 48 loads of a
      // wave.  The rows are read COALESCED instead (instruction i: rows 4 i .. + 3, 16 lanes x 16 bytes per row), delta is reduced
      // in that layout, and q / dO turn into fragments through the ring slots the first-tile request leaves free (K / V slots 1..3:
      // 12 KiB per wave in each ring; rows >= Sq read the last row -- their LSE2 = +inf zeroes every probability).
      bf16x8 qf[2][8], dof[2][8];
      f32x4 ld;                       // LSE2[qb 0, 1], delta[qb 0, 1]
      {
        const int r4 = lane >> 4, c16 = lane & 15;
        char* stA = smem + 16384 + wave * 12288;
        char* stB = smem + VBASE_LDS + 16384 + wave * 12288;
        float* rowc = reinterpret_cast<float*>(stA + 8192);       // [0, 64): delta, [64, 128): LSE2 of the wave's rows
        // every address = uniform base of the (batch, head) + a 32-bit lane offset (row stride x row + chunk)
        const bf16_t* qbase = p.q + (long)b * p.Sq * p.ldq + (long)hq * 128;
        const bf16_t* dobase = p.dout + (long)b * p.Sq * p.lddo + (long)hq * 128;
        const bf16_t* obase = p.o + (long)b * p.Sq * p.ldo + (long)hq * 128;
        auto ubase = [](const bf16_t* ptr) {
          const uint64_t v = reinterpret_cast<uint64_t>(ptr);
          uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
          asm volatile("" : "+s"(lo), "+s"(hi));
          return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
        };
        const char* qb8 = ubase(qbase); const char* dob8 = ubase(dobase); const char* ob8 = ubase(obase);
        typedef uint32_t raw4_t __attribute__((ext_vector_type(4)));
        auto ldrow = [&](const char* base8, long ld, int i) {
          const uint32_t off = (uint32_t)min(q0 + 4 * i + r4, p.Sq - 1) * (uint32_t)(ld * 2) + (uint32_t)(16 * c16);
          const raw4_t r = *(const __attribute__((address_space(1))) raw4_t*)(base8 + off);      // global_load ... s[base] (not FLAT)
          return make_uint4(r[0], r[1], r[2], r[3]);
        };
        uint4 qr[16], dr[16], ovr[16];
        float dl[16];
        // all rows requested in consumption order (q, dO, O, this lane's row statistics: loads return in order); q turns into
        // fragments while the rest flies
#pragma unroll
        for (int i = 0; i < 16; ++i) qr[i] = ldrow(qb8, p.ldq, i);
#pragma unroll
        for (int i = 0; i < 16; ++i) dr[i] = ldrow(dob8, p.lddo, i);
#pragma unroll
        for (int i = 0; i < 16; ++i) ovr[i] = ldrow(ob8, p.ldo, i);
        const int rl = 4 * c16 + r4, row = q0 + rl;             // this lane's row of the wave's 64 (delta / LSE bookkeeping)
        const bool ok = row < p.Sq;
        const long srow = ((long)b * p.nq + hq) * p.Sq + (ok ? row : 0);
        const float st_m = p.stats[srow * 2], st_inv = p.stats[srow * 2 + 1];
        __builtin_amdgcn_sched_barrier(0);
        // q: rows 0..31 through stA, rows 32..63 through stB (tile layout of the K ring: Cfg<128>::off)
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<uint4*>((i < 8 ? stA : stB) + C::off(4 * (i & 7) + r4, c16)) = qr[i];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          qf[0][ks] = *reinterpret_cast<const bf16x8*>(stA + C::off(l31, 2 * ks + h));
          qf[1][ks] = *reinterpret_cast<const bf16x8*>(stB + C::off(l31, 2 * ks + h));
        }
        __builtin_amdgcn_sched_barrier(0);
        {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const uint32_t dw4[4] = {dr[i].x, dr[i].y, dr[i].z, dr[i].w}, ow4[4] = {ovr[i].x, ovr[i].y, ovr[i].z, ovr[i].w};
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
            float acc = 0.f;                     // v_dot2c_f32_bf16: two bf16 products per instruction, f32 accumulation
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, dw4[e]), __builtin_bit_cast(bf2_t, ow4[e]), acc, false);
            dl[i] = acc;
          }
        }
        // sum over the 16 lanes of a row (one DPP row): quad swaps, then the mirrored half and the mirrored row (the partial sums are
        // uniform within what has been summed already, so a mirror reaches the other half); lane c16 keeps the total of dl[c16]
        float mine = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float v = dl[i];
          v += dpp_f32<0xB1>(v);        // quad_perm [1, 0, 3, 2]
          v += dpp_f32<0x4E>(v);        // quad_perm [2, 3, 0, 1]
          v += dpp_f32<0x141>(v);       // row_half_mirror
          v += dpp_f32<0x140>(v);       // row_mirror
          mine = c16 == i ? v : mine;
        }
        dl[0] = mine;
        {
          const float dlt = dl[0];
          const float m = st_m, inv = ok ? st_inv : 0.f;
          const float lse = m - __logf(inv);                 // natural log of the row's normaliser, scaled scores
          if (ok) {
            float* ws = const_cast<float*>(p.delta);
            ws[srow] = -dlt;
            ws[nrows + srow] = (inv > 0.f) ? (p.lse_log2 ? -lse * LOG2E : -lse / p.scale) : NEG_INF;
          }
          rowc[rl] = dlt;
          rowc[64 + rl] = (inv > 0.f) ? lse * LOG2E : __builtin_huge_valf();
        }
        // dO through the same areas (the wave's LDS operations execute in order: its q fragment reads precede these writes)
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<uint4*>((i < 8 ? stA : stB) + C::off(4 * (i & 7) + r4, c16)) = dr[i];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          dof[0][ks] = *reinterpret_cast<const bf16x8*>(stA + C::off(l31, 2 * ks + h));
          dof[1][ks] = *reinterpret_cast<const bf16x8*>(stB + C::off(l31, 2 * ks + h));
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) { ld[qb] = rowc[64 + 32 * qb + l31]; ld[2 + qb] = rowc[32 * qb + l31]; }
      }
      f32x32 d0, d1, d2, d3;
      advance(cur, nxt);              // scalar work: its results wait in scalar registers / spill lanes while the loop runs
#if UR_C128_STAMPS
      unsigned int* dbg = g_c128_stamps + ((size_t)((cur.item * 2 + cur.half) & 8191) * 4 + wave) * 32;
      const uint32_t db_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dbg), db_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)dbg >> 32));
      const unsigned long long st1 = __builtin_readcyclecounter(), sr1 = wall_clock64();
      unsigned long long st2 = st1;
#endif
      if (tfirst < tend) {
        const int tlast = __builtin_amdgcn_readfirstlane(q0 < p.Sq ? min(4 * x + wave, ntiles - 1) : -1);
        i32x32 qv0, qv1, dov0, dov1;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            qv0[4 * ks + jj] = (int)pack_bf2(bf2f((bf16_t)qf[0][ks][2 * jj]) * c, bf2f((bf16_t)qf[0][ks][2 * jj + 1]) * c);
            qv1[4 * ks + jj] = (int)pack_bf2(bf2f((bf16_t)qf[1][ks][2 * jj]) * c, bf2f((bf16_t)qf[1][ks][2 * jj + 1]) * c);
            dov0[4 * ks + jj] = (int)((uint32_t)(uint16_t)dof[0][ks][2 * jj] | ((uint32_t)(uint16_t)dof[0][ks][2 * jj + 1] << 16));
            dov1[4 * ks + jj] = (int)((uint32_t)(uint16_t)dof[1][ks][2 * jj] | ((uint32_t)(uint16_t)dof[1][ks][2 * jj + 1] << 16));
          }
        __syncthreads();        // the statement's entry LDS-DMA fills ring slots 1 and 2: every wave is done with its staging area there
        asm volatile(
            "s_mov_b32 s36, %[kbl]\n\ts_mov_b32 s37, %[kbh]\n\ts_mov_b32 s38, %[vbl]\n\ts_mov_b32 s39, %[vbh]\n\t"
            "s_mov_b32 s40, %[k16]\n\ts_mov_b32 s41, %[v16]\n\ts_mov_b32 s43, %[tend]\n\ts_mov_b32 s44, %[tlast]\n\t"
            "s_mov_b32 s45, %[tfirst]\n\ts_mov_b32 s46, %[mbl]\n\ts_mov_b32 s47, %[mbh]\n\ts_mov_b32 s57, %[waveb]\n\t"
#if UR_C128_STAMPS
            "s_mov_b32 s72, %[dbl]\n\ts_mov_b32 s73, %[dbh]\n\t"
#endif
            UR_ATTN_DQ_C128_ASM
            : "=&{a[0:31]}"(d0), "=&{a[32:63]}"(d1), "=&{a[64:95]}"(d2), "=&{a[96:127]}"(d3)
            : "{a[128:159]}"(qv0), "{a[160:191]}"(qv1), "{a[192:223]}"(dov0), "{a[224:255]}"(dov1), "{v[240:241]}"(kava), "{v[242:249]}"(tatb),
              "{v[250:253]}"(ld), "{v[8:9]}"(dw), "{v[12:13]}"(voff),
              [kbl] "s"(cur.kb_lo), [kbh] "s"(cur.kb_hi), [vbl] "s"(cur.vb_lo), [vbh] "s"(cur.vb_hi), [k16] "s"(k16b), [v16] "s"(v16b), [tend] "s"(tend),
              [tlast] "s"(tlast), [tfirst] "s"(tfirst), [mbl] "s"(cur.mb_lo), [mbh] "s"(cur.mb_hi), [waveb] "s"(waveb)
#if UR_C128_STAMPS
              , [dbl] "s"(db_lo), [dbh] "s"(db_hi)
#endif
            : UR_ATTN_DQ_C128_CLOBBERS);
#if UR_C128_STAMPS
        st2 = __builtin_readcyclecounter();
#endif
      } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) { d0[i] = 0.f; d1[i] = 0.f; d2[i] = 0.f; d3[i] = 0.f; }
      }
      // the next block (decoded before the loop): once every wave has left the loop its key state is built and its first tiles are requested
      // (the first 32 rows' dQ leave BEFORE that barrier: the waves with fewer diagonal tiles leave the loop up to ~4.7 k cycles early and
      // spend that wait on their own stores; the request then flies under the second 32 rows' stores and the next block's row loads)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int row = q0 + 32 * qb + l31;
        const bool ok = row < p.Sq;
        f32x16 acc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int idx = 16 * (dt & 1) + r;
            acc[dt][r] = (qb == 0 ? (dt < 2 ? d0[idx] : d1[idx]) : (dt < 2 ? d2[idx] : d3[idx])) * p.scale;
          }
        dq_store_block<128>(p, acc, b, hq, row, ok, lane);
        if (qb == 0) {
          asm volatile("" ::: "memory");
#if UR_C128_STAMPS
          sa = __builtin_readcyclecounter();
#endif
          __syncthreads();                   // every wave has left the loop: the key-state words and the rings are free
#if UR_C128_STAMPS
          sb = __builtin_readcyclecounter();
#endif
          if (nxt.valid) {
            if (nxt.item != cur.item) { key_state(nxt); if (p.kmask) __syncthreads(); }
            else { nxt.tfirst = cur.tfirst; nxt.mb_lo = cur.mb_lo; nxt.mb_hi = cur.mb_hi; }
          }
          request(nxt);
#if UR_C128_STAMPS
          sc = __builtin_readcyclecounter();
#endif
        }
      }
#if UR_C128_STAMPS
      if (lane == 0) {
        const unsigned long long st3 = __builtin_readcyclecounter(), sr3 = wall_clock64();
        dbg[12] = (unsigned int)(st1 - st0); dbg[13] = (unsigned int)(st2 - st1); dbg[14] = (unsigned int)(st3 - st2); dbg[15] = (unsigned int)x;
        dbg[16] = (unsigned int)(sa - st2); dbg[17] = (unsigned int)(sb - sa); dbg[18] = (unsigned int)(sc - sb); dbg[19] = (unsigned int)(st3 - sc); dbg[20] = 0;
        dbg[21] = (unsigned int)(st3 - st1); dbg[22] = (unsigned int)(sr3 - sr1);      // shader cycles and 100 MHz ticks of the same span: the clock
      }
#endif
    }
    cur = nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA pieces of the last loop iterations: landed before the workgroup gives its LDS back
}

// ================================================================================================
// Causal head_dim-128 backward dK / dV, hand-scheduled (tools/asmgen/attn_dkv.py -> gen/attn_dkv_c128_asm.h; CPU-emulated by
// tests/test_asmgen_attn_dkv.py).  A workgroup = 128 keys of one (batch, kv head), one wave per SIMD, 32 keys per wave (key = MFMA lane);
// per 64-query tile 64 MFMAs in the order X_a X_b Y_a Y_b with the vector work of a half in the gaps of the following block; Q / dO
// tiles and the row constants (-delta, -LSE2 from attn_bwd_dq_c128_kernel) arrive by LDS-DMA three tiles ahead.  k is pre-scaled by
// scale*log2e here, so P = exp2(S') is one instruction per score.  Padded keys are zeroed at the store (their lanes never mix).
// LDS: 4 ring slots x (1 KiB row constants | Q tile 16 KiB | dO tile 16 KiB).
namespace c128 { constexpr int DKV_SLOT = 33792, DKV_RING_BYTES = 4 * DKV_SLOT, DKV_LDS_BYTES = DKV_RING_BYTES + 16, DKV_HIGH = 2 * DKV_SLOT; }      // + the drawn queue item

// persist = 0: one workgroup per (batch, kv head, 128-key block), dealt by block_map.  persist = 1 (launch_dkv: one workgroup per CU,
// the groups a multiple of 8): the workgroup DRAWS its key blocks from the queue of its XCD lane (g_dkv_queue[slot][id & 7], zeroed
// by the call's dQ kernel) in block_map's order -- heaviest blocks first, the blocks of a (batch, kv head) group on one XCD at about
// the same time -- so the sweep balances under any padding like the hardware dispatcher's, but no workgroup launch (LDS allocation,
// kernel-argument loads, wave start) sits between two key blocks: 5 % of the kernel on dense inputs.  (A static rotation of the
// block sizes over the workgroups was as fast on dense inputs and 1.5 ms per step slower on the padded batch.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dkv_c128_kernel(AttnP p, int persist) {
  using namespace c128;
  using C = Cfg<128>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nx = p.Sk / 128;
  const int per_lane = persist ? (p.nkv * p.B / 8) * nx : 1;
  unsigned int* qitem = reinterpret_cast<unsigned int*>(smem + DKV_RING_BYTES);
  const uint32_t lds0 = lds_off(smem);
  const float c = p.scale * LOG2E;
  const long nrows = (long)p.B * p.nq * p.Sq;
  const int q16b = __builtin_amdgcn_readfirstlane((int)(p.ldq * 32)), d16b = __builtin_amdgcn_readfirstlane((int)(p.lddo * 32));
  const int sq = __builtin_amdgcn_readfirstlane(p.Sq);
  const uint32_t wsel = __builtin_amdgcn_readfirstlane((wave & 1) ? 0u : (uint32_t)(nrows * 4));
  const uint32_t cwave = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(wave & 1) * 256u), waveb = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 1024u);

  // The key blocks of this workgroup are software-pipelined: as soon as every wave has left the loop of block n, block n + 1 is drawn,
  // its first tile's LDS-DMA is requested, and block n's results are stored (with the k-norm + RoPE backward) under that latency; block
  // n + 1's K / V rows are loaded and packed at the top of its own pass (requested before the stores they had to sit in registers
  // across them: hipcc spilled each load the moment it was issued -- sixteen exposed round trips; packed but carried over the back
  // edge they went through scratch as well).  Carried into the next pass: the scalars of block n + 1 only.
  struct Blk { int kvh, b, kb, qstart, ntot; uint32_t qb_lo, qb_hi, do_lo, do_hi, ws_lo, ws_hi; bool more, run; };
  // draw a key block, request its first tile and its K / V rows.  Caller: every wave has left the previous loop (persist: the barriers
  // inside; the first draw: nothing has run yet).
  // (macros, not lambdas: through a by-reference capture hipcc keeps kf / vf in scratch memory)
#define UR_DKV_DRAW(d, first) do { \
    int ln = tid & 63; \
    asm volatile("" : "+v"(ln)); \
    int id = (int)blockIdx.x; \
    d.more = first; d.run = false; \
    if (persist) { \
      __syncthreads(); \
      if (tid == 0) *qitem = atomicAdd(&p.queue[blockIdx.x & 7], 1u); \
      __syncthreads(); \
      const unsigned int t = *qitem; \
      d.more = t < (unsigned int)per_lane; \
      id = (int)(t * 8u + (blockIdx.x & 7u)); \
    } \
    if (!d.more) break; \
    const BlockMap bm = block_map_id<false>(id, nx, 1, p.nkv, p.B); \
    d.kvh = bm.head; d.b = bm.b; d.kb = 128 * bm.x + 32 * wave; \
    const long ktok = (long)d.b * p.Sk + d.kb + (ln & 31); \
    const bool kvalid = p.kmask == nullptr || p.kmask[ktok] != 0; \
 \
    d.run = p.kmask == nullptr || __syncthreads_or(kvalid ? 1 : 0) != 0; \
    if (!d.run) break; \
    i32x2 voff; \
    { \
      const int row = 4 * wave + (ln >> 4), pos = ln & 15; \
      const int sw = ((row & 3) << 2) | ((row >> 2) & 3); \
      voff[0] = (int)((uint32_t)(row * p.ldq + (pos ^ sw) * 8) * 2u); \
      voff[1] = (int)((uint32_t)(row * p.lddo + (pos ^ sw) * 8) * 2u); \
    } \
    const int voffc = 4 * ln; \
    const int hq0 = d.kvh * p.rep; \
    auto sgpr64 = [](const void* ptr) { return (unsigned long long)(uintptr_t)ptr; }; \
    const unsigned long long qb_s = sgpr64(p.q + (long)d.b * p.Sq * p.ldq + (long)hq0 * 128), dob_s = sgpr64(p.dout + (long)d.b * p.Sq * p.lddo + (long)hq0 * 128), \
                             wsb_s = sgpr64(p.delta + ((long)d.b * p.nq + hq0) * p.Sq); \
    d.qb_lo = __builtin_amdgcn_readfirstlane((uint32_t)qb_s); d.qb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(qb_s >> 32)); \
    d.do_lo = __builtin_amdgcn_readfirstlane((uint32_t)dob_s); d.do_hi = __builtin_amdgcn_readfirstlane((uint32_t)(dob_s >> 32)); \
    d.ws_lo = __builtin_amdgcn_readfirstlane((uint32_t)wsb_s); d.ws_hi = __builtin_amdgcn_readfirstlane((uint32_t)(wsb_s >> 32)); \
    d.qstart = 128 * bm.x; \
    d.ntot = __builtin_amdgcn_readfirstlane(((p.Sq - d.qstart) / KT) * p.rep); \
    asm volatile( \
        "s_mov_b32 s36, %[qbl]\n\ts_mov_b32 s37, %[qbh]\n\ts_mov_b32 s38, %[dol]\n\ts_mov_b32 s39, %[doh]\n\ts_mov_b32 s40, %[wsl]\n\ts_mov_b32 s41, %[wsh]\n\t" \
        "s_mov_b32 s42, %[q16]\n\ts_mov_b32 s43, %[d16]\n\ts_mov_b32 s58, %[qstart]\n\t" \
        "s_lshl_b32 s60, %[sq], 2\n\ts_mov_b32 s34, %[wsel]\n\ts_mov_b32 s56, %[cwave]\n\ts_mov_b32 s57, %[waveb]\n\t" \
        UR_ATTN_DKV_C128_DMA_ASM \
        : \
        : "{v[8:9]}"(voff), "{v10}"(voffc), \
          [qbl] "s"(__builtin_amdgcn_readfirstlane(d.qb_lo)), [qbh] "s"(__builtin_amdgcn_readfirstlane(d.qb_hi)), [dol] "s"(__builtin_amdgcn_readfirstlane(d.do_lo)), [doh] "s"(__builtin_amdgcn_readfirstlane(d.do_hi)), [wsl] "s"(__builtin_amdgcn_readfirstlane(d.ws_lo)), [wsh] "s"(__builtin_amdgcn_readfirstlane(d.ws_hi)), [q16] "s"(q16b), [d16] "s"(d16b), \
          [qstart] "s"(__builtin_amdgcn_readfirstlane(d.qstart)), [sq] "s"(sq), [wsel] "s"(wsel), [cwave] "s"(cwave), [waveb] "s"(waveb) \
        : UR_ATTN_DKV_C128_DMA_CLOBBERS); \
  } while (0)
  // K~ = k * scale * log2(e) (rounded to bf16 once more) and V fragments of this lane's key
#define UR_DKV_PACK(d) do { \
    int lp = tid & 63; \
    asm volatile("" : "+v"(lp)); \
    const long ktokp = (long)d.b * p.Sk + d.kb + (lp & 31); \
    const bf16_t* krow = p.k + ktokp * p.ldk + (long)d.kvh * 128; \
    const bf16_t* vrow = p.v + ktokp * p.ldv + (long)d.kvh * 128; \
    bf16x8 kf[8], vf[8]; \
_Pragma("unroll") \
    for (int ks = 0; ks < 8; ++ks) { kf[ks] = g_frag(krow, ks, lp, true); vf[ks] = g_frag(vrow, ks, lp, true); } \
_Pragma("unroll") \
    for (int ks = 0; ks < 8; ++ks) \
_Pragma("unroll") \
      for (int jj = 0; jj < 4; ++jj) { \
        kv_[4 * ks + jj] = (int)pack_bf2(bf2f((bf16_t)kf[ks][2 * jj]) * c, bf2f((bf16_t)kf[ks][2 * jj + 1]) * c); \
        vv_[4 * ks + jj] = (int)((uint32_t)(uint16_t)vf[ks][2 * jj] | ((uint32_t)(uint16_t)vf[ks][2 * jj + 1] << 16)); \
      } \
  } while (0)
  Blk cur, nxt;
#if UR_C128_STAMPS
  unsigned long long acc_pack = 0, acc_asm = 0, acc_draw = 0, acc_dv = 0, acc_dk = 0, nblk = 0;
#endif
  UR_DKV_DRAW(cur, true);
  while (cur.more) {
    c128::f32x32 k0, k1, v0, v1;
#if UR_C128_STAMPS
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t1 = t0, t2 = t0;
#endif
    if (cur.run) {
      i32x32 kv_, vv_;
      UR_DKV_PACK(cur);
#if UR_C128_STAMPS
      asm volatile("" :: "v"(kv_), "v"(vv_));
      t1 = __builtin_readcyclecounter();
#endif
      int ln = tid & 63;
      asm volatile("" : "+v"(ln));
      const int h = ln >> 5, l31 = ln & 31;
      i32x2 ra, ca, voff;
      i32x8 ta, tb;
      ra[0] = (int)(lds0 + C::off(l31, h)); ra[1] = ra[0] + DKV_HIGH;
      ca[0] = (int)(lds0 + 16 * h); ca[1] = ca[0] + DKV_HIGH;
      {
        const int g16 = (ln >> 4) & 1, i = ln & 15;
        const int row = 4 * h + (i >> 2), sub8 = 8 * (i & 1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int ch = 4 * dt + 2 * g16 + ((i & 3) >> 1);
          ta[dt] = (int)(lds0 + C::off(row, ch) + sub8); ta[4 + dt] = ta[dt] + DKV_HIGH;
          tb[dt] = (int)(lds0 + C::off(row + 8, ch) + sub8); tb[4 + dt] = tb[dt] + DKV_HIGH;
        }
      }
      {
        const int row = 4 * wave + (ln >> 4), pos = ln & 15;
        const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
        voff[0] = (int)((uint32_t)(row * p.ldq + (pos ^ sw) * 8) * 2u);
        voff[1] = (int)((uint32_t)(row * p.lddo + (pos ^ sw) * 8) * 2u);
      }
      const int voffc = 4 * ln, xdiag = l31 - 4 * h;
      const int kb_s = __builtin_amdgcn_readfirstlane(cur.kb);
#if UR_C128_STAMPS
      // (the generated loop's own accumulators land behind the C++ ones: words 8..13 of the wave's record; one block per workgroup
      // -- UR_ATTN_DKV_PERSIST=0 -- gives one record per block)
      unsigned int* dbg_asm = g_c128_stamps + ((size_t)(blockIdx.x & 8191) * 4 + wave) * 32 + 8;
      const uint32_t db_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dbg_asm), db_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)dbg_asm >> 32));
#endif
      asm volatile(
          "s_mov_b32 s36, %[qbl]\n\ts_mov_b32 s37, %[qbh]\n\ts_mov_b32 s38, %[dol]\n\ts_mov_b32 s39, %[doh]\n\ts_mov_b32 s40, %[wsl]\n\ts_mov_b32 s41, %[wsh]\n\t"
          "s_mov_b32 s42, %[q16]\n\ts_mov_b32 s43, %[d16]\n\ts_mov_b32 s45, %[ntot]\n\ts_mov_b32 s47, %[kb]\n\ts_mov_b32 s58, %[qstart]\n\t"
          "s_mov_b32 s35, %[sq]\n\ts_lshl_b32 s60, %[sq], 2\n\ts_mov_b32 s34, %[wsel]\n\ts_mov_b32 s56, %[cwave]\n\ts_mov_b32 s57, %[waveb]\n\t"
#if UR_C128_STAMPS
          "s_mov_b32 s72, %[dbl]\n\ts_mov_b32 s73, %[dbh]\n\t"
#endif
          UR_ATTN_DKV_C128_ASM
          : "=&{a[0:31]}"(k0), "=&{a[32:63]}"(k1), "=&{a[64:95]}"(v0), "=&{a[96:127]}"(v1)
          : "{a[128:159]}"(kv_), "{a[160:191]}"(vv_), "{v[8:9]}"(voff), "{v10}"(voffc), "{v[144:145]}"(ra), "{v[146:153]}"(ta), "{v[154:161]}"(tb),
            "{v[162:163]}"(ca), "{v164}"(xdiag),
            [qbl] "s"(__builtin_amdgcn_readfirstlane(cur.qb_lo)), [qbh] "s"(__builtin_amdgcn_readfirstlane(cur.qb_hi)), [dol] "s"(__builtin_amdgcn_readfirstlane(cur.do_lo)), [doh] "s"(__builtin_amdgcn_readfirstlane(cur.do_hi)), [wsl] "s"(__builtin_amdgcn_readfirstlane(cur.ws_lo)), [wsh] "s"(__builtin_amdgcn_readfirstlane(cur.ws_hi)), [q16] "s"(q16b),
            [d16] "s"(d16b), [ntot] "s"(__builtin_amdgcn_readfirstlane(cur.ntot)), [kb] "s"(kb_s), [qstart] "s"(__builtin_amdgcn_readfirstlane(cur.qstart)), [sq] "s"(sq), [wsel] "s"(wsel), [cwave] "s"(cwave), [waveb] "s"(waveb)
#if UR_C128_STAMPS
            , [dbl] "s"(db_lo), [dbh] "s"(db_hi)
#endif
          : UR_ATTN_DKV_C128_CLOBBERS);
#if UR_C128_STAMPS
      t2 = __builtin_readcyclecounter();
#endif
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) { k0[i] = 0.f; k1[i] = 0.f; v0[i] = 0.f; v1[i] = 0.f; }
    }
    // the next block: drawn, requested
    nxt.more = false; nxt.run = false;
    if (persist) UR_DKV_DRAW(nxt, false);
#if UR_C128_STAMPS
    const unsigned long long t3 = __builtin_readcyclecounter();
    unsigned long long t4 = t3;
#endif
    // this block's results
    {
      int le = tid & 63;
      asm volatile("" : "+v"(le));
      const long ktok = (long)cur.b * p.Sk + cur.kb + (le & 31);
      const bool kvalid = p.kmask == nullptr || p.kmask[ktok] != 0;       // padded key: whatever its lane computed is dropped
      // (dV first, then dK with its RoPE backward: one 64-register accumulator at a time beside the next block's K / V rows in flight)
      {
        f32x16 dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const int idx = 16 * (dt & 1) + r; dv[dt][r] = kvalid ? (dt < 2 ? v0[idx] : v1[idx]) : 0.f; }
        store_T<128>(p.dv + ktok * p.lddv + (long)cur.kvh * 128, dv, 1.0f, le, true);
      }
      asm volatile("" ::: "memory");
#if UR_C128_STAMPS
      t4 = __builtin_readcyclecounter();
#endif
      {
        f32x16 dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const int idx = 16 * (dt & 1) + r; dk[dt][r] = kvalid ? (dt < 2 ? k0[idx] : k1[idx]) * p.scale : 0.f; }
        if (p.rk_src != nullptr && cur.run) {
          // k = rope(k_norm(k_raw)) ran in the q|k|v GEMM's epilogue: the lane owns its key's whole row, so dK leaves as the gradient of
          // the RAW k projection (the arithmetic of qknorm_rope_bwd_roped_kernel; a padded key's zero gradient stays zero)
          const int key_e = cur.kb + (le & 31);
          rope_bwd_from_roped<128>(dk, p.rk_src + ktok * p.rk_ld + (long)cur.kvh * 128, p.rk_w, p.rp_cos + (long)key_e * 64, p.rp_sin + (long)key_e * 64,
                                   p.rp_rstd[ktok * p.rp_rstd_ld + p.rk_rstd_h0 + cur.kvh], le >> 5);
        }
        if (p.rk_src != nullptr) store_T<128>(p.rk_dst + ktok * p.rk_lddst + (long)cur.kvh * 128, dk, 1.0f, le, true);
        else store_T<128>(p.dk + ktok * p.lddk + (long)cur.kvh * 128, dk, 1.0f, le, true);
      }
    }
#if UR_C128_STAMPS
    { const unsigned long long t5 = __builtin_readcyclecounter();
      acc_pack += t1 - t0; acc_asm += t2 - t1; acc_draw += t3 - t2; acc_dv += t4 - t3; acc_dk += t5 - t4; nblk += 1; }
#endif
    cur = nxt;
  }
#if UR_C128_STAMPS
  if ((tid & 63) == 0 && blockIdx.x < 8192) {
    unsigned int* dbg = g_c128_stamps + ((size_t)blockIdx.x * 4 + wave) * 32;
    dbg[0] = (unsigned int)acc_pack; dbg[1] = (unsigned int)acc_asm; dbg[2] = (unsigned int)acc_draw; dbg[3] = (unsigned int)acc_dv; dbg[4] = (unsigned int)acc_dk;
    dbg[5] = (unsigned int)nblk; dbg[6] = 0xD0C5u;
  }
#endif
#undef UR_DKV_DRAW
#undef UR_DKV_PACK
}

// ================================================================================================
template <int HD> constexpr int fwd_smem() { return 4 * Cfg<HD>::TILE + MAX_KTILES * 16; }
template <int HD> constexpr int dkv_smem() { return 2 * (2 * Cfg<HD>::TILE + 5 * KT * (int)sizeof(float)); }

template <typename K>
int set_smem(K kern, int bytes, const char* name) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) UR_FAIL((int)e, "%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
  return 0;
}

int fill(AttnP& p, const ur_attn_args* a) {
  UR_REQUIRE(a && a->q && a->k && a->v && a->stats, "ur_attn: null argument");
  UR_REQUIRE(a->B >= 0 && a->Sq > 0 && a->Sk > 0 && a->nq > 0 && a->nkv > 0 && (a->nq % a->nkv) == 0, "ur_attn: bad sizes");
  UR_REQUIRE(a->head_dim == 64 || a->head_dim == 128, "ur_attn: head_dim must be 64 or 128 (got %d)", a->head_dim);
  UR_REQUIRE((a->ldq % 8) == 0 && (a->ldk % 8) == 0 && (a->ldv % 8) == 0 && UR_ALIGNED16(a->q) && UR_ALIGNED16(a->k) && UR_ALIGNED16(a->v),
             "ur_attn: q/k/v need 16-byte aligned rows");
  UR_REQUIRE(a->ldq >= (int64_t)a->nq * a->head_dim && a->ldk >= (int64_t)a->nkv * a->head_dim && a->ldv >= (int64_t)a->nkv * a->head_dim,
             "ur_attn: token stride smaller than heads*head_dim");
  UR_REQUIRE(!a->causal || a->Sq == a->Sk, "ur_attn: causal mode needs Sq == Sk");
  UR_REQUIRE(a->dropout_p >= 0.f && a->dropout_p < 1.f && (!a->causal || a->dropout_p == 0.f), "ur_attn: bad dropout");
  UR_REQUIRE(a->scale > 0.f, "ur_attn: scale must be positive");
  UR_REQUIRE(a->Sk <= MAX_KTILES * KT, "ur_attn: Sk (%d) exceeds %d", a->Sk, MAX_KTILES * KT);
  memset(&p, 0, sizeof(p));
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.v = (const bf16_t*)a->v; p.o = (bf16_t*)a->o; p.stats = a->stats;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.kmask = a->key_mask;
  p.B = a->B; p.Sq = a->Sq; p.Sk = a->Sk; p.nq = a->nq; p.nkv = a->nkv; p.rep = a->nq / a->nkv;
  p.scale = a->scale; p.seed = a->seed;
  p.drow0 = (uint64_t)a->drop_batch0 * (uint64_t)a->nq * (uint64_t)a->Sq;
  p.drop_thr = a->dropout_p > 0.f ? std::max(1u, ur_drop_threshold16(a->dropout_p)) : 0u;
  p.drop_inv = a->dropout_p > 0.f ? 1.0f / (1.0f - a->dropout_p) : 1.0f;
  return 0;
}

// the hand-scheduled causal head_dim-128 BACKWARD pair (dQ + dK/dV) runs together or not at all (they share the -LSE2 plane)
inline bool c128_bwd_ok(const AttnP& p);
// CUs of the current device (persistent grids), cached per device
inline int device_cu_count() { return ur_device_cu_count(); }
// UR_ATTN_MODE_C128 = 0 sends the causal head_dim-128 launches back to the compiler-scheduled kernels
inline bool fwd_c128_enabled() { return attn_mode(UR_ATTN_MODE_C128) != 0; }
// UR_ATTN_MODE_DKV_PERSIST = 0 launches one workgroup per key block
inline bool dkv_persist_enabled() { return attn_mode(UR_ATTN_MODE_DKV_PERSIST) != 0; }
inline bool fwd_gq2_enabled() { static const bool on = ur_lab_int("UR_FWD_GQ2", 0) == 1; return on; }
template <int HD, bool CAUSAL, int NW>
int launch_fwd(const AttnP& p, hipStream_t st) {
  if constexpr (HD == 128 && CAUSAL && NW == 4) {
    if (p.rep == 2 && fwd_gq2_enabled()) {
      static std::atomic<uint64_t> once8{0};   // per device
      UR_ONCE_PER_DEVICE(once8) { int rc = set_smem(&attn_fwd_kernel<128, true, 8, true>, fwd_smem<128>(), "ur_attn_fwd(gq2)"); if (rc) return rc; }
      dim3 grid(ur_cdiv(p.Sq, 128) * (p.nq / 2) * p.B);
      hipLaunchKernelGGL((attn_fwd_kernel<128, true, 8, true>), grid, dim3(512), fwd_smem<128>(), st, p);
      UR_CHECK_LAUNCH("ur_attn_fwd(gq2)");
      return 0;
    }
  }
  if constexpr (HD == 128 && CAUSAL && NW == 4) {
    if (p.Sq == p.Sk && (p.Sk % KT) == 0 && p.Sk >= 128 && p.Sk <= c128::MAX_SK && fwd_c128_enabled() && (long)p.nq * p.B * 8 < (1L << 24) &&
        p.ldk * 2L * p.Sk < (1L << 31) && p.ldv * 2L * p.Sk < (1L << 31)) {
      static std::atomic<uint64_t> once_c{0};   // per device
      UR_ONCE_PER_DEVICE(once_c) { int rc = set_smem(&attn_fwd_c128_kernel, c128::LDS_BYTES, "ur_attn_fwd(c128)"); if (rc) return rc; }
      const int nx = ur_cdiv(p.Sq, 256), nch = (nx + 1) / 2, nitems = p.nq * p.B * nch;
      auto magic = [](uint32_t d) { return (uint32_t)(((1ull << 32) + d - 1) / d); };
      const C128Div dv{magic((uint32_t)(p.rep * nch)), magic((uint32_t)nch), magic((uint32_t)p.nkv)};
      hipLaunchKernelGGL(attn_fwd_c128_kernel, dim3(std::min(nitems, device_cu_count())), dim3(256), c128::LDS_BYTES, st, p, nitems, nch, dv);
      UR_CHECK_LAUNCH("ur_attn_fwd(c128)");
      return 0;
    }
  }
  static std::atomic<uint64_t> once{0};   // per device
  UR_ONCE_PER_DEVICE(once) { int rc = set_smem(&attn_fwd_kernel<HD, CAUSAL, NW>, fwd_smem<HD>(), "ur_attn_fwd"); if (rc) return rc; }
  dim3 grid(ur_cdiv(p.Sq, 32 * NW) * p.nq * p.B);
  // one K | V stage and one pair of key words when every key fits one tile (the Q-Formers' 32-query launches: twice the waves per CU)
  const int smem_bytes = p.Sk <= KT ? 2 * Cfg<HD>::TILE + 16 : fwd_smem<HD>();
  hipLaunchKernelGGL((attn_fwd_kernel<HD, CAUSAL, NW>), grid, dim3(NW * 64), smem_bytes, st, p);
  UR_CHECK_LAUNCH("ur_attn_fwd");
  return 0;
}
template <int HD, bool CAUSAL, int NW>
int launch_dq(const AttnP& p, hipStream_t st) {
  if constexpr (HD == 128 && CAUSAL && NW == 4) {
    if (c128_bwd_ok(p)) {
      static std::atomic<uint64_t> once_c{0};   // per device
      UR_ONCE_PER_DEVICE(once_c) { int rc = set_smem(&attn_bwd_dq_c128_kernel, c128::DQ_LDS_BYTES, "ur_attn_bwd(dq c128)"); if (rc) return rc; }
      const int nx = ur_cdiv(p.Sq, 256), nch = (nx + 1) / 2, nitems = p.nq * p.B * nch;
      auto magic = [](uint32_t d) { return (uint32_t)(((1ull << 32) + d - 1) / d); };
      const C128Div dv{magic((uint32_t)(p.rep * nch)), magic((uint32_t)nch), magic((uint32_t)p.nkv)};
      hipLaunchKernelGGL(attn_bwd_dq_c128_kernel, dim3(std::min(nitems, device_cu_count())), dim3(256), c128::DQ_LDS_BYTES, st, p, nitems, nch, dv);
      UR_CHECK_LAUNCH("ur_attn_bwd(dq c128)");
      return 0;
    }
  }
  static std::atomic<uint64_t> once{0};   // per device
  UR_ONCE_PER_DEVICE(once) { int rc = set_smem(&attn_bwd_dq_kernel<HD, CAUSAL, NW>, fwd_smem<HD>(), "ur_attn_bwd(dq)"); if (rc) return rc; }
  dim3 grid(ur_cdiv(p.Sq, 32 * NW) * p.nq * p.B);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, CAUSAL, NW>), grid, dim3(NW * 64), fwd_smem<HD>(), st, p);
  UR_CHECK_LAUNCH("ur_attn_bwd(dq)");
  return 0;
}
// UR_ATTN_MODE_FEWQ = 0 sends few-query shapes back to attn_bwd_dkv_kernel (one test process compares the two kernels bit for bit)
inline bool fewq_enabled() { return attn_mode(UR_ATTN_MODE_FEWQ) != 0; }

template <int HD, bool CAUSAL, int NW>
int launch_dkv(const AttnP& p, hipStream_t st) {
  if (HD == 64 && !CAUSAL && NW == 4 && p.rep == 1 && p.Sq <= KT && p.Sk >= 256 && fewq_enabled()) {
    // few queries, many keys: one workgroup per (batch, head) pair -- or per chunk of its key blocks while the pairs
    // alone do not fill the chip (>= 8 key blocks, i.e. two per wave, per workgroup)
    constexpr int SMF = 2 * Cfg<64>::TILE + 5 * KT * (int)sizeof(float) + 16 + 4 * 2 * 64 * (int)sizeof(float) + 256;      // + flag + the waves' column sums + one liveness byte per key block
    static_assert(MAX_KTILES * KT / 32 <= 256, "attn_bwd_dkv_fewq_kernel: one liveness byte per 32-key block in a 256-byte LDS tail (ur_attn caps Sk at MAX_KTILES * KT)");
    const int nblk = ur_cdiv(p.Sk, 32), pairs = p.nq * p.B;
    int nchunk = std::max(1, std::min(ur_cdiv(4096, pairs), nblk / 8));
    if (p.colsum_part != nullptr) nchunk = 1;          // the column sums leave as ONE partial per (batch, head): one workgroup per pair
    const int bpc = ur_cdiv(ur_cdiv(nblk, nchunk), 4) * 4;
    nchunk = ur_cdiv(nblk, bpc);
    hipLaunchKernelGGL((attn_bwd_dkv_fewq_kernel<64>), dim3(pairs * nchunk), dim3(256), SMF, st, p, nchunk, bpc);
    UR_CHECK_LAUNCH("ur_attn_bwd(dkv fewq)");
    return 0;
  }
  dim3 grid(ur_cdiv(p.Sk, 32 * NW) * p.nkv * p.B);
  if constexpr (HD == 128 && CAUSAL && NW == 4) {
    if (c128_bwd_ok(p)) {
      static std::atomic<uint64_t> once_c{0};   // per device
      UR_ONCE_PER_DEVICE(once_c) { int rc = set_smem(&attn_bwd_dkv_c128_kernel, c128::DKV_LDS_BYTES, "ur_attn_bwd(dkv c128)"); if (rc) return rc; }
      // persistent walk where the sweep divides evenly (see the kernel): one workgroup per CU for the whole launch
      const int ncu = device_cu_count(), ngroups = p.nkv * p.B;
      const bool pers = dkv_persist_enabled() && (ncu % 8) == 0 && (ngroups % 8) == 0 && (long)grid.x > ncu;
      if (pers) hipLaunchKernelGGL(attn_bwd_dkv_c128_kernel, dim3(ncu), dim3(256), c128::DKV_LDS_BYTES, st, p, 1);
      else hipLaunchKernelGGL(attn_bwd_dkv_c128_kernel, grid, dim3(256), c128::DKV_LDS_BYTES, st, p, 0);
      UR_CHECK_LAUNCH("ur_attn_bwd(dkv c128)");
      return 0;
    }
  }
  if (HD == 128 && NW == 4 && p.drop_thr == 0) {
    constexpr int SM2 = 2 * (2 * Cfg<128>::TILE + 4 * KT * (int)sizeof(float));      // = NB buffers of attn_bwd_dkv2_kernel
    static std::atomic<uint64_t> once2{0};   // per device
    UR_ONCE_PER_DEVICE(once2) { int rc = set_smem(&attn_bwd_dkv2_kernel<CAUSAL>, SM2, "ur_attn_bwd(dkv2)"); if (rc) return rc; }
    hipLaunchKernelGGL((attn_bwd_dkv2_kernel<CAUSAL>), grid, dim3(256), SM2, st, p);
    UR_CHECK_LAUNCH("ur_attn_bwd(dkv2)");
    return 0;
  }
  static std::atomic<uint64_t> once{0};   // per device
  UR_ONCE_PER_DEVICE(once) { int rc = set_smem(&attn_bwd_dkv_kernel<HD, CAUSAL, NW>, dkv_smem<HD>(), "ur_attn_bwd(dkv)"); if (rc) return rc; }
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, CAUSAL, NW>), grid, dim3(NW * 64), dkv_smem<HD>(), st, p);
  UR_CHECK_LAUNCH("ur_attn_bwd(dkv)");
  return 0;
}

inline int pick_nw(int S) { return S <= 32 ? 1 : (S <= 64 ? 2 : 4); }
// the shapes whose dK/dV run on attn_bwd_dkv_fewq_kernel (the only kernel that can also emit the column sums of dK | dV)
inline bool fewq_shape(const AttnP& p, int hd, bool causal) {
  return hd == 64 && !causal && pick_nw(p.Sk) == 4 && p.rep == 1 && p.Sq <= KT && p.Sk >= 256 && p.Sk <= MAX_KTILES * KT && fewq_enabled();
}
inline bool c128_bwd_ok(const AttnP& p) {
  return p.Sq == p.Sk && (p.Sk % 128) == 0 && p.Sk >= 128 && p.Sk <= c128::MAX_SK && fwd_c128_enabled() && (long)p.nq * p.B * 8 < (1L << 24) &&
         p.ldk * 2L * p.Sk < (1L << 31) && p.ldv * 2L * p.Sk < (1L << 31) && p.ldq * 2L * p.Sq < (1L << 31) && p.lddo * 2L * p.Sq < (1L << 31) &&
         p.drop_thr == 0 && (long)p.B * p.nq * p.Sq * 4 < (1L << 31);
}

#define UR_ATTN_DISPATCH(HD_, CAUSAL_, NW_, FN, ...)                                      \
  do {                                                                                    \
    if ((HD_) == 64) {                                                                    \
      if (CAUSAL_) { if ((NW_) == 1) return FN<64, true, 1>(__VA_ARGS__); if ((NW_) == 2) return FN<64, true, 2>(__VA_ARGS__); return FN<64, true, 4>(__VA_ARGS__); } \
      else { if ((NW_) == 1) return FN<64, false, 1>(__VA_ARGS__); if ((NW_) == 2) return FN<64, false, 2>(__VA_ARGS__); return FN<64, false, 4>(__VA_ARGS__); }      \
    } else {                                                                              \
      if (CAUSAL_) { if ((NW_) == 1) return FN<128, true, 1>(__VA_ARGS__); if ((NW_) == 2) return FN<128, true, 2>(__VA_ARGS__); return FN<128, true, 4>(__VA_ARGS__); } \
      else { if ((NW_) == 1) return FN<128, false, 1>(__VA_ARGS__); if ((NW_) == 2) return FN<128, false, 2>(__VA_ARGS__); return FN<128, false, 4>(__VA_ARGS__); }      \
    }                                                                                     \
  } while (0)

int do_fwd(const AttnP& p, int hd, bool causal, hipStream_t st) { UR_ATTN_DISPATCH(hd, causal, pick_nw(p.Sq), launch_fwd, p, st); }
int do_dq(const AttnP& p, int hd, bool causal, hipStream_t st) { UR_ATTN_DISPATCH(hd, causal, pick_nw(p.Sq), launch_dq, p, st); }
int do_dkv(const AttnP& p, int hd, bool causal, hipStream_t st) { UR_ATTN_DISPATCH(hd, causal, pick_nw(p.Sk), launch_dkv, p, st); }

}  // namespace

#if UR_C128_STAMPS
extern "C" int ur_lab_c128_stamps(unsigned int* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c128_stamps), sizeof(unsigned int) * n);   // 32 words per wave
}
#endif
#if UR_DKV2_STAMPS
extern "C" int ur_lab_attn_stamps(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), sizeof(long long) * n);
}
#endif

extern "C" int ur_attn_fwd(const ur_attn_args* a, void* stream) {
  AttnP p;
  int rc = fill(p, a);
  if (rc) return rc;
  if (a->B == 0) return 0;
  UR_REQUIRE(a->o && UR_ALIGNED16(a->o) && (a->ldo % 4) == 0 && a->ldo >= (int64_t)a->nq * a->head_dim, "ur_attn_fwd: bad output");
  if (tiny_shape(p, a->head_dim, a->causal != 0, false)) return launch_tiny(p, false, (hipStream_t)stream);
  return do_fwd(p, a->head_dim, a->causal != 0, (hipStream_t)stream);
}

extern "C" int64_t ur_attn_bwd_kv_colsum_floats(const ur_attn_args* a) {
  AttnP p;
  if (!a || fill(p, a) != 0 || a->B <= 0) return 0;
  if (tiny_shape(p, a->head_dim, a->causal != 0, true) || !fewq_shape(p, a->head_dim, a->causal != 0)) return 0;
  return 2 * (int64_t)a->B * a->nq * a->head_dim;
}

extern "C" int ur_attn_mode(int key, int value) {
  if (key < 0 || key >= UR_ATTN_MODE_COUNT) return -1;
  static const int dflt[UR_ATTN_MODE_COUNT] = {3, 1, 1, 1};
  const int hi = key == UR_ATTN_MODE_TINY ? 3 : 1;
  if (value == -2) return g_attn_mode[key].load(std::memory_order_relaxed);                     // query only
  return g_attn_mode[key].exchange(value < 0 || value > hi ? dflt[key] : value);
}

extern "C" int64_t ur_attn_bwd_workspace_floats(int32_t B, int32_t nq, int32_t Sq) {
  return 4 * (int64_t)B * nq * Sq + 16;      // two row-constant planes + two planes of dropout row keys + the dK/dV kernel's eight queue words (padded to 64 bytes)
}

extern "C" int ur_attn_bwd(const ur_attn_args* a, const ur_attn_bwd_args* g, void* stream) {
  AttnP p;
  int rc = fill(p, a);
  if (rc) return rc;
  if (a->B == 0) return 0;
  UR_REQUIRE(g && g->dout && (g->dq || g->rope_q_raw) && g->dk && g->dv && g->delta && a->o, "ur_attn_bwd: null argument");
  if (g->rope_q_raw) {
    UR_REQUIRE(a->head_dim == 128 && a->causal && a->Sq == a->Sk, "ur_attn_bwd: the fused q-norm / RoPE backward is built for the causal head_dim-128 shape");
    UR_REQUIRE(g->rope_q_weight && g->rope_cos && g->rope_sin && g->rope_dq_raw && UR_ALIGNED16(g->rope_q_raw) && UR_ALIGNED16(g->rope_dq_raw) &&
               UR_ALIGNED16(g->rope_q_weight) && UR_ALIGNED16(g->rope_cos) && UR_ALIGNED16(g->rope_sin) && (g->rope_ldraw % 8) == 0 &&
               (g->rope_lddraw % 8) == 0 && g->rope_ldraw >= (int64_t)a->nq * a->head_dim && g->rope_lddraw >= (int64_t)a->nq * a->head_dim,
               "ur_attn_bwd: bad q-norm / RoPE operands (16-byte aligned, row strides %% 8 == 0)");
    UR_REQUIRE(g->rope_rstd == nullptr || (g->rope_rstd_h0 >= 0 && g->rope_rstd_ld >= (int64_t)g->rope_rstd_h0 + a->nq),
               "ur_attn_bwd: rope_rstd needs rope_rstd_ld >= rope_rstd_h0 + nq");
  }
  UR_REQUIRE(UR_ALIGNED16(g->dout) && UR_ALIGNED16(a->o) && (g->lddo % 8) == 0 && (a->ldo % 8) == 0, "ur_attn_bwd: dout/o need 16-byte aligned rows");
  UR_REQUIRE((g->dq == nullptr || (g->lddq % 4) == 0) && (g->lddk % 4) == 0 && (g->lddv % 4) == 0 && ((uintptr_t)g->dq & 7) == 0 && ((uintptr_t)g->dk & 7) == 0 &&
             ((uintptr_t)g->dv & 7) == 0, "ur_attn_bwd: gradient outputs need 8-byte aligned rows");
  p.dout = (const bf16_t*)g->dout; p.dq = (bf16_t*)g->dq; p.dk = (bf16_t*)g->dk; p.dv = (bf16_t*)g->dv; p.delta = g->delta;
  p.lddo = g->lddo; p.lddq = g->lddq; p.lddk = g->lddk; p.lddv = g->lddv;
  p.rp_raw = (const bf16_t*)g->rope_q_raw; p.rp_ldraw = g->rope_ldraw; p.rp_w = g->rope_q_weight; p.rp_cos = g->rope_cos; p.rp_sin = g->rope_sin;
  p.rp_eps = g->rope_eps; p.rp_draw = (bf16_t*)g->rope_dq_raw; p.rp_lddraw = g->rope_lddraw;
  p.rp_rstd = g->rope_q_raw ? g->rope_rstd : nullptr; p.rp_rstd_ld = g->rope_rstd_ld; p.rp_rstd_h0 = g->rope_rstd_h0;
  const bool rope_k = p.rp_rstd != nullptr && g->rope_k != nullptr;
  p.rk_src = nullptr;
  if (rope_k) {
    UR_REQUIRE(g->rope_k_weight && g->rope_dk_raw && UR_ALIGNED16(g->rope_k) && UR_ALIGNED16(g->rope_dk_raw) && UR_ALIGNED16(g->rope_k_weight) &&
               (g->rope_ldk % 8) == 0 && (g->rope_lddkraw % 8) == 0 && g->rope_ldk >= (int64_t)a->nkv * a->head_dim &&
               g->rope_lddkraw >= (int64_t)a->nkv * a->head_dim && g->rope_rstd_hk0 >= 0 && g->rope_rstd_ld >= (int64_t)g->rope_rstd_hk0 + a->nkv,
               "ur_attn_bwd: bad k-norm / RoPE operands (rope_k, rope_k_weight, rope_dk_raw; 16-byte aligned, row strides %% 8 == 0)");
    p.rk_src = (const bf16_t*)g->rope_k; p.rk_ld = g->rope_ldk; p.rk_w = g->rope_k_weight; p.rk_rstd_h0 = g->rope_rstd_hk0;
    p.rk_dst = (bf16_t*)g->rope_dk_raw; p.rk_lddst = g->rope_lddkraw;
  }
  hipStream_t st = (hipStream_t)stream;
  p.colsum_part = nullptr;
  if (g->kv_colsum != nullptr) {
    UR_REQUIRE(g->kv_colsum_ws != nullptr && ur_attn_bwd_kv_colsum_floats(a) > 0,
               "ur_attn_bwd: kv_colsum is produced by the few-query dK/dV kernel only (ur_attn_bwd_kv_colsum_floats(a) == 0 for this shape) and needs kv_colsum_ws");
    p.colsum_part = g->kv_colsum_ws;
  }
  // the workspace behind the two row-constant planes: two planes of dropout row keys, then the queue words (ur_attn_bwd_workspace_floats)
  p.rowkeys = reinterpret_cast<uint32_t*>(g->delta + 2 * (int64_t)a->B * a->nq * a->Sq);
  p.queue = reinterpret_cast<unsigned int*>(g->delta + 4 * (int64_t)a->B * a->nq * a->Sq);
  p.lse_log2 = (a->head_dim == 128 && a->causal != 0 && c128_bwd_ok(p)) ? 1 : 0;
  if (tiny_shape(p, a->head_dim, a->causal != 0, true)) return launch_tiny(p, true, st);      // dQ, dK, dV in one kernel
  // the dQ kernel also computes the row constants (delta, -LSE/scale) and leaves them in `delta` for dK/dV
  // (only the generated dK/dV kernel carries the k heads' backward in its store; on every other path dk is written roped and the
  // stand-alone kernel turns it into the raw gradient here)
  const bool rope_k_fused = rope_k && p.lse_log2 != 0;
  if (rope_k && !rope_k_fused) {
    // (checked BEFORE anything is launched: a refused call leaves every output untouched)
    UR_REQUIRE(g->lddk == (int64_t)a->nkv * a->head_dim, "ur_attn_bwd: rope_k on this shape needs a dense dk [B*Sk, nkv*hd]");
    p.rk_src = nullptr;
  }
  rc = do_dq(p, a->head_dim, a->causal != 0, st);
  if (rc) return rc;
  rc = do_dkv(p, a->head_dim, a->causal != 0, st);
  if (rc) return rc;
  if (p.colsum_part != nullptr) {
    const int n = 2 * a->nq * a->head_dim;
    hipLaunchKernelGGL(kv_colsum_reduce_kernel, dim3(ur_cdiv(n, 64)), dim3(256), 0, st, (const float*)p.colsum_part, g->kv_colsum, (int)a->B, n);
    UR_CHECK_LAUNCH("ur_attn_bwd(kv colsum)");
  }
  if (rope_k && !rope_k_fused) {
    return ur_qknorm_rope_bwd_roped_k(g->dk, g->rope_k, g->rope_ldk, g->rope_rstd, g->rope_rstd_ld, g->rope_rstd_hk0, g->rope_k_weight, g->rope_cos,
                                      g->rope_sin, g->rope_dk_raw, g->rope_lddkraw, (int64_t)a->B * a->Sk, a->Sk, a->nkv, a->head_dim, stream);
  }
  return 0;
}

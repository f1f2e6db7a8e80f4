// LoRA adapter side kernels (rank 16), gfx950 only.
//
// peft LoraLayer (call site training/train_item_individual_token_joint.py:121-131, r=16, lora_dropout=0.1):
//     y = W x + s * B_a ( A_a dropout_a(x) )            one nn.Dropout per adapter a
// The rank-16 products are HBM-bound streams over an [M, W] activation (M = tokens): running them through
// the 128x128-tile GEMM wastes 8x of its MFMA work and LDS traffic, and masking the operand while it is
// staged de-pipelines it.  Three kernels replace that:
//   ur_lora_dropout_bits : the dropped flags of every adapter input element, ONE bit each, generated once per
//                          (layer, adapter group, step) by a counter-based generator and kept for the backward
//   ur_lora_project      : P[m, 16a+j] = alpha * sum_w keep_a(m,w) X[m, c0_a+w] U_a[j,w]   (reduction over columns)
//                          forward  t  = s * dropout(x) A^T   (adapters share x, one bit plane each)
//                          backward tb = s * dy_a B_a         (adapter a owns a column range of dy)
//   ur_lora_reduce       : G_a[j,w]  = alpha * sum_m V[m,16a+j] keep_a(m,w) X[m, c0_a+w]   (reduction over tokens)
//                          dA = tb^T dropout(x) ;  dB_a^T = t_a^T dy_a   (deterministic token split + slab sum)
// Both read X exactly once for all adapters that share it; the MFMA operands come straight from global memory
// (project) or through one LDS round trip with hardware-transposed reads (reduce).
//
// Bit layout (shared with ur_gemm's masked LoRA epilogue): plane a, row m, byte c/8 covers columns c..c+7
// (c % 8 == 0): bit i (i < 4) = element c+2i dropped, bit 4+i = element c+2i+1 dropped -- the order in which
// a 16-byte bf16x8 fragment keeps its elements in 32-bit pairs, so a byte expands to four pair masks with
// one shift/and + one packed arithmetic shift each.
#include <type_traits>
#include "common.hip.h"
#include "unirec_hip.h"

// Cache policy of the ring kernels' activation pieces (the aux operand of global_load_lds): 2 = nt.  They are read once per kernel, like the
// register-staged kernels' ld_stream loads; with the default policy the rings ran 4.4 TB/s where nt reaches 5.05 (lora_bgrad on a
// [131072, 4096] input: 244 -> 212 us; the ring DEPTH does not matter: 2 stages 252 us).  0 = default policy (A/B builds).
#ifndef UR_RING_AUX
#define UR_RING_AUX 2
#endif
namespace {

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
typedef short ss2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, (us2_t)(__builtin_bit_cast(us2_t, a) - __builtin_bit_cast(us2_t, b)));
}
__device__ __forceinline__ uint32_t pk_sign16(uint32_t a) {          // each 16-bit half -> 0xffff if its sign bit is set
  return __builtin_bit_cast(uint32_t, (ss2_t)(__builtin_bit_cast(ss2_t, a) >> 15));
}
// zero the dropped elements of a bf16x8 fragment (dropped-flag byte in the layout above)
__device__ __forceinline__ uint4 drop_apply(uint4 x, uint32_t byte) {
  const uint32_t z = (byte & 0xfu) | ((byte >> 4) << 16);
  x.x &= ~pk_sign16((z << 15) & 0x80008000u);
  x.y &= ~pk_sign16((z << 14) & 0x80008000u);
  x.z &= ~pk_sign16((z << 13) & 0x80008000u);
  x.w &= ~pk_sign16((z << 12) & 0x80008000u);
  return x;
}
// the activations these kernels stream are read once per kernel: non-temporal loads keep them out of the L2's way
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream(const bf16_t* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ int kc_g(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int f64sw(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); }
// 16-byte-chunk swizzle of a [tokens][64 columns] tile (128-byte rows) that is conflict-free for BOTH accesses of
// lora_bgrad: ds_write_b128 of one column chunk by 8 consecutive token rows (8-lane groups, banks mod 128 B: the 8 rows
// need 8 distinct chunk positions -> low 3 bits of the row), and ds_read_b64_tr_b16 of a 32-byte chunk pair by rows
// {0..3, 8..11} + 4 (32-lane groups, banks mod 256 B: rows of one parity need distinct pair positions -> bit 3 of the row
// moves the pair index by 2).  The f64sw form left the writes 4-way conflicted (PMC: 66 % of the kernel's LDS cycles).
__device__ __forceinline__ int sw16(int r) { return (r & 7) ^ ((r & 8) >> 1); }
// The RING kernels' image of the same tile has no LDS write instruction (LDS-DMA pieces land contiguously; the swizzle is applied on the
// source side), and two kinds of read: ds_read_b128 of a row's 16-byte chunk (4 s2 + g) by lanes (row l15, group g) -- the hardware serves
// that instruction in four 16-lane groups, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS table): rows
// {0-3, 12-15} at chunk c beside rows {4-11} at chunk c ^ 1, banks mod 256 B = two rows -- and the transposed reads above.  Under sw16 the
// row reads are 2-way conflicted (rows 0 / 12, 2 / 14, 4 / 8, 6 / 10 share a chunk position: PMC, 36-50 % of these kernels' LDS cycles).
// swr = the f64sw PAIR swizzle written as a chunk swizzle serves both: transposed reads need rows {0, 2, 8, 10} (and {4, 6, 12, 14}) on
// four different chunk pairs, row reads need the chunk positions of rows {0, 2, 12, 14} and (1 ^ those of) rows {4, 6, 8, 10} all different:
// positions (0, 2, 0, 2, 4, 6, 4, 6) for rows (0, 2, 4, .. 14); odd rows alike in the other half of the banks.
#ifndef UR_RING_SWR
#define UR_RING_SWR 1        // lab: 0 = sw16 in the ring kernels (round 5)
#endif
__device__ __forceinline__ int swr(int r) { return UR_RING_SWR ? ((((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2)) : sw16(r); }
// (ring kernels, [tokens][16] tiles of 32-byte rows: transposed reads of rows {0-3, 8-11} + 4 by a 32-lane group need rows r and r + 8 in
// different halves of the 256 bytes of banks: LDS row slot s holds source row vrow(s), an involution)
__device__ __forceinline__ int vrow(int s) { return UR_RING_SWR ? (s ^ (((s >> 3) & 1) << 2)) : s; }

// ---- dropped-flag bit planes ---------------------------------------------------------------------
// thread <-> (row m, group q of 32 columns): one 32-bit word per adapter plane.  Every element draws a 15-bit value u;
// dropped iff u < thr = p * 2^15.  The (row, group, adapter, seed) counter is murmur-finalised ONCE into the word's key h;
// UR_BITS_XORSHIFT = 2 (default): the 32 values are BIT-SLICED over up to 15 consecutive states of a multiply-xorshift chain started at h
//   (state k carries bit k of all 32 values, least significant first) and the 32 comparisons are one boolean step per state:
//   lt_k = t_k ? (~b_k | lt_{k-1}) : (~b_k & lt_{k-1});  bits below thr's lowest set bit cannot decide and are not drawn.
//   7 vector instructions per state for 32 decisions (the kernel is VALU-bound: profiles/README.md r2_lora_bits.txt).
// = 1: 16 states, two 15-bit fields of each compared by a packed subtract; = 0: a second finaliser per pair (round 1).
// Streams of different words start at hashed, unrelated points of the generator's single 2^32 - 1 cycle.
#ifndef UR_BITS_XORSHIFT
#define UR_BITS_XORSHIFT 2
#endif
__global__ void lora_bits_kernel(uint64_t seed, uint32_t thr15, int M, int W, int nad, long bits_ld, long bits_stride,
                                 uint8_t* __restrict__ bits, long row0) {
  // blockIdx.y counts chunks of 2^20 rows so that the (row, group) split is ONE 32-bit division (a 64-bit one costs ~150 vector
  // instructions per thread, a sixth of the kernel)
  const uint32_t ng = (uint32_t)(bits_ld >> 2);
  const uint32_t lid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t ml = lid / ng;
  const int q = (int)(lid - ml * ng);
  const long m = ((long)blockIdx.y << 20) + ml;
  if (ml >= (1u << 20) || m >= M) return;
  [[maybe_unused]] const uint32_t thr_pk = thr15 * 0x10001u;
  // seed-only key (scalar unit), entered between the two mixing rounds as ur_hash2 does: the streams of two seeds are
  // neither shifted nor XOR-permuted copies of each other even when the seeds differ in a few low bits only
  const uint32_t s_lo = (uint32_t)seed, s_hi = (uint32_t)(seed >> 32);
  const uint32_t k2 = fmix32((s_lo * 0x7FEB352Du) ^ ((s_hi << 13) | (s_hi >> 19)) ^ 0x5851F42Du);
  for (int a = 0; a < nad; ++a) {
    uint32_t out = 0;
    if (q * 32 < W) {
      const uint64_t ctr = (((uint64_t)(row0 + m) * (uint64_t)ng + (uint64_t)(uint32_t)q) << 2) + (uint64_t)a;     // row0: rows that precede row 0 in the global minibatch
      const uint32_t h = fmix32((((uint32_t)ctr ^ s_lo) + fmix32((uint32_t)(ctr >> 32) + s_hi + 0x9E3779B9u)) ^ k2) + k2;
      uint32_t w = h ? h : 0x9E3779B9u;            // xorshift32 state: never zero
#if UR_BITS_XORSHIFT == 2
      if (thr15 != 0) {
        const int k0 = __builtin_ctz(thr15);       // uniform (kernel argument): a scalar loop of 15 - k0 states
        uint32_t lt = 0;
#pragma unroll 1
        for (uint32_t tb = (thr15 >> k0) | (0x8000u >> k0); tb != 1; tb >>= 1) {     // sentinel above bit 14: the zero bits above thr's top bit count
          // one multiply-xorshift round per state (half of the lowbias32 finaliser; a bijection whose only fixed point is 0).
          // Round 2 used xorshift32 here: GF(2)-linear, so the 15 x 32 value bits of a word were linear functions of its
          // 32-bit key -- pairs of columns were independent, but the number of dropped columns per word was not binomial
          // (P(no column dropped) 6.5 % low at p = 0.1: tests/test_gpu_primitives.py popcount test).  The multiply's carries
          // break the linearity for one quarter-rate instruction per state.
          w ^= w >> 16; w *= 0x7FEB352Du; w ^= w >> 15;
          const uint32_t T = 0u - (tb & 1u);                             // all ones where thr has this bit
          lt = (~w & lt) | (T & (~w | lt));
        }
        out = lt;
      }
#else
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if UR_BITS_XORSHIFT == 1
          w ^= w << 13; w ^= w >> 17; w ^= w << 5;
#else
          w = fmix32(h + (uint32_t)(4 * b + i + 1) * 0x9E3779B9u);       // lab: the second finaliser per pair of decisions
#endif
          const uint32_t d = pk_sub16(w & 0x7fff7fffu, thr_pk);        // sign of each half set iff field < thr
          v |= ((d >> 15) & 0x10001u) << i;
        }
        out |= ((v | (v >> 12)) & 0xffu) << (8 * b);
      }
#endif
    }
    *reinterpret_cast<uint32_t*>(bits + (long)a * bits_stride + m * bits_ld + 4 * q) = out;
  }
}

__device__ __forceinline__ const char* bg_uniform_ptr(const char* p) {          // (gemm_common.hip.h: uniform_ptr)
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
// ---- column-reduction products -------------------------------------------------------------------
struct ProjP {
  const bf16_t* X; long ldx; int M;
  int col0[4], width[4];
  const bf16_t* U[4]; long ldu[4];
  const uint8_t* bits; long bits_ld, bits_stride;
  bf16_t* P; long ldp;
  float alpha;
};

// 256 threads = 4 waves, each wave 2 x 16 tokens, all of the entry's columns in chunks of 128.  NAD adapters share
// the X fragments (NAD > 1: gridDim.y == 1); separate column ranges run as NAD = 1 with blockIdx.y = adapter.
// X fragments are the MFMA column operand straight from global memory (lane: token lane&15, 8 consecutive
// columns); the U chunk is staged once per block through a 2-slot LDS ring (GEMM K-contiguous image).
template <int NAD, bool MASKED>
__global__ __launch_bounds__(256, 4) void lora_project_kernel(ProjP p) {
  constexpr int RB = 2, KC = 128;
  constexpr int SUB = NAD * 16 * 128;              // one 64-column sub-tile of the U chunk: rows x 128 B
  constexpr int STAGE = 2 * SUB;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int y = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int W = p.width[y], col0 = p.col0[y];
  const int tok0 = blockIdx.x * (4 * RB * 16) + wave * (RB * 16);

  f32x4 acc[RB][NAD];
  const bf16_t* xrow[RB];
  const uint8_t* brow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = min(tok0 + 16 * rb + l15, p.M - 1);
    xrow[rb] = p.X + (long)m * p.ldx + col0;
    brow[rb] = MASKED ? p.bits + (long)m * p.bits_ld : nullptr;
#pragma unroll
    for (int a = 0; a < NAD; ++a) acc[rb][a] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nchunks = (W + KC - 1) / KC;
  for (int c = 0; c < nchunks; ++c) {
    const int kc = c * KC;
    uint4 ureg[NAD];
#pragma unroll
    for (int i = 0; i < NAD; ++i) {
      const int pi = tid + 256 * i, row = pi >> 4, k = kc + (pi & 15) * 8;
      const int a = y + (row >> 4);
      ureg[i] = (k < W) ? *reinterpret_cast<const uint4*>(p.U[a] + (long)(row & 15) * p.ldu[a] + k) : make_uint4(0, 0, 0, 0);
    }
    uint4 xf[RB][4], bw[RB][NAD];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = kc + 32 * s + 8 * g;
        xf[rb][s] = (k < W) ? ld_stream(xrow[rb] + k) : make_uint4(0, 0, 0, 0);
      }
      if (MASKED) {
#pragma unroll
        for (int a = 0; a < NAD; ++a) bw[rb][a] = *reinterpret_cast<const uint4*>(brow[rb] + (long)a * p.bits_stride + (kc >> 3));
      }
    }
    char* st = smem + (c & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < NAD; ++i) {
      const int pi = tid + 256 * i, row = pi >> 4, c16 = pi & 15;
      *reinterpret_cast<uint4*>(st + (c16 >> 3) * SUB + row * 128 + (((c16 & 7) ^ kc_g(row)) << 4)) = ureg[i];
    }
    __syncthreads();       // slot c&1 is re-written two chunks later: every wave has passed the next barrier by then
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int a = 0; a < NAD; ++a) {
        const int row = a * 16 + l15, ch = 4 * (s & 1) + g;
        const bf16x8 uf = *reinterpret_cast<const bf16x8*>(st + (s >> 1) * SUB + row * 128 + ((ch ^ kc_g(row)) << 4));
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          uint4 x = xf[rb][s];
          if (MASKED) {
            const uint32_t wsel = s == 0 ? bw[rb][a].x : s == 1 ? bw[rb][a].y : s == 2 ? bw[rb][a].z : bw[rb][a].w;
            x = drop_apply(x, (wsel >> (8 * g)) & 0xffu);
          }
          acc[rb][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf, __builtin_bit_cast(bf16x8, x), acc[rb][a], 0, 0, 0);
        }
      }
    }
  }
  // lane holds P[token l15][16 (y + a) + 4 g .. + 3]
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = tok0 + 16 * rb + l15;
    if (m < p.M) {
#pragma unroll
      for (int a = 0; a < NAD; ++a) {
        const f32x4 v = acc[rb][a];
        *reinterpret_cast<uint2*>(p.P + (long)m * p.ldp + 16 * (y + a) + 4 * g) =
            make_uint2(pack_bf2(v[0] * p.alpha, v[1] * p.alpha), pack_bf2(v[2] * p.alpha, v[3] * p.alpha));
      }
    }
  }
}

// ---- the column reduction of ONE adapter with X streamed through an LDS-DMA ring ------------------------------------------------
// grid: x = block of 256 tokens, y = entry (widths multiples of 64).  The block walks the entry's columns in chunks of 64: a stage =
// the [256 x 64] piece of X (32 KiB, rows of 128 bytes, chunks swizzled by swr on the source side), the [16 x 64] chunk of U and
// -- MASKED -- the 8 flag bytes of each row, fetched by global_load_lds three stages ahead (96 KiB in flight per CU); one barrier
// per stage; wave w consumes tokens 64 w .. + 63 as MFMA column operands (the layout of lora_bgrad_ring_kernel's tb product).
#ifndef UR_P2_NST
#define UR_P2_NST 4          // lab: ring depth of lora_project_ring_kernel (2 = one stage in flight, two workgroups per CU)
#endif
constexpr int P2_NST = UR_P2_NST, P2_XS = 256 * 128, P2_STAGE = P2_XS + 2048 + 2048, P2_SMEM = P2_NST * P2_STAGE;
template <bool MASKED>
__global__ __launch_bounds__(256) void lora_project_ring_kernel(ProjP p) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
  constexpr int NP = 8 + 1 + (MASKED ? 2 : 0);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int y = blockIdx.y;
  const int W = p.width[y], col0 = p.col0[y];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tok0 = blockIdx.x * 256;
  const int nch = W / 64;
  const bool full = tok0 + 256 <= p.M;                        // uniform

  // ---- producer ----
  const int prow = lane >> 3, wpar = wave & 1;
  // X piece = rows 8 piece .. + 7 (piece = 4 i + wave): row = 8 (wave & 1) + prow (mod 16) for the swizzle
  const int schunk = (lane & 7) ^ swr(8 * wpar + prow);
  const uint32_t xlane = (uint32_t)((prow * p.ldx + schunk * 8) * 2);
  const uint32_t ulane = (uint32_t)((prow * p.ldu[y] + schunk * 8) * 2);
  const uint32_t blane = (uint32_t)((lane >> 1) * p.bits_ld + 4 * (lane & 1));
  // (Every block walks the chunks in the SAME order: a token's sum must not depend on where its row sits in the batch -- the shard
  // invariance the forward is tested for.  Measured cost: with a power-of-two row stride (2048 columns) the blocks, in lockstep, keep
  // every request in flight on the same 128 bytes of a row, i.e. on a few memory channels -- 141 us against 104 us with the chunk
  // order rotated per block, which a kernel whose sums must not depend on the row's position may not do.)
  auto issue = [&](int c) {
    char* st = smem + (c & (P2_NST - 1)) * P2_STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int piece = 4 * i + wave;
      if (full) {
        const char* ub = bg_uniform_ptr(reinterpret_cast<const char*>(p.X + (long)(tok0 + 8 * piece) * p.ldx + col0 + 64 * c));
        __builtin_amdgcn_global_load_lds((gbl_void*)(ub + xlane), (lds_void*)(st + piece * 1024), 16, 0, UR_RING_AUX);
      } else {
        const int m = min(tok0 + 8 * piece + prow, p.M - 1);
        __builtin_amdgcn_global_load_lds((gbl_void*)(p.X + (long)m * p.ldx + col0 + 64 * c + schunk * 8), (lds_void*)(st + piece * 1024), 16, 0, UR_RING_AUX);
      }
    }
    {
      const char* ub = bg_uniform_ptr(reinterpret_cast<const char*>(p.U[y] + (long)(8 * wpar) * p.ldu[y] + 64 * c));
      __builtin_amdgcn_global_load_lds((gbl_void*)(ub + ulane), (lds_void*)(st + P2_XS + wpar * 1024), 16, 0, 0);
    }
    if (MASKED) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {            // this wave's own rows 64 wave + 32 i + (lane >> 1), 4 of the chunk's 8 flag bytes per lane
        const int r0 = 64 * wave + 32 * i;
        if (full) {
          const char* ub = bg_uniform_ptr(reinterpret_cast<const char*>(p.bits + (long)(tok0 + r0) * p.bits_ld + 8 * c));
          __builtin_amdgcn_global_load_lds((gbl_void*)(ub + blane), (lds_void*)(st + P2_XS + 2048 + r0 * 8), 4, 0, 0);
        } else {
          const int m = min(tok0 + r0 + (lane >> 1), p.M - 1);
          __builtin_amdgcn_global_load_lds((gbl_void*)(p.bits + (long)m * p.bits_ld + 8 * c + 4 * (lane & 1)), (lds_void*)(st + P2_XS + 2048 + r0 * 8), 4, 0, 0);
        }
      }
    }
  };

  f32x4 acc[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int npro = min(P2_NST - 1, nch);
  for (int c = 0; c < npro; ++c) issue(c);
  const int swl = swr(l15);
  const uint32_t xrd = (uint32_t)((64 * wave + l15) * 128), urd = (uint32_t)(P2_XS + l15 * 128), brd = (uint32_t)(P2_XS + 2048 + (64 * wave + l15) * 8 + g);
  for (int c = 0; c < nch; ++c) {
    const int later = min(nch, c + P2_NST - 1) - (c + 1);
    if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NP) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (c + P2_NST - 1 < nch) issue(c + P2_NST - 1);          // (every wave has finished stage c - 1: its slot takes stage c + 3)
    __builtin_amdgcn_sched_barrier(0);
    // (LDS reads as asm: hipcc may answer a plain LDS load that could alias an LDS-DMA in flight with s_waitcnt vmcnt(0))
    const uint32_t sb = lds_off(smem) + (uint32_t)((c & (P2_NST - 1)) * P2_STAGE);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      u32x4_t xv[4], uv;
      uint32_t fb[4] = {0u, 0u, 0u, 0u};
      const uint32_t sw = (uint32_t)(((4 * s2 + g) ^ swl) << 4);
      const uint32_t ax = sb + xrd + sw, au = sb + urd + sw, ab = sb + brd + 4u * s2;
      if (MASKED)
        asm volatile("ds_read_b128 %0, %9\n\tds_read_b128 %1, %9 offset:2048\n\tds_read_b128 %2, %9 offset:4096\n\tds_read_b128 %3, %9 offset:6144\n\t"
                     "ds_read_b128 %4, %10\n\t"
                     "ds_read_u8 %5, %11\n\tds_read_u8 %6, %11 offset:128\n\tds_read_u8 %7, %11 offset:256\n\tds_read_u8 %8, %11 offset:384\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3]), "=&v"(uv), "=&v"(fb[0]), "=&v"(fb[1]), "=&v"(fb[2]), "=&v"(fb[3])
                     : "v"(ax), "v"(au), "v"(ab) : "memory");
      else
        asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:2048\n\tds_read_b128 %2, %5 offset:4096\n\tds_read_b128 %3, %5 offset:6144\n\t"
                     "ds_read_b128 %4, %6\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3]), "=&v"(uv) : "v"(ax), "v"(au) : "memory");
      const bf16x8 uf = __builtin_bit_cast(bf16x8, uv);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        uint4 x = make_uint4(xv[rb][0], xv[rb][1], xv[rb][2], xv[rb][3]);
        if (MASKED) x = drop_apply(x, fb[rb]);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf, __builtin_bit_cast(bf16x8, x), acc[rb], 0, 0, 0);
      }
    }
  }
  // lane holds P[token l15][16 y + 4 g .. + 3]
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    const int m = tok0 + 64 * wave + 16 * rb + l15;
    if (m < p.M) {
      const f32x4 v = acc[rb];
      *reinterpret_cast<uint2*>(p.P + (long)m * p.ldp + 16 * y + 4 * g) =
          make_uint2(pack_bf2(v[0] * p.alpha, v[1] * p.alpha), pack_bf2(v[2] * p.alpha, v[3] * p.alpha));
    }
  }
}

// ---- RMSNorm forward + the adapters' down projection in ONE pass over the residual stream ---------------------------------
// h = w * (x * rstd) (Qwen3RMSNorm, modeling_qwen3.py:59-64) is written once and never re-read by a projection kernel:
// t[m, 16a + j] = alpha * sum_c keep_a(m, c) h[m, c] A_a[j, c] for the NAD adapters that consume h (q|k|v: 3, gate|up: 2)
// comes out of the same registers.  A wave owns 16 tokens x D = 1024 columns: lane (token l15, column group g) holds its 32
// 16-byte pieces of the row from one burst of loads (32 KiB in flight per wave, no barrier in front of them), the sum of
// squares closes over the four lane groups with two shuffles, and the normalised pieces are the MFMA column operand
// directly (the lora_project layout).  A_a chunks of 128 columns go through the same 2-slot LDS ring as lora_project.
struct RmsLoraP {
  const bf16_t* X; const float* W; bf16_t* H; float* rstd; int M; float eps;
  const bf16_t* U[4]; long ldu[4];
  const uint8_t* bits; long bits_ld, bits_stride;
  bf16_t* P; long ldp; float alpha;
};
template <int NAD, bool MASKED>
__global__ __launch_bounds__(256, 2) void rms_lora_kernel(RmsLoraP p) {
  constexpr int D = 1024, KC = 128, NC = D / 32;
  constexpr int SUB = NAD * 16 * 128;
  constexpr int STAGE = 2 * SUB;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  __shared__ __attribute__((aligned(16))) float wlds[D];          // the norm weight, staged once: the chunk loop reads it from LDS (two global loads per
                                                                  // 32-column step sat in front of every normalisation: 64 L2 round trips per wave)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  *reinterpret_cast<float4*>(wlds + 4 * tid) = *reinterpret_cast<const float4*>(p.W + 4 * tid);          // 256 threads x 4 = D (made visible by the first chunk's barrier)
  const int tok = blockIdx.x * 64 + wave * 16 + l15;
  const int m = min(tok, p.M - 1);
  const bool mok = tok < p.M;
  const bf16_t* xrow = p.X + (long)m * D + 8 * g;
  uint4 xf[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) xf[c] = ld_stream(xrow + 32 * c);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const uint32_t wd[4] = {xf[c].x, xf[c].y, xf[c].z, xf[c].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float lo = bf_lo(wd[e]), hi = bf_hi(wd[e]); q += lo * lo; q += hi * hi; }
  }
  // the row stays PACKED between the two passes: without this hipcc keeps all 256 unpacked floats of the sum of squares for
  // the normalisation below (256 more registers -> 600-900 bytes of scratch per lane)
#pragma unroll
  for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(xf[c].x), "+v"(xf[c].y), "+v"(xf[c].z), "+v"(xf[c].w));
  q += __shfl_xor(q, 16, 64);
  q += __shfl_xor(q, 32, 64);
  const float rs = rsqrtf(q / (float)D + p.eps);
  if (g == 0 && mok) p.rstd[m] = rs;
  f32x4 acc[NAD];
#pragma unroll
  for (int a = 0; a < NAD; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16_t* hrow = p.H + (long)m * D + 8 * g;
  // per-lane bases with compile-time chunk offsets on top (no lane-indexed access to the argument arrays: that sends the
  // whole struct to scratch)
  const bf16_t* ulane[NAD]; const uint8_t* blane[NAD];
#pragma unroll
  for (int i = 0; i < NAD; ++i) {
    ulane[i] = p.U[i] + (long)((tid >> 4) & 15) * p.ldu[i] + (tid & 15) * 8;
    blane[i] = MASKED ? p.bits + (long)m * p.bits_ld + (long)i * p.bits_stride : nullptr;
  }
  // The A pieces of chunk c + 1 are requested right behind the barrier of chunk c, into the registers chunk c's pieces have
  // just left (they went to LDS): one exposed L2 round trip per chunk less (-2.4 %; the flag words as well: no further gain).
  // (raw vector types: arrays of HIP's uint4 class that live across the chunks are kept in scratch memory by hipcc)
  u32x4_t ureg[NAD];
#pragma unroll
  for (int i = 0; i < NAD; ++i) ureg[i] = *reinterpret_cast<const u32x4_t*>(ulane[i]);
  // compile-time chunk index: xf[] must stay in registers (hipcc does not unroll a loop with a barrier in it on request).  A macro, not a
  // lambda: an array the lambda would capture by reference (the A pieces carried from chunk to chunk) is kept in scratch memory by hipcc.
#define UR_RMS_CHUNK(C8) do { \
    constexpr int c8 = (C8); \
    const int kc = c8 * KC; \
    asm volatile("" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0); \
    u32x4_t bw[NAD]; \
    if (MASKED) { \
_Pragma("unroll") \
      for (int a = 0; a < NAD; ++a) bw[a] = *reinterpret_cast<const u32x4_t*>(blane[a] + (kc >> 3)); \
    } \
    char* st = smem + (c8 & 1) * STAGE; \
_Pragma("unroll") \
    for (int i = 0; i < NAD; ++i) { \
      const int pi = tid + 256 * i, row = pi >> 4, c16 = pi & 15; \
      *reinterpret_cast<u32x4_t*>(st + (c16 >> 3) * SUB + row * 128 + (((c16 & 7) ^ kc_g(row)) << 4)) = ureg[i]; \
    } \
    __syncthreads(); \
    if (c8 + 1 < D / KC) { \
_Pragma("unroll") \
      for (int i = 0; i < NAD; ++i) ureg[i] = *reinterpret_cast<const u32x4_t*>(ulane[i] + (c8 + 1) * KC); \
    } \
_Pragma("unroll") \
    for (int sx = 0; sx < 4; ++sx) { \
      const int c = 4 * c8 + sx; \
      asm volatile("" ::: "memory"); \
      __builtin_amdgcn_sched_barrier(0); \
 \
      float wv[8]; \
      { \
        const float4 w0 = *reinterpret_cast<const float4*>(wlds + 32 * c + 8 * g), w1 = *reinterpret_cast<const float4*>(wlds + 32 * c + 8 * g + 4); \
        wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w; \
      } \
      const uint32_t wd[4] = {xf[c].x, xf[c].y, xf[c].z, xf[c].w}; \
      uint32_t hw[4]; \
_Pragma("unroll") \
      for (int e = 0; e < 4; ++e) hw[e] = pack_bf2(wv[2 * e] * (bf_lo(wd[e]) * rs), wv[2 * e + 1] * (bf_hi(wd[e]) * rs)); \
      const uint4 hq = make_uint4(hw[0], hw[1], hw[2], hw[3]); \
      if (mok) *reinterpret_cast<uint4*>(hrow + 32 * c) = hq; \
_Pragma("unroll") \
      for (int a = 0; a < NAD; ++a) { \
        const int row = a * 16 + l15, ch = 4 * (sx & 1) + g; \
        const bf16x8 uf = *reinterpret_cast<const bf16x8*>(st + (sx >> 1) * SUB + row * 128 + ((ch ^ kc_g(row)) << 4)); \
        uint4 x = hq; \
        if (MASKED) { \
          const uint32_t wsel = bw[a][sx]; \
          x = drop_apply(x, (wsel >> (8 * g)) & 0xffu); \
        } \
        acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf, __builtin_bit_cast(bf16x8, x), acc[a], 0, 0, 0); \
      } \
    } \
  } while (0)
  UR_RMS_CHUNK(0); UR_RMS_CHUNK(1); UR_RMS_CHUNK(2); UR_RMS_CHUNK(3); UR_RMS_CHUNK(4); UR_RMS_CHUNK(5); UR_RMS_CHUNK(6); UR_RMS_CHUNK(7);
#undef UR_RMS_CHUNK
  // lane holds P[token l15][16 a + 4 g .. + 3]
  if (mok) {
#pragma unroll
    for (int a = 0; a < NAD; ++a) {
      const f32x4 v = acc[a];
      *reinterpret_cast<uint2*>(p.P + (long)m * p.ldp + 16 * a + 4 * g) =
          make_uint2(pack_bf2(v[0] * p.alpha, v[1] * p.alpha), pack_bf2(v[2] * p.alpha, v[3] * p.alpha));
    }
  }
}

// ---- SwiGLU forward + the down_proj adapter's down projection in ONE pass over gate|up ---------------------------------------
// act = silu(gate) * up (Qwen3MLP, modeling_qwen3.py:81-83) leaves as before; t[m, j] = alpha * sum_c keep(m, c) act[m, c] A[j, c]
// is taken from the same registers, so the adapter never re-reads act (805 MB per layer at C4).  lora_project's layout: a
// wave owns 2 x 16 tokens, lane (token l15, column group g) computes 8 consecutive columns of act per k-step from one
// 16-byte piece of gate and one of up; A chunks of 128 columns go through the 2-slot LDS ring.
struct SwiLoraP {
  const bf16_t* GU; long ldgu; bf16_t* ACT; int M, I;
  const bf16_t* U; long ldu;
  const uint8_t* bits; long bits_ld;
  bf16_t* P; long ldp; float alpha;
};
#ifndef UR_SWILORA_DIRECT
#define UR_SWILORA_DIRECT 1      // 1: A fragments straight from L2 (96 KB, resident), no LDS ring and no block barrier
#endif
template <bool MASKED>
__global__ __launch_bounds__(256, 4) void swiglu_lora_kernel(SwiLoraP p) {
  constexpr int RB = 2, KC = 128;
  [[maybe_unused]] constexpr int SUB = 16 * 128, STAGE = 2 * SUB;
#if !UR_SWILORA_DIRECT
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int tok0 = blockIdx.x * (4 * RB * 16) + wave * (RB * 16);
  f32x4 acc[RB];
  const bf16_t* grow[RB]; bf16_t* arow[RB]; const uint8_t* brow[RB]; bool ok[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int tk = tok0 + 16 * rb + l15, m = min(tk, p.M - 1);
    ok[rb] = tk < p.M;
    grow[rb] = p.GU + (long)m * p.ldgu + 8 * g;
    arow[rb] = p.ACT + (long)m * p.I + 8 * g;
    brow[rb] = MASKED ? p.bits + (long)m * p.bits_ld : nullptr;
    acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#if UR_SWILORA_DIRECT
  const bf16_t* ulane = p.U + (long)l15 * p.ldu + 8 * g;                     // MFMA row operand: A[l15, kc + 32 sx + 8 g ..]
#else
  const bf16_t* ulane = p.U + (long)(tid >> 4) * p.ldu + (tid & 15) * 8;      // piece tid = row tid >> 4 of A, 16-byte chunk tid & 15
#endif
  const int nchunks = p.I / KC;
  for (int c = 0; c < nchunks; ++c) {
    const int kc = c * KC;
#if UR_SWILORA_DIRECT
    bf16x8 afr[4];
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) afr[sx] = *reinterpret_cast<const bf16x8*>(ulane + kc + 32 * sx);
#else
    const uint4 ureg = *reinterpret_cast<const uint4*>(ulane + kc);
#endif
    uint4 gf[RB][4], uf4[RB][4], bw[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int sx = 0; sx < 4; ++sx) {
        gf[rb][sx] = ld_stream(grow[rb] + kc + 32 * sx);
        uf4[rb][sx] = ld_stream(grow[rb] + p.I + kc + 32 * sx);
      }
      if (MASKED) bw[rb] = *reinterpret_cast<const uint4*>(brow[rb] + (kc >> 3));
    }
#if !UR_SWILORA_DIRECT
    char* st = smem + (c & 1) * STAGE;
    {
      const int row = tid >> 4, c16 = tid & 15;
      *reinterpret_cast<uint4*>(st + (c16 >> 3) * SUB + row * 128 + (((c16 & 7) ^ kc_g(row)) << 4)) = ureg;
    }
    __syncthreads();       // slot c & 1 is re-written two chunks later: every wave has passed the next barrier by then
#endif
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
#if UR_SWILORA_DIRECT
      const bf16x8 af = afr[sx];
#else
      const int ch = 4 * (sx & 1) + g;
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(st + (sx >> 1) * SUB + l15 * 128 + ((ch ^ kc_g(l15)) << 4));
#endif
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const uint32_t gw[4] = {gf[rb][sx].x, gf[rb][sx].y, gf[rb][sx].z, gf[rb][sx].w};
        const uint32_t uw[4] = {uf4[rb][sx].x, uf4[rb][sx].y, uf4[rb][sx].z, uf4[rb][sx].w};
        uint32_t aw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) aw[e] = pack_bf2(silu_f(bf_lo(gw[e])) * bf_lo(uw[e]), silu_f(bf_hi(gw[e])) * bf_hi(uw[e]));
        uint4 x = make_uint4(aw[0], aw[1], aw[2], aw[3]);
        if (ok[rb]) *reinterpret_cast<uint4*>(arow[rb] + kc + 32 * sx) = x;
        if (MASKED) {
          const uint32_t wsel = sx == 0 ? bw[rb].x : sx == 1 ? bw[rb].y : sx == 2 ? bw[rb].z : bw[rb].w;
          x = drop_apply(x, (wsel >> (8 * g)) & 0xffu);
        }
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, x), acc[rb], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = tok0 + 16 * rb + l15;
    if (m < p.M) {
      const f32x4 v = acc[rb];
      *reinterpret_cast<uint2*>(p.P + (long)m * p.ldp + 4 * g) = make_uint2(pack_bf2(v[0] * p.alpha, v[1] * p.alpha), pack_bf2(v[2] * p.alpha, v[3] * p.alpha));
    }
  }
}

// ---- token-reduction products --------------------------------------------------------------------
struct RedP {
  const bf16_t* X; long ldx; int M;
  int col0[4], width[4]; long goff[4];
  const bf16_t* V; long ldv;
  const uint8_t* bits; long bits_ld, bits_stride;
  const uint32_t* bits_t; long bt_ld, bt_stride;       // token-packed flags (ring kernel)
  float* out; int transposed;
  int tok_per_block;
  float alpha;
};

// one pair of hardware-transposed reads: lane gets [column lane&15 of the 16-column segment][rows r0 .. r0+7]
__device__ __forceinline__ void tr_pair(bf16x8 (&f)[2], uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
  const uint32_t a[2] = {a0, a1}, b[2] = {b0, b1};
  tr_read(f, a, b);
}

// grid: x = 64-column block, y = token split, z = entry (separate column ranges, NAD == 1) or 0 (NAD adapters share X).
// Per 64-token step the block stages X[64 tokens][64 columns] (one masked copy per adapter) and V[64 tokens][16 NAD]
// row-major in LDS; ds_read_b64_tr_b16 turns both into token-packed MFMA operands.  Wave w owns columns 16w..16w+15.
#ifndef UR_RED_ABLATE
#define UR_RED_ABLATE 0     // lab (results WRONG when != 0): 1 = one LDS copy instead of one per adapter, 2 = no transposed reads / MFMAs,
#endif                      // 3 = no flag-byte loads, 4 = no barriers, 5 = no global loads of X
template <int NAD, bool MASKED>
__global__ __launch_bounds__(256) void lora_reduce_kernel(RedP p) {
  constexpr int TOK = NAD <= 3 ? 128 : 64;          // tokens per step (two barriers per step)
  constexpr int XP = TOK / 32;                       // X pieces (16 B) per thread and step
  constexpr int XT = TOK * 128;
  constexpr int VROW = NAD * 32;
  __shared__ __attribute__((aligned(16))) char smem[NAD * XT + TOK * VROW];
  char* vt = smem + NAD * XT;
  const int e0 = blockIdx.z;
  const int W = p.width[e0], col0 = p.col0[e0];
  const int cb = blockIdx.x * 64;
  if (cb >= W) return;                                   // uniform per block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int tbeg = blockIdx.y * p.tok_per_block, tend = min(p.M, tbeg + p.tok_per_block);

  // this thread's staging pieces
  int prow[XP], pch[XP], pcol[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int pi = tid + 256 * i;
    prow[i] = pi >> 3; pch[i] = pi & 7;
    pcol[i] = min(cb + 8 * pch[i], W - 8);
  }
  constexpr int VP = (TOK * 2 * NAD + 255) / 256;          // V pieces (16 B) per thread: TOK rows x 2 NAD pieces
  int vrow[VP], vpart[VP];
  bool vthr[VP];
#pragma unroll
  for (int i = 0; i < VP; ++i) {
    const int pi = tid + 256 * i;
    vthr[i] = pi < TOK * 2 * NAD;
    vrow[i] = pi / (2 * NAD); vpart[i] = pi % (2 * NAD);
  }

  uint4 xr[XP], vr[VP];
  uint32_t br[XP][NAD] = {};
  auto gload = [&](int t0) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int m = min(t0 + prow[i], p.M - 1);
      if (UR_RED_ABLATE != 5) xr[i] = ld_stream(p.X + (long)m * p.ldx + col0 + pcol[i]);
      else xr[i] = make_uint4(m, i, t0, 1);
      if (MASKED && UR_RED_ABLATE != 3) {
#pragma unroll
        for (int a = 0; a < NAD; ++a) br[i][a] = p.bits[(long)a * p.bits_stride + (long)m * p.bits_ld + (pcol[i] >> 3)];
      }
    }
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      vr[i] = make_uint4(0, 0, 0, 0);
      if (vthr[i] && t0 + vrow[i] < tend) vr[i] = *reinterpret_cast<const uint4*>(p.V + (long)(t0 + vrow[i]) * p.ldv + 16 * e0 + 8 * vpart[i]);
    }
  };

  f32x4 acc[NAD];
#pragma unroll
  for (int a = 0; a < NAD; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int q = l15 >> 2, pp = lane & 3;

  if (tbeg < tend) gload(tbeg);
  for (int t0 = tbeg; t0 < tend; t0 += TOK) {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int off = prow[i] * 128 + ((((pch[i] >> 1) ^ f64sw(prow[i])) << 5) | ((pch[i] & 1) << 4));
#pragma unroll
      for (int a = 0; a < (UR_RED_ABLATE == 1 ? 1 : NAD); ++a)
        *reinterpret_cast<uint4*>(smem + a * XT + off) = (MASKED && UR_RED_ABLATE != 3) ? drop_apply(xr[i], br[i][a]) : xr[i];
    }
#pragma unroll
    for (int i = 0; i < VP; ++i)
      if (vthr[i]) *reinterpret_cast<uint4*>(vt + vrow[i] * VROW + vpart[i] * 16) = vr[i];
    if (UR_RED_ABLATE != 4) __syncthreads();
    if (t0 + TOK < tend) gload(t0 + TOK);
#pragma unroll
    for (int ks = 0; ks < (UR_RED_ABLATE == 2 ? 0 : TOK / 32); ++ks) {
      const int ka = 32 * ks + 8 * g + q;
      const uint32_t xo = lds_off(smem) + ka * 128 + ((wave ^ f64sw(ka)) << 5) + pp * 8;
      const uint32_t vo = lds_off(vt) + ka * VROW + pp * 8;
#pragma unroll
      for (int a = 0; a < NAD; ++a) {
        bf16x8 f[2];      // f[0] = X fragment (index: column), f[1] = V fragment (index: rank row j)
        tr_pair(f, xo + a * XT, xo + a * XT + 4 * 128, vo + a * 32, vo + a * 32 + 4 * VROW);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 fa = p.transposed ? f[1] : f[0], fb = p.transposed ? f[0] : f[1];
        acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[a], 0, 0, 0);
      }
    }
    if (UR_RED_ABLATE != 4) __syncthreads();
  }
  // partial (or final) result of this token range, dense layout: entry e at goff[e], [16][W] or [W][16]
  long total = 0;
  {
    const int ne = gridDim.z > 1 ? (int)gridDim.z : NAD;
    total = p.goff[ne - 1] + 16L * p.width[gridDim.z > 1 ? ne - 1 : 0];
  }
  float* base = p.out + (long)blockIdx.y * total;
#pragma unroll
  for (int a = 0; a < NAD; ++a) {
    float* ge = base + p.goff[e0 + a];
    const f32x4 v = acc[a] * p.alpha;
    if (p.transposed) {            // D[j = 4g+e][w = l15]  ->  G[w][j]
      const int w = cb + 16 * wave + l15;
      if (w < W) *reinterpret_cast<float4*>(ge + (long)w * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
    } else {                       // D[w = 4g+e][j = l15]  ->  G[j][w]
      const int w = cb + 16 * wave + 4 * g;
      if (w < W) *reinterpret_cast<float4*>(ge + (long)l15 * W + w) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ---- the token reduction with X streamed through an LDS-DMA ring --------------------------------------------------------
// Same grid and output as lora_reduce_kernel (x = 64-column block, y = token split, z = entry).  A stage = 128 tokens: the raw
// [128 x 64] piece of X (16 KiB, the f64sw pair swizzle of lora_reduce_kernel applied on the SOURCE side of the LDS-DMA), the NAD
// [128 x 16] tiles of V and -- MASKED -- the token-packed flag words of the stage ([4 token groups x 64 columns] per adapter,
// ur_lora_bits_transpose), all fetched three stages ahead of their use.  One barrier per stage; wave w owns columns 16 w .. + 15:
// per 32-token step it reads the transposed X fragment ONCE, masks a copy per adapter in registers (the flag byte of its column
// and 8 tokens: drop_apply) and feeds NAD MFMAs.  No masked LDS copies, no global load between the barriers.
constexpr int R2_TOK = 128, R2_XS = R2_TOK * 128, R2_VS = R2_TOK * 32;
// ring depth: 4 stages and one workgroup per CU for a single adapter; two or more adapters (their flag expansion and MFMA chains make the
// consumer the longer side) run 2 stages and TWO workgroups per CU, one's loads under the other's chains (measured: 104 -> 86 us for 3 adapters)
#ifndef UR_R2_NST1
#define UR_R2_NST1 4         // lab: ring depth of the single-adapter launch
#endif
constexpr int r2_nst(int nad) { return nad == 1 ? UR_R2_NST1 : 2; }
template <int NAD, bool MASKED> constexpr int r2_stage() { return R2_XS + NAD * R2_VS + (MASKED ? NAD * 1024 : 0); }
template <int NAD, bool MASKED>
__global__ __launch_bounds__(256) void lora_reduce_ring_kernel(RedP p) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
  constexpr int STAGE = r2_stage<NAD, MASKED>(), NP = 4 + NAD + (MASKED ? 1 : 0);      // LDS-DMA instructions per wave and stage
  constexpr int R2_NST = r2_nst(NAD);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int e0 = blockIdx.z;
  const int W = p.width[e0], col0 = p.col0[e0];
  const int cb = blockIdx.x * 64;
  if (cb >= W) return;                                   // uniform per block
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = l15 >> 2, pp = lane & 3;
  const int tbeg = blockIdx.y * p.tok_per_block, tend = min(p.M, tbeg + p.tok_per_block);
  const int nst = (tend - tbeg) / R2_TOK;                 // (host: M and the split are multiples of 128)

  // ---- producer ----
  const int prow = lane >> 3, pos = lane & 7, wpar = wave & 1;
  // X piece = rows 8 piece .. + 7 (piece = 4 i + wave): f64sw(row) = bit 1 of prow | (piece & 1) << 1; LDS position pos holds
  // source chunk ((pos >> 1) ^ f64sw) << 1 | (pos & 1)
  const int fsw = ((prow >> 1) & 1) | (wpar << 1);
  const uint32_t xlane = (uint32_t)((prow * p.ldx + ((((pos >> 1) ^ fsw) << 1) | (pos & 1)) * 8) * 2);
  const uint32_t vlane = (uint32_t)((vrow(lane >> 1) * p.ldv + 8 * (lane & 1)) * 2);         // LDS row slot lane >> 1 of the wave's 32-row piece
  const int am = wave < NAD ? wave : NAD - 1;             // this wave's flag piece (waves >= NAD repeat the last adapter's: same bytes)
  const uint32_t mlane = (uint32_t)(((lane >> 4) * p.bt_ld + 4 * (lane & 15)) * 4);
  auto issue = [&](int s) {
    char* st = smem + (s & (R2_NST - 1)) * STAGE;
    const long t0 = tbeg + (long)s * R2_TOK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = 4 * i + wave;
      const char* ub = bg_uniform_ptr(reinterpret_cast<const char*>(p.X + (t0 + 8 * piece) * p.ldx + col0 + cb));
      __builtin_amdgcn_global_load_lds((gbl_void*)(ub + xlane), (lds_void*)(st + piece * 1024), 16, 0, UR_RING_AUX);
    }
#pragma unroll
    for (int a = 0; a < NAD; ++a) {
      const char* ub = bg_uniform_ptr(reinterpret_cast<const char*>(p.V + (t0 + 32 * wave) * p.ldv + 16 * (e0 + a)));
      __builtin_amdgcn_global_load_lds((gbl_void*)(ub + vlane), (lds_void*)(st + R2_XS + a * R2_VS + wave * 1024), 16, 0, 0);
    }
    if (MASKED) {
      const char* ub = bg_uniform_ptr(reinterpret_cast<const char*>(p.bits_t + (long)am * p.bt_stride + (t0 >> 5) * p.bt_ld + cb));
      __builtin_amdgcn_global_load_lds((gbl_void*)(ub + mlane), (lds_void*)(st + R2_XS + NAD * R2_VS + am * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[NAD];
#pragma unroll
  for (int a = 0; a < NAD; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int npro = min(R2_NST - 1, nst);
  for (int s = 0; s < npro; ++s) issue(s);
  // lane constants of the consumer: transposed reads of step ks at rows ka = 32 ks + 8 g + q (+ 4)
  uint32_t xo[4], vo[4], vo4[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int ka = 32 * ks + 8 * g + q;
    xo[ks] = (uint32_t)(ka * 128 + ((wave ^ f64sw(ka)) << 5) + pp * 8);
    vo[ks] = (uint32_t)(R2_XS + vrow(ka) * 32 + pp * 8);
    vo4[ks] = (uint32_t)(R2_XS + vrow(ka + 4) * 32 + pp * 8);
  }
  const uint32_t mo = (uint32_t)(R2_XS + NAD * R2_VS + (16 * wave + l15) * 4);

  for (int s = 0; s < nst; ++s) {
    const int later = min(nst, s + R2_NST - 1) - (s + 1);      // stages issued after stage s (at most R2_NST - 2)
    if (R2_NST > 3 && later >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NP) : "memory");
    else if (R2_NST > 2 && later == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (s + R2_NST - 1 < nst) issue(s + R2_NST - 1);
    __builtin_amdgcn_sched_barrier(0);
    const char* st = smem + (s & (R2_NST - 1)) * STAGE;
    const uint32_t sb = lds_off(st);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 xf, vf[NAD];
      if constexpr (NAD == 1) {
        bf16x8 f[2];
        const uint32_t a[2] = {sb + xo[ks], sb + vo[ks]}, b[2] = {sb + xo[ks] + 4 * 128, sb + vo4[ks]};
        tr_read(f, a, b);
        xf = f[0]; vf[0] = f[1];
      } else {
        bf16x8 f[4];
        uint32_t a[4], b[4];
        a[0] = sb + xo[ks]; b[0] = a[0] + 4 * 128;
#pragma unroll
        for (int i = 1; i < 4; ++i) { const int ad = i - 1 < NAD ? i - 1 : NAD - 1; a[i] = sb + vo[ks] + ad * R2_VS; b[i] = sb + vo4[ks] + ad * R2_VS; }
        tr_read(f, a, b);
        xf = f[0];
#pragma unroll
        for (int ad = 0; ad < NAD && ad < 3; ++ad) vf[ad] = f[1 + ad];
        if constexpr (NAD == 4) {
          bf16x8 f2[2];
          const uint32_t a2[2] = {sb + vo[ks] + 3 * R2_VS, sb + vo[ks] + 3 * R2_VS}, b2[2] = {sb + vo4[ks] + 3 * R2_VS, sb + vo4[ks] + 3 * R2_VS};
          tr_read(f2, a2, b2);
          vf[3] = f2[0];
        }
      }
#pragma unroll
      for (int ad = 0; ad < NAD; ++ad) {
        bf16x8 xm = xf;
        if (MASKED) {
          const uint32_t wd = *reinterpret_cast<const uint32_t*>(st + mo + ad * 1024 + ks * 256);
          xm = __builtin_bit_cast(bf16x8, drop_apply(__builtin_bit_cast(uint4, xf), (wd >> (8 * g)) & 0xffu));
        }
        const bf16x8 fa = p.transposed ? vf[ad] : xm, fb = p.transposed ? xm : vf[ad];
        acc[ad] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[ad], 0, 0, 0);
      }
    }
  }
  // partial (or final) result of this token range, dense layout: entry e at goff[e], [16][W] or [W][16]
  long total = 0;
  {
    const int ne = gridDim.z > 1 ? (int)gridDim.z : NAD;
    total = p.goff[ne - 1] + 16L * p.width[gridDim.z > 1 ? ne - 1 : 0];
  }
  float* base = p.out + (long)blockIdx.y * total;
#pragma unroll
  for (int a = 0; a < NAD; ++a) {
    float* ge = base + p.goff[e0 + a];
    const f32x4 v = acc[a] * p.alpha;
    if (p.transposed) {            // D[j = 4g+e][w = l15]  ->  G[w][j]
      const int w = cb + 16 * wave + l15;
      *reinterpret_cast<float4*>(ge + (long)w * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
    } else {                       // D[w = 4g+e][j = l15]  ->  G[j][w]
      const int w = cb + 16 * wave + 4 * g;
      *reinterpret_cast<float4*>(ge + (long)l15 * W + w) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// token-packed flag words from the row planes: thread = (adapter, 32-token group, 32-column word).  The 32 row words of the group are a
// 32 x 32 bit matrix; its transpose (five masked-swap rounds, Hacker's Delight 7-3 -- that routine yields the ANTI-transpose, so the rows
// enter in reverse order and word 31 - p leaves as the one of input bit p) is the 32 token-packed words up to the two pair orders: input
// bit p = byte p >> 3, element order (0, 2, 4, 6, 1, 3, 5, 7) -> column; a word's token bits leave natural order for the same pair order
// by one in-byte unshuffle (two masked swaps).  Row reads are 4 bytes per lane, contiguous across the wave; 128-byte writes per thread.
__global__ __launch_bounds__(256) void lora_bits_transpose_kernel(const uint8_t* __restrict__ bits, long bits_ld, long bits_stride, int M, int W, int nad,
                                                                  uint32_t* __restrict__ bt, long bt_ld, long bt_stride) {
  const int nquad = (int)(bits_ld >> 2);
  const long per = (long)(M / 32) * nquad;
  const long idx0 = (long)blockIdx.x * 256 + threadIdx.x;
  const bool valid = idx0 < per * nad;                      // (every lane stays: the stores below are a wave-wide copy out of LDS)
  const long idx = valid ? idx0 : per * nad - 1;
  const int a = (int)(idx / per);
  const long r = idx - (long)a * per;
  const int tg = (int)(r / nquad), qd = (int)(r - (long)tg * nquad);
  const uint8_t* src = bits + (long)a * bits_stride + (long)(32 * tg) * bits_ld + 4 * qd;
  uint32_t A[32];
#pragma unroll
  for (int rr = 0; rr < 32; ++rr) A[31 - rr] = *reinterpret_cast<const uint32_t*>(src + (long)rr * bits_ld);
  {
    uint32_t m = 0x0000FFFFu;
#pragma unroll
    for (int j = 16; j != 0; j >>= 1, m ^= m << j) {
#pragma unroll
      for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
        const uint32_t t = (A[k] ^ (A[k + j] >> j)) & m;
        A[k] ^= t; A[k + j] ^= t << j;
      }
    }
  }
  // the thread's 32 words (128 contiguous bytes of the packed row) leave through a wave-private LDS tile so that a store instruction
  // writes 256 contiguous bytes (one word per lane 128 bytes apart costs 64 partial lines per instruction: 52 us against 34)
  __shared__ uint32_t stage[4][64 * 33];
  __shared__ unsigned long long dbase[4][64];
  __shared__ int dqd[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* dst = bt + (long)a * bt_stride + (long)tg * bt_ld + 32 * qd;
  dbase[wave][lane] = valid ? (unsigned long long)(uintptr_t)dst : 0ull;
  dqd[wave][lane] = qd;
#pragma unroll
  for (int p = 0; p < 32; ++p) {
    const int i = p & 7, col = 8 * (p >> 3) + (i < 4 ? 2 * i : 2 * (i - 4) + 1);
    uint32_t x = A[31 - p];
    uint32_t t = (x ^ (x >> 1)) & 0x22222222u; x ^= t ^ (t << 1);
    t = (x ^ (x >> 2)) & 0x0C0C0C0Cu; x ^= t ^ (t << 2);
    stage[wave][lane * 33 + col] = x;
  }
  // (wave-private tile, the wave's LDS operations execute in order: no barrier; lanes past the end carry a zero base)
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int id = 64 * k + lane, sl = id >> 5, col = id & 31;
    if (dbase[wave][sl] != 0ull) {
      uint32_t* d = reinterpret_cast<uint32_t*>((uintptr_t)dbase[wave][sl]);
      if (32 * dqd[wave][sl] + col < (int)bt_ld) d[col] = stage[wave][sl * 33 + col];      // (the source lane's column range decides the padding test)
    }
  }
}

// ---- backward of the B side in ONE pass over dy: tb = alpha * dy_e B_e (column reduction) and the partial
// dB_e^T = t_e^T dy_e of this block's tokens (token reduction) from the same fragments ---------------------
struct BgradP {
  const bf16_t* X; long ldx; int M;
  int col0[4], width[4]; long goff[4];
  const bf16_t* U[4]; long ldu[4];        // B_e^T [16, width_e]
  const bf16_t* V; long ldv;              // t [M, 16 nad]
  bf16_t* P; long ldp;                    // tb [M, 16 nad]
  float* slabs; long total;               // [gridDim.x][total]: partial dB, entry e at goff[e], layout [w][16]
  float alpha;
};

// grid: x = block of 512 tokens, y = adapter entry.  4 waves, each 128 tokens (8 row blocks of 16).  Per 64-column
// chunk a wave loads its [128 tokens x 64 columns] of dy ONCE as MFMA column operands (token on the lane), feeds the
// tb accumulators, parks the same fragments row-major in its private LDS tile and reads them back transposed
// (ds_read_b64_tr_b16) as the operands of dB^T[16 x 64] += t^T[16 x 128] dy[128 x 64]; the four waves' partials are
// summed through LDS and leave as this block's slab.  dy is read once where ur_lora_project + ur_lora_reduce read it twice.
#ifndef UR_BG_ABLATE
#define UR_BG_ABLATE 0      // lab (tools/lab/bgrad_ablate.sh; results WRONG by construction): 1 = no dB phase, 2 = loads + tb MFMAs only, 3 = no cross-wave exchange / barriers
#endif
constexpr int BG_TOK = 512, BG_WTOK = 128;
constexpr int BG_XT = BG_WTOK * 128;                 // a wave's X tile: 128 tokens x 64 columns bf16
constexpr int BG_SMEM = 4 * BG_XT + 4 * 16 * 64 * 4; // + cross-wave reduction of the [16 x 64] f32 partials
__global__ __launch_bounds__(256, 2) void lora_bgrad_kernel(BgradP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int e = blockIdx.y;
  const int W = p.width[e], col0 = p.col0[e];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int q = l15 >> 2, pp = lane & 3;
  const int tok0 = blockIdx.x * BG_TOK + wave * BG_WTOK;
  char* xt = smem + wave * BG_XT;
  float* red = reinterpret_cast<float*>(smem + 4 * BG_XT);

  // t^T fragments of this wave's 128 tokens (MFMA row operand of the dB product: rank row j on l15, 8 tokens per lane):
  // staged once through the wave's tile as [128 tokens][16] rows of 32 bytes, read back transposed
  bf16x8 tT[4];
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pi = lane + 64 * i, r = pi >> 1, part2 = pi & 1;
      const int m = tok0 + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < p.M) v = *reinterpret_cast<const uint4*>(p.V + (long)m * p.ldv + 16 * e + 8 * part2);
      *reinterpret_cast<uint4*>(xt + r * 32 + part2 * 16) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; k += 2) {
      bf16x8 f[2];
      const uint32_t a0 = lds_off(xt) + (32 * k + 8 * g + q) * 32 + pp * 8, a1 = a0 + 32 * 32;
      tr_pair(f, a0, a0 + 4 * 32, a1, a1 + 4 * 32);
      tT[k] = f[0]; tT[k + 1] = f[1];
    }
  }
  f32x4 tb[8];
#pragma unroll
  for (int rb = 0; rb < 8; ++rb) tb[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16_t* xrow[8];
#pragma unroll
  for (int rb = 0; rb < 8; ++rb) xrow[rb] = p.X + (long)min(tok0 + 16 * rb + l15, p.M - 1) * p.ldx + col0;
  const bf16_t* urow = p.U[e] + (long)l15 * p.ldu[e];
  float* slab = p.slabs + (long)blockIdx.x * p.total + p.goff[e];

  for (int c0 = 0; c0 < W; c0 += 64) {
    // this chunk's fragments: dy (token l15 of row block rb, 8 columns) and B^T (rank row l15, the same 8 columns)
    uint4 xf[8][2], uf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int k = c0 + 32 * s2 + 8 * g;
      const bool ok = k < W;
#if UR_BG_ABLATE >= 4
      uf[s2] = *reinterpret_cast<const uint4*>(urow + k);
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) xf[rb][s2] = *reinterpret_cast<const uint4*>(xrow[rb] + k);
#else
      uf[s2] = ok ? *reinterpret_cast<const uint4*>(urow + k) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) xf[rb][s2] = ok ? *reinterpret_cast<const uint4*>(xrow[rb] + k) : make_uint4(0, 0, 0, 0);   // (non-temporal measured slower here)
#endif
    }
    // (the previous chunk's transposed reads of this tile are complete: lgkmcnt(0) below precedes the MFMAs)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) {
#if UR_BG_ABLATE == 5
        tb[rb][0] += __uint_as_float((xf[rb][s2].x ^ xf[rb][s2].y ^ xf[rb][s2].z ^ xf[rb][s2].w ^ uf[s2].x) & 0x3f800000u);
#else
        tb[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, uf[s2]), __builtin_bit_cast(bf16x8, xf[rb][s2]), tb[rb], 0, 0, 0);
#endif
        const int row = 16 * rb + l15, ch = 4 * s2 + g;
#if UR_BG_ABLATE < 2
        *reinterpret_cast<uint4*>(xt + row * 128 + ((ch ^ sw16(row)) << 4)) = xf[rb][s2];
#endif
      }
#if UR_BG_ABLATE == 1 || UR_BG_ABLATE == 2 || UR_BG_ABLATE >= 4
    continue;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own writes have landed (private tile: no barrier)
    f32x4 db[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) db[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ka = 32 * k + 8 * g + q;
#pragma unroll
      for (int cb = 0; cb < 4; cb += 2) {
        bf16x8 f[2];
        // 8-byte piece pp of the 32-byte chunk pair cb: 16-byte chunk 2 cb + (pp >> 1), swizzled per row; row ka + 4 flips
        // bit 2 of the swizzle
        const int swk = sw16(ka), c0 = 2 * cb + (pp >> 1), c1 = c0 + 2;
        const uint32_t base = lds_off(xt) + ka * 128 + ((pp & 1) << 3);
        const uint32_t a0 = base + ((c0 ^ swk) << 4), b0 = base + 4 * 128 + ((c0 ^ swk ^ 4) << 4);
        const uint32_t a1 = base + ((c1 ^ swk) << 4), b1 = base + 4 * 128 + ((c1 ^ swk ^ 4) << 4);
        tr_pair(f, a0, b0, a1, b1);
        db[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tT[k], f[0], db[cb], 0, 0, 0);            // D[j = 4g+e][w = l15]
        db[cb + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tT[k], f[1], db[cb + 1], 0, 0, 0);
      }
    }
#if UR_BG_ABLATE == 3
    if (c0 + 64 < W) { if (db[0][0] + db[1][0] + db[2][0] + db[3][0] == 123.456f) slab[0] = 1.f; continue; }
#endif
    // cross-wave sum: red[wave][w (64)][j (16)]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
      *reinterpret_cast<float4*>(red + wave * 1024 + (16 * cb + l15) * 16 + 4 * g) = make_float4(db[cb][0], db[cb][1], db[cb][2], db[cb][3]);
    __syncthreads();
    {
      const float4 r0 = *reinterpret_cast<const float4*>(red + tid * 4), r1 = *reinterpret_cast<const float4*>(red + 1024 + tid * 4);
      const float4 r2 = *reinterpret_cast<const float4*>(red + 2048 + tid * 4), r3 = *reinterpret_cast<const float4*>(red + 3072 + tid * 4);
      const int w = c0 + (tid >> 2);                        // element 4 tid = (w = tid / 4, j = 4 (tid % 4))
      if (w < W)
        *reinterpret_cast<float4*>(slab + (long)w * 16 + 4 * (tid & 3)) =
            make_float4(r0.x + r1.x + r2.x + r3.x, r0.y + r1.y + r2.y + r3.y, r0.z + r1.z + r2.z + r3.z, r0.w + r1.w + r2.w + r3.w);
    }
    __syncthreads();
  }
  // tb: lane holds rows j = 4 g .. + 3 of token l15
#pragma unroll
  for (int rb = 0; rb < 8; ++rb) {
    const int m = tok0 + 16 * rb + l15;
    if (m < p.M)
      *reinterpret_cast<uint2*>(p.P + (long)m * p.ldp + 16 * e + 4 * g) =
          make_uint2(pack_bf2(tb[rb][0] * p.alpha, tb[rb][1] * p.alpha), pack_bf2(tb[rb][2] * p.alpha, tb[rb][3] * p.alpha));
  }
}

// ---- the same products with dy streamed through an LDS-DMA ring ------------------------------------------------------------
// grid: x = block of 512 tokens, y = adapter entry (every width a multiple of 64).  The block's stream is a sequence of stages
// (column chunk c, token tile t): [256 tokens x 64 columns] of dy (32 KiB, rows of 128 bytes, 16-byte chunks swizzled by
// swr) + the [16 x 64] chunk of B^T, fetched by global_load_lds three stages ahead of their use (no registers, no LDS write
// instructions; 96 KiB in flight per CU).  One barrier per stage makes the tile visible to the four waves, which then split it
// TWO ways: tokens for tb (wave w: tokens 64 w .. + 63 of the tile as MFMA column operands, all 64 columns), columns for dB
// (wave w: columns 16 w .. + 15 as hardware-transposed operands, all 256 tokens) -- the dB partial of a column is complete
// inside one wave, so it leaves straight from the accumulators after the block's last token tile: no cross-wave sum.
#ifndef UR_B2_NST
#define UR_B2_NST 4          // lab: ring depth of lora_bgrad_ring_kernel (a power of two; 2 = one stage in flight)
#endif
constexpr int B2_TILE = 256, B2_NST = UR_B2_NST;
constexpr int B2_XS = B2_TILE * 128, B2_STAGE = B2_XS + 2048;                   // dy tile + B^T chunk
constexpr int b2_smem(int nt) { return B2_NST * B2_STAGE + (nt > 2 ? 0 : nt * B2_TILE * 32); }   // + t [tokens][16] (prologue staging; NT 4: inside the ring)
template <int B2_NT>
__global__ __launch_bounds__(256) void lora_bgrad_ring_kernel(BgradP p) {
  constexpr int B2_TOK = B2_NT * B2_TILE;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int e = blockIdx.y;
  const int W = p.width[e], col0 = p.col0[e];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = l15 >> 2, pp = lane & 3;
  const int tok0 = blockIdx.x * B2_TOK;
  const int nch = W / 64, nst = nch * B2_NT;
  char* tstage = B2_NT > 2 ? smem : smem + B2_NST * B2_STAGE;       // (NT 4: 32 KiB inside the ring, before the first stage is issued)

  // ---- producer: this wave's 8 pieces (8 rows x 128 B each) of a stage's dy tile + one piece of its B^T chunk ----
  // every address = uniform base (scalar registers) + ONE 32-bit lane offset; only the launch's last token block clamps rows per lane
  const int prow = lane >> 3;                                 // row within a piece; rows 8 piece .. of the tile (piece = 4 i + wave): swizzle row 8 (wave & 1) + prow
  const int wpar = wave & 1;
  const int schunk = (lane & 7) ^ swr(8 * wpar + prow);
  const uint32_t xlane = (uint32_t)((prow * p.ldx + schunk * 8) * 2);
  const uint32_t ulane = (uint32_t)((prow * p.ldu[e] + schunk * 8) * 2);
  const char* ubase = reinterpret_cast<const char*>(p.U[e] + (long)(8 * wpar) * p.ldu[e]);
  const bool full = tok0 + B2_TOK <= p.M;                     // uniform
  // (every block walks the column chunks in the same order: a token's tb must not depend on where its row sits in the batch -- the
  // backward keeps the per-sample invariance the data-parallel equivalence tests rest on; rotated orders measured no gain here)
  auto issue = [&](int s) {
    const int c = s / B2_NT, t = s - c * B2_NT;
    char* st = smem + (s & (B2_NST - 1)) * B2_STAGE;
    if (full) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int piece = 4 * i + wave;                       // rows 8 piece .. + 7
        const char* ub = bg_uniform_ptr(reinterpret_cast<const char*>(p.X + (long)(tok0 + t * B2_TILE + 8 * piece) * p.ldx + col0 + 64 * c));
        __builtin_amdgcn_global_load_lds((gbl_void*)(ub + xlane), (lds_void*)(st + piece * 1024), 16, 0, UR_RING_AUX);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int piece = 4 * i + wave;
        const int m = min(tok0 + t * B2_TILE + 8 * piece + prow, p.M - 1);
        const char* src = reinterpret_cast<const char*>(p.X + (long)m * p.ldx + col0 + 64 * c + schunk * 8);
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(st + piece * 1024), 16, 0, UR_RING_AUX);
      }
    }
    const char* uu = bg_uniform_ptr(ubase + 128 * c);
    __builtin_amdgcn_global_load_lds((gbl_void*)(uu + ulane), (lds_void*)(st + B2_XS + wpar * 1024), 16, 0, 0);
  };

  // ---- t^T fragments of the block's 512 tokens (MFMA row operand of the dB product), staged once through LDS ----
  bf16x8 tT[B2_NT][8];
  {
#pragma unroll
    for (int i = 0; i < 2 * B2_NT; ++i) {
      const int pi = tid + 256 * i, r = pi >> 1, part2 = pi & 1;
      const int m = tok0 + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < p.M) v = *reinterpret_cast<const uint4*>(p.V + (long)m * p.ldv + 16 * e + 8 * part2);
      *reinterpret_cast<uint4*>(tstage + r * 32 + part2 * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < B2_NT; ++t)
#pragma unroll
      for (int k = 0; k < 8; k += 4) {
        bf16x8 f[4];
        uint32_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = lds_off(tstage) + (256 * t + 32 * (k + i) + 8 * g + q) * 32 + pp * 8; b[i] = a[i] + 4 * 32; }
        tr_read(f, a, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) tT[t][k + i] = f[i];
      }
  }
  if (B2_NT > 2) __syncthreads();                 // the staging area is ring space
  const int npro = min(B2_NST - 1, nst);
  for (int s = 0; s < npro; ++s) issue(s);

  f32x4 tb[B2_NT][4];
#pragma unroll
  for (int t = 0; t < B2_NT; ++t)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) tb[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float* slab = p.slabs + (long)blockIdx.x * p.total + p.goff[e];
  // lane constants of the consumers
  const uint32_t xrd0 = (uint32_t)((64 * wave + l15) * 128);             // row of row block 0; chunk position (4 s2 + g) ^ swr(l15)
  const int swl = swr(l15);
  uint32_t tra[8], trb[8];                                                // transposed reads of k-step k: rows 32 k + 8 g + q (+ 4)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ka = 32 * k + 8 * g + q, c0 = 2 * wave + (pp >> 1);
    tra[k] = (uint32_t)(ka * 128 + ((pp & 1) << 3) + ((c0 ^ swr(ka)) << 4));
    trb[k] = (uint32_t)((ka + 4) * 128 + ((pp & 1) << 3) + ((c0 ^ swr(ka + 4)) << 4));
  }

  int s = 0;
  for (int c = 0; c < nch; ++c) {
    f32x4 db = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < B2_NT; ++t, ++s) {
      // stage s has landed (this wave's pieces: the stages issued after it may be in flight), then everyone's
      const int later = min(nst, s + B2_NST - 1) - (s + 1);
      if (later >= 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // (every wave has finished stage s - 1: its slot takes stage s + 3)
      if (s + B2_NST - 1 < nst) issue(s + B2_NST - 1);
      __builtin_amdgcn_sched_barrier(0);
      const char* st = smem + (s & (B2_NST - 1)) * B2_STAGE;
      bf16x8 uf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) uf[s2] = *reinterpret_cast<const bf16x8*>(st + B2_XS + l15 * 128 + (((4 * s2 + g) ^ swl) << 4));
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(st + xrd0 + rb * 2048 + (((4 * s2 + g) ^ swl) << 4));
          tb[t][rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uf[s2], xf, tb[t][rb], 0, 0, 0);
        }
      const uint32_t sb = lds_off(st);
#pragma unroll
      for (int k = 0; k < 8; k += 4) {
        bf16x8 f[4];
        const uint32_t a[4] = {sb + tra[k], sb + tra[k + 1], sb + tra[k + 2], sb + tra[k + 3]};
        const uint32_t b[4] = {sb + trb[k], sb + trb[k + 1], sb + trb[k + 2], sb + trb[k + 3]};
        tr_read(f, a, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) db = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tT[t][k + i], f[i], db, 0, 0, 0);     // D[j = 4g+e][w = l15]
      }
    }
    *reinterpret_cast<float4*>(slab + (long)(64 * c + 16 * wave + l15) * 16 + 4 * g) = make_float4(db[0], db[1], db[2], db[3]);
  }
  // tb: lane holds rows j = 4 g .. + 3 of token l15
#pragma unroll
  for (int t = 0; t < B2_NT; ++t)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int m = tok0 + t * B2_TILE + 64 * wave + 16 * rb + l15;
      if (m < p.M)
        *reinterpret_cast<uint2*>(p.P + (long)m * p.ldp + 16 * e + 4 * g) =
            make_uint2(pack_bf2(tb[t][rb][0] * p.alpha, tb[t][rb][1] * p.alpha), pack_bf2(tb[t][rb][2] * p.alpha, tb[t][rb][3] * p.alpha));
    }
}

// out = sum over `splits` slabs of total4 float4 each: 1024 threads = 64 float4 x 16 groups of slabs (a thread per
// element walking 256 slabs serially took 64 us for 64 K floats)
__global__ __launch_bounds__(1024) void slab_sum_kernel(const float* __restrict__ ws, float* __restrict__ out, long total4, int splits) {
  __shared__ float4 red[16][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  const float4* w4 = reinterpret_cast<const float4*>(ws);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < total4)
    for (int z = g; z < splits; z += 16) {
      const float4 b = w4[i + (long)z * total4];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
  red[g][lane] = a;
  __syncthreads();
  if (g == 0 && i < total4) {
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 b = red[k][lane]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

int validate_common(const ur_lora_args* a, const char* who) {
  UR_REQUIRE(a != nullptr, "%s: null args", who);
  UR_REQUIRE(a->M >= 0 && a->nad >= 1 && a->nad <= 4, "%s: M >= 0 and 1 <= nad <= 4 required (M=%d nad=%d)", who, a->M, a->nad);
  UR_REQUIRE(a->rank == 16, "%s: the dedicated LoRA kernels are built for rank 16 (got %d)", who, a->rank);
  UR_REQUIRE(a->X && UR_ALIGNED16(a->X) && (a->ldx % 8) == 0, "%s: X must be 16-byte aligned with ldx %% 8 == 0", who);
  const int ne = a->shared ? 1 : a->nad;
  for (int e = 0; e < ne; ++e) {
    UR_REQUIRE(a->width[e] >= 8 && (a->width[e] % 8) == 0 && a->col0[e] >= 0 && (a->col0[e] % 8) == 0 &&
               (int64_t)a->col0[e] + a->width[e] <= a->ldx, "%s: entry %d column range [%d, +%d) invalid (multiples of 8 inside ldx)",
               who, e, a->col0[e], a->width[e]);
  }
  if (a->drop_bits) {
    UR_REQUIRE(a->shared, "%s: dropout bit planes need adapters that share their input", who);
    UR_REQUIRE(a->col0[0] == 0, "%s: dropout bit planes cover the input from column 0", who);
    UR_REQUIRE((a->bits_ld % 16) == 0 && a->bits_ld * 8 >= a->width[0] && UR_ALIGNED16(a->drop_bits) && (a->bits_stride % 16) == 0,
               "%s: bit planes need 16-byte aligned rows (bits_ld %% 16 == 0, bits_ld * 8 >= width)", who);
  }
  return 0;
}

}  // namespace

extern "C" int64_t ur_lora_bits_ld(int32_t W) { return ((int64_t)W + 127) / 128 * 16; }

extern "C" int ur_lora_dropout_bits(uint64_t seed, float p, int32_t M, int32_t W, int32_t nad, uint8_t* bits, int64_t bits_ld,
                                    int64_t bits_stride, int64_t row0, void* stream) {
  UR_REQUIRE(p >= 0.f && p < 1.f && M >= 0 && W > 0 && nad >= 1 && nad <= 4, "ur_lora_dropout_bits: bad argument");
  UR_REQUIRE(bits_ld == ur_lora_bits_ld(W) && bits_stride >= (int64_t)M * bits_ld && (bits_stride % 16) == 0 && (M == 0 || (bits && UR_ALIGNED16(bits))),
             "ur_lora_dropout_bits: bits_ld must be ur_lora_bits_ld(W), planes 16-byte aligned and at least M rows apart");
  if (M == 0) return 0;
  double t = (double)p * 32768.0 + 0.5;
  const uint32_t thr15 = t > 32767.0 ? 32767u : (uint32_t)t;
  UR_REQUIRE(bits_ld / 4 <= 2048, "ur_lora_dropout_bits: W up to 65536 columns");
  const long rows_y = 1L << 20;
  const long n = (long)(M < rows_y ? M : rows_y) * (bits_ld / 4);         // threads per grid.y slice (< 2^31)
  hipLaunchKernelGGL(lora_bits_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)((M + rows_y - 1) / rows_y)), dim3(256), 0, (hipStream_t)stream, seed, thr15, (int)M, (int)W,
                     (int)nad, (long)bits_ld, (long)bits_stride, bits, (long)row0);
  UR_CHECK_LAUNCH("ur_lora_dropout_bits");
  return 0;
}

extern "C" int ur_lora_project(const ur_lora_args* a, void* stream) {
  if (int rc = validate_common(a, "ur_lora_project")) return rc;
  UR_REQUIRE(a->P && (((uintptr_t)a->P) & 7) == 0 && (a->ldp % 4) == 0 && a->ldp >= 16 * a->nad, "ur_lora_project: P must be 8-byte aligned, ldp %% 4 == 0, ldp >= 16 nad");
  for (int e = 0; e < a->nad; ++e)
    UR_REQUIRE(a->U[e] && UR_ALIGNED16(a->U[e]) && (a->ldu[e] % 8) == 0 && a->ldu[e] >= a->width[a->shared ? 0 : e],
               "ur_lora_project: U[%d] must be a 16-byte aligned [16, width] bf16 matrix (ldu %% 8 == 0)", e);
  if (a->M == 0) return 0;
  ProjP p;
  p.X = (const bf16_t*)a->X; p.ldx = a->ldx; p.M = a->M;
  for (int e = 0; e < 4; ++e) {
    const int s = a->shared ? 0 : (e < a->nad ? e : 0);
    p.col0[e] = a->col0[s]; p.width[e] = a->width[s];
    p.U[e] = (const bf16_t*)a->U[e < a->nad ? e : 0]; p.ldu[e] = a->ldu[e < a->nad ? e : 0];
  }
  p.bits = (const uint8_t*)a->drop_bits; p.bits_ld = a->bits_ld; p.bits_stride = a->bits_stride;
  p.P = (bf16_t*)a->P; p.ldp = a->ldp; p.alpha = a->alpha;
  hipStream_t st = (hipStream_t)stream;
  const unsigned gx = (unsigned)ur_cdiv(a->M, 128);
  const bool masked = a->drop_bits != nullptr;
  bool ring = !a->shared || a->nad == 1;           // one adapter per entry, every width a multiple of 64: wave-private LDS-DMA rings
  for (int e = 0; ring && e < (a->shared ? 1 : a->nad); ++e) ring = (a->width[e] % 64) == 0;
  if (ring) {
    static std::atomic<uint64_t> attr_set[2];      // per device, per kernel
    const void* fn = masked ? reinterpret_cast<const void*>(&lora_project_ring_kernel<true>) : reinterpret_cast<const void*>(&lora_project_ring_kernel<false>);
    UR_ONCE_PER_DEVICE(attr_set[masked ? 1 : 0]) {
      hipError_t er = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, P2_SMEM);
      if (er != hipSuccess) UR_FAIL((int)er, "ur_lora_project: hipFuncSetAttribute failed: %s", hipGetErrorString(er));
    }
    dim3 grid((unsigned)ur_cdiv(a->M, 256), a->nad);
    if (masked) hipLaunchKernelGGL((lora_project_ring_kernel<true>), grid, dim3(256), P2_SMEM, st, p);
    else hipLaunchKernelGGL((lora_project_ring_kernel<false>), grid, dim3(256), P2_SMEM, st, p);
  } else
  if (!a->shared || a->nad == 1) {
    dim3 grid(gx, a->nad);
    if (masked) hipLaunchKernelGGL((lora_project_kernel<1, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((lora_project_kernel<1, false>), grid, dim3(256), 0, st, p);
  } else {
    dim3 grid(gx, 1);
#define UR_PROJ(NAD)                                                                          \
    if (masked) hipLaunchKernelGGL((lora_project_kernel<NAD, true>), grid, dim3(256), 0, st, p); \
    else hipLaunchKernelGGL((lora_project_kernel<NAD, false>), grid, dim3(256), 0, st, p)
    if (a->nad == 2) { UR_PROJ(2); } else if (a->nad == 3) { UR_PROJ(3); } else { UR_PROJ(4); }
#undef UR_PROJ
  }
  UR_CHECK_LAUNCH("ur_lora_project");
  return 0;
}

extern "C" int ur_rmsnorm_lora_fwd(const void* x, const float* w, void* out, float* rstd, int32_t M, int32_t D, float eps,
                                   const ur_lora_args* a, void* stream) {
  UR_REQUIRE(D == 1024, "ur_rmsnorm_lora_fwd: built for the hidden size 1024 (D=%d)", D);
  UR_REQUIRE(M >= 0 && x && w && out && rstd && UR_ALIGNED16(x) && UR_ALIGNED16(w) && UR_ALIGNED16(out), "ur_rmsnorm_lora_fwd: null / misaligned");
  UR_REQUIRE(a && a->nad >= 2 && a->nad <= 3 && a->rank == 16 && a->shared == 1, "ur_rmsnorm_lora_fwd: 2 or 3 rank-16 adapters sharing the normalised input");
  UR_REQUIRE(a->P && (((uintptr_t)a->P) & 7) == 0 && (a->ldp % 4) == 0 && a->ldp >= 16 * a->nad, "ur_rmsnorm_lora_fwd: P must be 8-byte aligned, ldp %% 4 == 0, ldp >= 16 nad");
  UR_REQUIRE(!a->drop_bits || (UR_ALIGNED16(a->drop_bits) && (a->bits_ld % 16) == 0 && a->bits_ld >= D / 8 && (a->bits_stride % 16) == 0),
             "ur_rmsnorm_lora_fwd: bad dropout bit planes");
  for (int e = 0; e < a->nad; ++e)
    UR_REQUIRE(a->U[e] && UR_ALIGNED16(a->U[e]) && (a->ldu[e] % 8) == 0 && a->ldu[e] >= D, "ur_rmsnorm_lora_fwd: U[%d] must be a 16-byte aligned [16, D] bf16 matrix", e);
  if (M == 0) return 0;
  RmsLoraP p;
  p.X = (const bf16_t*)x; p.W = w; p.H = (bf16_t*)out; p.rstd = rstd; p.M = M; p.eps = eps;
  for (int e = 0; e < 4; ++e) { p.U[e] = (const bf16_t*)a->U[e < a->nad ? e : 0]; p.ldu[e] = a->ldu[e < a->nad ? e : 0]; }
  p.bits = (const uint8_t*)a->drop_bits; p.bits_ld = a->bits_ld; p.bits_stride = a->bits_stride;
  p.P = (bf16_t*)a->P; p.ldp = a->ldp; p.alpha = a->alpha;
  const dim3 grid((unsigned)ur_cdiv(M, 64));
  hipStream_t st = (hipStream_t)stream;
  const bool masked = a->drop_bits != nullptr;
  if (a->nad == 2) {
    if (masked) hipLaunchKernelGGL((rms_lora_kernel<2, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((rms_lora_kernel<2, false>), grid, dim3(256), 0, st, p);
  } else {
    if (masked) hipLaunchKernelGGL((rms_lora_kernel<3, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((rms_lora_kernel<3, false>), grid, dim3(256), 0, st, p);
  }
  UR_CHECK_LAUNCH("ur_rmsnorm_lora_fwd");
  return 0;
}

extern "C" int ur_swiglu_lora_fwd(const void* gu, void* act, int32_t M, int32_t I, const ur_lora_args* a, void* stream) {
  UR_REQUIRE(M >= 0 && I > 0 && (I % 128) == 0, "ur_swiglu_lora_fwd: I must be a multiple of 128 (I=%d)", I);
  UR_REQUIRE(gu && act && UR_ALIGNED16(gu) && UR_ALIGNED16(act), "ur_swiglu_lora_fwd: null / misaligned");
  UR_REQUIRE(a && a->nad == 1 && a->rank == 16, "ur_swiglu_lora_fwd: one rank-16 adapter (down_proj)");
  UR_REQUIRE(a->P && (((uintptr_t)a->P) & 7) == 0 && (a->ldp % 4) == 0 && a->ldp >= 16, "ur_swiglu_lora_fwd: P must be 8-byte aligned, ldp %% 4 == 0, ldp >= 16");
  UR_REQUIRE(a->U[0] && UR_ALIGNED16(a->U[0]) && (a->ldu[0] % 8) == 0 && a->ldu[0] >= I, "ur_swiglu_lora_fwd: U must be a 16-byte aligned [16, I] bf16 matrix");
  UR_REQUIRE(!a->drop_bits || (UR_ALIGNED16(a->drop_bits) && (a->bits_ld % 16) == 0 && a->bits_ld >= I / 8), "ur_swiglu_lora_fwd: bad dropout bit plane");
  if (M == 0) return 0;
  SwiLoraP p;
  p.GU = (const bf16_t*)gu; p.ldgu = 2L * I; p.ACT = (bf16_t*)act; p.M = M; p.I = I;
  p.U = (const bf16_t*)a->U[0]; p.ldu = a->ldu[0];
  p.bits = (const uint8_t*)a->drop_bits; p.bits_ld = a->bits_ld;
  p.P = (bf16_t*)a->P; p.ldp = a->ldp; p.alpha = a->alpha;
  const dim3 grid((unsigned)ur_cdiv(M, 128));
  if (a->drop_bits) hipLaunchKernelGGL((swiglu_lora_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((swiglu_lora_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, p);
  UR_CHECK_LAUNCH("ur_swiglu_lora_fwd");
  return 0;
}

// the ring kernel takes launches whose token count and every width are multiples of 128 / 64 and, under dropout, come with the
// token-packed flags; (the split below is then a multiple of 128 tokens by construction)
static inline bool lora_reduce_ring_ok(const ur_lora_args* a) {
  if ((a->M % 128) != 0 || a->M < 128) return false;
  const int ne = a->shared ? 1 : a->nad;
  for (int e = 0; e < ne; ++e)
    if ((a->width[e] % 64) != 0) return false;
  if (a->drop_bits != nullptr) {
    if (a->drop_bits_t == nullptr || (a->bits_t_ld % 4) != 0 || a->bits_t_ld < a->width[0] || !UR_ALIGNED16(a->drop_bits_t) || (a->bits_t_stride % 4) != 0) return false;
  }
  return true;
}
template <int NAD, bool MASKED>
static int launch_reduce_ring(const RedP& p, dim3 grid, hipStream_t st) {
  constexpr int SMEM = r2_nst(NAD) * r2_stage<NAD, MASKED>();
  static std::atomic<uint64_t> attr_set{0};      // per device
  UR_ONCE_PER_DEVICE(attr_set) {
    hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(&lora_reduce_ring_kernel<NAD, MASKED>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (er != hipSuccess) UR_FAIL((int)er, "ur_lora_reduce: hipFuncSetAttribute failed: %s", hipGetErrorString(er));
  }
  hipLaunchKernelGGL((lora_reduce_ring_kernel<NAD, MASKED>), grid, dim3(256), SMEM, st, p);
  return 0;
}
static inline int lora_reduce_splits(const ur_lora_args* a) {
  int wmax = 0;
  const int ne = a->shared ? 1 : a->nad;
  for (int e = 0; e < ne; ++e) wmax = a->width[e] > wmax ? a->width[e] : wmax;
  const long colblocks = (long)ur_cdiv(wmax, 64) * ne;
  const int tiles = ur_cdiv(a->M, 128);
  // ~8 blocks per CU for the register-staged kernel; the ring kernel runs one workgroup per CU: 2 rounds of them
  // ring: two rounds of the workgroups a CU holds (one for a single adapter, two otherwise), never a partial third
  const long ring_wg = ((a->shared && a->nad > 1) || r2_nst(1) == 2 ? 4L : 2L) * ur_device_cu_count();
  long want = lora_reduce_ring_ok(a) ? ring_wg / colblocks : (2048L + colblocks - 1) / colblocks;
  if (want < 1) want = 1;
  if (want > tiles) want = tiles;
  if (want > 256) want = 256;
  const int tiles_per = ur_cdiv(tiles, (int)want);
  return ur_cdiv(tiles, tiles_per);
}
static inline int64_t lora_reduce_total(const ur_lora_args* a) {
  int64_t t = 0;
  for (int e = 0; e < a->nad; ++e) t += 16LL * a->width[a->shared ? 0 : e];
  return t;
}

extern "C" int64_t ur_lora_bits_t_ld(int32_t W) { return ((int64_t)W + 3) / 4 * 4; }

extern "C" int ur_lora_bits_transpose(const uint8_t* bits, int64_t bits_ld, int64_t bits_stride, int32_t M, int32_t W, int32_t nad,
                                      uint32_t* bits_t, int64_t bits_t_ld, int64_t bits_t_stride, void* stream) {
  UR_REQUIRE(bits && bits_t && M >= 0 && (M % 32) == 0 && W > 0 && nad >= 1 && nad <= 4, "ur_lora_bits_transpose: bad argument (M %% 32 == 0, 1 <= nad <= 4)");
  UR_REQUIRE(bits_ld * 8 >= W && (bits_ld % 4) == 0 && (((uintptr_t)bits) & 3) == 0 && (bits_stride % 4) == 0 && bits_t_ld >= W && (bits_t_ld % 4) == 0 && UR_ALIGNED16(bits_t) && (bits_t_stride % 4) == 0 &&
             bits_t_stride >= (int64_t)(M / 32) * bits_t_ld, "ur_lora_bits_transpose: row / plane strides");
  if (M == 0) return 0;
  const long n = (long)(M / 32) * (bits_ld / 4) * nad;
  hipLaunchKernelGGL(lora_bits_transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bits, (long)bits_ld, (long)bits_stride,
                     (int)M, (int)W, (int)nad, bits_t, (long)bits_t_ld, (long)bits_t_stride);
  UR_CHECK_LAUNCH("ur_lora_bits_transpose");
  return 0;
}

extern "C" int64_t ur_lora_reduce_workspace_bytes(const ur_lora_args* a) {
  if (!a || a->M <= 0 || a->nad < 1 || a->nad > 4) return 0;
  const int splits = lora_reduce_splits(a);
  return splits > 1 ? (int64_t)splits * lora_reduce_total(a) * (int64_t)sizeof(float) : 0;
}

extern "C" int ur_lora_reduce(const ur_lora_args* a, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = validate_common(a, "ur_lora_reduce")) return rc;
  UR_REQUIRE(a->V && UR_ALIGNED16(a->V) && (a->ldv % 8) == 0 && a->ldv >= 16 * a->nad, "ur_lora_reduce: V must be a 16-byte aligned [M, 16 nad] bf16 matrix");
  UR_REQUIRE(a->G && UR_ALIGNED16(a->G), "ur_lora_reduce: G must be 16-byte aligned");
  const int64_t total = lora_reduce_total(a);
  hipStream_t st = (hipStream_t)stream;
  if (a->M == 0) {
    hipError_t e = hipMemsetAsync(a->G, 0, (size_t)total * sizeof(float), st);
    if (e != hipSuccess) UR_FAIL((int)e, "ur_lora_reduce: memset failed");
    return 0;
  }
  const int splits = lora_reduce_splits(a);
  if (splits > 1)
    UR_REQUIRE(workspace && UR_ALIGNED16(workspace) && workspace_bytes >= ur_lora_reduce_workspace_bytes(a),
               "ur_lora_reduce: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)ur_lora_reduce_workspace_bytes(a));
  RedP p;
  p.X = (const bf16_t*)a->X; p.ldx = a->ldx; p.M = a->M;
  long off = 0;
  int wmax = 0;
  for (int e = 0; e < 4; ++e) {
    const int s = a->shared ? 0 : (e < a->nad ? e : 0);
    p.col0[e] = a->col0[s]; p.width[e] = a->width[s];
    p.goff[e] = off;
    if (e < a->nad) { off += 16L * a->width[s]; wmax = a->width[s] > wmax ? a->width[s] : wmax; }
  }
  p.V = (const bf16_t*)a->V; p.ldv = a->ldv;
  p.bits = (const uint8_t*)a->drop_bits; p.bits_ld = a->bits_ld; p.bits_stride = a->bits_stride;
  p.bits_t = (const uint32_t*)a->drop_bits_t; p.bt_ld = a->bits_t_ld; p.bt_stride = a->bits_t_stride;
  p.out = splits > 1 ? (float*)workspace : (float*)a->G;
  p.transposed = a->g_transposed ? 1 : 0;
  const int tiles = ur_cdiv(a->M, 128);
  p.tok_per_block = ur_cdiv(tiles, splits) * 128;
  p.alpha = a->alpha;
  const bool masked = a->drop_bits != nullptr;
  if (lora_reduce_ring_ok(a)) {
    // X through the LDS-DMA ring (one workgroup per CU: the splits above were sized for it)
    const bool per_entry = !a->shared || a->nad == 1;
    dim3 grid(ur_cdiv(wmax, 64), splits, per_entry ? a->nad : 1);
    const int nad_k = per_entry ? 1 : a->nad;
    int rc = 0;
#define UR_RING(NAD, MK) rc = launch_reduce_ring<NAD, MK>(p, grid, st)
    if (masked) { if (nad_k == 1) UR_RING(1, true); else if (nad_k == 2) UR_RING(2, true); else if (nad_k == 3) UR_RING(3, true); else UR_RING(4, true); }
    else { if (nad_k == 1) UR_RING(1, false); else if (nad_k == 2) UR_RING(2, false); else if (nad_k == 3) UR_RING(3, false); else UR_RING(4, false); }
#undef UR_RING
    if (rc) return rc;
  } else
  if (!a->shared || a->nad == 1) {
    dim3 grid(ur_cdiv(wmax, 64), splits, a->nad);
    if (masked) hipLaunchKernelGGL((lora_reduce_kernel<1, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((lora_reduce_kernel<1, false>), grid, dim3(256), 0, st, p);
  } else {
    dim3 grid(ur_cdiv(wmax, 64), splits, 1);
#define UR_RED(NAD)                                                                          \
    if (masked) hipLaunchKernelGGL((lora_reduce_kernel<NAD, true>), grid, dim3(256), 0, st, p); \
    else hipLaunchKernelGGL((lora_reduce_kernel<NAD, false>), grid, dim3(256), 0, st, p)
    if (a->nad == 2) { UR_RED(2); } else if (a->nad == 3) { UR_RED(3); } else { UR_RED(4); }
#undef UR_RED
  }
  UR_CHECK_LAUNCH("ur_lora_reduce");
  if (splits > 1) {
    const long total4 = total / 4;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((total4 + 63) / 64)), dim3(1024), 0, st, (const float*)workspace, (float*)a->G, total4, splits);
    UR_CHECK_LAUNCH("ur_lora_reduce(slab_sum)");
  }
  return 0;
}

extern "C" int64_t ur_lora_bgrad_workspace_bytes(const ur_lora_args* a) {
  if (!a || a->M <= 0 || a->nad < 1 || a->nad > 4) return 0;
  int64_t total = 0;
  for (int e = 0; e < a->nad; ++e) total += 16LL * a->width[e];
  return (int64_t)ur_cdiv(a->M, BG_TOK) * total * (int64_t)sizeof(float);
}

extern "C" int ur_lora_bgrad(const ur_lora_args* a, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = validate_common(a, "ur_lora_bgrad")) return rc;
  UR_REQUIRE(!a->shared && !a->drop_bits, "ur_lora_bgrad: adapters own column ranges of X (shared = 0), no dropout planes");
  UR_REQUIRE(a->P && (((uintptr_t)a->P) & 7) == 0 && (a->ldp % 4) == 0 && a->ldp >= 16 * a->nad, "ur_lora_bgrad: P must be 8-byte aligned, ldp %% 4 == 0, ldp >= 16 nad");
  UR_REQUIRE(a->V && UR_ALIGNED16(a->V) && (a->ldv % 8) == 0 && a->ldv >= 16 * a->nad, "ur_lora_bgrad: V must be a 16-byte aligned [M, 16 nad] bf16 matrix");
  UR_REQUIRE(a->G && UR_ALIGNED16(a->G), "ur_lora_bgrad: G must be 16-byte aligned");
  for (int e = 0; e < a->nad; ++e)
    UR_REQUIRE(a->U[e] && UR_ALIGNED16(a->U[e]) && (a->ldu[e] % 8) == 0 && a->ldu[e] >= a->width[e],
               "ur_lora_bgrad: U[%d] must be a 16-byte aligned [16, width] bf16 matrix (ldu %% 8 == 0)", e);
  hipStream_t st = (hipStream_t)stream;
  BgradP p;
  long off = 0;
  for (int e = 0; e < 4; ++e) {
    const int s = e < a->nad ? e : 0;
    p.col0[e] = a->col0[s]; p.width[e] = a->width[s]; p.goff[e] = off;
    p.U[e] = (const bf16_t*)a->U[s]; p.ldu[e] = a->ldu[s];
    if (e < a->nad) off += 16L * a->width[s];
  }
  const int64_t total = off;
  if (a->M == 0) {
    hipError_t er = hipMemsetAsync(a->G, 0, (size_t)total * sizeof(float), st);
    if (er != hipSuccess) UR_FAIL((int)er, "ur_lora_bgrad: memset failed");
    return 0;
  }
  UR_REQUIRE(workspace && UR_ALIGNED16(workspace) && workspace_bytes >= ur_lora_bgrad_workspace_bytes(a),
             "ur_lora_bgrad: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)ur_lora_bgrad_workspace_bytes(a));
  p.X = (const bf16_t*)a->X; p.ldx = a->ldx; p.M = a->M;
  p.V = (const bf16_t*)a->V; p.ldv = a->ldv;
  p.P = (bf16_t*)a->P; p.ldp = a->ldp;
  p.slabs = (float*)workspace; p.total = total; p.alpha = a->alpha;
  // every width a multiple of 64 (the decoder's are): dy goes through the LDS-DMA ring, in blocks of 1024 tokens where that still
  // gives every CU a block (half the slabs), else 512; ragged widths keep the register-staged kernel
  bool ring = true;
  for (int e = 0; e < a->nad; ++e) ring = ring && (a->width[e] % 64) == 0;
  const int kind = !ring ? 0 : ((long)ur_cdiv(a->M, 1024) * a->nad >= ur_device_cu_count() ? 2 : 1);
  const void* fn = kind == 0 ? reinterpret_cast<const void*>(&lora_bgrad_kernel)
                 : kind == 1 ? reinterpret_cast<const void*>(&lora_bgrad_ring_kernel<2>) : reinterpret_cast<const void*>(&lora_bgrad_ring_kernel<4>);
  const int smem = kind == 0 ? BG_SMEM : b2_smem(kind == 1 ? 2 : 4);
  static std::atomic<uint64_t> attr_set[3];      // per device, per kernel
  UR_ONCE_PER_DEVICE(attr_set[kind]) {
    hipError_t er = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (er != hipSuccess) UR_FAIL((int)er, "ur_lora_bgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(er));
  }
  const int nblk = ur_cdiv(a->M, kind == 2 ? 1024 : BG_TOK);
  if (kind == 2) hipLaunchKernelGGL(lora_bgrad_ring_kernel<4>, dim3(nblk, a->nad), dim3(256), smem, st, p);
  else if (kind == 1) hipLaunchKernelGGL(lora_bgrad_ring_kernel<2>, dim3(nblk, a->nad), dim3(256), smem, st, p);
  else hipLaunchKernelGGL(lora_bgrad_kernel, dim3(nblk, a->nad), dim3(256), BG_SMEM, st, p);
  UR_CHECK_LAUNCH("ur_lora_bgrad");
  const long total4 = total / 4;
  hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((total4 + 63) / 64)), dim3(1024), 0, st, (const float*)workspace, (float*)a->G, total4, nblk);
  UR_CHECK_LAUNCH("ur_lora_bgrad(slab_sum)");
  return 0;
}

// Shared between gemm.hip (generic tiles) and gemm_pers.hip (persistent 256x256 projection kernel).
#pragma once
#include "common.hip.h"

namespace urgemm {

struct GemmP {
  const bf16_t* R; const bf16_t* S; long ldr, lds; int K;
  const bf16_t* R2; const bf16_t* S2; long ldr2, lds2; int K2;
  void* C; long ldc; int M, N; float alpha;
  const float* bias; const bf16_t* res; long ldres;
  bf16_t* gelu_out; long ldg; const bf16_t* aux; long ldaux;
  int ksplit_len; long slab_stride;
  int gm, gn;
  int gcw;    // column-chunk width (tiles) of the per-XCD tile order; 0 = plain row-major runs
  int stagger;   // cycles between the start groups of the launch's first wave of workgroups (0 = all start together)
  // LoRA dropout (ur_gemm_args.drop_*): masked rank-r LoRA epilogue driven by the adapters' dropped-flag bit planes
  const uint8_t* drop_bits; long drop_bits_ld, drop_bits_stride; int drop_rank; float drop_inv_keep;
  // SwiGLU backward epilogue (ur_gemm_args.swiglu_*): the result is d(act); dgate / dup leave instead of C
  // (forward epilogue, sw_mode 2: sw_gu = gate, sw_dgu = act)
  const bf16_t* sw_gu; long sw_ldgu; bf16_t* sw_dgu; long sw_lddgu; int sw_I; int sw_mode;
  // q/k-norm + RoPE epilogue of the q|k|v projection (ur_gemm_args.qkr_*; persistent kernel only)
  bf16_t* qk_q; long qk_ldq; bf16_t* qk_k; long qk_ldk; bf16_t* qk_v; long qk_ldv; float* qk_rstd;
  const float* qk_qw; const float* qk_kw; const float* qk_cos; const float* qk_sin; int qk_S, qk_nq, qk_nk; float qk_eps;
  // SwiGLU forward epilogue of the merged gate|up projection with 128-row interleaved weights (ur_gemm_args.swp_*; persistent kernel only)
  bf16_t* sp_act; long sp_ldact; int sp_I;
};

// grouped launches (gemm.hip: gemm_grouped_kernel): what differs between the groups of one grid
constexpr int UR_GEMM_MAX_GROUPS = 8;
struct GemmGroupSlot {
  const bf16_t* R; const bf16_t* S; void* C;
  long ldr, lds, ldc, slab_stride;
  int M, N, gm, gn;
  int wg0, pad_;        // first workgroup of the group's run
};
struct GemmGroups { int n, pad_; GemmGroupSlot g[UR_GEMM_MAX_GROUPS]; };

__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));     // opaque SGPR pair: keeps loop strength reduction from turning
                                             // (uniform base + lane offset) into per-lane 64-bit pointers
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// ---- shared by the persistent kernels (gemm_pers.hip, gemm_ws.hip) ----
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ char* uniform_wptr(char* p) { return const_cast<char*>(uniform_ptr(p)); }
// Epilogue accesses name the GLOBAL address space: a pointer rebuilt from scalar halves (uniform_ptr) or read out of the
// by-value argument struct is a generic pointer to hipcc, which then emits FLAT loads / stores -- and the wait-count pass answers
// any pending FLAT access with s_waitcnt vmcnt(0) lgkmcnt(0) instead of the exact in-order count (no load could stay in flight
// across a batch of stores).  global_load / global_store get exact counts.
template <int W> struct raw_words { typedef uint32_t type __attribute__((ext_vector_type(W))); };
template <> struct raw_words<1> { typedef uint32_t type; };
template <class T> __device__ __forceinline__ T ld_g(const void* p) {
  typedef typename raw_words<sizeof(T) / 4>::type raw_t;                   // (HIP's uint4 / float4 classes do not copy out of an address space)
  const raw_t r = *(const __attribute__((address_space(1))) raw_t*)(p);
  return __builtin_bit_cast(T, r);
}
template <class T> __device__ __forceinline__ T ld_g_nt(const void* p) {      // read-once streams (the SwiGLU backward's gate | up, a residual)
  typedef typename raw_words<sizeof(T) / 4>::type raw_t;
  const raw_t r = __builtin_nontemporal_load((const __attribute__((address_space(1))) raw_t*)(p));
  return __builtin_bit_cast(T, r);
}
template <class T> __device__ __forceinline__ void st_g(void* p, const T& v) {
  typedef typename raw_words<sizeof(T) / 4>::type raw_t;
  *(__attribute__((address_space(1))) raw_t*)(p) = __builtin_bit_cast(raw_t, v);
}

// XCD-aware tile order of gemm.hip, as a function of the (virtual) block id: ids equal mod 8 share an XCD
// n / d for n * d < 2^32 by one scalar multiply-high: magic = ceil(2^32 / d) (host, TileOrder)
struct TileOrder { int nwg, gn, gcw, rows_x, per; uint32_t m_gn, m_per, m_gcw; };
__device__ __forceinline__ int fdiv(int n, uint32_t magic) { return (int)__umulhi((uint32_t)n, magic); }
__device__ __forceinline__ void tile_coords(const TileOrder& o, int vid, int& bm, int& bn) {
  const int q = o.nwg >> 3, r = o.nwg & 7, x = vid & 7;
  const int id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (vid >> 3);
  if (o.gcw > 0) {
    const int j = id - x * q;                        // gm % 8 == 0: every XCD owns rows_x whole tile rows (r == 0)
    const int ch = fdiv(j, o.m_per), rem = j - ch * o.per;
    const int rr = fdiv(rem, o.m_gcw);
    bm = x * o.rows_x + rr;
    bn = ch * o.gcw + (rem - rr * o.gcw);
  } else {
    bm = fdiv(id, o.m_gn); bn = id - bm * o.gn;
  }
}

// two lanes 16 apart exchange halves: afterwards (a, b) of a lane in 16-lane row rho hold 2 x 4 CONSECUTIVE columns
//   a' = [a.row0, b.row0, a.row2, b.row2], b' = [a.row1, b.row1, a.row3, b.row3]
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
  const u32x2_t r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ void swap16f(float& a, float& b) {
  uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
  swap16(ua, ub);
  a = __uint_as_float(ua); b = __uint_as_float(ub);
}

// gemm_pers.hip: persistent 256x256 kernel for interior, K-contiguous, bf16-output launches (see its header comment).
// Returns 0 and launches when the shape qualifies, 1 when the caller should use the generic kernel, < 0 / > 0 on error.
bool gemm_pers_eligible(const GemmP& p, int splits, bool rk, bool sk, bool outf32);
int gemm_pers_launch(GemmP p, hipStream_t st);

// tools/lab/gemm_ws.hip (lab builds, -DUR_LAB=1): wave-specialised 128x256 kernel (the epilogue of a tile beside the next tile's K loop); a subset of the persistent kernel's launches
bool gemm_ws_eligible(const GemmP& p);
int gemm_ws_launch(GemmP p, hipStream_t st);

}  // namespace urgemm

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

__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));     // opaque SGPR pair: keeps loop strength reduction from turning
                                             // (uniform base + lane offset) into per-lane 64-bit pointers
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// gemm_pers.hip: persistent 256x256 kernel for interior, K-contiguous, bf16-output launches (see its header comment).
// Returns 0 and launches when the shape qualifies, 1 when the caller should use the generic kernel, < 0 / > 0 on error.
bool gemm_pers_eligible(const GemmP& p, int splits, bool rk, bool sk, bool outf32);
int gemm_pers_launch(GemmP p, hipStream_t st);

}  // namespace urgemm

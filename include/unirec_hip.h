/* unirec_hip.h -- C ABI of libunirec_hip.so: the MI355X (gfx950) implementation of UniRec's
 * nested Q-Former + Qwen3/LoRA hot path.
 *
 * The reference (ulab-uiuc/UniRec) is pure Python/PyTorch and has NO native interface for this
 * path: every entry point below is new and replaces the eager torch ops the
 * cited reference lines execute.  A reference-side maintainer binds it with ctypes exactly as
 * unirec_amd/_lib.py does.
 *
 * Conventions
 *   - plain pointers + sizes; no torch types.  All pointers are DEVICE pointers unless noted.
 *   - the caller (PyTorch caching allocator) owns every buffer, including workspaces, whose size is
 *     queried with the matching *_workspace_bytes call.  The library never allocates device memory.
 *   - tensors are row-major, batch-major; weights keep nn.Linear's [out, in] layout.
 *   - "bf16" = raw bfloat16 bits (torch.bfloat16 storage); statistics / losses / gradients of
 *     parameters are float32.
 *   - every call takes the HIP stream explicitly (pass torch.cuda.current_stream().cuda_stream);
 *     no call synchronises the device.  Re-entrant: no global mutable state besides an init-once
 *     kernel-attribute cache.
 *   - return value: 0 ok; < 0 invalid argument (shape / alignment / null); > 0 hipError_t.
 *     ur_last_error() returns the calling thread's last message.  Nothing throws or aborts.
 */
#ifndef UNIREC_HIP_H
#define UNIREC_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UR_ABI_VERSION 12

int ur_version(void);
const char* ur_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM: C[m][n] = alpha * ( sum_k R(m,k) S(n,k) + sum_k2 R2(m,k2) S2(n,k2) ) (+ epilogue)
 * Replaces nn.Linear forward / backward everywhere on the path:
 *   models/qformer.py:126-130,186-195 (query/key/value), :281-288 (attention output dense),
 *   :352-375 (FFN), models/qformer_utils.py:32-35,50-54 (heads); Qwen3 q/k/v/o/gate/up/down
 *   projections (transformers modeling_qwen3.py:81-83,227-238) incl. the LoRA A/B products
 *   (training/train_item_individual_token_joint.py:121-131).
 * r_kcontig: R stored [M][K] (1) or [K][M] (0).  s_kcontig: S stored [N][K] (1) or [K][N] (0).
 * The optional second range (R2,S2,K2) uses the same layout flags (LoRA low-rank term).
 * Epilogue (bf16 output): v = alpha*acc + bias[n] + residual[m][n]; v *= gelu'(aux[m][n]);
 *   C = bf16(v); gelu_out = bf16(gelu(bf16(v))).  f32 output: v = alpha*acc + bias[n].
 * split_k > 1: f32 output only, ldc == N, deterministic slab reduction in `workspace`.
 * Constraints: K,K2 % 8 == 0 unless both operands are K-strided; N, ldc % 4 == 0; leading dims
 * % 8 == 0; 16-byte aligned bases. */
typedef struct {
  const void* R; int64_t ldr; int32_t r_kcontig;
  const void* S; int64_t lds; int32_t s_kcontig;
  int32_t K;
  const void* R2; int64_t ldr2; const void* S2; int64_t lds2; int32_t K2;
  void* C; int64_t ldc; int32_t c_f32;
  int32_t M, N;
  float alpha;
  const float* bias;
  const void* residual; int64_t ldres;
  void* gelu_out; int64_t ldg;
  const void* gelu_grad_aux; int64_t ldaux;
  int32_t split_k;
  /* LoRA dropout in the backward to the adapter input -- peft LoraLayer: result = base(x) + lora_B(lora_A(dropout(x)))
   * * scaling, one nn.Dropout per adapter (training/train_item_individual_token_joint.py:121-131: lora_dropout=0.1).
   * drop_bits != NULL: the second pair holds K2/drop_rank adapters that share the input x [M, N]; instead of joining
   * the main reduction,  C(m,n) += sum_a keep_a(m,n)/(1-drop_p) * R2[m, a*r:(a+1)*r] . S2[n, a*r:(a+1)*r]
   * (dx = dy W + sum_a mask_a * (tb_a A_a)); keep_a comes from bit plane a of ur_lora_dropout_bits (row stride
   * drop_bits_ld = ur_lora_bits_ld(N) bytes, plane stride drop_bits_stride bytes). */
  const void* drop_bits; int64_t drop_bits_ld; int64_t drop_bits_stride;
  int32_t drop_rank; float drop_p;
  /* SwiGLU backward as the epilogue of the down-projection's dX GEMM (Qwen3MLP: down(act_fn(gate(x)) * up(x)),
   * transformers modeling_qwen3.py:81-91).  swiglu_gu != NULL: the GEMM result v[m][n] (n < N = swiglu_I, f32, never
   * rounded or stored) is d(act); with g = gu[m][n], u = gu[m][swiglu_I + n] the epilogue writes
   *   dgu[m][n] = v * u * silu'(g)      dgu[m][swiglu_I + n] = v * silu(g)
   * and C is not written (pass any valid pointer).  bf16 output, no split_k, no residual / gelu modes. */
  const void* swiglu_gu; int64_t swiglu_ldgu;
  void* swiglu_dgu; int64_t swiglu_lddgu;
  int32_t swiglu_I;
  /* SwiGLU forward as the epilogue of the up-projection GEMM (the gate projection has run): swiglu_gate != NULL:
   * C[m][n] = bf16(v) is up(x) as usual, and additionally  swiglu_act[m][n] = silu(gate[m][n]) * C[m][n]  (from the
   * ROUNDED C, the value the backward reads).  bf16 output, no split_k, no residual / gelu modes. */
  const void* swiglu_gate; int64_t swiglu_ldgate;
  void* swiglu_act; int64_t swiglu_ldact;
  /* q/k-norm + RoPE as the epilogue of the merged q|k|v projection (Qwen3Attention: q_norm / k_norm over each head, then
   * apply_rotary_pos_emb -- transformers modeling_qwen3.py:59-64,107-137,227-245).  qkr_q != NULL: the N output columns are
   * qkr_nq_cols of q heads, then qkr_nk_cols of k heads, then the v columns (head_dim 128).  The ROWS of S (and S2) of every
   * q / k head must be stored in the paired order ur_qkrope_perm (tile column c of a head holds feature
   * ((c >> 4) & 1) * 64 + (c >> 5) * 16 + (c & 15): a lane of the MFMA tile then owns the rotate-half partners d, d + 64).
   * C is not written.  qkr_q [M, qkr_nq_cols] and qkr_k [M, qkr_nk_cols] receive RoPE(RMSNorm(x) * weight) of the f32
   * accumulators in the STANDARD feature order, qkr_v [M, N - nq_cols - nk_cols] the plain projection, qkr_rstd
   * [M, (nq_cols + nk_cols) / 128] f32 the 1 / rms of every (token, head) -- what ur_qknorm_rope_bwd_roped needs instead of
   * the raw q, k, which are never stored.  Position of row m = m % qkr_S; cos / sin tables [S, 64] f32 (ur_rope_table).
   * Runs on the persistent kernel only (M, N multiples of 256, K of 64, >= 128 tiles, S a multiple of 256, nq / nk columns multiples
   * of 256, no other epilogue): ur_gemm_qkrope_supported tells; ur_gemm fails loudly otherwise. */
  void* qkr_q; int64_t qkr_ldq; void* qkr_k; int64_t qkr_ldk; void* qkr_v; int64_t qkr_ldv;
  float* qkr_rstd;
  const float* qkr_qw; const float* qkr_kw; const float* qkr_cos; const float* qkr_sin;
  int32_t qkr_S, qkr_nq_cols, qkr_nk_cols; float qkr_eps;
  /* SwiGLU forward as the epilogue of the MERGED gate|up projection (Qwen3MLP: act_fn(gate_proj(x)) * up_proj(x), transformers
   * modeling_qwen3.py:81-91).  swp_act != NULL: N == 2 * swp_I, and the ROWS of S (and S2) are interleaved in blocks of 128:
   * tile column block t holds [gate rows 128 t .. 128 t + 127 | up rows 128 t .. 128 t + 127], so one lane of the 256-column
   * tile owns gate and up of the same feature.  C [M, 2 I] receives gate | up in the STANDARD order (what the backward reads)
   * and swp_act [M, swp_I] = silu(gate) * up computed from the bf16-rounded C values (bit-identical to ur_swiglu_fwd).
   * Persistent kernel only: ur_gemm_swiglu_paired_supported tells; ur_gemm fails loudly otherwise. */
  void* swp_act; int64_t swp_ldact; int32_t swp_I;
} ur_gemm_args;
int ur_gemm_swiglu_paired_supported(const ur_gemm_args* a);
/* 1 when ur_gemm would run the q/k-norm + RoPE epilogue for these arguments, 0 when the caller must use the separate
 * ur_qknorm_rope_fwd pass. */
int ur_gemm_qkrope_supported(const ur_gemm_args* a);
/* feature index stored at tile column c (0..127) of a head under the paired order */
int ur_qkrope_perm(int c);
int64_t ur_gemm_workspace_bytes(const ur_gemm_args* a);
int ur_gemm(const ur_gemm_args* a, void* workspace, int64_t workspace_bytes, void* stream);
/* `count` (1..8) independent products of ONE kind in one grid: K-strided ("token-major") bf16 operands, f32 output, the same K,
 * split_k and alpha, no epilogue -- the weight gradients dW = dY^T X of the Linear layers of one Q-Former layer (autograd of
 * nn.Linear under /root/reference/models/qformer.py:56-92 attention projections, :238-275 feed-forward; the reference issues one
 * addmm per weight).  Alone none of those [out, in] = [768..3072, 768..3072] products over 8192 tokens fills the chip without
 * slicing the token axis into pieces too short for a tile; together they are one round of 256 x 256 tiles.  Results per product
 * equal ur_gemm's for the same tile size and split (same order of summation).  Workspace: ur_gemm_grouped_workspace_bytes. */
int64_t ur_gemm_grouped_workspace_bytes(const ur_gemm_args* a, int32_t count);
int ur_gemm_grouped(const ur_gemm_args* a, int32_t count, void* workspace, int64_t workspace_bytes, void* stream);
/* Launches with K-contiguous operands, bf16 output, M, N multiples of 256, K a multiple of 64 (>= 256), >= 128 output tiles and
 * a plain / bias / residual / masked-LoRA / SwiGLU-backward epilogue run on the persistent kernel (csrc/gemm_pers.hip: one
 * workgroup per CU walks its tiles, the LDS-DMA ring never drains, epilogue from registers) -- bit-identical to the generic
 * kernel.  ur_gemm_persistent_mode(0) keeps every launch on the generic kernel, (1) enables the persistent one (2 behaves as 1 in
 * the product library; round 5's wave-specialised lab kernel left the product build in round 6: tools/lab/gemm_ws.hip), (-1) returns
 * to the default (1); returns the previous setting.  Process-wide; for A/B timing and the bit-identity tests.  The library reads
 * no environment variable. */
int ur_gemm_persistent_mode(int mode);

/* ------------------------------------------------------------------------------------------------
 * LoRA adapter products, rank 16 (peft LoraLayer; call site training/train_item_individual_token_joint.py:121-131,
 * r=16, lora_alpha=32, lora_dropout=0.1).  HBM-bound streams over one [M, W] activation X:
 *   ur_lora_project:  P[m, 16a + j] = alpha * sum_w keep_a(m,w) X[m, col0_a + w] U_a[j, w]
 *       forward  t  = s * dropout_a(x) A_a^T   (shared = 1: the nad adapters read the same columns, one bit plane each)
 *       backward tb = s * dy_a B_a             (shared = 0: adapter a owns columns [col0[a], col0[a]+width[a]) of dy;
 *                                               U_a = B_a^T stored [16, width[a]])
 *   ur_lora_reduce:   G_a[j, w] = alpha * sum_m V[m, 16a + j] keep_a(m,w) X[m, col0_a + w]
 *       dA_a = tb_a^T dropout_a(x)  (shared = 1, g_transposed = 0: G = [16 nad, W] f32, dense)
 *       dB_a = dy_a^T t_a           (shared = 0, g_transposed = 1: G = [sum width, 16] f32, dense, adapter ranges in order)
 *       Token reduction split deterministically over blocks; partial slabs live in the caller's workspace.
 *   ur_lora_dropout_bits: dropped flags of nad adapters over an [M, W] input: plane a at bits + a*bits_stride, row m at
 *       + m*bits_ld (bits_ld = ur_lora_bits_ld(W) = 16 * ceil(W/128) bytes); the byte at column c/8 (c % 8 == 0) holds
 *       bit i (i<4) = element c+2i dropped, bit 4+i = element c+2i+1 dropped.  A pure function of (seed, p, row0 + m, c, a).
 * drop_bits == NULL: no dropout.  The 1/(1-p) scale is the caller's (alpha).
 * Constraints: rank == 16; column ranges and ldx multiples of 8; X, U, V, G 16-byte aligned; P 8-byte aligned. */
typedef struct {
  const void* X; int64_t ldx; int32_t M;
  int32_t nad; int32_t rank; int32_t shared;
  int32_t col0[4]; int32_t width[4];
  const void* drop_bits; int64_t bits_ld; int64_t bits_stride;
  float alpha;
  const void* U[4]; int64_t ldu[4];       /* project: U_a bf16 [16, width_a] */
  void* P; int64_t ldp;                   /* project: bf16 [M, >= 16 nad] */
  const void* V; int64_t ldv;             /* reduce: bf16 [M, >= 16 nad] */
  void* G; int32_t g_transposed;          /* reduce: f32, dense */
  /* reduce only, optional: the TOKEN-packed copy of the dropped flags (ur_lora_bits_transpose).  With it (and M, the token split and
   * every width multiples of 128 / 64) ur_lora_reduce streams X through an LDS-DMA ring and masks the transposed fragments in
   * registers; without it the register-staged kernel runs.  Same flags, same result up to the order of the f32 token sum. */
  const void* drop_bits_t; int64_t bits_t_ld; int64_t bits_t_stride;
} ur_lora_args;
int64_t ur_lora_bits_ld(int32_t W);
/* Token-packed flags: plane a at bits_t + a*bits_t_stride (32-bit words), token group tg = m / 32 at + tg*bits_t_ld, column c at + c:
 * byte g of the word = tokens 32 tg + 8 g .. + 7 of column c, bit i (i<4) = token 8g+2i dropped, bit 4+i = token 8g+2i+1 dropped (the
 * pair order of a transposed bf16x8 fragment).  bits_t_ld = ur_lora_bits_t_ld(W) = W rounded up to 4 words; M % 32 == 0. */
int64_t ur_lora_bits_t_ld(int32_t W);
int ur_lora_bits_transpose(const uint8_t* bits, int64_t bits_ld, int64_t bits_stride, int32_t M, int32_t W, int32_t nad,
                           uint32_t* bits_t, int64_t bits_t_ld, int64_t bits_t_stride, void* stream);
/* row0: rows that precede row 0 of this call in the GLOBAL minibatch (row m draws the flags of global row row0 + m) */
int ur_lora_dropout_bits(uint64_t seed, float p, int32_t M, int32_t W, int32_t nad, uint8_t* bits, int64_t bits_ld,
                         int64_t bits_stride, int64_t row0, void* stream);
int ur_lora_project(const ur_lora_args* a, void* stream);
/* RMSNorm forward (ur_rmsnorm_fwd: out = w * (x * rstd), Qwen3RMSNorm modeling_qwen3.py:59-64) fused with the down
   projection of the 2 or 3 adapters that read the normalised activation (q|k|v or gate|up): `a` as for ur_lora_project with
   shared = 1 and X ignored (the adapters read `out`): P[m, 16a + j] = alpha * sum_c keep_a(m,c) out[m,c] U_a[j,c].  `out` is
   written once and not re-read.  D == 1024 (the Qwen3-0.6B hidden size). */
/* SwiGLU forward (ur_swiglu_fwd: act = silu(gate) * up over gu = [gate | up], Qwen3MLP modeling_qwen3.py:81-83) fused with
   the down projection of the down_proj adapter, which reads act: P[m, j] = alpha * sum_c keep(m,c) act[m,c] U[j,c]
   (`a`: nad = 1, U[0] = A [16, I], optional bit plane, X ignored).  I a multiple of 128. */
int ur_swiglu_lora_fwd(const void* gu, void* act, int32_t M, int32_t I, const ur_lora_args* a, void* stream);
int ur_rmsnorm_lora_fwd(const void* x, const float* w, void* out, float* rstd, int32_t M, int32_t D, float eps,
                        const ur_lora_args* a, void* stream);
int64_t ur_lora_reduce_workspace_bytes(const ur_lora_args* a);
int ur_lora_reduce(const ur_lora_args* a, void* workspace, int64_t workspace_bytes, void* stream);
/* ur_lora_bgrad: the B side of the backward in ONE pass over dy (shared = 0, no dropout planes): P = tb = alpha * dy_a B_a
 * (as ur_lora_project with U_a = B_a^T) AND G = dB, [sum width, 16] f32 dense (as ur_lora_reduce with V = t,
 * g_transposed = 1, scale 1); partial slabs of ceil(M / 512) token blocks live in the caller's workspace. */
int64_t ur_lora_bgrad_workspace_bytes(const ur_lora_args* a);
int ur_lora_bgrad(const ur_lora_args* a, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (+ fused dropout / residual) -- models/qformer.py:64,106-107 (embeddings: LN then
 * dropout) and :285-289, :371-375 (dense -> dropout -> LayerNorm(h + input)).
 * z = dropout_pre(y) + residual;  out = dropout_post(LN(z) * gamma + beta)
 *   y [M,H] bf16 (row m reads y[(m % y_rows)], so a [Q,H] query table broadcasts over the batch);
 *   residual [M,H] bf16 or NULL; z_save [M,H] bf16 or NULL (kept for backward);
 *   mean, rstd [M] f32.  H % 8 == 0, H <= 8192.  Dropout masks are a pure function of
 *   (seed, element index) and are regenerated by the backward.  drop_row0: index of row 0 of this launch in the GLOBAL
 *   minibatch (element index = (drop_row0 + row) * H + h): a data-parallel rank passes the rows that precede its shard, so
 *   the masks -- and the training run -- do not depend on the number of ranks; 0 for a single process. */
int ur_layernorm_fwd(const void* y, int32_t y_rows, const void* residual, const float* gamma, const float* beta,
                     void* out, void* z_save, float* mean, float* rstd, int32_t M, int32_t H, float eps,
                     float p_pre, uint64_t seed_pre, float p_post, uint64_t seed_post, int64_t drop_row0, void* stream);
/* Backward.  dz [M,H] bf16 = gradient w.r.t. z (the residual branch); dy [M,H] bf16 = dz with the
 * pre-dropout mask applied (may alias dz when p_pre == 0; may be NULL when not needed).
 * dgamma, dbeta [H] f32 are OVERWRITTEN; dbias [H] f32 (column sum of dy, i.e. the gradient of the
 * preceding dense bias) is written when non-NULL.  workspace: ur_layernorm_bwd_workspace_bytes(H).
 * dgamma == NULL (round 6): the call stops at the per-block partial sums in `workspace` and the caller finishes them with
 * ur_layernorm_bwd_reduce(workspace, M, H, ...) -- on ANY stream ordered behind this call: the three parameter gradients hang off the
 * backward's dX chain (unirec_amd/qformer.py runs the reduction on its side stream beside the next layer's products).  The workspace then
 * belongs to that pending reduction until it has run. */
int64_t ur_layernorm_bwd_workspace_bytes(int32_t H);
int ur_layernorm_bwd(const void* dout, const void* z, const float* mean, const float* rstd, const float* gamma,
                     void* dz, void* dy, float* dgamma, float* dbeta, float* dbias, int32_t M, int32_t H,
                     float p_pre, uint64_t seed_pre, float p_post, uint64_t seed_post, int64_t drop_row0,
                     void* workspace, int64_t workspace_bytes, void* stream);
int ur_layernorm_bwd_reduce(const void* workspace, int32_t M, int32_t H, float* dgamma, float* dbeta, float* dbias, void* stream);

/* out[r][h] (f32, overwritten) = sum_{b<nb} in[(b*rows + r)][h]   (bf16 in).  Used for the
 * gradient of the batch-broadcast query_embeddings (models/qformer_utils.py:39) and, with rows=1,
 * for bias gradients (column sums). */
int ur_batch_reduce(const void* in, float* out, int32_t nb, int32_t rows, int32_t H, void* workspace,
                    int64_t workspace_bytes, void* stream);
int64_t ur_batch_reduce_workspace_bytes(int32_t nb, int32_t rows, int32_t H);

/* ------------------------------------------------------------------------------------------------
 * RMSNorm -- transformers modeling_qwen3.py:59-64.  out = w * (x * rsqrt(mean(x^2)+eps)).
 * x,out [M,D] bf16; w [D] f32; rstd [M] f32 saved.  Backward (weights frozen under LoRA):
 * dx = add (bf16 [M,D] or NULL) + rstd*(g - xhat*mean(g*xhat)), g = dout*w. */
int ur_rmsnorm_fwd(const void* x, const float* w, void* out, float* rstd, int32_t M, int32_t D, float eps, void* stream);
int ur_rmsnorm_bwd(const void* dout, const void* x, const float* w, const float* rstd, const void* add, void* dx,
                   int32_t M, int32_t D, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused attention (MFMA, online softmax, K/V tiles staged through LDS).
 *   causal == 0 : Q-Former attention -- models/qformer.py:169-275 (BertSelfAttention.forward):
 *       scores/sqrt(d) + mask, softmax, dropout(probs), probs @ V.  key_mask [B,Sk] (1 = attend)
 *       reproduces invert_attention_mask's ADDITIVE finfo(float32).min: a fully masked row is a
 *       uniform softmax over all Sk keys, never NaN.
 *   causal == 1 : Qwen3 attention -- transformers modeling_qwen3.py:185-208,244-280 with the SDPA
 *       mask semantics (causal AND key padding; a query row with no allowed key outputs 0).
 *       GQA: kv head = q head / (nq / nkv).  Needs Sq == Sk, dropout_p == 0.
 * q [B,Sq,nq,hd], k/v [B,Sk,nkv,hd], o [B,Sq,nq,hd] bf16 with token strides ldq/ldk/ldv/ldo
 * (elements; lets q,k,v live in one fused projection buffer).  head_dim 64 or 128.
 * stats [B,nq,Sq,2] f32 = (row max, 1/row sum) saved for the backward. */
typedef struct {
  const void* q; const void* k; const void* v; void* o; float* stats;
  int64_t ldq, ldk, ldv, ldo;
  const uint8_t* key_mask;
  int32_t B, Sq, Sk, nq, nkv, head_dim;
  int32_t causal;
  float scale;
  float dropout_p; uint64_t seed;
  int64_t drop_batch0;   /* batch rows that precede b = 0 of this call in the GLOBAL minibatch: the dropout ROW of (b, h, q) is
                            ((drop_batch0 + b) * nq + h) * Sq + q (ur_attn_dropout_keep), so a data-parallel rank draws the masks of ITS samples */
} ur_attn_args;
/* Backward: dout [B,Sq,nq,hd] -> dq [B,Sq,nq,hd], dk/dv [B,Sk,nkv,hd] (bf16, strides in elements).
 * delta: caller-provided f32 scratch of ur_attn_bwd_workspace_floats(B, nq, Sq) = 4*B*nq*Sq + 16 words: the row constants of
 * the backward (-rowsum(dO*O) and -LSE/scale), two planes of per-row dropout keys (written by the dQ kernel, read by the dK/dV
 * kernel; untouched without dropout) and the work-queue words of the persistent dK/dV kernel (zeroed by the
 * call itself) -- the library keeps no mutable device state of its own, so calls on different streams never interfere as long
 * as each brings its own workspace.  `a` must be the forward's arguments (o and stats filled by ur_attn_fwd).
 * No floating-point atomics: results are bitwise reproducible. */
typedef struct {
  const void* dout; void* dq; void* dk; void* dv;
  int64_t lddo, lddq, lddk, lddv;
  float* delta;
  /* q-norm + RoPE backward fused into the dQ kernel (Qwen3Attention: q = rope(q_norm(q_proj(x))), modeling_qwen3.py:59-64,
   * 107-170,244-252; causal, head_dim 128).  rope_q_raw != NULL: q in ur_attn_args was produced from the raw projection
   * rope_q_raw [B*Sq, >= nq*hd] (row stride rope_ldraw) with norm weight rope_q_weight [hd] (f32), eps rope_eps and the
   * cos / sin tables [Sq, hd/2] (f32, position = row index inside the sequence); the kernel then writes the gradient of
   * the RAW projection to rope_dq_raw (row stride rope_lddraw) and dq is not written (may be NULL). */
  const void* rope_q_raw; int64_t rope_ldraw;
  const float* rope_q_weight; const float* rope_cos; const float* rope_sin; float rope_eps;
  void* rope_dq_raw; int64_t rope_lddraw;
  /* rope_rstd != NULL (with rope_q_raw): the forward ran q-norm + RoPE as the q|k|v GEMM's epilogue (ur_gemm_args.qkr_*) and the raw
   * projection was never stored.  rope_q_raw then holds the ROPED, normed q (that epilogue's qkr_q: the q of ur_attn_args), the
   * normalised row is recovered by rotating back (x^ = R^T(q) / weight; non-zero norm weights) and 1 / rms of (token m, head h) is
   * rope_rstd[m * rope_rstd_ld + rope_rstd_h0 + h] (qkr_rstd); rope_eps is not used.  Same arithmetic as
   * ur_qknorm_rope_bwd_roped on the q heads. */
  const float* rope_rstd; int64_t rope_rstd_ld; int32_t rope_rstd_h0;
  /* with rope_rstd, optional: the k heads as well.  rope_k = the ROPED, normed k (qkr_k: the k of ur_attn_args, row stride rope_ldk),
   * rope_k_weight its norm weight, 1 / rms of (token m, k head h) = rope_rstd[m * rope_rstd_ld + rope_rstd_hk0 + h]; rope_dk_raw
   * (row stride rope_lddkraw) receives the gradient of the RAW k projection.  The generated causal head_dim-128 dK/dV kernel applies it
   * in its store (dk is then not written); other shapes write dk (dense [B*Sk, nkv*hd] required) and run ur_qknorm_rope_bwd_roped_k. */
  const void* rope_k; int64_t rope_ldk; const float* rope_k_weight; int32_t rope_rstd_hk0;
  void* rope_dk_raw; int64_t rope_lddkraw;
  /* kv_colsum != NULL: the call also writes the column sums over ALL keys of all batch rows of dK and dV, f32 [2][nq * head_dim]
   * ([dK sums | dV sums], feature h * head_dim + d) -- the bias gradients of the K | V projections that feed this attention
   * (models/qformer.py:186-188: key / value are nn.Linear WITH bias), which otherwise cost the caller a second pass over dK | dV (13 GB
   * per step for the user Q-Former at C3).  Produced by the few-query dK/dV kernel only (<= 64 queries, >= 256 keys, head_dim 64, one
   * kv head per query head, non-causal): ur_attn_bwd_kv_colsum_floats(a) is the size in f32 words of the scratch kv_colsum_ws the
   * caller must then provide, or 0 when the shape does not take that kernel (pass kv_colsum = NULL and sum dK | dV yourself). */
  float* kv_colsum; float* kv_colsum_ws;
} ur_attn_bwd_args;
int ur_attn_fwd(const ur_attn_args* a, void* stream);
int ur_attn_bwd(const ur_attn_args* a, const ur_attn_bwd_args* g, void* stream);
/* Kernel selection of ur_attn_fwd / ur_attn_bwd (round 6: replaces the UR_ATTN_* environment variables the library used to read -- a
 * product library's launch sequence must not depend on its caller's environment).  Process-wide words, defaults = the product path;
 * every alternative is a complete path kept for the bit-identity / oracle tests (tests/test_gpu_switches.py) and for A/B timing:
 *   UR_ATTN_MODE_TINY        3 (default): <= 4-query x <= 16-key shapes on the VALU kernels, forward (bit 0) and backward (bit 1); 0: MFMA kernels
 *   UR_ATTN_MODE_C128        1: causal head_dim-128 launches on the generated loops (tools/asmgen); 0: the compiler-scheduled kernels
 *   UR_ATTN_MODE_DKV_PERSIST 1: one dK/dV workgroup per CU drawing key blocks from the call's work queue; 0: one workgroup per key block
 *   UR_ATTN_MODE_FEWQ        1: few-query x many-key shapes on attn_bwd_dkv_fewq_kernel; 0: attn_bwd_dkv_kernel
 * ur_attn_mode(key, value) sets the word and returns the previous value (value -1: back to the default; -2: query only); -1 for an
 * unknown key.  Replaces nothing of the reference (SDPA picks its kernels itself, modeling_qwen3.py:185-208). */
enum { UR_ATTN_MODE_TINY = 0, UR_ATTN_MODE_C128 = 1, UR_ATTN_MODE_DKV_PERSIST = 2, UR_ATTN_MODE_FEWQ = 3, UR_ATTN_MODE_COUNT = 4 };
int ur_attn_mode(int key, int value);
/* f32 words the `delta` workspace of ur_attn_bwd must hold (row constants + the call's own work-queue words) */
int64_t ur_attn_bwd_workspace_floats(int32_t B, int32_t nq, int32_t Sq);
int64_t ur_attn_bwd_kv_colsum_floats(const ur_attn_args* a);
/* Keep flags of the counter-based dropout every kernel here regenerates instead of storing (nn.Dropout at models/qformer.py:107,
 * 258, 287, 373).  Test / inspection entries: a parity test feeds these masks to the reference's own nn.Dropout modules and to the
 * CPU oracle, so that a TRAINING-mode step can be compared value for value.
 *   ur_dropout_keep: hidden dropout of ur_layernorm_fwd/bwd: keep[i] = 1 iff element idx0 + i of the stream `seed` survives
 *     probability p; the counter of (row, column) is (drop_row0 + row) * H + column.
 *   ur_attn_dropout_keep: attention probabilities of ur_attn_fwd/bwd: keep[r * Sk + key] for the rows row0 .. row0 + nrows - 1, a
 *     row being R = ((drop_batch0 + b) * nq + h) * Sq + query.  One 32-bit word decides a PAIR of keys (2 kp, 2 kp + 1) of a row
 *     (two 16-bit fields against p * 65536: unirec_amd/csrc/common.hip.h ur_attn_pair_word; oracle/dropout_ref.py restates it). */
int ur_dropout_keep(uint64_t seed, float p, uint64_t idx0, int64_t n, uint8_t* keep, void* stream);
int ur_attn_dropout_keep(uint64_t seed, float p, uint64_t row0, int64_t nrows, int32_t Sk, uint8_t* keep, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Qwen3 per-head q/k RMSNorm + rotary embedding -- modeling_qwen3.py:59-64 (norm over head_dim),
 * :140-170 (rotate_half / apply_rotary_pos_emb), :107-137 (inv_freq, positions arange(S)).
 * cos/sin tables are [S][head_dim/2] f32 (ur_rope_table).  qkv_raw [M, (nq+2nkv)*hd] bf16 is the
 * fused projection output (q heads | k heads | v heads); token m has position m % S.
 * fwd: q_out [M,nq*hd], k_out [M,nkv*hd] bf16.  bwd: dqkv_raw q and k sections are written (the v
 * section is written by ur_attn_bwd). */
int ur_rope_table(float* cos_out, float* sin_out, int32_t S, int32_t head_dim, float theta, void* stream);
int ur_qknorm_rope_fwd(const void* qkv_raw, int64_t ldraw, const float* q_norm_w, const float* k_norm_w,
                       const float* cos_tab, const float* sin_tab, void* q_out, void* k_out, int64_t M, int32_t S,
                       int32_t nq, int32_t nkv, int32_t head_dim, float eps, void* stream);
int ur_qknorm_rope_bwd(const void* dq_out, const void* dk_out, const void* qkv_raw, int64_t ldraw,
                       const float* q_norm_w, const float* k_norm_w, const float* cos_tab, const float* sin_tab,
                       void* dqkv_raw, int64_t lddraw, int64_t M, int32_t S, int32_t nq, int32_t nkv,
                       int32_t head_dim, float eps, void* stream);
/* The same backward when the forward ran as the q|k|v GEMM's epilogue (ur_gemm_args.qkr_*) and the raw projections were never
 * stored: q_roped [M, >= nq*hd] / k_roped [M, >= nkv*hd] are the forward's outputs, rstd [M, nq + nkv] its row constants; the
 * normalised rows are recovered by rotating back (x^ = R^T(o) / weight: the norm weights must be non-zero).  head_dim 128. */
int ur_qknorm_rope_bwd_roped(const void* dq_out, const void* dk_out, const void* q_roped, int64_t ldq, const void* k_roped, int64_t ldk,
                             const float* rstd, const float* q_norm_w, const float* k_norm_w, const float* cos_tab, const float* sin_tab,
                             void* dqkv_raw, int64_t lddraw, int64_t M, int32_t S, int32_t nq, int32_t nkv, int32_t head_dim, void* stream);
/* The k heads alone (nq = 0 above would lose the row-constant layout): the q heads' backward rode in the dQ kernel
 * (ur_attn_bwd_args.rope_rstd).  dk_out [M, nkv*hd] dense, k_roped [M, >= nkv*hd], 1 / rms of (m, k head h) = rstd[m * rstd_ld + rstd_h0 + h],
 * dk_raw [M, >= nkv*hd] (row stride lddraw) receives the gradient of the raw k projection. */
int ur_qknorm_rope_bwd_roped_k(const void* dk_out, const void* k_roped, int64_t ldk, const float* rstd, int64_t rstd_ld, int32_t rstd_h0,
                               const float* k_norm_w, const float* cos_tab, const float* sin_tab, void* dk_raw, int64_t lddraw,
                               int64_t M, int32_t S, int32_t nkv, int32_t head_dim, void* stream);

/* Embedding gather fused with Q-Former token injection --
 * training/train_item_individual_token_joint.py:143 (embed) and :160-171 (triple python loop with a
 * host sync per (item, query, sample)): out[b][s] = item_tokens[b][id - first] if first <= id <
 * first+T else embed[id].  input_ids int64 [B,S]; embed [vocab,D] bf16; item_tokens [B,T,D] bf16
 * (T = hist*Q_item; may be NULL/0).  Zero host syncs.
 * Backward: d_item_tokens[b][t] = sum of dx rows at the positions holding token t (0 when the token
 * was truncated away); the overwritten embedding rows receive no gradient, as in the reference. */
int ur_embed_inject_fwd(const void* embed, int64_t vocab, const int64_t* input_ids, const void* item_tokens,
                        int64_t first_special_id, int32_t T, void* out, int32_t B, int32_t S, int32_t D, void* stream);
int ur_inject_bwd(const void* dx, const int64_t* input_ids, int64_t first_special_id, int32_t T, void* d_item_tokens,
                  int32_t B, int32_t S, int32_t D, void* stream);

/* U0: user-sequence assembly -- models/user_sequence_encoder.py:128-142 (tokens + (time+geo) context
 * broadcast over the item's query tokens, flatten, + sinusoidal PE over the flat index :16-33, dropout)
 * fused with the collate's right-padding + mask (training/user_qformer_training.py:153-161).
 * item_tokens [B,L,Qi,H], context [B,L,H] bf16; lengths int32 [B] (events per user);
 * out [B,L*Qi,H] bf16 (zeros past the user's length), mask [B,L*Qi] f32 (1 = real token). */
int ur_user_sequence_assemble(const void* item_tokens, const void* context, const int32_t* lengths, void* out, float* mask,
                              int32_t B, int32_t L, int32_t Qi, int32_t H, float dropout_p, uint64_t seed, int64_t drop_batch0,
                              void* stream);      /* drop_batch0: users that precede b = 0 in the global minibatch (dropout counter) */

/* Mean over the middle axis, x [B,S,D] bf16 -> [B,D] (f32 and/or bf16 output) --
 * train_item_individual_token_joint.py:179-181 (mean over ALL S positions, padding included),
 * models/qformer_utils.py:50 and training/user_qformer_training.py:60 (mean over query tokens). */
int64_t ur_mean_pool_workspace_bytes(int32_t B, int32_t D);
int ur_mean_pool_fwd(const void* x, float* out_f32, void* out_bf16, int32_t B, int32_t S, int32_t D, void* workspace,
                     int64_t workspace_bytes, void* stream);
int ur_mean_pool_bwd(const float* dout_f32, const void* dout_bf16, void* dx, int32_t B, int32_t S, int32_t D, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data path either side of the hot path.
 * ur_gather_rows: out[i] = src[idx[i]] for rows of row_elems elements; idx < 0 or >= n_src gives a zero row -- the
 *   packed form of training/train_item_individual_token_joint.py:557-577 (_get_history_qformer_inputs: per history slot
 *   the cached [F,1024] field vectors and [F] mask, zeros for padding / unknown items) and :246-255 (cached item query
 *   tokens).  Kinds: UR_KIND_U8 / UR_KIND_BF16 / UR_KIND_F32; f32 rows may be written as bf16.
 * ur_catalog_scores: scores[b][n] = cos(user_b, item_n) against a shared catalogue [N,D] f32 (F.normalize eps 1e-12,
 *   :408-415 with pool = all items); writes user_inv_norm [B], and cat_inv_norm [N] unless cat_norm_ready.
 * ur_rank_of_index: rank_b = 1 + #{n : s_bn > s_b,gt_b} (:416-417; the positive wins ties as in ur_mrr_rank). */
#define UR_KIND_U8 0
#define UR_KIND_BF16 1
#define UR_KIND_F32 2
int ur_gather_rows(const void* src, int32_t src_kind, void* out, int32_t out_kind, const int64_t* idx, int64_t row_elems,
                   int64_t n_out, int64_t n_src, void* stream);
int ur_catalog_scores(const float* user, const float* catalog, float* scores, float* user_inv_norm, float* cat_inv_norm,
                      int32_t cat_norm_ready, int32_t B, int64_t N, int32_t D, void* stream);
int ur_rank_of_index(const float* scores, const int64_t* gt_index, int32_t* rank, int32_t B, int64_t N, void* stream);
/* ur_context_mlp1: first layer of the event-context MLPs of models/user_sequence_encoder.py:118-121:
 *   kind 0 = TimestampEncoder (models/mwne.py:525-565; in = timestamps.float() [n]; 9 features),
 *   kind 1 = GeoCoordinateEncoder (models/mwne.py:586-607; in = (lat, lon) degrees [n,2]; 3 features);
 *   out[n][j] = bf16(gelu(b1[j] + sum_f W1[j][f] feat_f)), W1 f32 [H2, 9 or 3].  The second Linear is ur_gemm. */
int ur_context_mlp1(const float* in, int32_t kind, const float* W1, const float* b1, void* out, int64_t n, int32_t H2, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Ranking head -- train_item_individual_token_joint.py:331-352 (InfoNCELoss), :392-419 (MRR).
 * ur_cosine_scores: scores[b][0] = cos(user_b, pos_b), scores[b][1+n] = cos(user_b, neg_{b,n})
 *   (F.normalize eps 1e-12), all f32; user [B,D], pos [B,D], neg [B,N,D]; cand_inv_norm [B,N+1].
 * ur_infonce_fwd_bwd: loss = mean_b( -s_b0/tau + logsumexp over {0} U valid negatives of s/tau );
 *   neg_mask uint8 [B,N] (1 = valid) or NULL; d_user [B,D] = grad_scale * dloss/duser (may be NULL).
 * ur_mrr_rank: rank_b = 1 + #{valid n : s_{b,1+n} > s_b0}  (positive wins ties; the reference's
 *   unstable argsort leaves ties undefined).
 * ur_topk: descending top-K over scores [B,C], lowest index first among equal scores. */
int ur_cosine_scores(const float* user, const float* pos, const float* neg, float* scores, float* cand_inv_norm,
                     int32_t B, int32_t N, int32_t D, void* stream);
int64_t ur_infonce_workspace_bytes(int32_t B, int32_t N, int32_t D);
int ur_infonce_fwd_bwd(const float* user, const float* pos, const float* neg, const uint8_t* neg_mask,
                       const float* scores, const float* cand_inv_norm, float temperature, float grad_scale,
                       float* loss, float* d_user, int32_t B, int32_t N, int32_t D, void* workspace,
                       int64_t workspace_bytes, void* stream);
int ur_mrr_rank(const float* scores, const uint8_t* neg_mask, int32_t* rank, int32_t B, int32_t N, void* stream);
int ur_topk(const float* scores, int32_t B, int32_t C, int32_t K, int32_t* idx_out, float* val_out, void* workspace,
            int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Item / user Q-Former heads and losses.
 * field projection over the QUERY axis -- models/qformer_utils.py:54:
 *   out[b][f][e] = sum_q Wf[f][q] * rec[b][q][e] + bf[f]   (rec = reconstruction_head(query_outputs),
 *   bf16 [B,Q,E]; out f32 [B,F,E]; Wf [F,Q], bf [F] f32; Q <= 64, F <= 32).
 * QFormerLoss -- training/item_qformer_training.py:49-56: sum(mask*(rec-x)^2)/sum(mask) (divides by
 *   the number of valid FIELDS) + TripletMarginLoss(margin, p=2, eps 1e-6, mean).
 * eval metrics -- evaluation/evaluate_item_qformer.py:75-92: ur_recon_stats returns
 *   sums3 = { sum mask*(rec-x)^2, sum mask, sum over valid fields of cos(x_f, rec_f) }.
 * ur_mse_loss -- training/user_qformer_training.py:193,209 (nn.MSELoss, mean over all elements).
 * `coef` scales the gradient outputs (loss weight x upstream gradient).  All f32. */
int64_t ur_heads_workspace_bytes(int32_t Q, int32_t F);
int ur_field_projection_fwd(const void* rec, const float* Wf, const float* bf, float* out, int32_t B, int32_t Q, int32_t F,
                            int32_t E, void* stream);
int ur_field_projection_bwd(const float* dout, const void* rec, const float* Wf, void* drec, float* dWf, float* dbf,
                            int32_t B, int32_t Q, int32_t F, int32_t E, void* workspace, int64_t workspace_bytes, void* stream);
int ur_recon_stats(const float* rec, const float* x, const float* mask, float* sums3, int64_t rows, int32_t E, void* workspace,
                   int64_t workspace_bytes, void* stream);
int ur_recon_grad(const float* rec, const float* x, const float* mask, const float* sums3, float coef, float* drec,
                  int64_t rows, int32_t E, void* stream);
int ur_triplet_margin(const float* anchor, const float* pos, const float* neg, float margin, float coef, float* loss,
                      float* d_anchor, int32_t B, int32_t E, void* workspace, int64_t workspace_bytes, void* stream);
int ur_mse_loss(const float* a, const float* b, int64_t n, float coef, float* loss, float* d_a, void* workspace,
                int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise helpers. */
/* fp32 master -> bf16 shadow (n elements), used once per step on the flat parameter buffer. */
int ur_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int ur_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* dst [cols,rows] = src [rows,cols]^T (bf16).  Used for the per-step LoRA A^T operands so that the
 * dX GEMMs against the frozen Qwen3 weights run on K-contiguous operands. */
int ur_transpose_bf16(const void* src, void* dst, int32_t rows, int32_t cols, void* stream);
/* n small bf16 transposes in one launch.  desc (device memory, 8-byte aligned): n records of 32 bytes
   {const void* src; void* dst; int32 rows, cols, tile0, ntx}: dst[c][r] = src[r][c]; tile0 = index of the matrix's first
   32x32 tile in the launch (ascending, desc[0].tile0 = 0), ntx = ceil(cols / 32); total_tiles = sum of all tiles.
   Replaces the per-adapter A^T / B^T operand copies of the LoRA backward (peft keeps A, B as nn.Linear weights;
   train_item_individual_token_joint.py:121-131). */
int ur_transpose_bf16_batched(const void* desc, int32_t n, int32_t total_tiles, void* stream);
/* out = a + b (bf16, n % 8 == 0). */
int ur_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* dx = dy * gelu'(u) (erf GELU, nn.GELU() of the user head, training/user_qformer_training.py:40). */
int ur_gelu_bwd(const void* dy, const void* u, void* dx, int64_t n, void* stream);
/* SwiGLU -- modeling_qwen3.py:82: act = silu(gate) * up.  gu [M, 2I] bf16 (gate | up), act [M,I]. */
int ur_swiglu_fwd(const void* gu, void* act, int32_t M, int32_t I, void* stream);
int ur_swiglu_bwd(const void* dact, const void* gu, void* dgu, int32_t M, int32_t I, void* stream);

/* fused AdamW on a flat f32 parameter buffer (torch.optim.AdamW semantics, decoupled decay):
 * training/item_qformer_training.py:108, training/user_qformer_training.py:196,
 * TrainingArguments(...) default optimizer (:755-773).  grad_scale multiplies the gradient first
 * (1/world_size after a sum all-reduce). */
int ur_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                  void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient all-reduce of the data-parallel step (SURVEY 8(b) communicator calls, 8(e)).
 * The reference trains on ONE device (training/train_item_individual_token_joint.py:755-773 hands the model to the HF
 * Trainer; training/item_qformer_training.py:105-125 and training/user_qformer_training.py:170-208 are plain single-process
 * loops), so these replace nothing: they are the north-star's pure data parallelism -- one process per GPU, an in-place SUM
 * all-reduce per gradient bucket over RCCL / xGMI, overlapped with the backward.
 *   ur_comm_unique_id      rank 0 makes the 128-byte RCCL id and hands it to every rank by any channel
 *                          (unirec_amd.dp: torch.distributed broadcast_object_list / the launcher's store).
 *   ur_comm_init           collective over all `world` ranks: RCCL communicator + ONE library-owned side stream + two events
 *                          on `device`.  world = 1 is valid (the all-reduce is then the identity).
 *   ur_comm_allreduce_async  buf (device, `count` elements of dtype, in place) was written on producer_stream: the side stream
 *                          waits for that stream's work up to this call, then reduces.  Calls queue in order on the side
 *                          stream; every rank must issue the same sequence of (count, dtype).  No host synchronisation.
 *   ur_comm_ticket         number of all-reduces queued so far = the ticket of the last one (0: none yet).
 *   ur_comm_wait_ticket    consumer_stream waits for the all-reduce with that ticket (and, the side stream being in order, every
 *                          earlier one): a consumer can start on bucket k while k + 1 is still in flight.  The library keeps a
 *                          ring of UR_COMM_RING completion events; a ticket older than the ring waits for the oldest one kept
 *                          (later in stream order, so still correct).
 *   ur_comm_wait           = ur_comm_wait_ticket(last ticket): consumer_stream waits for every all-reduce queued so far
 *                          (event fence, no host sync).  buf must stay allocated and untouched by other streams until waited for.
 *   ur_comm_destroy        synchronises the side stream, frees communicator, stream and events.  NULL is a no-op.
 * Every call selects the communicator's device for its own duration and restores the caller's.
 * Errors: < 0 invalid argument (-2: RCCL could not be loaded -- it is resolved with dlopen at the first ur_comm_* call, so
 * the library itself does not depend on it), 1..999 HIP error codes, 1000 + ncclResult_t for RCCL failures. */
#define UR_COMM_ID_BYTES 128
#define UR_COMM_F32 0
#define UR_COMM_BF16 1
int ur_comm_unique_id(void* id_out);
int ur_comm_init(void** comm_out, int32_t rank, int32_t world, const void* unique_id, int32_t device);
int ur_comm_allreduce_async(void* comm, void* buf, int64_t count, int32_t dtype, void* producer_stream);
#define UR_COMM_RING 64
int64_t ur_comm_ticket(void* comm);
int ur_comm_wait_ticket(void* comm, int64_t ticket, void* consumer_stream);
int ur_comm_wait(void* comm, void* consumer_stream);
int ur_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* UNIREC_HIP_H */

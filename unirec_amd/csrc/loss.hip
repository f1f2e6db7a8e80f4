// Ranking head of the joint model: fused L2-normalise + dot over the candidate pool (each candidate
// element is read ONCE -> HBM-bound), InfoNCE loss forward/backward, MRR rank and
// deterministic top-K.  training/train_item_individual_token_joint.py:331-352 (InfoNCELoss.forward),
// :392-419 (MRREvaluator._compute_batch_mrr).  All arithmetic f32.  gfx950 only.
#include "common.hip.h"
#include "unirec_hip.h"

namespace {

constexpr float NORM_EPS = 1e-12f;   // F.normalize eps

// scores[b][j] = cos(user[b], cand(b,j)), j = 0 is the positive, j >= 1 the negatives.
// One wave per candidate row; also stores 1/max(||cand||, eps) for the backward.
__global__ __launch_bounds__(256) void cos_scores_kernel(const float* __restrict__ user, const float* __restrict__ pos,
                                                         const float* __restrict__ neg, float* __restrict__ scores,
                                                         float* __restrict__ inv_norm, int B, int N, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long total = (long)B * (N + 1);
  for (long row = (long)blockIdx.x * 4 + wave; row < total; row += (long)gridDim.x * 4) {
    const int b = (int)(row / (N + 1)), j = (int)(row - (long)b * (N + 1));
    const float* c = (j == 0) ? pos + (long)b * D : neg + ((long)b * N + (j - 1)) * D;
    const float* u = user + (long)b * D;
    float dot = 0.f, cc = 0.f, uu = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
      const float4 cv = *reinterpret_cast<const float4*>(c + d), uv = *reinterpret_cast<const float4*>(u + d);
      dot += cv.x * uv.x + cv.y * uv.y + cv.z * uv.z + cv.w * uv.w;
      cc += cv.x * cv.x + cv.y * cv.y + cv.z * cv.z + cv.w * cv.w;
      uu += uv.x * uv.x + uv.y * uv.y + uv.z * uv.z + uv.w * uv.w;
    }
    dot = wave_sum(dot); cc = wave_sum(cc); uu = wave_sum(uu);
    if (lane == 0) {
      const float ic = 1.0f / fmaxf(sqrtf(cc), NORM_EPS), iu = 1.0f / fmaxf(sqrtf(uu), NORM_EPS);
      scores[row] = dot * ic * iu;
      inv_norm[row] = ic;
    }
  }
}

// per sample: loss_b = -s0/tau + logsumexp_{valid j}(s_j/tau); w[b][j] = d loss_b / d (s_j)  (already / tau)
__global__ __launch_bounds__(256) void infonce_row_kernel(const float* __restrict__ scores, const uint8_t* __restrict__ nmask,
                                                          float* __restrict__ row_loss, float* __restrict__ w, int N, float inv_tau) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* s = scores + (long)b * (N + 1);
  float mx = -__builtin_huge_valf();
  for (int j = tid; j <= N; j += 256) {
    const bool valid = (j == 0) || nmask == nullptr || nmask[(long)b * N + (j - 1)];
    if (valid) mx = fmaxf(mx, s[j] * inv_tau);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j <= N; j += 256) {
    const bool valid = (j == 0) || nmask == nullptr || nmask[(long)b * N + (j - 1)];
    if (valid) sum += __expf(s[j] * inv_tau - mx);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = red[0] + red[1] + red[2] + red[3];
  const float lse = mx + logf(sum);
  if (tid == 0) row_loss[b] = -s[0] * inv_tau + lse;
  for (int j = tid; j <= N; j += 256) {
    const bool valid = (j == 0) || nmask == nullptr || nmask[(long)b * N + (j - 1)];
    float p = valid ? __expf(s[j] * inv_tau - lse) : 0.f;
    if (j == 0) p -= 1.0f;
    w[(long)b * (N + 1) + j] = p * inv_tau;
  }
}

__global__ void mean_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += x[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

// partial[b][chunk][:] = sum_{j in chunk} w[b][j] * cand_hat(b,j)[:]     (cand_hat = cand * inv_norm)
__global__ __launch_bounds__(256) void infonce_bwd_stage1(const float* __restrict__ pos, const float* __restrict__ neg,
                                                          const float* __restrict__ w, const float* __restrict__ inv_norm,
                                                          float* __restrict__ part, int N, int D, int per) {
  const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const int j0 = ch * per, j1 = min(N + 1, j0 + per);
  for (int d = tid * 4; d < D; d += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = j0; j < j1; ++j) {
      const float coef = w[(long)b * (N + 1) + j] * inv_norm[(long)b * (N + 1) + j];
      const float* c = (j == 0) ? pos + (long)b * D : neg + ((long)b * N + (j - 1)) * D;
      const float4 cv = *reinterpret_cast<const float4*>(c + d);
      acc.x += coef * cv.x; acc.y += coef * cv.y; acc.z += coef * cv.z; acc.w += coef * cv.w;
    }
    *reinterpret_cast<float4*>(part + (((long)b * gridDim.x + ch) * D) + d) = acc;
  }
}
// du[b] = gscale * ( sum_chunks part - uhat * sum_j w_j s_j ) / max(||u||, eps)
__global__ __launch_bounds__(256) void infonce_bwd_stage2(const float* __restrict__ user, const float* __restrict__ scores,
                                                          const float* __restrict__ w, const float* __restrict__ part,
                                                          float* __restrict__ du, int N, int D, int nchunks, float gscale) {
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float ws = 0.f, uu = 0.f;
  for (int j = tid; j <= N; j += 256) ws += w[(long)b * (N + 1) + j] * scores[(long)b * (N + 1) + j];
  for (int d = tid; d < D; d += 256) { const float u = user[(long)b * D + d]; uu += u * u; }
  ws = wave_sum(ws); uu = wave_sum(uu);
  if (lane == 0) { red[wave] = ws; red[4 + wave] = uu; }
  __syncthreads();
  ws = red[0] + red[1] + red[2] + red[3];
  uu = red[4] + red[5] + red[6] + red[7];
  const float iu = 1.0f / fmaxf(sqrtf(uu), NORM_EPS);
  for (int d = tid; d < D; d += 256) {
    float s = 0.f;
    for (int c = 0; c < nchunks; ++c) s += part[((long)b * nchunks + c) * D + d];
    du[(long)b * D + d] = gscale * iu * (s - user[(long)b * D + d] * iu * ws);
  }
}

// rank[b] = 1 + #{valid negatives with score strictly greater than the positive's}
__global__ __launch_bounds__(256) void mrr_rank_kernel(const float* __restrict__ scores, const uint8_t* __restrict__ nmask,
                                                       int* __restrict__ rank, int N) {
  __shared__ int red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* s = scores + (long)b * (N + 1);
  const float s0 = s[0];
  int cnt = 0;
  for (int j = 1 + tid; j <= N; j += 256)
    if ((nmask == nullptr || nmask[(long)b * N + (j - 1)]) && s[j] > s0) ++cnt;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) red[wave] = cnt;
  __syncthreads();
  if (tid == 0) rank[b] = 1 + red[0] + red[1] + red[2] + red[3];
}

// descending top-K, lowest index first among equal scores.  One workgroup per row, K selection rounds;
// `taken` marks are kept in a caller-provided byte scratch [B][C].
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ scores, int C, int K, int* __restrict__ idx_out,
                                                   float* __restrict__ val_out, uint8_t* __restrict__ taken) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* s = scores + (long)b * C;
  uint8_t* tk = taken + (long)b * C;
  for (int j = tid; j < C; j += 256) tk[j] = 0;
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    float best = -__builtin_huge_valf(); int besti = 0x7fffffff;
    for (int j = tid; j < C; j += 256) {
      if (tk[j]) continue;
      const float v = s[j];
      if (besti == 0x7fffffff || v > best || (v == best && j < besti)) { best = v; besti = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(besti, o, 64);
      if (oi != 0x7fffffff && (besti == 0x7fffffff || ov > best || (ov == best && oi < besti))) { best = ov; besti = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
      float v = bv[0]; int i = bi[0];
      for (int wv = 1; wv < 4; ++wv)
        if (bi[wv] != 0x7fffffff && (i == 0x7fffffff || bv[wv] > v || (bv[wv] == v && bi[wv] < i))) { v = bv[wv]; i = bi[wv]; }
      idx_out[(long)b * K + k] = (i == 0x7fffffff) ? -1 : i;
      if (val_out) val_out[(long)b * K + k] = v;
      if (i != 0x7fffffff) tk[i] = 1;
    }
    __syncthreads();
  }
}

constexpr int BWD_CHUNKS = 16;

}  // namespace

extern "C" int ur_cosine_scores(const float* user, const float* pos, const float* neg, float* scores, float* cand_inv_norm,
                                int32_t B, int32_t N, int32_t D, void* stream) {
  UR_REQUIRE(user && pos && scores && cand_inv_norm && (N == 0 || neg) && B >= 0 && N >= 0 && D > 0 && (D % 4) == 0, "ur_cosine_scores: bad argument");
  UR_REQUIRE(UR_ALIGNED16(user) && UR_ALIGNED16(pos) && (!neg || UR_ALIGNED16(neg)), "ur_cosine_scores: alignment");
  if (B == 0) return 0;
  const long rows = (long)B * (N + 1);
  long g = (rows + 3) / 4; if (g > 256 * 16) g = 256 * 16;
  hipLaunchKernelGGL(cos_scores_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, user, pos, neg, scores, cand_inv_norm, B, N, D);
  UR_CHECK_LAUNCH("ur_cosine_scores");
  return 0;
}

extern "C" int64_t ur_infonce_workspace_bytes(int32_t B, int32_t N, int32_t D) {
  // row_loss [B] + w [B][N+1] + partial [B][BWD_CHUNKS][D]
  return (int64_t)sizeof(float) * ((int64_t)((B + 3) / 4 * 4) + (int64_t)(((long)B * (N + 1) + 3) / 4 * 4) + (int64_t)B * BWD_CHUNKS * D);
}

extern "C" int ur_infonce_fwd_bwd(const float* user, const float* pos, const float* neg, const uint8_t* neg_mask,
                                  const float* scores, const float* cand_inv_norm, float temperature, float grad_scale,
                                  float* loss, float* d_user, int32_t B, int32_t N, int32_t D, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  UR_REQUIRE(user && pos && scores && cand_inv_norm && loss && (N == 0 || neg) && B > 0 && N >= 0 && D > 0 && (D % 4) == 0 && temperature > 0.f,
             "ur_infonce_fwd_bwd: bad argument");
  UR_REQUIRE(workspace && UR_ALIGNED16(workspace) && workspace_bytes >= ur_infonce_workspace_bytes(B, N, D), "ur_infonce_fwd_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* row_loss = (float*)workspace;
  float* w = row_loss + (B + 3) / 4 * 4;
  float* part = w + ((long)B * (N + 1) + 3) / 4 * 4;
  hipLaunchKernelGGL(infonce_row_kernel, dim3(B), dim3(256), 0, st, scores, neg_mask, row_loss, w, N, 1.0f / temperature);
  UR_CHECK_LAUNCH("ur_infonce(row)");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, st, (const float*)row_loss, loss, B);
  UR_CHECK_LAUNCH("ur_infonce(mean)");
  if (d_user) {
    const int per = ur_cdiv(N + 1, BWD_CHUNKS);
    hipLaunchKernelGGL(infonce_bwd_stage1, dim3(BWD_CHUNKS, B), dim3(256), 0, st, pos, neg, (const float*)w, cand_inv_norm, part, N, D, per);
    UR_CHECK_LAUNCH("ur_infonce(bwd1)");
    hipLaunchKernelGGL(infonce_bwd_stage2, dim3(B), dim3(256), 0, st, user, scores, (const float*)w, (const float*)part, d_user, N, D,
                       BWD_CHUNKS, grad_scale / (float)B);
    UR_CHECK_LAUNCH("ur_infonce(bwd2)");
  }
  return 0;
}

extern "C" int ur_mrr_rank(const float* scores, const uint8_t* neg_mask, int32_t* rank, int32_t B, int32_t N, void* stream) {
  UR_REQUIRE(scores && rank && B >= 0 && N >= 0, "ur_mrr_rank: bad argument");
  if (B == 0) return 0;
  hipLaunchKernelGGL(mrr_rank_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, neg_mask, rank, N);
  UR_CHECK_LAUNCH("ur_mrr_rank");
  return 0;
}

extern "C" int ur_topk(const float* scores, int32_t B, int32_t C, int32_t K, int32_t* idx_out, float* val_out, void* workspace,
                       int64_t workspace_bytes, void* stream) {
  UR_REQUIRE(scores && idx_out && B >= 0 && C > 0 && K > 0 && K <= C, "ur_topk: bad argument");
  UR_REQUIRE(workspace && workspace_bytes >= (int64_t)B * C, "ur_topk: workspace must hold B*C bytes");
  if (B == 0) return 0;
  hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, C, K, idx_out, val_out, (uint8_t*)workspace);
  UR_CHECK_LAUNCH("ur_topk");
  return 0;
}

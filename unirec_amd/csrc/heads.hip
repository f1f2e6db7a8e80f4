// Item Q-Former heads and losses (small, HBM-bound):
//   field_projection over the QUERY axis: out[b][f][e] = sum_q Wf[f][q] * rec[b][q][e] + bf[f]
//     (models/qformer_utils.py:54: Linear(Q -> F) applied to reconstruction_head(qo)^T, transposed back)
//   QFormerLoss (training/item_qformer_training.py:49-56): masked MSE / #valid fields + TripletMargin
//   eval metrics (evaluation/evaluate_item_qformer.py:75-92): masked MSE + cosine sum over valid fields
//   MSE loss of the user Q-Former (training/user_qformer_training.py:193,209)
#include "common.cuh"
#include "unirec_hip.h"

namespace {

constexpr int MAXQ = 64, MAXF = 32;

// rec [B][Q][E] bf16 -> out [B][F][E] f32
__global__ __launch_bounds__(256) void fproj_fwd_kernel(const bf16_t* __restrict__ rec, const float* __restrict__ Wf,
                                                        const float* __restrict__ bf, float* __restrict__ out, int B, int Q, int F,
                                                        int E) {
  __shared__ float w[MAXF * MAXQ];
  __shared__ float bb[MAXF];
  for (int i = threadIdx.x; i < F * Q; i += 256) w[i] = Wf[i];
  for (int i = threadIdx.x; i < F; i += 256) bb[i] = bf[i];
  __syncthreads();
  const long total = (long)B * E;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / E; const int e = (int)(i - b * E);
    float r[MAXQ];
    for (int q = 0; q < Q; ++q) r[q] = bf2f(rec[(b * Q + q) * E + e]);
    for (int f = 0; f < F; ++f) {
      float s = bb[f];
      for (int q = 0; q < Q; ++q) s += w[f * Q + q] * r[q];
      out[(b * F + f) * E + e] = s;
    }
  }
}

// dout [B][F][E] f32 -> drec [B][Q][E] bf16 ; partial dWf/dbf per block -> part[nblk][F*Q + F]
__global__ __launch_bounds__(256) void fproj_bwd_kernel(const float* __restrict__ dout, const bf16_t* __restrict__ rec,
                                                        const float* __restrict__ Wf, bf16_t* __restrict__ drec,
                                                        float* __restrict__ part, int B, int Q, int F, int E) {
  __shared__ float w[MAXF * MAXQ];
  __shared__ float accw[4][MAXF * MAXQ + MAXF];      // one accumulator set per wave, summed in wave order: bitwise reproducible
  const int nacc = F * Q + F;
  float* acc = accw[threadIdx.x >> 6];
  for (int i = threadIdx.x; i < F * Q; i += 256) w[i] = Wf[i];
  for (int i = threadIdx.x; i < nacc; i += 256) { accw[0][i] = 0.f; accw[1][i] = 0.f; accw[2][i] = 0.f; accw[3][i] = 0.f; }
  __syncthreads();
  const long total = (long)B * E;
  for (long i0 = (long)blockIdx.x * 256; i0 < total; i0 += (long)gridDim.x * 256) {
    const long i = i0 + threadIdx.x;
    const bool ok = i < total;
    const long b = ok ? i / E : 0; const int e = ok ? (int)(i - b * E) : 0;
    float g[MAXF];
    for (int f = 0; f < F; ++f) g[f] = ok ? dout[(b * F + f) * E + e] : 0.f;
    for (int q = 0; q < Q; ++q) {
      const float r = ok ? bf2f(rec[(b * Q + q) * E + e]) : 0.f;
      float s = 0.f;
      for (int f = 0; f < F; ++f) {
        s += w[f * Q + q] * g[f];
        const float c = wave_sum(g[f] * r);
        if ((threadIdx.x & 63) == 0) acc[f * Q + q] += c;
      }
      if (ok) drec[(b * Q + q) * E + e] = f2bf(s);
    }
    for (int f = 0; f < F; ++f) {
      const float c = wave_sum(g[f]);
      if ((threadIdx.x & 63) == 0) acc[F * Q + f] += c;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nacc; i += 256) part[(long)blockIdx.x * nacc + i] = (accw[0][i] + accw[1][i]) + (accw[2][i] + accw[3][i]);
}
__global__ void part_reduce_kernel(const float* __restrict__ part, int nparts, int n, float* __restrict__ o0, int n0,
                                   float* __restrict__ o1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[(long)p * n + i];
  if (i < n0) o0[i] = s; else o1[i - n0] = s;
}

// ---- scalar reductions over big tensors: two-stage, deterministic ------------------------------
// masked squared error: sums[0] = sum mask*(rec-x)^2, sums[1] = sum mask (per FIELD, not x E),
// sums[2] = sum over valid fields of cos(x_f, rec_f), one wave per (b,f) row.
__global__ __launch_bounds__(256) void recon_stats_kernel(const float* __restrict__ rec, const float* __restrict__ x,
                                                          const float* __restrict__ mask, float* __restrict__ part, long rows, int E) {
  __shared__ float red[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float m = mask[row];
    float se = 0.f, dot = 0.f, xx = 0.f, rr = 0.f;
    for (int e = lane; e < E; e += 64) {
      const float r = rec[row * E + e], v = x[row * E + e];
      se += (r - v) * (r - v); dot += r * v; xx += v * v; rr += r * r;
    }
    se = wave_sum(se); dot = wave_sum(dot); xx = wave_sum(xx); rr = wave_sum(rr);
    a0 += m * se; a1 += m;
    if (m != 0.f) a2 += dot / (fmaxf(sqrtf(xx), 1e-12f) * fmaxf(sqrtf(rr), 1e-12f));
  }
  if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = a2; }
  __syncthreads();
  if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void small_sum_kernel(const float* __restrict__ part, int nparts, int k, float* __restrict__ out) {
  const int j = threadIdx.x;
  if (j >= k) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[p * k + j];
  out[j] = s;
}
// d_rec = coef * 2 * mask * (rec - x) / sum(mask)     (coef = recon weight * upstream grad)
__global__ void recon_grad_kernel(const float* __restrict__ rec, const float* __restrict__ x, const float* __restrict__ mask,
                                  const float* __restrict__ sums, float coef, float* __restrict__ drec, long rows, int E) {
  const long total = rows * E, stride = (long)gridDim.x * blockDim.x;
  const float sc = 2.0f * coef / sums[1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) drec[i] = sc * mask[i / E] * (rec[i] - x[i]);
}
// triplet margin (p=2, eps=1e-6 added to the difference, mean over batch): per-row loss + grad wrt anchor
__global__ __launch_bounds__(256) void triplet_kernel(const float* __restrict__ a, const float* __restrict__ p, const float* __restrict__ n,
                                                      float margin, float coef, float* __restrict__ row_loss, float* __restrict__ da,
                                                      int B, int E) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
    float sp = 0.f, sn = 0.f;
    for (int e = lane; e < E; e += 64) {
      const float dp = a[(long)b * E + e] - p[(long)b * E + e] + 1e-6f, dn = a[(long)b * E + e] - n[(long)b * E + e] + 1e-6f;
      sp += dp * dp; sn += dn * dn;
    }
    sp = sqrtf(wave_sum(sp)); sn = sqrtf(wave_sum(sn));
    const float l = sp - sn + margin;
    if (lane == 0) row_loss[b] = fmaxf(l, 0.f);
    if (da) {
      const float act = (l > 0.f) ? coef / (float)B : 0.f;
      for (int e = lane; e < E; e += 64) {
        const float dp = a[(long)b * E + e] - p[(long)b * E + e] + 1e-6f, dn = a[(long)b * E + e] - n[(long)b * E + e] + 1e-6f;
        da[(long)b * E + e] = act * (dp / fmaxf(sp, 1e-30f) - dn / fmaxf(sn, 1e-30f));
      }
    }
  }
}
// plain MSE (mean over all elements): part sums, then grad
__global__ __launch_bounds__(256) void mse_part_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                       float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float d = a[i] - b[i]; s += d * d; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void mse_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float coef, float* __restrict__ da) {
  const float sc = 2.0f * coef / (float)n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) da[i] = sc * (a[i] - b[i]);
}
__global__ void scale_sum_kernel(const float* __restrict__ part, int nparts, float scale, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s * scale;
}

constexpr int NBLK = 256;

}  // namespace

extern "C" int64_t ur_heads_workspace_bytes(int32_t Q, int32_t F) {
  const int64_t a = (int64_t)NBLK * ((int64_t)F * Q + F);
  return (a > NBLK * 3 ? a : NBLK * 3) * (int64_t)sizeof(float);
}

extern "C" int ur_field_projection_fwd(const void* rec, const float* Wf, const float* bf, float* out, int32_t B, int32_t Q, int32_t F,
                                       int32_t E, void* stream) {
  UR_REQUIRE(rec && Wf && bf && out && B >= 0 && Q > 0 && Q <= MAXQ && F > 0 && F <= MAXF && E > 0, "ur_field_projection_fwd: need Q <= 64, F <= 32");
  if (B == 0) return 0;
  long g = ((long)B * E + 255) / 256; if (g > 2048) g = 2048;
  hipLaunchKernelGGL(fproj_fwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)rec, Wf, bf, out, B, Q, F, E);
  UR_CHECK_LAUNCH("ur_field_projection_fwd");
  return 0;
}

extern "C" int ur_field_projection_bwd(const float* dout, const void* rec, const float* Wf, void* drec, float* dWf, float* dbf,
                                       int32_t B, int32_t Q, int32_t F, int32_t E, void* workspace, int64_t workspace_bytes, void* stream) {
  UR_REQUIRE(dout && rec && Wf && drec && dWf && dbf && B > 0 && Q > 0 && Q <= MAXQ && F > 0 && F <= MAXF && E > 0, "ur_field_projection_bwd: bad argument");
  UR_REQUIRE(workspace && workspace_bytes >= ur_heads_workspace_bytes(Q, F), "ur_field_projection_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  long g = ((long)B * E + 255) / 256; if (g > NBLK) g = NBLK;
  hipLaunchKernelGGL(fproj_bwd_kernel, dim3((int)g), dim3(256), 0, st, dout, (const bf16_t*)rec, Wf, (bf16_t*)drec, (float*)workspace, B, Q, F, E);
  UR_CHECK_LAUNCH("ur_field_projection_bwd");
  const int n = F * Q + F;
  hipLaunchKernelGGL(part_reduce_kernel, dim3(ur_cdiv(n, 256)), dim3(256), 0, st, (const float*)workspace, (int)g, n, dWf, F * Q, dbf);
  UR_CHECK_LAUNCH("ur_field_projection_bwd(reduce)");
  return 0;
}

extern "C" int ur_recon_stats(const float* rec, const float* x, const float* mask, float* sums3, int64_t rows, int32_t E, void* workspace,
                              int64_t workspace_bytes, void* stream) {
  UR_REQUIRE(rec && x && mask && sums3 && rows > 0 && E > 0, "ur_recon_stats: bad argument");
  UR_REQUIRE(workspace && workspace_bytes >= (int64_t)NBLK * 3 * (int64_t)sizeof(float), "ur_recon_stats: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  long g = (rows + 3) / 4; if (g > NBLK) g = NBLK;
  hipLaunchKernelGGL(recon_stats_kernel, dim3((int)g), dim3(256), 0, st, rec, x, mask, (float*)workspace, (long)rows, E);
  UR_CHECK_LAUNCH("ur_recon_stats");
  hipLaunchKernelGGL(small_sum_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, (int)g, 3, sums3);
  UR_CHECK_LAUNCH("ur_recon_stats(sum)");
  return 0;
}

extern "C" int ur_recon_grad(const float* rec, const float* x, const float* mask, const float* sums3, float coef, float* drec,
                             int64_t rows, int32_t E, void* stream) {
  UR_REQUIRE(rec && x && mask && sums3 && drec && rows > 0 && E > 0, "ur_recon_grad: bad argument");
  long g = (rows * E + 255) / 256; if (g > 2048) g = 2048;
  hipLaunchKernelGGL(recon_grad_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, rec, x, mask, sums3, coef, drec, (long)rows, E);
  UR_CHECK_LAUNCH("ur_recon_grad");
  return 0;
}

extern "C" int ur_triplet_margin(const float* anchor, const float* pos, const float* neg, float margin, float coef, float* loss,
                                 float* d_anchor, int32_t B, int32_t E, void* workspace, int64_t workspace_bytes, void* stream) {
  UR_REQUIRE(anchor && pos && neg && loss && B > 0 && E > 0, "ur_triplet_margin: bad argument");
  UR_REQUIRE(workspace && workspace_bytes >= (int64_t)B * (int64_t)sizeof(float), "ur_triplet_margin: workspace must hold B floats");
  hipStream_t st = (hipStream_t)stream;
  int g = (B + 3) / 4; if (g > 1024) g = 1024;
  hipLaunchKernelGGL(triplet_kernel, dim3(g), dim3(256), 0, st, anchor, pos, neg, margin, coef, (float*)workspace, d_anchor, B, E);
  UR_CHECK_LAUNCH("ur_triplet_margin");
  hipLaunchKernelGGL(scale_sum_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, B, 1.0f / (float)B, loss);
  UR_CHECK_LAUNCH("ur_triplet_margin(mean)");
  return 0;
}

extern "C" int ur_mse_loss(const float* a, const float* b, int64_t n, float coef, float* loss, float* d_a, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  UR_REQUIRE(a && b && loss && n > 0, "ur_mse_loss: bad argument");
  UR_REQUIRE(workspace && workspace_bytes >= (int64_t)NBLK * (int64_t)sizeof(float), "ur_mse_loss: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_part_kernel, dim3(NBLK), dim3(256), 0, st, a, b, (long)n, (float*)workspace);
  UR_CHECK_LAUNCH("ur_mse_loss");
  hipLaunchKernelGGL(scale_sum_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, NBLK, 1.0f / (float)n, loss);
  UR_CHECK_LAUNCH("ur_mse_loss(mean)");
  if (d_a) {
    long g = (n + 255) / 256; if (g > 2048) g = 2048;
    hipLaunchKernelGGL(mse_grad_kernel, dim3((int)g), dim3(256), 0, st, a, b, (long)n, coef, d_a);
    UR_CHECK_LAUNCH("ur_mse_loss(grad)");
  }
  return 0;
}

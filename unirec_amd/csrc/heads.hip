// Item Q-Former heads and losses (small, HBM-bound):
//   field_projection over the QUERY axis: out[b][f][e] = sum_q Wf[f][q] * rec[b][q][e] + bf[f]
//     (models/qformer_utils.py:54: Linear(Q -> F) applied to reconstruction_head(qo)^T, transposed back)
//   QFormerLoss (training/item_qformer_training.py:49-56): masked MSE / #valid fields + TripletMargin
//   eval metrics (evaluation/evaluate_item_qformer.py:75-92): masked MSE + cosine sum over valid fields
//   MSE loss of the user Q-Former (training/user_qformer_training.py:193,209)
#include "common.hip.h"
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
// ---- fast paths for the reference's shape (Q = 32 query tokens, F <= 16 fields, E a multiple of 256) --------------------
// The generic kernels above index per-thread arrays with runtime bounds (they live in scratch) and, in the backward, pay
// F*Q wave reductions per element: 165 / 527 us at B 256, E 1024.  Here everything is compile-time indexed; the weight
// gradient dWf[f][q] = sum_{b,e} dout[b][f][e] rec[b][q][e] is an MFMA contraction over e (both operands are e-contiguous
// as they lie: dout rows f -> the row operand, rounded to bf16; rec rows q -> the column operand).
constexpr int FQ = 32, FPAD = 16;
__global__ __launch_bounds__(256) void fproj_fwd32_kernel(const bf16_t* __restrict__ rec, const float* __restrict__ Wf,
                                                          const float* __restrict__ bf, float* __restrict__ out, int B, int F, int E) {
  __shared__ float w[FPAD * FQ];
  __shared__ float bb[FPAD];
  for (int i = threadIdx.x; i < FPAD * FQ; i += 256) w[i] = i < F * FQ ? Wf[i] : 0.f;
  if (threadIdx.x < FPAD) bb[threadIdx.x] = threadIdx.x < F ? bf[threadIdx.x] : 0.f;
  __syncthreads();
  const int e4 = E >> 2;
  const long total = (long)B * e4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / e4; const int e = (int)(i - b * e4) * 4;
    float acc[FPAD][4];
#pragma unroll
    for (int f = 0; f < FPAD; ++f) { acc[f][0] = acc[f][1] = acc[f][2] = acc[f][3] = bb[f]; }
#pragma unroll 4
    for (int q = 0; q < FQ; ++q) {
      const uint2 rv = *reinterpret_cast<const uint2*>(rec + (b * FQ + q) * E + e);
      const float r0 = bf_lo(rv.x), r1 = bf_hi(rv.x), r2 = bf_lo(rv.y), r3 = bf_hi(rv.y);
#pragma unroll
      for (int f = 0; f < FPAD; ++f) {
        const float wv = w[f * FQ + q];
        acc[f][0] = fmaf(wv, r0, acc[f][0]); acc[f][1] = fmaf(wv, r1, acc[f][1]);
        acc[f][2] = fmaf(wv, r2, acc[f][2]); acc[f][3] = fmaf(wv, r3, acc[f][3]);
      }
    }
#pragma unroll
    for (int f = 0; f < FPAD; ++f)
      if (f < F) *reinterpret_cast<float4*>(out + (b * F + f) * E + e) = make_float4(acc[f][0], acc[f][1], acc[f][2], acc[f][3]);
  }
}

// one block per batch row at a time; part[block][F*32 + F] holds the block's dWf | dbf partial sums
__global__ __launch_bounds__(256) void fproj_bwd32_kernel(const float* __restrict__ dout, const bf16_t* __restrict__ rec,
                                                          const float* __restrict__ Wf, bf16_t* __restrict__ drec,
                                                          float* __restrict__ part, int B, int F, int E) {
  __shared__ float w[FPAD * FQ];
  __shared__ float red[4][FPAD * FQ + FPAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  for (int i = tid; i < FPAD * FQ; i += 256) w[i] = i < F * FQ ? Wf[i] : 0.f;
  __syncthreads();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int ew = E >> 2;                                     // e range of one wave
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    // ---- dWf / dbf: D[f][q] += sum_e dout[b][f][e] rec[b][q][e], 32 e per MFMA step
    const float* drow = dout + ((long)b * F + min(l15, F - 1)) * E + wave * ew + 8 * g;
    const bf16_t* r0 = rec + ((long)b * FQ + l15) * E + wave * ew + 8 * g;
    const bf16_t* r1 = r0 + 16L * E;
    for (int e0 = 0; e0 < ew; e0 += 32) {
      float4 d0 = *reinterpret_cast<const float4*>(drow + e0), d1 = *reinterpret_cast<const float4*>(drow + e0 + 4);
      if (l15 >= F) { d0 = make_float4(0.f, 0.f, 0.f, 0.f); d1 = d0; }
      bsum += ((d0.x + d0.y) + (d0.z + d0.w)) + ((d1.x + d1.y) + (d1.z + d1.w));
      const uint4 ap = make_uint4(pack_bf2(d0.x, d0.y), pack_bf2(d0.z, d0.w), pack_bf2(d1.x, d1.y), pack_bf2(d1.z, d1.w));
      const bf16x8 af = __builtin_bit_cast(bf16x8, ap);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(r0 + e0), b1 = *reinterpret_cast<const bf16x8*>(r1 + e0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b1, acc1, 0, 0, 0);
    }
    // ---- drec[b][q][e] = sum_f Wf[f][q] dout[b][f][e]: thread <-> 4 consecutive e
    for (int e = tid * 4; e < E; e += 1024) {
      float gv[FPAD][4];
#pragma unroll
      for (int f = 0; f < FPAD; ++f) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < F) v = *reinterpret_cast<const float4*>(dout + ((long)b * F + f) * E + e);
        gv[f][0] = v.x; gv[f][1] = v.y; gv[f][2] = v.z; gv[f][3] = v.w;
      }
#pragma unroll 4
      for (int q = 0; q < FQ; ++q) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int f = 0; f < FPAD; ++f) {
          const float wv = w[f * FQ + q];
          s0 = fmaf(wv, gv[f][0], s0); s1 = fmaf(wv, gv[f][1], s1); s2 = fmaf(wv, gv[f][2], s2); s3 = fmaf(wv, gv[f][3], s3);
        }
        *reinterpret_cast<uint2*>(drec + ((long)b * FQ + q) * E + e) = make_uint2(pack_bf2(s0, s1), pack_bf2(s2, s3));
      }
    }
  }
  // accumulator layout of v_mfma_f32_16x16x32: lane holds D[row = 4 g + i][col = l15], i = 0..3  (row = f, col = q)
  bsum += __shfl_xor(bsum, 16, 64); bsum += __shfl_xor(bsum, 32, 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    red[wave][(4 * g + i) * FQ + l15] = acc0[i];
    red[wave][(4 * g + i) * FQ + 16 + l15] = acc1[i];
  }
  if (g == 0) red[wave][FPAD * FQ + l15] = bsum;
  __syncthreads();
  const int nacc = F * FQ + F;
  for (int i = tid; i < nacc; i += 256) {
    const int j = i < F * FQ ? i : FPAD * FQ + (i - F * FQ);
    part[(long)blockIdx.x * nacc + i] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
  }
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
  if (Q == FQ && F <= FPAD && (E % 4) == 0 && UR_ALIGNED16(out) && (((uintptr_t)rec) & 7) == 0) {
    long g4 = ((long)B * (E / 4) + 255) / 256; if (g4 > 4096) g4 = 4096;
    hipLaunchKernelGGL(fproj_fwd32_kernel, dim3((int)g4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)rec, Wf, bf, out, B, F, E);
    UR_CHECK_LAUNCH("ur_field_projection_fwd");
    return 0;
  }
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
  if (Q == FQ && F <= FPAD && (E % 128) == 0 && UR_ALIGNED16(dout) && UR_ALIGNED16(rec) && (((uintptr_t)drec) & 7) == 0) {
    g = B < NBLK ? B : NBLK;
    hipLaunchKernelGGL(fproj_bwd32_kernel, dim3((int)g), dim3(256), 0, st, dout, (const bf16_t*)rec, Wf, (bf16_t*)drec, (float*)workspace, B, F, E);
  } else {
    hipLaunchKernelGGL(fproj_bwd_kernel, dim3((int)g), dim3(256), 0, st, dout, (const bf16_t*)rec, Wf, (bf16_t*)drec, (float*)workspace, B, Q, F, E);
  }
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

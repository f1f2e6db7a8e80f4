// LayerNorm / RMSNorm forward+backward and batch reductions (HBM-bound, one wave per row,
// 16-byte vector loads, wave-shuffle row reductions, f32 statistics).  gfx950 only.
#include "common.hip.h"
#include "unirec_hip.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;   // 256 threads = 4 waves = 4 rows in flight per block

struct Drop {
  uint32_t thr; float inv_keep; uint64_t seed;
  uint64_t idx0;      // element index of this launch's first row in the GLOBAL batch (drop_row0 * H): a data-parallel rank draws the
                      // masks its samples would get in a single-process run over the whole minibatch
  __device__ __forceinline__ bool on() const { return thr != 0; }
};
static Drop make_drop(float p, uint64_t seed, int64_t row0, int H) {
  Drop d; d.thr = (p > 0.f) ? ur_drop_threshold(p) : 0u; d.inv_keep = (p > 0.f) ? 1.0f / (1.0f - p) : 1.0f; d.seed = seed;
  d.idx0 = (uint64_t)row0 * (uint64_t)H;
  return d;
}

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
  f[4] = bf_lo(u.z); f[5] = bf_hi(u.z); f[6] = bf_lo(u.w); f[7] = bf_hi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}
__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ y, int y_rows,
                                                     const bf16_t* __restrict__ res, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                     bf16_t* __restrict__ zsave, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int M, int H, float eps, Drop pre,
                                                     Drop post) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += gridDim.x * ROWS_PER_BLOCK) {
    const long yoff = (long)(row % y_rows) * H, roff = (long)row * H;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e0 = (lane + i * 64) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
      if (e0 < H) {
        unpack8(*reinterpret_cast<const uint4*>(y + yoff + e0), v[i]);
        if (pre.on()) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[i][e] *= ur_dropout_scale(pre.seed, pre.idx0 + (uint64_t)(roff + e0 + e), pre.thr, pre.inv_keep);
        }
        if (res) {
          float r[8];
          unpack8(*reinterpret_cast<const uint4*>(res + roff + e0), r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[i][e] += r[e];
        }
        {   // round z to bf16 first so forward and backward see the same z (also when no z is kept: a no-grad forward is bit-identical)
          uint4 pk = pack8(v[i]);
          if (zsave) *reinterpret_cast<uint4*>(zsave + roff + e0) = pk;
          unpack8(pk, v[i]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
      }
    }
    const float mu = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e0 = (lane + i * 64) * 8;
      if (e0 < H) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)H + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e0 = (lane + i * 64) * 8;
      if (e0 < H) {
        float g[8], b[8], o[8];
        load8f(gamma + e0, g); load8f(beta + e0, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
          if (post.on()) o[e] *= ur_dropout_scale(post.seed, post.idx0 + (uint64_t)(roff + e0 + e), post.thr, post.inv_keep);
        }
        *reinterpret_cast<uint4*>(out + roff + e0) = pack8(o);
      }
    }
  }
}

// backward: per-wave column partials (dgamma, dbeta, dbias) go to workspace [nwaves_total][3][H]
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, bf16_t* __restrict__ dz,
                                                     bf16_t* __restrict__ dy, float* __restrict__ part, int M, int H,
                                                     Drop pre, Drop post) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pg[NCH][8], pb[NCH][8], pd[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { pg[i][e] = 0.f; pb[i][e] = 0.f; pd[i][e] = 0.f; }

  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += gridDim.x * ROWS_PER_BLOCK) {
    const long roff = (long)row * H;
    const float mu = mean[row], rs = rstd[row];
    float go[NCH][8], xh[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e0 = (lane + i * 64) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { go[i][e] = 0.f; xh[i][e] = 0.f; }
      if (e0 < H) {
        float zz[8], g[8];
        unpack8(*reinterpret_cast<const uint4*>(dout + roff + e0), go[i]);
        unpack8(*reinterpret_cast<const uint4*>(z + roff + e0), zz);
        load8f(gamma + e0, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (post.on()) go[i][e] *= ur_dropout_scale(post.seed, post.idx0 + (uint64_t)(roff + e0 + e), post.thr, post.inv_keep);
          xh[i][e] = (zz[e] - mu) * rs;
          pg[i][e] += go[i][e] * xh[i][e];
          pb[i][e] += go[i][e];
          go[i][e] *= g[e];                       // g = dout * gamma
          s1 += go[i][e];
          s2 += go[i][e] * xh[i][e];
        }
      }
    }
    const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e0 = (lane + i * 64) * 8;
      if (e0 < H) {
        float d[8], dd[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          d[e] = rs * (go[i][e] - m1 - xh[i][e] * m2);
          dd[e] = d[e];
        }
        const uint4 pk = pack8(d);
        *reinterpret_cast<uint4*>(dz + roff + e0) = pk;
        if (pre.on()) {
#pragma unroll
          for (int e = 0; e < 8; ++e) dd[e] *= ur_dropout_scale(pre.seed, pre.idx0 + (uint64_t)(roff + e0 + e), pre.thr, pre.inv_keep);
          if (dy) *reinterpret_cast<uint4*>(dy + roff + e0) = pack8(dd);
        } else if (dy && dy != dz) {
          *reinterpret_cast<uint4*>(dy + roff + e0) = pk;
        }
        float dr[8];
        unpack8(pre.on() ? pack8(dd) : pk, dr);   // bias grad sums the bf16 values the dense bwd will see
#pragma unroll
        for (int e = 0; e < 8; ++e) pd[i][e] += dr[e];
      }
    }
  }
  // one partial row per BLOCK: the four waves fold their column sums in LDS, in wave order (deterministic), so the grid can be
  // four times larger at the same number of partial rows (M = 8192 rows used to be 16 dependent rows per wave on half the CUs)
  __shared__ __attribute__((aligned(16))) float comb[3 * NCH * 512];
#pragma unroll 1
  for (int w = 0; w < ROWS_PER_BLOCK; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int e0 = (lane + i * 64) * 8;
        float* c0 = comb + e0; float* c1 = comb + NCH * 512 + e0; float* c2 = comb + 2 * NCH * 512 + e0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (w == 0) { c0[e] = pg[i][e]; c1[e] = pb[i][e]; c2[e] = pd[i][e]; }
          else { c0[e] += pg[i][e]; c1[e] += pb[i][e]; c2[e] += pd[i][e]; }
        }
      }
    }
    __syncthreads();
  }
  float* base = part + (long)blockIdx.x * 3 * H;
  for (int c = threadIdx.x; c < H; c += 256) {
    base[c] = comb[c];
    base[H + c] = comb[NCH * 512 + c];
    base[2 * H + c] = comb[2 * NCH * 512 + c];
  }
}

// out[k][c] = sum_p part[p][k][c]   (k in 0..nout-1), out pointers may be null.  1024 threads = 64 columns x 16 groups
// of partial rows (a thread per column walked thousands of partial rows one load at a time: 119 us for 3072 columns).
__global__ __launch_bounds__(1024) void colpart_reduce_kernel(const float* __restrict__ part, int nparts, int ncols_total, float* o0,
                                                             float* o1, float* o2, int H) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < ncols_total)
    for (int p = g; p < nparts; p += 16) s += part[(long)p * ncols_total + c];
  red[g][lane] = s;
  __syncthreads();
  if (g == 0 && c < ncols_total) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][lane];
    const int k = c / H, h = c - k * H;
    float* o = (k == 0) ? o0 : (k == 1 ? o1 : o2);
    if (o) o[h] = s;
  }
}

// ---- batch reduce: out[r*H + h] = sum_b in[(b*rows + r)*H + h] ---------------------------------
__global__ void batch_reduce_stage1(const bf16_t* __restrict__ in, float* __restrict__ part, int nb, long rh8,
                                    int per_slice) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;   // 8-element chunk of the [rows*H] vector
  if (c >= rh8) return;
  const int slice = blockIdx.y;
  const int b0 = slice * per_slice, b1 = min(nb, b0 + per_slice);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = b0;
  for (; b + 4 <= b1; b += 4) {                     // four independent loads in flight
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(in + ((long)(b + u) * rh8 + c) * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[8];
      unpack8(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
  }
  for (; b < b1; ++b) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(in + ((long)b * rh8 + c) * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
  }
  float* o = part + ((long)slice * rh8 + c) * 8;
  *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}
// 1024 threads = 64 elements x 16 groups of slices
__global__ __launch_bounds__(1024) void batch_reduce_stage2(const float* __restrict__ part, float* __restrict__ out, long n, int nslices) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (i < n)
    for (int k = g; k < nslices; k += 16) s += part[(long)k * n + i];
  red[g][lane] = s;
  __syncthreads();
  if (g == 0 && i < n) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][lane];
    out[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void rms_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                      bf16_t* __restrict__ out, float* __restrict__ rstd, int M, int D,
                                                      float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += gridDim.x * ROWS_PER_BLOCK) {
    const long roff = (long)row * D;
    float v[NCH][8];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e0 = (lane + i * 64) * 8;
      if (e0 < D) {
        unpack8(*reinterpret_cast<const uint4*>(x + roff + e0), v[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) q += v[i][e] * v[i][e];
      }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) rstd[row] = rs;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e0 = (lane + i * 64) * 8;
      if (e0 < D) {
        float g[8], o[8];
        load8f(w + e0, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = g[e] * (v[i][e] * rs);
        *reinterpret_cast<uint4*>(out + roff + e0) = pack8(o);
      }
    }
  }
}

#ifndef UR_RMS_NT
#define UR_RMS_NT 7             // bits: 1 = x, 2 = dout, 4 = add read non-temporally (each is read once; 10.55 -> 9.85 ms per C4 step), 8 = dx stored non-temporally (no gain)
#endif
typedef unsigned int rms_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 rms_ld16(const bf16_t* p) {
  if (!NT) return *reinterpret_cast<const uint4*>(p);
  const rms_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const rms_u32x4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
template <int NCH>
__global__ __launch_bounds__(256) void rms_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x,
                                                      const float* __restrict__ w, const float* __restrict__ rstd,
                                                      const bf16_t* __restrict__ add, bf16_t* __restrict__ dx, int M,
                                                      int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += gridDim.x * ROWS_PER_BLOCK) {
    const long roff = (long)row * D;
    const float rs = rstd[row];
    float g[NCH][8], xh[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e0 = (lane + i * 64) * 8;
      if (e0 < D) {
        float ww[8];
        unpack8(rms_ld16<(UR_RMS_NT & 2) != 0>(dout + roff + e0), g[i]);
        unpack8(rms_ld16<(UR_RMS_NT & 1) != 0>(x + roff + e0), xh[i]);
        load8f(w + e0, ww);
#pragma unroll
        for (int e = 0; e < 8; ++e) { g[i][e] *= ww[e]; xh[i][e] *= rs; s += g[i][e] * xh[i][e]; }
      }
    }
    const float m = wave_sum(s) / (float)D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e0 = (lane + i * 64) * 8;
      if (e0 < D) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rs * (g[i][e] - xh[i][e] * m);
        if (add) {
          float a[8];
          unpack8(rms_ld16<(UR_RMS_NT & 4) != 0>(add + roff + e0), a);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += a[e];
        }
        if (UR_RMS_NT & 8) {
          const uint4 pv = pack8(o);
          const rms_u32x4 tv = {pv.x, pv.y, pv.z, pv.w};
          __builtin_nontemporal_store(tv, reinterpret_cast<rms_u32x4*>(dx + roff + e0));
        } else {
          *reinterpret_cast<uint4*>(dx + roff + e0) = pack8(o);
        }
      }
    }
  }
}

inline int nch_for(int H) { int c = ur_cdiv(H, 512); return c <= 1 ? 1 : (c <= 2 ? 2 : 4); }
inline int row_grid(int M, int cap) { int g = ur_cdiv(M, ROWS_PER_BLOCK); return g < cap ? (g > 0 ? g : 1) : cap; }
#ifndef UR_LN_BWD_BLOCKS
#define UR_LN_BWD_BLOCKS 512
#endif
constexpr int LN_BWD_BLOCKS = UR_LN_BWD_BLOCKS;   // partial rows of the column sums = blocks (ln_bwd_kernel folds its four waves)

}  // namespace

#define UR_NCH_DISPATCH(H, CALL)                  \
  switch (nch_for(H)) {                           \
    case 1: { constexpr int NCH = 1; CALL; } break; \
    case 2: { constexpr int NCH = 2; CALL; } break; \
    default: { constexpr int NCH = 4; CALL; } break; \
  }

extern "C" int ur_layernorm_fwd(const void* y, int32_t y_rows, const void* residual, const float* gamma,
                                const float* beta, void* out, void* z_save, float* mean, float* rstd, int32_t M,
                                int32_t H, float eps, float p_pre, uint64_t seed_pre, float p_post, uint64_t seed_post,
                                int64_t drop_row0, void* stream) {
  UR_REQUIRE(M >= 0 && H > 0 && (H % 8) == 0 && H <= 2048, "ur_layernorm_fwd: need H %% 8 == 0 and H <= 2048 (H=%d)", H);
  if (M == 0) return 0;
  UR_REQUIRE(y && gamma && beta && out && mean && rstd && y_rows > 0, "ur_layernorm_fwd: null argument");
  UR_REQUIRE(UR_ALIGNED16(y) && UR_ALIGNED16(out) && UR_ALIGNED16(gamma) && UR_ALIGNED16(beta) &&
             (!residual || UR_ALIGNED16(residual)) && (!z_save || UR_ALIGNED16(z_save)), "ur_layernorm_fwd: 16-byte alignment");
  UR_REQUIRE(p_pre >= 0.f && p_pre < 1.f && p_post >= 0.f && p_post < 1.f, "ur_layernorm_fwd: dropout p out of range");
  const Drop pre = make_drop(p_pre, seed_pre, drop_row0, H), post = make_drop(p_post, seed_post, drop_row0, H);
  const int grid = row_grid(M, 4096);
  UR_NCH_DISPATCH(H, hipLaunchKernelGGL((ln_fwd_kernel<NCH>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                        (const bf16_t*)y, y_rows, (const bf16_t*)residual, gamma, beta, (bf16_t*)out,
                                        (bf16_t*)z_save, mean, rstd, M, H, eps, pre, post));
  UR_CHECK_LAUNCH("ur_layernorm_fwd");
  return 0;
}

extern "C" int64_t ur_layernorm_bwd_workspace_bytes(int32_t H) {
  return (int64_t)LN_BWD_BLOCKS * 3 * H * (int64_t)sizeof(float);
}

extern "C" int ur_layernorm_bwd(const void* dout, const void* z, const float* mean, const float* rstd,
                                const float* gamma, void* dz, void* dy, float* dgamma, float* dbeta, float* dbias,
                                int32_t M, int32_t H, float p_pre, uint64_t seed_pre, float p_post, uint64_t seed_post,
                                int64_t drop_row0, void* workspace, int64_t workspace_bytes, void* stream) {
  UR_REQUIRE(M >= 0 && H > 0 && (H % 8) == 0 && H <= 2048, "ur_layernorm_bwd: need H %% 8 == 0 and H <= 2048 (H=%d)", H);
  UR_REQUIRE(dout && z && mean && rstd && gamma && dz && ((dgamma && dbeta) || (!dgamma && !dbeta)), "ur_layernorm_bwd: null argument");
  UR_REQUIRE(workspace && workspace_bytes >= ur_layernorm_bwd_workspace_bytes(H), "ur_layernorm_bwd: workspace too small");
  UR_REQUIRE(UR_ALIGNED16(dout) && UR_ALIGNED16(z) && UR_ALIGNED16(dz) && UR_ALIGNED16(gamma) && (!dy || UR_ALIGNED16(dy)),
             "ur_layernorm_bwd: 16-byte alignment");
  UR_REQUIRE(p_pre >= 0.f && p_pre < 1.f && p_post >= 0.f && p_post < 1.f, "ur_layernorm_bwd: dropout p out of range");
  const Drop pre = make_drop(p_pre, seed_pre, drop_row0, H), post = make_drop(p_post, seed_post, drop_row0, H);
  const int grid = row_grid(M, LN_BWD_BLOCKS);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  UR_NCH_DISPATCH(H, hipLaunchKernelGGL((ln_bwd_kernel<NCH>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dout,
                                        (const bf16_t*)z, mean, rstd, gamma, (bf16_t*)dz, (bf16_t*)dy, part, M, H, pre, post));
  UR_CHECK_LAUNCH("ur_layernorm_bwd");
  if (!dgamma) return 0;            // partial sums only: ur_layernorm_bwd_reduce finishes them (on a stream of the caller's choice)
  const int ncols = 3 * H;
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3(ur_cdiv(ncols, 64)), dim3(1024), 0, st, (const float*)part,
                     grid, ncols, dgamma, dbeta, dbias, H);
  UR_CHECK_LAUNCH("ur_layernorm_bwd(reduce)");
  return 0;
}

extern "C" int ur_layernorm_bwd_reduce(const void* workspace, int32_t M, int32_t H, float* dgamma, float* dbeta, float* dbias, void* stream) {
  UR_REQUIRE(M >= 0 && H > 0 && (H % 8) == 0 && H <= 2048 && workspace && dgamma && dbeta, "ur_layernorm_bwd_reduce: bad argument");
  const int ncols = 3 * H;
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3(ur_cdiv(ncols, 64)), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace,
                     row_grid(M, LN_BWD_BLOCKS), ncols, dgamma, dbeta, dbias, H);
  UR_CHECK_LAUNCH("ur_layernorm_bwd_reduce");
  return 0;
}

// slices of the batch axis: enough of them that slices x (rows*H/8) threads fill the chip (a bias gradient over
// 819,200 tokens x 2048 columns used to run on 256 waves, each walking 12,800 rows one load at a time)
static inline int br_slices(int nb, long n) {
  if (nb <= 0) return 1;
  const long rh8 = n / 8 > 0 ? n / 8 : 1;
  long want = (262144 + rh8 - 1) / rh8;
  if (want < 64) want = 64;
  if (want > 2048) want = 2048;
  const long by_rows = nb / 32 > 0 ? nb / 32 : 1;      // at least 32 rows per slice: stage 2 walks the slices serially
  if (want > by_rows) want = by_rows;
  if (want < 64) want = 64;
  return (int)(nb < want ? nb : want);
}

extern "C" int64_t ur_batch_reduce_workspace_bytes(int32_t nb, int32_t rows, int32_t H) {
  return (int64_t)br_slices(nb, (long)rows * H) * rows * H * (int64_t)sizeof(float);
}

extern "C" int ur_batch_reduce(const void* in, float* out, int32_t nb, int32_t rows, int32_t H, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  UR_REQUIRE(nb >= 0 && rows > 0 && H > 0 && ((long)rows * H) % 8 == 0, "ur_batch_reduce: rows*H must be a multiple of 8");
  UR_REQUIRE(in && out && UR_ALIGNED16(in) && UR_ALIGNED16(out), "ur_batch_reduce: null / misaligned argument");
  UR_REQUIRE(workspace && UR_ALIGNED16(workspace) && workspace_bytes >= ur_batch_reduce_workspace_bytes(nb, rows, H),
             "ur_batch_reduce: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)rows * H, rh8 = n / 8;
  const int ns = br_slices(nb, n);
  const int per = ur_cdiv(nb > 0 ? nb : 1, ns);
  hipLaunchKernelGGL(batch_reduce_stage1, dim3(ur_cdiv(rh8, 128), ns), dim3(128), 0, st, (const bf16_t*)in,
                     (float*)workspace, nb, rh8, per);
  UR_CHECK_LAUNCH("ur_batch_reduce(stage1)");
  hipLaunchKernelGGL(batch_reduce_stage2, dim3(ur_cdiv(n, 64)), dim3(1024), 0, st, (const float*)workspace, out, n, ns);
  UR_CHECK_LAUNCH("ur_batch_reduce(stage2)");
  return 0;
}

extern "C" int ur_rmsnorm_fwd(const void* x, const float* w, void* out, float* rstd, int32_t M, int32_t D, float eps,
                              void* stream) {
  UR_REQUIRE(M >= 0 && D > 0 && (D % 8) == 0 && D <= 2048, "ur_rmsnorm_fwd: need D %% 8 == 0 and D <= 2048 (D=%d)", D);
  if (M == 0) return 0;
  UR_REQUIRE(x && w && out && rstd && UR_ALIGNED16(x) && UR_ALIGNED16(w) && UR_ALIGNED16(out), "ur_rmsnorm_fwd: null / misaligned");
  const int grid = row_grid(M, 8192);
  UR_NCH_DISPATCH(D, hipLaunchKernelGGL((rms_fwd_kernel<NCH>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                        (const bf16_t*)x, w, (bf16_t*)out, rstd, M, D, eps));
  UR_CHECK_LAUNCH("ur_rmsnorm_fwd");
  return 0;
}

extern "C" int ur_rmsnorm_bwd(const void* dout, const void* x, const float* w, const float* rstd, const void* add,
                              void* dx, int32_t M, int32_t D, void* stream) {
  UR_REQUIRE(M >= 0 && D > 0 && (D % 8) == 0 && D <= 2048, "ur_rmsnorm_bwd: need D %% 8 == 0 and D <= 2048 (D=%d)", D);
  if (M == 0) return 0;
  UR_REQUIRE(dout && x && w && rstd && dx && UR_ALIGNED16(dout) && UR_ALIGNED16(x) && UR_ALIGNED16(w) && UR_ALIGNED16(dx) &&
             (!add || UR_ALIGNED16(add)), "ur_rmsnorm_bwd: null / misaligned");
  const int grid = row_grid(M, 8192);
  UR_NCH_DISPATCH(D, hipLaunchKernelGGL((rms_bwd_kernel<NCH>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                        (const bf16_t*)dout, (const bf16_t*)x, w, rstd, (const bf16_t*)add, (bf16_t*)dx, M, D));
  UR_CHECK_LAUNCH("ur_rmsnorm_bwd");
  return 0;
}

// HBM-bound element-wise kernels: casts, adds, SwiGLU, fused AdamW.  16-byte accesses, grid-stride.
#include "common.hip.h"
#include "unirec_hip.h"

thread_local char g_ur_err[512] = {0};

extern "C" int ur_version(void) { return UR_ABI_VERSION; }
extern "C" const char* ur_last_error(void) { return g_ur_err; }

namespace {

inline int ew_grid(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;   // 8 blocks/CU, grid-stride the rest
  return (int)g;
}

__global__ void cast_f2b_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n) {
  const long n8 = n / 8, stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const float4 a = *reinterpret_cast<const float4*>(s + i * 8), b = *reinterpret_cast<const float4*>(s + i * 8 + 4);
    *reinterpret_cast<uint4*>(d + i * 8) =
        make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
  }
  if (blockIdx.x == 0) for (long i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) d[i] = f2bf(s[i]);
}
__global__ void cast_b2f_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, long n) {
  const long n8 = n / 8, stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const uint4 u = *reinterpret_cast<const uint4*>(s + i * 8);
    *reinterpret_cast<float4*>(d + i * 8) = make_float4(bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y));
    *reinterpret_cast<float4*>(d + i * 8 + 4) = make_float4(bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w));
  }
  if (blockIdx.x == 0) for (long i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) d[i] = bf2f(s[i]);
}
__global__ void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ o, long n8) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const uint4 x = *reinterpret_cast<const uint4*>(a + i * 8), y = *reinterpret_cast<const uint4*>(b + i * 8);
    *reinterpret_cast<uint4*>(o + i * 8) =
        make_uint4(pack_bf2(bf_lo(x.x) + bf_lo(y.x), bf_hi(x.x) + bf_hi(y.x)), pack_bf2(bf_lo(x.y) + bf_lo(y.y), bf_hi(x.y) + bf_hi(y.y)),
                   pack_bf2(bf_lo(x.z) + bf_lo(y.z), bf_hi(x.z) + bf_hi(y.z)), pack_bf2(bf_lo(x.w) + bf_lo(y.w), bf_hi(x.w) + bf_hi(y.w)));
  }
}

__device__ __forceinline__ void un8(const uint4& u, float (&f)[8]) {
  f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
  f[4] = bf_lo(u.z); f[5] = bf_hi(u.z); f[6] = bf_lo(u.w); f[7] = bf_hi(u.w);
}
__device__ __forceinline__ uint4 pk8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

// gu [M][2I]: gate = cols [0,I), up = cols [I,2I)
// read-once / written-for-a-much-later-kernel streams: non-temporal accesses keep them from displacing L2 lines
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream(const bf16_t* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st_stream(bf16_t* p, uint4 v) {
  u32x4_t t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p));
}
__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act, long M, int I8, int I) {
  const long total = M * I8, stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long m = i / I8; const int c = (int)(i - m * I8) * 8;
    float g[8], u[8], o[8];
    un8(ld_stream(gu + m * 2 * I + c), g);
    un8(ld_stream(gu + m * 2 * I + I + c), u);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = silu_f(g[e]) * u[e];
    *reinterpret_cast<uint4*>(act + m * I + c) = pk8(o);
  }
}
__global__ void swiglu_bwd_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ gu, bf16_t* __restrict__ dgu,
                                  long M, int I8, int I) {
  const long total = M * I8, stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long m = i / I8; const int c = (int)(i - m * I8) * 8;
    float g[8], u[8], d[8], dg[8], du[8];
    un8(ld_stream(gu + m * 2 * I + c), g);
    un8(ld_stream(gu + m * 2 * I + I + c), u);
    un8(ld_stream(dact + m * I + c), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sg = sigmoid_f(g[e]);
      const float si = g[e] * sg;
      du[e] = d[e] * si;
      dg[e] = d[e] * u[e] * (sg * (1.0f + g[e] * (1.0f - sg)));
    }
    st_stream(dgu + m * 2 * I + c, pk8(dg));
    st_stream(dgu + m * 2 * I + I + c, pk8(du));
  }
}

__global__ void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ u, bf16_t* __restrict__ dx, long n8) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float a[8], b[8], o[8];
    un8(*reinterpret_cast<const uint4*>(dy + i * 8), a);
    un8(*reinterpret_cast<const uint4*>(u + i * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = a[e] * gelu_erf_grad_f(b[e]);
    *reinterpret_cast<uint4*>(dx + i * 8) = pk8(o);
  }
}

// dst[c][r] = src[r][c]  (bf16, 32x32 tiles through LDS; small matrices: LoRA A^T per step)
__global__ void transpose_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols) {
  __shared__ bf16_t t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    t[i][tx] = (r < rows && c < cols) ? src[(long)r * cols + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(long)c * rows + r] = t[tx][i];
  }
}

// many small transposes in ONE launch: desc[i] = {src, dst, rows, cols, first 32x32 tile, tiles per row}; a workgroup
// finds its matrix by bisection over the first-tile column (the LoRA A^T / B^T operands of a whole backward: 308 matrices)
struct TransposeDesc { const bf16_t* src; bf16_t* dst; int rows, cols, tile0, ntx; };
__global__ void transpose_batched_kernel(const TransposeDesc* __restrict__ desc, int n) {
  __shared__ bf16_t t[32][33];
  int lo = 0, hi = n - 1;
  const int blk = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].tile0 <= blk) lo = mid; else hi = mid - 1; }
  const TransposeDesc d = desc[lo];
  const int tl = blk - d.tile0;
  const int c0 = (tl % d.ntx) * 32, r0 = (tl / d.ntx) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    t[i][tx] = (r < d.rows && c < d.cols) ? d.src[(long)r * d.cols + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < d.cols && r < d.rows) d.dst[(long)c * d.rows + r] = t[tx][i];
  }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                             float gscale) {
  const long n4 = n / 4, stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; float* G = &gg.x; float* Mo = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = G[e] * gscale;
      P[e] *= (1.0f - lr * wd);
      Mo[e] = b1 * Mo[e] + (1.0f - b1) * gr;
      V[e] = b2 * V[e] + (1.0f - b2) * gr * gr;
      const float denom = sqrtf(V[e]) / bc2_sqrt + eps;
      P[e] -= (lr / bc1) * (Mo[e] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0) {
    for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float gr = g[i] * gscale;
      float pp = p[i] * (1.0f - lr * wd);
      const float mm = b1 * m[i] + (1.0f - b1) * gr, vv = b2 * v[i] + (1.0f - b2) * gr * gr;
      pp -= (lr / bc1) * (mm / (sqrtf(vv) / bc2_sqrt + eps));
      p[i] = pp; m[i] = mm; v[i] = vv;
    }
  }
}

}  // namespace

extern "C" int ur_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  UR_REQUIRE(n >= 0, "ur_cast_f32_to_bf16: negative n");
  if (n == 0) return 0;
  UR_REQUIRE(src && dst && UR_ALIGNED16(src) && UR_ALIGNED16(dst), "ur_cast_f32_to_bf16: null / misaligned");
  hipLaunchKernelGGL(cast_f2b_kernel, dim3(ew_grid(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
  UR_CHECK_LAUNCH("ur_cast_f32_to_bf16");
  return 0;
}
extern "C" int ur_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  UR_REQUIRE(n >= 0, "ur_cast_bf16_to_f32: negative n");
  if (n == 0) return 0;
  UR_REQUIRE(src && dst && UR_ALIGNED16(src) && UR_ALIGNED16(dst), "ur_cast_bf16_to_f32: null / misaligned");
  hipLaunchKernelGGL(cast_b2f_kernel, dim3(ew_grid(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, (long)n);
  UR_CHECK_LAUNCH("ur_cast_bf16_to_f32");
  return 0;
}
extern "C" int ur_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  UR_REQUIRE(n >= 0 && (n % 8) == 0, "ur_add_bf16: n must be a multiple of 8");
  if (n == 0) return 0;
  UR_REQUIRE(a && b && out && UR_ALIGNED16(a) && UR_ALIGNED16(b) && UR_ALIGNED16(out), "ur_add_bf16: null / misaligned");
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b,
                     (bf16_t*)out, (long)(n / 8));
  UR_CHECK_LAUNCH("ur_add_bf16");
  return 0;
}
extern "C" int ur_gelu_bwd(const void* dy, const void* u, void* dx, int64_t n, void* stream) {
  UR_REQUIRE(n >= 0 && (n % 8) == 0, "ur_gelu_bwd: n must be a multiple of 8");
  if (n == 0) return 0;
  UR_REQUIRE(dy && u && dx && UR_ALIGNED16(dy) && UR_ALIGNED16(u) && UR_ALIGNED16(dx), "ur_gelu_bwd: null / misaligned");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)u,
                     (bf16_t*)dx, (long)(n / 8));
  UR_CHECK_LAUNCH("ur_gelu_bwd");
  return 0;
}
extern "C" int ur_transpose_bf16(const void* src, void* dst, int32_t rows, int32_t cols, void* stream) {
  UR_REQUIRE(src && dst && rows > 0 && cols > 0, "ur_transpose_bf16: bad argument");
  hipLaunchKernelGGL(transpose_kernel, dim3(ur_cdiv(cols, 32), ur_cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     (bf16_t*)dst, rows, cols);
  UR_CHECK_LAUNCH("ur_transpose_bf16");
  return 0;
}
extern "C" int ur_transpose_bf16_batched(const void* desc, int32_t n, int32_t total_tiles, void* stream) {
  UR_REQUIRE(desc && n > 0 && total_tiles > 0 && (((uintptr_t)desc) & 7) == 0, "ur_transpose_bf16_batched: bad argument");
  static_assert(sizeof(TransposeDesc) == 32, "descriptor layout: two pointers + four int32 (unirec_hip.h)");
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const TransposeDesc*)desc, n);
  UR_CHECK_LAUNCH("ur_transpose_bf16_batched");
  return 0;
}
extern "C" int ur_swiglu_fwd(const void* gu, void* act, int32_t M, int32_t I, void* stream) {
  UR_REQUIRE(M >= 0 && I > 0 && (I % 8) == 0, "ur_swiglu_fwd: I must be a multiple of 8");
  if (M == 0) return 0;
  UR_REQUIRE(gu && act && UR_ALIGNED16(gu) && UR_ALIGNED16(act), "ur_swiglu_fwd: null / misaligned");
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(ew_grid((long)M * (I / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)gu, (bf16_t*)act, (long)M, I / 8, I);
  UR_CHECK_LAUNCH("ur_swiglu_fwd");
  return 0;
}
extern "C" int ur_swiglu_bwd(const void* dact, const void* gu, void* dgu, int32_t M, int32_t I, void* stream) {
  UR_REQUIRE(M >= 0 && I > 0 && (I % 8) == 0, "ur_swiglu_bwd: I must be a multiple of 8");
  if (M == 0) return 0;
  UR_REQUIRE(dact && gu && dgu && UR_ALIGNED16(dact) && UR_ALIGNED16(gu) && UR_ALIGNED16(dgu), "ur_swiglu_bwd: null / misaligned");
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(ew_grid((long)M * (I / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dact, (const bf16_t*)gu, (bf16_t*)dgu, (long)M, I / 8, I);
  UR_CHECK_LAUNCH("ur_swiglu_bwd");
  return 0;
}
extern "C" int ur_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                             void* stream) {
  UR_REQUIRE(n >= 0 && step >= 1, "ur_adamw_step: need n >= 0 and step >= 1");
  if (n == 0) return 0;
  UR_REQUIRE(param && grad && exp_avg && exp_avg_sq && UR_ALIGNED16(param) && UR_ALIGNED16(grad) && UR_ALIGNED16(exp_avg) &&
             UR_ALIGNED16(exp_avg_sq), "ur_adamw_step: null / misaligned");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
  UR_CHECK_LAUNCH("ur_adamw_step");
  return 0;
}

// ---- keep flags of the counter-based dropout (test / inspection entry; include/unirec_hip.h) ----------------------------------
namespace {
__global__ void dropout_keep_kernel(uint64_t seed, uint32_t thr, uint64_t idx0, long n, uint8_t* __restrict__ keep) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    keep[i] = ur_dropout_scale(seed, idx0 + (uint64_t)i, thr, 1.0f) != 0.f ? 1 : 0;
}
}  // namespace

namespace {
__global__ void attn_dropout_keep_kernel(uint64_t seed, uint32_t thr16, uint64_t row0, long nrows, int Sk, uint8_t* __restrict__ keep) {
  const long n = nrows * Sk, stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long row = i / Sk;
    const uint32_t key = (uint32_t)(i - row * Sk);
    const ur_rowkey rk = ur_attn_row_key(seed, row0 + (uint64_t)row);
    keep[i] = ur_attn_keep_scale(ur_attn_pair_word(rk.k1, rk.k2, key >> 1), key, thr16, 1.0f) != 0.f ? 1 : 0;
  }
}
}  // namespace

extern "C" int ur_attn_dropout_keep(uint64_t seed, float p, uint64_t row0, int64_t nrows, int32_t Sk, uint8_t* keep, void* stream) {
  UR_REQUIRE(nrows >= 0 && Sk > 0 && (nrows == 0 || keep != nullptr), "ur_attn_dropout_keep: bad arguments");
  UR_REQUIRE(p >= 0.f && p < 1.f, "ur_attn_dropout_keep: p must be in [0, 1)");
  if (nrows == 0) return 0;
  const uint32_t thr16 = p > 0.f ? (ur_drop_threshold16(p) > 1u ? ur_drop_threshold16(p) : 1u) : 0u;
  hipLaunchKernelGGL(attn_dropout_keep_kernel, dim3(ew_grid(nrows * Sk, 256)), dim3(256), 0, (hipStream_t)stream, seed, thr16, row0, (long)nrows, (int)Sk, keep);
  UR_CHECK_LAUNCH("ur_attn_dropout_keep");
  return 0;
}

extern "C" int ur_dropout_keep(uint64_t seed, float p, uint64_t idx0, int64_t n, uint8_t* keep, void* stream) {
  UR_REQUIRE(n >= 0 && (n == 0 || keep != nullptr), "ur_dropout_keep: null output");
  UR_REQUIRE(p >= 0.f && p < 1.f, "ur_dropout_keep: p must be in [0, 1)");
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_keep_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, seed, p > 0.f ? ur_drop_threshold(p) : 0u, idx0, (long)n, keep);
  UR_CHECK_LAUNCH("ur_dropout_keep");
  return 0;
}

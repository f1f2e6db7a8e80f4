// Data-path kernels either side of the hot path, gfx950 only.  All HBM-bound.
//
//   ur_gather_rows    : out[i] = src[idx[i]] (a zero row for idx < 0): the packed replacement of the reference's
//                       per-sample torch.stack of cached tensors --
//                       training/train_item_individual_token_joint.py:557-577 (_get_history_qformer_inputs: history
//                       slot -> [F,1024] field vectors + [F] mask, zero padding for missing items / empty slots),
//                       :246-255 (history item query tokens), models/qformer_utils.py:150-155 (__getitem__).
//                       f32 rows may leave as bf16 (the dtype the Q-Former kernels consume) in the same pass.
//   ur_catalog_scores : scores[b][n] = cos(user_b, item_n) over a SHARED catalogue [N,D] f32 with
//                       F.normalize(p=2, eps=1e-12) semantics (training/train_item_individual_token_joint.py:408-415,
//                       evaluation over pool = all items); the catalogue is read once per 16 users.
//   ur_rank_of_index  : rank_b = 1 + #{n : s_bn > s_b,gt_b}  (:416-417: position of the positive in the descending
//                       argsort; ties resolved for the positive, as ur_mrr_rank).
#include "common.hip.h"
#include "unirec_hip.h"

namespace {

// one wave per output row; 16-byte pieces
template <int SRC_BYTES, bool TO_BF16>
__global__ void gather_rows_kernel(const char* __restrict__ src, const long* __restrict__ idx, char* __restrict__ out,
                                   long row_elems, long n_out, long n_src) {
  const long row = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (row >= n_out) return;
  const int lane = threadIdx.x & 63;
  const long s = idx[row];
  const bool valid = s >= 0 && s < n_src;
  constexpr int EPP = 16 / SRC_BYTES;                   // source elements per 16-byte piece
  const long pieces = row_elems / EPP;                  // host guarantees divisibility
  const char* sp = src + (valid ? s : 0) * row_elems * SRC_BYTES;
  if (TO_BF16) {                                        // f32 -> bf16: 2 source pieces -> one 16-byte output piece
    char* op = out + row * row_elems * 2;
    for (long p = lane; p < pieces / 2; p += 64) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (valid) { a = *reinterpret_cast<const float4*>(sp + p * 32); b = *reinterpret_cast<const float4*>(sp + p * 32 + 16); }
      *reinterpret_cast<uint4*>(op + p * 16) = make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
    }
  } else {
    char* op = out + row * row_elems * SRC_BYTES;
    for (long p = lane; p < pieces; p += 64) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (valid) v = *reinterpret_cast<const uint4*>(sp + p * 16);
      *reinterpret_cast<uint4*>(op + p * 16) = v;
    }
  }
}
// narrow rows (masks: F bytes per item): one thread per output element
__global__ void gather_bytes_kernel(const uint8_t* __restrict__ src, const long* __restrict__ idx, uint8_t* __restrict__ out,
                                    long row_bytes, long n_out, long n_src) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out * row_bytes) return;
  const long row = i / row_bytes, c = i - row * row_bytes;
  const long s = idx[row];
  out[i] = (s >= 0 && s < n_src) ? src[s * row_bytes + c] : (uint8_t)0;
}

__global__ void row_inv_norm_kernel(const float* __restrict__ x, float* __restrict__ inv, long rows, int D) {
  const long row = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = x + row * D;
  float s = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 v = *reinterpret_cast<const float4*>(p + d);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s = wave_sum(s);
  if (lane == 0) inv[row] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
}

// scores[b][n] = (u_b . c_n) * inv_u[b] * inv_c[n], f32 FMA.  Block = 256 threads = 4 waves; a block owns 16 users
// (their vectors staged once in LDS) and walks catalogue rows, one row per wave per step: each lane holds 4-element
// pieces of the row and accumulates 16 dot products, reduced across the wave at the end of the row.
constexpr int CU_USERS = 16;
__global__ __launch_bounds__(256) void catalog_scores_kernel(const float* __restrict__ user, const float* __restrict__ inv_u,
                                                             const float* __restrict__ cat, const float* __restrict__ inv_c,
                                                             float* __restrict__ scores, int B, long N, int D, long rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float us[];      // [CU_USERS][D]
  const int b0 = blockIdx.y * CU_USERS;
  const int nb = min(CU_USERS, B - b0);
  for (int i = threadIdx.x * 4; i < CU_USERS * D; i += 256 * 4) {
    const int u = i / D, d = i - u * D;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u < nb) v = *reinterpret_cast<const float4*>(user + (long)(b0 + u) * D + d);
    *reinterpret_cast<float4*>(us + i) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  for (long n = r0 + wave; n < r1; n += 4) {
    float acc[CU_USERS];
#pragma unroll
    for (int u = 0; u < CU_USERS; ++u) acc[u] = 0.f;
    const float* cp = cat + n * D;
    for (int d = lane * 4; d < D; d += 256) {
      const float4 c = *reinterpret_cast<const float4*>(cp + d);
#pragma unroll
      for (int u = 0; u < CU_USERS; ++u) {
        const float4 x = *reinterpret_cast<const float4*>(us + u * D + d);
        acc[u] = fmaf(c.x, x.x, fmaf(c.y, x.y, fmaf(c.z, x.z, fmaf(c.w, x.w, acc[u]))));
      }
    }
    const float ic = inv_c[n];
#pragma unroll
    for (int u = 0; u < CU_USERS; ++u) {
      const float s = wave_sum(acc[u]);
      if (lane == 0 && u < nb) scores[(long)(b0 + u) * N + n] = s * inv_u[b0 + u] * ic;
    }
  }
}

__global__ void rank_of_index_kernel(const float* __restrict__ scores, const long* __restrict__ gt, int* __restrict__ rank, long N) {
  const int b = blockIdx.x;
  const float* s = scores + (long)b * N;
  const long g = min(max(gt[b], 0L), N - 1);
  const float ref = s[g];
  int cnt = 0;
  for (long n = threadIdx.x; n < N; n += blockDim.x) cnt += (s[n] > ref) ? 1 : 0;
  __shared__ int part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) part[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) rank[b] = 1 + part[0] + part[1] + part[2] + part[3];
}

// Context MLP, first layer: h1[n][j] = gelu(b1[j] + sum_f W1[j][f] * feat_f(n)), bf16 out.
//   KIND 0: TimestampEncoder features, models/mwne.py:525-565 (9 = secular + 4 sin/cos pairs, all in f32 as the
//           reference computes them from timestamps.float()).
//   KIND 1: GeoCoordinateEncoder features, models/mwne.py:586-607 (lat/lon degrees -> unit-sphere x, y, z).
// The 9- / 3-wide contraction stays on the vector units in f32 (exact features, no padding to an MFMA k-step); the
// second Linear (2H -> H) is an ordinary ur_gemm.
template <int KIND>
__global__ void context_mlp1_kernel(const float* __restrict__ in, const float* __restrict__ W1, const float* __restrict__ b1,
                                    bf16_t* __restrict__ out, long n, int H2) {
  const long row = blockIdx.x;
  if (row >= n) return;
  constexpr int NF = KIND == 0 ? 9 : 3;
  float f[NF];
  if (KIND == 0) {
    const float x = in[row];
    const float year = 31557600.0f, day = 86400.0f, two_pi = 6.283185307179586f;
    f[0] = x / year;
    float r = fmodf(x, day); if (r < 0.f) r += day;              // torch `%` is a floor-mod
    const float day_phase = r / day;
    f[1] = sinf(two_pi * day_phase); f[2] = cosf(two_pi * day_phase);
    const float week_phase = ((x / day) + 4.0f) / 7.0f;
    f[3] = sinf(two_pi * week_phase); f[4] = cosf(two_pi * week_phase);
    float ry = fmodf(x, year); if (ry < 0.f) ry += year;
    const float year_phase = ry / year;
    f[5] = sinf(two_pi * year_phase); f[6] = cosf(two_pi * year_phase);
    const float month_phase = year_phase * 12.0f;
    f[7] = sinf(two_pi * month_phase); f[8] = cosf(two_pi * month_phase);
  } else {
    const float d2r = 0.017453292519943295f;
    const float lat = in[2 * row] * d2r, lon = in[2 * row + 1] * d2r;
    f[0] = cosf(lat) * cosf(lon); f[1] = cosf(lat) * sinf(lon); f[2] = sinf(lat);
  }
  for (int j = threadIdx.x; j < H2; j += blockDim.x) {
    float a = b1[j];
#pragma unroll
    for (int k = 0; k < NF; ++k) a = fmaf(W1[(long)j * NF + k], f[k], a);
    out[row * H2 + j] = f2bf(gelu_erf_f(a));
  }
}

}  // namespace

extern "C" int ur_gather_rows(const void* src, int32_t src_kind, void* out, int32_t out_kind, const int64_t* idx, int64_t row_elems,
                              int64_t n_out, int64_t n_src, void* stream) {
  UR_REQUIRE(n_out >= 0 && n_src >= 0 && row_elems > 0, "ur_gather_rows: bad sizes");
  UR_REQUIRE((src_kind == UR_KIND_U8 || src_kind == UR_KIND_BF16 || src_kind == UR_KIND_F32) &&
             (out_kind == src_kind || (src_kind == UR_KIND_F32 && out_kind == UR_KIND_BF16)),
             "ur_gather_rows: kinds must match, or f32 -> bf16");
  if (n_out == 0) return 0;
  UR_REQUIRE(src && out && idx, "ur_gather_rows: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (src_kind == UR_KIND_U8) {
    const long total = n_out * row_elems;
    hipLaunchKernelGGL(gather_bytes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint8_t*)src, (const long*)idx,
                       (uint8_t*)out, (long)row_elems, (long)n_out, (long)n_src);
  } else {
    UR_REQUIRE(UR_ALIGNED16(src) && UR_ALIGNED16(out), "ur_gather_rows: rows must be 16-byte aligned");
    const dim3 grid((unsigned)((n_out + 3) / 4)), block(256);
    if (src_kind == UR_KIND_BF16) {
      UR_REQUIRE((row_elems % 8) == 0, "ur_gather_rows: bf16 rows must be multiples of 8 elements");
      hipLaunchKernelGGL((gather_rows_kernel<2, false>), grid, block, 0, st, (const char*)src, (const long*)idx, (char*)out, (long)row_elems, (long)n_out, (long)n_src);
    } else if (out_kind == UR_KIND_F32) {
      UR_REQUIRE((row_elems % 4) == 0, "ur_gather_rows: f32 rows must be multiples of 4 elements");
      hipLaunchKernelGGL((gather_rows_kernel<4, false>), grid, block, 0, st, (const char*)src, (const long*)idx, (char*)out, (long)row_elems, (long)n_out, (long)n_src);
    } else {
      UR_REQUIRE((row_elems % 8) == 0, "ur_gather_rows: f32 -> bf16 rows must be multiples of 8 elements");
      hipLaunchKernelGGL((gather_rows_kernel<4, true>), grid, block, 0, st, (const char*)src, (const long*)idx, (char*)out, (long)row_elems, (long)n_out, (long)n_src);
    }
  }
  UR_CHECK_LAUNCH("ur_gather_rows");
  return 0;
}

extern "C" int ur_catalog_scores(const float* user, const float* catalog, float* scores, float* user_inv_norm, float* cat_inv_norm,
                                 int32_t cat_norm_ready, int32_t B, int64_t N, int32_t D, void* stream) {
  UR_REQUIRE(B >= 0 && N >= 0 && D > 0 && (D % 4) == 0 && D <= 2048, "ur_catalog_scores: need D %% 4 == 0 and D <= 2048 (got %d)", D);
  if (B == 0 || N == 0) return 0;
  UR_REQUIRE(user && catalog && scores && user_inv_norm && cat_inv_norm, "ur_catalog_scores: null pointer");
  UR_REQUIRE(UR_ALIGNED16(user) && UR_ALIGNED16(catalog), "ur_catalog_scores: operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(row_inv_norm_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, user, user_inv_norm, (long)B, (int)D);
  if (!cat_norm_ready)
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, catalog, cat_inv_norm, (long)N, (int)D);
  const int ub = ur_cdiv(B, CU_USERS);
  long blocks_x = 2048 / ub;                                  // ~8 workgroups per CU in all
  if (blocks_x < 1) blocks_x = 1;
  long rpb = (N + blocks_x - 1) / blocks_x;
  rpb = (rpb + 3) / 4 * 4;
  blocks_x = (N + rpb - 1) / rpb;
  const size_t smem = (size_t)CU_USERS * D * sizeof(float);
  static std::atomic<uint64_t> attr_set{0};   // per device
  UR_ONCE_PER_DEVICE(attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&catalog_scores_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CU_USERS * 2048 * 4);
    if (e != hipSuccess) UR_FAIL((int)e, "ur_catalog_scores: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(catalog_scores_kernel, dim3((unsigned)blocks_x, (unsigned)ub), dim3(256), smem, st, user, user_inv_norm, catalog,
                     cat_inv_norm, scores, (int)B, (long)N, (int)D, (long)rpb);
  UR_CHECK_LAUNCH("ur_catalog_scores");
  return 0;
}

extern "C" int ur_rank_of_index(const float* scores, const int64_t* gt_index, int32_t* rank, int32_t B, int64_t N, void* stream) {
  UR_REQUIRE(B >= 0 && N > 0, "ur_rank_of_index: bad sizes");
  if (B == 0) return 0;
  UR_REQUIRE(scores && gt_index && rank, "ur_rank_of_index: null pointer");
  hipLaunchKernelGGL(rank_of_index_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, scores, (const long*)gt_index, rank, (long)N);
  UR_CHECK_LAUNCH("ur_rank_of_index");
  return 0;
}

extern "C" int ur_context_mlp1(const float* in, int32_t kind, const float* W1, const float* b1, void* out, int64_t n, int32_t H2, void* stream) {
  UR_REQUIRE((kind == 0 || kind == 1) && n >= 0 && H2 > 0, "ur_context_mlp1: kind must be 0 (timestamp) or 1 (lat/lon), n >= 0");
  if (n == 0) return 0;
  UR_REQUIRE(in && W1 && b1 && out, "ur_context_mlp1: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL((context_mlp1_kernel<0>), dim3((unsigned)n), dim3(256), 0, st, in, W1, b1, (bf16_t*)out, (long)n, (int)H2);
  else hipLaunchKernelGGL((context_mlp1_kernel<1>), dim3((unsigned)n), dim3(256), 0, st, in, W1, b1, (bf16_t*)out, (long)n, (int)H2);
  UR_CHECK_LAUNCH("ur_context_mlp1");
  return 0;
}

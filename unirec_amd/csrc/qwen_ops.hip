// Qwen3-side HBM-bound kernels: RoPE table, fused per-head q/k RMSNorm + RoPE (fwd/bwd), embedding
// gather fused with Q-Former token injection (fwd/bwd), mean pooling (fwd/bwd).  gfx950 only.
#include "common.hip.h"
#include "unirec_hip.h"

namespace {

__device__ __forceinline__ void un8(const uint4& u, float (&f)[8]) {
  f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
  f[4] = bf_lo(u.z); f[5] = bf_hi(u.z); f[6] = bf_lo(u.w); f[7] = bf_hi(u.w);
}
__device__ __forceinline__ uint4 pk8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}
__device__ __forceinline__ void ld8f(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// Lane exchanges inside the 16-lane group that owns a token row of head_dim 128 as DPP modifiers (no LDS crossbar: __shfl_xor
// compiles to ds_bpermute_b32, eight of them per lane and head were a third of the roped backward's time)
template <int CTRL> __device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// the rotate-half partner: LPH / 2 lanes away inside the group
template <int LPH> __device__ __forceinline__ float partner_of(float x) {
  if constexpr (LPH == 16) return dpp_f<0x128>(x);                    // row_ror:8 == lane ^ 8 inside a 16-lane row
  else return __shfl_xor(x, LPH / 2, 64);
}
// sum over the group's lanes, result in every lane
template <int LPH> __device__ __forceinline__ float group_sum_dpp(float x) {
  if constexpr (LPH == 16) {
    x += dpp_f<0xB1>(x);        // quad_perm [1,0,3,2]
    x += dpp_f<0x4E>(x);        // quad_perm [2,3,0,1]
    x += dpp_f<0x141>(x);       // row_half_mirror
    x += dpp_f<0x140>(x);       // row_mirror
    return x;
  } else {
    return group_sum<LPH>(x);
  }
}

// cos/sin [S][hd/2] f32 : inv_freq_i = theta^(-2i/hd), angle = pos * inv_freq_i  (modeling_qwen3.py:107-137)
__global__ void rope_table_kernel(float* __restrict__ cs, float* __restrict__ sn, int S, int half, float theta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * half) return;
  const int pos = i / half, j = i - pos * half;
  const float inv = 1.0f / powf(theta, (float)(2 * j) / (float)(2 * half));
  const float ang = (float)pos * inv;
  cs[i] = cosf(ang);
  sn[i] = sinf(ang);
}

// One token per group of LPH = hd/8 lanes; the group walks the token's nq + nkv heads (0..nq-1 are q, the rest k), so
// the position's cos / sin row and the two norm weight vectors are loaded ONCE per token instead of once per
// (token, head) row -- they were 6 of the 7 loads of a row (table reads 2x the payload bytes).
template <int HD, bool BWD>
__global__ __launch_bounds__(256) void qknorm_rope_kernel(const bf16_t* __restrict__ raw, long ldraw,
                                                          const float* __restrict__ qw, const float* __restrict__ kw,
                                                          const float* __restrict__ cs, const float* __restrict__ sn,
                                                          bf16_t* __restrict__ qo, bf16_t* __restrict__ ko,   // fwd outputs / bwd: dq_out, dk_out inputs
                                                          bf16_t* __restrict__ draw, long lddraw,             // bwd output (q,k sections)
                                                          long M, int S, int nq, int nkv, float eps) {
  constexpr int LPH = HD / 8, HALF = HD / 2;
  const int lane = threadIdx.x & 63;
  const int li = lane % LPH;
  const long toks_per_block = 256 / LPH;
  float wq[8], wk[8];
  ld8f(qw + li * 8, wq);
  ld8f(kw + li * 8, wk);
  const float sign = (li < LPH / 2) ? -1.f : 1.f;       // rotate_half: first half gets -x[d+half]
  // all LPH lanes of a group share the token, so the group shuffles below never see a diverged partner
  for (long m = (long)blockIdx.x * toks_per_block + threadIdx.x / LPH; m < M; m += (long)gridDim.x * toks_per_block) {
    const int pos = (int)(m % S);
    float c[8], s[8];
    ld8f(cs + (long)pos * HALF + (li % (LPH / 2)) * 8, c);
    ld8f(sn + (long)pos * HALF + (li % (LPH / 2)) * 8, s);
    // heads in batches of UH: the batch's loads are issued together (one 16-byte load per lane and head in flight was the
    // kernel's whole memory-level parallelism: 3.3 TB/s on q|k raw + outputs; the reductions below are a dependent chain per head)
#ifndef UR_ROPE_UH
#define UR_ROPE_UH 4            // lab: heads per load batch of the forward (1 = the round-2 kernel)
#endif
    constexpr int UH = BWD ? (UR_ROPE_UH > 1 ? 2 : 1) : UR_ROPE_UH;        // (backward: two loads per head, its operands mostly come from L2 -- and 4 would cost a wave per SIMD)
    const int nh = nq + nkv;
    for (int h0 = 0; h0 < nh; h0 += UH) {
      uint4 xr[UH], dyr[UH];
#pragma unroll
      for (int u = 0; u < UH; ++u) {
        const int hh = min(h0 + u, nh - 1);
        xr[u] = *reinterpret_cast<const uint4*>(raw + m * ldraw + (long)hh * HD + li * 8);
        if (BWD) {
          const bool isq = hh < nq;
          const bf16_t* op = isq ? (qo + m * (long)nq * HD + (long)hh * HD) : (ko + m * (long)nkv * HD + (long)(hh - nq) * HD);
          dyr[u] = *reinterpret_cast<const uint4*>(op + li * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < UH; ++u) {
        const int hh = h0 + u;
        if (hh >= nh) break;
        const bool isq = hh < nq;
        float x[8], ww[8];
        un8(xr[u], x);
#pragma unroll
        for (int e = 0; e < 8; ++e) ww[e] = isq ? wq[e] : wk[e];
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
        ss = group_sum_dpp<LPH>(ss);
        const float rs = rsqrtf(ss / (float)HD + eps);
        bf16_t* op = isq ? (qo + m * (long)nq * HD + (long)hh * HD) : (ko + m * (long)nkv * HD + (long)(hh - nq) * HD);
        if (!BWD) {
          float xn[8], o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) xn[e] = x[e] * rs * ww[e];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float partner = partner_of<LPH>(xn[e]);
            o[e] = xn[e] * c[e] + sign * partner * s[e];
          }
          *reinterpret_cast<uint4*>(op + li * 8) = pk8(o);
        } else {
          float dy[8], g[8], xh[8];
          un8(dyr[u], dy);
          float t = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float partner = partner_of<LPH>(dy[e] * s[e]);   // (dout*sin) of the paired element
            const float dxn = dy[e] * c[e] - sign * partner;               // d<half: +partner, d>=half: -partner
            g[e] = dxn * ww[e];
            xh[e] = x[e] * rs;
            t += g[e] * xh[e];
          }
          t = group_sum_dpp<LPH>(t) / (float)HD;
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rs * (g[e] - xh[e] * t);
          *reinterpret_cast<uint4*>(draw + m * lddraw + (long)hh * HD + li * 8) = pk8(o);
        }
      }
    }
  }
}

// Backward of q/k-norm + RoPE when the forward ran as the q|k|v GEMM's epilogue (ur_gemm_args.qkr_*): the raw projections were
// never stored.  The normalised row is recovered from the ROPED output (the rotation is orthogonal): xn = R^T o, x^ = xn / w,
// and 1 / rms comes from the forward (rstd [M, nq + nkv]).  Same lane layout as qknorm_rope_kernel (LPH lanes per token, 8
// consecutive features each; the rotate-half partner sits LPH / 2 lanes away).  Needs non-zero norm weights (host check).
template <int HD>
__global__ __launch_bounds__(256) void qknorm_rope_bwd_roped_kernel(const bf16_t* __restrict__ dqo, const bf16_t* __restrict__ dko,
                                                                    const bf16_t* __restrict__ qr, long ldqr, const bf16_t* __restrict__ kr, long ldkr,
                                                                    const float* __restrict__ rstd, const float* __restrict__ qw, const float* __restrict__ kw,
                                                                    const float* __restrict__ cs, const float* __restrict__ sn,
                                                                    bf16_t* __restrict__ draw, long lddraw, long M, int S, int nq, int nkv,
                                                                    long rstd_ld, int rstd_h0) {
  constexpr int LPH = HD / 8, HALF = HD / 2;
  const int lane = threadIdx.x & 63;
  const int li = lane % LPH;
  const long toks_per_block = 256 / LPH;
  float wq[8], wk[8], iwq[8], iwk[8];
  ld8f(qw + li * 8, wq);
  ld8f(kw + li * 8, wk);
#pragma unroll
  for (int e = 0; e < 8; ++e) { iwq[e] = 1.0f / wq[e]; iwk[e] = 1.0f / wk[e]; }
  const float sign = (li < LPH / 2) ? -1.f : 1.f;
  const int nh = nq + nkv;
  for (long m = (long)blockIdx.x * toks_per_block + threadIdx.x / LPH; m < M; m += (long)gridDim.x * toks_per_block) {
    const int pos = (int)(m % S);
    float c[8], s[8];
    ld8f(cs + (long)pos * HALF + (li % (LPH / 2)) * 8, c);
    ld8f(sn + (long)pos * HALF + (li % (LPH / 2)) * 8, s);
    constexpr int UH = 2;
    for (int h0 = 0; h0 < nh; h0 += UH) {
      uint4 orr[UH], dyr[UH];
      float rsv[UH];
#pragma unroll
      for (int u = 0; u < UH; ++u) {
        const int hh = min(h0 + u, nh - 1);
        const bool isq = hh < nq;
        orr[u] = *reinterpret_cast<const uint4*>((isq ? qr + m * ldqr + (long)hh * HD : kr + m * ldkr + (long)(hh - nq) * HD) + li * 8);
        dyr[u] = *reinterpret_cast<const uint4*>((isq ? dqo + m * (long)nq * HD + (long)hh * HD : dko + m * (long)nkv * HD + (long)(hh - nq) * HD) + li * 8);
        rsv[u] = rstd[m * rstd_ld + rstd_h0 + hh];
      }
#pragma unroll
      for (int u = 0; u < UH; ++u) {
        const int hh = h0 + u;
        if (hh >= nh) break;
        const bool isq = hh < nq;
        float o[8], dy[8], g[8], xh[8];
        un8(orr[u], o);
        un8(dyr[u], dy);
        const float rs = rsv[u];
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float ww = isq ? wq[e] : wk[e], iw = isq ? iwq[e] : iwk[e];
          const float po = partner_of<LPH>(o[e] * s[e]);            // forward: o = xn c + sign * partner(xn) s  =>  xn = o c - sign * partner(o s)
          xh[e] = (o[e] * c[e] - sign * po) * iw;                           // x^ = x * rstd
          const float pd = partner_of<LPH>(dy[e] * s[e]);
          g[e] = (dy[e] * c[e] - sign * pd) * ww;
          t += g[e] * xh[e];
        }
        t = group_sum_dpp<LPH>(t) / (float)HD;
        float dx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dx[e] = rs * (g[e] - xh[e] * t);
        *reinterpret_cast<uint4*>(draw + m * lddraw + (long)hh * HD + li * 8) = pk8(dx);
      }
    }
  }
}

// out[b][s][:] = special(ids) ? tokens[b][ids - first][:] : embed[ids][:]
__global__ __launch_bounds__(256) void embed_inject_kernel(const bf16_t* __restrict__ embed, const long* __restrict__ ids,
                                                           const bf16_t* __restrict__ tokens, long first, int T,
                                                           bf16_t* __restrict__ out, long rows, int S, int D, long vocab) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D8 = D / 8;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const long id = ids[row];
    const long rel = id - first;
    const bf16_t* src;
    if (tokens != nullptr && rel >= 0 && rel < T) src = tokens + ((row / S) * T + rel) * (long)D;
    else src = embed + (id < 0 ? 0 : (id >= vocab ? vocab - 1 : id)) * (long)D;
    for (int c = lane; c < D8; c += 64)
      *reinterpret_cast<uint4*>(out + row * D + c * 8) = *reinterpret_cast<const uint4*>(src + c * 8);
  }
}

// d_tokens[b][t][:] = sum over positions s with ids[b][s] == first + t of dx[b][s][:]   (0 if absent)
__global__ __launch_bounds__(256) void inject_bwd_kernel(const bf16_t* __restrict__ dx, const long* __restrict__ ids, long first,
                                                         int T, bf16_t* __restrict__ dtok, int S, int D) {
  __shared__ unsigned char match[256];
  const int b = blockIdx.x / T, t = blockIdx.x - b * T;
  const int tid = threadIdx.x;
  const long want = first + t;
  const int D8 = D / 8;
  float acc[2][8];   // up to D = 4096 with 256 threads
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[i][e] = 0.f;
  for (int s0 = 0; s0 < S; s0 += 256) {
    const int s = s0 + tid;
    const int mine = (s < S && ids[(long)b * S + s] == want) ? 1 : 0;
    match[tid] = (unsigned char)mine;
    // (a special token appears once per sample: seven of eight chunks hold no match and skip the 256-step scan -- 545 -> 80 us per launch at C4)
    if (__syncthreads_or(mine) == 0) continue;          // uniform: every thread sees the same result; match[] is rewritten behind the NEXT barrier only
    for (int j = 0; j < 256; ++j) {          // ascending position order: bitwise reproducible
      if (!match[j]) continue;
      const bf16_t* rowp = dx + ((long)b * S + s0 + j) * D;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 256;
        if (c < D8) {
          float f[8];
          un8(*reinterpret_cast<const uint4*>(rowp + c * 8), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[i][e] += f[e];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256;
    if (c < D8) *reinterpret_cast<uint4*>(dtok + ((long)b * T + t) * D + c * 8) = pk8(acc[i]);
  }
}

// mean over the middle axis: x [B][S][D] bf16 -> out [B][D] (f32 and/or bf16)
constexpr int POOL_SLICES = 16;
__global__ void pool_stage1(const bf16_t* __restrict__ x, float* __restrict__ part, int S, int D8, int per) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D8) return;
  const int sl = blockIdx.y, b = blockIdx.z;
  const int s0 = sl * per, s1 = min(S, s0 + per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = s0; s < s1; ++s) {
    float f[8];
    un8(*reinterpret_cast<const uint4*>(x + (((long)b * S + s) * D8 + c) * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
  }
  float* o = part + (((long)b * POOL_SLICES + sl) * D8 + c) * 8;
  *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}
__global__ void pool_stage2(const float* __restrict__ part, float* __restrict__ out32, bf16_t* __restrict__ out16, long BD, int D,
                            float inv) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BD) return;
  const long b = i / D; const int d = (int)(i - b * D);
  float s = 0.f;
  for (int k = 0; k < POOL_SLICES; ++k) s += part[(b * POOL_SLICES + k) * D + d];
  s *= inv;
  if (out32) out32[i] = s;
  if (out16) out16[i] = f2bf(s);
}
// dx[b][s][:] = dout[b][:] * inv   (dout f32 or bf16)
__global__ void pool_bwd_kernel(const float* __restrict__ d32, const bf16_t* __restrict__ d16, bf16_t* __restrict__ dx, long B,
                                int S, int D8, float inv) {
  const long total = B * S * D8, stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % D8); const long b = i / ((long)S * D8);
    float f[8];
    if (d32) ld8f(d32 + (b * D8 + c) * 8, f);
    else un8(*reinterpret_cast<const uint4*>(d16 + (b * D8 + c) * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= inv;
    *reinterpret_cast<uint4*>(dx + i * 8) = pk8(f);
  }
}

// U0 -- user-sequence assembly (models/user_sequence_encoder.py:128-142 + the collate's padding,
// training/user_qformer_training.py:153-161), one pass:
//   out[b][l*Qi + j][:] = tokens[b][l][j][:] + ctx[b][l][:] + PE[l*Qi + j][:]   for l < len[b]
//                         0                                                    for l >= len[b] (padding)
//   mask[b][l*Qi + j]   = l < len[b]
// PE = sinusoidal table over the FLAT index (PositionalEncoding :20-25): even d -> sin(pos * w_d),
// odd d -> cos(pos * w_{d-1}), w_d = exp(-d * ln(1e4) / H).  Dropout (p=0.1, the reference leaves the
// module in train mode) is the same counter-based hash as everywhere else.
__global__ __launch_bounds__(256) void user_seq_kernel(const bf16_t* __restrict__ tok, const bf16_t* __restrict__ ctx,
                                                       const int* __restrict__ lens, bf16_t* __restrict__ out, float* __restrict__ mask,
                                                       int B, int L, int Qi, int H, uint32_t thr, float inv_keep, uint64_t seed, uint64_t idx0) {
  const int H8 = H / 8;
  const long rows = (long)B * L * Qi;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float wstep = -9.210340371976184f / (float)H;          // -ln(10000)/H
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int b = (int)(row / ((long)L * Qi));
    const int pos = (int)(row - (long)b * L * Qi);              // flat position l*Qi + j
    const int l = pos / Qi;
    const bool valid = l < lens[b];
    if (lane == 0) mask[row] = valid ? 1.0f : 0.0f;
    for (int c = lane; c < H8; c += 64) {
      float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (valid) {
        float t[8], cx[8];
        un8(*reinterpret_cast<const uint4*>(tok + row * H + c * 8), t);
        un8(*reinterpret_cast<const uint4*>(ctx + ((long)b * L + l) * H + c * 8), cx);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const int d = c * 8 + e;                               // even index of the (sin, cos) pair
          const float ang = (float)pos * __expf((float)d * wstep);
          o[e] = t[e] + cx[e] + sinf(ang);
          o[e + 1] = t[e + 1] + cx[e + 1] + cosf(ang);
        }
        if (thr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] *= ur_dropout_scale(seed, idx0 + (uint64_t)(row * H + c * 8 + e), thr, inv_keep);
        }
      }
      *reinterpret_cast<uint4*>(out + row * H + c * 8) = pk8(o);
    }
  }
}

inline int grid_cap(long n, int cap) { return (int)(n < 1 ? 1 : (n > cap ? cap : n)); }

}  // namespace

extern "C" int ur_rope_table(float* cos_out, float* sin_out, int32_t S, int32_t head_dim, float theta, void* stream) {
  UR_REQUIRE(cos_out && sin_out && S > 0 && head_dim > 0 && (head_dim % 2) == 0, "ur_rope_table: bad argument");
  const int n = S * (head_dim / 2);
  hipLaunchKernelGGL(rope_table_kernel, dim3(ur_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, cos_out, sin_out, S, head_dim / 2, theta);
  UR_CHECK_LAUNCH("ur_rope_table");
  return 0;
}

static int qk_common_check(const void* raw, int64_t ldraw, const float* qw, const float* kw, const float* c, const float* s,
                           int64_t M, int32_t S, int32_t nq, int32_t nkv, int32_t hd, const char* who) {
  UR_REQUIRE(hd == 64 || hd == 128, "%s: head_dim must be 64 or 128", who);
  UR_REQUIRE(raw && qw && kw && c && s && M >= 0 && S > 0 && nq >= 0 && nkv > 0, "%s: null / bad argument", who);   // (nq == 0: the k heads alone)
  UR_REQUIRE((ldraw % 8) == 0 && ldraw >= (int64_t)(nq + 2 * nkv) * hd && UR_ALIGNED16(raw) && UR_ALIGNED16(qw) && UR_ALIGNED16(kw) &&
             UR_ALIGNED16(c) && UR_ALIGNED16(s), "%s: alignment / stride", who);
  return 0;
}

extern "C" int ur_qknorm_rope_fwd(const void* qkv_raw, int64_t ldraw, const float* q_norm_w, const float* k_norm_w,
                                  const float* cos_tab, const float* sin_tab, void* q_out, void* k_out, int64_t M, int32_t S,
                                  int32_t nq, int32_t nkv, int32_t head_dim, float eps, void* stream) {
  int rc = qk_common_check(qkv_raw, ldraw, q_norm_w, k_norm_w, cos_tab, sin_tab, M, S, nq, nkv, head_dim, "ur_qknorm_rope_fwd");
  if (rc) return rc;
  if (M == 0) return 0;
  UR_REQUIRE(q_out && k_out && UR_ALIGNED16(q_out) && UR_ALIGNED16(k_out), "ur_qknorm_rope_fwd: bad outputs");
  const int tpb = 256 / (head_dim / 8);                    // tokens per workgroup
  const int grid = grid_cap(((long)M + tpb - 1) / tpb, 256 * 16);
  if (head_dim == 128)
    hipLaunchKernelGGL((qknorm_rope_kernel<128, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv_raw, (long)ldraw,
                       q_norm_w, k_norm_w, cos_tab, sin_tab, (bf16_t*)q_out, (bf16_t*)k_out, (bf16_t*)nullptr, 0L, (long)M, S, nq, nkv, eps);
  else
    hipLaunchKernelGGL((qknorm_rope_kernel<64, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv_raw, (long)ldraw,
                       q_norm_w, k_norm_w, cos_tab, sin_tab, (bf16_t*)q_out, (bf16_t*)k_out, (bf16_t*)nullptr, 0L, (long)M, S, nq, nkv, eps);
  UR_CHECK_LAUNCH("ur_qknorm_rope_fwd");
  return 0;
}

extern "C" int ur_qknorm_rope_bwd(const void* dq_out, const void* dk_out, const void* qkv_raw, int64_t ldraw,
                                  const float* q_norm_w, const float* k_norm_w, const float* cos_tab, const float* sin_tab,
                                  void* dqkv_raw, int64_t lddraw, int64_t M, int32_t S, int32_t nq, int32_t nkv,
                                  int32_t head_dim, float eps, void* stream) {
  int rc = qk_common_check(qkv_raw, ldraw, q_norm_w, k_norm_w, cos_tab, sin_tab, M, S, nq, nkv, head_dim, "ur_qknorm_rope_bwd");
  if (rc) return rc;
  if (M == 0) return 0;
  UR_REQUIRE(dq_out && dk_out && dqkv_raw && UR_ALIGNED16(dq_out) && UR_ALIGNED16(dk_out) && UR_ALIGNED16(dqkv_raw) && (lddraw % 8) == 0 &&
             lddraw >= (int64_t)(nq + nkv) * head_dim, "ur_qknorm_rope_bwd: bad gradient buffers");
  const int tpb = 256 / (head_dim / 8);                    // tokens per workgroup
  const int grid = grid_cap(((long)M + tpb - 1) / tpb, 256 * 16);
  if (head_dim == 128)
    hipLaunchKernelGGL((qknorm_rope_kernel<128, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv_raw, (long)ldraw,
                       q_norm_w, k_norm_w, cos_tab, sin_tab, (bf16_t*)dq_out, (bf16_t*)dk_out, (bf16_t*)dqkv_raw, (long)lddraw, (long)M, S, nq, nkv, eps);
  else
    hipLaunchKernelGGL((qknorm_rope_kernel<64, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv_raw, (long)ldraw,
                       q_norm_w, k_norm_w, cos_tab, sin_tab, (bf16_t*)dq_out, (bf16_t*)dk_out, (bf16_t*)dqkv_raw, (long)lddraw, (long)M, S, nq, nkv, eps);
  UR_CHECK_LAUNCH("ur_qknorm_rope_bwd");
  return 0;
}

extern "C" int ur_qknorm_rope_bwd_roped(const void* dq_out, const void* dk_out, const void* q_roped, int64_t ldq, const void* k_roped, int64_t ldk,
                                        const float* rstd, const float* q_norm_w, const float* k_norm_w, const float* cos_tab, const float* sin_tab,
                                        void* dqkv_raw, int64_t lddraw, int64_t M, int32_t S, int32_t nq, int32_t nkv, int32_t head_dim, void* stream) {
  UR_REQUIRE(head_dim == 128, "ur_qknorm_rope_bwd_roped: head_dim must be 128 (the q|k|v epilogue it pairs with)");
  UR_REQUIRE(dq_out && dk_out && q_roped && k_roped && rstd && q_norm_w && k_norm_w && cos_tab && sin_tab && dqkv_raw && M >= 0 && S > 0 && nq > 0 && nkv > 0,
             "ur_qknorm_rope_bwd_roped: null / bad argument");
  UR_REQUIRE((ldq % 8) == 0 && (ldk % 8) == 0 && (lddraw % 8) == 0 && ldq >= (int64_t)nq * head_dim && ldk >= (int64_t)nkv * head_dim &&
             lddraw >= (int64_t)(nq + nkv) * head_dim && UR_ALIGNED16(dq_out) && UR_ALIGNED16(dk_out) && UR_ALIGNED16(q_roped) && UR_ALIGNED16(k_roped) &&
             UR_ALIGNED16(q_norm_w) && UR_ALIGNED16(k_norm_w) && UR_ALIGNED16(cos_tab) && UR_ALIGNED16(sin_tab) && UR_ALIGNED16(dqkv_raw),
             "ur_qknorm_rope_bwd_roped: alignment / stride");
  if (M == 0) return 0;
  const int tpb = 256 / (head_dim / 8);
  const int grid = grid_cap(((long)M + tpb - 1) / tpb, 256 * 16);
  hipLaunchKernelGGL((qknorm_rope_bwd_roped_kernel<128>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dq_out, (const bf16_t*)dk_out,
                     (const bf16_t*)q_roped, (long)ldq, (const bf16_t*)k_roped, (long)ldk, rstd, q_norm_w, k_norm_w, cos_tab, sin_tab,
                     (bf16_t*)dqkv_raw, (long)lddraw, (long)M, S, nq, nkv, (long)(nq + nkv), 0);
  UR_CHECK_LAUNCH("ur_qknorm_rope_bwd_roped");
  return 0;
}

extern "C" int ur_qknorm_rope_bwd_roped_k(const void* dk_out, const void* k_roped, int64_t ldk, const float* rstd, int64_t rstd_ld, int32_t rstd_h0,
                                          const float* k_norm_w, const float* cos_tab, const float* sin_tab, void* dk_raw, int64_t lddraw,
                                          int64_t M, int32_t S, int32_t nkv, int32_t head_dim, void* stream) {
  UR_REQUIRE(head_dim == 128, "ur_qknorm_rope_bwd_roped_k: head_dim must be 128 (the q|k|v epilogue it pairs with)");
  UR_REQUIRE(dk_out && k_roped && rstd && k_norm_w && cos_tab && sin_tab && dk_raw && M >= 0 && S > 0 && nkv > 0 && rstd_h0 >= 0 && rstd_ld >= rstd_h0 + nkv,
             "ur_qknorm_rope_bwd_roped_k: null / bad argument");
  UR_REQUIRE((ldk % 8) == 0 && (lddraw % 8) == 0 && ldk >= (int64_t)nkv * head_dim && lddraw >= (int64_t)nkv * head_dim && UR_ALIGNED16(dk_out) &&
             UR_ALIGNED16(k_roped) && UR_ALIGNED16(k_norm_w) && UR_ALIGNED16(cos_tab) && UR_ALIGNED16(sin_tab) && UR_ALIGNED16(dk_raw),
             "ur_qknorm_rope_bwd_roped_k: alignment / stride");
  if (M == 0) return 0;
  const int tpb = 256 / (head_dim / 8);
  const int grid = grid_cap(((long)M + tpb - 1) / tpb, 256 * 16);
  // (nq = 0: every head of the kernel's walk is a k head; the q operands are never dereferenced)
  hipLaunchKernelGGL((qknorm_rope_bwd_roped_kernel<128>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dk_out, (const bf16_t*)dk_out,
                     (const bf16_t*)k_roped, (long)ldk, (const bf16_t*)k_roped, (long)ldk, rstd, k_norm_w, k_norm_w, cos_tab, sin_tab,
                     (bf16_t*)dk_raw, (long)lddraw, (long)M, S, 0, nkv, (long)rstd_ld, (int)rstd_h0);
  UR_CHECK_LAUNCH("ur_qknorm_rope_bwd_roped_k");
  return 0;
}

extern "C" int ur_embed_inject_fwd(const void* embed, int64_t vocab, const int64_t* input_ids, const void* item_tokens,
                                   int64_t first_special_id, int32_t T, void* out, int32_t B, int32_t S, int32_t D, void* stream) {
  UR_REQUIRE(embed && input_ids && out && vocab > 0 && B >= 0 && S > 0 && D > 0 && (D % 8) == 0 && T >= 0, "ur_embed_inject_fwd: bad argument");
  UR_REQUIRE(UR_ALIGNED16(embed) && UR_ALIGNED16(out) && (!item_tokens || UR_ALIGNED16(item_tokens)), "ur_embed_inject_fwd: alignment");
  if (B == 0) return 0;
  const long rows = (long)B * S;
  hipLaunchKernelGGL(embed_inject_kernel, dim3(grid_cap((rows + 3) / 4, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)embed,
                     (const long*)input_ids, (const bf16_t*)(T > 0 ? item_tokens : nullptr), (long)first_special_id, T, (bf16_t*)out, rows, S, D,
                     (long)vocab);
  UR_CHECK_LAUNCH("ur_embed_inject_fwd");
  return 0;
}

extern "C" int ur_inject_bwd(const void* dx, const int64_t* input_ids, int64_t first_special_id, int32_t T, void* d_item_tokens,
                             int32_t B, int32_t S, int32_t D, void* stream) {
  UR_REQUIRE(dx && input_ids && d_item_tokens && B >= 0 && S > 0 && T >= 0 && D > 0 && (D % 8) == 0 && D <= 4096, "ur_inject_bwd: bad argument");
  UR_REQUIRE(UR_ALIGNED16(dx) && UR_ALIGNED16(d_item_tokens), "ur_inject_bwd: alignment");
  if (B == 0 || T == 0) return 0;
  hipLaunchKernelGGL(inject_bwd_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx, (const long*)input_ids,
                     (long)first_special_id, T, (bf16_t*)d_item_tokens, S, D);
  UR_CHECK_LAUNCH("ur_inject_bwd");
  return 0;
}

extern "C" int64_t ur_mean_pool_workspace_bytes(int32_t B, int32_t D) { return (int64_t)B * POOL_SLICES * D * (int64_t)sizeof(float); }

extern "C" int ur_mean_pool_fwd(const void* x, float* out_f32, void* out_bf16, int32_t B, int32_t S, int32_t D, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  UR_REQUIRE(x && (out_f32 || out_bf16) && B >= 0 && S > 0 && D > 0 && (D % 8) == 0 && UR_ALIGNED16(x), "ur_mean_pool_fwd: bad argument");
  UR_REQUIRE(workspace && UR_ALIGNED16(workspace) && workspace_bytes >= ur_mean_pool_workspace_bytes(B, D), "ur_mean_pool_fwd: workspace too small");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int D8 = D / 8, per = ur_cdiv(S, POOL_SLICES);
  hipLaunchKernelGGL(pool_stage1, dim3(ur_cdiv(D8, 64), POOL_SLICES, B), dim3(64), 0, st, (const bf16_t*)x, (float*)workspace, S, D8, per);
  UR_CHECK_LAUNCH("ur_mean_pool_fwd(stage1)");
  const long BD = (long)B * D;
  hipLaunchKernelGGL(pool_stage2, dim3(ur_cdiv(BD, 256)), dim3(256), 0, st, (const float*)workspace, out_f32, (bf16_t*)out_bf16, BD, D,
                     1.0f / (float)S);
  UR_CHECK_LAUNCH("ur_mean_pool_fwd(stage2)");
  return 0;
}

extern "C" int ur_mean_pool_bwd(const float* dout_f32, const void* dout_bf16, void* dx, int32_t B, int32_t S, int32_t D, void* stream) {
  UR_REQUIRE((dout_f32 || dout_bf16) && dx && B >= 0 && S > 0 && D > 0 && (D % 8) == 0 && UR_ALIGNED16(dx), "ur_mean_pool_bwd: bad argument");
  UR_REQUIRE((!dout_f32 || UR_ALIGNED16(dout_f32)) && (!dout_bf16 || UR_ALIGNED16(dout_bf16)), "ur_mean_pool_bwd: alignment");
  if (B == 0) return 0;
  const long total = (long)B * S * (D / 8);
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(grid_cap((total + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, dout_f32,
                     (const bf16_t*)dout_bf16, (bf16_t*)dx, (long)B, S, D / 8, 1.0f / (float)S);
  UR_CHECK_LAUNCH("ur_mean_pool_bwd");
  return 0;
}

extern "C" int ur_user_sequence_assemble(const void* item_tokens, const void* context, const int32_t* lengths, void* out, float* mask,
                                         int32_t B, int32_t L, int32_t Qi, int32_t H, float dropout_p, uint64_t seed, int64_t drop_batch0,
                                         void* stream) {
  UR_REQUIRE(item_tokens && context && lengths && out && mask && B >= 0 && L > 0 && Qi > 0 && H > 0 && (H % 8) == 0,
             "ur_user_sequence_assemble: bad argument");
  UR_REQUIRE(UR_ALIGNED16(item_tokens) && UR_ALIGNED16(context) && UR_ALIGNED16(out), "ur_user_sequence_assemble: alignment");
  UR_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "ur_user_sequence_assemble: dropout p out of range");
  if (B == 0) return 0;
  const long rows = (long)B * L * Qi;
  hipLaunchKernelGGL(user_seq_kernel, dim3(grid_cap((rows + 3) / 4, 256 * 16)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)item_tokens,
                     (const bf16_t*)context, (const int*)lengths, (bf16_t*)out, mask, B, L, Qi, H,
                     dropout_p > 0.f ? ur_drop_threshold(dropout_p) : 0u, dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f, seed,
                     (uint64_t)drop_batch0 * (uint64_t)L * (uint64_t)Qi * (uint64_t)H);
  UR_CHECK_LAUNCH("ur_user_sequence_assemble");
  return 0;
}

// Shared device/host helpers for the UniRec MI355X (gfx950 / CDNA4) hot-path library.
// Wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef uint16_t bf16_t;   // raw bfloat16 bits (matches torch.bfloat16 storage)

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define UR_WAVE 64

// Lab knobs (tools/lab): experiment switches read from the environment and the kernel branches behind them exist only in builds
// made with -DUR_LAB=1 (tools/lab/lib_variant.sh ... -DUR_LAB=1); the product library compiles them out.
#ifndef UR_LAB
#define UR_LAB 0
#endif
static inline int ur_lab_int(const char* name, int dflt) {
#if UR_LAB
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

// ---- per-device once flags -------------------------------------------------------------------------
// hipFuncSetAttribute and the CU count are properties of (function, DEVICE): a process that drives several GPUs needs them per
// device.  Bit d of the mask = done on device d (the guarded calls are idempotent: a race only repeats one; UR_ONCE_PER_DEVICE).
static inline int ur_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev;
}
// UR_ONCE_PER_DEVICE(mask) { body }: the body (a hipFuncSetAttribute) runs while bit d of the mask is clear, and the bit is set only
// AFTER the body has completed: a second thread on the same device (PyTorch issues backward launches from its autograd thread) either
// sees the bit -- then the attribute is applied -- or repeats the idempotent call; a body that leaves early (failure: `return rc`)
// leaves the bit clear, so the next call tries again instead of launching without the attribute.
struct ur_once_scope {
  std::atomic<uint64_t>& mask;
  uint64_t bit;
  bool todo;
  explicit ur_once_scope(std::atomic<uint64_t>& m) : mask(m), bit(1ull << (ur_current_device() & 63)), todo(!(m.load(std::memory_order_acquire) & bit)) {}
  bool pending() const { return todo; }
  void done() { mask.fetch_or(bit, std::memory_order_release); todo = false; }
};
#define UR_ONCE_PER_DEVICE(mask) for (ur_once_scope ur_once_(mask); ur_once_.pending(); ur_once_.done())
static inline int ur_device_cu_count() {
  static std::atomic<int> cached[64];
  const int dev = ur_current_device() & 63;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// ---- error plumbing (C ABI: 0 ok, <0 invalid argument, >0 HIP error code) -------------------
extern thread_local char g_ur_err[512];
#define UR_FAIL(code, ...)                                   \
  do {                                                       \
    snprintf(g_ur_err, sizeof(g_ur_err), __VA_ARGS__);       \
    return (code);                                           \
  } while (0)
#define UR_REQUIRE(cond, ...)                                \
  do {                                                       \
    if (!(cond)) UR_FAIL(-1, __VA_ARGS__);                   \
  } while (0)
#define UR_CHECK_LAUNCH(name)                                                        \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) UR_FAIL((int)e_, "%s: launch failed: %s", name, hipGetErrorString(e_)); \
  } while (0)
#define UR_ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)

// ---- bf16 <-> f32 ---------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  f2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t));
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- transposed LDS reads (ds_read_b64_tr_b16) as inline asm -----------------------------------
// hipcc (ROCm 7.2) drains every in-flight LDS-DMA (s_waitcnt vmcnt(0)) in front of the
// __builtin_amdgcn_ds_read_tr16_b64 builtin, which kills the prefetch ring.  The asm form is invisible
// to that pass.  Each statement issues all its reads, then waits lgkmcnt(0) itself, so its outputs are
// valid when the statement ends (cdna_hip_programming.md §5.7 form (i)).  EXEC must be all ones.
__device__ __forceinline__ uint32_t lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}
// N fragments (N = 1, 2 or 4), each = two 4-row transposed reads at LDS byte offsets a[i], b[i]
__device__ __forceinline__ void tr_read(bf16x8 (&f)[4], const uint32_t (&a)[4], const uint32_t (&b)[4]) {
  bf16x4 l0, h0, l1, h1, l2, h2, l3, h3;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\t"
      "ds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
      "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\t"
      "ds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
      : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]), "v"(a[2]), "v"(b[2]), "v"(a[3]), "v"(b[3]));
  f[0] = cat4(l0, h0); f[1] = cat4(l1, h1); f[2] = cat4(l2, h2); f[3] = cat4(l3, h3);
}
__device__ __forceinline__ void tr_read(bf16x8 (&f)[2], const uint32_t (&a)[2], const uint32_t (&b)[2]) {
  bf16x4 l0, h0, l1, h1;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\t"
      "ds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1)
      : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]));
  f[0] = cat4(l0, h0); f[1] = cat4(l1, h1);
}

// ---- wave / block reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the lanes that share (lane / W) for power-of-two W <= 64
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- counter-based RNG for dropout (mask is a pure function of (seed, element index), so the
// backward regenerates it instead of storing it; independent of grid shape and rank count) ------
// 32-bit decision word of element idx: a keyed murmur3 finaliser over the low counter word -- three 32-bit multiplies per
// element where the splitmix64 round it replaces cost three 64-bit ones (ten quarter-rate v_mul per probability: the
// user Q-Former's attention kernels spent half their time there).  The seed enters three times (xor into the counter,
// add after the first multiply, xor between the two mixing rounds), so the streams of two seeds are neither shifted nor
// xor-permuted copies of each other; the high counter word (element index >= 2^32) moves the additive key.
__device__ __forceinline__ uint32_t ur_hash2(uint64_t seed, uint64_t idx) {
  const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  const uint32_t k2 = (s0 * 0x7FEB352Du) ^ ((s1 << 13) | (s1 >> 19)) ^ 0x5851F42Du;      // seed only: scalar unit
  uint32_t h = ((uint32_t)idx ^ s0) * 0x9E3779B1u + ((uint32_t)(idx >> 32) + s1);
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= k2;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
// Full 64-bit mix of the same generator: four independent 16-bit fields per element (LoRA dropout:
// one field per adapter that shares an input; keep iff field >= p * 65536).
__device__ __forceinline__ uint64_t ur_hash64(uint64_t seed, uint64_t idx) {
  uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
__device__ __forceinline__ bool ur_keep16(uint64_t seed, uint64_t idx, int slot, uint32_t thr16) {
  return (uint32_t)((ur_hash64(seed, idx) >> (16 * slot)) & 0xffffu) >= thr16;
}
static inline uint32_t ur_drop_threshold16(float p) {
  double t = (double)p * 65536.0 + 0.5;
  if (t < 0) t = 0;
  if (t > 65535.0) t = 65535.0;
  return (uint32_t)t;
}
// keep-probability test: returns scale (1/(1-p)) if kept, 0 if dropped.  thr = p * 2^32.
__device__ __forceinline__ float ur_dropout_scale(uint64_t seed, uint64_t idx, uint32_t thr, float inv_keep) {
  return (ur_hash2(seed, idx) >= thr) ? inv_keep : 0.0f;
}
// ---- attention-probability dropout: one 32-bit word per PAIR of keys ----------------------------------------------------------
// The attention kernels draw one decision per (query row, key).  With ur_hash2 per element that was three quarter-rate 32-bit
// multiplies + ~12 vector instructions per probability -- more than the softmax itself (the C3 cross-attention forward / dQ launches
// ran 611 / 644 us with dropout against 417 / 450 us without).  Here a ROW (global index R = ((batch0 + b) * heads + h) * Sq + q) owns
// two 32-bit keys (two ur_hash2 draws, once per row: per lane and kernel where the lane is the query, by the dQ kernel into the
// backward's workspace where the lane is the key), and the word of key pair kp = key >> 1 is a two-multiply finaliser over kp ^ k1
// with k2 added between the rounds (the row enters twice: two rows whose first keys differ only in low bits do not share a shifted
// stream).  Its halves are the 16-bit decision fields of keys 2 kp and 2 kp + 1: keep iff field >= p * 65536.
// oracle/dropout_ref.py restates it; ur_attn_dropout_keep exports the flags.
struct ur_rowkey { uint32_t k1, k2; };
__device__ __forceinline__ ur_rowkey ur_attn_row_key(uint64_t seed, uint64_t row) {
  ur_rowkey r;
  r.k1 = ur_hash2(seed, 2ull * row);
  r.k2 = ur_hash2(seed, 2ull * row + 1ull);
  return r;
}
__device__ __forceinline__ uint32_t ur_attn_pair_word(uint32_t k1, uint32_t k2, uint32_t kp) {
  uint32_t x = kp ^ k1;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x += k2;
  x ^= x >> 15; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
// scale (1 / (1 - p)) of key `key` if kept, 0 if dropped; `word` = ur_attn_pair_word(k1, k2, key >> 1)
__device__ __forceinline__ float ur_attn_keep_scale(uint32_t word, uint32_t key, uint32_t thr16, float inv_keep) {
  const uint32_t f = (key & 1u) ? (word >> 16) : (word & 0xffffu);
  return f >= thr16 ? inv_keep : 0.0f;
}
static inline uint32_t ur_drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

// ---- math -------------------------------------------------------------------------------------
// Phi(x) = 0.5 erfc(-x / sqrt 2), branch-free: erfc(t) = exp(-t g(t)) with g a degree-7 fit of -ln(erfc(t)) / t on [0, 4], continued
// with the slope of -ln erfc at 4 (so Phi -> 0 / 1 in the tails instead of stopping at erfc(4) / 2).  13 vector instructions and one
// v_exp_f32 against ~50 for 1 + erff(x / sqrt 2) (both of erff's branches run in a wave with mixed |x|): the Q-Former's
// FFN launches with a GELU epilogue (M 8192 / 16384, N 3072, K 768) went 90 -> 68 us and 126 -> 97 us.  Exhaustively over all bf16 inputs (tools/lab/gelu_fit.py): gelu(x)
// rounds to a different bf16 than the exact value for 225 inputs (1 + erff: 197), every one of them a tail value below 1e-6 in
// magnitude or a 1-ulp rounding tie; |gelu'(x) error| <= 7e-7.  Unlike 1 + erf, the negative side keeps its RELATIVE accuracy.
// Every GELU path (stand-alone kernels, both GEMM kernels' epilogues) uses these helpers, so they stay bit-identical to one another.
__device__ __forceinline__ float norm_cdf_f(float x) {
  const float tu = fabsf(x) * 0.70710678118654752f;
  const float t = fminf(tu, 4.0f);
  float g = 4.092421022505732e-06f;
  g = fmaf(g, t, -4.843266651732847e-05f);
  g = fmaf(g, t, 4.698846532846801e-05f);
  g = fmaf(g, t, 0.00241068028844893f);
  g = fmaf(g, t, -0.021436134353280067f);
  g = fmaf(g, t, 0.10379933565855026f);
  g = fmaf(g, t, 0.6364415287971497f);
  g = fmaf(g, t, 1.1283843517303467f);
  const float ex = fmaf(tu - t, 8.22f, g * t);
  const float e = 0.5f * __builtin_amdgcn_exp2f(-1.4426950408889634f * ex);
  return x < 0.f ? e : 1.0f - e;
}
__device__ __forceinline__ float gelu_erf_f(float x) { return x * norm_cdf_f(x); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
  const float pdf = 0.39894228040143268f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
  return norm_cdf_f(x) + x * pdf;
}
// sigmoid through the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the IEEE division sequence (v_div_scale x2, v_rcp,
// four fused multiply-adds, v_div_fmas, v_div_fixup: ~10 of the 21 vector instructions per element of the SwiGLU-backward
// GEMM epilogue, which is bound by exactly that arithmetic).  Every SwiGLU path (stand-alone kernels, both GEMM kernels'
// epilogues, the fused SwiGLU + adapter kernel) uses these two helpers, so they stay bit-identical to one another.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }

static inline int ur_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

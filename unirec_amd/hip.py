"""Tensor-level wrappers over the C ABI (include/unirec_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every computation is a
call into libunirec_hip.so with raw ``data_ptr()``s.  All functions require CUDA(HIP) tensors and
raise on anything else -- there is no eager fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import AttnArgs, AttnBwdArgs, GemmArgs, LoraArgs, check

BF16 = torch.bfloat16
F32 = torch.float32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream of the current device as an integer handle.  (torch.cuda.current_stream() builds a Stream object per
    call: 2.7 us, on every one of the ~870 launches of an item-stage step; the raw query is 0.3 us.)"""
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _need(t, dtype, name):
    if not t.is_cuda:
        raise _lib.UniRecHipError(f"{name}: expected a device tensor (the UniRec HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")


_ws_cache = {}

# Optional live kernel timing (bench.py roofline leg): when PROFILE is a list, every ur_gemm launch is
# bracketed by HIP events recorded on the launching (current) stream and logged with its shape.
PROFILE = None
PROFILE_STREAM = None   # bench.py: list of (event0, event1, family, algorithmic bytes) around the HBM-bound launches wrapped by _stream_family
PROFILE_ATTN = None     # bench.py: list of (event0, event1, 'fwd' | 'bwd', B, Sq, Sk, nq, head_dim, causal) around every attention launch


def _stream_family(family, nbytes):
    """Decorator for a wrapper of an HBM-bound launch: when PROFILE_STREAM is a list (bench.py's roofline leg) HIP events on the
    launching stream bracket the call and (events, family, ALGORITHMIC bytes = every operand read / written once) is appended.
    nbytes(result, *args, **kwargs) -> bytes."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            if PROFILE_STREAM is None:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            PROFILE_STREAM.append((e0, e1, family, int(nbytes(r, *a, **k))))
            return r
        return wrapped
    return deco


def _nb(*ts):
    return sum(t.numel() * t.element_size() for t in ts if t is not None and hasattr(t, "numel"))


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (device, tag, STREAM); owned by the caller side (torch allocator).  Per stream since round 6: launches of
    one family run on two streams at once (the Q-Formers' weight-gradient reductions on their side stream beside the dX chain), and a scratch
    buffer shared across streams would be a race (the C3 full-size determinism test caught exactly that)."""
    key = (device, tag, _stream() if device.type == "cuda" else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ------------------------------------------------------------------------------------------------
def gemm(R, S, *, r_kcontig=True, s_kcontig=True, M=None, N=None, K=None, out=None, out_f32=False, alpha=1.0,
         bias=None, residual=None, gelu_out=None, gelu_grad_aux=None, R2=None, S2=None, split_k=1, drop=None,
         swiglu_bwd=None, swiglu_fwd=None, swiglu_paired=None):
    """C[M,N] = alpha*(R(m,k) S(n,k) + R2 S2) + epilogue.  R/S are 2-D bf16 (row stride = stride(0)).
    drop = (bits, p, rank): masked LoRA epilogue (bits = hip.lora_dropout_bits planes of the adapters' shared input),
    swiglu_bwd = (gu, dgu): the result is d(act) of SwiGLU; dgate | dup are written to dgu [M, 2N] and no C is produced
    (returns dgu).  swiglu_fwd = (gate, act): C is up(x) as usual and act[M, N] = silu(gate) * C is written beside it.
    swiglu_paired = act [M, N/2]: the merged gate|up projection with 128-row interleaved weight rows (swiglu_pair_rows); C is
    gate | up in the standard order and act = silu(gate) * up leaves beside it (persistent kernel only: gemm_swiglu_paired_supported).
    See ur_gemm_args in include/unirec_hip.h."""
    lib = _lib.load()
    for t, n in ((R, "R"), (S, "S")):
        if t.dtype != BF16 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"gemm: {n} must be a 2-D bf16 device tensor with unit inner stride")
    if M is None:
        M = R.shape[0] if r_kcontig else R.shape[1]
    if N is None:
        N = S.shape[0] if s_kcontig else S.shape[1]
    if K is None:
        K = R.shape[1] if r_kcontig else R.shape[0]
    Ks = S.shape[1] if s_kcontig else S.shape[0]
    if Ks != K:
        raise ValueError(f"gemm: K mismatch R:{K} S:{Ks}")
    if swiglu_bwd is not None:
        gu_, dgu_ = swiglu_bwd
        for t, n in ((gu_, "gu"), (dgu_, "dgu")):
            if t.dtype != BF16 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1 or t.shape[0] != M or t.shape[1] != 2 * N:
                raise ValueError(f"gemm: swiglu_bwd {n} must be a bf16 [M, 2N] device tensor with unit inner stride")
        out = dgu_            # C is not written in this mode; any valid bf16 pointer with ldc >= N
    if out is None:
        out = torch.empty((M, N), dtype=F32 if out_f32 else BF16, device=R.device)
    a = GemmArgs()
    a.R, a.ldr, a.r_kcontig = R.data_ptr(), R.stride(0), int(r_kcontig)
    a.S, a.lds, a.s_kcontig = S.data_ptr(), S.stride(0), int(s_kcontig)
    a.K = K
    if R2 is not None:
        a.R2, a.ldr2, a.S2, a.lds2 = R2.data_ptr(), R2.stride(0), S2.data_ptr(), S2.stride(0)
        a.K2 = R2.shape[1] if r_kcontig else R2.shape[0]
    a.C, a.ldc, a.c_f32 = out.data_ptr(), out.stride(0), int(out.dtype == F32)
    a.M, a.N, a.alpha = M, N, float(alpha)
    a.bias = _p(bias)
    if residual is not None:
        a.residual, a.ldres = residual.data_ptr(), residual.stride(0)
    if gelu_out is not None:
        a.gelu_out, a.ldg = gelu_out.data_ptr(), gelu_out.stride(0)
    if gelu_grad_aux is not None:
        a.gelu_grad_aux, a.ldaux = gelu_grad_aux.data_ptr(), gelu_grad_aux.stride(0)
    a.split_k = int(split_k)
    if swiglu_bwd is not None:
        a.swiglu_gu, a.swiglu_ldgu = gu_.data_ptr(), gu_.stride(0)
        a.swiglu_dgu, a.swiglu_lddgu, a.swiglu_I = dgu_.data_ptr(), dgu_.stride(0), N
    if swiglu_fwd is not None:
        gate_, act_ = swiglu_fwd
        for t, n in ((gate_, "gate"), (act_, "act")):
            if t.dtype != BF16 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1 or t.shape[0] != M or t.shape[1] != N:
                raise ValueError(f"gemm: swiglu_fwd {n} must be a bf16 [M, N] device tensor with unit inner stride")
        a.swiglu_gate, a.swiglu_ldgate, a.swiglu_act, a.swiglu_ldact = gate_.data_ptr(), gate_.stride(0), act_.data_ptr(), act_.stride(0)
    if swiglu_paired is not None:
        act_ = swiglu_paired
        if act_.dtype != BF16 or not act_.is_cuda or act_.dim() != 2 or act_.stride(1) != 1 or act_.shape[0] != M or act_.shape[1] * 2 != N:
            raise ValueError("gemm: swiglu_paired act must be a bf16 [M, N/2] device tensor with unit inner stride")
        a.swp_act, a.swp_ldact, a.swp_I = act_.data_ptr(), act_.stride(0), N // 2
    if drop is not None:
        bits, pdrop, rank = drop
        a.drop_bits, a.drop_bits_ld, a.drop_bits_stride = bits.data_ptr(), bits.stride(1), bits.stride(0)
        a.drop_p, a.drop_rank = float(pdrop), int(rank)
    ws, wsb = 0, 0
    if split_k > 1:
        wsb = lib.ur_gemm_workspace_bytes(ctypes.byref(a))
        ws = workspace(wsb, R.device, "gemm").data_ptr()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ur_gemm(ctypes.byref(a), ws, wsb, _stream()), "ur_gemm")
        e1.record()
        PROFILE.append((e0, e1, int(r_kcontig), int(s_kcontig), int(out.dtype == F32), M, N, K + int(a.K2), int(split_k),
                        1 if swiglu_bwd is not None else (2 if swiglu_fwd is not None else (4 if swiglu_paired is not None else 0))))
        return out
    check(lib.ur_gemm(ctypes.byref(a), ws, wsb, _stream()), "ur_gemm")
    return out


def qkrope_perm(head_dim=128):
    """Feature stored at tile column c of a head under the paired row order of the fused q|k|v + q/k-norm + RoPE launch
    (ur_qkrope_perm, include/unirec_hip.h): index tensor `perm` with W_paired[h * 128 + c] = W[h * 128 + perm[c]]."""
    if head_dim != 128:
        raise ValueError("the fused q/k-norm + RoPE epilogue is built for head_dim 128")
    lib = _lib.load()
    return torch.tensor([lib.ur_qkrope_perm(c) for c in range(128)], dtype=torch.long)


def _qkrope_args(R, S, R2, S2, qw, kw, cos, sin, Sseq, nq_cols, nk_cols, eps, outs):
    M, N, K = R.shape[0], S.shape[0], R.shape[1]
    a = GemmArgs()
    a.R, a.ldr, a.r_kcontig = R.data_ptr(), R.stride(0), 1
    a.S, a.lds, a.s_kcontig = S.data_ptr(), S.stride(0), 1
    a.K, a.M, a.N, a.alpha, a.split_k = K, M, N, 1.0, 1
    if R2 is not None:
        a.R2, a.ldr2, a.S2, a.lds2, a.K2 = R2.data_ptr(), R2.stride(0), S2.data_ptr(), S2.stride(0), R2.shape[1]
    q, k, v, rstd = outs
    a.C, a.ldc, a.c_f32 = q.data_ptr(), q.stride(0), 0                # not written in this mode: any valid pointer
    a.qkr_q, a.qkr_ldq, a.qkr_k, a.qkr_ldk, a.qkr_v, a.qkr_ldv = q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0)
    a.qkr_rstd = rstd.data_ptr()
    a.qkr_qw, a.qkr_kw, a.qkr_cos, a.qkr_sin = qw.data_ptr(), kw.data_ptr(), cos.data_ptr(), sin.data_ptr()
    a.qkr_S, a.qkr_nq_cols, a.qkr_nk_cols, a.qkr_eps = int(Sseq), int(nq_cols), int(nk_cols), float(eps)
    return a


def swiglu_pair_rows(I):
    """Row order of the merged gate|up weight (and of its LoRA B operand) for the paired SwiGLU epilogue: blocks of 128 gate rows
    alternate with the 128 up rows of the same features -- index tensor `rows` with W_paired = W[rows], W = [gate; up] [2 I, K]."""
    t = torch.arange(I // 128)[:, None, None] * 128 + torch.arange(128)[None, None, :]          # [I/128, 1, 128]
    return torch.cat([t, t + I], dim=1).reshape(-1)


GEMM_MAX_GROUPS = 8


def gemm_grouped(products, split_k=1):
    """ur_gemm_grouped: products = [(R [K, M], S [K, N], out f32 [M, N]), ...] (1..8 token-major pairs: out_i = R_i^T S_i over the shared
    token axis K) as ONE launch -- the weight gradients of one Q-Former layer.  Every out_i is written (split_k > 1: through the
    per-stream workspace and one grouped reduction)."""
    lib = _lib.load()
    n = len(products)
    if not 1 <= n <= GEMM_MAX_GROUPS:
        raise ValueError(f"gemm_grouped: 1 .. {GEMM_MAX_GROUPS} products per launch (got {n})")
    arr = (GemmArgs * n)()
    for a, (R, S, out) in zip(arr, products):
        for t_, nm in ((R, "R"), (S, "S")):
            if t_.dtype != BF16 or not t_.is_cuda or t_.dim() != 2 or t_.stride(1) != 1:
                raise ValueError(f"gemm_grouped: {nm} must be a 2-D bf16 device tensor with unit inner stride")
        if R.shape[0] != S.shape[0]:
            raise ValueError(f"gemm_grouped: token counts differ ({R.shape[0]} vs {S.shape[0]})")
        _need(out, F32, "out")
        if out.dim() != 2 or tuple(out.shape) != (R.shape[1], S.shape[1]) or out.stride(1) != 1:
            raise ValueError(f"gemm_grouped: out must be f32 [{R.shape[1]}, {S.shape[1]}], got {tuple(out.shape)}")
        a.R, a.ldr, a.r_kcontig = R.data_ptr(), R.stride(0), 0
        a.S, a.lds, a.s_kcontig = S.data_ptr(), S.stride(0), 0
        a.K, a.M, a.N, a.alpha = R.shape[0], R.shape[1], S.shape[1], 1.0
        a.C, a.ldc, a.c_f32 = out.data_ptr(), out.stride(0), 1
        a.split_k = int(split_k)
    ws, wsb = 0, 0
    if split_k > 1:
        wsb = lib.ur_gemm_grouped_workspace_bytes(arr, n)
        ws = workspace(wsb, products[0][0].device, "gemm_grouped").data_ptr()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ur_gemm_grouped(arr, n, ws, wsb, _stream()), "ur_gemm_grouped")
        e1.record()
        # (one entry for the launch: M = the products' output rows summed at the widest N -- FLOP-exact only for equal N; see flops below)
        fl = sum(2.0 * R.shape[0] * R.shape[1] * S.shape[1] for R, S, _ in products)
        K = products[0][0].shape[0]
        N = max(S.shape[1] for _, S, _ in products)
        PROFILE.append((e0, e1, 0, 0, 1, int(round(fl / (2.0 * K * N))), N, K, int(split_k), 0))
        return
    check(lib.ur_gemm_grouped(arr, n, ws, wsb, _stream()), "ur_gemm_grouped")


def gemm_swiglu_paired_supported(M, I, K, K2, device):
    """True when the merged gate|up projection of these sizes can carry the SwiGLU forward in its epilogue."""
    lib = _lib.load()
    d = torch.empty(16, dtype=BF16, device=device)
    a = GemmArgs()
    a.R = a.S = a.C = a.swp_act = d.data_ptr()
    a.ldr = a.lds = K
    a.r_kcontig = a.s_kcontig = 1
    a.K, a.M, a.N, a.alpha, a.split_k, a.ldc = K, M, 2 * I, 1.0, 1, 2 * I
    if K2:
        a.R2 = a.S2 = d.data_ptr(); a.ldr2 = a.lds2 = K2; a.K2 = K2
    a.swp_ldact, a.swp_I = I, I
    return bool(lib.ur_gemm_swiglu_paired_supported(ctypes.byref(a)))


def gemm_qkrope_supported(M, N, K, K2, Sseq, nq_cols, nk_cols, device):
    """True when the q|k|v projection of these sizes can carry q/k-norm + RoPE in its epilogue (ur_gemm_qkrope_supported)."""
    lib = _lib.load()
    d = torch.empty(16, dtype=BF16, device=device)          # any valid 16-byte aligned pointer: only sizes / flags are inspected
    f = torch.empty(4, dtype=F32, device=device)
    a = GemmArgs()
    a.R = a.S = a.C = a.qkr_q = a.qkr_k = a.qkr_v = d.data_ptr()
    a.qkr_rstd = a.qkr_qw = a.qkr_kw = a.qkr_cos = a.qkr_sin = f.data_ptr()
    a.ldr = a.lds = K
    a.r_kcontig = a.s_kcontig = 1
    a.K, a.M, a.N, a.alpha, a.split_k, a.ldc = K, M, N, 1.0, 1, N
    if K2:
        a.R2 = a.S2 = d.data_ptr(); a.ldr2 = a.lds2 = K2; a.K2 = K2
    a.qkr_ldq, a.qkr_ldk, a.qkr_ldv = nq_cols, nk_cols, N - nq_cols - nk_cols
    a.qkr_S, a.qkr_nq_cols, a.qkr_nk_cols, a.qkr_eps = int(Sseq), int(nq_cols), int(nk_cols), 1e-6
    return bool(lib.ur_gemm_qkrope_supported(ctypes.byref(a)))


def gemm_qkv_rope(R, S, qw, kw, cos, sin, Sseq, nq_cols, nk_cols, eps, R2=None, S2=None):
    """The merged q|k|v projection with q/k-norm + RoPE in its epilogue (ur_gemm_args.qkr_*): R [M,K] activations, S [N,K]
    weights whose q / k head rows are in the paired order (qkrope_perm), optional LoRA pair (S2 rows paired likewise).
    Returns (q_r [M,nq_cols], k_r [M,nk_cols], v [M,N-nq-nk], rstd [M,(nq+nk)/128]); the raw q, k are never stored."""
    lib = _lib.load()
    M, N = R.shape[0], S.shape[0]
    dev = R.device
    outs = (torch.empty((M, nq_cols), dtype=BF16, device=dev), torch.empty((M, nk_cols), dtype=BF16, device=dev),
            torch.empty((M, N - nq_cols - nk_cols), dtype=BF16, device=dev), torch.empty((M, (nq_cols + nk_cols) // 128), dtype=F32, device=dev))
    a = _qkrope_args(R, S, R2, S2, qw, kw, cos, sin, Sseq, nq_cols, nk_cols, eps, outs)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ur_gemm(ctypes.byref(a), 0, 0, _stream()), "ur_gemm(qkrope)")
        e1.record()
        PROFILE.append((e0, e1, 1, 1, 0, M, N, R.shape[1] + int(a.K2), 1, 3))
        return outs
    check(lib.ur_gemm(ctypes.byref(a), 0, 0, _stream()), "ur_gemm(qkrope)")
    return outs


@_stream_family("qknorm_rope_bwd", lambda r, dq_out, dk_out, q_r, k_r, rstd, qw, kw, cos, sin, dqkv_raw, S, nq, nkv, hd:
                3 * (_nb(k_r) if dq_out is None else _nb(q_r, k_r)))      # gradient in, roped output in, raw gradient out
def qknorm_rope_bwd_roped(dq_out, dk_out, q_r, k_r, rstd, qw, kw, cos, sin, dqkv_raw, S, nq, nkv, hd):
    """Backward of the fused q|k|v epilogue: dq|dk raw into dqkv_raw[:, :(nq+nkv)*hd] from the ROPED forward outputs + rstd."""
    lib = _lib.load()
    M = q_r.shape[0]
    check(lib.ur_qknorm_rope_bwd_roped(dq_out.data_ptr(), dk_out.data_ptr(), q_r.data_ptr(), q_r.stride(0), k_r.data_ptr(), k_r.stride(0),
                                       rstd.data_ptr(), qw.data_ptr(), kw.data_ptr(), cos.data_ptr(), sin.data_ptr(), dqkv_raw.data_ptr(),
                                       dqkv_raw.stride(0), M, S, nq, nkv, hd, _stream()), "ur_qknorm_rope_bwd_roped")
    return dqkv_raw


@_stream_family("qknorm_rope_bwd", lambda r, dk_out, k_r, rstd, h0, kw, cos, sin, dk_raw, S, nkv, hd: 3 * _nb(dk_out))      # gradient in, roped k in, raw gradient out
def qknorm_rope_bwd_roped_k(dk_out, k_r, rstd, h0, kw, cos, sin, dk_raw, S, nkv, hd):
    """The k heads of qknorm_rope_bwd_roped alone (the q heads rode in the dQ kernel: attn_bwd(..., rope_rstd=...)): dk_raw [M, >= nkv*hd]
    view receives the gradient of the raw k projection; 1 / rms of (m, k head h) = rstd[m, h0 + h]."""
    lib = _lib.load()
    M = k_r.shape[0]
    check(lib.ur_qknorm_rope_bwd_roped_k(dk_out.data_ptr(), k_r.data_ptr(), k_r.stride(0), rstd.data_ptr(), rstd.stride(0), int(h0), kw.data_ptr(),
                                         cos.data_ptr(), sin.data_ptr(), dk_raw.data_ptr(), dk_raw.stride(0), M, S, nkv, hd, _stream()),
          "ur_qknorm_rope_bwd_roped_k")
    return dk_raw


ATTN_MODE_TINY, ATTN_MODE_C128, ATTN_MODE_DKV_PERSIST, ATTN_MODE_FEWQ = 0, 1, 2, 3      # include/unirec_hip.h: UR_ATTN_MODE_*


def attn_mode(key, value):
    """Kernel selection of the attention entry points (ur_attn_mode, include/unirec_hip.h): sets the process-wide word `key` and returns
    the previous value; value -1 = default, -2 = query only.  The library reads no environment variable."""
    prev = int(_lib.load().ur_attn_mode(int(key), int(value)))
    if prev < 0:
        raise ValueError(f"ur_attn_mode: unknown key {key}")
    return prev


class attn_mode_set:
    """with hip.attn_mode_set(hip.ATTN_MODE_C128, 0): ...   -- the word is restored on exit (tests, A/B timing)"""

    def __init__(self, key, value):
        self.key, self.value = key, value

    def __enter__(self):
        self.prev = attn_mode(self.key, self.value)
        return self

    def __exit__(self, *exc):
        attn_mode(self.key, self.prev)
        return False


def gemm_persistent_mode(mode):
    """0 = generic GEMM kernel only, 1 = persistent kernel where eligible, -1 = default; returns the previous setting
    (ur_gemm_persistent_mode, include/unirec_hip.h)."""
    return int(_lib.load().ur_gemm_persistent_mode(int(mode)))


# ---- LoRA adapter products (rank 16): include/unirec_hip.h, csrc/lora.hip ---------------------------
def lora_bits_ld(W):
    return int(_lib.load().ur_lora_bits_ld(int(W)))


def lora_bits_t_ld(W):
    return int(_lib.load().ur_lora_bits_t_ld(int(W)))


@_stream_family("lora_bits", lambda r, *a, **k: _nb(r))
def lora_dropout_bits(seed, p, M, W, nad, device, out=None, row0=0):
    """Dropped-flag bit planes uint8 [nad, M, ur_lora_bits_ld(W)] of nad adapters that share an [M, W] input.  row0: rows that
    precede row 0 in the global minibatch (a data-parallel rank draws the flags of ITS rows)."""
    lib = _lib.load()
    ld = int(lib.ur_lora_bits_ld(int(W)))
    bits = torch.empty((nad, M, ld), dtype=torch.uint8, device=device) if out is None else out
    check(lib.ur_lora_dropout_bits(int(seed), float(p), int(M), int(W), int(nad), bits.data_ptr(), ld, bits.stride(0), int(row0), _stream()),
          "ur_lora_dropout_bits")
    return bits


def lora_bits_transpose(bits, W, out=None):
    """Token-packed copy int32 [nad, M / 32, ur_lora_bits_t_ld(W)] of the dropped-flag planes (ur_lora_bits_transpose): what the token
    reduction (lora_reduce(..., bits_t=...)) masks its transposed fragments with.  M % 32 == 0."""
    lib = _lib.load()
    nad, M, ld = bits.shape
    if M % 32:
        raise ValueError("lora_bits_transpose: M must be a multiple of 32")
    ldt = int(lib.ur_lora_bits_t_ld(int(W)))
    bt = torch.empty((nad, M // 32, ldt), dtype=torch.int32, device=bits.device) if out is None else out
    check(lib.ur_lora_bits_transpose(bits.data_ptr(), ld, bits.stride(0), int(M), int(W), int(nad), bt.data_ptr(), ldt, bt.stride(0), _stream()),
          "ur_lora_bits_transpose")
    return bt


def lora_bits_to_keep(bits, W):
    """Unpack bit planes into keep masks uint8 [nad, M, W] (1 = kept) -- test / inspection helper (torch ops)."""
    nad, M, ld = bits.shape
    b = bits.to(torch.int32)
    pos = torch.tensor([0, 4, 1, 5, 2, 6, 3, 7], device=bits.device, dtype=torch.int32)     # element e of a byte -> bit
    dropped = (b.unsqueeze(-1) >> pos) & 1
    return (1 - dropped).reshape(nad, M, ld * 8)[:, :, :W].to(torch.uint8)


def _lora_args(X, cols, shared, bits, alpha):
    a = LoraArgs()
    if X.dtype != BF16 or not X.is_cuda or X.dim() != 2 or X.stride(1) != 1:
        raise ValueError("lora: X must be a 2-D bf16 device tensor with unit inner stride")
    a.X, a.ldx, a.M = X.data_ptr(), X.stride(0), X.shape[0]
    a.nad, a.rank, a.shared = len(cols), 16, int(shared)
    for e, (c0, w) in enumerate(cols):
        a.col0[e], a.width[e] = int(c0), int(w)
    if bits is not None:
        a.drop_bits, a.bits_ld, a.bits_stride = bits.data_ptr(), bits.stride(1), bits.stride(0)
    a.alpha = float(alpha)
    return a


def _cols_bytes(X, cols):
    return X.shape[0] * 2 * (X.shape[1] if cols is None else sum(w for _, w in cols))


@_stream_family("lora_project", lambda r, X, U, cols=None, alpha=1.0, bits=None, out=None: _cols_bytes(X, cols) + _nb(r) + (_nb(bits) if bits is not None else 0))
def lora_project(X, U, cols=None, alpha=1.0, bits=None, out=None):
    """P[m, 16a+j] = alpha * sum_w keep_a(m,w) X[m, c0_a+w] U_a[j,w].  U: list of bf16 [16, width_a] matrices.
    cols=None: the adapters share all of X's columns (optionally with dropout bit planes `bits`);
    cols=[(c0, width), ...]: adapter a owns that column range of X."""
    lib = _lib.load()
    shared = cols is None
    if shared:
        cols = [(0, X.shape[1])] * len(U)
    a = _lora_args(X, cols, shared, bits, alpha)
    for e, u in enumerate(U):
        if u.dtype != BF16 or u.shape[0] != 16 or u.stride(1) != 1 or u.shape[1] != cols[e][1]:
            raise ValueError("lora_project: U[a] must be bf16 [16, width_a]")
        a.U[e], a.ldu[e] = u.data_ptr(), u.stride(0)
    if out is None:
        out = torch.empty((X.shape[0], 16 * len(U)), dtype=BF16, device=X.device)
    a.P, a.ldp = out.data_ptr(), out.stride(0)
    check(lib.ur_lora_project(ctypes.byref(a), _stream()), "ur_lora_project")
    return out


@_stream_family("swiglu_lora", lambda r, gu, I, U, alpha=1.0, bits=None: _nb(gu) + _nb(*r) + (_nb(bits) if bits is not None else 0))
def swiglu_lora_fwd(gu, I, U, alpha=1.0, bits=None):
    """SwiGLU forward + the down_proj adapter's down projection in one pass over gu [M, 2I]: -> (act bf16 [M, I], t bf16 [M, 16])."""
    lib = _lib.load()
    _need(gu, BF16, "gu")
    M = gu.shape[0]
    act = torch.empty((M, I), dtype=BF16, device=gu.device)
    a = _lora_args(act, [(0, I)], True, bits, alpha)
    if U.dtype != BF16 or U.shape[0] != 16 or U.stride(1) != 1 or U.shape[1] != I:
        raise ValueError("swiglu_lora_fwd: U must be bf16 [16, I]")
    a.U[0], a.ldu[0] = U.data_ptr(), U.stride(0)
    t = torch.empty((M, 16), dtype=BF16, device=gu.device)
    a.P, a.ldp = t.data_ptr(), t.stride(0)
    check(lib.ur_swiglu_lora_fwd(gu.data_ptr(), act.data_ptr(), M, I, ctypes.byref(a), _stream()), "ur_swiglu_lora_fwd")
    return act, t


@_stream_family("rms_lora", lambda r, x, w, eps, U, alpha=1.0, bits=None: _nb(x) + _nb(*r) + (_nb(bits) if bits is not None else 0))
def rmsnorm_lora_fwd(x, w, eps, U, alpha=1.0, bits=None):
    """RMSNorm forward + the down projection of the 2 or 3 adapters that read the normalised activation, one pass over x:
    -> (h bf16 like x, rstd f32 [M], t bf16 [M, 16 len(U)]).  D must be 1024."""
    lib = _lib.load()
    _need(x, BF16, "x")
    D = x.shape[-1]
    M = x.numel() // D
    out = torch.empty_like(x)
    rstd = torch.empty((M,), dtype=F32, device=x.device)
    a = _lora_args(out.view(M, D), [(0, D)] * len(U), True, bits, alpha)
    for e, u in enumerate(U):
        if u.dtype != BF16 or u.shape[0] != 16 or u.stride(1) != 1 or u.shape[1] != D:
            raise ValueError("rmsnorm_lora_fwd: U[a] must be bf16 [16, D]")
        a.U[e], a.ldu[e] = u.data_ptr(), u.stride(0)
    t = torch.empty((M, 16 * len(U)), dtype=BF16, device=x.device)
    a.P, a.ldp = t.data_ptr(), t.stride(0)
    check(lib.ur_rmsnorm_lora_fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), rstd.data_ptr(), M, D, eps, ctypes.byref(a), _stream()),
          "ur_rmsnorm_lora_fwd")
    return out, rstd, t


@_stream_family("lora_reduce", lambda r, X, V, out, cols=None, nad=None, alpha=1.0, bits=None, transposed=False, bits_t=None:
                _cols_bytes(X, cols) + X.shape[0] * 2 * 16 * (len(cols) if cols is not None else int(nad)) + (_nb(bits) if bits is not None else 0))
def lora_reduce(X, V, out, cols=None, nad=None, alpha=1.0, bits=None, transposed=False, bits_t=None):
    """G_a[j,w] = alpha * sum_m V[m,16a+j] keep_a(m,w) X[m, c0_a+w] into the dense f32 tensor `out`
    ([16 nad, W] for shared columns, or [sum width, 16] with transposed=True for per-adapter column ranges).
    bits_t: lora_bits_transpose(bits, W) -- with it the launch streams X through the LDS-DMA ring kernel."""
    lib = _lib.load()
    shared = cols is None
    if shared:
        cols = [(0, X.shape[1])] * int(nad)
    a = _lora_args(X, cols, shared, bits, alpha)
    if V.dtype != BF16 or V.stride(1) != 1 or V.shape[0] != X.shape[0] or V.shape[1] < 16 * len(cols):
        raise ValueError("lora_reduce: V must be bf16 [M, >= 16 nad]")
    _need(out, F32, "out")
    if out.numel() != 16 * sum(w for _, w in cols):
        raise ValueError("lora_reduce: out has the wrong size")
    a.V, a.ldv = V.data_ptr(), V.stride(0)
    a.G, a.g_transposed = out.data_ptr(), int(transposed)
    if bits is not None and bits_t is not None:
        if bits_t.dtype != torch.int32 or bits_t.dim() != 3 or bits_t.shape[0] != bits.shape[0] or bits_t.shape[1] * 32 != X.shape[0] or bits_t.stride(2) != 1:
            raise ValueError("lora_reduce: bits_t must be the int32 [nad, M / 32, ld] tensor of lora_bits_transpose")
        a.drop_bits_t, a.bits_t_ld, a.bits_t_stride = bits_t.data_ptr(), bits_t.stride(1), bits_t.stride(0)
    wsb = lib.ur_lora_reduce_workspace_bytes(ctypes.byref(a))
    ws = workspace(wsb, X.device, "lora").data_ptr() if wsb else 0
    check(lib.ur_lora_reduce(ctypes.byref(a), ws, wsb, _stream()), "ur_lora_reduce")
    return out


def layernorm_fwd(y, gamma, beta, eps, residual=None, save_z=True, p_pre=0.0, seed_pre=0, p_post=0.0, seed_post=0,
                  M=None, drop_row0=0):
    """Returns (out, z, mean, rstd).  y may have fewer rows than M (row m reads y[m % y.shape[0]]).  drop_row0: rows that
    precede this launch's row 0 in the global minibatch (dropout masks keyed on the global element index)."""
    lib = _lib.load()
    _need(y, BF16, "y")
    H = y.shape[-1]
    y_rows = y.numel() // H
    if M is None:
        M = y_rows
    dev = y.device
    out = torch.empty((M, H), dtype=BF16, device=dev)
    z = torch.empty((M, H), dtype=BF16, device=dev) if save_z else None
    mean = torch.empty((M,), dtype=F32, device=dev)
    rstd = torch.empty((M,), dtype=F32, device=dev)
    check(lib.ur_layernorm_fwd(y.data_ptr(), y_rows, _p(residual), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), _p(z),
                               mean.data_ptr(), rstd.data_ptr(), M, H, eps, p_pre, seed_pre, p_post, seed_post, int(drop_row0), _stream()),
          "ur_layernorm_fwd")
    return out, z, mean, rstd


def layernorm_bwd(dout, z, mean, rstd, gamma, dgamma, dbeta, dbias=None, p_pre=0.0, seed_pre=0, p_post=0.0, seed_post=0,
                  need_dy=True, drop_row0=0, defer_reduce=False):
    """Returns (dz, dy); dgamma/dbeta/dbias (f32 [H]) are overwritten in place.
    defer_reduce=True: returns (dz, dy, finish) -- the call stops at the per-block partial sums (a scratch tensor of its own) and
    finish() launches their reduction into dgamma / dbeta / dbias on the stream that is current WHEN IT IS CALLED (ordering against this
    call is the caller's: qformer.py runs it on its side stream behind an event)."""
    lib = _lib.load()
    M, H = z.shape
    dz = torch.empty_like(z)
    dy = dz if (p_pre == 0.0 or not need_dy) else torch.empty_like(z)
    wsb = lib.ur_layernorm_bwd_workspace_bytes(H)
    if defer_reduce:
        ws = torch.empty(int(wsb), dtype=torch.uint8, device=z.device)
        check(lib.ur_layernorm_bwd(dout.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                   dz.data_ptr(), dy.data_ptr() if need_dy else 0, 0, 0, 0,
                                   M, H, p_pre, seed_pre, p_post, seed_post, int(drop_row0), ws.data_ptr(), wsb, _stream()), "ur_layernorm_bwd")

        def finish():
            check(lib.ur_layernorm_bwd_reduce(ws.data_ptr(), M, H, dgamma.data_ptr(), dbeta.data_ptr(), _p(dbias), _stream()), "ur_layernorm_bwd_reduce")
        finish.scratch = ws
        return dz, dy, finish
    ws = workspace(wsb, z.device, "ln")
    check(lib.ur_layernorm_bwd(dout.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                               dz.data_ptr(), dy.data_ptr() if need_dy else 0, dgamma.data_ptr(), dbeta.data_ptr(), _p(dbias),
                               M, H, p_pre, seed_pre, p_post, seed_post, int(drop_row0), ws.data_ptr(), wsb, _stream()), "ur_layernorm_bwd")
    return dz, dy


def batch_reduce(x, nb, rows, H, out=None):
    lib = _lib.load()
    _need(x, BF16, "x")
    if out is None:
        out = torch.empty((rows, H), dtype=F32, device=x.device)
    wsb = lib.ur_batch_reduce_workspace_bytes(nb, rows, H)
    ws = workspace(wsb, x.device, "br")
    check(lib.ur_batch_reduce(x.data_ptr(), out.data_ptr(), nb, rows, H, ws.data_ptr(), wsb, _stream()), "ur_batch_reduce")
    return out


def colsum(x2d, out=None):
    """f32 column sums of a contiguous bf16 [M,N] (bias gradients)."""
    M, N = x2d.shape
    o = batch_reduce(x2d, M, 1, N, out=None if out is None else out.view(1, N))
    return o.view(N)


def rmsnorm_fwd(x, w, eps):
    lib = _lib.load()
    _need(x, BF16, "x")
    D = x.shape[-1]
    M = x.numel() // D
    out = torch.empty_like(x)
    rstd = torch.empty((M,), dtype=F32, device=x.device)
    check(lib.ur_rmsnorm_fwd(x.data_ptr(), w.data_ptr(), out.data_ptr(), rstd.data_ptr(), M, D, eps, _stream()), "ur_rmsnorm_fwd")
    return out, rstd


@_stream_family("rms_bwd", lambda r, dout, x, w, rstd, add=None: _nb(dout, x, r, add))
def rmsnorm_bwd(dout, x, w, rstd, add=None):
    lib = _lib.load()
    D = x.shape[-1]
    M = x.numel() // D
    dx = torch.empty_like(x)
    check(lib.ur_rmsnorm_bwd(dout.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), _p(add), dx.data_ptr(), M, D, _stream()),
          "ur_rmsnorm_bwd")
    return dx


class AttnCtx:
    """What ur_attn_bwd needs from the forward (argument struct + the tensors it points into)."""
    __slots__ = ("args", "keep", "o", "stats")


def attn_ctx_prefix(ctx, Bg):
    """The forward context restricted to its first Bg samples (a backward over the leading rows of a larger forward): same pointers
    and strides -- batch b of every operand sits at b * (tokens * stride) -- with B = Bg."""
    a = AttnArgs.from_buffer_copy(ctx.args)
    a.B = int(Bg)
    c = AttnCtx()
    c.args, c.keep, c.o, c.stats = a, ctx.keep, ctx.o, ctx.stats
    return c


def _tok_stride(t):
    # [B,S,heads,hd] view (possibly a slice of a fused projection buffer): elements between tokens
    if t.stride(3) != 1 or t.stride(2) != t.shape[3] or t.stride(0) != t.shape[1] * t.stride(1):
        raise ValueError("attention operand must be [B,S,heads,hd] with dense heads and uniform token stride")
    return t.stride(1)


def attn_fwd(q, k, v, *, causal, key_mask=None, scale=None, dropout_p=0.0, seed=0, out=None, drop_batch0=0):
    """q [B,Sq,nq,hd], k/v [B,Sk,nkv,hd] bf16 (strided views allowed).  key_mask uint8 [B,Sk] or None.
    Returns (o [B,Sq,nq,hd] contiguous, ctx)."""
    lib = _lib.load()
    B, Sq, nq, hd = q.shape
    Sk, nkv = k.shape[1], k.shape[2]
    if out is None:
        out = torch.empty((B, Sq, nq, hd), dtype=BF16, device=q.device)
    stats = torch.empty((B, nq, Sq, 2), dtype=F32, device=q.device)
    if key_mask is not None and (key_mask.dtype != torch.uint8 or not key_mask.is_contiguous()):
        raise TypeError("key_mask must be a contiguous uint8 [B,Sk] tensor")
    a = AttnArgs()
    a.q, a.k, a.v, a.o, a.stats = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), stats.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo = _tok_stride(q), _tok_stride(k), _tok_stride(v), _tok_stride(out)
    a.key_mask = _p(key_mask)
    a.B, a.Sq, a.Sk, a.nq, a.nkv, a.head_dim = B, Sq, Sk, nq, nkv, hd
    a.causal = int(causal)
    a.scale = float(scale if scale is not None else hd ** -0.5)
    a.dropout_p, a.seed, a.drop_batch0 = float(dropout_p), int(seed), int(drop_batch0)
    if PROFILE_ATTN is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ur_attn_fwd(ctypes.byref(a), _stream()), "ur_attn_fwd")
        e1.record()
        PROFILE_ATTN.append((e0, e1, "fwd", B, Sq, Sk, nq, hd, int(causal)))
    else:
        check(lib.ur_attn_fwd(ctypes.byref(a), _stream()), "ur_attn_fwd")
    ctx = AttnCtx()
    ctx.args, ctx.keep, ctx.o, ctx.stats = a, (q, k, v, key_mask), out, stats
    return out, ctx


def attn_bwd_kv_colsum_supported(ctx):
    """True when attn_bwd(ctx, ..., kv_colsum=) can also produce the column sums of dK | dV (the few-query dK/dV kernel's shapes)."""
    return int(_lib.load().ur_attn_bwd_kv_colsum_floats(ctypes.byref(ctx.args))) > 0


def attn_bwd(ctx, dout, dq=None, dk=None, dv=None, rope_q=None, rope_rstd=None, rope_k=None, kv_colsum=None):
    """dout [B,Sq,nq,hd] -> (dq, dk, dv); outputs may be strided views into a fused gradient buffer.
    rope_q = (q_raw [M, >= nq*hd] view, q_norm_weight f32 [hd], cos, sin, eps, dq_raw [M, >= nq*hd] view): the dQ kernel carries
    the q-norm + RoPE backward and writes the gradient of the RAW q projection into dq_raw; no dq is produced (returns None).
    rope_rstd = (rstd f32 [M, nh], first q head's column): the forward ran q-norm + RoPE in the q|k|v GEMM epilogue -- rope_q[0]
    is then the ROPED q (the attention's own q) and 1 / rms comes from rstd (ur_attn_bwd_args.rope_rstd).
    rope_k = (k_r [M, >= nkv*hd] view, k_norm_weight, first k head's column of rstd, dk_raw [M, >= nkv*hd] view) with rope_rstd: the
    k heads' backward too (in the dK/dV kernel's store where the generated kernel runs, else by the stand-alone kernel inside the call;
    dk is scratch then).
    kv_colsum = f32 [2 * nq * hd] (contiguous): also receives [column sums of dK | of dV] over all keys and batch rows -- the K | V
    projections' bias gradients -- when attn_bwd_kv_colsum_supported(ctx)."""
    lib = _lib.load()
    q, k, v, _ = ctx.keep
    if dq is None and rope_q is None:
        dq = torch.empty(q.shape, dtype=BF16, device=q.device)
    if dk is None:
        dk = torch.empty(k.shape, dtype=BF16, device=q.device)
    if dv is None:
        dv = torch.empty(v.shape, dtype=BF16, device=q.device)
    a = ctx.args
    # workspace: the row constants (-rowsum(dO*O), -LSE/scale) + the call's own work-queue words (the library owns no device state)
    delta = torch.empty((int(lib.ur_attn_bwd_workspace_floats(a.B, a.nq, a.Sq)),), dtype=F32, device=q.device)
    g = AttnBwdArgs()
    g.dout, g.dq, g.dk, g.dv = dout.data_ptr(), (0 if dq is None else dq.data_ptr()), dk.data_ptr(), dv.data_ptr()
    g.lddo, g.lddq, g.lddk, g.lddv = _tok_stride(dout), (0 if dq is None else _tok_stride(dq)), _tok_stride(dk), _tok_stride(dv)
    if rope_q is not None:
        q_raw, qw, cos, sin, eps, dq_raw = rope_q
        g.rope_q_raw, g.rope_ldraw, g.rope_q_weight = q_raw.data_ptr(), q_raw.stride(0), qw.data_ptr()
        g.rope_cos, g.rope_sin, g.rope_eps = cos.data_ptr(), sin.data_ptr(), float(eps)
        g.rope_dq_raw, g.rope_lddraw = dq_raw.data_ptr(), dq_raw.stride(0)
        if rope_rstd is not None:
            rstd, h0 = rope_rstd
            _need(rstd, F32, "rope_rstd")
            g.rope_rstd, g.rope_rstd_ld, g.rope_rstd_h0 = rstd.data_ptr(), rstd.stride(0), int(h0)
            if rope_k is not None:
                k_r, kw, hk0, dk_raw = rope_k
                g.rope_k, g.rope_ldk, g.rope_k_weight, g.rope_rstd_hk0 = k_r.data_ptr(), k_r.stride(0), kw.data_ptr(), int(hk0)
                g.rope_dk_raw, g.rope_lddkraw = dk_raw.data_ptr(), dk_raw.stride(0)
    g.delta = delta.data_ptr()
    if kv_colsum is not None:
        nws = int(lib.ur_attn_bwd_kv_colsum_floats(ctypes.byref(a)))
        if nws <= 0:
            raise ValueError("attn_bwd: kv_colsum is not available for this shape (attn_bwd_kv_colsum_supported)")
        _need(kv_colsum, F32, "kv_colsum")
        if kv_colsum.numel() != 2 * a.nq * a.head_dim or not kv_colsum.is_contiguous():
            raise ValueError("attn_bwd: kv_colsum must be a contiguous f32 tensor of 2 * nq * head_dim elements")
        cws = torch.empty((nws,), dtype=F32, device=q.device)
        g.kv_colsum, g.kv_colsum_ws = kv_colsum.data_ptr(), cws.data_ptr()
    if PROFILE_ATTN is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.ur_attn_bwd(ctypes.byref(a), ctypes.byref(g), _stream()), "ur_attn_bwd")
        e1.record()
        PROFILE_ATTN.append((e0, e1, "bwd", a.B, a.Sq, a.Sk, a.nq, a.head_dim, a.causal))
    else:
        check(lib.ur_attn_bwd(ctypes.byref(a), ctypes.byref(g), _stream()), "ur_attn_bwd")
    return dq, dk, dv


def dropout_keep(seed, p, idx0, n, device):
    """uint8 [n]: keep flags of elements idx0 .. idx0 + n - 1 of the dropout stream `seed` (ur_dropout_keep) -- test / inspection
    helper: hidden dropout counters are (drop_row0 + row) * H + col (attention probabilities: attn_dropout_keep)."""
    lib = _lib.load()
    out = torch.empty((int(n),), dtype=torch.uint8, device=device)
    check(lib.ur_dropout_keep(int(seed), float(p), int(idx0), int(n), out.data_ptr(), _stream()), "ur_dropout_keep")
    return out


def attn_dropout_keep(seed, p, row0, nrows, Sk, device):
    """uint8 [nrows, Sk]: keep flags of the attention-probability dropout for the rows row0 .. (ur_attn_dropout_keep); a row is
    ((drop_batch0 + b) * heads + h) * Sq + query -- test / inspection helper."""
    lib = _lib.load()
    out = torch.empty((int(nrows), int(Sk)), dtype=torch.uint8, device=device)
    check(lib.ur_attn_dropout_keep(int(seed), float(p), int(row0), int(nrows), int(Sk), out.data_ptr(), _stream()), "ur_attn_dropout_keep")
    return out


def cast_f32_to_bf16(src, dst=None):
    lib = _lib.load()
    _need(src, F32, "src")
    if dst is None:
        dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    check(lib.ur_cast_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "ur_cast_f32_to_bf16")
    return dst


def cast_bf16_to_f32(src, dst=None):
    lib = _lib.load()
    _need(src, BF16, "src")
    if dst is None:
        dst = torch.empty(src.shape, dtype=F32, device=src.device)
    check(lib.ur_cast_bf16_to_f32(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "ur_cast_bf16_to_f32")
    return dst


class BatchedTranspose:
    """Many small bf16 transposes as ONE launch (ur_transpose_bf16_batched).  Built once over persistent source tensors
    (views of a parameter pack's bf16 shadow); run() refreshes every destination; out[i] is the transposed view."""

    def __init__(self, sources):
        import numpy as np
        dev = sources[0].device
        total = sum(t.numel() for t in sources)
        self.buf = torch.empty(total, dtype=BF16, device=dev)
        self.out, rec, off, tile = [], np.zeros((len(sources), 4), dtype=np.int64), 0, 0
        for i, t in enumerate(sources):
            if t.dim() != 2 or not t.is_contiguous() or t.dtype != BF16:
                raise ValueError("BatchedTranspose: contiguous 2-D bf16 sources only")
            rows, cols = t.shape
            d = self.buf[off:off + rows * cols].view(cols, rows)
            off += rows * cols
            self.out.append(d)
            ntx = (cols + 31) // 32
            rec[i] = (t.data_ptr(), d.data_ptr(), rows | (cols << 32), tile | (ntx << 32))
            tile += ntx * ((rows + 31) // 32)
        self.sources = sources                       # keeps the storages alive
        self.n, self.tiles = len(sources), tile
        self.desc = torch.from_numpy(rec).to(dev)
        self.key = tuple(t.data_ptr() for t in sources)

    def run(self):
        check(_lib.load().ur_transpose_bf16_batched(self.desc.data_ptr(), self.n, self.tiles, _stream()), "ur_transpose_bf16_batched")
        return self.out


def transpose_bf16(src, dst=None):
    lib = _lib.load()
    _need(src, BF16, "src")
    rows, cols = src.shape
    if dst is None:
        dst = torch.empty((cols, rows), dtype=BF16, device=src.device)
    check(lib.ur_transpose_bf16(src.data_ptr(), dst.data_ptr(), rows, cols, _stream()), "ur_transpose_bf16")
    return dst


def add_bf16(a, b, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(a)
    check(lib.ur_add_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "ur_add_bf16")
    return out


def swiglu_fwd(gu, I, out=None):
    lib = _lib.load()
    M = gu.numel() // (2 * I)
    act = torch.empty((M, I), dtype=BF16, device=gu.device) if out is None else out
    assert act.is_contiguous() and act.numel() == M * I and act.dtype == BF16
    check(lib.ur_swiglu_fwd(gu.data_ptr(), act.data_ptr(), M, I, _stream()), "ur_swiglu_fwd")
    return act


def swiglu_bwd(dact, gu, I):
    lib = _lib.load()
    M = gu.numel() // (2 * I)
    dgu = torch.empty_like(gu)
    check(lib.ur_swiglu_bwd(dact.data_ptr(), gu.data_ptr(), dgu.data_ptr(), M, I, _stream()), "ur_swiglu_bwd")
    return dgu


@_stream_family("adamw", lambda r, param, grad, exp_avg, exp_avg_sq, *a, **k: 2 * _nb(param, exp_avg, exp_avg_sq) + _nb(grad))
def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    lib = _lib.load()
    check(lib.ur_adamw_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), param.numel(), lr,
                            beta1, beta2, eps, weight_decay, step, grad_scale, _stream()), "ur_adamw_step")


# ---- Qwen3-side ops ------------------------------------------------------------------------------
def rope_table(S, head_dim, theta, device):
    lib = _lib.load()
    cos = torch.empty((S, head_dim // 2), dtype=F32, device=device)
    sin = torch.empty((S, head_dim // 2), dtype=F32, device=device)
    check(lib.ur_rope_table(cos.data_ptr(), sin.data_ptr(), S, head_dim, theta, _stream()), "ur_rope_table")
    return cos, sin


def qknorm_rope_fwd(qkv_raw, qw, kw, cos, sin, S, nq, nkv, hd, eps):
    """qkv_raw [M,(nq+2nkv)*hd] -> (q_out [M,nq*hd], k_out [M,nkv*hd])."""
    lib = _lib.load()
    M = qkv_raw.shape[0]
    q_out = torch.empty((M, nq * hd), dtype=BF16, device=qkv_raw.device)
    k_out = torch.empty((M, nkv * hd), dtype=BF16, device=qkv_raw.device)
    check(lib.ur_qknorm_rope_fwd(qkv_raw.data_ptr(), qkv_raw.stride(0), qw.data_ptr(), kw.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                 q_out.data_ptr(), k_out.data_ptr(), M, S, nq, nkv, hd, eps, _stream()), "ur_qknorm_rope_fwd")
    return q_out, k_out


def qknorm_rope_bwd(dq_out, dk_out, qkv_raw, qw, kw, cos, sin, dqkv_raw, S, nq, nkv, hd, eps):
    lib = _lib.load()
    M = qkv_raw.shape[0]
    check(lib.ur_qknorm_rope_bwd(dq_out.data_ptr(), dk_out.data_ptr(), qkv_raw.data_ptr(), qkv_raw.stride(0), qw.data_ptr(), kw.data_ptr(),
                                 cos.data_ptr(), sin.data_ptr(), dqkv_raw.data_ptr(), dqkv_raw.stride(0), M, S, nq, nkv, hd, eps,
                                 _stream()), "ur_qknorm_rope_bwd")
    return dqkv_raw


def embed_inject_fwd(embed, input_ids, item_tokens, first_special_id):
    """embed [V,D] bf16, input_ids int64 [B,S], item_tokens [B,T,D] bf16 or None -> [B,S,D] bf16."""
    lib = _lib.load()
    _need(embed, BF16, "embed")
    _need(input_ids, torch.int64, "input_ids")
    B, S = input_ids.shape
    V, D = embed.shape
    T = 0 if item_tokens is None else item_tokens.shape[1]
    out = torch.empty((B, S, D), dtype=BF16, device=embed.device)
    check(lib.ur_embed_inject_fwd(embed.data_ptr(), V, input_ids.data_ptr(), _p(item_tokens), first_special_id, T, out.data_ptr(),
                                  B, S, D, _stream()), "ur_embed_inject_fwd")
    return out


def inject_bwd(dx, input_ids, first_special_id, T):
    lib = _lib.load()
    B, S, D = dx.shape
    d_tok = torch.empty((B, T, D), dtype=BF16, device=dx.device)
    check(lib.ur_inject_bwd(dx.data_ptr(), input_ids.data_ptr(), first_special_id, T, d_tok.data_ptr(), B, S, D, _stream()),
          "ur_inject_bwd")
    return d_tok


def mean_pool_fwd(x, out_f32=True, out_bf16=False):
    """x [B,S,D] bf16 -> (f32 [B,D] or None, bf16 [B,D] or None)."""
    lib = _lib.load()
    _need(x, BF16, "x")
    B, S, D = x.shape
    o32 = torch.empty((B, D), dtype=F32, device=x.device) if out_f32 else None
    o16 = torch.empty((B, D), dtype=BF16, device=x.device) if out_bf16 else None
    wsb = lib.ur_mean_pool_workspace_bytes(B, D)
    ws = workspace(wsb, x.device, "pool")
    check(lib.ur_mean_pool_fwd(x.data_ptr(), _p(o32), _p(o16), B, S, D, ws.data_ptr(), wsb, _stream()), "ur_mean_pool_fwd")
    return o32, o16


def mean_pool_bwd(dout, S):
    """dout [B,D] (f32 or bf16) -> dx [B,S,D] bf16 = dout / S broadcast."""
    lib = _lib.load()
    B, D = dout.shape
    dx = torch.empty((B, S, D), dtype=BF16, device=dout.device)
    d32 = dout.data_ptr() if dout.dtype == F32 else 0
    d16 = dout.data_ptr() if dout.dtype == BF16 else 0
    check(lib.ur_mean_pool_bwd(d32, d16, dx.data_ptr(), B, S, D, _stream()), "ur_mean_pool_bwd")
    return dx


# ---- ranking head --------------------------------------------------------------------------------
@_stream_family("J6_candidate_scores", lambda r, user, pos, neg: _nb(user, pos, neg) + _nb(*r))
def cosine_scores(user, pos, neg):
    """user [B,D], pos [B,D], neg [B,N,D] f32 -> (scores [B,1+N], cand_inv_norm [B,1+N])."""
    lib = _lib.load()
    for t, n in ((user, "user"), (pos, "pos"), (neg, "neg")):
        _need(t, F32, n)
    B, D = user.shape
    N = neg.shape[1]
    scores = torch.empty((B, N + 1), dtype=F32, device=user.device)
    inv = torch.empty((B, N + 1), dtype=F32, device=user.device)
    check(lib.ur_cosine_scores(user.data_ptr(), pos.data_ptr(), neg.data_ptr(), scores.data_ptr(), inv.data_ptr(), B, N, D, _stream()),
          "ur_cosine_scores")
    return scores, inv


def infonce_fwd_bwd(user, pos, neg, neg_mask, scores, inv, temperature=0.07, grad_scale=1.0, need_grad=True):
    """Returns (loss [1] f32, d_user [B,D] f32 or None)."""
    lib = _lib.load()
    B, D = user.shape
    N = neg.shape[1]
    loss = torch.empty((1,), dtype=F32, device=user.device)
    du = torch.empty((B, D), dtype=F32, device=user.device) if need_grad else None
    wsb = lib.ur_infonce_workspace_bytes(B, N, D)
    ws = workspace(wsb, user.device, "infonce")
    check(lib.ur_infonce_fwd_bwd(user.data_ptr(), pos.data_ptr(), neg.data_ptr(), _p(neg_mask), scores.data_ptr(), inv.data_ptr(),
                                 temperature, grad_scale, loss.data_ptr(), _p(du), B, N, D, ws.data_ptr(), wsb, _stream()),
          "ur_infonce_fwd_bwd")
    return loss, du


def mrr_rank(scores, neg_mask=None):
    lib = _lib.load()
    B, C = scores.shape
    rank = torch.empty((B,), dtype=torch.int32, device=scores.device)
    check(lib.ur_mrr_rank(scores.data_ptr(), _p(neg_mask), rank.data_ptr(), B, C - 1, _stream()), "ur_mrr_rank")
    return rank


def topk(scores, K):
    lib = _lib.load()
    _need(scores, F32, "scores")
    B, C = scores.shape
    idx = torch.empty((B, K), dtype=torch.int32, device=scores.device)
    val = torch.empty((B, K), dtype=F32, device=scores.device)
    ws = workspace(B * C, scores.device, "topk")
    check(lib.ur_topk(scores.data_ptr(), B, C, K, idx.data_ptr(), val.data_ptr(), ws.data_ptr(), B * C, _stream()), "ur_topk")
    return idx, val


# ---- data path: csrc/catalog.hip ----------------------------------
_KIND = {torch.uint8: 0, BF16: 1, F32: 2}


def gather_rows(src, idx, out_dtype=None):
    """out[i] = src[idx[i]] over the leading axis (idx int64, any shape; negative / out-of-range -> zero row).
    f32 sources may be gathered straight to bf16.  Returns idx.shape + src.shape[1:]."""
    lib = _lib.load()
    if not src.is_cuda or not src.is_contiguous() or src.dtype not in _KIND:
        raise ValueError("gather_rows: src must be a contiguous uint8 / bf16 / f32 device tensor")
    out_dtype = src.dtype if out_dtype is None else out_dtype
    idx = idx.to(device=src.device, dtype=torch.int64).contiguous()
    row = 1
    for s_ in src.shape[1:]:
        row *= s_
    out = torch.empty(tuple(idx.shape) + tuple(src.shape[1:]), dtype=out_dtype, device=src.device)
    check(lib.ur_gather_rows(src.data_ptr(), _KIND[src.dtype], out.data_ptr(), _KIND[out_dtype], idx.data_ptr(), row, idx.numel(),
                             src.shape[0], _stream()), "ur_gather_rows")
    return out


def catalog_scores(user, catalog, cat_inv_norm=None):
    """cosine scores [B,N] of user [B,D] f32 against a shared catalogue [N,D] f32; returns (scores, cat_inv_norm)
    so repeated calls over the same catalogue skip its norm pass."""
    lib = _lib.load()
    _need(user, F32, "user")
    _need(catalog, F32, "catalog")
    B, D = user.shape
    N = catalog.shape[0]
    scores = torch.empty((B, N), dtype=F32, device=user.device)
    uinv = torch.empty((B,), dtype=F32, device=user.device)
    ready = cat_inv_norm is not None
    if not ready:
        cat_inv_norm = torch.empty((N,), dtype=F32, device=user.device)
    check(lib.ur_catalog_scores(user.data_ptr(), catalog.data_ptr(), scores.data_ptr(), uinv.data_ptr(), cat_inv_norm.data_ptr(),
                                int(ready), B, N, D, _stream()), "ur_catalog_scores")
    return scores, cat_inv_norm


def rank_of_index(scores, gt_index):
    """1-based rank of column gt_index[b] in the descending order of scores[b] (ties go to the ground truth)."""
    lib = _lib.load()
    _need(scores, F32, "scores")
    B, N = scores.shape
    gt = gt_index.to(device=scores.device, dtype=torch.int64).contiguous()
    rank = torch.empty((B,), dtype=torch.int32, device=scores.device)
    check(lib.ur_rank_of_index(scores.data_ptr(), gt.data_ptr(), rank.data_ptr(), B, N, _stream()), "ur_rank_of_index")
    return rank


@_stream_family("lora_bgrad", lambda r, dy, t, Bt, cols, gB, alpha=1.0, out=None: _cols_bytes(dy, cols) + 2 * dy.shape[0] * 2 * 16 * len(cols))
def lora_bgrad(dy, t, Bt, cols, gB, alpha=1.0, out=None):
    """One pass over dy: returns tb [M, 16 nad] = alpha * dy_a B_a (Bt: list of B_a^T [16, width_a]) and fills
    gB [sum width, 16] f32 with dB_a = dy_a^T t_a (adapter a owns columns cols[a] of dy, t holds t_a at columns 16a)."""
    lib = _lib.load()
    a = _lora_args(dy, cols, False, None, alpha)
    for e, u in enumerate(Bt):
        if u.dtype != BF16 or u.shape[0] != 16 or u.stride(1) != 1 or u.shape[1] != cols[e][1]:
            raise ValueError("lora_bgrad: Bt[a] must be bf16 [16, width_a]")
        a.U[e], a.ldu[e] = u.data_ptr(), u.stride(0)
    if t.dtype != BF16 or t.stride(1) != 1 or t.shape[0] != dy.shape[0] or t.shape[1] < 16 * len(cols):
        raise ValueError("lora_bgrad: t must be bf16 [M, >= 16 nad]")
    _need(gB, F32, "gB")
    if gB.numel() != 16 * sum(w for _, w in cols):
        raise ValueError("lora_bgrad: gB has the wrong size")
    if out is None:
        out = torch.empty((dy.shape[0], 16 * len(cols)), dtype=BF16, device=dy.device)
    a.V, a.ldv = t.data_ptr(), t.stride(0)
    a.P, a.ldp = out.data_ptr(), out.stride(0)
    a.G, a.g_transposed = gB.data_ptr(), 1
    wsb = lib.ur_lora_bgrad_workspace_bytes(ctypes.byref(a))
    ws = workspace(wsb, dy.device, "lora").data_ptr() if wsb else 0
    check(lib.ur_lora_bgrad(ctypes.byref(a), ws, wsb, _stream()), "ur_lora_bgrad")
    return out


def context_mlp1(x, kind, W1, b1):
    """bf16 [n, 2H] = gelu(W1 feat(x) + b1); kind 0: x = f32 timestamps [n]; kind 1: x = f32 (lat, lon) [n,2]."""
    lib = _lib.load()
    _need(x, F32, "x")
    _need(W1, F32, "W1")
    _need(b1, F32, "b1")
    n = x.shape[0]
    out = torch.empty((n, W1.shape[0]), dtype=BF16, device=x.device)
    check(lib.ur_context_mlp1(x.data_ptr(), int(kind), W1.data_ptr(), b1.data_ptr(), out.data_ptr(), n, W1.shape[0], _stream()), "ur_context_mlp1")
    return out


# ---- heads / losses ------------------------------------------------------------------------------
def gelu_bwd(dy, u):
    lib = _lib.load()
    dx = torch.empty_like(u)
    check(lib.ur_gelu_bwd(dy.data_ptr(), u.data_ptr(), dx.data_ptr(), u.numel(), _stream()), "ur_gelu_bwd")
    return dx


def _heads_ws(Q, F, device):
    lib = _lib.load()
    n = max(lib.ur_heads_workspace_bytes(Q, F), 4096 * 4)
    return workspace(n, device, "heads"), n


def field_projection_fwd(rec16, Wf, bf):
    """rec16 [B,Q,E] bf16, Wf [F,Q] f32, bf [F] f32 -> [B,F,E] f32."""
    lib = _lib.load()
    B, Q, E = rec16.shape
    F_ = Wf.shape[0]
    out = torch.empty((B, F_, E), dtype=F32, device=rec16.device)
    check(lib.ur_field_projection_fwd(rec16.data_ptr(), Wf.data_ptr(), bf.data_ptr(), out.data_ptr(), B, Q, F_, E, _stream()),
          "ur_field_projection_fwd")
    return out


def field_projection_bwd(dout, rec16, Wf, dWf, dbf):
    lib = _lib.load()
    B, Q, E = rec16.shape
    F_ = Wf.shape[0]
    drec = torch.empty_like(rec16)
    ws, n = _heads_ws(Q, F_, rec16.device)
    check(lib.ur_field_projection_bwd(dout.data_ptr(), rec16.data_ptr(), Wf.data_ptr(), drec.data_ptr(), dWf.data_ptr(), dbf.data_ptr(),
                                      B, Q, F_, E, ws.data_ptr(), n, _stream()), "ur_field_projection_bwd")
    return drec


def recon_stats(rec, x, mask):
    """f32 [rows,E] x2 + f32 mask [rows] -> sums3 = (sum mask*se, sum mask, sum cos over valid rows)."""
    lib = _lib.load()
    E = rec.shape[-1]
    rows = rec.numel() // E
    sums = torch.empty((3,), dtype=F32, device=rec.device)
    ws, n = _heads_ws(1, 1, rec.device)
    check(lib.ur_recon_stats(rec.data_ptr(), x.data_ptr(), mask.data_ptr(), sums.data_ptr(), rows, E, ws.data_ptr(), n, _stream()),
          "ur_recon_stats")
    return sums


def recon_grad(rec, x, mask, sums, coef):
    lib = _lib.load()
    E = rec.shape[-1]
    rows = rec.numel() // E
    d = torch.empty_like(rec)
    check(lib.ur_recon_grad(rec.data_ptr(), x.data_ptr(), mask.data_ptr(), sums.data_ptr(), coef, d.data_ptr(), rows, E, _stream()),
          "ur_recon_grad")
    return d


def triplet_margin(anchor, pos, neg, margin, coef, need_grad=True):
    lib = _lib.load()
    B, E = anchor.shape
    loss = torch.empty((1,), dtype=F32, device=anchor.device)
    da = torch.empty_like(anchor) if need_grad else None
    ws, n = _heads_ws(1, 1, anchor.device)
    if n < 4 * B:
        ws, n = workspace(4 * B, anchor.device, "heads"), 4 * B
    check(lib.ur_triplet_margin(anchor.data_ptr(), pos.data_ptr(), neg.data_ptr(), margin, coef, loss.data_ptr(), _p(da), B, E,
                                ws.data_ptr(), n, _stream()), "ur_triplet_margin")
    return loss, da


def mse_loss(a, b, coef=1.0, need_grad=True):
    lib = _lib.load()
    loss = torch.empty((1,), dtype=F32, device=a.device)
    da = torch.empty_like(a) if need_grad else None
    ws, n = _heads_ws(1, 1, a.device)
    check(lib.ur_mse_loss(a.data_ptr(), b.data_ptr(), a.numel(), coef, loss.data_ptr(), _p(da), ws.data_ptr(), n, _stream()),
          "ur_mse_loss")
    return loss, da


def user_sequence_assemble(item_tokens, context, lengths, dropout_p=0.0, seed=0, drop_batch0=0):
    """item_tokens [B,L,Qi,H] bf16, context [B,L,H] bf16, lengths int32 [B] -> (out [B,L*Qi,H] bf16, mask [B,L*Qi] f32)."""
    lib = _lib.load()
    _need(item_tokens, BF16, "item_tokens")
    _need(context, BF16, "context")
    _need(lengths, torch.int32, "lengths")
    B, L, Qi, H = item_tokens.shape
    out = torch.empty((B, L * Qi, H), dtype=BF16, device=item_tokens.device)
    mask = torch.empty((B, L * Qi), dtype=F32, device=item_tokens.device)
    check(lib.ur_user_sequence_assemble(item_tokens.data_ptr(), context.data_ptr(), lengths.data_ptr(), out.data_ptr(), mask.data_ptr(),
                                        B, L, Qi, H, dropout_p, seed, int(drop_batch0), _stream()), "ur_user_sequence_assemble")
    return out, mask

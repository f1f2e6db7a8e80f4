"""Qwen3 decoder + LoRA on the MI355X: the third-party half of the joint path.

The reference loads ``AutoModel.from_pretrained("Qwen/Qwen3-Embedding-0.6B")`` and wraps it with
``peft.get_peft_model`` (training/train_item_individual_token_joint.py:98-132); neither package's
source is part of the reference tree, so this module restates the published arithmetic
(transformers ``modeling_qwen3.py``: RMSNorm :59-64, MLP :81-83, RoPE :107-170, attention :185-280,
decoder layer :306-331, model :367-425) and the LoRA definition
``y = W x + (alpha/r) B A x`` with HF-compatible parameter names:
    embed_tokens.weight, layers.{i}.input_layernorm.weight, layers.{i}.self_attn.{q,k,v,o}_proj.weight,
    layers.{i}.self_attn.{q,k}_norm.weight, layers.{i}.post_attention_layernorm.weight,
    layers.{i}.mlp.{gate,up,down}_proj.weight, norm.weight
LoRA tensors are ``<proj>.lora_A.weight [r,in]`` / ``<proj>.lora_B.weight [out,r]`` (the only trainable
tensors, bias none: :123-129); ``peft_state_dict()`` exports them under peft's adapter key names.

Execution: the whole stack (embed + Q-Former token injection, 28 decoder layers, final norm, mean
pool over ALL positions :179-181) is ONE autograd node.  Base weights are frozen, so the backward is
dX GEMMs on the frozen bf16 weights plus rank-r reductions for dA/dB; every LoRA B-product rides in
the base GEMM's accumulators (second K-range of ur_gemm).  Activations are kept in HBM (288 GB:
no gradient checkpointing); only the RMSNorm outputs are recomputed in the backward.
"""
import math
import os

import torch
import torch.nn as nn

from . import hip
from .packing import ParamPack, norm_device

BF16, F32 = torch.bfloat16, torch.float32
LORA_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


class Qwen3Config:
    """Shape of Qwen3-Embedding-0.6B by default (public model card)."""

    def __init__(self, vocab_size=151669, hidden_size=1024, intermediate_size=3072, num_hidden_layers=28,
                 num_attention_heads=16, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6, rope_theta=1e6,
                 lora_r=16, lora_alpha=32.0, lora_dropout=0.1, initializer_range=0.02):
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads, self.num_key_value_heads = num_hidden_layers, num_attention_heads, num_key_value_heads
        self.head_dim, self.rms_norm_eps, self.rope_theta = head_dim, rms_norm_eps, rope_theta
        self.lora_r, self.lora_alpha, self.lora_dropout = lora_r, lora_alpha, lora_dropout
        self.initializer_range = initializer_range


class _Proj(nn.Module):
    def __init__(self, fin, fout, r):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin), requires_grad=False)
        if r > 0:
            self.lora_A = nn.Linear(fin, r, bias=False)
            self.lora_B = nn.Linear(r, fout, bias=False)


class _Norm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d), requires_grad=False)


class _Attn(nn.Module):
    def __init__(self, c, r):
        super().__init__()
        D, hd = c.hidden_size, c.head_dim
        self.q_proj = _Proj(D, c.num_attention_heads * hd, r)
        self.k_proj = _Proj(D, c.num_key_value_heads * hd, r)
        self.v_proj = _Proj(D, c.num_key_value_heads * hd, r)
        self.o_proj = _Proj(c.num_attention_heads * hd, D, r)
        self.q_norm = _Norm(hd)
        self.k_norm = _Norm(hd)


class _MLP(nn.Module):
    def __init__(self, c, r):
        super().__init__()
        self.gate_proj = _Proj(c.hidden_size, c.intermediate_size, r)
        self.up_proj = _Proj(c.hidden_size, c.intermediate_size, r)
        self.down_proj = _Proj(c.intermediate_size, c.hidden_size, r)


class _Layer(nn.Module):
    def __init__(self, c, r):
        super().__init__()
        self.self_attn = _Attn(c, r)
        self.mlp = _MLP(c, r)
        self.input_layernorm = _Norm(c.hidden_size)
        self.post_attention_layernorm = _Norm(c.hidden_size)


class _TokenTable(nn.Module):
    def __init__(self, v, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(v, d), requires_grad=False)


# UNIREC_SWIGLU_FWD_FUSED=1 runs SwiGLU forward as the up projection's epilogue (ur_gemm swiglu_gate) instead of its own launch.
# Measured neutral on the joint step (115.1 vs 115.5 seq/s on one box, alternating runs: the epilogue's extra gate read and act
# write are not overlapped with MFMA work at one workgroup per CU, and the stand-alone kernel already streams at 5.4 TB/s), so
# the separate launch stays the default; the backward fusion (swiglu_gu), which removes 6 of 15 activation passes, is always on.
# (Needs the per-adapter launches: it has no effect unless UNIREC_MERGE_PROJ=0 as well.)
_FUSE_SWIGLU_FWD = os.environ.get("UNIREC_SWIGLU_FWD_FUSED", "0") == "1"


# UNIREC_MERGE_PROJ=0 (lab): one projection launch per LoRA adapter instead of the merged q|k|v and gate|up launches
_MERGE_PROJ = os.environ.get("UNIREC_MERGE_PROJ", "1") != "0"
# UNIREC_FUSE_NORM_LORA=0 (lab): RMSNorm forward and the q|k|v / gate|up adapters' down projection as two kernels again
_FUSE_NORM_LORA = os.environ.get("UNIREC_FUSE_NORM_LORA", "1") != "0"
# UNIREC_FUSE_QK_ROPE=0 (lab): q/k-norm + RoPE as their own pass over the raw q|k|v again (the fused form needs the persistent GEMM:
# >= 128 output tiles, S >= 256, head_dim 128; smaller launches take the separate pass anyway)
_FUSE_QK_ROPE = os.environ.get("UNIREC_FUSE_QK_ROPE", "1") != "0"
_QK_FUSE_MAX_RATIO = 4.0      # largest |w_d| / |w_{d+64}| spread of a rotate-half pair of the q / k norm weights under which the fused epilogue is used
# UNIREC_FUSE_SWIGLU_GEMM=0 (lab): SwiGLU forward as its own pass over gate|up again (the fused form rides in the merged gate|up launch on
# the persistent GEMM: interleaved weight rows put gate and up of a feature into one lane; the down adapter's t = dropout(act) A^T is
# then a lora_project pass over act)
_FUSE_SWIGLU_GEMM = os.environ.get("UNIREC_FUSE_SWIGLU_GEMM", "1") != "0"
# UNIREC_FUSE_SWIGLU_LORA=0 (lab): SwiGLU forward and the down_proj adapter's down projection as two kernels again
_FUSE_SWIGLU_LORA = os.environ.get("UNIREC_FUSE_SWIGLU_LORA", "1") != "0"


def _split_k(red, out_rows, out_cols):
    tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128)
    return int(max(1, min(1024 // max(tiles, 1), red // 512, 128)))


class _JointFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, item_tokens16, input_ids, mask_u8, first_special_id, anchor):
        pooled, saved = model._forward_impl(item_tokens16, input_ids, mask_u8, first_special_id)
        ctx.model, ctx.saved = model, saved
        return pooled

    @staticmethod
    def backward(ctx, d_pooled):
        d_tok = ctx.model._backward_impl(ctx.saved, d_pooled)
        ctx.saved = None
        return None, d_tok, None, None, None, None


class Qwen3LoRAModel(nn.Module):
    """Frozen Qwen3 decoder with LoRA adapters on all 7 projections."""

    def __init__(self, config: Qwen3Config, use_lora=True):
        super().__init__()
        self.config = config
        r = config.lora_r if use_lora else 0
        self.use_lora = r > 0
        self.embed_tokens = _TokenTable(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([_Layer(config, r) for _ in range(config.num_hidden_layers)])
        self.norm = _Norm(config.hidden_size)
        self._pack = None
        self._frozen = None
        self._rope = None
        self.grad_ready_hook = None
        self.lora_seed = 0x5EED        # base seed of the LoRA dropout masks (per run; identical on every data-parallel rank)
        # index of this forward's first sequence in the GLOBAL minibatch = dp_rank * B (unirec_amd.dp.set_dp_rank) unless
        # sample_offset is given: masks are keyed on the global token row (first * S + local row), so a data-parallel rank
        # draws exactly the masks its sequences would get in a single-process run over the whole minibatch
        self.dp_rank = 0
        self.sample_offset = None
        self._lora_step = 0            # forward calls with dropout so far: every step draws new masks
        self._bcomb = None             # block-diagonal LoRA B operands of the merged q|k|v and gate|up launches
        self._bits_stream = None       # side stream + planes of prefetch_lora_bits
        self._bits_pre = None
        self.recompute_mlp = os.environ.get("UNIREC_RECOMPUTE_MLP", "0") == "1"   # drop gate|up and act after the forward, rebuild them in the backward
        self.keep_norm_outputs = True   # keep the two RMSNorm outputs per layer for the backward (memory for time); False recomputes them
        self.reset_parameters()

    def reset_parameters(self, lora_b_std=0.0):
        """Base N(0, initializer_range) (weights are random: no network for the checkpoint); LoRA A
        kaiming-uniform(a=sqrt(5)), B zeros -- peft's default init.  lora_b_std > 0 draws B ~ N(0, std)
        so the LoRA path is numerically exercised."""
        std = self.config.initializer_range
        for n, p in self.named_parameters():
            if n.endswith("lora_A.weight"):
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            elif n.endswith("lora_B.weight"):
                if lora_b_std > 0:
                    nn.init.normal_(p, std=lora_b_std)
                else:
                    nn.init.zeros_(p)
            elif n.endswith("norm.weight") or n.endswith("layernorm.weight"):
                nn.init.ones_(p)
            else:
                nn.init.normal_(p, std=std)

    def resize_token_embeddings(self, new_size):
        """train_item_individual_token_joint.py:112-119 (rows for the added special tokens)."""
        old = self.embed_tokens.weight.data
        if new_size == old.shape[0]:
            return
        new = torch.empty(new_size, old.shape[1], dtype=old.dtype, device=old.device).normal_(std=self.config.initializer_range)
        n = min(new_size, old.shape[0])
        new[:n] = old[:n]
        self.embed_tokens.weight = nn.Parameter(new, requires_grad=False)
        self.config.vocab_size = new_size
        self._frozen = None

    def load_base_weights(self, state_dict, strict=True):
        """Frozen base weights from a Qwen3Model / Qwen3-Embedding state_dict (HF key names, optional ``model.`` or
        ``base_model.model.`` prefix).  A checkpoint vocabulary smaller than the (resized) table fills the first rows and
        leaves the added special-token rows alone -- ``load_state_dict(strict=False)`` cannot do that: it raises on the
        shape mismatch.  LoRA tensors in the dict are loaded too.  strict: every base tensor must be present."""
        sd = {}
        for k, v in state_dict.items():
            for pre in ("base_model.model.", "model."):
                if k.startswith(pre):
                    k = k[len(pre):]
            sd[k] = v
        own = dict(self.named_parameters())
        missing, unexpected = [], [k for k in sd if k not in own and k != "lm_head.weight"]
        with torch.no_grad():
            for k, p in own.items():
                if k not in sd:
                    if ".lora_" not in k:
                        missing.append(k)
                    continue
                v = sd[k]
                if k == "embed_tokens.weight" and v.shape[0] != p.shape[0]:
                    if v.shape[0] > p.shape[0] or v.shape[1] != p.shape[1]:
                        raise RuntimeError(f"embed_tokens.weight: checkpoint {tuple(v.shape)} does not fit the table {tuple(p.shape)}")
                    p[:v.shape[0]].copy_(v.to(p.device, p.dtype))
                    continue
                if tuple(v.shape) != tuple(p.shape):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(p.shape)}")
                p.copy_(v.to(p.device, p.dtype))
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_base_weights: missing {missing[:5]}{'...' if len(missing) > 5 else ''}, unexpected {unexpected[:5]}")
        self._frozen = None
        if self._pack is not None:
            self._pack.mark_dirty()
        return missing, unexpected

    # ---- parameter plumbing ---------------------------------------------------------------------
    def lora_named_parameters(self):
        named = dict(self.named_parameters())
        order = []
        for i in range(self.config.num_hidden_layers):
            lp = f"layers.{i}."
            for grp in (("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"), ("self_attn.o_proj",),
                        ("mlp.gate_proj", "mlp.up_proj"), ("mlp.down_proj",)):
                order += [lp + g + ".lora_A.weight" for g in grp]
            for g in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj",
                      "mlp.up_proj", "mlp.down_proj"):
                order.append(lp + g + ".lora_B.weight")
        return [(n, named[n]) for n in order]

    def _ensure_pack(self, device):
        if not self.use_lora:
            return None
        if self._pack is None or not self._pack.is_current() or self._pack.device != norm_device(device):
            self._pack = ParamPack(self.lora_named_parameters(), device)
        return self._pack

    @property
    def pack(self):
        return self._pack

    def _ensure_frozen(self, device):
        """bf16 operands of the frozen base weights, fused per layer: [q|k|v], o, [gate|up], down."""
        # keyed on EVERY frozen tensor (storage + in-place version): loading any of them after a forward has run must not
        # leave stale bf16 / transposed copies behind
        key = (str(device), _FUSE_QK_ROPE, _FUSE_SWIGLU_GEMM) + tuple((p.data_ptr(), p._version) for n, p in self.named_parameters() if ".lora_" not in n)
        if self._frozen is not None and self._frozen["key"] == key:
            return self._frozen
        fz = {"key": key, "layers": []}
        dev = torch.device(device)

        def c16(t):
            return hip.cast_f32_to_bf16(t.detach().to(dev, F32).contiguous())
        fz["embed"] = c16(self.embed_tokens.weight)
        fz["norm"] = self.norm.weight.detach().to(dev, F32).contiguous()
        for lyr in self.layers:
            a, m = lyr.self_attn, lyr.mlp
            wqkv = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0)
            wgu = torch.cat([m.gate_proj.weight, m.up_proj.weight], 0)
            rp = self._qk_row_perm(dev)
            fz["layers"].append({
                "qkv": c16(wqkv),
                # q|k|v with the rows of every q / k head in the paired order of the fused q/k-norm + RoPE epilogue (hip.qkrope_perm)
                # (only when that path can be selected at all: the duplicates cost ~0.6 GB at Qwen3-0.6B)
                "qkvP": c16(wqkv.to(dev)[rp]) if (rp is not None and _FUSE_QK_ROPE) else None,
                # gate|up with 128-row blocks of gate and up interleaved (hip.swiglu_pair_rows): the paired SwiGLU forward epilogue
                "guP": c16(wgu.to(dev)[self._swiglu_rows(m.gate_proj.weight.shape[0], dev)]) if (_FUSE_SWIGLU_GEMM and m.gate_proj.weight.shape[0] % 128 == 0) else None,
                "o": c16(a.o_proj.weight), "gu": c16(wgu), "d": c16(m.down_proj.weight),
                # frozen => one-time transposed copies, so every dX GEMM is K-contiguous on both operands
                "qkvT": c16(wqkv.t()), "oT": c16(a.o_proj.weight.t()), "guT": c16(wgu.t()), "dT": c16(m.down_proj.weight.t()),
                "qn": a.q_norm.weight.detach().to(dev, F32).contiguous(), "kn": a.k_norm.weight.detach().to(dev, F32).contiguous(),
                "ln1": lyr.input_layernorm.weight.detach().to(dev, F32).contiguous(),
                "ln2": lyr.post_attention_layernorm.weight.detach().to(dev, F32).contiguous()})
        # The fused epilogue's backward recovers the normalised rows from the bf16 roped outputs (x^ = R^T(o) / w): the rounding
        # error of a rotate-half pair (d, d + 64) scales with the larger of its two entries and is divided by each one's own weight,
        # i.e. it is amplified by max(|w_d|, |w_{d+64}|) / min(...).  Fuse only while that conditioning stays small; a checkpoint
        # with widely spread norm weights takes the separate q/k-norm + RoPE pass (which keeps the raw q, k).
        def well_conditioned(w):
            a = w.abs().view(2, -1)
            lo, hi = torch.minimum(a[0], a[1]), torch.maximum(a[0], a[1])
            return bool((lo > 0).all()) and bool((hi <= _QK_FUSE_MAX_RATIO * lo).all())
        fz["qk_norm_fusable"] = all(well_conditioned(l["qn"]) and well_conditioned(l["kn"]) for l in fz["layers"])
        self._frozen = fz
        return fz

    def _swiglu_rows(self, I, device):
        """hip.swiglu_pair_rows(I) on `device`, built once (a host -> device copy per forward would block the host on the stream)"""
        key = (str(device), int(I))
        if getattr(self, "_sp", None) is None or self._sp[0] != key:
            self._sp = (key, hip.swiglu_pair_rows(I).to(device))
        return self._sp[1]

    def _qk_row_perm(self, device):
        """Row order of the paired q|k|v operand: inside every q / k head tile column c holds feature qkrope_perm[c]; v rows stay."""
        c = self.config
        if c.head_dim != 128:
            return None
        key = (str(device), c.num_attention_heads, c.num_key_value_heads)
        if getattr(self, "_rp", None) is None or self._rp[0] != key:
            perm = hip.qkrope_perm(128)
            nqk = c.num_attention_heads + c.num_key_value_heads
            rows = torch.cat([(torch.arange(nqk)[:, None] * 128 + perm[None, :]).reshape(-1),
                              torch.arange(nqk * 128, (nqk + c.num_key_value_heads) * 128)])
            self._rp = (key, rows.to(device))
        return self._rp[1]

    def _rope_tables(self, S, device):
        if self._rope is None or self._rope[0] != (S, str(device)):
            self._rope = ((S, str(device)), hip.rope_table(S, self.config.head_dim, float(self.config.rope_theta), device))
        return self._rope[1]

    def peft_state_dict(self):
        """LoRA tensors under peft's adapter key names (save_pretrained compatibility, :183-200)."""
        out = {}
        for n, p in self.named_parameters():
            if ".lora_A." in n or ".lora_B." in n:
                out["base_model.model." + n] = p.detach().cpu()
        return out

    # ---- public forward ---------------------------------------------------------------------------
    def forward_pooled(self, input_ids, attention_mask=None, item_tokens16=None, first_special_id=0):
        """input_ids int64 [B,S]; attention_mask [B,S] or None; item_tokens16 [B,T,D] bf16 (Q-Former
        query tokens to inject at ids first_special_id + t) or None -> mean-pooled f32 [B,D]."""
        if not input_ids.is_cuda:
            raise hip._lib.UniRecHipError("Qwen3LoRAModel runs on the MI355X only (no CPU fallback in the product path)")
        B, S = input_ids.shape
        mask_u8 = None if attention_mask is None else (attention_mask != 0).to(torch.uint8).contiguous()
        if item_tokens16 is None:
            item_tokens16 = torch.zeros((B, 0, self.config.hidden_size), dtype=BF16, device=input_ids.device)
        # the LoRA gradients are side effects of this node's backward: an anchor input keeps the node
        # alive even when the injected tokens do not require grad (frozen / absent Q-Former)
        if not torch.is_grad_enabled():
            # inference (MRREvaluator, CatalogEvaluator, token caches): nothing is kept for a backward, so the peak is one
            # layer's activations instead of all of them (178 GB at B=64, S=2048 in training)
            return self._forward_impl(item_tokens16.detach(), input_ids.contiguous(), mask_u8, int(first_special_id), keep=False)[0]
        anchor = torch.zeros(1, device=input_ids.device, requires_grad=True) if self.use_lora else None
        return _JointFn.apply(self, item_tokens16, input_ids.contiguous(), mask_u8, int(first_special_id), anchor)

    # ---- implementation ---------------------------------------------------------------------------
    def _lora(self, pack, name):
        return None if pack is None else pack.w16(name)

    def _drop_p(self):
        """peft applies lora_dropout only in training mode (nn.Dropout inside every LoraLayer)."""
        return float(self.config.lora_dropout) if (self.training and self.use_lora) else 0.0

    def lora_dropout_seed(self, step, layer, group):
        """Seed of the dropout masks of one adapter group (the adapters that share an input: 0 = q|k|v, 1 = o,
        2 = gate|up, 3 = down) -- a pure function of (base seed, step, layer, group); tests regenerate the bit
        planes from it (hip.lora_dropout_bits) and feed the unpacked masks to the oracle."""
        # splitmix64 of the combined index: the seeds of neighbouring (step, layer, group) differ in every bit, not in a
        # few low ones (the mask streams of two adapter groups must be unrelated, as peft's per-module nn.Dropout are)
        z = (int(self.lora_seed) * 0x9E3779B97F4A7C15 + int(step) * 0xD1B54A32D192ED03 + (layer * 8 + group + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return (z ^ (z >> 31)) & 0x7FFFFFFFFFFFFFFF

    def first_sample(self, B):
        return int(self.sample_offset) if self.sample_offset is not None else int(self.dp_rank) * int(B)

    def _bits_groups(self):
        """(input width, adapters sharing it) of the four adapter groups of a layer: q|k|v, o, gate|up, down."""
        c = self.config
        return ((c.hidden_size, 3), (c.num_attention_heads * c.head_dim, 1), (c.hidden_size, 2), (c.intermediate_size, 1))

    def prefetch_lora_bits(self, M, device, row0=0):
        """Generate the NEXT forward's LoRA dropout bit planes (all layers, all adapter groups) on a side stream.  The planes
        are pure functions of (seed, step, layer, group) -- nothing on the main stream feeds them -- so the caller starts
        this before the item Q-Former's forward, whose small launches leave most of the chip idle.  From the second step on the
        planes are already there: the decoder's backward regenerates them for the following step the moment it has finished with
        this step's (`_prefetch_next_step`), under the Q-Former's backward.  The planes are allocated on the caller's stream (the
        caching allocator keys blocks by stream) and written on the side stream after it has caught up with the caller's stream."""
        p = self._drop_p()
        if p <= 0.0 or self.config.lora_r != 16 or not torch.is_grad_enabled():
            self._release_prefetched(device)
            return
        pre = self._bits_pre
        if pre is not None and pre["step"] == self._lora_step and pre["M"] == M and pre["row0"] == int(row0) and pre["p"] == p:
            return                                 # made under the previous step's backward
        self._release_prefetched(device)
        main = torch.cuda.current_stream(device)
        if self._bits_stream is None:
            self._bits_stream = torch.cuda.Stream(device=device)
        planes = {(i, g): torch.empty((nad, M, hip.lora_bits_ld(W)), dtype=torch.uint8, device=device)
                  for i in range(self.config.num_hidden_layers) for g, (W, nad) in enumerate(self._bits_groups())}
        # token-packed copies for the backward's token reductions (hip.lora_reduce's ring kernel), made on the side stream as well
        packed = {}
        if M % 128 == 0 and os.environ.get("UNIREC_BITS_T", "1") != "0":      # (lab switch: 0 = no token-packed copies, the register-staged reduction)
            packed = {(i, g): torch.empty((nad, M // 32, hip.lora_bits_t_ld(W)), dtype=torch.int32, device=device)
                      for i in range(self.config.num_hidden_layers) for g, (W, nad) in enumerate(self._bits_groups()) if W % 64 == 0}
        self._bits_stream.wait_stream(main)
        self._generate_bits(planes, packed, self._lora_step, M, device, int(row0), p)

    def _release_prefetched(self, device=None):
        """Drop a prefetched plane set that no forward will consume (last partial batch of an epoch, train -> eval, a different
        micro-batch).  The planes were allocated on the caller's stream but are WRITTEN on the side stream: the caller's stream
        first waits for the generator's last event, so the caching allocator cannot hand the blocks to main-stream tensors while
        the side stream is still writing masks into them."""
        pre, self._bits_pre = self._bits_pre, None
        if pre is not None and pre.get("event_t") is not None:
            torch.cuda.current_stream(device).wait_event(pre["event_t"])

    def _generate_bits(self, planes, packed, step, M, device, row0, p):
        side = self._bits_stream
        with torch.cuda.stream(side):
            # one event per layer: the forward of layer i waits for ITS planes only, so the generator keeps running under the first
            # layers of the decoder when the Q-Former's forward is shorter than it; the token-packed copies (backward only) come last
            events = []
            for i in range(self.config.num_hidden_layers):
                for g, (W, nad) in enumerate(self._bits_groups()):
                    hip.lora_dropout_bits(self.lora_dropout_seed(step, i, g), p, M, W, nad, device, out=planes[(i, g)], row0=row0)
                events.append(side.record_event())
            for (i, g), bt in packed.items():
                hip.lora_bits_transpose(planes[(i, g)], self._bits_groups()[g][0], out=bt)
            ev_t = side.record_event()
        self._bits_pre = {"step": step, "M": M, "planes": planes, "packed": packed, "events": events, "event_t": ev_t, "row0": int(row0), "p": p}

    def _prefetch_next_step(self, cur, device):
        """Called by the decoder's backward when it has consumed this step's planes: the next step's are written into the SAME buffers
        on the side stream (which first waits for the main stream to get here), under the Q-Former's backward and the optimizer."""
        if os.environ.get("UNIREC_BITS_NEXT", "1") == "0" or not self.training:
            return
        if self.sample_offset is not None:
            # micro-batching / explicit shards: the next forward's first row is not this one's, so planes written here would only be
            # thrown away (and every backward would generate a full set for nothing)
            return
        self._bits_stream.wait_stream(torch.cuda.current_stream(device))
        self._generate_bits(cur["planes"], cur["packed"], self._lora_step, cur["M"], device, cur["row0"], cur["p"])

    def _lora_bcomb(self, pack, device):
        """Second-K-range operands of the merged projection launches: y[q|k|v] = h W^T + [t_q|t_k|t_v] Bc^T with
        Bc [N, 16 nad] block-diagonal (rows of adapter a carry B_a in columns 16a..16a+15, zeros elsewhere -- exact zeros, so
        the sum is bit-identical to the per-adapter launches).  q|k|v and gate|up each leave as ONE launch that reads its
        input once (1.09 -> 1.01 ms and 1.58 -> 1.49 ms per layer at C4).  The zero blocks are written once; each step
        refreshes the diagonal blocks from the bf16 shadow with one multi-tensor copy."""
        c = self.config
        r, NQ, NKV, I = c.lora_r, c.num_attention_heads * c.head_dim, c.num_key_value_heads * c.head_dim, c.intermediate_size
        if self._bcomb is None or self._bcomb["dev"] != torch.device(device) or self._bcomb["pack"] is not pack:
            qkv = torch.zeros((c.num_hidden_layers, NQ + 2 * NKV, 3 * r), dtype=BF16, device=device)
            gu = torch.zeros((c.num_hidden_layers, 2 * I, 2 * r), dtype=BF16, device=device)
            dst, src = [], []
            for i in range(c.num_hidden_layers):
                lp = f"layers.{i}."
                row = 0
                for j, (pn, n) in enumerate((("q", NQ), ("k", NKV), ("v", NKV))):
                    dst.append(qkv[i, row:row + n, j * r:(j + 1) * r]); src.append(pack.w16(lp + f"self_attn.{pn}_proj.lora_B.weight")); row += n
                for j, pn in enumerate(("gate", "up")):
                    dst.append(gu[i, j * I:(j + 1) * I, j * r:(j + 1) * r]); src.append(pack.w16(lp + f"mlp.{pn}_proj.lora_B.weight"))
            self._bcomb = {"dev": torch.device(device), "pack": pack, "qkv": qkv, "gu": gu, "dst": dst, "src": src}
        torch._foreach_copy_(self._bcomb["dst"], self._bcomb["src"])
        rp = self._qk_row_perm(device) if _FUSE_QK_ROPE else None
        self._bcomb["qkvP"] = self._bcomb["qkv"].index_select(1, rp) if rp is not None else None      # rows paired like fz["qkvP"]
        self._bcomb["guP"] = (self._bcomb["gu"].index_select(1, self._swiglu_rows(I, device)) if (_FUSE_SWIGLU_GEMM and I % 128 == 0) else None)
        return self._bcomb["qkv"], self._bcomb["gu"]

    def _lora_transposes(self, pack):
        """A^T (per adapter group) and B^T (per adapter) of every layer for the backward, refreshed by ONE launch per
        backward (308 separate 5 us launches before).  The table is built once per pack: the sources are views of the pack's
        persistent bf16 shadow."""
        bt = getattr(self, "_lt", None)
        if bt is None or bt[0] is not pack:
            names, srcs = [], []
            for i in range(self.config.num_hidden_layers):
                lp = f"layers.{i}."
                for grp in (("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"), ("self_attn.o_proj",),
                            ("mlp.gate_proj", "mlp.up_proj"), ("mlp.down_proj",)):
                    an = tuple(lp + g + ".lora_A.weight" for g in grp)
                    names.append(an)
                    srcs.append(pack.fused16(list(an)) if len(an) > 1 else pack.w16(an[0]))
                    for g in grp:
                        names.append(lp + g + ".lora_B.weight")
                        srcs.append(pack.w16(lp + g + ".lora_B.weight"))
            bt = (pack, names, hip.BatchedTranspose(srcs))
            self._lt = bt
        outs = bt[2].run()
        return dict(zip(bt[1], outs))

    def _norm_lora_down(self, x, w, eps, a_names, pack, sc, seed, p, pre=None, row0=0):
        """(h, rstd, t, bits) = RMSNorm forward + _lora_down of the adapters that read h, as one kernel (ur_rmsnorm_lora_fwd)."""
        bits = pre
        if bits is None and p > 0.0:
            bits = hip.lora_dropout_bits(seed, p, x.shape[0], x.shape[1], len(a_names), x.device, row0=row0)
        h, rstd, t = hip.rmsnorm_lora_fwd(x, w, eps, [pack.w16(n) for n in a_names], alpha=sc / (1.0 - p), bits=bits)
        return h, rstd, t, bits

    def _lora_down(self, xin, a_names, pack, sc, seed, p, pre=None, row0=0):
        """(t, bits): t[M, nb*r] = s * dropout_j(x) A_j^T for the nb adapters that share the input x (one dropped-flag
        bit plane per adapter, generated once here and kept for the backward)."""
        if self.config.lora_r != 16:
            if p > 0.0:
                raise hip._lib.UniRecHipError("LoRA dropout is implemented for rank 16 (the reference's r) only")
            A = pack.fused16(a_names) if len(a_names) > 1 else pack.w16(a_names[0])
            return hip.gemm(xin, A, alpha=sc), None
        bits = pre
        if bits is None and p > 0.0:
            bits = hip.lora_dropout_bits(seed, p, xin.shape[0], xin.shape[1], len(a_names), xin.device, row0=row0)
        t = hip.lora_project(xin, [pack.w16(n) for n in a_names], alpha=sc / (1.0 - p), bits=bits)
        return t, bits

    def _forward_impl(self, item_tokens16, input_ids, mask_u8, first_special_id, keep=True):
        c = self.config
        dev = input_ids.device
        fz = self._ensure_frozen(dev)
        pack = self._ensure_pack(dev)
        bc_qkv = bc_gu = None
        if pack is not None:
            pack.refresh_shadow()
            if _MERGE_PROJ and c.lora_r == 16:
                bc_qkv, bc_gu = self._lora_bcomb(pack, dev)
        B, S = input_ids.shape
        D, I, nq, nkv, hd, r = c.hidden_size, c.intermediate_size, c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.lora_r
        NQ, NKV = nq * hd, nkv * hd
        M = B * S
        eps, sc = c.rms_norm_eps, (c.lora_alpha / c.lora_r if self.use_lora else 0.0)
        cos, sin = self._rope_tables(S, dev)
        T = item_tokens16.shape[1]
        tok = item_tokens16.detach().contiguous() if T > 0 else None
        x = hip.embed_inject_fwd(fz["embed"], input_ids, tok, first_special_id).view(M, D)
        pdrop = self._drop_p()
        step = self._lora_step
        if pdrop > 0.0:
            self._lora_step += 1
        saved = {"B": B, "S": S, "T": T, "ids": input_ids, "first": first_special_id, "mask": mask_u8, "layers": [],
                 "pdrop": pdrop, "step": step}
        row0 = self.first_sample(B) * S                # token rows that precede this shard in the global minibatch
        pre, self._bits_pre = self._bits_pre, None
        if pre is not None and not (pdrop > 0.0 and pre["step"] == step and pre["M"] == M and pre["row0"] == row0 and pack is not None):
            self._bits_pre = pre
            self._release_prefetched(dev)           # not this forward's planes: order their last writes before the blocks are reused
            pre = None
        if pre is not None:
            saved["bits_t"] = pre.get("packed", {})                      # (layer, group) -> token-packed copy for the backward
            saved["bits_t_event"] = pre["event_t"]
            saved["bits_pre"] = pre                                      # (its buffers take the next step's planes after the backward)
            pre_events = pre["events"]                                   # planes prefetched on the side stream, one event per layer
            pre = pre["planes"]
        else:
            pre, pre_events = None, None
        bp = (lambda i, g: pre[(i, g)]) if pre is not None else (lambda i, g: None)
        fuse_norm = _FUSE_NORM_LORA and pack is not None and c.lora_r == 16 and D == 1024
        for i, fl in enumerate(fz["layers"]):
            lp = f"layers.{i}."
            L = {"x": x}
            if pre_events is not None:
                # (lab switch UNIREC_BITS_ONE_EVENT=1: wait for every layer's planes before the first layer, the former behaviour)
                torch.cuda.current_stream(dev).wait_event(pre_events[-1] if (i == 0 and os.environ.get("UNIREC_BITS_ONE_EVENT") == "1") else pre_events[i])
            # q/k-norm + RoPE inside the q|k|v launch (the raw q, k are never written or re-read) when the persistent GEMM takes it
            fuse_rope = (_FUSE_QK_ROPE and hd == 128 and fl["qkvP"] is not None and fz["qk_norm_fusable"] and (pack is None or bc_qkv is not None) and
                         hip.gemm_qkrope_supported(M, NQ + 2 * NKV, D, 3 * r if pack is not None else 0, S, NQ, NKV, dev))
            qkv = None if fuse_rope else torch.empty((M, NQ + 2 * NKV), dtype=BF16, device=dev)
            if fuse_norm:      # RMSNorm + the q|k|v adapters' down projection in one pass over x (h is written once, never re-read by a projection kernel)
                h, rstd1, t_qkv, L["bits_qkv"] = self._norm_lora_down(x, fl["ln1"], eps, [lp + f"self_attn.{p}_proj.lora_A.weight" for p in "qkv"],
                                                                      pack, sc, self.lora_dropout_seed(step, i, 0), pdrop, bp(i, 0), row0=row0)
            else:
                h, rstd1 = hip.rmsnorm_fwd(x, fl["ln1"], eps)
            if pack is not None:
                if not fuse_norm:
                    t_qkv, L["bits_qkv"] = self._lora_down(h, [lp + f"self_attn.{p}_proj.lora_A.weight" for p in "qkv"], pack, sc,
                                                           self.lora_dropout_seed(step, i, 0), pdrop, bp(i, 0), row0=row0)     # [M,3r] = s * dropout(h) A^T
                if fuse_rope:
                    q_r, k_r, v2, rstd_qk = hip.gemm_qkv_rope(h, fl["qkvP"], fl["qn"], fl["kn"], cos, sin, S, NQ, NKV, eps, R2=t_qkv, S2=self._bcomb["qkvP"][i])
                elif bc_qkv is not None:
                    hip.gemm(h, fl["qkv"], out=qkv, R2=t_qkv, S2=bc_qkv[i])        # one launch, block-diagonal B
                else:
                    col = 0
                    for j, (p, n) in enumerate((("q", NQ), ("k", NKV), ("v", NKV))):
                        hip.gemm(h, fl["qkv"][col:col + n], out=qkv[:, col:col + n], R2=t_qkv[:, j * r:(j + 1) * r],
                                 S2=pack.w16(lp + f"self_attn.{p}_proj.lora_B.weight"))
                        col += n
                L["t_qkv"] = t_qkv
            elif fuse_rope:
                q_r, k_r, v2, rstd_qk = hip.gemm_qkv_rope(h, fl["qkvP"], fl["qn"], fl["kn"], cos, sin, S, NQ, NKV, eps)
            else:
                hip.gemm(h, fl["qkv"], out=qkv)
            if fuse_rope:
                v4 = v2.view(B, S, nkv, hd)
                L.update(q_r=q_r, k_r=k_r, rstd_qk=rstd_qk)
            else:
                q_r, k_r = hip.qknorm_rope_fwd(qkv, fl["qn"], fl["kn"], cos, sin, S, nq, nkv, hd, eps)
                v4 = qkv[:, NQ + NKV:].view(B, S, nkv, hd)
            # the attention output's rows are padded by 64 columns when their length is a power of two: the kernels that stream it
            # in 128-byte column chunks (the o_proj adapter's projection and its token reduction) otherwise keep every request in
            # flight on the same bytes of a 4 KiB-strided row, i.e. on a few memory channels (lora_project_ring_kernel: 141 -> 104 us)
            att_out = None
            if (NQ & (NQ - 1)) == 0 and NQ >= 1024 and os.environ.get("UNIREC_PAD_ATT", "1") != "0":
                att_out = torch.empty((M, NQ + 64), dtype=BF16, device=dev)[:, :NQ].view(B, S, nq, hd)
            att, actx = hip.attn_fwd(q_r.view(B, S, nq, hd), k_r.view(B, S, nkv, hd), v4, causal=True, key_mask=mask_u8, out=att_out)
            att2 = att.view(M, NQ)
            if pack is not None:
                t_o, L["bits_o"] = self._lora_down(att2, [lp + "self_attn.o_proj.lora_A.weight"], pack, sc, self.lora_dropout_seed(step, i, 1), pdrop, bp(i, 1), row0=row0)
                x2 = hip.gemm(att2, fl["o"], residual=x, R2=t_o, S2=pack.w16(lp + "self_attn.o_proj.lora_B.weight"))
                L["t_o"] = t_o
            else:
                x2 = hip.gemm(att2, fl["o"], residual=x)
            gu = torch.empty((M, 2 * I), dtype=BF16, device=dev)
            if fuse_norm:
                h2, rstd2, t_gu, L["bits_gu"] = self._norm_lora_down(x2, fl["ln2"], eps, [lp + "mlp.gate_proj.lora_A.weight", lp + "mlp.up_proj.lora_A.weight"],
                                                                     pack, sc, self.lora_dropout_seed(step, i, 2), pdrop, bp(i, 2), row0=row0)
            else:
                h2, rstd2 = hip.rmsnorm_fwd(x2, fl["ln2"], eps)
            if pack is not None:
                if not fuse_norm:
                    t_gu, L["bits_gu"] = self._lora_down(h2, [lp + "mlp.gate_proj.lora_A.weight", lp + "mlp.up_proj.lora_A.weight"], pack, sc,
                                                         self.lora_dropout_seed(step, i, 2), pdrop, bp(i, 2), row0=row0)
                # gate first; the up projection's epilogue then reads the gate tile and writes act = silu(gate) * up beside up
                fused = _FUSE_SWIGLU_FWD and bc_gu is None
                act = torch.empty((M, I), dtype=BF16, device=dev) if fused else None
                pair = (_FUSE_SWIGLU_GEMM and bc_gu is not None and fl["guP"] is not None and self._bcomb.get("guP") is not None and
                        hip.gemm_swiglu_paired_supported(M, I, D, 2 * r, dev))
                if pair:            # ONE launch: gate|up (standard order, for the backward) and act = silu(gate) * up from its registers
                    act = torch.empty((M, I), dtype=BF16, device=dev)
                    hip.gemm(h2, fl["guP"], out=gu, R2=t_gu, S2=self._bcomb["guP"][i], swiglu_paired=act)
                elif bc_gu is not None:
                    hip.gemm(h2, fl["gu"], out=gu, R2=t_gu, S2=bc_gu[i])            # one launch, block-diagonal B
                for j, p in enumerate(("gate", "up") if bc_gu is None else ()):
                    hip.gemm(h2, fl["gu"][j * I:(j + 1) * I], out=gu[:, j * I:(j + 1) * I], R2=t_gu[:, j * r:(j + 1) * r],
                             S2=pack.w16(lp + f"mlp.{p}_proj.lora_B.weight"), swiglu_fwd=(gu[:, :I], act) if (j == 1 and fused) else None)
                fuse_act = _FUSE_SWIGLU_LORA and not fused and not pair and r == 16 and I % 128 == 0
                if fuse_act:      # act and t_d = s * dropout(act) A_d^T from one pass over gate|up (ur_swiglu_lora_fwd)
                    bits_d = bp(i, 3)
                    if bits_d is None and pdrop > 0.0:
                        bits_d = hip.lora_dropout_bits(self.lora_dropout_seed(step, i, 3), pdrop, M, I, 1, dev, row0=row0)
                    act, t_d = hip.swiglu_lora_fwd(gu, I, pack.w16(lp + "mlp.down_proj.lora_A.weight"), alpha=sc / (1.0 - pdrop), bits=bits_d)
                    L["bits_d"] = bits_d
                elif not fused and not pair:
                    act = hip.swiglu_fwd(gu, I)
                L["t_gu"] = t_gu
            else:
                hip.gemm(h2, fl["gu"], out=gu)
                act = hip.swiglu_fwd(gu, I)
            if pack is not None:
                if not fuse_act:
                    t_d, L["bits_d"] = self._lora_down(act, [lp + "mlp.down_proj.lora_A.weight"], pack, sc, self.lora_dropout_seed(step, i, 3), pdrop, bp(i, 3), row0=row0)
                x3 = hip.gemm(act, fl["d"], residual=x2, R2=t_d, S2=pack.w16(lp + "mlp.down_proj.lora_B.weight"))
                L["t_d"] = t_d
            else:
                x3 = hip.gemm(act, fl["d"], residual=x2)
            L.update(rstd1=rstd1, qkv=qkv, actx=actx, att=att2, x2=x2, rstd2=rstd2, gu=gu, act=act)      # (qkv is None under the fused q/k-norm + RoPE epilogue)
            # (recompute_mlp = True: every layer; an int k: layers 0 .. k - 1 only -- as much memory as the shape needs, no more recomputation than that)
            if self.recompute_mlp and keep and (self.recompute_mlp is True or i < int(self.recompute_mlp)):
                # memory for time: gate|up and act (2 x [M, 3I] bf16 over the stack: 67 GB at C4) are dropped and rebuilt in the
                # backward by the very launch that made them (bit-identical: same kernel, same operands).
                how = "pair" if (pack is not None and pair) else ("merged" if (pack is not None and bc_gu is not None and not fused) else ("plain" if pack is None else None))
                if how is not None:
                    L.update(gu=None, act=None, mlp_recompute=how)
            if self.keep_norm_outputs:       # 2 x [M,D] bf16 per layer (15 GB at C4) instead of two RMSNorm recomputes
                L.update(h=h, h2=h2)
            if keep:
                saved["layers"].append(L)
            x = x3
        last, rstd_f = hip.rmsnorm_fwd(x, fz["norm"], eps)
        pooled, _ = hip.mean_pool_fwd(last.view(B, S, D))
        if not keep:
            return pooled, None
        saved["xf"], saved["rstd_f"] = x, rstd_f
        return pooled, saved

    def _backward_impl(self, saved, d_pooled):
        c = self.config
        fz = self._frozen
        pack = self._pack if self.use_lora else None
        B, S, T = saved["B"], saved["S"], saved["T"]
        D, I, nq, nkv, hd, r = c.hidden_size, c.intermediate_size, c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.lora_r
        NQ, NKV = nq * hd, nkv * hd
        M = B * S
        eps, sc = c.rms_norm_eps, (c.lora_alpha / c.lora_r if self.use_lora else 0.0)
        dev = d_pooled.device
        cos, sin = self._rope_tables(S, dev)
        dlast = hip.mean_pool_bwd(d_pooled.contiguous().to(F32), S).view(M, D)
        dx = hip.rmsnorm_bwd(dlast, saved["xf"], fz["norm"], saved["rstd_f"])
        touched = []

        pdrop, step = saved["pdrop"], saved["step"]
        lt = self._lora_transposes(pack) if (pack is not None and r == 16) else None      # name(s) -> transposed bf16 operand

        packed_bits = saved.get("bits_t", {})
        if saved.get("bits_t_event") is not None:
            torch.cuda.current_stream(dev).wait_event(saved["bits_t_event"])

        def lora_grads(dy, t, xin, a_names, b_specs, bits, group=None):
            """dB_p = dy_p^T t_p ; tb = s * dy B ; dA_p = tb_p^T dropout_p(x).  Returns tb [M, len(b)*r] (bf16).
            group = (layer, adapter group): the key of the prefetched token-packed flags."""
            nb = len(b_specs)
            touched.extend([b for b, _, _ in b_specs] + list(a_names))
            if r != 16:           # generic tiles (no dropout: _lora_down refused it)
                tb = torch.empty((M, nb * r), dtype=BF16, device=dev)
                for j, (bname, c0, n) in enumerate(b_specs):
                    dyp = dy[:, c0:c0 + n]
                    hip.gemm(dyp, t[:, j * r:(j + 1) * r], r_kcontig=False, s_kcontig=False, out=pack.g32(bname), split_k=_split_k(M, n, r))
                    hip.gemm(dyp, pack.w16(bname), s_kcontig=False, out=tb[:, j * r:(j + 1) * r], alpha=sc)
                gA = pack.fusedg(a_names) if len(a_names) > 1 else pack.g32(a_names[0])
                hip.gemm(tb, xin, r_kcontig=False, s_kcontig=False, out=gA, split_k=_split_k(M, nb * r, xin.shape[1]))
                return tb
            cols = [(c0, n) for _, c0, n in b_specs]
            bnames = [b for b, _, _ in b_specs]
            gB = pack.fusedg(bnames) if nb > 1 else pack.g32(bnames[0])            # [sum n, r]: adapter ranges in order
            tb = hip.lora_bgrad(dy, t, [lt[b] for b in bnames], cols, gB, alpha=sc)     # dB and tb, dy read once
            gA = pack.fusedg(a_names) if len(a_names) > 1 else pack.g32(a_names[0])
            bits_t = packed_bits.get(group) if bits is not None else None
            if bits is not None and bits_t is None and M % 128 == 0 and xin.shape[1] % 64 == 0 and os.environ.get("UNIREC_BITS_T", "1") != "0":
                bits_t = hip.lora_bits_transpose(bits, xin.shape[1])          # (no prefetch this step: made here)
            hip.lora_reduce(xin, tb, gA, nad=len(a_names), alpha=1.0 / (1.0 - pdrop), bits=bits, bits_t=bits_t)
            return tb

        def dx_gemm(dy, wT, tb, a_names, bits, swiglu=None):
            """dx = dy W + sum_j mask_j * (tb_j A_j): the adapters' part joins the main reduction when there is no
            dropout, and is a masked rank-r epilogue (ur_gemm drop_bits) when there is.  swiglu = (gu, dgu): dx is d(act)
            and leaves the GEMM as dgate | dup (SwiGLU backward in the epilogue: d(act) is never stored)."""
            if lt is not None:
                AT = lt[tuple(a_names)]
            else:
                AT = hip.transpose_bf16(pack.fused16(a_names) if len(a_names) > 1 else pack.w16(a_names[0]))
            drop = (bits, pdrop, r) if bits is not None else None
            return hip.gemm(dy, wT, R2=tb, S2=AT, drop=drop, swiglu_bwd=swiglu)

        scratch = {}
        for i in reversed(range(len(fz["layers"]))):
            fl, L = fz["layers"][i], saved["layers"][i]
            lp = f"layers.{i}."
            x, x2, gu, qkv = L["x"], L["x2"], L["gu"], L["qkv"]
            # ---- MLP: x3 = x2 + down(silu(gate) * up)
            act = L["act"]
            h2 = L["h2"] if "h2" in L else hip.rmsnorm_fwd(x2, fl["ln2"], eps)[0]          # kept, or recomputed
            if L.get("mlp_recompute"):           # recompute_mlp: one [M, 2I] + [M, I] scratch pair serves every layer
                if "gu" not in scratch:
                    scratch["gu"] = torch.empty((M, 2 * I), dtype=BF16, device=dev)
                    scratch["act"] = torch.empty((M, I), dtype=BF16, device=dev)
                gu, act = scratch["gu"], scratch["act"]
                how = L["mlp_recompute"]
                if how == "pair":
                    hip.gemm(h2, fl["guP"], out=gu, R2=L["t_gu"], S2=self._bcomb["guP"][i], swiglu_paired=act)
                else:
                    if how == "merged":
                        hip.gemm(h2, fl["gu"], out=gu, R2=L["t_gu"], S2=self._bcomb["gu"][i])
                    else:
                        hip.gemm(h2, fl["gu"], out=gu)
                    hip.swiglu_fwd(gu, I, out=act)
            dgu = torch.empty_like(gu)
            if pack is not None:
                tb = lora_grads(dx, L["t_d"], act, [lp + "mlp.down_proj.lora_A.weight"], [(lp + "mlp.down_proj.lora_B.weight", 0, D)], L["bits_d"], group=(i, 3))
                dx_gemm(dx, fl["dT"], tb, [lp + "mlp.down_proj.lora_A.weight"], L["bits_d"], swiglu=(gu, dgu))
            else:
                hip.gemm(dx, fl["dT"], swiglu_bwd=(gu, dgu))
            if pack is not None:
                a_names = [lp + "mlp.gate_proj.lora_A.weight", lp + "mlp.up_proj.lora_A.weight"]
                tb = lora_grads(dgu, L["t_gu"], h2, a_names, [(lp + "mlp.gate_proj.lora_B.weight", 0, I), (lp + "mlp.up_proj.lora_B.weight", I, I)], L["bits_gu"], group=(i, 2))
                dh2 = dx_gemm(dgu, fl["guT"], tb, a_names, L["bits_gu"])
            else:
                dh2 = hip.gemm(dgu, fl["guT"])
            dx2 = hip.rmsnorm_bwd(dh2, x2, fl["ln2"], L["rstd2"], add=dx)
            # ---- attention: x2 = x + o(attn)
            att = L["att"]
            if pack is not None:
                tb = lora_grads(dx2, L["t_o"], att, [lp + "self_attn.o_proj.lora_A.weight"], [(lp + "self_attn.o_proj.lora_B.weight", 0, D)], L["bits_o"], group=(i, 1))
                datt = dx_gemm(dx2, fl["oT"], tb, [lp + "self_attn.o_proj.lora_A.weight"], L["bits_o"])
            else:
                datt = hip.gemm(dx2, fl["oT"])
            dqkv = torch.empty((M, NQ + 2 * NKV), dtype=BF16, device=dev)
            dk_r = torch.empty((M, NKV), dtype=BF16, device=dev)
            if "rstd_qk" in L and hd == 128 and os.environ.get("UNIREC_ROPE_BWD_FUSED", "1") != "0":
                # the forward ran q/k-norm + RoPE in the q|k|v launch: no raw q, k exist; the rows are recovered from the roped outputs.
                # The q heads' backward rides in the dQ kernel's store (its lanes own whole rows of q_r, which it has just read as
                # its q operand): dq never makes the round trip through HBM.
                # (the k heads' likewise in the dK/dV kernel's store; dk_r is scratch for the shapes the generated kernels do not take)
                k_in_call = os.environ.get("UNIREC_ROPE_K_FUSED", "1") != "0"      # test / lab switch: 0 = the k heads by a separate launch
                hip.attn_bwd(L["actx"], datt.view(B, S, nq, hd), dk=dk_r.view(B, S, nkv, hd), dv=dqkv[:, NQ + NKV:].view(B, S, nkv, hd),
                             rope_q=(L["q_r"], fl["qn"], cos, sin, eps, dqkv[:, :NQ]), rope_rstd=(L["rstd_qk"], 0),
                             rope_k=(L["k_r"], fl["kn"], nq, dqkv[:, NQ:NQ + NKV]) if k_in_call else None)
                if not k_in_call:
                    hip.qknorm_rope_bwd_roped_k(dk_r, L["k_r"], L["rstd_qk"], nq, fl["kn"], cos, sin, dqkv[:, NQ:NQ + NKV], S, nkv, hd)
            elif "rstd_qk" in L:
                dq_r = torch.empty((M, NQ), dtype=BF16, device=dev)
                hip.attn_bwd(L["actx"], datt.view(B, S, nq, hd), dq=dq_r.view(B, S, nq, hd), dk=dk_r.view(B, S, nkv, hd),
                             dv=dqkv[:, NQ + NKV:].view(B, S, nkv, hd))
                hip.qknorm_rope_bwd_roped(dq_r, dk_r, L["q_r"], L["k_r"], L["rstd_qk"], fl["qn"], fl["kn"], cos, sin, dqkv, S, nq, nkv, hd)
            elif hd == 128 and os.environ.get("UNIREC_ROPE_BWD_FUSED", "0") == "1":
                # UNIREC_ROPE_BWD_FUSED=1: the dQ kernel carries the q-norm + RoPE backward of the q heads (its lanes own whole
                # rows) and writes straight into dqkv; the stand-alone kernel is left with the k heads (nq = 0, operands offset
                # to the k columns).  Parity-tested; measured neutral on the joint step (115.7 vs 115.9 seq/s, alternating
                # same-box runs: the ~1000 vector instructions per wave in the dQ kernel's store cost what the 4 saved
                # activation passes return), so the separate launch stays the default.
                hip.attn_bwd(L["actx"], datt.view(B, S, nq, hd), dk=dk_r.view(B, S, nkv, hd), dv=dqkv[:, NQ + NKV:].view(B, S, nkv, hd),
                             rope_q=(qkv[:, :NQ], fl["qn"], cos, sin, eps, dqkv[:, :NQ]))
                hip.qknorm_rope_bwd(dk_r, dk_r, qkv[:, NQ:], fl["qn"], fl["kn"], cos, sin, dqkv[:, NQ:], S, 0, nkv, hd, eps)
            else:
                dq_r = torch.empty((M, NQ), dtype=BF16, device=dev)
                hip.attn_bwd(L["actx"], datt.view(B, S, nq, hd), dq=dq_r.view(B, S, nq, hd), dk=dk_r.view(B, S, nkv, hd),
                             dv=dqkv[:, NQ + NKV:].view(B, S, nkv, hd))
                hip.qknorm_rope_bwd(dq_r, dk_r, qkv, fl["qn"], fl["kn"], cos, sin, dqkv, S, nq, nkv, hd, eps)
            h = L["h"] if "h" in L else hip.rmsnorm_fwd(x, fl["ln1"], eps)[0]              # kept, or recomputed
            if pack is not None:
                a_names = [lp + f"self_attn.{p}_proj.lora_A.weight" for p in "qkv"]
                specs = [(lp + "self_attn.q_proj.lora_B.weight", 0, NQ), (lp + "self_attn.k_proj.lora_B.weight", NQ, NKV),
                         (lp + "self_attn.v_proj.lora_B.weight", NQ + NKV, NKV)]
                tb = lora_grads(dqkv, L["t_qkv"], h, a_names, specs, L["bits_qkv"], group=(i, 0))
                dh = dx_gemm(dqkv, fl["qkvT"], tb, a_names, L["bits_qkv"])
            else:
                dh = hip.gemm(dqkv, fl["qkvT"])
            dx = hip.rmsnorm_bwd(dh, x, fl["ln1"], L["rstd1"], add=dx2)
            L.clear()
            if self.grad_ready_hook is not None:      # dp.GradBuckets: layer i's LoRA gradients are final
                self.grad_ready_hook(i)
        if pack is not None:
            pack.publish_grads(touched)
        if saved.get("bits_pre") is not None and self._bits_pre is None:
            self._prefetch_next_step(saved.pop("bits_pre"), dev)
        if T > 0:
            return hip.inject_bwd(dx.view(B, S, D), saved["ids"], saved["first"], T)
        return None

"""MI355X-native drop-in for the reference's ``models/qformer.py`` (query-only Q-Former path).

Public surface kept: ``BertConfig`` (re-export), ``BertModel(config,
add_pooling_layer=False)`` with ``.forward(query_embeds=, encoder_hidden_states=,
encoder_attention_mask=, attention_mask=, return_dict=True).last_hidden_state`` and the reference's
exact ``state_dict`` keys (dead tensors included, models/qformer.py:56-61,396-397), so reference
checkpoints load unchanged.

Underneath, one autograd node runs the whole encoder forward and one runs the whole backward, each a
sequence of libunirec_hip.so calls (bf16 MFMA GEMMs with fused epilogues, fused MFMA attention, fused
dropout+residual+LayerNorm); torch only owns the memory.  There is no eager fallback.

Reference lines restated here: embeddings models/qformer.py:78-108; masks :784-802 + HF
invert_attention_mask; BertSelfAttention :169-275; BertSelfOutput :285-289; BertLayer (query branch)
:402-484; BertEncoder loop :517-566; BertModel.forward :804-972.
"""
import math
import os
import re
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import hip
from .packing import ParamPack, norm_device

try:  # the reference re-exports transformers' BertConfig (models/qformer.py:46); checkpoints pickle it
    from transformers.models.bert.configuration_bert import BertConfig
except Exception:  # pragma: no cover - transformers is optional for the compute path
    class BertConfig:  # minimal stand-in with the fields the path reads
        def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                     intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                     attention_probs_dropout_prob=0.1, max_position_embeddings=512, initializer_range=0.02,
                     layer_norm_eps=1e-12, pad_token_id=0, **kw):
            self.__dict__.update(dict(vocab_size=vocab_size, hidden_size=hidden_size, num_hidden_layers=num_hidden_layers,
                                      num_attention_heads=num_attention_heads, intermediate_size=intermediate_size,
                                      hidden_act=hidden_act, hidden_dropout_prob=hidden_dropout_prob,
                                      attention_probs_dropout_prob=attention_probs_dropout_prob,
                                      max_position_embeddings=max_position_embeddings, initializer_range=initializer_range,
                                      layer_norm_eps=layer_norm_eps, pad_token_id=pad_token_id))
            self.__dict__.update(kw)

BF16, F32 = torch.bfloat16, torch.float32


# -------------------------------------------------------------------------------------------------
# Parameter containers: plain nn modules used ONLY for their parameters / state_dict keys.
# -------------------------------------------------------------------------------------------------
class _SelfAttentionParams(nn.Module):
    def __init__(self, config, is_cross):
        super().__init__()
        H = config.hidden_size
        kin = config.encoder_width if is_cross else H
        self.query = nn.Linear(H, H)
        self.key = nn.Linear(kin, H)
        self.value = nn.Linear(kin, H)


class _SelfOutputParams(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class _AttentionParams(nn.Module):
    def __init__(self, config, is_cross=False):
        super().__init__()
        self.self = _SelfAttentionParams(config, is_cross)
        self.output = _SelfOutputParams(config)


class _IntermediateParams(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class _OutputParams(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class _LayerParams(nn.Module):
    def __init__(self, config, layer_num):
        super().__init__()
        self.attention = _AttentionParams(config)
        # models/qformer.py:386-393
        self.has_cross_attention = bool(config.add_cross_attention and layer_num % config.cross_attention_freq == 0)
        if self.has_cross_attention:
            self.crossattention = _AttentionParams(config, is_cross=True)
        self.intermediate = _IntermediateParams(config)          # dead on the query-only path
        self.output = _OutputParams(config)                      # dead
        self.intermediate_query = _IntermediateParams(config)
        self.output_query = _OutputParams(config)


class _EmbeddingParams(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)  # dead
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)                  # dead
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))


class _EncoderParams(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([_LayerParams(config, i) for i in range(config.num_hidden_layers)])


_KV_COLSUM = os.environ.get("UNIREC_KV_COLSUM", "1") != "0"     # test / lab: 0 = the K | V bias gradients by a column-sum pass over dK | dV again
_USE_WT = os.environ.get("UNIREC_QF_WT", "1") != "0"     # lab: 0 = dX products read the [out, in] weights as K-strided operands again
# round 6: the weight gradients dW = dY^T X (token reductions, split-K) and the bias column sums of the backward are issued on a SIDE stream:
# they hang off the dX chain (nothing in the backward reads them) and neither they nor the dX products fill 256 CUs at the Q-Formers' row counts
# (item Q-Former of the joint step: 6400 rows = 100 output tiles), so the two streams' kernels run beside each other.  Same kernels, same
# arithmetic: results bit-identical (tests/test_gpu_r6_parity.py).  0 = everything on the caller's stream, as before.
_DW_SIDE = os.environ.get("UNIREC_QF_DW_STREAM", "1") != "0"
_DW_SIDE_MAX_ROWS = 65536       # (beyond: the inputs held until the join are whole gigabytes; C3 runs 32768 rows)
_DW_GROUPED = os.environ.get("UNIREC_QF_DW_GROUPED", "1") != "0"
_side_streams = {}


def _side_stream(device):
    key = (device.type, device.index)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st
_DEAD_RE = re.compile(r"(^|\.)layer\.\d+\.(intermediate\.dense|output\.dense|output\.LayerNorm)\.")


def _dead(name):
    """Reference tensors the query-only path never touches: the text FFN directly under a layer
    (``layer.N.intermediate`` / ``layer.N.output`` -- NOT ``attention.output`` / ``crossattention.output``, which are live)
    and the word / position embedding tables."""
    return bool(_DEAD_RE.search(name)) or "word_embeddings" in name or "position_embeddings" in name


def _split_k_for(out_rows, out_cols, red):
    """dW GEMMs have tiny output grids; split the token reduction so the launch fills 256 CUs."""
    tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128)
    want = max(1, 512 // max(tiles, 1))
    sp = int(max(1, min(want, red // 256, 64)))
    t256 = ((out_rows + 255) // 256) * ((out_cols + 255) // 256)
    one_round = 256 // max(t256, 1)
    if red < 16384 and out_rows >= 256 and out_cols >= 256 and t256 * one_round >= 224 and red // max(one_round, 1) >= 768:
        # ONE round of 224-256 workgroups on the 256x256 tile (ur_gemm takes it from 224 for token reductions) beats 1.7 rounds of
        # the small tile when each split keeps >= 12 K tiles: C2 (tools/lab/dw_split_c2.py) [768, 3072] over 8192 tokens split 3
        # 85.6 us -> split 7 75.3 us, [2304, 768] split 4 66.6 -> split 9 63.8
        return int(one_round)
    if red >= 16384:
        # long token reductions (the user Q-Former: 32 768 query tokens, 819 200 keys): enough splits for ur_gemm to take the
        # 256x256 tile (tiles * splits >= 256) in WHOLE rounds of 256 workgroups.  C3, out [2048, 1024] over 819 200 tokens: split 4
        # on 128x128 tiles 5.5 ms, split 8 4.0 ms; over 32 768 tokens (tools/lab/dw_split.py): [1024, 4096] split 2 454 us ->
        # split 4 335 us, [3072, 1024] split 2 416 -> split 16 298 (split 6 = 1.125 rounds: 392), [1024, 1024] split 8 118 -> 16 104
        whole = 256 // math.gcd(256, t256)
        if red // whole >= 1024:
            sp = max(1, min(64, whole))
        else:
            sp = max(sp, min(64, -(-256 // t256)))
    return sp


# -------------------------------------------------------------------------------------------------
# The encoder as ONE autograd node.
# -------------------------------------------------------------------------------------------------
class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, query_embeds, enc, enc_mask_u8, B, qe_param_name, grad_items=None):
        out, saved = model._forward_impl(query_embeds, enc, enc_mask_u8, B, grad_items=grad_items)
        ctx.model = model
        ctx.saved = saved
        ctx.enc_needs_grad = bool(enc.requires_grad)
        ctx.qe_param_name = qe_param_name
        return out

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        d_qe, d_enc = model._backward_impl(ctx.saved, dout, ctx.enc_needs_grad, ctx.qe_param_name)
        ctx.saved = None
        return None, d_qe, d_enc, None, None, None, None


class BertModel(nn.Module):
    """Query-only Q-Former backbone (``input_ids`` is not supported: the reference's hot path never
    passes it)."""

    def __init__(self, config, add_pooling_layer=False):
        super().__init__()
        if add_pooling_layer:
            raise NotImplementedError("pooler is not on the UniRec hot path (models/qformer.py:592 unused)")
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        if config.hidden_size // config.num_attention_heads != 64:
            raise ValueError("the MI355X Q-Former attention kernel is built for head_dim 64 "
                             "(every UniRec configuration: 256/4, 768/12, 1024/16)")
        self.config = config
        self.embeddings = _EmbeddingParams(config)
        self.encoder = _EncoderParams(config)
        self.pooler = None
        self._init_weights_like_reference()
        self._pack = None
        object.__setattr__(self, "_pack_owner", None)   # wrapper owning a bigger pack (not a submodule: no cycle)
        self._pack_prefix = ""
        self._step = 0
        self.seed = 0x5EED
        # index of this forward's first sample in the GLOBAL minibatch (unirec_amd.dp.set_sample_offset): every dropout mask is
        # keyed on (seed, global element index), so a data-parallel rank draws exactly the masks its samples would get in a
        # single-process run over the whole minibatch -- training does not depend on the number of ranks
        self.dp_rank = 0               # unirec_amd.dp.set_dp_rank: first sample of a forward over B local samples = dp_rank * B ...
        self.sample_offset = None      # ... unless given explicitly (unequal shards)
        self.grad_ready_hook = None

    # models/qformer.py:664-674
    def _init_weights_like_reference(self):
        std = getattr(self.config, "initializer_range", 0.02)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                m.weight.data.normal_(mean=0.0, std=std)
            elif isinstance(m, nn.LayerNorm):
                m.bias.data.zero_()
                m.weight.data.fill_(1.0)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()

    # ---- live-parameter ordering (q|k|v and k|v adjacent so they fuse into one GEMM operand) ----
    def live_named_parameters(self, prefix=""):
        named = dict(self.named_parameters())
        order = ["embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]
        # the cross-attention K | V projections of ALL layers read the same encoder states (models/qformer.py:186-188): their weights
        # (and biases) sit side by side so that one GEMM serves every layer (_cross_kv_names; dp.layer_boundaries skips them)
        order += self._cross_kv_names("")[0] + self._cross_kv_names("")[1]
        for i, lyr in enumerate(self.encoder.layer):
            lp = f"encoder.layer.{i}."
            blocks = ["attention."] + (["crossattention."] if lyr.has_cross_attention else [])
            for b in blocks:
                qkv = ("query", "key", "value") if b == "attention." else ("query",)
                order += [lp + b + f"self.{n}.weight" for n in qkv]
                order += [lp + b + f"self.{n}.bias" for n in qkv]
                order += [lp + b + "output.dense.weight", lp + b + "output.dense.bias",
                          lp + b + "output.LayerNorm.weight", lp + b + "output.LayerNorm.bias"]
            order += [lp + "intermediate_query.dense.weight", lp + "intermediate_query.dense.bias",
                      lp + "output_query.dense.weight", lp + "output_query.dense.bias",
                      lp + "output_query.LayerNorm.weight", lp + "output_query.LayerNorm.bias"]
        return [(prefix + n, named[n]) for n in order]

    def _cross_kv_names(self, pre):
        """([k0.weight, v0.weight, k1.weight, ...], [k0.bias, v0.bias, ...]) over the layers with cross attention, in layer order."""
        cl = [i for i, lyr in enumerate(self.encoder.layer) if lyr.has_cross_attention]
        ws = [f"{pre}encoder.layer.{i}.crossattention.self.{n}.weight" for i in cl for n in ("key", "value")]
        bs = [f"{pre}encoder.layer.{i}.crossattention.self.{n}.bias" for i in cl for n in ("key", "value")]
        return ws, bs

    def dead_parameters(self):
        return [p for n, p in self.named_parameters() if _dead(n)]

    def _ensure_pack(self, device, prefix=""):
        if self._pack_owner is not None:
            return self._pack_owner._ensure_pack(device)
        if self._pack is None or not self._pack.is_current() or self._pack.device != norm_device(device):
            for p in self.dead_parameters():
                p.requires_grad_(False)
            self._pack = ParamPack(self.live_named_parameters(), device)
        return self._pack

    def _set_pack_owner(self, owner, prefix):
        object.__setattr__(self, "_pack_owner", owner)
        self._pack_prefix = prefix

    def _live_names(self):
        if getattr(self, "_live_names_cache", None) is None or self._live_names_cache[0] != self._pack_prefix:
            self._live_names_cache = (self._pack_prefix, [n for n, _ in self.live_named_parameters(self._pack_prefix)])
        return self._live_names_cache[1]

    def _names(self):
        return self._pack_prefix

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, head_mask=None, query_embeds=None,
                encoder_hidden_states=None, encoder_attention_mask=None, past_key_values=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, is_decoder=False):
        if input_ids is not None or past_key_values is not None or is_decoder or output_attentions or head_mask is not None:
            raise NotImplementedError("only the query-only encoder path of models/qformer.py is on the UniRec hot path")
        assert query_embeds is not None, "You have to specify query_embeds when input_ids is None"
        hidden = self.encode(query_embeds, encoder_hidden_states, encoder_attention_mask, attention_mask)
        out = hidden if getattr(self, "return_bf16", False) else _CastFn.apply(hidden)
        if return_dict is False:
            return (out, None)
        return SimpleNamespace(last_hidden_state=out, pooler_output=None, past_key_values=None, hidden_states=None,
                               attentions=None, cross_attentions=None)

    def encode(self, query_embeds, encoder_hidden_states, encoder_attention_mask=None, attention_mask=None,
               qe_param_name=None, grad_items=None):
        """bf16 [B,Q,H] hidden states (internal fast path used by the wrappers).  qe_param_name: the
        pack entry that receives the query-table gradient directly (wrappers); None returns it through
        autograd (free-standing BertModel use).  grad_items = Bg < B: only the FIRST Bg samples carry gradient -- one forward
        serves anchor | positives | negatives of a triplet step (training/item_qformer_training.py:117-131 runs three), and the
        backward walks the leading Bg samples' rows only."""
        if not query_embeds.is_cuda:
            raise hip._lib.UniRecHipError("BertModel runs on the MI355X only: move the model and inputs to 'cuda' "
                                          "(there is no CPU fallback in the product path)")
        B, Q, H = query_embeds.shape
        if attention_mask is not None and not bool((attention_mask != 0).all()):
            raise NotImplementedError("query attention_mask with zeros is not on the UniRec path "
                                      "(models/qformer_utils.py:43 passes all ones)")
        enc = encoder_hidden_states
        assert enc is not None, "encoder_hidden_states must be given for cross-attention layers"
        if enc.dim() != 3 or enc.shape[0] != B or enc.shape[2] != self.config.encoder_width:
            raise ValueError(f"encoder_hidden_states must be [B,T,{self.config.encoder_width}], got {tuple(enc.shape)}")
        mask_u8 = None
        if encoder_attention_mask is not None:
            if tuple(encoder_attention_mask.shape) != (B, enc.shape[1]):
                raise ValueError("Wrong shape for encoder_attention_mask")
            mask_u8 = (encoder_attention_mask != 0).to(torch.uint8).contiguous()
        if B == 0:          # an empty batch is an empty result (as the reference's modules return on empty tensors); nothing to launch
            return torch.empty((0, Q, H), dtype=BF16, device=query_embeds.device)
        # the reference expands one [1,Q,H] parameter over the batch (models/qformer_utils.py:39):
        # keep it un-expanded so the LayerNorm kernel broadcasts it and the gradient reduces over B.
        qe = query_embeds
        if qe.stride(0) == 0 or B == 1:
            qe_src = qe[0:1] if qe.stride(0) == 0 else qe
        else:
            qe_src = qe
        self._ensure_pack(query_embeds.device)
        if not torch.is_grad_enabled():
            # inference (token caches, evaluators, the no-grad positive / negative forward of the item step): no layer's
            # activations are kept, the peak is one layer instead of all of them
            return self._forward_impl(qe_src.detach(), enc.detach(), mask_u8, B, keep=False)[0]
        if grad_items is not None and not (0 < int(grad_items) <= B):
            raise ValueError(f"grad_items must be in 1..{B}")
        return _EncoderFn.apply(self, qe_src, enc, mask_u8, B, qe_param_name, None if grad_items is None or int(grad_items) == B else int(grad_items))

    # ---- implementation (sequence of HIP calls) --------------------------------------------------
    def _drop(self):
        p = float(self.config.hidden_dropout_prob) if self.training else 0.0
        pa = float(self.config.attention_probs_dropout_prob) if self.training else 0.0
        return p, pa

    def _seed(self, layer, site, step=None):
        return (self.seed * 1000003 + (self._step if step is None else step) * 8191 + layer * 64 + site) & 0x7FFFFFFFFFFFFFFF

    def _forward_impl(self, query_embeds, enc, mask_u8, B, keep=True, grad_items=None):
        cfg = self.config
        pack = self._ensure_pack(query_embeds.device)
        pre = self._names()
        pack.refresh_shadow()
        H, eps = cfg.hidden_size, cfg.layer_norm_eps
        Qn = query_embeds.shape[1]
        T = enc.shape[1]
        M, Me = B * Qn, B * T
        p_h, p_a = self._drop()
        if self.training:
            self._step += 1
        qe16 = hip.cast_f32_to_bf16(query_embeds.detach().contiguous().view(-1, H)) if query_embeds.dtype == F32 \
            else query_embeds.detach().contiguous().view(-1, H)
        enc16 = (hip.cast_f32_to_bf16(enc.detach().contiguous()) if enc.dtype == F32 else enc.detach().contiguous()).view(Me, -1)
        b0 = int(self.sample_offset) if self.sample_offset is not None else int(self.dp_rank) * B
        row0 = b0 * Qn                                   # rows of [B*Q, H] activations that precede this shard
        # models/qformer.py:525-548: `config.gradient_checkpointing` in training keeps each layer's INPUT only; the backward re-runs the
        # layer's forward (same kernels, same dropout seeds: bit-identical activations) before it walks the layer
        ckpt = bool(keep and self.training and getattr(cfg, "gradient_checkpointing", False))
        S = {"B": B, "Q": Qn, "T": T, "p_h": p_h, "p_a": p_a, "layers": [], "enc16": enc16, "mask": mask_u8,
             "qe_rows": qe16.shape[0], "step_seed": self._step, "row0": row0, "b0": b0, "ckpt": ckpt, "Bg": B if grad_items is None else int(grad_items)}
        w = lambda n: pack.w32(pre + n)
        x, z0, mean0, rstd0 = hip.layernorm_fwd(qe16, w("embeddings.LayerNorm.weight"), w("embeddings.LayerNorm.bias"), eps,
                                                M=M, p_post=p_h, seed_post=self._seed(1023, 0), drop_row0=row0, save_z=keep)
        S["emb"] = (z0, mean0, rstd0, self._seed(1023, 0))
        # ---- the cross-attention keys | values of EVERY layer: one projection of the encoder states (its input does not depend on the layer)
        kvw, kvb = self._cross_kv_names(pre)
        ncross = len(kvw) // 2
        kv_all = hip.gemm(enc16, pack.fused16(kvw), bias=pack.fused32(kvb)) if ncross else None      # [B*T, ncross * 2H]
        S["kv_all"] = kv_all
        for i in range(len(self.encoder.layer)):
            x_in = x
            x, L = self._layer_forward(i, x, S, pack, pre, save=keep and not ckpt)
            if keep:
                S["layers"].append({"x_in": x_in} if ckpt else L)
        return x.view(B, Qn, H), (S if keep else None)

    def _layer_forward(self, i, x, S, pack, pre, save):
        """One encoder layer on x [B*Q, H] (models/qformer.py:409-511) -> (output, what its backward needs).  Dropout seeds come from the
        step counter recorded in S, so a re-run under `gradient_checkpointing` reproduces the first run bit for bit."""
        cfg = self.config
        lyr = self.encoder.layer[i]
        H, nh, I, eps = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size, cfg.layer_norm_eps
        dh = H // nh
        B, Qn, T, p_h, p_a, row0, b0, mask_u8, kv_all = S["B"], S["Q"], S["T"], S["p_h"], S["p_a"], S["row0"], S["b0"], S["mask"], S["kv_all"]
        M = B * Qn
        seed = lambda site: self._seed(i, site, step=S["step_seed"])
        lp = pre + f"encoder.layer.{i}."
        L = {}
        # ---- self attention
        a = lp + "attention."
        Wqkv = pack.fused16([a + "self.query.weight", a + "self.key.weight", a + "self.value.weight"])
        bqkv = pack.fused32([a + "self.query.bias", a + "self.key.bias", a + "self.value.bias"])
        qkv = hip.gemm(x, Wqkv, bias=bqkv)
        q5 = qkv.view(B, Qn, 3, nh, dh)
        ctx_o, actx = hip.attn_fwd(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], causal=False, dropout_p=p_a, seed=seed(1), drop_batch0=b0)
        y = hip.gemm(ctx_o.view(M, H), pack.w16(a + "output.dense.weight"), bias=pack.w32(a + "output.dense.bias"))
        s_h = seed(2)
        x1, z1, m1, r1 = hip.layernorm_fwd(y, pack.w32(a + "output.LayerNorm.weight"), pack.w32(a + "output.LayerNorm.bias"),
                                           eps, residual=x, p_pre=p_h, seed_pre=s_h, drop_row0=row0, save_z=save)
        L["self"] = (x, qkv, actx, ctx_o, z1, m1, r1, s_h)
        xc = x1
        # ---- cross attention (models/qformer.py:432-447)
        if lyr.has_cross_attention:
            c = lp + "crossattention."
            ncross = kv_all.shape[1] // (2 * H)
            jc = sum(1 for l2 in self.encoder.layer[:i] if l2.has_cross_attention)
            qc = hip.gemm(x1, pack.w16(c + "self.query.weight"), bias=pack.w32(c + "self.query.bias"))
            kv5 = kv_all.view(B, T, ncross, 2, nh, dh)[:, :, jc]
            ctx2, actx2 = hip.attn_fwd(qc.view(B, Qn, nh, dh), kv5[:, :, 0], kv5[:, :, 1], causal=False, key_mask=mask_u8,
                                       dropout_p=p_a, seed=seed(3), drop_batch0=b0)
            y2 = hip.gemm(ctx2.view(M, H), pack.w16(c + "output.dense.weight"), bias=pack.w32(c + "output.dense.bias"))
            s_h2 = seed(4)
            x2, z2, m2, r2 = hip.layernorm_fwd(y2, pack.w32(c + "output.LayerNorm.weight"),
                                               pack.w32(c + "output.LayerNorm.bias"), eps, residual=x1, p_pre=p_h, seed_pre=s_h2, drop_row0=row0, save_z=save)
            L["cross"] = (x1, qc, jc, actx2, ctx2, z2, m2, r2, s_h2)
            xc = x2
        # ---- query FFN (models/qformer.py:449-454, 481-484)
        f1, f2 = lp + "intermediate_query.dense.", lp + "output_query."
        hbuf = torch.empty((M, I), dtype=BF16, device=x.device)
        u = hip.gemm(xc, pack.w16(f1 + "weight"), bias=pack.w32(f1 + "bias"), gelu_out=hbuf)
        y3 = hip.gemm(hbuf, pack.w16(f2 + "dense.weight"), bias=pack.w32(f2 + "dense.bias"))
        s_h3 = seed(5)
        x3, z3, m3, r3 = hip.layernorm_fwd(y3, pack.w32(f2 + "LayerNorm.weight"), pack.w32(f2 + "LayerNorm.bias"), eps,
                                           residual=xc, p_pre=p_h, seed_pre=s_h3, drop_row0=row0, save_z=save)
        L["ffn"] = (xc, u, hbuf, z3, m3, r3, s_h3)
        return x3, L

    def _weight_transposes(self, pack, pre, with_enc):
        """W^T of every weight a dX product reads (dx = dy W), refreshed by ONE launch per backward: with [in, out] copies the
        dX launches are K-contiguous on both operands and take the 8-phase loop like the forward (the token-major [out, in]
        weight as a K-strided operand runs the 2-slot loop: 110 vs 84 us at M 8192, N 3072, K 768).  UNIREC_QF_WT=0: lab."""
        if not _USE_WT:
            return None
        bt = getattr(self, "_wt", None)
        if bt is None or bt[0] is not pack or bt[1] != (pre, with_enc):
            keys, srcs = [], []
            for i, lyr in enumerate(self.encoder.layer):
                lp = pre + f"encoder.layer.{i}."
                a = lp + "attention."
                groups = [((a + "self.query.weight", a + "self.key.weight", a + "self.value.weight"),), ((a + "output.dense.weight",),),
                          ((lp + "intermediate_query.dense.weight",),), ((lp + "output_query.dense.weight",),)]
                if lyr.has_cross_attention:
                    c = lp + "crossattention."
                    groups += [((c + "self.query.weight",),), ((c + "output.dense.weight",),)]
                if i == 0 and with_enc and self._cross_kv_names(pre)[0]:
                    groups += [(tuple(self._cross_kv_names(pre)[0]),)]          # all layers' K | V: one dX for the encoder states
                for (names,) in groups:
                    keys.append(names if len(names) > 1 else names[0])
                    srcs.append(pack.fused16(list(names)) if len(names) > 1 else pack.w16(names[0]))
            bt = (pack, (pre, with_enc), keys, hip.BatchedTranspose(srcs))
            self._wt = bt
        return dict(zip(bt[2], bt[3].run()))

    @staticmethod
    def _prefix(S, Bg):
        """The saved state of a forward over B samples restricted to its first Bg (every saved tensor is row-major over samples: the
        leading rows are a contiguous view); attention contexts get B = Bg (hip.attn_ctx_prefix)."""
        Qn, T = S["Q"], S["T"]
        Mg, Tg = Bg * Qn, Bg * T
        P = dict(S)
        P["B"] = Bg
        P["enc16"] = S["enc16"][:Tg]
        P["mask"] = None if S["mask"] is None else S["mask"][:Bg]
        P["kv_all"] = None if S["kv_all"] is None else S["kv_all"][:Tg]
        z0, mean0, rstd0, s0 = S["emb"]
        P["emb"] = (None if z0 is None else z0[:Mg], mean0[:Mg], rstd0[:Mg], s0)
        layers = []
        for L in S["layers"]:
            if "x_in" in L:                    # gradient checkpointing: the layer is re-run on the leading rows only
                layers.append({"x_in": L["x_in"][:Mg]})
                continue
            x, qkv, actx, ctx_o, z1, m1, r1, s_h = L["self"]
            N = {"self": (x[:Mg], qkv[:Mg], hip.attn_ctx_prefix(actx, Bg), ctx_o[:Bg], z1[:Mg], m1[:Mg], r1[:Mg], s_h)}
            if "cross" in L:
                x1, qc, jc, actx2, ctx2, z2, m2, r2, s_h2 = L["cross"]
                N["cross"] = (x1[:Mg], qc[:Mg], jc, hip.attn_ctx_prefix(actx2, Bg), ctx2[:Bg], z2[:Mg], m2[:Mg], r2[:Mg], s_h2)
            xc, u, hbuf, z3, m3, r3, s_h3 = L["ffn"]
            N["ffn"] = (xc[:Mg], u[:Mg], hbuf[:Mg], z3[:Mg], m3[:Mg], r3[:Mg], s_h3)
            layers.append(N)
        P["layers"] = layers
        return P

    def _backward_impl(self, S, dout, enc_needs_grad, qe_param_name=None):
        if S.get("Bg", S["B"]) < S["B"]:
            if enc_needs_grad or S["qe_rows"] == S["B"] * S["Q"]:
                raise NotImplementedError("grad_items: the encoder states and per-sample query embeddings take no gradient on this path")
            Bg = S["Bg"]
            dout = dout[:Bg]
            S = self._prefix(S, Bg)
        cfg = self.config
        pack = self._ensure_pack(dout.device)
        pre = self._names()
        H, nh, I = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size
        dh = H // nh
        B, Qn, T = S["B"], S["Q"], S["T"]
        M, Me = B * Qn, B * T
        p_h = S["p_h"]
        row0 = S["row0"]
        enc16 = S["enc16"]
        dx = dout.contiguous().view(M, H)
        if dx.dtype != BF16:
            dx = hip.cast_f32_to_bf16(dx)
        d_enc = None
        g = lambda n: pack.g32(pre + n)
        wt = self._weight_transposes(pack, pre, enc_needs_grad)

        def dX(dy, names, **kw):
            """dy W for one weight (or several adjacent ones): through the transposed copy when there is one."""
            key = tuple(names) if len(names) > 1 else names[0]
            if wt is not None:
                return hip.gemm(dy, wt[key], **kw)
            return hip.gemm(dy, pack.fused16(list(names)) if len(names) > 1 else pack.w16(names[0]), s_kcontig=False, **kw)

        # (at 32768 rows -- the user Q-Former of C3 -- both streams' kernels fill the chip: 50.7-51.0 -> 50.4 ms per step, no more)
        side = _side_stream(dout.device) if (_DW_SIDE and dout.is_cuda and M <= _DW_SIDE_MAX_ROWS) else None
        main = torch.cuda.current_stream(dout.device) if side is not None else None
        held = []
        if side is not None:
            side.wait_stream(main)         # (the first use: everything the caller has queued so far, incl. the zeroing of the gradient buffers)

        def off_chain(fn, *tensors):
            """run fn() on the side stream behind what the main stream has queued so far (its inputs `tensors` were produced there)"""
            if side is None:
                return fn()
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            # The caching allocator must not hand the inputs' memory to the main stream before the side stream is done with it: they are kept
            # alive until the next join() (after which the main stream is behind the side stream).  tensor.record_stream() instead makes the
            # allocator poll an event per block at every later allocation: at 32768 rows (C3) the host then stalled for up to 400 ms per step.
            held.extend(tensors)
            with torch.cuda.stream(side):
                return fn()

        def join():
            flush_dW()
            if side is not None:
                main.wait_stream(side)
                del held[:]

        def dW(dy, xin, names):
            """grad of an [out,in] weight (or several adjacent ones): dY^T X, token reduction split over CUs."""
            out = pack.fusedg(names) if len(names) > 1 else pack.g32(names[0])
            if grouped is not None and dy.shape[0] == M and dy.shape[1] % 8 == 0 and xin.shape[1] % 8 == 0 and dy.stride(0) % 8 == 0 and xin.stride(0) % 8 == 0:
                grouped.append((dy, xin, out.view(dy.shape[1], xin.shape[1])))        # leaves with the layer's other weight gradients: flush_dW()
                if len(grouped) == hip.GEMM_MAX_GROUPS:
                    flush_dW()
                return
            off_chain(lambda: hip.gemm(dy, xin, r_kcontig=False, s_kcontig=False, out=out, split_k=_split_k_for(out.shape[0], out.shape[1], dy.shape[0])), dy, xin)

        # The weight gradients of ONE layer as one grouped launch (ur_gemm_grouped): 5-7 products of 9-36 big tiles each over the same tokens.
        # Alone each needs 7-14 token slices to fill the chip (K tiles too short to pay for a tile's prologue, and a split-K reduction launch each);
        # together they are one round of 256 x 256 tiles at split 1-2.  UNIREC_QF_DW_GROUPED=0: one ur_gemm per weight, as before.
        grouped = [] if (_DW_GROUPED and dout.is_cuda and M >= 256) else None

        def flush_dW():
            if not grouped:
                return
            prods = list(grouped)
            del grouped[:]
            t256 = sum(-(-o.shape[0] // 256) * -(-o.shape[1] // 256) for _, _, o in prods)
            sp = max(1, min(256 // max(t256, 1), M // 2048))          # at most one round of big tiles, never fewer than 32 K tiles per slice
            off_chain(lambda: hip.gemm_grouped(prods, split_k=sp), *[t for d_, x_, _ in prods for t in (d_, x_)])

        def colsum(x, out):
            off_chain(lambda: hip.colsum(x, out=out), x)

        def ln_bwd(*a, **kw):
            """hip.layernorm_bwd with the reduction of its per-block partial sums (dgamma, dbeta, dbias: parameter gradients only) on the side stream"""
            if side is None:
                return hip.layernorm_bwd(*a, **kw)
            dz_, dy_, finish = hip.layernorm_bwd(*a, defer_reduce=True, **kw)
            off_chain(finish, finish.scratch)
            return dz_, dy_

        kvw, kvb = self._cross_kv_names(pre)
        ncross = len(kvw) // 2
        dkv_all = torch.empty_like(S["kv_all"]) if ncross else None          # every layer's dK | dV, reduced by ONE launch after the loop
        kv_colsum_fused = None             # True: the attention backward wrote the K | V bias gradients itself (every cross layer)
        for i in reversed(range(len(self.encoder.layer))):
            lyr = self.encoder.layer[i]
            L = S["layers"][i]
            if S["ckpt"]:          # gradient checkpointing: the layer's activations are rebuilt from its saved input
                L = self._layer_forward(i, L["x_in"], S, pack, pre, save=True)[1]
            lp = pre + f"encoder.layer.{i}."
            # ---- FFN
            xc, u, hbuf, z3, m3, r3, s_h3 = L["ffn"]
            f1, f2 = lp + "intermediate_query.dense.", lp + "output_query."
            dz3, dy3 = ln_bwd(dx, z3, m3, r3, pack.w32(f2 + "LayerNorm.weight"), pack.g32(f2 + "LayerNorm.weight"),
                                         pack.g32(f2 + "LayerNorm.bias"), dbias=pack.g32(f2 + "dense.bias"), p_pre=p_h, seed_pre=s_h3, drop_row0=row0)
            dW(dy3, hbuf, [f2 + "dense.weight"])
            du = dX(dy3, [f2 + "dense.weight"], gelu_grad_aux=u)
            colsum(du, pack.g32(f1 + "bias"))
            dW(du, xc, [f1 + "weight"])
            dx = dX(du, [f1 + "weight"], residual=dz3)
            # ---- cross attention
            if lyr.has_cross_attention:
                c = lp + "crossattention."
                x1, qc, jc, actx2, ctx2, z2, m2, r2, s_h2 = L["cross"]
                dz2, dy2 = ln_bwd(dx, z2, m2, r2, pack.w32(c + "output.LayerNorm.weight"),
                                             pack.g32(c + "output.LayerNorm.weight"), pack.g32(c + "output.LayerNorm.bias"),
                                             dbias=pack.g32(c + "output.dense.bias"), p_pre=p_h, seed_pre=s_h2, drop_row0=row0)
                dW(dy2, ctx2.view(M, H), [c + "output.dense.weight"])
                dctx2 = dX(dy2, [c + "output.dense.weight"])
                dkv5 = dkv_all.view(B, T, ncross, 2, nh, dh)[:, :, jc]
                dqc = torch.empty_like(qc)
                # the K | V bias gradients of this layer = column sums of its dK | dV: the few-query dK/dV kernel emits them with the
                # gradients (user Q-Former: 64 queries x 1600 keys) and the second pass over every layer's dK | dV below disappears
                cs = None
                if kv_colsum_fused is not False and _KV_COLSUM and hip.attn_bwd_kv_colsum_supported(actx2):
                    cs = pack.fusedg(kvb)[jc * 2 * H:(jc + 1) * 2 * H]
                    kv_colsum_fused = True
                else:
                    assert kv_colsum_fused is not True, "cross-attention layers of one model take the same kernel"
                    kv_colsum_fused = False
                hip.attn_bwd(actx2, dctx2.view(B, Qn, nh, dh), dq=dqc.view(B, Qn, nh, dh), dk=dkv5[:, :, 0], dv=dkv5[:, :, 1], kv_colsum=cs)
                dW(dqc, x1, [c + "self.query.weight"])
                colsum(dqc, pack.g32(c + "self.query.bias"))
                dx = dX(dqc, [c + "self.query.weight"], residual=dz2)
            # ---- self attention
            a = lp + "attention."
            x0, qkv, actx, ctx_o, z1, m1, r1, s_h = L["self"]
            dz1, dy1 = ln_bwd(dx, z1, m1, r1, pack.w32(a + "output.LayerNorm.weight"),
                                         pack.g32(a + "output.LayerNorm.weight"), pack.g32(a + "output.LayerNorm.bias"),
                                         dbias=pack.g32(a + "output.dense.bias"), p_pre=p_h, seed_pre=s_h, drop_row0=row0)
            dW(dy1, ctx_o.view(M, H), [a + "output.dense.weight"])
            dctx = dX(dy1, [a + "output.dense.weight"])
            dqkv = torch.empty_like(qkv)
            d5 = dqkv.view(B, Qn, 3, nh, dh)
            hip.attn_bwd(actx, dctx.view(B, Qn, nh, dh), dq=d5[:, :, 0], dk=d5[:, :, 1], dv=d5[:, :, 2])
            names = [a + "self.query.weight", a + "self.key.weight", a + "self.value.weight"]
            dW(dqkv, x0, names)
            colsum(dqkv, pack.fusedg([a + "self.query.bias", a + "self.key.bias", a + "self.value.bias"]))
            dx = dX(dqkv, names, residual=dz1)
            L.clear()
            S["layers"][i].clear()
            flush_dW()                                # the layer's weight gradients: one grouped launch (beside the next layer's dX chain)
            if self.grad_ready_hook is not None:      # dp.GradBuckets: layer i's gradients are final
                join()                                # (... once the side stream's token reductions of this layer have run)
                self.grad_ready_hook(i)
        # ---- the cross-attention K | V projections of all layers: one token reduction, one bias sum (and one dX for the encoder states)
        if ncross:
            dW(dkv_all, enc16, kvw)
            if kv_colsum_fused is not True:
                colsum(dkv_all, pack.fusedg(kvb))
            if enc_needs_grad:
                d_enc = dX(dkv_all, kvw)
            if self.grad_ready_hook is not None:
                join()
                self.grad_ready_hook(-2)              # the hoisted K | V gradients of every layer are final (their own bucket: dp.bucket_hook)
        # ---- embeddings LayerNorm; gradient of the batch-broadcast query table reduces over B
        z0, mean0, rstd0, s0 = S["emb"]
        dz0, _ = hip.layernorm_bwd(dx, z0, mean0, rstd0, pack.w32(pre + "embeddings.LayerNorm.weight"),
                                   g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), p_post=p_h, seed_post=s0,
                                   need_dy=False, drop_row0=row0)
        rows = S["qe_rows"]
        touched = list(self._live_names())
        if qe_param_name is not None:      # wrapper-owned [1,Q,H] table: gradient goes straight into the pack
            hip.batch_reduce(dz0, M // rows, rows, H, out=pack.g32(qe_param_name).view(rows, H))
            touched.append(qe_param_name)
            d_qe = None
        elif rows == M:
            d_qe = hip.cast_bf16_to_f32(dz0).view(B, Qn, H)
        else:
            d_qe = hip.batch_reduce(dz0, M // rows, rows, H).view(1, Qn, H)
        if d_enc is not None:
            d_enc = hip.cast_bf16_to_f32(d_enc).view(B, T, -1)
        join()                                        # every gradient of this backward is on the caller's stream from here on
        pack.publish_grads(touched)
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(-1)                  # query table + embedding LayerNorm
        return d_qe, d_enc


class _CastFn(torch.autograd.Function):
    """bf16 -> f32 view of the encoder output for callers that expect the reference's fp32 tensors."""

    @staticmethod
    def forward(ctx, x16):
        return hip.cast_bf16_to_f32(x16)

    @staticmethod
    def backward(ctx, g):
        return hip.cast_f32_to_bf16(g.contiguous())

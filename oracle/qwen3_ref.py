"""fp32 CPU restatement of the Qwen3 decoder + LoRA + joint head (TEST INFRASTRUCTURE ONLY).

The Qwen3 arithmetic is third-party to the reference (``transformers``, unpinned by
README.md:60-63; the copy installed in the build container is 5.15.0,
models/qwen3/modeling_qwen3.py).  Call sites in the reference:
training/train_item_individual_token_joint.py:98-103,143,173-181.
LoRA is ``peft`` (unpinned, not installed): PARITY UNPINNED -- restated from the
published definition  y = W x + (alpha/r) * B(A(dropout(x)))  (call site :121-131).
"""
import math
from dataclasses import dataclass

import torch

from .qformer_ref import linear


@dataclass
class Qwen3Cfg:
    hidden_size: int = 1024
    num_hidden_layers: int = 28
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 3072
    vocab_size: int = 151669
    rope_theta: float = 1e6
    rms_norm_eps: float = 1e-6
    lora_r: int = 16
    lora_alpha: float = 32.0
    lora_dropout: float = 0.0        # reference call site :121-131 trains with 0.1; masks are an explicit input here

    @property
    def lora_scale(self):
        return self.lora_alpha / self.lora_r


LORA_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def rms_norm(x, w, eps):
    # Qwen3RMSNorm.forward, modeling_qwen3.py:59-64
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def rope_cos_sin(S, head_dim, theta):
    # Qwen3RotaryEmbedding, modeling_qwen3.py:107-137; positions arange(S) (:382-385)
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    # modeling_qwen3.py:140-144
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def lora_linear(P, name, x, cfg: Qwen3Cfg, masks=None):
    """Base Linear (bias-free) + LoRA delta if adapter tensors are present.
    peft LoraLayer (training mode): result = base(x) + lora_B(lora_A(dropout(x))) * scaling, one nn.Dropout per
    adapter.  The Bernoulli keep mask is an explicit input (masks[name], 0/1, shape of x): nn.Dropout's RNG
    stream is not part of the algorithm; dropout(x) = x * mask / (1 - p)."""
    y = linear(x, P[name + ".weight"])
    a = P.get(name + ".lora_A.weight")
    if a is not None:
        xd = x
        if masks is not None and name in masks:
            xd = x * masks[name].to(x.dtype).reshape(x.shape) / (1.0 - cfg.lora_dropout)
        y = y + cfg.lora_scale * linear(linear(xd, a), P[name + ".lora_B.weight"])
    return y


def attention_allowed(attention_mask, S):
    """Boolean [B|1,1,S,S]: causal AND key-padding (masking_utils.sdpa_mask :372-536)."""
    ok = torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None]
    if attention_mask is not None:
        ok = ok & attention_mask.bool()[:, None, None, :]
    return ok


def masked_softmax(scores, ok, fully_masked):
    """Two conventions exist for a query row with NO allowed key (a left-pad position), and the
    reference does not pin one (attn implementation / torch version are unpinned):
      "zero"    -- torch SDPA safe-softmax: probabilities 0, attention output 0.  This is what the
                   reference produces with the installed defaults (transformers 5.15 `sdpa`, torch
                   2.10) and is the PRODUCT semantics (a flash kernel that skips masked keys).
      "uniform" -- eager path: additive finfo.min bias (masking_utils.eager_mask :601-603) -> all
                   scores collapse -> uniform over all S keys.
    Rows with at least one allowed key are identical under both."""
    if fully_masked == "uniform":
        return torch.softmax(scores + torch.where(ok, 0.0, torch.finfo(torch.float32).min), dim=-1)
    # (the scores are this call's own temporary: masked in place, and the fully masked rows zeroed by ONE masked_fill -- at S 2048 every
    # [B, heads, S, S] temporary is 0.27 GB per sequence and the page faults of fresh ones were two thirds of the oracle's forward time)
    w = torch.softmax(scores.masked_fill_(~ok, float("-inf")), dim=-1)
    return w.masked_fill(~ok.any(dim=-1, keepdim=True), 0.0)


def decoder_layer(P, pre, x, cos, sin, ok, cfg: Qwen3Cfg, fully_masked="zero", masks=None):
    """Qwen3DecoderLayer.forward modeling_qwen3.py:306-331 with Qwen3Attention :244-280,
    eager_attention_forward :185-208, Qwen3MLP :81-83."""
    B, S, _ = x.shape
    nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    h = rms_norm(x, P[pre + "input_layernorm.weight"], cfg.rms_norm_eps)
    q = lora_linear(P, pre + "self_attn.q_proj", h, cfg, masks).view(B, S, nq, hd)
    k = lora_linear(P, pre + "self_attn.k_proj", h, cfg, masks).view(B, S, nkv, hd)
    v = lora_linear(P, pre + "self_attn.v_proj", h, cfg, masks).view(B, S, nkv, hd)
    q = rms_norm(q, P[pre + "self_attn.q_norm.weight"], cfg.rms_norm_eps).transpose(1, 2)
    k = rms_norm(k, P[pre + "self_attn.k_norm.weight"], cfg.rms_norm_eps).transpose(1, 2)
    v = v.transpose(1, 2)
    c, s = cos[None, None], sin[None, None]
    q = q * c + rotate_half(q) * s
    k = k * c + rotate_half(k) * s
    rep = nq // nkv
    k = k[:, :, None].expand(B, nkv, rep, S, hd).reshape(B, nq, S, hd)      # repeat_kv :173-182
    v = v[:, :, None].expand(B, nkv, rep, S, hd).reshape(B, nq, S, hd)
    w = masked_softmax((q @ k.transpose(2, 3)).mul_(hd ** -0.5), ok, fully_masked)
    a = (w @ v).transpose(1, 2).reshape(B, S, nq * hd)
    x = x + lora_linear(P, pre + "self_attn.o_proj", a, cfg, masks)
    h = rms_norm(x, P[pre + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
    g = lora_linear(P, pre + "mlp.gate_proj", h, cfg, masks)
    u = lora_linear(P, pre + "mlp.up_proj", h, cfg, masks)
    x = x + lora_linear(P, pre + "mlp.down_proj", torch.nn.functional.silu(g) * u, cfg, masks)
    return x


def qwen3_forward(P, cfg: Qwen3Cfg, inputs_embeds, attention_mask=None, prefix="", fully_masked="zero", lora_masks=None, checkpoint_layers=False):
    """Qwen3Model.forward on inputs_embeds, modeling_qwen3.py:367-425 -> last_hidden_state
    (== hidden_states[-1], i.e. AFTER the final norm; SURVEY §3.3 [probe]).
    checkpoint_layers (tests at the full depth x length only): every decoder layer is re-run in the backward instead of keeping its S x S
    probabilities (0.5 GB per layer and sequence at S 2048) -- the same arithmetic in the same order, identical results, a third more
    compute and far less memory churn."""
    B, S, _ = inputs_embeds.shape
    cos, sin = rope_cos_sin(S, cfg.head_dim, cfg.rope_theta)
    ok = attention_allowed(attention_mask, S)
    x = inputs_embeds
    for i in range(cfg.num_hidden_layers):
        if checkpoint_layers:
            from torch.utils.checkpoint import checkpoint
            x = checkpoint(lambda xi, i=i: decoder_layer(P, f"{prefix}layers.{i}.", xi, cos, sin, ok, cfg, fully_masked, lora_masks), x, use_reentrant=False)
        else:
            x = decoder_layer(P, f"{prefix}layers.{i}.", x, cos, sin, ok, cfg, fully_masked, lora_masks)
    return rms_norm(x, P[prefix + "norm.weight"], cfg.rms_norm_eps)


def inject_tokens(text_embeds, input_ids, item_tokens, first_special_id):
    """training/train_item_individual_token_joint.py:160-171: every position of sample b whose id is
    tok(i,j) = first_special_id + i*Qi + j gets item_tokens[b,i,j,:] (out-of-place restatement of
    the in-place index_put; gradient flows to item_tokens, not to the overwritten rows)."""
    B, hist, Qi, D = item_tokens.shape
    rel = input_ids - first_special_id
    is_sp = (rel >= 0) & (rel < hist * Qi)
    src = item_tokens.reshape(B, hist * Qi, D)
    gathered = torch.gather(src, 1, rel.clamp(0, hist * Qi - 1)[..., None].expand(-1, -1, D))
    return torch.where(is_sp[..., None], gathered, text_embeds)


def joint_forward(P, cfg: Qwen3Cfg, input_ids, attention_mask, item_tokens, first_special_id, prefix="",
                  fully_masked="zero", lora_masks=None, checkpoint_layers=False):
    """MultiModalQwenEmbedding.forward after the Q-Former call (:143,160-181):
    embed -> inject -> Qwen3(+LoRA) -> mean over ALL S positions."""
    text = P[prefix + "embed_tokens.weight"][input_ids]
    if item_tokens is not None:
        text = inject_tokens(text, input_ids, item_tokens, first_special_id)
    h = qwen3_forward(P, cfg, text, attention_mask, prefix, fully_masked, lora_masks, checkpoint_layers)
    return h.mean(dim=1)


def l2_normalize(x, eps=1e-12):
    # F.normalize(p=2, dim=-1): x / max(||x||, eps)
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def infonce_loss(user, pos, neg, neg_mask=None, temperature=0.07):
    """InfoNCELoss.forward, training/train_item_individual_token_joint.py:331-352."""
    u, p, n = l2_normalize(user), l2_normalize(pos), l2_normalize(neg)
    pos_sim = (u * p).sum(-1) / temperature
    neg_sim = torch.einsum("bd,bnd->bn", u, n) / temperature
    if neg_mask is not None:
        neg_sim = neg_sim.masked_fill(~neg_mask, float("-inf"))   # == dropping the invalid negatives
    allsim = torch.cat([pos_sim[:, None], neg_sim], dim=1)
    return (-pos_sim + torch.logsumexp(allsim, dim=1)).mean()


def mrr_ranks(user, pos, neg):
    """MRREvaluator._compute_batch_mrr, :392-419.  Returns (scores[B,1+N], rank[B]) with the
    positive at candidate index 0.  Tie rule (the reference's argsort is unstable):
    rank = 1 + #{candidates with score strictly greater than the positive's} -- the value any
    stable-descending sort with the positive first would give."""
    u = l2_normalize(user)
    cands = l2_normalize(torch.cat([pos[:, None, :], neg], dim=1))
    scores = torch.einsum("bd,bnd->bn", u, cands)
    rank = 1 + (scores[:, 1:] > scores[:, :1]).sum(dim=1)
    return scores, rank


def topk_indices(scores, k):
    """Descending top-k with the lowest index first among equal scores (build-defined tie rule)."""
    order = torch.argsort(-scores, dim=-1, stable=True)
    return order[..., :k]


def qwen3_shapes(cfg: Qwen3Cfg, lora=True, prefix=""):
    D, I = cfg.hidden_size, cfg.intermediate_size
    nq, nkv, hd, r = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.lora_r
    s = {prefix + "embed_tokens.weight": (cfg.vocab_size, D), prefix + "norm.weight": (D,)}
    dims = {"self_attn.q_proj": (nq * hd, D), "self_attn.k_proj": (nkv * hd, D), "self_attn.v_proj": (nkv * hd, D),
            "self_attn.o_proj": (D, nq * hd), "mlp.gate_proj": (I, D), "mlp.up_proj": (I, D), "mlp.down_proj": (D, I)}
    for i in range(cfg.num_hidden_layers):
        lp = f"{prefix}layers.{i}."
        s[lp + "input_layernorm.weight"] = (D,)
        s[lp + "post_attention_layernorm.weight"] = (D,)
        s[lp + "self_attn.q_norm.weight"] = (hd,)
        s[lp + "self_attn.k_norm.weight"] = (hd,)
        for n, (o, k) in dims.items():
            s[lp + n + ".weight"] = (o, k)
            if lora:
                s[lp + n + ".lora_A.weight"] = (r, k)
                s[lp + n + ".lora_B.weight"] = (o, r)
    return s

"""fp32 CPU restatement of the reference Q-Former path (TEST INFRASTRUCTURE ONLY).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Parameters are plain dicts keyed by the reference's own
``state_dict`` names, so golden fixtures, the oracle and the HIP product path
all speak the same checkpoint vocabulary.  torch (CPU, fp32) is used only as a
tensor library + autograd; none of the reference's classes are imported.
"""
import math
from dataclasses import dataclass

import torch

F32_MIN = torch.finfo(torch.float32).min  # invert_attention_mask uses finfo(dtype).min


@dataclass
class QFormerCfg:
    """Mirror of the BertConfig fields the path reads (models/qformer_utils.py:22-27)."""
    hidden_size: int = 1024
    num_hidden_layers: int = 12
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    query_length: int = 32
    encoder_width: int = 1024
    cross_attention_freq: int = 2
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.0  # oracle parity runs are eval / p=0

    def has_cross(self, i: int) -> bool:
        # models/qformer.py:386-393
        return i % self.cross_attention_freq == 0


def layer_norm(x, w, b, eps):
    # nn.LayerNorm semantics (biased variance), models/qformer.py:64,281,367
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    # ACT2FN["gelu"] = exact erf GELU (models/qformer.py:352-356)
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def _split_heads(x, nh):
    # transpose_for_scores, models/qformer.py:132-138
    B, T, H = x.shape
    return x.view(B, T, nh, H // nh).permute(0, 2, 1, 3)


def bert_attention(P, pre, hidden, cfg, enc=None, enc_mask_bias=None, self_mask_bias=None):
    """BertAttention = BertSelfAttention + BertSelfOutput.
    models/qformer.py:169-275 (scores / softmax / context), :285-289 (dense+LN)."""
    nh = cfg.num_attention_heads
    dh = cfg.hidden_size // nh
    q = _split_heads(linear(hidden, P[pre + "self.query.weight"], P[pre + "self.query.bias"]), nh)
    kv_src = hidden if enc is None else enc
    k = _split_heads(linear(kv_src, P[pre + "self.key.weight"], P[pre + "self.key.bias"]), nh)
    v = _split_heads(linear(kv_src, P[pre + "self.value.weight"], P[pre + "self.value.bias"]), nh)
    scores = q @ k.transpose(-1, -2) / math.sqrt(dh)             # :198, :238
    bias = self_mask_bias if enc is None else enc_mask_bias        # :188 / :241
    if bias is not None:
        scores = scores + bias
    probs = torch.softmax(scores, dim=-1)                          # :244
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(hidden.shape[0], hidden.shape[1], cfg.hidden_size)
    out = linear(ctx, P[pre + "output.dense.weight"], P[pre + "output.dense.bias"])
    return layer_norm(out + hidden, P[pre + "output.LayerNorm.weight"], P[pre + "output.LayerNorm.bias"],
                      cfg.layer_norm_eps)


def bert_model(P, pre, cfg: QFormerCfg, query_embeds, enc, enc_mask, query_mask=None):
    """BertModel.forward on the query-only path (input_ids=None).
    models/qformer.py:804-972, embeddings :78-108, masks :784-802 + HF invert_attention_mask,
    layer loop :517-566, layer body :402-484."""
    x = layer_norm(query_embeds, P[pre + "embeddings.LayerNorm.weight"], P[pre + "embeddings.LayerNorm.bias"],
                   cfg.layer_norm_eps)
    B, Q, _ = x.shape
    if query_mask is None:
        query_mask = torch.ones(B, Q)
    self_bias = (1.0 - query_mask.to(torch.float32))[:, None, None, :] * -10000.0       # :801
    if enc_mask is None:
        enc_mask = torch.ones(enc.shape[:2])
    enc_bias = (1.0 - enc_mask.to(torch.float32))[:, None, None, :] * F32_MIN           # invert_attention_mask
    for i in range(cfg.num_hidden_layers):
        lp = f"{pre}encoder.layer.{i}."
        x = bert_attention(P, lp + "attention.", x, cfg, self_mask_bias=self_bias)
        if cfg.has_cross(i):
            x = bert_attention(P, lp + "crossattention.", x, cfg, enc=enc, enc_mask_bias=enc_bias)
        inter = gelu_erf(linear(x, P[lp + "intermediate_query.dense.weight"], P[lp + "intermediate_query.dense.bias"]))
        out = linear(inter, P[lp + "output_query.dense.weight"], P[lp + "output_query.dense.bias"])
        x = layer_norm(out + x, P[lp + "output_query.LayerNorm.weight"], P[lp + "output_query.LayerNorm.bias"],
                       cfg.layer_norm_eps)
    return x


def item_qformer_forward(P, cfg: QFormerCfg, field_embeddings, attention_mask=None):
    """QFormerForItemRepresentation.forward, models/qformer_utils.py:37-60
    (dup models/qformer_model.py:27-50)."""
    B = field_embeddings.shape[0]
    qe = P["query_embeddings"].expand(B, -1, -1)
    qo = bert_model(P, "qformer.", cfg, qe, field_embeddings, attention_mask)
    item_rep = linear(qo.mean(dim=1), P["item_representation_head.weight"], P["item_representation_head.bias"])
    rec_q = linear(qo, P["reconstruction_head.weight"], P["reconstruction_head.bias"])          # [B,Q,E]
    rec = linear(rec_q.transpose(1, 2), P["field_projection.weight"], P["field_projection.bias"]).transpose(1, 2)
    return {"query_outputs": qo, "item_representation": item_rep, "reconstructed_fields": rec}


def qformer_loss(out, field_embeddings, attention_mask, pos_rep, neg_rep,
                 recon_w=1.0, cont_w=0.5, margin=0.5):
    """QFormerLoss.forward, training/item_qformer_training.py:49-56.
    masked MSE divides by #valid fields (NOT x E); TripletMarginLoss p=2 eps=1e-6 mean."""
    m = attention_mask.to(torch.float32)
    se = (out["reconstructed_fields"] - field_embeddings) ** 2
    recon = (se * m.unsqueeze(-1)).sum() / m.sum()
    a = out["item_representation"]
    d_ap = torch.sqrt(((a - pos_rep + 1e-6) ** 2).sum(-1))   # F.pairwise_distance adds eps to the difference
    d_an = torch.sqrt(((a - neg_rep + 1e-6) ** 2).sum(-1))
    cont = torch.clamp(d_ap - d_an + margin, min=0.0).mean()
    return recon_w * recon + cont_w * cont, recon, cont


def eval_reconstruction(rec, field_embeddings, attention_mask):
    """evaluation/evaluate_item_qformer.py:75-92 (one batch): masked MSE, summed cosine, #valid."""
    m = attention_mask.to(torch.float32)
    mse = (((rec - field_embeddings) ** 2) * m.unsqueeze(-1)).sum() / m.sum()
    valid = attention_mask.bool()
    o = field_embeddings[valid]
    r = rec[valid]
    on = o / o.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    rn = r / r.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return mse, (on * rn).sum(-1).sum(), int(valid.sum())


def user_qformer_forward(P, cfg: QFormerCfg, user_tokens, attention_mask, n_pred, head_eps=1e-5):
    """UserQFormer.forward, training/user_qformer_training.py:47-68; head :38-43
    (Linear, nn.GELU() exact, nn.LayerNorm default eps 1e-5, Linear)."""
    B = user_tokens.shape[0]
    qe = P["query_embeddings"].expand(B, -1, -1)
    qo = bert_model(P, "qformer.", cfg, qe, user_tokens, attention_mask)
    u = qo.mean(dim=1)
    h = gelu_erf(linear(u, P["prediction_head.0.weight"], P["prediction_head.0.bias"]))
    h = layer_norm(h, P["prediction_head.2.weight"], P["prediction_head.2.bias"], head_eps)
    flat = linear(h, P["prediction_head.3.weight"], P["prediction_head.3.bias"])
    return flat.view(B, n_pred, cfg.encoder_width), qo


def sinusoidal_pe(length, d_model):
    """PositionalEncoding buffer, models/user_sequence_encoder.py:20-25."""
    pos = torch.arange(length, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(length, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def assemble_user_sequence(item_tokens, context_embs):
    """models/user_sequence_encoder.py:128-142 (dropout off): tokens[L,Qi,H] + ctx[L,1,H]
    -> flat [L*Qi,H] + sinusoidal PE over the flat index."""
    L, Qi, H = item_tokens.shape
    flat = (item_tokens + context_embs.unsqueeze(1)).reshape(L * Qi, H)
    return flat + sinusoidal_pe(L * Qi, H)


# ---- parameter-shape helpers (state_dict key inventory) -------------------------

def qformer_live_shapes(cfg: QFormerCfg, prefix="qformer."):
    """Live (gradient-receiving) tensors of BertModel on the query-only path."""
    H, I, E = cfg.hidden_size, cfg.intermediate_size, cfg.encoder_width
    s = {prefix + "embeddings.LayerNorm.weight": (H,), prefix + "embeddings.LayerNorm.bias": (H,)}
    for i in range(cfg.num_hidden_layers):
        lp = f"{prefix}encoder.layer.{i}."
        blocks = [("attention.", H)]
        if cfg.has_cross(i):
            blocks.append(("crossattention.", E))
        for b, kin in blocks:
            s[lp + b + "self.query.weight"] = (H, H)
            s[lp + b + "self.query.bias"] = (H,)
            s[lp + b + "self.key.weight"] = (H, kin)
            s[lp + b + "self.key.bias"] = (H,)
            s[lp + b + "self.value.weight"] = (H, kin)
            s[lp + b + "self.value.bias"] = (H,)
            s[lp + b + "output.dense.weight"] = (H, H)
            s[lp + b + "output.dense.bias"] = (H,)
            s[lp + b + "output.LayerNorm.weight"] = (H,)
            s[lp + b + "output.LayerNorm.bias"] = (H,)
        s[lp + "intermediate_query.dense.weight"] = (I, H)
        s[lp + "intermediate_query.dense.bias"] = (I,)
        s[lp + "output_query.dense.weight"] = (H, I)
        s[lp + "output_query.dense.bias"] = (H,)
        s[lp + "output_query.LayerNorm.weight"] = (H,)
        s[lp + "output_query.LayerNorm.bias"] = (H,)
    return s


def item_qformer_shapes(cfg: QFormerCfg, num_fields: int):
    s = {"query_embeddings": (1, cfg.query_length, cfg.hidden_size)}
    s.update(qformer_live_shapes(cfg))
    H, E, Q = cfg.hidden_size, cfg.encoder_width, cfg.query_length
    s["item_representation_head.weight"] = (E, H)
    s["item_representation_head.bias"] = (E,)
    s["reconstruction_head.weight"] = (E, H)
    s["reconstruction_head.bias"] = (E,)
    s["field_projection.weight"] = (num_fields, Q)
    s["field_projection.bias"] = (num_fields,)
    return s


def user_qformer_shapes(cfg: QFormerCfg, n_pred: int):
    s = {"query_embeddings": (1, cfg.query_length, cfg.hidden_size)}
    s.update(qformer_live_shapes(cfg))
    H, E = cfg.hidden_size, cfg.encoder_width
    s["prediction_head.0.weight"] = (H, H)
    s["prediction_head.0.bias"] = (H,)
    s["prediction_head.2.weight"] = (H,)
    s["prediction_head.2.bias"] = (H,)
    s["prediction_head.3.weight"] = (n_pred * E, H)
    s["prediction_head.3.bias"] = (n_pred * E,)
    return s

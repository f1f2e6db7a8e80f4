"""Training-mode (dropout ON) restatement of the reference Q-Former path with the keep masks FED IN (TEST INFRASTRUCTURE ONLY).

oracle/qformer_ref.py restates the path at p = 0 (eval, or dropout=0.0 fixtures).  The timed configurations train with dropout
(/root/reference/models/qformer_utils.py:17-27: dropout 0.2 for both probabilities; training/user_qformer_training.py:23-28: 0.1),
and nn.Dropout draws from torch's generator, which no other implementation can reproduce.  Here every nn.Dropout of the path is
``x * keep / (1 - p)`` with `keep` an explicit 0/1 tensor -- the semantics of nn.Dropout in training mode -- at exactly the
reference's sites:

    embeddings            models/qformer.py:106-107   LayerNorm, then dropout
    attention probs       models/qformer.py:244-264   softmax, dropout, then probs @ V   (self and cross attention)
    BertSelfOutput        models/qformer.py:285-289   dense, dropout, LayerNorm(. + input)  (self and cross attention)
    BertOutput (query)    models/qformer.py:371-375   dense, dropout, LayerNorm(. + input)

Pinned by tests/golden/*_train.npz: the reference's own classes run in train() mode with their nn.Dropout modules applying the
same masks (tests/golden/make_golden_r5.py); with masks = None every function here equals its oracle/qformer_ref.py counterpart.
Mask keys: "emb", "{i}.self.probs", "{i}.self.out", "{i}.cross.probs", "{i}.cross.out", "{i}.ffn.out" (oracle/dropout_ref.py).
"""
import math

import torch

from .qformer_ref import F32_MIN, QFormerCfg, _split_heads, gelu_erf, layer_norm, linear


def _drop(x, masks, key, p):
    if masks is None or p <= 0.0:
        return x
    keep = torch.as_tensor(masks[key]).to(x.dtype)
    assert keep.shape == x.shape, (key, tuple(keep.shape), tuple(x.shape))
    return x * keep / (1.0 - p)


def bert_attention_train(P, pre, hidden, cfg, masks, kprobs, kout, p, enc=None, enc_mask_bias=None, self_mask_bias=None):
    """BertAttention in training mode: models/qformer.py:169-275 (:258 probability dropout), :285-289 (:287 hidden dropout)."""
    nh = cfg.num_attention_heads
    dh = cfg.hidden_size // nh
    q = _split_heads(linear(hidden, P[pre + "self.query.weight"], P[pre + "self.query.bias"]), nh)
    kv_src = hidden if enc is None else enc
    k = _split_heads(linear(kv_src, P[pre + "self.key.weight"], P[pre + "self.key.bias"]), nh)
    v = _split_heads(linear(kv_src, P[pre + "self.value.weight"], P[pre + "self.value.bias"]), nh)
    scores = q @ k.transpose(-1, -2) / math.sqrt(dh)
    bias = self_mask_bias if enc is None else enc_mask_bias
    if bias is not None:
        scores = scores + bias
    probs = _drop(torch.softmax(scores, dim=-1), masks, kprobs, p)                 # :244, :258
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(hidden.shape[0], hidden.shape[1], cfg.hidden_size)
    out = _drop(linear(ctx, P[pre + "output.dense.weight"], P[pre + "output.dense.bias"]), masks, kout, p)      # :286-287
    return layer_norm(out + hidden, P[pre + "output.LayerNorm.weight"], P[pre + "output.LayerNorm.bias"], cfg.layer_norm_eps)


def bert_model_train(P, pre, cfg: QFormerCfg, query_embeds, enc, enc_mask, masks, p):
    """BertModel.forward on the query-only path in training mode (models/qformer.py:804-972; layer body :402-484)."""
    x = layer_norm(query_embeds, P[pre + "embeddings.LayerNorm.weight"], P[pre + "embeddings.LayerNorm.bias"], cfg.layer_norm_eps)
    x = _drop(x, masks, "emb", p)                                                    # :106-107
    B, Q, _ = x.shape
    self_bias = torch.zeros(B, 1, 1, Q)                                              # query mask of ones (:801)
    if enc_mask is None:
        enc_mask = torch.ones(enc.shape[:2])
    enc_bias = (1.0 - enc_mask.to(torch.float32))[:, None, None, :] * F32_MIN
    for i in range(cfg.num_hidden_layers):
        lp = f"{pre}encoder.layer.{i}."
        x = bert_attention_train(P, lp + "attention.", x, cfg, masks, f"{i}.self.probs", f"{i}.self.out", p, self_mask_bias=self_bias)
        if cfg.has_cross(i):
            x = bert_attention_train(P, lp + "crossattention.", x, cfg, masks, f"{i}.cross.probs", f"{i}.cross.out", p,
                                     enc=enc, enc_mask_bias=enc_bias)
        inter = gelu_erf(linear(x, P[lp + "intermediate_query.dense.weight"], P[lp + "intermediate_query.dense.bias"]))
        out = _drop(linear(inter, P[lp + "output_query.dense.weight"], P[lp + "output_query.dense.bias"]), masks, f"{i}.ffn.out", p)
        x = layer_norm(out + x, P[lp + "output_query.LayerNorm.weight"], P[lp + "output_query.LayerNorm.bias"], cfg.layer_norm_eps)
    return x


def item_qformer_forward_train(P, cfg: QFormerCfg, field_embeddings, attention_mask, masks, p):
    """QFormerForItemRepresentation.forward in training mode (models/qformer_utils.py:37-60; the heads carry no dropout)."""
    B = field_embeddings.shape[0]
    qe = P["query_embeddings"].expand(B, -1, -1)
    qo = bert_model_train(P, "qformer.", cfg, qe, field_embeddings, attention_mask, masks, p)
    item_rep = linear(qo.mean(dim=1), P["item_representation_head.weight"], P["item_representation_head.bias"])
    rec_q = linear(qo, P["reconstruction_head.weight"], P["reconstruction_head.bias"])
    rec = linear(rec_q.transpose(1, 2), P["field_projection.weight"], P["field_projection.bias"]).transpose(1, 2)
    return {"query_outputs": qo, "item_representation": item_rep, "reconstructed_fields": rec}


def user_qformer_forward_train(P, cfg: QFormerCfg, user_tokens, attention_mask, n_pred, masks, p, head_eps=1e-5):
    """UserQFormer.forward in training mode (training/user_qformer_training.py:47-68; the prediction head has no dropout :38-43)."""
    B = user_tokens.shape[0]
    qe = P["query_embeddings"].expand(B, -1, -1)
    qo = bert_model_train(P, "qformer.", cfg, qe, user_tokens, attention_mask, masks, p)
    u = qo.mean(dim=1)
    h = gelu_erf(linear(u, P["prediction_head.0.weight"], P["prediction_head.0.bias"]))
    h = layer_norm(h, P["prediction_head.2.weight"], P["prediction_head.2.bias"], head_eps)
    flat = linear(h, P["prediction_head.3.weight"], P["prediction_head.3.bias"])
    return flat.view(B, n_pred, cfg.encoder_width), qo

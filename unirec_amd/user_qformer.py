"""Drop-in for ``UserQFormer`` (training/user_qformer_training.py:17-68).

4-layer Q-Former, 64 learned queries, cross-attention in EVERY layer over the user's history of
item query tokens [B,T,E] (T = hist*Q_item, ragged -> attention_mask), mean over queries, MLP head
(Linear, exact GELU, LayerNorm(eps 1e-5), Linear) predicting the next item's query tokens.
"""
import torch
import torch.nn as nn

from . import hip
from .packing import ParamPack, norm_device
from .qformer import BertConfig, BertModel, _split_k_for

BF16, F32 = torch.bfloat16, torch.float32


class _UserHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, h16):
        pack = model._ensure_pack(h16.device)
        B, Q, H = h16.shape
        _, u16 = hip.mean_pool_fwd(h16, out_f32=False, out_bf16=True)                       # :60
        g = torch.empty((B, H), dtype=BF16, device=h16.device)
        a = hip.gemm(u16, pack.w16("prediction_head.0.weight"), bias=pack.w32("prediction_head.0.bias"), gelu_out=g)
        ln, _, mean, rstd = hip.layernorm_fwd(g, pack.w32("prediction_head.2.weight"), pack.w32("prediction_head.2.bias"),
                                              model.prediction_head[2].eps, save_z=False)
        flat = hip.gemm(ln, pack.w16("prediction_head.3.weight"), bias=pack.w32("prediction_head.3.bias"), out_f32=True)
        ctx.model, ctx.saved = model, (u16, a, g, ln, mean, rstd, Q)
        return flat

    @staticmethod
    def backward(ctx, d_flat):
        model = ctx.model
        u16, a, g, ln, mean, rstd, Q = ctx.saved
        pack = model._ensure_pack(u16.device)
        B, H = u16.shape
        d16 = hip.cast_f32_to_bf16(d_flat.contiguous())
        hip.gemm(d16, ln, r_kcontig=False, s_kcontig=False, out=pack.g32("prediction_head.3.weight"))
        hip.colsum(d16, out=pack.g32("prediction_head.3.bias"))
        # [B, Q_pred * E] x [Q_pred * E, H]: 32 output tiles only, 32 768 deep -- split the reduction over the CUs (575 -> ~90 us at C3)
        W3 = pack.w16("prediction_head.3.weight")
        sp = max(1, min(16, W3.shape[0] // 2048))
        dln = hip.gemm(d16, W3, s_kcontig=False) if sp == 1 else hip.cast_f32_to_bf16(hip.gemm(d16, W3, s_kcontig=False, out_f32=True, split_k=sp))
        dz, _ = hip.layernorm_bwd(dln, g, mean, rstd, pack.w32("prediction_head.2.weight"), pack.g32("prediction_head.2.weight"),
                                  pack.g32("prediction_head.2.bias"), need_dy=False)
        da = hip.gelu_bwd(dz, a)
        hip.gemm(da, u16, r_kcontig=False, s_kcontig=False, out=pack.g32("prediction_head.0.weight"))
        hip.colsum(da, out=pack.g32("prediction_head.0.bias"))
        du = hip.gemm(da, pack.w16("prediction_head.0.weight"), s_kcontig=False)
        dh = hip.mean_pool_bwd(du, Q)
        pack.publish_grads([f"prediction_head.{i}.{k}" for i in (0, 2, 3) for k in ("weight", "bias")])
        return None, dh


class UserQFormer(nn.Module):
    """A Q-Former model to create a fixed-length representation of a variable-length user sequence."""

    def __init__(self, hidden_size: int = 1024, num_hidden_layers: int = 4, num_attention_heads: int = 16,
                 intermediate_size: int = 4096, num_query_tokens: int = 64, input_embedding_dim: int = 1024,
                 num_item_tokens_to_predict: int = 32, dropout: float = 0.1):
        super().__init__()
        self.config = BertConfig(
            hidden_size=hidden_size, num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
            intermediate_size=intermediate_size, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout,
            add_cross_attention=True, query_length=num_query_tokens, encoder_width=input_embedding_dim,
            cross_attention_freq=1)
        self.num_query_tokens = num_query_tokens
        self.query_embeddings = nn.Parameter(torch.randn(1, num_query_tokens, hidden_size))
        self.qformer = BertModel(self.config, add_pooling_layer=False)
        self.prediction_head = nn.Sequential(
            nn.Linear(hidden_size, hidden_size), nn.GELU(), nn.LayerNorm(hidden_size),
            nn.Linear(hidden_size, num_item_tokens_to_predict * input_embedding_dim))
        self.num_item_tokens_to_predict = num_item_tokens_to_predict
        self.input_embedding_dim = input_embedding_dim
        self.qformer._set_pack_owner(self, "qformer.")
        self._pack = None

    def live_named_parameters(self):
        named = dict(self.named_parameters())
        head = [f"prediction_head.{i}.{k}" for i in (0, 2, 3) for k in ("weight", "bias")]
        return ([("query_embeddings", self.query_embeddings)] + self.qformer.live_named_parameters("qformer.")
                + [(n, named[n]) for n in head])

    def _ensure_pack(self, device):
        if self._pack is None or not self._pack.is_current() or self._pack.device != norm_device(device):
            for p in self.qformer.dead_parameters():
                p.requires_grad_(False)
            self._pack = ParamPack(self.live_named_parameters(), device)
        return self._pack

    @property
    def pack(self):
        return self._pack

    def encode_bf16(self, user_sequence_tokens, attention_mask):
        """[B,64,H] bf16 user query tokens (the tokens C5 injects into Qwen3)."""
        B = user_sequence_tokens.shape[0]
        query_embeds = self.query_embeddings.expand(B, -1, -1)
        return self.qformer.encode(query_embeds, user_sequence_tokens, attention_mask, None, qe_param_name="query_embeddings")

    def forward(self, user_sequence_tokens: torch.Tensor, attention_mask: torch.Tensor):
        batch_size = user_sequence_tokens.shape[0]
        h16 = self.encode_bf16(user_sequence_tokens, attention_mask)
        if batch_size == 0:
            return torch.zeros((0, self.num_item_tokens_to_predict, self.input_embedding_dim), dtype=torch.float32, device=h16.device)
        flat = _UserHeadFn.apply(self, h16)
        return flat.view(batch_size, self.num_item_tokens_to_predict, self.input_embedding_dim)

"""Drop-in for ``models/qformer_utils.py:16-60`` (class part): ``QFormerForItemRepresentation``.

Same constructor, same ``forward(field_embeddings, attention_mask=None)`` -> dict with
``query_outputs [B,Q,H]``, ``item_representation [B,E]``, ``reconstructed_fields [B,F,E]`` (fp32
tensors carrying autograd), same attributes (``config``, ``num_query_tokens``, ``qformer``,
``query_embeddings``) and the same ``state_dict`` keys.  The dataset part of the reference file
(QFormerDataset, load_real_data) is upstream of the hot path and out of scope.
"""
import torch
import torch.nn as nn

from . import hip
from .packing import ParamPack, norm_device
from .qformer import BertConfig, BertModel, _CastFn, _split_k_for

BF16, F32 = torch.bfloat16, torch.float32


class _ItemHeadsFn(torch.autograd.Function):
    """item_representation_head(mean_Q) + field_projection(reconstruction_head(.)^T)^T as one node
    (models/qformer_utils.py:50-54)."""

    @staticmethod
    def forward(ctx, model, h16):
        pack = model._ensure_pack(h16.device)
        B, Q, H = h16.shape
        E = model.config.encoder_width
        _, pooled = hip.mean_pool_fwd(h16, out_f32=False, out_bf16=True)
        item = hip.gemm(pooled, pack.w16("item_representation_head.weight"), bias=pack.w32("item_representation_head.bias"),
                        out_f32=True)
        rec_q = hip.gemm(h16.view(B * Q, H), pack.w16("reconstruction_head.weight"), bias=pack.w32("reconstruction_head.bias"))
        rec = hip.field_projection_fwd(rec_q.view(B, Q, E), pack.w32("field_projection.weight"), pack.w32("field_projection.bias"))
        ctx.model, ctx.saved = model, (h16, pooled, rec_q)
        ctx.set_materialize_grads(False)
        return item, rec

    @staticmethod
    def backward(ctx, d_item, d_rec):
        model = ctx.model
        h16, pooled, rec_q = ctx.saved
        pack = model._ensure_pack(h16.device)
        B, Q, H = h16.shape
        E = model.config.encoder_width
        dh = None
        touched = []
        if d_rec is not None:
            drq = hip.field_projection_bwd(d_rec.contiguous(), rec_q.view(B, Q, E), pack.w32("field_projection.weight"),
                                           pack.g32("field_projection.weight"), pack.g32("field_projection.bias")).view(B * Q, E)
            gW = pack.g32("reconstruction_head.weight")
            hip.gemm(drq, h16.view(B * Q, H), r_kcontig=False, s_kcontig=False, out=gW, split_k=_split_k_for(E, H, B * Q))
            hip.colsum(drq, out=pack.g32("reconstruction_head.bias"))
            dh = hip.gemm(drq, pack.w16("reconstruction_head.weight"), s_kcontig=False)
            touched += ["field_projection.weight", "field_projection.bias", "reconstruction_head.weight", "reconstruction_head.bias"]
        if d_item is not None:
            d16 = hip.cast_f32_to_bf16(d_item.contiguous())
            hip.gemm(d16, pooled, r_kcontig=False, s_kcontig=False, out=pack.g32("item_representation_head.weight"))
            hip.colsum(d16, out=pack.g32("item_representation_head.bias"))
            dpool = hip.gemm(d16, pack.w16("item_representation_head.weight"), s_kcontig=False)
            dh2 = hip.mean_pool_bwd(dpool, Q).view(B * Q, H)
            dh = dh2 if dh is None else hip.add_bf16(dh, dh2)
            touched += ["item_representation_head.weight", "item_representation_head.bias"]
        pack.publish_grads(touched)
        return None, (None if dh is None else dh.view(B, Q, H))


class QFormerForItemRepresentation(nn.Module):
    def __init__(self, hidden_size: int = 1024, num_hidden_layers: int = 12, num_attention_heads: int = 16,
                 intermediate_size: int = 4096, num_query_tokens: int = 32, field_embedding_dim: int = 1024,
                 num_fields: int = None, dropout: float = 0.2):
        super().__init__()
        if num_fields is None:
            raise ValueError("num_fields must be provided")
        self.config = BertConfig(
            hidden_size=hidden_size, num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
            intermediate_size=intermediate_size, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout,
            add_cross_attention=True, query_length=num_query_tokens, encoder_width=field_embedding_dim,
            cross_attention_freq=2)
        self.num_query_tokens = num_query_tokens
        self.query_embeddings = nn.Parameter(torch.randn(1, num_query_tokens, hidden_size))
        self.qformer = BertModel(self.config, add_pooling_layer=False)
        self.item_representation_head = nn.Linear(hidden_size, field_embedding_dim)
        self.reconstruction_head = nn.Linear(hidden_size, field_embedding_dim)
        self.field_projection = nn.Linear(num_query_tokens, num_fields)
        self.qformer._set_pack_owner(self, "qformer.")
        self._pack = None

    # ---- flat parameter pack over every live tensor of the wrapper ------------------------------
    def live_named_parameters(self):
        named = dict(self.named_parameters())
        heads = ["item_representation_head.weight", "item_representation_head.bias", "reconstruction_head.weight",
                 "reconstruction_head.bias", "field_projection.weight", "field_projection.bias"]
        return ([("query_embeddings", self.query_embeddings)] + self.qformer.live_named_parameters("qformer.")
                + [(n, named[n]) for n in heads])

    def _ensure_pack(self, device):
        if self._pack is None or not self._pack.is_current() or self._pack.device != norm_device(device):
            for p in self.qformer.dead_parameters():
                p.requires_grad_(False)
            self._pack = ParamPack(self.live_named_parameters(), device)
        return self._pack

    @property
    def pack(self):
        return self._pack

    def encode_bf16(self, field_embeddings, attention_mask=None):
        """[B,Q,H] bf16 query tokens (what the joint model injects); carries autograd."""
        B = field_embeddings.shape[0]
        query_embeds = self.query_embeddings.expand(B, -1, -1)
        return self.qformer.encode(query_embeds, field_embeddings, attention_mask, None, qe_param_name="query_embeddings")

    def forward_triplet(self, anchor_fields, anchor_mask, other_fields, other_mask):
        """The triplet step's forwards as ONE launch sequence (the reference runs anchor, positives and negatives through the model
        one after the other, training/item_qformer_training.py:117-131, the last two without gradient): the encoder runs once over
        anchor | others, only the anchor rows are walked by its backward.  Returns (the anchor's output dict, with autograd; the
        others' item_representation [n_others, E], detached).  Dropout masks are keyed on the row index inside this merged batch."""
        B = anchor_fields.shape[0]
        x = torch.cat([anchor_fields, other_fields])
        # each side's padding mask on its own terms, as the separate forwards apply them: a missing one is all ones
        if anchor_mask is None and other_mask is None:
            mask = None
        else:
            ones = lambda f, like: torch.ones(f.shape[:2], dtype=like.dtype, device=f.device)
            am = anchor_mask if anchor_mask is not None else ones(anchor_fields, other_mask)
            om = other_mask if other_mask is not None else ones(other_fields, anchor_mask)
            mask = torch.cat([am, om.to(am.dtype)])
        q = self.query_embeddings.expand(x.shape[0], -1, -1)
        h_all = self.qformer.encode(q, x, mask, None, qe_param_name="query_embeddings", grad_items=B)
        h_a = h_all[:B]
        item, rec = _ItemHeadsFn.apply(self, h_a)
        with torch.no_grad():
            other_item, _ = _ItemHeadsFn.apply(self, h_all[B:].detach())
        return {"query_outputs": _CastFn.apply(h_a), "item_representation": item, "reconstructed_fields": rec}, other_item

    def forward(self, field_embeddings: torch.Tensor, attention_mask: torch.Tensor = None):
        h16 = self.encode_bf16(field_embeddings, attention_mask)
        if h16.shape[0] == 0:
            z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=h16.device)
            E = self.qformer.config.encoder_width
            return {"query_outputs": z(0, h16.shape[1], h16.shape[2]), "item_representation": z(0, E),
                    "reconstructed_fields": z(0, self.field_projection.out_features, E)}
        item, rec = _ItemHeadsFn.apply(self, h16)
        return {"query_outputs": _CastFn.apply(h16), "item_representation": item, "reconstructed_fields": rec}

"""Device-side tensor path of ``UserSequenceEncoder`` (models/user_sequence_encoder.py:36-142).

The reference class loads frozen modality encoders (text / CLIP / MWNE) from the network to turn raw
item dicts into field embeddings; that part is upstream of the hot path and out of scope (SURVEY.md §2
rows 9-11).  What is on the path (row U0) is the tensor assembly of the user sequence from *cached*
item query tokens, which is what BASELINE config 3 measures: tokens + (timestamp + geo) context +
sinusoidal positional encoding over the flat token index + right-padding with a mask.  This module
keeps the reference's method names for that part and runs it as one HIP kernel.
"""
import torch

from . import hip


class PositionalEncoding:
    """Parameter-free stand-in: the sinusoidal table is generated inside ur_user_sequence_assemble."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 5000):
        self.d_model, self.p, self.max_len = d_model, dropout, max_len


class UserSequenceAssembler:
    def __init__(self, embedding_dim: int = 1024, num_query_tokens: int = 32, dropout: float = 0.1, training: bool = False):
        self.embedding_dim = embedding_dim
        self.num_query_tokens = num_query_tokens
        self.positional_encoder = PositionalEncoding(embedding_dim, dropout)
        self.training = training       # the reference leaves nn.Dropout in train mode (user_sequence_encoder.py:50)
        self._step = 0

    def encode_user_sequences(self, item_query_tokens, context_embs, lengths):
        """Batched encode_user_sequence + collate padding.
        item_query_tokens [B,L,Qi,H] (cached item tokens), context_embs [B,L,H] (time + geo embeddings),
        lengths [B] events per user -> (padded_inputs [B,L*Qi,H] bf16, attention_mask [B,L*Qi] float)."""
        if not item_query_tokens.is_cuda:
            raise hip._lib.UniRecHipError("UserSequenceAssembler runs on the MI355X only")
        t = item_query_tokens if item_query_tokens.dtype == torch.bfloat16 else hip.cast_f32_to_bf16(item_query_tokens.contiguous())
        c = context_embs if context_embs.dtype == torch.bfloat16 else hip.cast_f32_to_bf16(context_embs.contiguous())
        self._step += 1
        p = self.positional_encoder.p if self.training else 0.0
        return hip.user_sequence_assemble(t.contiguous(), c.contiguous(), lengths.to(torch.int32).contiguous(), p, 0xA55E + self._step)

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
    """Parameter-free stand-in: the sinusoidal table is generated inside ur_user_sequence_assemble.  `training` mirrors
    the nn.Module flag: the reference never calls .eval() on it, so its dropout(0.1) is ACTIVE in encode_user_sequence
    (models/user_sequence_encoder.py:50,140)."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 5000):
        self.d_model, self.p, self.max_len = d_model, dropout, max_len
        self.training = True

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def to(self, device):
        return self


class UserSequenceAssembler:
    def __init__(self, embedding_dim: int = 1024, num_query_tokens: int = 32, dropout: float = 0.1, training: bool = False):
        self.embedding_dim = embedding_dim
        self.num_query_tokens = num_query_tokens
        self.positional_encoder = PositionalEncoding(embedding_dim, dropout)
        self.training = training       # the reference leaves nn.Dropout in train mode (user_sequence_encoder.py:50)
        self._step = 0
        self.dp_rank, self.sample_offset = 0, None      # dropout counters keyed on the global user index (unirec_amd.dp.set_dp_rank)

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
        b0 = int(self.sample_offset) if self.sample_offset is not None else int(self.dp_rank) * int(t.shape[0])
        return hip.user_sequence_assemble(t.contiguous(), c.contiguous(), lengths.to(torch.int32).contiguous(), p, 0xA55E + self._step,
                                          drop_batch0=b0)


# ---- event context encoders + batched sequence encoding from the token cache --------
class _ContextMLP(torch.nn.Module):
    """Linear(nf, 2H) -> GELU -> Linear(2H, H) with the reference's parameter names (``projection.0`` / ``projection.2``).
    Forward only: the reference never trains these (models/user_sequence_encoder.py:118-121 runs them under no_grad)."""
    kind, nfeat = 0, 9

    def __init__(self, embedding_dim: int):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.projection = torch.nn.Sequential(torch.nn.Linear(self.nfeat, embedding_dim * 2), torch.nn.GELU(),
                                              torch.nn.Linear(embedding_dim * 2, embedding_dim))
        self._w2 = None

    def _second(self, dev):
        w = self.projection[2].weight
        key = (w.data_ptr(), w._version, str(dev))
        if self._w2 is None or self._w2[0] != key:
            self._w2 = (key, hip.cast_f32_to_bf16(w.detach().to(dev, torch.float32).contiguous()),
                        self.projection[2].bias.detach().to(dev, torch.float32).contiguous())
        return self._w2[1], self._w2[2]

    def _run(self, x32):
        if not x32.is_cuda:
            raise hip._lib.UniRecHipError(f"{type(self).__name__} runs on the MI355X only")
        l0 = self.projection[0]
        h1 = hip.context_mlp1(x32.contiguous(), self.kind, l0.weight.detach().to(x32.device, torch.float32).contiguous(),
                              l0.bias.detach().to(x32.device, torch.float32).contiguous())
        w2, b2 = self._second(x32.device)
        return hip.gemm(h1, w2, bias=b2)            # [n, H] bf16


class TimestampEncoder(_ContextMLP):
    """models/mwne.py:504-565: secular + time-of-day / day-of-week / day-of-year / month sin-cos features (f32, as the
    reference computes them from ``timestamps.float()``), then the two-layer projection."""
    kind, nfeat = 0, 9

    def forward(self, timestamps):
        return self._run(timestamps.float().view(-1))


class GeoCoordinateEncoder(_ContextMLP):
    """models/mwne.py:568-607: (lat, lon) degrees -> unit-sphere (x, y, z) -> projection."""
    kind, nfeat = 1, 3

    def forward(self, coordinates):
        if coordinates.dim() != 2 or coordinates.shape[1] != 2:
            raise ValueError("Input coordinates must be of shape [batch_size, 2]")
        return self._run(coordinates.float())


class CachedUserSequenceEncoder:
    """Batched ``UserSequenceEncoder.encode_user_sequence`` (models/user_sequence_encoder.py:101-142) + the collate
    padding of training/user_qformer_training.py:138-163, from CACHED item query tokens (data.ItemTokenCache) instead
    of re-running the frozen modality encoders: tokens + (time + geo) context, flatten, positional encoding, pad, mask
    -- one gather, two tiny MLPs and one assembly kernel per batch of users, no per-user Python tensor work."""

    def __init__(self, token_cache, embedding_dim=1024, dropout=0.1, training=False):
        self.token_cache = token_cache
        self.embedding_dim = embedding_dim
        self.timestamp_encoder = TimestampEncoder(embedding_dim)
        self.geo_encoder = GeoCoordinateEncoder(embedding_dim)
        self.assembler = UserSequenceAssembler(embedding_dim, token_cache.tokens.shape[1], dropout, training)

    def to(self, device):
        self.timestamp_encoder.to(device)
        self.geo_encoder.to(device)
        return self

    def encode_user_sequences(self, user_histories, max_events):
        """user_histories: list (users) of lists of events {'item_id', 'timestamp', 'coordinates': (lat, lon)}.
        -> (padded_inputs [B, max_events*Q, H] bf16, attention_mask [B, max_events*Q] float)."""
        dev = self.token_cache.tokens.device
        B, L = len(user_histories), max_events
        ts = torch.zeros((B, L), dtype=torch.float64)
        co = torch.zeros((B, L, 2), dtype=torch.float32)
        for b, h in enumerate(user_histories):
            for i, ev in enumerate(h[:L]):
                ts[b, i] = float(ev["timestamp"])
                co[b, i, 0], co[b, i, 1] = float(ev["coordinates"][0]), float(ev["coordinates"][1])
        toks, n = self.token_cache.history_tokens([[ev["item_id"] for ev in h] for h in user_histories], L)
        ctx = hip.add_bf16(self.timestamp_encoder(ts.to(dev).view(-1)), self.geo_encoder(co.to(dev).view(-1, 2)))
        return self.assembler.encode_user_sequences(toks, ctx.view(B, L, self.embedding_dim), n.to(dev))


class UserSequenceEncoder:
    """Drop-in for ``models/user_sequence_encoder.py:36-142``: same constructor arguments, attributes
    (``item_encoder``, ``item_qformer``, ``item_qformer_fields``, ``embedding_dim``, ``timestamp_encoder``, ``geo_encoder``,
    ``positional_encoder``, ``device``) and methods (``_get_item_query_tokens_batch``, ``encode_user_sequence``), so
    ``training/user_qformer_training.py:138-163`` (the collate) calls it unchanged.  The item Q-Former, the two context
    MLPs and the sequence assembly run on the HIP path; the frozen modality encoders behind ``ItemEncoder`` are upstream
    (see item_encoder_pure_value.py: inject a backend or a field cache via ``item_encoder=`` / ``item_encoder_kwargs=``).

    Deliberate difference: the reference rebuilds the item Q-Former with DEFAULT hyper-parameters and so only
    loads checkpoints trained with the defaults (:65); here ``checkpoint['config']`` -- the pickled BertConfig that
    training/item_qformer_training.py:178-186 stores -- is honoured when present (defaults otherwise, i.e. the
    reference's behaviour for the checkpoints it can load)."""

    def __init__(self, item_qformer_checkpoint_path: str, item_encoder_config_path: str, item_encoder=None, item_encoder_kwargs=None):
        from .item_encoder_pure_value import ItemEncoder
        if not torch.cuda.is_available():
            raise hip._lib.UniRecHipError("UserSequenceEncoder runs on the MI355X only (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.item_encoder = item_encoder if item_encoder is not None else ItemEncoder(config_path=item_encoder_config_path,
                                                                                    **(item_encoder_kwargs or {}))
        self._load_item_qformer(item_qformer_checkpoint_path)
        self.embedding_dim = self.item_qformer.config.hidden_size
        self.timestamp_encoder = TimestampEncoder(embedding_dim=self.embedding_dim).to(self.device)
        self.geo_encoder = GeoCoordinateEncoder(embedding_dim=self.embedding_dim).to(self.device)
        self.positional_encoder = PositionalEncoding(d_model=self.embedding_dim).to(self.device)
        self._step = 0

    def _load_item_qformer(self, checkpoint_path: str):
        """:56-70.  The checkpoint holds a pickled config object, hence weights_only=False (as the reference's plain
        torch.load did when it was written); only load checkpoints you trust."""
        from .qformer_utils import QFormerForItemRepresentation
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        num_fields = len(checkpoint["field_names"])
        cfg = checkpoint.get("config", None)
        kw = {}
        if cfg is not None:
            get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
            for ours, theirs in (("hidden_size", "hidden_size"), ("num_hidden_layers", "num_hidden_layers"),
                                 ("num_attention_heads", "num_attention_heads"), ("intermediate_size", "intermediate_size"),
                                 ("num_query_tokens", "query_length"), ("field_embedding_dim", "encoder_width"),
                                 ("dropout", "hidden_dropout_prob")):
                v = get(theirs)
                if v is not None:
                    kw[ours] = v
        self.item_qformer = QFormerForItemRepresentation(num_fields=num_fields, **kw).to(self.device)
        self.item_qformer.load_state_dict(checkpoint["model_state_dict"])
        self.item_qformer.eval()
        self.item_qformer_fields = checkpoint["field_names"]

    @torch.no_grad()
    def _get_item_query_tokens_batch(self, item_samples):
        """:72-99 -- field vectors from the ItemEncoder, mask = np.any per field, item Q-Former (eval) -> query_outputs
        [n, Q, H] float32 on the device."""
        import numpy as np
        encoded = self.item_encoder.encode_batch_by_field(item_samples, self.item_qformer_fields)
        block = np.stack([np.asarray(encoded[f], dtype=np.float32)[:len(item_samples)] for f in self.item_qformer_fields], axis=1)   # [n,F,E]
        mask = np.any(block != 0, axis=-1).astype(np.int64)
        x = torch.from_numpy(block).to(self.device)
        m = torch.from_numpy(mask).to(self.device)
        return self.item_qformer(x, m)["query_outputs"]

    def encode_user_sequence(self, user_history):
        """:101-142 -- [n_events * Q, H] float32: item tokens + (time + geo) context, flattened, + sinusoidal positional
        encoding over the flat index (+ the positional encoder's dropout while it is in training mode)."""
        Q = self.item_qformer.num_query_tokens
        if not user_history:
            return torch.empty(0, Q, self.embedding_dim)
        item_samples = [event["item_data"] for event in user_history]
        timestamps = torch.tensor([event["timestamp"] for event in user_history], device=self.device)
        coords = torch.tensor([event["coordinates"] for event in user_history], device=self.device)
        toks = self._get_item_query_tokens_batch(item_samples)                                   # [L,Q,H] f32
        with torch.no_grad():
            ctx = hip.add_bf16(self.timestamp_encoder(timestamps), self.geo_encoder(coords))   # [L,H] bf16
        L = len(user_history)
        t16 = hip.cast_f32_to_bf16(toks.detach().contiguous()).view(1, L, Q, self.embedding_dim)
        self._step += 1
        p = self.positional_encoder.p if self.positional_encoder.training else 0.0
        out, _ = hip.user_sequence_assemble(t16, ctx.view(1, L, self.embedding_dim).contiguous(),
                                            torch.tensor([L], dtype=torch.int32, device=self.device), p, 0xA55E + self._step)
        return out.view(L * Q, self.embedding_dim).float()

"""Cache formats and batch assembly either side of the hot path.

Mirrors, with the reference's names and on-disk formats:
  * ``QFormerDataset``           models/qformer_utils.py:62-155  (field-embedding cache: ``embeddings.pt`` dict[int ->
                                  [F,1024] f32], ``masks.pt`` dict[int -> [F] i64], ``fields.json``)
  * history batch assembly        training/train_item_individual_token_joint.py:557-577 (_get_history_qformer_inputs)
  * ``run_inference`` token cache data_processing/qformer_inference.py:112-176 (pickle: item_id -> np.float32 [Q,H])

MI355X-first difference: the cache lives PACKED ([N,F,E] / [N,F] / [N,Q,H], one allocation each) and resident in
HBM; a training batch is one index tensor plus one gather kernel (``ur_gather_rows``) instead of ``B * hist``
per-sample ``torch.stack`` calls on the host, and field vectors leave the gather already in bf16, the dtype the
Q-Former kernels consume.  The frozen modality encoders that produce the field vectors are upstream and out of
scope: ``item_encoder`` is whatever object offers ``encode_batch_by_field(samples, fields)`` (:106).
"""
import json
import os
import pickle
from collections.abc import Mapping

import numpy as np
import torch

from . import hip


class _RowsAsDict(Mapping):
    """dict[int -> row tensor] view of a packed [N, ...] tensor (the reference's cache objects are such dicts)."""

    def __init__(self, packed):
        self._t = packed

    def __getitem__(self, idx):
        if not (0 <= int(idx) < self._t.shape[0]):
            raise KeyError(idx)
        return self._t[int(idx)]

    def __iter__(self):
        return iter(range(self._t.shape[0]))

    def __len__(self):
        return self._t.shape[0]


class QFormerDataset(torch.utils.data.Dataset):
    """models/qformer_utils.py:62-155 with a packed cache.  ``embedding_cache`` / ``mask_cache`` keep their dict
    interface; ``packed(device)`` hands the whole cache to the GPU as a ``PackedFieldStore``."""

    def __init__(self, samples, item_encoder, cache_dir=None, precompute_batch_size=8192):
        self.samples = samples
        self.item_encoder = item_encoder
        self.cache_dir = cache_dir
        self.precompute_batch_size = precompute_batch_size
        self.available_fields = self._analyze_fields()
        self.fields = None          # [N,F,E] f32 (host)
        self.masks = None           # [N,F] i64 (host)
        if not (cache_dir and self._load_cache()):
            self._precompute()
            if cache_dir:
                self._save_cache()

    # -- reference attribute names ---------------------------------------------------------------
    @property
    def embedding_cache(self):
        return _RowsAsDict(self.fields)

    @property
    def mask_cache(self):
        return _RowsAsDict(self.masks)

    def _analyze_fields(self):
        """:83-96 -- every key of every sample except item_id, sorted."""
        all_fields = set()
        for sample in self.samples:
            all_fields.update(sample.keys())
        return sorted(f for f in all_fields if f != "item_id")

    def _precompute(self):
        """:98-124 -- batched encode_batch_by_field; a field is valid iff its vector is not all zeros (:116)."""
        n, bs, F = len(self.samples), self.precompute_batch_size, len(self.available_fields)
        fields, masks = None, torch.zeros((n, F), dtype=torch.long)
        for i in range(0, n, bs):
            batch = self.samples[i:i + bs]
            enc = self.item_encoder.encode_batch_by_field(batch, self.available_fields)
            block = np.stack([np.asarray(enc[f], dtype=np.float32)[:len(batch)] for f in self.available_fields], axis=1)   # [b,F,E]
            if fields is None:
                fields = torch.zeros((n, F, block.shape[-1]), dtype=torch.float32)
            fields[i:i + len(batch)] = torch.from_numpy(block)
            masks[i:i + len(batch)] = torch.from_numpy(np.any(block != 0, axis=-1).astype(np.int64))
        if fields is None:
            fields = torch.zeros((0, F, 0), dtype=torch.float32)
        self.fields, self.masks = fields, masks

    def _load_cache(self):
        """:126-144 -- valid only when fields.json equals the current field list."""
        emb_path, mask_path = os.path.join(self.cache_dir, "embeddings.pt"), os.path.join(self.cache_dir, "masks.pt")
        fields_path = os.path.join(self.cache_dir, "fields.json")
        if not (os.path.exists(emb_path) and os.path.exists(mask_path) and os.path.exists(fields_path)):
            return False
        with open(fields_path, "r") as f:
            if json.load(f) != self.available_fields:
                return False
        emb, msk = torch.load(emb_path), torch.load(mask_path)
        n = len(emb)
        self.fields = torch.stack([emb[i].to(torch.float32) for i in range(n)]) if n else torch.zeros((0, len(self.available_fields), 0))
        self.masks = torch.stack([msk[i].to(torch.long) for i in range(n)]) if n else torch.zeros((0, len(self.available_fields)), dtype=torch.long)
        return True

    def _save_cache(self):
        """:146-151 -- the reference's dict-of-tensors files, so its own loader reads them."""
        os.makedirs(self.cache_dir, exist_ok=True)
        n = self.fields.shape[0]
        torch.save({i: self.fields[i].clone() for i in range(n)}, os.path.join(self.cache_dir, "embeddings.pt"))
        torch.save({i: self.masks[i].clone() for i in range(n)}, os.path.join(self.cache_dir, "masks.pt"))
        with open(os.path.join(self.cache_dir, "fields.json"), "w") as f:
            json.dump(self.available_fields, f)

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        return {"field_embeddings": self.fields[idx], "attention_mask": self.masks[idx],
                "item_id": self.samples[idx].get("item_id", str(idx))}

    def packed(self, device="cuda"):
        ids = [str(s.get("item_id", str(i))) for i, s in enumerate(self.samples)]
        return PackedFieldStore(self.fields, self.masks, ids, device)


class PackedFieldStore:
    """The field-embedding cache resident in HBM: fields [N,F,E] f32, masks [N,F] u8, item_id -> row."""

    def __init__(self, fields, masks, item_ids, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise hip._lib.UniRecHipError("PackedFieldStore lives in HBM (the UniRec HIP path has no CPU fallback)")
        self.fields = fields.to(dev, torch.float32).contiguous()
        self.masks = (masks != 0).to(dev, torch.uint8).contiguous()
        self.item_id_to_idx = {str(k): i for i, k in enumerate(item_ids)}      # :551
        self.num_fields, self.field_dim = self.fields.shape[1], self.fields.shape[2]

    def __len__(self):
        return self.fields.shape[0]

    def history_index(self, histories, num_history_items):
        """[B,hist] int64 rows for a batch of history id lists: -1 for empty slots and unknown items (:560-575)."""
        idx = np.full((len(histories), num_history_items), -1, dtype=np.int64)
        for b, h in enumerate(histories):
            for i, item_id in enumerate(h[:num_history_items]):
                idx[b, i] = self.item_id_to_idx.get(str(item_id), -1)
        return torch.from_numpy(idx)

    def gather(self, index, dtype=torch.bfloat16):
        """index [...] int64 -> (field_embeddings [...,F,E] in `dtype` (bf16 or f32), attention_mask [...,F] int64)."""
        idx = index.to(self.fields.device)
        emb = hip.gather_rows(self.fields, idx, out_dtype=dtype)
        msk = hip.gather_rows(self.masks, idx)
        return emb, msk.to(torch.long)

    def history_inputs(self, histories, num_history_items, dtype=torch.bfloat16):
        """(history_field_embeddings [B,hist,F,E], history_attention_mask [B,hist,F]) of the joint batch (:557-577 + collate)."""
        return self.gather(self.history_index(histories, num_history_items), dtype)


class ItemTokenCache:
    """Cached item query tokens [N,Q,H] in HBM: what data_processing/qformer_inference.py:112-176 writes as
    a pickle of item_id -> np.float32 [Q,H] and what training/user_qformer_training.py / ValidationDataset (:246-255)
    read back per history item.  ``build`` runs the item Q-Former in eval mode over a PackedFieldStore in batches."""

    def __init__(self, tokens, item_ids):
        self.tokens = tokens                   # [N,Q,H] bf16, device
        self.item_ids = [str(k) for k in item_ids]
        self.item_id_to_idx = {k: i for i, k in enumerate(self.item_ids)}

    @classmethod
    def build(cls, model, store, batch_size=4096):
        was_training = model.training
        model.eval()
        outs = []
        with torch.no_grad():
            for i in range(0, len(store), batch_size):
                idx = torch.arange(i, min(len(store), i + batch_size), dtype=torch.int64)
                emb, msk = store.gather(idx)
                outs.append(model.encode_bf16(emb, msk).detach())
        model.train(was_training)
        ids = sorted(store.item_id_to_idx, key=store.item_id_to_idx.get)
        q, h = model.num_query_tokens, model.config.hidden_size
        return cls(torch.cat(outs, 0) if outs else torch.zeros((0, q, h), dtype=torch.bfloat16, device=store.fields.device), ids)

    def save(self, output_path):
        """The reference's pickle (:172-174): item_id -> np.float32 [Q,H]."""
        os.makedirs(os.path.dirname(os.path.abspath(output_path)), exist_ok=True)
        t = self.tokens.float().cpu().numpy()
        with open(output_path, "wb") as f:
            pickle.dump({k: t[i] for i, k in enumerate(self.item_ids)}, f)

    @classmethod
    def load(cls, path, device="cuda"):
        with open(path, "rb") as f:
            d = pickle.load(f)
        ids = list(d.keys())
        t = torch.from_numpy(np.stack([np.asarray(d[k], dtype=np.float32) for k in ids])) if ids else torch.zeros((0, 0, 0))
        return cls(t.to(device).to(torch.bfloat16).contiguous(), ids)

    def history_tokens(self, histories, num_history_items):
        """[B,hist,Q,H] bf16 query tokens of a batch of history id lists, zeros for empty slots / unknown items
        (:241-255), plus the per-user number of real events [B]."""
        idx = np.full((len(histories), num_history_items), -1, dtype=np.int64)
        n = np.zeros((len(histories),), dtype=np.int64)
        for b, h in enumerate(histories):
            n[b] = min(len(h), num_history_items)
            for i, item_id in enumerate(h[:num_history_items]):
                idx[b, i] = self.item_id_to_idx.get(str(item_id), -1)
        return hip.gather_rows(self.tokens, torch.from_numpy(idx)), torch.from_numpy(n)

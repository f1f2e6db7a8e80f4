"""Boundary surface of ``models/item_encoder_pure_value.py:15-465`` (``ItemEncoder``).

The reference class owns three FROZEN modality encoders (SentenceTransformer Qwen3-Embedding-0.6B for text /
category, CLIP ViT-L/14 zero-padded 768 -> 1024 for images, the MWNE Fourier encoder for numbers) that it downloads
from the network at construction time (:43-72).  They are upstream of the hot path and out of scope (SURVEY.md section 2
row 9, section 8(b)): neither the weights nor a network exist where this package runs.  What the hot path's callers touch is
the *surface*: ``ItemEncoder(config_path)``, ``.embedding_dim``, ``.field_mapping``, ``.modality_ids``,
``.encode_batch_by_field(samples, fields) -> {field: np.float32[B, 1024]}`` (:359-389), ``.encode_batch`` (:391-409),
``.get_embedding_dimensions`` (:411-...).  This class keeps that surface, the YAML configuration format
(config/triplet_config.yaml) and the per-modality dispatch, and takes the field vectors from one of two sources the
caller supplies:

  * ``backend``: an object with ``encode_text_batch(list[str])``, ``encode_image_batch(list)``,
    ``encode_number_batch(list)`` (each -> np.float32 [B, embedding_dim]) -- e.g. the reference's own encoders wrapped
    by a deployment that has the weights -- or with ``encode_batch_by_field`` itself;
  * ``field_cache``: a mapping ``item_id -> {field: vector}`` or a ``(QFormerDataset-style) PackedFieldStore`` whose rows
    were produced offline (training/precompute_full_field_embeddings.py) -- the normal case for training, where the field
    vectors are cached anyway.

With neither, ``encode_batch_by_field`` raises ``ModalityEncodersUnavailable`` naming what is missing instead of trying
to download models.
"""
from typing import Any, Dict, List

import numpy as np
import yaml


class ModalityEncodersUnavailable(RuntimeError):
    pass


class ItemEncoder:
    def __init__(self, config_path: str = "config/triplet_config.yaml", backend=None, field_cache=None):
        self.config_path = config_path
        self.field_mapping = None
        self.modality_ids = None
        self.embedding_dim = 1024
        self.backend = backend
        self.field_cache = field_cache
        self._load_config()

    def _load_config(self):
        """:34-41 -- FIELD_MAPPING: field -> [field_id, modality_id, modality_type]; MODALITY_IDS."""
        with open(self.config_path, "r") as f:
            config = yaml.safe_load(f)
        self.field_mapping = config["FIELD_MAPPING"]
        self.modality_ids = config["MODALITY_IDS"]

    # ---- sources of field vectors -----------------------------------------------------------------------------
    def _from_cache(self, samples, field_name):
        out = np.zeros((len(samples), self.embedding_dim), dtype=np.float32)
        store = self.field_cache
        packed = hasattr(store, "item_id_to_idx") and hasattr(store, "fields")
        if packed:
            names = getattr(store, "field_names", None)
            if names is None:
                raise ModalityEncodersUnavailable("a PackedFieldStore used as ItemEncoder.field_cache needs `.field_names`")
            if field_name not in names:
                return out
            col = names.index(field_name)
        for i, s in enumerate(samples):
            key = str(s.get("item_id", ""))
            if packed:
                row = store.item_id_to_idx.get(key, -1)
                if row >= 0:
                    out[i] = store.fields[row, col].float().cpu().numpy()
            else:
                rec = store.get(key) if hasattr(store, "get") else None
                if rec is not None and field_name in rec and rec[field_name] is not None:
                    out[i] = np.asarray(rec[field_name], dtype=np.float32)
        return out

    def _encode_modality(self, modality_type, field_name, data_batch, samples):
        if self.field_cache is not None:
            return self._from_cache(samples, field_name)
        b = self.backend
        if b is None:
            raise ModalityEncodersUnavailable(
                f"ItemEncoder cannot encode field '{field_name}' ({modality_type}): the frozen modality encoders of "
                "models/item_encoder_pure_value.py:43-72 (SentenceTransformer Qwen/Qwen3-Embedding-0.6B, openai/clip-vit-large-patch14, "
                "number_encoders/mathematical_encoder_1024d_normalized.pth) are upstream of the MI355X hot path and their weights are "
                "not available here.  Pass backend=<object with encode_text_batch / encode_image_batch / encode_number_batch> or "
                "field_cache=<item_id -> {field: vector} | PackedFieldStore> to ItemEncoder.")
        if modality_type not in self._BACKEND_METHOD:
            raise ValueError(f"Unknown modality type: {modality_type}")
        return np.asarray(getattr(b, self._BACKEND_METHOD[modality_type])(data_batch), dtype=np.float32)

    # modality (third entry of a FIELD_MAPPING value) -> the backend method that embeds a batch of raw field values
    _BACKEND_METHOD = {"text": "encode_text_batch", "category": "encode_text_batch", "image": "encode_image_batch", "number": "encode_number_batch"}

    # ---- reference surface ------------------------------------------------------------------------------------
    def encode_batch_by_field(self, samples: List[Dict[str, Any]], fields_to_encode: List[str]) -> Dict[str, np.ndarray]:
        """Contract of models/item_encoder_pure_value.py:359-389: {field: float array [B, embedding_dim]}; a field that FIELD_MAPPING does
        not know is all zeros (and is reported once per call), an unknown modality raises ValueError, an empty batch gives empty arrays.
        Columns come from the field cache when there is one, else from the backend's method for the field's modality."""
        if not samples:
            return {name: np.array([]) for name in fields_to_encode}
        b = self.backend
        if b is not None and self.field_cache is None and hasattr(b, "encode_batch_by_field"):
            return b.encode_batch_by_field(samples, fields_to_encode)          # a backend that batches by field itself
        modality = {name: (self.field_mapping[name][2] if self.field_mapping.get(name) else None) for name in fields_to_encode}
        bad = [m for m in modality.values() if m is not None and m not in self._BACKEND_METHOD]
        if bad:
            raise ValueError(f"Unknown modality type: {bad[0]}")
        for name in (n for n, m in modality.items() if m is None):
            print(f"Warning: Field '{name}' not in field_mapping. Skipping.")
        zeros = np.zeros((len(samples), self.embedding_dim))
        return {name: zeros.copy() if m is None else self._encode_modality(m, name, [s.get(name, "") for s in samples], samples)
                for name, m in modality.items()}

    def encode_batch(self, samples: List[Dict[str, Any]]) -> List[Dict[str, np.ndarray]]:
        """Contract of :391-409: the per-field columns of encode_batch_by_field over every mapped field, regrouped per sample."""
        if not samples:
            return []
        columns = self.encode_batch_by_field(samples, list(self.field_mapping.keys()))
        return [{name: col[i] for name, col in columns.items()} for i in range(len(samples))]

    def get_embedding_dimensions(self) -> Dict[str, int]:
        """every field vector is 1024-d whatever its modality (CLIP's 768 are zero-padded, :163,257)."""
        return {field_name: self.embedding_dim for field_name in self.field_mapping}

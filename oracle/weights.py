"""Deterministic weight generator owned by the build (oracle side).

A tensor is a pure function of (name, shape, seed, std): fixtures therefore
only need to hold seeds + outputs, never weights.
Test infrastructure only -- see oracle/__init__.py.
"""
import hashlib
import numpy as np


def _rng(name: str, seed: int) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.Philox(key=int.from_bytes(h[:8], "little")))


def normal(name: str, shape, seed: int, std: float = 0.02, mean: float = 0.0) -> np.ndarray:
    return (mean + std * _rng(name, seed).standard_normal(size=tuple(shape))).astype(np.float32)


def uniform(name: str, shape, seed: int, lo: float, hi: float) -> np.ndarray:
    return _rng(name, seed).uniform(lo, hi, size=tuple(shape)).astype(np.float32)


def fill_state_dict(shapes: dict, seed: int, rules=None) -> dict:
    """shapes: {key: shape}.  Default rule mirrors the reference init
    (models/qformer.py:664-674: Linear/Embedding N(0,0.02), bias 0, LN (1,0);
    models/qformer_utils.py:28: query_embeddings N(0,1)) but with NON-trivial
    biases / LN affine so every term is numerically exercised."""
    out = {}
    for k, shp in shapes.items():
        if rules is not None:
            r = rules(k, shp)
            if r is not None:
                out[k] = r
                continue
        if k.endswith("position_ids"):
            out[k] = np.arange(shp[-1], dtype=np.int64).reshape(shp)
        elif k == "query_embeddings" or k.endswith(".query_embeddings"):
            out[k] = normal(k, shp, seed, std=1.0)
        elif "LayerNorm.weight" in k or k.endswith("norm.weight") or k.endswith("layernorm.weight") \
                or (k.startswith("prediction_head.2") and k.endswith("weight")):
            out[k] = normal(k, shp, seed, std=0.1, mean=1.0)
        elif k.endswith(".bias"):
            out[k] = normal(k, shp, seed, std=0.02)
        else:
            out[k] = normal(k, shp, seed, std=0.02)
    return out

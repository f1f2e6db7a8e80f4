"""CPU: the data path either side of the hot path -- the oracle restatement and
the host side of the product classes against golden vectors produced by the reference's own classes
(tests/golden/make_golden_data.py)."""
import json
import os

import numpy as np
import torch

from oracle import data_ref as D
from tests.golden import data_cases as dc

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "data_path.npz"))
T = json.load(open(os.path.join(HERE, "golden", "data_path.json")))


def test_oracle_field_cache_matches_reference():
    samples, _ = dc.item_samples()
    fields = D.analyze_fields(samples)
    assert fields == T["available_fields"]
    emb, msk = D.precompute_cache(samples, dc.FakeItemEncoder().encode_batch_by_field, fields, 5)
    assert np.array_equal(emb, G["embeddings"]) and np.array_equal(msk, G["masks"])
    assert msk[5, fields.index("brand")] == 0 and msk[9].sum() == 2          # missing fields are masked out


def test_oracle_history_assembly_and_prompt_match_reference():
    samples, item_dict = dc.item_samples()
    idx = {s["item_id"]: i for i, s in enumerate(samples)}
    for b, h in enumerate(dc.histories()):
        e, m = D.history_qformer_inputs(h, idx, G["embeddings"], G["masks"], dc.HIST)
        assert np.array_equal(e, G["history_field_embeddings"][b]) and np.array_equal(m, G["history_attention_mask"][b])
        assert D.construct_input_text(h, item_dict, dc.HIST, dc.QI) == T["input_texts"][b]


def test_oracle_mrr_matches_reference():
    users, pos, negs = dc.mrr_inputs()
    ranks = [D.mrr_rank(users[b], pos[b], negs[b])[0] for b in range(len(users))]
    assert np.allclose(1.0 / np.array(ranks), G["batch_mrr"], rtol=0, atol=1e-12)
    assert ranks[2] >= 2                                                       # the planted stronger negative counts


def test_product_dataset_host_side_and_cache_format(tmp_path):
    """QFormerDataset (no GPU involved): same fields / vectors / masks as the reference, the reference's cache files,
    and a cache written in the reference's dict-of-tensors format loads back."""
    from unirec_amd.data import QFormerDataset
    from unirec_amd.evaluation import construct_input_text
    samples, item_dict = dc.item_samples()
    cache = str(tmp_path / "cache")
    ds = QFormerDataset(samples, dc.FakeItemEncoder(), cache_dir=cache, precompute_batch_size=5)
    assert ds.available_fields == T["available_fields"] and len(ds) == len(samples)
    assert np.array_equal(ds.fields.numpy(), G["embeddings"]) and np.array_equal(ds.masks.numpy(), G["masks"])
    assert sorted(os.listdir(cache)) == T["cache_files"]
    it = ds[3]
    assert sorted(it.keys()) == T["getitem_keys"] and it["item_id"] == T["getitem_item_id"]
    assert torch.equal(ds.embedding_cache[3], it["field_embeddings"]) and len(ds.mask_cache) == len(samples)
    emb = torch.load(os.path.join(cache, "embeddings.pt"))
    assert isinstance(emb, dict) and emb[0].dtype == torch.float32 and tuple(emb[0].shape) == G["embeddings"].shape[1:]

    class Boom:
        def encode_batch_by_field(self, *a):
            raise AssertionError("a valid cache must not re-encode")
    ds2 = QFormerDataset(samples, Boom(), cache_dir=cache)
    assert np.array_equal(ds2.fields.numpy(), G["embeddings"]) and np.array_equal(ds2.masks.numpy(), G["masks"])
    # a different field list invalidates the cache (qformer_utils.py:137-143)
    other = [dict(s, extra="1") for s in samples]
    try:
        QFormerDataset(other, Boom(), cache_dir=cache)
        assert False, "stale cache was accepted"
    except AssertionError as e:
        assert "re-encode" in str(e)
    for b, h in enumerate(dc.histories()):
        assert construct_input_text(h, item_dict, dc.HIST, dc.QI) == T["input_texts"][b]


def test_oracle_context_encoders_match_reference():
    """TimestampEncoder / GeoCoordinateEncoder (models/mwne.py:504-607) and one user's encode_user_sequence
    (models/user_sequence_encoder.py:101-142, dropout off) from cached item tokens."""
    from oracle import weights as W
    from oracle import qformer_ref as R
    H = dc.CTX_H
    PT = W.fill_state_dict(D.context_mlp_shapes(H, 9), dc.CTX_SEED)
    PG = W.fill_state_dict(D.context_mlp_shapes(H, 3), dc.CTX_SEED + 1)
    ts, co = dc.context_inputs()
    te, ge = D.context_mlp(D.timestamp_features(ts), PT), D.context_mlp(D.geo_features(co), PG)
    assert np.abs(te - G["time_emb"]).max() < 2e-4 * np.abs(G["time_emb"]).max()
    assert np.abs(ge - G["geo_emb"]).max() < 2e-5
    ev = dc.user_events()[0]
    toks = dc.event_tokens()
    ctx = torch.from_numpy(te[:len(ev)] + ge[:len(ev)])
    seq = R.assemble_user_sequence(torch.from_numpy(np.stack([toks[e["item_id"]] for e in ev])), ctx).numpy()
    assert np.abs(seq - G["encoded_user_sequence"]).max() < 5e-4

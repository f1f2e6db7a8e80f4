#!/usr/bin/env python3
"""Golden vectors for the data path either side of the hot path, produced by the
REFERENCE's own classes on CPU (build container only; same shim as make_golden.py).  Writes data_path.npz / .json
(inputs are regenerated from seeds by tests/golden/data_cases.py; only expected outputs are stored).

Usage:  python tests/golden/make_golden_data.py
"""
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import numpy as np
import torch

from tests.golden import make_golden as mg          # installs the shim and imports the reference modules
from tests.golden import data_cases as dc

rj = mg.rj
from models.qformer_utils import QFormerDataset as RefQFormerDataset   # noqa: E402  (reference, via the shim's sys.path)


def main():
    res, txt = {}, {}
    samples, item_dict = dc.item_samples()
    enc = dc.FakeItemEncoder()
    with tempfile.TemporaryDirectory() as td:
        ds = RefQFormerDataset(samples, enc, cache_dir=td, precompute_batch_size=5)
        txt["available_fields"] = ds.available_fields
        n = len(ds)
        res["embeddings"] = np.stack([ds.embedding_cache[i].numpy() for i in range(n)])
        res["masks"] = np.stack([ds.mask_cache[i].numpy() for i in range(n)])
        txt["cache_files"] = sorted(os.listdir(td))
        item0 = ds[3]
        txt["getitem_keys"] = sorted(item0.keys())
        txt["getitem_item_id"] = item0["item_id"]
        # joint dataset helpers on an instance built without its heavy constructor
        jd = rj.AmazonBeautyJointDataset.__new__(rj.AmazonBeautyJointDataset)
        jd.item_dict = item_dict
        jd.num_history_items, jd.num_query_tokens_per_item = dc.HIST, dc.QI
        jd.qformer_field_dataset = ds
        jd.item_id_to_idx = {s["item_id"]: i for i, s in enumerate(samples)}
        F, E = res["embeddings"].shape[1:]
        jd.zero_field_embeddings = torch.zeros((F, E), dtype=torch.float32)
        jd.zero_attention_mask = torch.zeros(F, dtype=torch.long)
        hs = dc.histories()
        he, hm, texts = [], [], []
        for h in hs:
            e, m = jd._get_history_qformer_inputs(h)
            he.append(e.numpy()); hm.append(m.numpy())
            texts.append(jd._construct_input_text(h))
        res["history_field_embeddings"] = np.stack(he)
        res["history_attention_mask"] = np.stack(hm)
        txt["input_texts"] = texts
    # MRR evaluator on a fake model that returns fixed user embeddings
    users, pos, negs = dc.mrr_inputs()

    class FakeModel:
        def eval(self):
            return self

        def __call__(self, **kw):
            return torch.from_numpy(users)
    ev = rj.MRREvaluator(FakeModel(), None, None)
    batch = {"input_ids": torch.zeros((len(users), 4), dtype=torch.long), "attention_mask": torch.ones((len(users), 4), dtype=torch.long),
             "history_field_embeddings": torch.zeros((len(users), 1, 1, 1)), "history_attention_mask": torch.zeros((len(users), 1, 1)),
             "positive_item_embeddings": torch.from_numpy(pos), "negative_item_embeddings": [torch.from_numpy(n) for n in negs]}
    with torch.no_grad():
        res["batch_mrr"] = np.array(ev._compute_batch_mrr(batch), dtype=np.float64)
    # event-context encoders and one user's encoded sequence (models/user_sequence_encoder.py:101-142)
    import types
    from models.mwne import TimestampEncoder as RefTime, GeoCoordinateEncoder as RefGeo
    import models.user_sequence_encoder as ruse
    from oracle import weights as W
    from oracle import data_ref as D
    H = dc.CTX_H
    te, ge = RefTime(H), RefGeo(H)
    te.load_state_dict({k: torch.from_numpy(v) for k, v in W.fill_state_dict(D.context_mlp_shapes(H, 9), dc.CTX_SEED).items()})
    ge.load_state_dict({k: torch.from_numpy(v) for k, v in W.fill_state_dict(D.context_mlp_shapes(H, 3), dc.CTX_SEED + 1).items()})
    ts, co = dc.context_inputs()
    with torch.no_grad():
        res["time_emb"] = te(torch.tensor(ts)).numpy()
        res["geo_emb"] = ge(torch.tensor(co)).numpy()
    toks = dc.event_tokens()
    use = ruse.UserSequenceEncoder.__new__(ruse.UserSequenceEncoder)
    use.device = torch.device("cpu")
    use.embedding_dim = H
    use.timestamp_encoder, use.geo_encoder = te, ge
    use.positional_encoder = ruse.PositionalEncoding(d_model=H).eval()          # dropout off for the fixture
    use.item_qformer = types.SimpleNamespace(num_query_tokens=dc.QI)
    use._get_item_query_tokens_batch = lambda item_samples: torch.from_numpy(np.stack([toks[s["item_id"]] for s in item_samples]))
    hist = dc.user_events()[0]
    with torch.no_grad():
        res["encoded_user_sequence"] = use.encode_user_sequence(hist).numpy()
    np.savez_compressed(os.path.join(HERE, "data_path.npz"), **res)
    with open(os.path.join(HERE, "data_path.json"), "w") as f:
        json.dump(txt, f, indent=1)
    print({k: v.shape for k, v in res.items()}, list(txt))


if __name__ == "__main__":
    main()

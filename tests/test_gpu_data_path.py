"""GPU: packed caches, gather kernel and the ranking evaluators against the
reference's golden vectors and the CPU oracle.  Index / rank / top-K results are exact."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import data_ref as D  # noqa: E402
from tests.golden import data_cases as dc  # noqa: E402

DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "data_path.npz"))


def _store():
    from unirec_amd.data import QFormerDataset
    samples, _ = dc.item_samples()
    return QFormerDataset(samples, dc.FakeItemEncoder(), precompute_batch_size=5).packed(DEV), samples


def test_history_inputs_match_reference_stack():
    store, _ = _store()
    e32, m = store.history_inputs(dc.histories(), dc.HIST, dtype=torch.float32)
    assert np.array_equal(e32.cpu().numpy(), G["history_field_embeddings"])
    assert m.dtype == torch.long and np.array_equal(m.cpu().numpy(), G["history_attention_mask"])
    e16, _ = store.history_inputs(dc.histories(), dc.HIST)
    assert e16.dtype == torch.bfloat16
    assert torch.equal(e16.cpu(), torch.from_numpy(G["history_field_embeddings"]).to(torch.bfloat16))


@pytest.mark.parametrize("dtype,row", [(torch.float32, 20), (torch.bfloat16, 4096), (torch.uint8, 14), (torch.float32, 14336)])
def test_gather_rows_all_kinds(dtype, row):
    from unirec_amd import hip
    g = torch.Generator().manual_seed(1)
    src = (torch.randn(37, row, generator=g) * 4).to(dtype).to(DEV)
    idx = torch.tensor([[0, 36, -1, 5], [99, 7, 7, -5]])
    out = hip.gather_rows(src, idx)
    want = torch.zeros((2, 4, row), dtype=dtype)
    for b in range(2):
        for i in range(4):
            j = int(idx[b, i])
            if 0 <= j < 37:
                want[b, i] = src[j].cpu()
    assert torch.equal(out.cpu(), want)
    if dtype == torch.float32 and row % 8 == 0:
        assert torch.equal(hip.gather_rows(src, idx, out_dtype=torch.bfloat16).cpu(), want.to(torch.bfloat16))


def test_item_token_cache_build_save_load_and_history(tmp_path):
    from unirec_amd.data import ItemTokenCache
    from unirec_amd.qformer_utils import QFormerForItemRepresentation
    store, samples = _store()
    torch.manual_seed(0)
    model = QFormerForItemRepresentation(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                         num_query_tokens=dc.QI, field_embedding_dim=dc.E, num_fields=store.num_fields, dropout=0.2).to(DEV).train()
    cache = ItemTokenCache.build(model, store, batch_size=5)
    assert model.training                                     # restored
    model.eval()
    with torch.no_grad():
        emb, msk = store.gather(torch.arange(len(store)))
        full = model.encode_bf16(emb, msk)
    assert torch.equal(cache.tokens, full)                    # batching does not change the tokens
    path = str(tmp_path / "tokens" / "query_tokens.pkl")
    cache.save(path)
    import pickle
    d = pickle.load(open(path, "rb"))
    assert sorted(d) == [s["item_id"] for s in samples] and d["B003"].dtype == np.float32 and d["B003"].shape == (dc.QI, 128)
    back = ItemTokenCache.load(path, DEV)
    assert torch.equal(back.tokens, cache.tokens)
    toks, n = cache.history_tokens(dc.histories(), dc.HIST)
    assert n.tolist() == [3, 0, dc.HIST, dc.HIST]
    for b, h in enumerate(dc.histories()):
        want = D.history_query_tokens(h, {k: v for k, v in d.items()}, dc.HIST, dc.QI, 128)
        assert np.array_equal(toks[b].float().cpu().numpy(), want)


def test_mrr_evaluator_matches_reference():
    from unirec_amd.evaluation import MRREvaluator
    users, pos, negs = dc.mrr_inputs()
    rank = MRREvaluator.ranks_from_embeddings(torch.from_numpy(users).to(DEV), pos, negs)
    assert np.allclose(1.0 / rank.cpu().numpy().astype(np.float64), G["batch_mrr"], rtol=0, atol=1e-12)

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1, device=DEV))

        def forward(self, **kw):
            return torch.from_numpy(users).to(DEV)
    batch = {"input_ids": torch.zeros((5, 4), dtype=torch.long), "attention_mask": torch.ones((5, 4), dtype=torch.long),
             "history_field_embeddings": torch.zeros((5, 1, 1, 1)), "history_attention_mask": torch.zeros((5, 1, 1)),
             "positive_item_embeddings": torch.from_numpy(pos), "negative_item_embeddings": [torch.from_numpy(n) for n in negs]}
    got = MRREvaluator(Fake())._compute_batch_mrr(batch)
    assert np.allclose(got, G["batch_mrr"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("B,N,Dm", [(5, 1000, 48), (37, 5003, 1024)])
def test_catalog_evaluator_matches_oracle(B, N, Dm):
    from unirec_amd.evaluation import CatalogEvaluator
    rng = np.random.RandomState(3)
    cat = rng.randn(N, Dm).astype(np.float32)
    gt = rng.randint(0, N, size=B)
    users = (cat[gt] * 0.5 + rng.randn(B, Dm) * 0.7).astype(np.float32)
    users[1] = cat[gt[1]] * 2.0                                # an exact hit: rank 1
    cat[7] = 0.0                                               # a zero vector: cosine 0 through the eps clamp, no NaN
    s, rank, order = D.catalog_eval(users, cat, gt, 10)
    ev = CatalogEvaluator(cat, device=DEV)
    out = ev.evaluate(torch.from_numpy(users).to(DEV), gt, k=10)
    got = ev.scores(torch.from_numpy(users).to(DEV)).cpu().numpy()          # second call reuses the catalogue norms
    assert np.isfinite(got).all() and np.abs(got - s).max() < 2e-6
    # exactness of the integer results needs the oracle's own ordering to be robust to 2e-6: check that first
    srt = -np.sort(-s, axis=1)[:, :11]
    robust = (np.abs(np.diff(srt, axis=1)).min() > 1e-5)
    assert out["rank"].cpu().tolist()[1] == 1
    if robust:
        assert out["rank"].cpu().numpy().tolist() == rank.tolist()
        assert out["topk_index"].cpu().numpy().tolist() == order.tolist()
    else:       # fall back to self-consistency on the device scores
        s2, rank2, order2 = got, None, np.argsort(-got, axis=1, kind="stable")[:, :10]
        assert out["topk_index"].cpu().numpy().tolist() == order2.tolist()
    assert abs(out["mrr"] - float(np.mean(1.0 / out["rank"].cpu().numpy()))) < 1e-12
    assert abs(out["hit_at_k"] - float(np.mean(out["rank"].cpu().numpy() <= 10))) < 1e-12


def test_special_token_positions():
    from unirec_amd.evaluation import special_token_positions
    ids = torch.tensor([[5, 100, 7, 101, 9, 103], [102, 1, 1, 100, 1, 1]], device=DEV)
    pos = special_token_positions(ids, 100, 4)
    assert pos.cpu().tolist() == [[1, 3, -1, 5], [3, -1, 0, -1]]


def test_cached_user_sequence_encoder_matches_reference():
    """timestamp / geo encoders and the batched sequence encoding from the token cache against the
    reference's own TimestampEncoder / GeoCoordinateEncoder / encode_user_sequence outputs (golden)."""
    from oracle import weights as W
    from unirec_amd.data import ItemTokenCache
    from unirec_amd.user_sequence_encoder import CachedUserSequenceEncoder
    H = dc.CTX_H
    toks = dc.event_tokens()
    ids = sorted(toks)
    cache = ItemTokenCache(torch.from_numpy(np.stack([toks[k] for k in ids])).to(DEV).to(torch.bfloat16), ids)
    enc = CachedUserSequenceEncoder(cache, embedding_dim=H).to(DEV)
    enc.timestamp_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in W.fill_state_dict(D.context_mlp_shapes(H, 9), dc.CTX_SEED).items()})
    enc.geo_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in W.fill_state_dict(D.context_mlp_shapes(H, 3), dc.CTX_SEED + 1).items()})
    enc.to(DEV)
    ts, co = dc.context_inputs()
    te = enc.timestamp_encoder(torch.tensor(ts).to(DEV)).float().cpu().numpy()
    ge = enc.geo_encoder(torch.tensor(co).to(DEV)).float().cpu().numpy()
    assert np.linalg.norm(te - G["time_emb"]) <= 2e-2 * np.linalg.norm(G["time_emb"])
    assert np.linalg.norm(ge - G["geo_emb"]) <= 2e-2 * np.linalg.norm(G["geo_emb"])
    users = dc.user_events()
    L = 6
    x, mask = enc.encode_user_sequences(users, L)
    assert tuple(x.shape) == (2, L * dc.QI, H) and tuple(mask.shape) == (2, L * dc.QI)
    assert mask.sum(1).tolist() == [5 * dc.QI, 2 * dc.QI]
    want = G["encoded_user_sequence"]                                      # user 0: 5 events
    got = x[0, :5 * dc.QI].float().cpu().numpy()
    assert np.linalg.norm(got - want) <= 2e-2 * np.linalg.norm(want)
    assert (x[0, 5 * dc.QI:].float().abs().max().item() == 0) and (x[1, 2 * dc.QI:].float().abs().max().item() == 0)

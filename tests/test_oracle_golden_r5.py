"""CPU: the training-mode oracle (oracle/qformer_train_ref.py, masks fed in) reproduces the vectors the REFERENCE produced in train()
mode with the same masks applied by its own nn.Dropout modules (tests/golden/make_golden_r5.py), the numpy mask generator
(oracle/dropout_ref.py) has the statistics of a Bernoulli(1 - p) stream, and with no masks the training-mode functions ARE the pinned
eval-mode oracle."""
import numpy as np
import pytest
import torch

from oracle import dropout_ref as DR
from oracle import qformer_ref as R
from oracle import qformer_train_ref as RT
from tests.golden import cases
from tests.test_oracle_golden import _check_grads, _close, _load, _params


def test_item_qformer_training_mode(golden_dir):
    case = cases.TRAIN["item_c1_train"]
    c, p = case["cfg"], case["p"]
    g = _load(golden_dir, "item_c1_train")
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    P = _params(R.item_qformer_shapes(cfg, c["F"]), case["seed"])
    x, mask = cases.item_inputs(case)
    xt, mt = torch.from_numpy(x), torch.from_numpy(mask)
    out = RT.item_qformer_forward_train(P, cfg, xt, mt, cases.train_masks(case), p)
    for k in ("query_outputs", "item_representation", "reconstructed_fields"):
        _close(out[k].detach().numpy(), g[k], what=k)
    pos, neg = cases.triplet_reps(case)
    loss, rl, cl = R.qformer_loss(out, xt, mt, torch.from_numpy(pos), torch.from_numpy(neg))
    _close(loss.detach(), g["loss"], what="loss")
    loss.backward()
    _check_grads(P, g, cases.item_grad_keys(c))
    # the masks matter: the eval-mode forward is far from the training-mode fixture
    ev = R.item_qformer_forward(P, cfg, xt, mt)["query_outputs"].detach().numpy()
    assert np.linalg.norm(ev - g["query_outputs"]) > 0.1 * np.linalg.norm(g["query_outputs"])


def test_user_qformer_training_mode(golden_dir):
    case = cases.TRAIN["user_t96_train"]
    c, p = case["cfg"], case["p"]
    g = _load(golden_dir, "user_t96_train")
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 1)
    P = _params(R.user_qformer_shapes(cfg, c["n_pred"]), case["seed"])
    x, mask, tgt = cases.user_inputs(case)
    pred, _ = RT.user_qformer_forward_train(P, cfg, torch.from_numpy(x), torch.from_numpy(mask), c["n_pred"], cases.train_masks(case), p)
    _close(pred.detach().numpy(), g["predicted_item_tokens"], what="predicted_item_tokens")
    loss = ((pred - torch.from_numpy(tgt)) ** 2).mean()
    _close(loss.detach(), g["loss"], what="loss")
    loss.backward()
    _check_grads(P, g, cases.user_grad_keys(c))


@pytest.mark.parametrize("name", ["item_c1", "user_t96"])
def test_without_masks_the_training_oracle_is_the_eval_oracle(name):
    case = cases.ALL[name]
    c = case["cfg"]
    if case["kind"] == "item":
        cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
        P = _params(R.item_qformer_shapes(cfg, c["F"]), case["seed"], requires_grad=False)
        x, mask = cases.item_inputs(case)
        a = R.item_qformer_forward(P, cfg, torch.from_numpy(x), torch.from_numpy(mask))["query_outputs"]
        b = RT.item_qformer_forward_train(P, cfg, torch.from_numpy(x), torch.from_numpy(mask), None, 0.0)["query_outputs"]
    else:
        cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 1)
        P = _params(R.user_qformer_shapes(cfg, c["n_pred"]), case["seed"], requires_grad=False)
        x, mask, _ = cases.user_inputs(case)
        a = R.user_qformer_forward(P, cfg, torch.from_numpy(x), torch.from_numpy(mask), c["n_pred"])[0]
        b = RT.user_qformer_forward_train(P, cfg, torch.from_numpy(x), torch.from_numpy(mask), c["n_pred"], None, 0.0)[0]
    assert torch.equal(a, b)


@pytest.mark.parametrize("p", [0.1, 0.2, 0.5])
def test_mask_generator_statistics(p):
    """Bernoulli(1 - p) per element, unrelated across seeds and across the two counter layouts' neighbouring ranges."""
    n = 1 << 18
    k0 = DR.keep_range(12345, p, 0, n).astype(np.float64)
    k1 = DR.keep_range(12346, p, 0, n).astype(np.float64)
    sd = (p * (1 - p) / n) ** 0.5
    assert abs(k0.mean() - (1 - p)) < 5 * sd and abs(k1.mean() - (1 - p)) < 5 * sd
    cov = ((k0 - k0.mean()) * (k1 - k1.mean())).mean()
    assert abs(cov) < 5 * p * (1 - p) / n ** 0.5
    lag = ((k0[1:] - k0.mean()) * (k0[:-1] - k0.mean())).mean()
    assert abs(lag) < 5 * p * (1 - p) / n ** 0.5
    # counters beyond 2^32 move the stream (the high word enters the key)
    assert not np.array_equal(DR.keep_range(7, p, 0, 4096), DR.keep_range(7, p, 1 << 32, 4096))
    assert np.array_equal(DR.keep_range(7, p, 100, 64), DR.keep_range(7, p, 0, 164)[100:])
    assert DR.site_seed(0x5EED, 1, 0, 1) != DR.site_seed(0x5EED, 1, 0, 2) != DR.site_seed(0x5EED, 2, 0, 1)


@pytest.mark.parametrize("p", [0.1, 0.2, 0.5])
def test_attention_pair_word_statistics(p):
    """The attention sites draw one 32-bit word per PAIR of keys of a row: the two 16-bit fields of a word, neighbouring words,
    neighbouring rows, diagonal neighbours and two seeds must be uncorrelated, every column / row mean within its binomial range."""
    nrows, Sk = 4096, 1600
    K = DR.attn_keep_rows(0x5EED, p, 0, nrows, Sk).astype(np.float64)
    n = K.size
    assert abs(K.mean() - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5
    c = K - K.mean()
    lim = 5 * p * (1 - p) / n ** 0.5
    cov = lambda a, b: float((a * b).mean())
    assert abs(cov(c[:, 0::2], c[:, 1::2])) < lim            # the two fields of one word
    assert abs(cov(c[:, 1:-1:2], c[:, 2::2])) < lim          # neighbouring words
    assert abs(cov(c[:-1], c[1:])) < lim and abs(cov(c[:-2], c[2:])) < lim      # rows
    assert abs(cov(c[:-1, :-1], c[1:, 1:])) < lim
    K2 = DR.attn_keep_rows(0x5EEE, p, 0, nrows, Sk).astype(np.float64)
    assert abs(cov(c, K2 - K2.mean())) < lim
    assert np.abs(K.mean(0) - (1 - p)).max() < 6 * (p * (1 - p) / nrows) ** 0.5
    assert np.abs(K.mean(1) - (1 - p)).max() < 6 * (p * (1 - p) / Sk) ** 0.5
    # rows beyond 2^31 and an odd key count
    a = DR.attn_keep_rows(7, p, (1 << 33) + 5, 8, 13)
    assert a.shape == (8, 13) and np.array_equal(a[3], DR.attn_keep_rows(7, p, (1 << 33) + 8, 1, 13)[0])
    assert np.array_equal(DR.attn_keep(9, p, 2, 3, 4, 6, b0=5).reshape(-1, 6), DR.attn_keep_rows(9, p, 5 * 3 * 4, 2 * 3 * 4, 6))

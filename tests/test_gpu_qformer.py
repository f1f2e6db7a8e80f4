"""GPU parity: unirec_amd Q-Former classes (HIP path) vs the golden vectors the reference produced and
vs the oracle, on the same seeded inputs -- forward outputs, losses and parameter gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import qformer_ref as R  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.parity_utils import GRAD_REL, OUT_REL, assert_close, grad_scale, load_generated, load_golden  # noqa: E402

DEV = "cuda"
ITEM = [n for n, c in cases.ALL.items() if c["kind"] == "item"]
USER = [n for n, c in cases.ALL.items() if c["kind"] == "user"]


def _grad_np(p):
    assert p.grad is not None
    return p.grad.detach().float().cpu().numpy()


@pytest.mark.parametrize("name", ITEM)
def test_item_qformer_matches_reference(name):
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    case = cases.ALL[name]
    c = case["cfg"]
    g = load_golden(name)
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    m = QFormerForItemRepresentation(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"],
                                     intermediate_size=c["I"], num_query_tokens=c["Q"], field_embedding_dim=c["E"],
                                     num_fields=c["F"], dropout=0.0)
    m = load_generated(m, R.item_qformer_shapes(cfg, c["F"]), case["seed"])
    m.train()
    x, mask = cases.item_inputs(case)
    xt, mt = torch.from_numpy(x).to(DEV), torch.from_numpy(mask).to(DEV)
    out = m(xt, mt)
    print(name)
    for k in ("query_outputs", "item_representation", "reconstructed_fields"):
        assert out[k].dtype == torch.float32
        assert_close(out[k], g[k], OUT_REL, k)
    # the reference's own loss formula on the caller side (training/item_qformer_training.py:49-56)
    pos, neg = cases.triplet_reps(case)
    oc = {k: v for k, v in out.items()}
    loss, rl, cl = R.qformer_loss(oc, xt, mt, torch.from_numpy(pos).to(DEV), torch.from_numpy(neg).to(DEV))
    assert_close(loss, g["loss"], OUT_REL, "loss")
    loss.backward()
    named = dict(m.named_parameters())
    gs = grad_scale(g, cases.item_grad_keys(c))
    for k in cases.item_grad_keys(c):
        want = g["grad/" + k]
        got = cases.trim_like(_grad_np(named[k]))
        assert_close(got, want, GRAD_REL, "grad/" + k, floor=1e-6, ref_scale=gs)
    # dead reference tensors never receive gradients
    assert named["qformer.embeddings.word_embeddings.weight"].grad is None
    assert named["qformer.encoder.layer.0.intermediate.dense.weight"].grad is None


@pytest.mark.parametrize("name", ITEM)
def test_item_qformer_hip_losses_match_reference(name):
    """QFormerLoss + eval metrics computed by the HIP loss kernels (bench / training fast path)."""
    from unirec_amd import hip
    case = cases.ALL[name]
    g = load_golden(name)
    x, mask = cases.item_inputs(case)
    rec = torch.from_numpy(g["reconstructed_fields"]).to(DEV)
    xt, mt = torch.from_numpy(x).to(DEV), torch.from_numpy(mask).to(DEV).float()
    sums = hip.recon_stats(rec, xt, mt).cpu().numpy()
    assert_close(sums[0] / sums[1], g["eval_mse"], 1e-4, "eval_mse")
    assert_close(sums[2], g["eval_cos_sum"], 1e-4, "eval_cos_sum")
    assert sums[1] == mask.sum()
    pos, neg = cases.triplet_reps(case)
    a = torch.from_numpy(g["item_representation"]).to(DEV)
    tl, _ = hip.triplet_margin(a, torch.from_numpy(pos).to(DEV), torch.from_numpy(neg).to(DEV), 0.5, 0.5)
    assert_close(tl.cpu().numpy()[0], g["cont_loss"], 1e-4, "cont_loss")
    assert_close(sums[0] / sums[1] + 0.5 * tl.cpu().numpy()[0], g["loss"], 1e-4, "loss")


@pytest.mark.parametrize("name", USER)
def test_user_qformer_matches_reference(name):
    from unirec_amd.user_qformer import UserQFormer
    case = cases.ALL[name]
    c = case["cfg"]
    g = load_golden(name)
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 1)
    m = UserQFormer(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                    num_query_tokens=c["Q"], input_embedding_dim=c["E"], num_item_tokens_to_predict=c["n_pred"], dropout=0.0)
    m = load_generated(m, R.user_qformer_shapes(cfg, c["n_pred"]), case["seed"])
    m.train()
    x, mask, tgt = cases.user_inputs(case)
    pred = m(torch.from_numpy(x).to(DEV), torch.from_numpy(mask).to(DEV))
    print(name)
    assert_close(pred, g["predicted_item_tokens"], OUT_REL, "predicted_item_tokens")
    loss = ((pred - torch.from_numpy(tgt).to(DEV)) ** 2).mean()
    assert_close(loss, g["loss"], OUT_REL, "loss")
    loss.backward()
    named = dict(m.named_parameters())
    gs = grad_scale(g, cases.user_grad_keys(c))
    for k in cases.user_grad_keys(c):
        assert_close(cases.trim_like(_grad_np(named[k])), g["grad/" + k], GRAD_REL, "grad/" + k, floor=1e-6, ref_scale=gs)


def test_fully_masked_item_is_finite_and_uniform():
    """SURVEY I3 on the HIP path: fully masked rows are finite and depend on the masked content."""
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    case = cases.ALL["item_c1"]
    c = case["cfg"]
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    m = QFormerForItemRepresentation(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"],
                                     intermediate_size=c["I"], num_query_tokens=c["Q"], field_embedding_dim=c["E"],
                                     num_fields=c["F"], dropout=0.0)
    m = load_generated(m, R.item_qformer_shapes(cfg, c["F"]), case["seed"]).eval()
    x, mask = cases.item_inputs(case)
    xt, mt = torch.from_numpy(x).to(DEV), torch.from_numpy(mask).to(DEV)
    with torch.no_grad():
        base = m(xt, mt)["query_outputs"]
        x2 = xt.clone()
        x2[2] = x2[2] * 0.5 + 0.1
        alt = m(x2, mt)["query_outputs"]
    assert torch.isfinite(base).all()
    assert not torch.allclose(alt[2], base[2], atol=1e-3)
    assert torch.equal(alt[0], base[0])


def test_dropout_training_mode_is_reproducible_per_step():
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    torch.manual_seed(0)
    m = QFormerForItemRepresentation(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                     num_query_tokens=4, field_embedding_dim=64, num_fields=6, dropout=0.2).to(DEV).train()
    x = torch.randn(8, 6, 64, device=DEV)
    m.qformer._step = 0
    a = m(x)["query_outputs"]
    b = m(x)["query_outputs"]           # next step -> different masks
    m.qformer._step = 0
    c = m(x)["query_outputs"]
    assert torch.equal(a, c) and not torch.equal(a, b)
    a.sum().backward()
    assert torch.isfinite(m.query_embeddings.grad).all()


def test_gradient_checkpointing_is_bit_identical():
    """/root/reference/models/qformer.py:525-548: `config.gradient_checkpointing` in training re-runs each layer's forward inside the
    backward.  Here the re-run uses the same kernels and the dropout seeds of the recorded step: outputs and every parameter
    gradient equal the plain run bit for bit (dropout on, cross-attention in every second layer, key mask), and the switch is
    inert in eval mode / without grad."""
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    torch.manual_seed(0)
    m = QFormerForItemRepresentation(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256,
                                     num_query_tokens=8, field_embedding_dim=64, num_fields=6, dropout=0.2).to(DEV).train()
    x = torch.randn(16, 6, 64, device=DEV)
    mask = (torch.rand(16, 6, device=DEV) < 0.7).long()
    mask[:, 0] = 1
    w = torch.randn(16, 8, 128, device=DEV)

    def run(ckpt):
        m.qformer.config.gradient_checkpointing = ckpt
        m.qformer._step = 0
        m.zero_grad(set_to_none=True)
        out = m(x, mask)["query_outputs"]
        (out.float() * w).sum().backward()
        return out.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    o0, g0 = run(False)
    o1, g1 = run(True)
    assert torch.equal(o0, o1)
    assert g0.keys() == g1.keys() and len(g0) > 40
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    m.qformer._step = 0
    with torch.no_grad():
        assert torch.equal(m(x, mask)["query_outputs"], o0)
    m.qformer.config.gradient_checkpointing = False


def test_forward_triplet_equals_the_separate_forwards():
    """One forward over anchor | positives | negatives with the backward on the anchor rows (QFormerForItemRepresentation.forward_triplet)
    against the reference's schedule (training/item_qformer_training.py:117-131: anchor with grad, the others without): without dropout
    every kernel is row-independent, so outputs and ALL parameter gradients are equal bit for bit; with dropout the merged step is
    reproducible and finite; gradient checkpointing composes with it."""
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    from unirec_amd.losses import QFormerLoss
    torch.manual_seed(0)
    B = 24
    m = QFormerForItemRepresentation(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256,
                                     num_query_tokens=8, field_embedding_dim=64, num_fields=6, dropout=0.0).to(DEV).train()

    def fields(n):
        x = torch.randn(n, 6, 64, device=DEV)
        mk = (torch.rand(n, 6, device=DEV) < 0.7).long()
        mk[:, 0] = 1
        return x * mk[..., None], mk
    (xa, ma), (xo, mo) = fields(B), fields(2 * B)
    loss_fn = QFormerLoss()

    def separate():
        m.zero_grad(set_to_none=True)
        out = m(xa, ma)
        with torch.no_grad():
            rep = m(xo, mo)["item_representation"]
        loss, _, _ = loss_fn(out, {"field_embeddings": xa}, rep[:B], rep[B:], ma)
        loss.backward()
        return out, rep, loss.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

    def merged():
        m.zero_grad(set_to_none=True)
        out, rep = m.forward_triplet(xa, ma, xo, mo)
        loss, _, _ = loss_fn(out, {"field_embeddings": xa}, rep[:B], rep[B:], ma)
        loss.backward()
        return out, rep, loss.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    o0, r0, l0, g0 = separate()
    o1, r1, l1, g1 = merged()
    for k in ("query_outputs", "item_representation", "reconstructed_fields"):
        assert torch.equal(o0[k], o1[k]), k
    assert torch.equal(r0, r1) and torch.equal(l0, l1)
    assert g0.keys() == g1.keys() and len(g0) > 40
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    m.qformer.config.gradient_checkpointing = True
    _, _, l2, g2 = merged()
    m.qformer.config.gradient_checkpointing = False
    assert torch.equal(l2, l0) and all(torch.equal(g0[n], g2[n]) for n in g0)
    # dropout on: reproducible per step, finite
    m2 = QFormerForItemRepresentation(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                      num_query_tokens=8, field_embedding_dim=64, num_fields=6, dropout=0.2).to(DEV).train()
    outs = []
    for _ in range(2):
        m2.qformer._step = 0
        m2.zero_grad(set_to_none=True)
        out, rep = m2.forward_triplet(xa, ma, xo, mo)
        loss, _, _ = loss_fn(out, {"field_embeddings": xa}, rep[:B], rep[B:], ma)
        loss.backward()
        outs.append((loss.detach().clone(), m2.query_embeddings.grad.detach().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.isfinite(outs[0][1]).all()

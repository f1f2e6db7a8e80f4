"""GPU, round 6: the HEADLINE configuration's exact architecture against a vector the reference itself produced.

`joint_c4` (tests/golden/make_golden_r6.py): the reference's 12-layer H 1024 Q 2 F 14 item Q-Former inside the reference's
MultiModalQwenEmbedding.forward over the installed Qwen3Model at the 0.6B shape (28 layers, 16 / 8 heads of 128, I 3072) with merged LoRA
r 16, B 2 x S 2048, hist 50, left padding, InfoNCE over a pool of 1000, MRR rank (training/train_item_individual_token_joint.py:134-181,
331-352,392-419).  The product runs the same two sequences through the launches the bench step runs: 16 key blocks per head through the
generated causal kernels, 100 injected rows per sequence, all-S mean pool, pool-1000 InfoNCE, gradients back through 28 + 12 layers.
Measured errors are printed: they are the margin of the composite chain where it is longest."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import qformer_ref as R  # noqa: E402
from oracle import qwen3_ref as Q  # noqa: E402
from oracle import weights as W  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.parity_utils import GRAD_REL, OUT_REL, assert_close, load_generated, load_golden, rel_err  # noqa: E402

DEV = "cuda"


def build_joint_c4(case):
    from tests.test_gpu_joint import _qwen_cfg
    from unirec_amd.joint import MultiModalQwenEmbedding
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    c = case["cfg"]
    qc = cases.qwen_cfg(case)
    qc.lora_r, qc.lora_alpha = case["lora_r"], case["lora_alpha"]
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    qf = QFormerForItemRepresentation(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                                      num_query_tokens=c["Q"], field_embedding_dim=c["E"], num_fields=c["F"], dropout=0.0)
    qf = load_generated(qf, R.item_qformer_shapes(cfg, c["F"]), case["seed"])
    hc = _qwen_cfg(qc, True)
    hc.vocab_size = case["first_special_id"]
    m = MultiModalQwenEmbedding(qformer_model=qf, use_lora=True, qwen_config=hc, num_history_items=case["hist"], num_query_tokens_per_item=c["Q"])
    assert m.first_special_id == case["first_special_id"]
    gen = W.fill_state_dict(Q.qwen3_shapes(qc, lora=True), case["seed"] + 1, rules=cases.lora_weight_rules(case))
    missing, unexpected = m.base_model.load_state_dict({k: torch.from_numpy(v) for k, v in gen.items()}, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return m.to(DEV).train(), qf


def test_joint_c4_exact_architecture_matches_the_reference():
    from unirec_amd.joint import InfoNCELoss, mrr_ranks
    case = cases.C4["joint_c4"]
    g = load_golden("joint_c4")
    m, qf = build_joint_c4(case)
    assert m.base_model._drop_p() == 0.0
    ids, am, hfe, ham, pos, neg, nmask = cases.joint_inputs(case)
    assert ((am == 0).sum(axis=1) == g["n_pad"]).all()
    t = lambda a: torch.from_numpy(a).to(DEV)
    user = m(t(ids), t(am), t(hfe), t(ham))
    print("joint_c4 (12-layer Q-Former x 50 items -> 28 layers + LoRA at S 2048 -> pool 1000)")
    assert_close(user, g["user_embeddings"], OUT_REL, "user_embeddings")
    loss = InfoNCELoss()(user, t(pos), t(neg), t(nmask))
    assert_close(loss, g["loss"], OUT_REL, "infonce loss")
    # MRR rank: integer work, exact on the product's own scores; against the reference the rank may move only across candidates whose
    # reference score lies within the measured score error of the positive's (1000 random candidates sit ~2e-4 apart in cosine)
    scores, rank = mrr_ranks(user, t(pos), t(neg))
    sc = scores.float().cpu().numpy().astype(np.float64)
    un = g["user_embeddings"].astype(np.float64)
    un /= np.linalg.norm(un, axis=-1, keepdims=True)
    for b in range(case["B"]):
        cand = np.concatenate([pos[b][None], neg[b]], 0).astype(np.float64)
        cand /= np.maximum(np.linalg.norm(cand, axis=-1, keepdims=True), 1e-12)
        s_ref = cand @ un[b]
        d = float(np.abs(sc[b] - s_ref).max())
        close = int((np.abs(s_ref[1:] - s_ref[0]) <= 2 * d).sum())
        own = 1 + int((sc[b][1:] > sc[b][0]).sum())
        got, want = int(rank[b]), int(g["ranks"][b])
        print(f"  rank[{b}]: product {got}, reference {want}; max |score error| {d:.2e}, {close} candidates within 2x of it of the positive")
        assert got == own, "the rank kernel must be exact on its own scores"
        assert abs(got - want) <= close, (got, want, close)
    loss.backward()
    torch.cuda.synchronize()
    qn = dict(qf.named_parameters())
    joint_tol = GRAD_REL * 1.5           # 6e-2 through the joint chain (tests/parity_utils.py)
    assert_close(qn["query_embeddings"].grad, g["grad/query_embeddings"], joint_tol, "grad/query_embeddings")
    for k in cases.C4_QF_KEYS:
        got = qn[k].grad.float().cpu().numpy()
        assert_close(cases.c4_rows(got), g["grad/" + k], joint_tol, "grad/" + k)
        gn, ref = float(np.linalg.norm(got.astype(np.float64))), float(g["gnorm/" + k])
        assert abs(gn - ref) <= joint_tol * ref, (k, gn, ref)
    named = dict(m.base_model.named_parameters())
    worst = 0.0
    for i in range(28):
        for pj in cases.LORA_PROJ:
            for ab in ("lora_A", "lora_B"):
                k = f"layers.{i}.{pj}.{ab}.weight"
                gk = named[k].grad.float().cpu().numpy().astype(np.float64)
                ref = float(g["gnorm/" + k])
                e = abs(float(np.linalg.norm(gk)) - ref) / ref
                worst = max(worst, e)
                assert e <= joint_tol, (k, float(np.linalg.norm(gk)), ref)
                if "grad/" + k in g:
                    assert_close(gk, g["grad/" + k], joint_tol, "grad/" + k)
    print(f"  worst relative error of an adapter gradient's NORM over 28 x 7 x 2 adapters: {worst:.3e}")
    assert named["layers.0.self_attn.q_proj.weight"].grad is None      # base weights stay frozen


@pytest.mark.parametrize("kind", ["item", "user"])
def test_weight_gradients_on_the_side_stream_are_bit_identical(kind, monkeypatch):
    """Round 6: the Q-Formers' token reductions dW = dY^T X and bias column sums run on a side stream beside the dX chain
    (unirec_amd/qformer.py:_DW_SIDE).  Same kernels, same arithmetic: every gradient equals the single-stream run bit for bit, also when the
    step is repeated (the second step reuses memory the first one's side-stream work read)."""
    import unirec_amd.qformer as qformer
    from unirec_amd.losses import mse_loss

    def run(side):
        monkeypatch.setattr(qformer, "_DW_SIDE", side)
        torch.manual_seed(7)
        if kind == "item":
            from unirec_amd.qformer_utils import QFormerForItemRepresentation
            m = QFormerForItemRepresentation(hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=1024, num_query_tokens=32,
                                             field_embedding_dim=256, num_fields=14, dropout=0.2).to(DEV).train()
            g = torch.Generator().manual_seed(3)
            x = torch.randn(192, 14, 256, generator=g).to(DEV)
            mk = (torch.rand(192, 14, generator=g) < 0.8).long(); mk[:, 0] = 1
            mk = mk.to(DEV)
            fwd = lambda: m(x, mk)["query_outputs"].float().pow(2).mean() + m(x, mk)["reconstructed_fields"].float().pow(2).mean()
        else:
            from unirec_amd.user_qformer import UserQFormer
            m = UserQFormer(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, num_query_tokens=64, input_embedding_dim=256,
                            num_item_tokens_to_predict=4, dropout=0.1).to(DEV).train()
            g = torch.Generator().manual_seed(4)
            x = torch.randn(24, 320, 256, generator=g).to(DEV)
            mask = (torch.arange(320)[None, :] < torch.randint(160, 321, (24,), generator=g)[:, None]).float().to(DEV)
            tgt = torch.randn(24, 4, 256, generator=g).to(DEV)
            fwd = lambda: mse_loss(m(x, mask), tgt)
        outs = []
        for _ in range(2):
            m.zero_grad(set_to_none=True)
            fwd().backward()
            torch.cuda.synchronize()
            outs.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        return outs
    a, b = run(True), run(False)
    for step in range(2):
        assert a[step].keys() == b[step].keys() and len(a[step]) > 20
        for n in a[step]:
            assert torch.equal(a[step][n], b[step][n]), (kind, step, n)

"""GPU, round 5: the parity seams VERDICT round 4 listed.

(i)  the generated causal head_dim-128 backward (attn_bwd_dq_c128 / attn_bwd_dkv_c128) against an independent fp32 reference at the
     sequence lengths the bench runs (S = 2048, 4096: 16-32 key blocks per (batch, kv head), the per-XCD work queue, every ring phase);
(ii) Q-Former TRAINING mode (hidden + attention-probability dropout on) against vectors the reference itself produced with the
     kernels' masks fed to its nn.Dropout modules (tests/golden/*_train.npz, make_golden_r5.py) and against the training-mode oracle;
(iii) the exported keep flags (ur_dropout_keep) against the numpy restatement bit for bit;
(iv) two ur_attn_bwd calls in flight on two streams (each brings its own work-queue words: no library-owned device state);
(v)  the LoRA bit-plane prefetch when the next forward does not match (another M / row0, train -> eval back to back);
(vi) forward_triplet with only one of the two padding masks given.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dropout_ref as DR  # noqa: E402
from oracle import qformer_ref as R  # noqa: E402
from oracle import qformer_train_ref as RT  # noqa: E402
from oracle import weights as W  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.parity_utils import GRAD_REL, OUT_REL, assert_close, grad_scale, load_generated, load_golden  # noqa: E402
from unirec_amd import hip  # noqa: E402

DEV = "cuda"


# ---------------------------------------------------------------------------------------------------------------------------------
# (i) causal GQA attention, forward + backward, at the bench's sequence lengths
# ---------------------------------------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, km):
    """fp32, one (batch, kv head) group at a time (S = 4096 scores are 64 MiB per head).  SDPA semantics (modeling_qwen3.py:185-208
    behind /root/reference/training/train_item_individual_token_joint.py:173-177): a query row without an allowed key gives zero."""
    B, S, nq, hd = q.shape
    nkv = k.shape[2]
    rep = nq // nkv
    ok0 = torch.tril(torch.ones(S, S, dtype=torch.bool, device=q.device))
    out = torch.empty(B, S, nq, hd, dtype=torch.float32, device=q.device)
    for b in range(B):
        ok = ok0 if km is None else ok0 & km[b].bool()[None, :]
        for h in range(nq):
            s = (q[b, :, h] @ k[b, :, h // rep].t()) * hd ** -0.5
            w = torch.softmax(s.masked_fill(~ok, float("-inf")), dim=-1)
            w = torch.where(ok.any(-1, keepdim=True), w, torch.zeros_like(w))
            out[b, :, h] = w @ v[b, :, h // rep]
    return out


def _mask(kind, B, S, seed):
    if kind == "none":
        return None
    km = torch.ones(B, S, dtype=torch.uint8)
    g = torch.Generator().manual_seed(seed)
    if kind == "left":          # left padding (the tokenizer's side): leading keys masked -> fully masked causal rows
        for b in range(B):
            km[b, : (97 + 411 * b) % (S // 2)] = 0
    else:                       # holes anywhere, whole 128-key blocks among them
        km = (torch.rand(B, S, generator=g) < 0.8).to(torch.uint8)
        km[:, 0] = 1
        km[0, 256:512] = 0
        km[B - 1, S - 384:S - 128] = 0
    return km.to(DEV)


@pytest.mark.parametrize("B,S,nq,nkv,mask", [(2, 2048, 4, 2, "none"), (2, 2048, 4, 2, "left"), (2, 2048, 4, 2, "holes"),
                                             (1, 4096, 4, 2, "none"), (2, 4096, 4, 2, "left"), (1, 4096, 4, 2, "holes"),
                                             (1, 2048, 16, 8, "holes"), (1, 4096, 16, 8, "left")])
def test_causal_gqa_backward_at_bench_lengths(B, S, nq, nkv, mask):
    hd = 128
    g = torch.Generator().manual_seed(S + nq)
    buf = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).to(DEV).to(torch.bfloat16)
    q = buf[..., :nq * hd].view(B, S, nq, hd)
    k = buf[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd)
    v = buf[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
    km = _mask(mask, B, S, S + 7)
    dout = torch.randn(B, S, nq, hd, generator=g).to(DEV).to(torch.bfloat16)
    o, ctx = hip.attn_fwd(q, k, v, causal=True, key_mask=km)
    dq, dk, dv = hip.attn_bwd(ctx, dout)
    qf, kf, vf = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qf, kf, vf, km)
    ref.backward(dout.float())
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    assert torch.allclose(o.float(), ref.detach(), rtol=2e-2, atol=2e-2), f"o max err {(o.float() - ref).abs().max().item()}"
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        err = (got.float() - want).abs().max().item()
        scale = want.abs().max().item()
        assert err <= 2e-2 * scale + 2e-2, f"{name}: max err {err} (scale {scale})"
        rel = ((got.float() - want).norm() / want.norm()).item()
        assert rel <= 1e-2, f"{name}: relative Frobenius error {rel}"


# ---------------------------------------------------------------------------------------------------------------------------------
# (iii) exported keep flags
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,p,idx0,n", [(0x5EED, 0.2, 0, 100003), (0x7FFFFFFFFFFFFFF1, 0.1, (1 << 32) - 777, 5000), (3, 0.5, 12345678901234, 4096),
                                           (0xC0FFEE, 0.0, 0, 513)])
def test_dropout_keep_export_matches_the_numpy_generator(seed, p, idx0, n):
    got = hip.dropout_keep(seed, p, idx0, n, DEV).cpu().numpy()
    assert np.array_equal(got, DR.keep_range(seed, p, idx0, n))


@pytest.mark.parametrize("seed,p,row0,nrows,Sk", [(0x5EED, 0.2, 0, 513, 96), (0x7FFFFFFFFFFFFFF1, 0.1, (1 << 33) - 3, 40, 1600), (3, 0.5, 77, 64, 13),
                                                  (0xC0FFEE, 0.0, 0, 5, 8)])
def test_attention_dropout_keep_export_matches_the_numpy_generator(seed, p, row0, nrows, Sk):
    got = hip.attn_dropout_keep(seed, p, row0, nrows, Sk, DEV).cpu().numpy()
    assert np.array_equal(got, DR.attn_keep_rows(seed, p, row0, nrows, Sk))


# ---------------------------------------------------------------------------------------------------------------------------------
# (ii) training mode against the reference (masks fed) and the training-mode oracle
# ---------------------------------------------------------------------------------------------------------------------------------
def _grad_np(p):
    assert p.grad is not None
    return p.grad.detach().float().cpu().numpy()


def _arm(bert, case):
    """The product draws this forward's masks from (seed, step): set both so that it runs as the fixture's step."""
    bert.seed = case["drop_seed"]
    bert._step = case["step"] - 1
    for layer, site in ((0, 1), (1, 5), (DR.EMB_LAYER, DR.EMB_SITE)):
        assert bert._seed(layer, site, step=case["step"]) == DR.site_seed(case["drop_seed"], case["step"], layer, site)


def test_site_masks_are_the_kernels_masks():
    """One hidden site and one attention site, end to end: the primitive's output under dropout equals its p = 0 output times the
    fed mask / (1 - p) -- i.e. the counter layouts restated in oracle/dropout_ref.py are the kernels'."""
    p, seed = 0.2, DR.site_seed(0x5EED, 1, 0, DR.SITE_SELF_OUT)
    rows, H = 48, 256
    g = torch.Generator().manual_seed(1)
    y = torch.randn(rows, H, generator=g).to(DEV).to(torch.bfloat16)
    gamma, beta = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    _, z, _, _ = hip.layernorm_fwd(y, gamma, beta, 1e-12, p_pre=p, seed_pre=seed, drop_row0=5)      # no residual: z = dropout(y)
    keep = torch.from_numpy(DR.hidden_keep(seed, p, rows, H, row0=5)).to(DEV)
    want = (y.float() * keep * 1.25).to(torch.bfloat16)          # 1 / (1 - 0.2f) rounds to 1.25f
    assert torch.equal(z, want)
    # attention: V = identity-like probe -> o = dropped probabilities; compare against softmax * keep / (1 - p)
    B, nh, Sq, Sk, hd = 2, 2, 64, 64, 64
    q = (torch.randn(B, Sq, nh, hd, generator=g) * 0.3).to(DEV).to(torch.bfloat16)
    k = (torch.randn(B, Sk, nh, hd, generator=g) * 0.3).to(DEV).to(torch.bfloat16)
    v = torch.eye(Sk, hd)[None, :, None, :].expand(B, Sk, nh, hd).contiguous().to(DEV).to(torch.bfloat16)      # v[b, key, h, d] = (key == d)
    seed_a = DR.site_seed(0x5EED, 1, 0, DR.SITE_SELF_PROBS)
    o, _ = hip.attn_fwd(q, k, v, causal=False, dropout_p=p, seed=seed_a, drop_batch0=3)
    probs = torch.softmax((q.float().permute(0, 2, 1, 3) @ k.float().permute(0, 2, 3, 1)) * hd ** -0.5, dim=-1)
    keep_a = torch.from_numpy(DR.attn_keep(seed_a, p, B, nh, Sq, Sk, b0=3)).to(DEV)
    want_o = (probs * keep_a / (1 - p)).permute(0, 2, 1, 3)      # [B, Sq, nh, Sk == hd]
    assert torch.allclose(o.float(), want_o, rtol=2e-2, atol=2e-3)
    dropped = keep_a.permute(0, 2, 1, 3) == 0
    assert float(o.float()[dropped].abs().max()) == 0.0            # exactly the fed positions are zero


def test_item_qformer_training_mode_matches_reference():
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    case = cases.TRAIN["item_c1_train"]
    c, p = case["cfg"], case["p"]
    g = load_golden("item_c1_train")
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    m = QFormerForItemRepresentation(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                                     num_query_tokens=c["Q"], field_embedding_dim=c["E"], num_fields=c["F"], dropout=p)
    m = load_generated(m, R.item_qformer_shapes(cfg, c["F"]), case["seed"])
    m.train()
    _arm(m.qformer, case)
    x, mask = cases.item_inputs(case)
    xt, mt = torch.from_numpy(x).to(DEV), torch.from_numpy(mask).to(DEV)
    out = m(xt, mt)
    assert m.qformer._step == case["step"]
    for k in ("query_outputs", "item_representation", "reconstructed_fields"):
        assert_close(out[k], g[k], OUT_REL, k)
    pos, neg = cases.triplet_reps(case)
    # the PRODUCT's loss (losses.QFormerLoss: HIP masked-MSE + triplet kernels, training/item_qformer_training.py:49-56), not the oracle's, under dropout
    from unirec_amd.losses import QFormerLoss
    loss, rl, cl = QFormerLoss()(out, {"field_embeddings": xt}, torch.from_numpy(pos).to(DEV), torch.from_numpy(neg).to(DEV), mt)
    assert_close(loss, g["loss"], OUT_REL, "loss")
    assert_close(rl, g["recon_loss"], OUT_REL, "recon_loss")
    assert_close(cl, g["cont_loss"], OUT_REL, "cont_loss")
    loss.backward()
    named = dict(m.named_parameters())
    gs = grad_scale(g, cases.item_grad_keys(c))
    for k in cases.item_grad_keys(c):
        assert_close(cases.trim_like(_grad_np(named[k])), g["grad/" + k], GRAD_REL, "grad/" + k, floor=1e-6, ref_scale=gs)
    # and the oracle with the same masks (the full gradients, not the trimmed fixture rows)
    P = {k_: torch.from_numpy(v).requires_grad_(True) for k_, v in W.fill_state_dict(R.item_qformer_shapes(cfg, c["F"]), case["seed"]).items()}
    oo = RT.item_qformer_forward_train(P, cfg, torch.from_numpy(x), torch.from_numpy(mask), cases.train_masks(case), p)
    ol, _, _ = R.qformer_loss(oo, torch.from_numpy(x), torch.from_numpy(mask), torch.from_numpy(pos), torch.from_numpy(neg))
    ol.backward()
    for k in cases.item_grad_keys(c):
        if float(P[k].grad.norm()) > 1e-6:
            assert_close(_grad_np(named[k]), P[k].grad.numpy(), GRAD_REL, "oracle grad/" + k)
    # a different step draws different masks; eval mode ignores them
    out2 = m(xt, mt)["query_outputs"]
    assert not torch.equal(out2, out["query_outputs"])


def test_user_qformer_training_mode_matches_reference():
    from unirec_amd.user_qformer import UserQFormer
    case = cases.TRAIN["user_t96_train"]
    c, p = case["cfg"], case["p"]
    g = load_golden("user_t96_train")
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 1)
    m = UserQFormer(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                    num_query_tokens=c["Q"], input_embedding_dim=c["E"], num_item_tokens_to_predict=c["n_pred"], dropout=p)
    m = load_generated(m, R.user_qformer_shapes(cfg, c["n_pred"]), case["seed"])
    m.train()
    _arm(m.qformer, case)
    x, mask, tgt = cases.user_inputs(case)
    pred = m(torch.from_numpy(x).to(DEV), torch.from_numpy(mask).to(DEV))
    assert_close(pred, g["predicted_item_tokens"], OUT_REL, "predicted_item_tokens")
    from unirec_amd.losses import mse_loss          # the product's loss kernel (training/user_qformer_training.py:193,209)
    loss = mse_loss(pred, torch.from_numpy(tgt).to(DEV))
    assert_close(loss, g["loss"], OUT_REL, "loss")
    loss.backward()
    named = dict(m.named_parameters())
    gs = grad_scale(g, cases.user_grad_keys(c))
    for k in cases.user_grad_keys(c):
        assert_close(cases.trim_like(_grad_np(named[k])), g["grad/" + k], GRAD_REL, "grad/" + k, floor=1e-6, ref_scale=gs)


# ---------------------------------------------------------------------------------------------------------------------------------
# (iv) two backward calls in flight on two streams
# ---------------------------------------------------------------------------------------------------------------------------------
def test_attention_backward_calls_on_two_streams_do_not_share_state():
    """The persistent dK/dV kernel draws its key blocks from work-queue words; they live in each call's own workspace, so two calls
    running concurrently on two streams (many calls apart or not) give exactly the results of the same calls run alone."""
    hd, nq, nkv = 128, 16, 8
    cases_ = []
    for i, (B, S) in enumerate(((8, 1024), (4, 2048))):
        g = torch.Generator().manual_seed(40 + i)
        buf = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).to(DEV).to(torch.bfloat16)
        q, k, v = buf[..., :nq * hd].view(B, S, nq, hd), buf[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd), buf[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
        dout = torch.randn(B, S, nq, hd, generator=g).to(DEV).to(torch.bfloat16)
        o, ctx = hip.attn_fwd(q, k, v, causal=True)
        alone = [t.clone() for t in hip.attn_bwd(ctx, dout)]
        cases_.append((ctx, dout, alone))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for rep in range(6):
        for si, (ctx, dout, _) in enumerate(cases_):
            with torch.cuda.stream(streams[si]):
                outs[si].append(hip.attn_bwd(ctx, dout))
    torch.cuda.synchronize()
    for si, (_, _, alone) in enumerate(cases_):
        for got in outs[si]:
            for a, b in zip(got, alone):
                assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------------------------------
# (v) the LoRA bit-plane prefetch when the next forward does not match
# ---------------------------------------------------------------------------------------------------------------------------------
def _small_decoder():
    from unirec_amd.qwen3 import Qwen3Config, Qwen3LoRAModel
    cfg = Qwen3Config(vocab_size=128, hidden_size=1024, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=16,
                      num_key_value_heads=8, head_dim=128, lora_r=16, lora_alpha=32.0, lora_dropout=0.1)
    torch.manual_seed(0)
    m = Qwen3LoRAModel(cfg).to(DEV).train()
    m.reset_parameters(lora_b_std=0.05)
    return m


def _step(m, ids, prefetch=True):
    B, S = ids.shape
    if prefetch:
        m.prefetch_lora_bits(B * S, DEV, row0=m.first_sample(B) * S)
    m.zero_grad(set_to_none=True)
    pooled = m.forward_pooled(ids)
    pooled.float().pow(2).sum().backward()
    return pooled.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def test_lora_bit_prefetch_survives_shape_changes_and_eval():
    """qwen3._prefetch_next_step writes the NEXT step's planes on a side stream; when the next forward is a different M (last
    partial batch), another first row, or an eval pass, the planes are dropped -- after the main stream has been ordered behind the
    generator's last event.  Every step must equal the same step computed without any prefetch (planes made inline)."""
    g = torch.Generator().manual_seed(3)
    ids_a = torch.randint(0, 128, (4, 512), generator=g).to(DEV)
    ids_b = torch.randint(0, 128, (2, 512), generator=g).to(DEV)           # the epoch's last, smaller batch

    def run(prefetch):
        m = _small_decoder()
        m.lora_seed = 99
        res = []
        res.append(_step(m, ids_a, prefetch))
        res.append(_step(m, ids_a, prefetch))        # consumes the planes made under step 0's backward
        res.append(_step(m, ids_b, prefetch))        # another M: the prefetched set is released
        m.eval()
        with torch.no_grad():
            ev = m.forward_pooled(ids_a).clone()     # train -> eval back to back: released again, no dropout
        m.train()
        res.append(_step(m, ids_a, prefetch))
        m.sample_offset = 7                          # micro-batching: nothing is generated for a "next step"
        res.append(_step(m, ids_b, prefetch))
        assert m._bits_pre is None
        torch.cuda.synchronize()
        return res, ev
    r1, e1 = run(True)
    r0, e0 = run(False)
    assert torch.equal(e0, e1)
    for (p1, g1), (p0, g0) in zip(r1, r0):
        assert torch.equal(p1, p0)
        assert g1.keys() == g0.keys() and len(g1) > 0
        for n in g1:
            assert torch.equal(g1[n], g0[n]), n


# ---------------------------------------------------------------------------------------------------------------------------------
# (vi) forward_triplet with one padding mask missing
# ---------------------------------------------------------------------------------------------------------------------------------
def test_forward_triplet_with_only_one_mask():
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    torch.manual_seed(0)
    B = 8
    m = QFormerForItemRepresentation(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                                     num_query_tokens=8, field_embedding_dim=64, num_fields=6, dropout=0.0).to(DEV).train()
    xa, xo = torch.randn(B, 6, 64, device=DEV), torch.randn(2 * B, 6, 64, device=DEV)
    mo = (torch.rand(2 * B, 6, device=DEV) < 0.6).long()
    mo[:, 0] = 1
    ma = (torch.rand(B, 6, device=DEV) < 0.6).long()
    ma[:, 0] = 1
    for am, om in ((None, mo), (ma, None)):
        out, rep = m.forward_triplet(xa, am, xo, om)
        with torch.no_grad():
            want_rep = m(xo, om)["item_representation"]
            want_a = m(xa, am)["item_representation"]
        assert torch.equal(rep, want_rep)
        assert torch.equal(out["item_representation"].detach(), want_a)


# ---------------------------------------------------------------------------------------------------------------------------------
# (vii) right-padded keys (ragged histories): the masked tail is not swept
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Sq,Sk,p_drop", [(64, 1600, 0.0), (64, 1600, 0.1), (32, 200, 0.2), (64, 320, 0.0)])
def test_right_padded_key_tail_is_skipped_without_changing_a_bit(Sq, Sk, p_drop, monkeypatch):
    """Additive-mask semantics (models/qformer.py:241-244): a masked key has probability exactly 0 once its row has one allowed key, so
    the forward / dQ sweeps end at the last key tile holding a valid key and the few-query dK/dV kernel writes zeros for key blocks
    without one.  Against the fp32 reference (no dropout), against the SAME call with the padded keys' mask bytes set but their K / V
    rows replaced by garbage (nothing of a trimmed tile may be read into the result), and -- a sample WITHOUT any valid key is the
    uniform softmax over all keys -- against the reference for that sample too."""
    from tests.test_gpu_attention import _ref, _randn
    B, nh, hd = 4, 2, 64
    q, k, v = _randn((B, Sq, nh, hd), 1, 0.5), _randn((B, Sk, nh, hd), 2, 0.5), _randn((B, Sk, nh, hd), 3)
    lens = [Sk, Sk // 2 + 3, max(1, Sk // 5), 0]                  # full, two ragged, one sample without a valid key
    km = torch.zeros(B, Sk, dtype=torch.uint8)
    for b, n in enumerate(lens):
        km[b, :n] = 1
    km = km.to(DEV)
    dout = _randn((B, Sq, nh, hd), 4)

    def run(kk, vv):
        o, ctx = hip.attn_fwd(q, kk, vv, causal=False, key_mask=km, dropout_p=p_drop, seed=21, drop_batch0=2)
        dq, dk, dv = hip.attn_bwd(ctx, dout)
        torch.cuda.synchronize()
        return o, dq, dk, dv
    o, dq, dk, dv = run(k, v)
    if p_drop == 0.0:
        qf, kf, vf = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
        ref = _ref(qf, kf, vf, km, False)
        ref.backward(dout.float())
        assert torch.allclose(o.float(), ref.detach(), rtol=2e-2, atol=2e-2)
        for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
            err, scale = (got.float() - want).abs().max().item(), want.abs().max().item()
            assert err <= 2e-2 * scale + 2e-2, f"{name}: max err {err} (scale {scale})"
    # padded keys of samples that have a valid key: gradients exactly zero
    for b, n in enumerate(lens):
        if 0 < n < Sk:
            assert float(dk[b, n:].float().abs().max()) == 0.0 and float(dv[b, n:].float().abs().max()) == 0.0
    # garbage in the padded rows of those samples must not reach any output (whole trimmed tiles are never read; partial ones are masked)
    k2, v2 = k.clone(), v.clone()
    for b, n in enumerate(lens):
        if 0 < n < Sk:
            k2[b, n:] = 37.0
            v2[b, n:] = -53.0
    o2, dq2, dk2, dv2 = run(k2, v2)
    sel = [b for b, n in enumerate(lens) if n > 0]
    assert torch.equal(o[sel], o2[sel]) and torch.equal(dq[sel], dq2[sel]) and torch.equal(dk[sel], dk2[sel]) and torch.equal(dv[sel], dv2[sel])
    # the few-query kernel against the generic one, bit for bit (same arithmetic, the generic kernel sweeps everything)
    if Sq <= 64 and Sk >= 256:
        with hip.attn_mode_set(hip.ATTN_MODE_FEWQ, 0):
            o3, dq3, dk3, dv3 = run(k, v)
        assert torch.equal(dk, dk3) and torch.equal(dv, dv3) and torch.equal(dq, dq3)


# ---------------------------------------------------------------------------------------------------------------------------------
# (viii) the K | V bias gradients out of the few-query dK/dV kernel
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,nh,Sq,Sk,p_drop", [(3, 2, 64, 1600, 0.1), (5, 4, 40, 333, 0.0), (70, 16, 64, 256, 0.2)])
def test_few_query_kernel_emits_the_column_sums_of_dk_dv(B, nh, Sq, Sk, p_drop):
    """ur_attn_bwd_args.kv_colsum: [sum over batch and keys of dK | of dV] per (head, feature) -- the bias gradients of the K | V
    projections (models/qformer.py:186-188) -- from the kernel's f32 accumulators, against the column sums of the bf16 dK / dV it
    stores (ragged masks incl. a sample without a valid key, dropout, key counts that are not multiples of 32); dq / dk / dv themselves
    do not change by a bit, and shapes the few-query kernel does not take report so."""
    from tests.test_gpu_attention import _randn
    q, k, v = _randn((B, Sq, nh, 64), 1, 0.5), _randn((B, Sk, nh, 64), 2, 0.5), _randn((B, Sk, nh, 64), 3)
    g = torch.Generator().manual_seed(Sk)
    lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
    km = (torch.arange(Sk)[None, :] < lens[:, None]).to(torch.uint8)
    km[1] = 0
    km = km.to(DEV)
    dout = _randn((B, Sq, nh, 64), 4)
    o, ctx = hip.attn_fwd(q, k, v, causal=False, key_mask=km, dropout_p=p_drop, seed=5)
    assert hip.attn_bwd_kv_colsum_supported(ctx)
    dq0, dk0, dv0 = hip.attn_bwd(ctx, dout)
    cs = torch.full((2 * nh * 64,), float("nan"), dtype=torch.float32, device=DEV)
    dq1, dk1, dv1 = hip.attn_bwd(ctx, dout, kv_colsum=cs)
    torch.cuda.synchronize()
    assert torch.equal(dq0, dq1) and torch.equal(dk0, dk1) and torch.equal(dv0, dv1)
    want = torch.cat([dk1.float().sum(dim=(0, 1)).reshape(-1), dv1.float().sum(dim=(0, 1)).reshape(-1)])
    scale = want.abs().max().item()
    err = (cs - want).abs().max().item()
    # the stored gradients are bf16-rounded per element (2^-9 relative each, random sign): the sums agree to a few 1e-3 of the largest
    assert torch.isfinite(cs).all() and err <= 4e-3 * scale + 1e-4, (err, scale)
    # a shape that takes another dK/dV kernel says so
    o2, ctx2 = hip.attn_fwd(q, k[:, :64].contiguous(), v[:, :64].contiguous(), causal=False)
    assert not hip.attn_bwd_kv_colsum_supported(ctx2)
    with pytest.raises(ValueError):
        hip.attn_bwd(ctx2, dout, kv_colsum=cs)

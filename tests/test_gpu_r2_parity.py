"""GPU parity, round 2: mid-size reference fixtures (the 256x256 8-phase GEMM, multi-tile causal head_dim-128 attention
and the few-query dK/dV kernel meet reference-produced vectors), the gradients w.r.t. the decoder's input embeddings
(golden `grad_inputs_embeds`), the real UserSequenceEncoder boundary, and full-size C2 / C3 / C5 steps through
size-independent properties."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import data_ref as D  # noqa: E402
from oracle import qformer_ref as R  # noqa: E402
from oracle import qwen3_ref as Q  # noqa: E402
from oracle import weights as W  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.golden import data_cases as dc  # noqa: E402
from tests.parity_utils import GRAD_REL, OUT_REL, assert_close, grad_scale, load_generated, load_golden  # noqa: E402
from tests.test_gpu_joint import _qwen_cfg  # noqa: E402

DEV = "cuda"
QWEN_SMALL = [n for n, c in cases.ALL.items() if c["kind"] == "qwen"]


def _decoder_with_inputs_as_injected_tokens(case):
    """The product decoder consumes token ids; to differentiate w.r.t. `inputs_embeds` every position carries its own
    special id and the embeddings ride in as the injected tokens (the J2 path), whose gradient the backward returns."""
    from unirec_amd.qwen3 import Qwen3LoRAModel
    qc = cases.qwen_cfg(case)
    m = Qwen3LoRAModel(_qwen_cfg(qc, False), use_lora=False)
    m = load_generated(m, Q.qwen3_shapes(qc, lora=False), case["seed"] + 1)
    x, am = cases.qwen_inputs(case)
    B, S, Dm = x.shape
    m.resize_token_embeddings(qc.vocab_size + S)
    m = m.to(DEV).train()
    ids = (qc.vocab_size + torch.arange(S, device=DEV)).expand(B, S).contiguous()
    tok = torch.from_numpy(x).to(DEV).to(torch.bfloat16).requires_grad_(True)
    pooled = m.forward_pooled(ids, torch.from_numpy(am).to(DEV), tok, qc.vocab_size)
    return pooled, tok, x


@pytest.mark.parametrize("name", QWEN_SMALL)
def test_decoder_input_gradient_matches_transformers(name):
    case = cases.ALL[name]
    g = load_golden(name)
    pooled, tok, _ = _decoder_with_inputs_as_injected_tokens(case)
    assert_close(pooled, g["sdpa/last_hidden_state"].mean(axis=1), OUT_REL, "pooled")
    pooled.pow(2).sum().backward()
    assert_close(tok.grad.float(), g["sdpa/grad_inputs_embeds"], GRAD_REL, "grad_inputs_embeds")


@pytest.mark.parametrize("name", ["qwen_mid", "qwen_deep"])
def test_qwen3_mid_size_matches_transformers(name):
    """qwen_mid: 4 layers of the 0.6B shape at B*S = 2048 tokens: 256x256 GEMM tiles, 8 causal key tiles per head, GQA 16/8.
    qwen_deep: the model's full depth (28 layers) on 2 x 128 tokens: the accumulated bf16 error of the whole pre-norm stack."""
    case = cases.MID[name]
    g = load_golden(name)
    pooled, tok, x = _decoder_with_inputs_as_injected_tokens(case)
    assert_close(pooled, g["sdpa/pooled"], OUT_REL, "pooled")
    pooled.pow(2).sum().backward()
    got = tok.grad.float().cpu().numpy()
    assert_close(cases.mid_sample(got), g["sdpa/grad_inputs_embeds_s"], GRAD_REL, "grad_inputs_embeds (every 16th position)")
    gn = float(np.linalg.norm(got.astype(np.float64)))
    assert abs(gn - float(g["sdpa/grad_inputs_embeds_norm"])) <= GRAD_REL * gn


@pytest.mark.parametrize("name", ["qwen_lora", "qwen_lora_big"])
def test_lora_matches_merged_transformers(name):
    """(qwen_lora_big: 2048 tokens -- the q|k|v and gate|up launches then run on gemm_pers_kernel<3|4, 1> with the fused epilogues.)
    J4 pinned to a reference-held implementation: the HIP LoRA path (fused RMSNorm / SwiGLU + adapter passes, the second
    K range of the merged q|k|v and gate|up launches, lora_bgrad / lora_reduce; dropout off) against the installed
    transformers Qwen3Model run with MERGED weights W + (alpha / r) B A (tests/golden/qwen_lora.npz): pooled output,
    gradient w.r.t. the input embeddings, and dA = (alpha / r) B^T dW', dB = (alpha / r) dW' A^T of every adapter
    (peft call site training/train_item_individual_token_joint.py:121-131)."""
    from unirec_amd.qwen3 import Qwen3LoRAModel
    case = cases.LORA_CASES[name]
    g = load_golden(name)
    qc = cases.qwen_cfg(case)
    qc.lora_r, qc.lora_alpha = case["lora_r"], case["lora_alpha"]
    m = Qwen3LoRAModel(_qwen_cfg(qc, True), use_lora=True)
    gen = W.fill_state_dict(Q.qwen3_shapes(qc, lora=True), case["seed"] + 1, rules=cases.lora_weight_rules(case))
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in gen.items()}, strict=False)
    assert not unexpected and not [k for k in missing if "embed_tokens" not in k], (missing, unexpected)
    x, am = cases.qwen_inputs(case)
    B, S, _ = x.shape
    m.resize_token_embeddings(qc.vocab_size + S)
    m = m.to(DEV).train()
    assert m._drop_p() == 0.0
    ids = (qc.vocab_size + torch.arange(S, device=DEV)).expand(B, S).contiguous()
    tok = torch.from_numpy(x).to(DEV).to(torch.bfloat16).requires_grad_(True)
    pooled = m.forward_pooled(ids, torch.from_numpy(am).to(DEV), tok, qc.vocab_size)
    assert_close(pooled, g["pooled"], OUT_REL, "pooled")
    pooled.pow(2).sum().backward()
    got = tok.grad.float().cpu().numpy()
    assert_close(cases.mid_sample(got), g["grad_inputs_embeds_s"], GRAD_REL, "grad_inputs_embeds (every 16th position)")
    gn = float(np.linalg.norm(got.astype(np.float64)))
    assert abs(gn - float(g["grad_inputs_embeds_norm"])) <= GRAD_REL * gn
    named = dict(m.named_parameters())
    for i in range(qc.num_hidden_layers):
        for pj in cases.LORA_PROJ:
            for ab in ("lora_A", "lora_B"):
                k = f"layers.{i}.{pj}.{ab}.weight"
                gk = named[k].grad.float().cpu().numpy().astype(np.float64)
                ref = float(g["gnorm/" + k])
                assert abs(float(np.linalg.norm(gk)) - ref) <= GRAD_REL * ref, (k, float(np.linalg.norm(gk)), ref)
                if "grad/" + k in g:
                    assert_close(gk, g["grad/" + k], GRAD_REL, "grad/" + k)
    assert named["layers.0.self_attn.q_proj.weight"].grad is None      # base weights stay frozen


@pytest.mark.parametrize("spread", ["narrow", "pair_ratio_3", "log_uniform_0.05_20"])
def test_fused_qknorm_rope_epilogue_matches_the_separate_pass(monkeypatch, spread):
    """(spread: the q / k norm weights.  The fused backward recovers x^ = R^T(o) / w from bf16 outputs, which amplifies rounding by the
    spread of a rotate-half pair's weights: up to a ratio of 4 the fused path runs and stays within tolerance; beyond it -- trained
    checkpoints may spread widely -- the model falls back to the separate pass and the results are IDENTICAL.)
    Decoder level: with >= 8192 tokens the q|k|v launch carries q/k-norm + RoPE in its epilogue (csrc/gemm_pers.hip, the raw
    q, k are never stored; the backward recovers the rows from the roped outputs).  Same weights, inputs and LoRA dropout
    masks: pooled output, input gradient and every LoRA gradient against the path with the separate ur_qknorm_rope pass."""
    import unirec_amd.qwen3 as qmod
    from unirec_amd.qwen3 import Qwen3Config, Qwen3LoRAModel
    B, S, L = 4, 2048, 2
    cfg = Qwen3Config(vocab_size=512, num_hidden_layers=L, lora_dropout=0.1)
    torch.manual_seed(7)
    m = Qwen3LoRAModel(cfg, use_lora=True)
    m.reset_parameters(lora_b_std=0.02)
    def norm_weights():
        if spread == "narrow":
            return 1.0 + 0.1 * torch.randn(128)
        if spread == "pair_ratio_3":          # every rotate-half pair (d, d + 64) spread by up to 3x, magnitudes 0.3 .. 3
            base = torch.exp(torch.empty(64).uniform_(-1.2, 0.0))
            return torch.cat([base, base * torch.empty(64).uniform_(1.0, 3.0)]) * torch.where(torch.rand(128) < 0.5, -1.0, 1.0)
        return torch.exp(torch.empty(128).uniform_(-3.0, 3.0))          # log-uniform 0.05 .. 20
    with torch.no_grad():
        for lyr in m.layers:
            lyr.self_attn.q_norm.weight.copy_(norm_weights())
            lyr.self_attn.k_norm.weight.copy_(norm_weights())
    m = m.to(DEV).train()
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 512, (B, S), generator=g).to(DEV)
    am = torch.ones(B, S, dtype=torch.long)
    am[1, :300] = 0
    am[3, :77] = 0
    am = am.to(DEV)
    calls = []
    real = qmod.hip.gemm_qkv_rope
    monkeypatch.setattr(qmod.hip, "gemm_qkv_rope", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def run(fused):
        monkeypatch.setattr(qmod, "_FUSE_QK_ROPE", fused)
        m._lora_step, m._bcomb = 11, None
        for p_ in m.parameters():
            p_.grad = None
        if m.pack is not None:
            m.pack.clear_grads()
        pooled = m.forward_pooled(ids, am)
        pooled.pow(2).sum().backward()
        torch.cuda.synchronize()
        return pooled.detach().clone(), {n: p_.grad.detach().clone() for n, p_ in m.named_parameters() if p_.grad is not None}
    p0, g0 = run(False)
    assert not calls
    p1, g1 = run(True)
    if spread == "log_uniform_0.05_20":
        assert not calls, "badly conditioned norm weights must take the separate pass"
        assert torch.equal(p0, p1) and all(torch.equal(g0[n], g1[n]) for n in g0)
        return
    assert len(calls) == L, "the fused epilogue did not run"
    assert float((p1 - p0).norm() / p0.norm()) <= 5e-3
    assert set(g0) == set(g1) and len(g0) == 14 * L
    for n in g0:
        rel = float((g1[n] - g0[n]).norm() / g0[n].norm())
        assert rel <= 2e-2, (n, rel)


@pytest.mark.parametrize("tokens", [(4, 2048), (2, 256)])
def test_recompute_mlp_switch_is_bit_identical_and_frees_the_mlp_activations(tokens):
    """recompute_mlp drops gate|up and act after the forward and rebuilds them in the backward with the launch that made them:
    pooled output and every gradient are bit-identical, the saved state holds no [M, 2I] / [M, I] tensor, and the peak memory
    of forward + backward falls.  (4, 2048) takes the paired SwiGLU epilogue of the persistent GEMM, (2, 256) the merged
    gate|up launch + stand-alone SwiGLU pass."""
    from unirec_amd.qwen3 import Qwen3Config, Qwen3LoRAModel
    B, S = tokens
    L = 3
    cfg = Qwen3Config(vocab_size=512, num_hidden_layers=L, lora_dropout=0.1)
    torch.manual_seed(5)
    m = Qwen3LoRAModel(cfg, use_lora=True)
    m.reset_parameters(lora_b_std=0.02)
    m = m.to(DEV).train()
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(0, 512, (B, S), generator=g).to(DEV)
    am = torch.ones(B, S, dtype=torch.long)
    am[1, :100] = 0
    am = am.to(DEV)
    seen = {}
    real_bwd = m._backward_impl

    def spy(saved, d_pooled):
        seen["kinds"] = [lyr.get("mlp_recompute") for lyr in saved["layers"]]
        seen["big"] = sum(1 for lyr in saved["layers"] for k in ("gu", "act") if lyr.get(k) is not None)
        return real_bwd(saved, d_pooled)
    m._backward_impl = spy

    def run(recompute):
        m.recompute_mlp = recompute
        m._lora_step = 4
        for p_ in m.parameters():
            p_.grad = None
        if m.pack is not None:
            m.pack.clear_grads()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        pooled = m.forward_pooled(ids, am)
        pooled.pow(2).sum().backward()
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated()
        return pooled.detach().clone(), {n: p_.grad.detach().clone() for n, p_ in m.named_parameters() if p_.grad is not None}, peak
    p0, g0, peak0 = run(False)
    assert seen["big"] == 2 * L and not any(seen["kinds"])
    p1, g1, peak1 = run(True)
    assert seen["big"] == 0 and all(seen["kinds"]), seen
    assert seen["kinds"][0] == ("pair" if B * S >= 8192 else "merged"), seen
    assert torch.equal(p0, p1)
    assert set(g0) == set(g1) and len(g0) == 14 * L
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    M, I = B * S, cfg.intermediate_size
    assert peak0 - peak1 >= (L - 2) * M * 3 * I * 2, (peak0, peak1)          # L - 1 layers' worth less, give or take where the peak falls
    # round 6: recompute_mlp = k rebuilds layers 0 .. k - 1 only (bench.py's C5 line keeps as many layers as 288 GB allow)
    p2, g2, _ = run(2)
    assert seen["big"] == 2 * (L - 2) and [bool(k) for k in seen["kinds"]] == [True, True, False], seen
    assert torch.equal(p0, p2) and all(torch.equal(g0[n], g2[n]) for n in g0)


def test_user_qformer_mid_size_matches_reference():
    """The reference's default UserQFormer (L4 Q64 H1024 I4096) over T = 1600 keys: C3's shapes at B = 2."""
    from unirec_amd.user_qformer import UserQFormer
    case = cases.MID["user_mid"]
    c = case["cfg"]
    g = load_golden("user_mid")
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 1)
    m = UserQFormer(dropout=0.0)
    m = load_generated(m, R.user_qformer_shapes(cfg, c["n_pred"]), case["seed"]).train()
    x, mask, tgt = cases.user_inputs(case)
    pred = m(torch.from_numpy(x).to(DEV), torch.from_numpy(mask).to(DEV))
    assert_close(pred, g["predicted_item_tokens"], OUT_REL, "predicted_item_tokens")
    loss = ((pred - torch.from_numpy(tgt).to(DEV)) ** 2).mean()
    assert_close(loss, g["loss"], OUT_REL, "loss")
    loss.backward()
    named = dict(m.named_parameters())
    gs = grad_scale(g, cases.user_grad_keys(c))
    for k in cases.user_grad_keys(c):
        assert_close(cases.trim_like(named[k].grad.float().cpu().numpy()), g["grad/" + k], GRAD_REL, "grad/" + k, floor=1e-6, ref_scale=gs)


def test_item_qformer_mid_size_matches_reference():
    """BASELINE configs[1]'s architecture exactly (C2: L12 Q32 H768 nh12 I3072 F14 E1024) at B = 16, against vectors the reference's
    own QFormerForItemRepresentation / QFormerLoss produced: outputs, loss, eval metrics (HIP loss kernels) and gradients."""
    from unirec_amd import hip
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    case = cases.MID["item_mid"]
    c = case["cfg"]
    g = load_golden("item_mid")
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    m = QFormerForItemRepresentation(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                                     num_query_tokens=c["Q"], field_embedding_dim=c["E"], num_fields=c["F"], dropout=0.0)
    m = load_generated(m, R.item_qformer_shapes(cfg, c["F"]), case["seed"]).train()
    x, mask = cases.item_inputs(case)
    xt, mt = torch.from_numpy(x).to(DEV), torch.from_numpy(mask).to(DEV)
    out = m(xt, mt)
    got = cases.item_mid_sample({k: out[k].detach().float().cpu().numpy() for k in ("query_outputs", "item_representation", "reconstructed_fields")})
    for k in ("query_outputs", "item_representation", "reconstructed_fields"):
        assert_close(got[k], g[k], OUT_REL, k)
    sums = hip.recon_stats(out["reconstructed_fields"].detach(), xt, mt.float()).cpu().numpy()
    assert_close(sums[0] / sums[1], g["eval_mse"], OUT_REL, "eval_mse")
    assert_close(sums[2], g["eval_cos_sum"], OUT_REL, "eval_cos_sum")
    pos, neg = cases.triplet_reps(case)
    loss, rl, cl = R.qformer_loss({k: v for k, v in out.items()}, xt, mt, torch.from_numpy(pos).to(DEV), torch.from_numpy(neg).to(DEV))
    assert_close(loss, g["loss"], OUT_REL, "loss")
    loss.backward()
    named = dict(m.named_parameters())
    gs = grad_scale(g, cases.item_grad_keys(c))
    for k in cases.item_grad_keys(c):
        assert_close(cases.trim_like(named[k].grad.float().cpu().numpy()), g["grad/" + k], GRAD_REL, "grad/" + k, floor=1e-6, ref_scale=gs)


def test_joint_mid_size_matches_reference():
    """The headline path against reference-produced vectors at the 0.6B decoder's layer shape: item Q-Former (H = D = 1024, Q 2, F 14)
    -> injection -> 2 Qwen3 layers (16 / 8 heads of 128, I 3072), S 512, left padding, hist 10: user embeddings, InfoNCE loss,
    MRR ranks (exact) and the item Q-Former's gradients THROUGH the decoder (train_item_individual_token_joint.py:133-212,326-352,408-419)."""
    from tests.test_gpu_joint import _build_joint
    from unirec_amd.joint import InfoNCELoss, mrr_ranks
    case = cases.MID["joint_mid"]
    c = case["cfg"]
    g = {k[5:]: v for k, v in load_golden("joint_mid").items() if k.startswith("sdpa/")}
    m, qf = _build_joint(case, use_lora=False)
    ids, am, hfe, ham, pos, neg, nmask = cases.joint_inputs(case)
    t = lambda a: torch.from_numpy(a).to(DEV)
    user = m(t(ids), t(am), t(hfe), t(ham))
    assert_close(user, g["user_embeddings"], OUT_REL, "user_embeddings")
    loss = InfoNCELoss()(user, t(pos), t(neg), t(nmask))
    assert_close(loss, g["loss"], OUT_REL, "infonce loss")
    scores, rank = mrr_ranks(user, t(pos), t(neg))
    assert rank.cpu().tolist() == g["ranks"].tolist()
    loss.backward()
    named = dict(qf.named_parameters())
    ill = 0.05 * float(np.linalg.norm(g["grad/query_embeddings"]))
    ILL_KEYS = ("qformer.encoder.layer.0.attention.self.query.weight", "qformer.encoder.layer.0.attention.self.key.bias")       # see test_gpu_joint.py
    gs = grad_scale(g, cases.item_grad_keys(c, heads=False))
    for k in cases.item_grad_keys(c, heads=False):
        assert_close(cases.trim_like(named[k].grad.float().cpu().numpy()), g["grad/" + k], GRAD_REL * 1.5, "grad/" + k, floor=1e-6,
                     abs_scale=ill if k in ILL_KEYS else 0.0, ref_scale=gs)


# ---- the UserSequenceEncoder boundary (models/user_sequence_encoder.py:36-142) ----------------------------------------
USE_CFG = dict(H=dc.CTX_H, L=2, nh=2, I=256, Q=dc.QI, E=dc.E, seed=61)      # = tests/golden/make_golden_r2.py:USE_CFG


def _write_item_checkpoint(path, fields, config_as):
    """A checkpoint in the format training/item_qformer_training.py:178-186 writes: {'model_state_dict','config','field_names'}."""
    from unirec_amd.qformer_utils import QFormerForItemRepresentation
    c = USE_CFG
    m = QFormerForItemRepresentation(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                                     num_query_tokens=c["Q"], field_embedding_dim=c["E"], num_fields=len(fields), dropout=0.2)
    gen = W.fill_state_dict(R.item_qformer_shapes(R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2), len(fields)), c["seed"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gen.items()}, strict=False)
    cfg = m.config if config_as == "object" else dict(m.config.__dict__)
    torch.save({"model_state_dict": m.state_dict(), "config": cfg, "field_names": fields}, path)


@pytest.mark.parametrize("config_as", ["object", "dict"])
def test_user_sequence_encoder_drop_in(tmp_path, config_as):
    from unirec_amd.user_sequence_encoder import UserSequenceEncoder
    g = load_golden("use_real")
    samples, item_dict = dc.item_samples()
    fields = [str(f) for f in g["fields"]]
    ck = tmp_path / "best_qformer_model.pth"
    _write_item_checkpoint(str(ck), fields, config_as)
    cfgp = tmp_path / "triplet_config.yaml"
    cfgp.write_text("FIELD_MAPPING:\n" + "".join(f"  {f}: [{i}, 0, text]\n" for i, f in enumerate(fields)) + "MODALITY_IDS:\n  text: 0\n")
    enc = UserSequenceEncoder(item_qformer_checkpoint_path=str(ck), item_encoder_config_path=str(cfgp), item_encoder=dc.FakeItemEncoder())
    H = USE_CFG["H"]
    assert enc.embedding_dim == H and enc.item_qformer_fields == fields and not enc.item_qformer.training
    assert enc.item_qformer.config.num_hidden_layers == USE_CFG["L"] and enc.item_qformer.num_query_tokens == dc.QI
    enc.timestamp_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in W.fill_state_dict(D.context_mlp_shapes(H, 9), dc.CTX_SEED).items()})
    enc.geo_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in W.fill_state_dict(D.context_mlp_shapes(H, 3), dc.CTX_SEED + 1).items()})
    enc.timestamp_encoder.to(DEV); enc.geo_encoder.to(DEV)
    events = [dict(e, item_data=item_dict[e["item_id"]]) for e in dc.user_events()[0]]
    toks = enc._get_item_query_tokens_batch([e["item_data"] for e in events])
    assert toks.dtype == torch.float32 and tuple(toks.shape) == (5, dc.QI, H)
    assert_close(toks, g["item_query_tokens"], OUT_REL, "_get_item_query_tokens_batch")
    enc.positional_encoder.eval()                      # the fixture was generated with the positional dropout off
    seq = enc.encode_user_sequence(events)
    assert tuple(seq.shape) == (5 * dc.QI, H)          # the reference's own shape check (:187-189)
    assert_close(seq, g["encoded_user_sequence"], OUT_REL, "encode_user_sequence")
    enc.positional_encoder.train()                     # reference default: dropout(0.1) active
    seq2 = enc.encode_user_sequence(events)
    zeros = float((seq2 == 0).float().mean())
    assert 0.03 < zeros < 0.25, zeros
    assert tuple(enc.encode_user_sequence([]).shape) == (0, dc.QI, H)


def test_user_sequence_encoder_without_encoders_raises(tmp_path):
    from unirec_amd.item_encoder_pure_value import ModalityEncodersUnavailable
    from unirec_amd.user_sequence_encoder import UserSequenceEncoder
    fields = ["title", "price"]
    ck = tmp_path / "ck.pth"
    _write_item_checkpoint(str(ck), fields, "object")
    cfgp = tmp_path / "cfg.yaml"
    cfgp.write_text("FIELD_MAPPING:\n  title: [0, 0, text]\n  price: [1, 3, number]\nMODALITY_IDS:\n  text: 0\n  number: 3\n")
    enc = UserSequenceEncoder(str(ck), str(cfgp))
    with pytest.raises(ModalityEncodersUnavailable):
        enc._get_item_query_tokens_batch([{"item_id": "a", "title": "x", "price": 1.0}])


# ---- full-size configurations through size-independent properties -------------------------------------------------------
def _grads(m):
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def _worst_rel(ga, gb, g):
    worst = 0.0
    for k in g:
        s = ga[k].float() + gb[k].float()
        worst = max(worst, ((s - g[k].float()).norm() / (g[k].float().norm() + 1e-20)).item())
    return worst


def test_c2_full_size_item_step_properties():
    """C2: item Q-Former L12 Q32 H768 nh12 I3072 F14 E1024, B=256, the triplet step's loss (models/qformer_utils.py:17-60,
    training/item_qformer_training.py:117-131): determinism, shard invariance of the forward, additivity of gradients."""
    from unirec_amd.losses import QFormerLoss
    from unirec_amd.qformer_utils import QFormerForItemRepresentation
    torch.manual_seed(3)
    B = 256
    m = QFormerForItemRepresentation(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                                     num_query_tokens=32, field_embedding_dim=1024, num_fields=14, dropout=0.0).to(DEV).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 14, 1024, generator=g); x = x / x.norm(dim=-1, keepdim=True)
    mk = (torch.rand(B, 14, generator=g) < 0.8).long(); mk[:, 0] = 1
    x, mk = (x * mk[..., None]).to(DEV), mk.to(DEV)
    pr, nr = (torch.randn(B, 1024, generator=g) * 0.05).to(DEV), (torch.randn(B, 1024, generator=g) * 0.05).to(DEV)
    loss_fn = QFormerLoss()

    def step(sl, recon_only=False):
        for p in m.parameters():
            p.grad = None
        out = m(x[sl], mk[sl])
        loss, recon, cont = loss_fn(out, {"field_embeddings": x[sl]}, pr[sl], nr[sl], mk[sl])
        (cont if recon_only else loss).backward()
        return out, loss.detach().clone(), _grads(m)
    full = slice(0, B)
    out, loss, gr = step(full)
    assert all(torch.isfinite(v).all() for v in out.values()) and torch.isfinite(loss)
    out2, loss2, gr2 = step(full)
    assert torch.equal(loss, loss2) and all(torch.equal(out[k], out2[k]) for k in out)
    assert [k for k in gr if not torch.equal(gr[k], gr2[k])] == []                       # run-to-run bitwise determinism
    assert len(gr) == len([n for n, p in m.named_parameters() if p.requires_grad])       # every live tensor receives a gradient
    with torch.no_grad():
        o = m(x[64:128], mk[64:128])
    for k in out:
        assert torch.equal(o[k], out[k][64:128]), k                                      # shard invariance, bit for bit
    # additivity on the per-sample-mean triplet term (the masked MSE has a batch-global denominator: covered by the DP test)
    _, _, g_all = step(full, recon_only=True)
    _, _, ga = step(slice(0, 128), recon_only=True)
    _, _, gb = step(slice(128, 256), recon_only=True)
    ga = {k: v * 0.5 for k, v in ga.items()}
    gb = {k: v * 0.5 for k, v in gb.items()}
    assert _worst_rel(ga, gb, g_all) < 2e-2


def test_c3_full_size_user_step_properties():
    """C3: UserQFormer L4 Q64 H1024 over T = 1600 cached item tokens, B = 512, ragged (training/user_qformer_training.py:21-68)."""
    from unirec_amd.losses import mse_loss
    from unirec_amd.user_qformer import UserQFormer
    torch.manual_seed(4)
    B, T = 512, 1600
    m = UserQFormer(dropout=0.0).to(DEV).train()
    g = torch.Generator().manual_seed(6)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g)
    lens[0], lens[1] = T, 1
    mask = (torch.arange(T)[None, :] < lens[:, None]).float().to(DEV)
    x = torch.empty(B, T, 1024, dtype=torch.bfloat16, device=DEV)
    for i in range(0, B, 64):
        x[i:i + 64] = (torch.randn(64, T, 1024, generator=g) * 0.8).to(torch.bfloat16).to(DEV)
    x = x * mask[..., None].to(torch.bfloat16)
    tgt = (torch.randn(B, 32, 1024, generator=g) * 0.8).to(DEV)

    def step(sl, scale=1.0):
        for p in m.parameters():
            p.grad = None
        pred = m(x[sl], mask[sl])
        loss = mse_loss(pred, tgt[sl]) * scale
        loss.backward()
        return pred.detach().clone(), loss.detach().clone(), _grads(m)
    pred, loss, gr = step(slice(0, B))
    assert tuple(pred.shape) == (B, 32, 1024) and torch.isfinite(pred).all() and torch.isfinite(loss)
    pred2, loss2, gr2 = step(slice(0, B))
    assert torch.equal(pred, pred2) and torch.equal(loss, loss2) and all(torch.equal(gr[k], gr2[k]) for k in gr)
    with torch.no_grad():
        p0 = m(x[:64], mask[:64])
    assert torch.equal(p0, pred[:64])                                # shard invariance incl. the 1-key and full-length rows
    _, la, ga = step(slice(0, 256), 0.5)
    _, lb, gb = step(slice(256, 512), 0.5)
    assert abs((la + lb).item() - loss.item()) <= 1e-5 * abs(loss.item())
    assert _worst_rel(ga, gb, gr) < 2e-2
    # padded keys are inert: garbage behind a sample's length changes nothing
    x2 = x.clone()
    x2[1, 1:] = 7.0
    with torch.no_grad():
        assert torch.equal(m(x2[:64], mask[:64]), p0)


def test_c5_shaped_step_properties():
    """C5: hist = 100, S = 4096, pool 10000, the User Q-Former's 64 tokens injected (U4), 28 layers -- at B = 16 per
    launch (the full B = 64 is the micro-batched / recomputed step of bench.py): determinism, shard invariance, exact
    ranks and top-K over the 10000-candidate pool."""
    import bench
    from unirec_amd import hip
    from unirec_amd.joint import InfoNCELoss, mrr_ranks
    args = argparse.Namespace(layers=28, hist=100, seq=4096, pool=10000, no_dropout=True, lora_dropout=0.0, user_tokens=True)
    model, qf, cfg, (Qi, F, E, Dm) = bench.build(args, torch.device(DEV))
    B = 16
    b = bench.make_batch(B, args.hist, args.seq, args.pool, F, E, Dm, Qi, model.first_special_id, model.first_special_id, 99, DEV,
                         n_user=model.num_user_query_tokens)
    model.train()

    def fwd(sl):
        return model(b["input_ids"][sl], b["attention_mask"][sl], b["history_field_embeddings"][sl], b["history_attention_mask"][sl],
                     b["user_sequence_tokens"][sl], b["user_attention_mask"][sl])

    def step():
        for p in model.parameters():
            p.grad = None
        u = fwd(slice(0, B))
        loss = InfoNCELoss()(u, b["positive_item_embeddings"], b["negative_item_embeddings"])
        loss.backward()
        return u.detach().clone(), loss.detach().clone(), _grads(model)
    u, loss, gr = step()
    assert torch.isfinite(u).all() and torch.isfinite(loss)
    u2, loss2, gr2 = step()
    assert torch.equal(u, u2) and torch.equal(loss, loss2) and all(torch.equal(gr[k], gr2[k]) for k in gr)
    assert any(k.startswith("user_qformer.") for k in gr) and any(k.startswith("qformer_model.") for k in gr) and any(".lora_" in k for k in gr)
    with torch.no_grad():
        assert torch.equal(fwd(slice(4, 8)), u[4:8])
    scores, rank = mrr_ranks(u, b["positive_item_embeddings"], b["negative_item_embeddings"])
    assert tuple(scores.shape) == (B, 10000)
    assert torch.equal(rank.long(), 1 + (scores[:, 1:] > scores[:, :1]).sum(1))
    idx, val = hip.topk(scores, 10)
    order = torch.sort(scores, dim=1, descending=True, stable=True)
    assert torch.equal(idx.long(), order.indices[:, :10]) and torch.equal(val, order.values[:, :10])


def test_c5_contract_batch_single_launch_with_recompute():
    """C5 at the CONTRACT per-GPU batch: B = 64, hist = 100, S = 4096, pool 10000, user tokens, 28 layers, as ONE launch with
    recompute_mlp (gate|up and act are rebuilt in the backward: ~193 GB instead of > 288 GB).  Determinism, shard invariance of the
    forward against the B = 16 rows, exact ranks and top-K over the 10000-candidate pool."""
    import bench
    from unirec_amd import hip
    from unirec_amd.joint import InfoNCELoss, mrr_ranks
    if torch.cuda.get_device_properties(0).total_memory < 250 * 2**30:
        pytest.skip("needs the 288 GB of an MI355X")
    args = argparse.Namespace(layers=28, hist=100, seq=4096, pool=10000, no_dropout=True, lora_dropout=0.0, user_tokens=True)
    model, qf, cfg, (Qi, F, E, Dm) = bench.build(args, torch.device(DEV))
    model.base_model.recompute_mlp = True
    B = 64
    b = bench.make_batch(B, args.hist, args.seq, args.pool, F, E, Dm, Qi, model.first_special_id, model.first_special_id, 99, DEV,
                         n_user=model.num_user_query_tokens)
    model.train()

    def fwd(sl):
        return model(b["input_ids"][sl], b["attention_mask"][sl], b["history_field_embeddings"][sl], b["history_attention_mask"][sl],
                     b["user_sequence_tokens"][sl], b["user_attention_mask"][sl])

    def step():
        for p in model.parameters():
            p.grad = None
        u = fwd(slice(0, B))
        loss = InfoNCELoss()(u, b["positive_item_embeddings"], b["negative_item_embeddings"])
        loss.backward()
        gr = {k: v.clone() for k, v in _grads(model).items() if ".lora_" in k or k.startswith("user_qformer.qformer.encoder.layer.0.")}
        return u.detach().clone(), loss.detach().clone(), gr
    u, loss, gr = step()
    assert torch.isfinite(u).all() and torch.isfinite(loss) and torch.cuda.max_memory_allocated() < 260 * 2**30
    u2, loss2, gr2 = step()
    assert torch.equal(u, u2) and torch.equal(loss, loss2) and all(torch.equal(gr[k], gr2[k]) for k in gr)
    assert any(k.startswith("user_qformer.") for k in gr) and any(".lora_" in k for k in gr)
    with torch.no_grad():
        assert torch.equal(fwd(slice(16, 32)), u[16:32])           # a 16-row shard of the batch: the same rows, bit for bit
    scores, rank = mrr_ranks(u, b["positive_item_embeddings"], b["negative_item_embeddings"])
    assert tuple(scores.shape) == (B, 10000)
    assert torch.equal(rank.long(), 1 + (scores[:, 1:] > scores[:, :1]).sum(1))
    idx, val = hip.topk(scores, 10)
    order = torch.sort(scores, dim=1, descending=True, stable=True)
    assert torch.equal(idx.long(), order.indices[:, :10]) and torch.equal(val, order.values[:, :10])


# ---- empty and degenerate inputs at the class boundary -------------------------------------------------------------------
def test_empty_and_single_sample_batches():
    """B = 0 goes through every stage without a launch fault and returns correctly shaped empty tensors (the reference's modules do
    the same on empty tensors); B = 1 with a fully masked item / a length-1 user history matches the oracle."""
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    from unirec_amd.user_qformer import UserQFormer
    c = cases.ALL["item_c1"]["cfg"]
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    m = QFormerForItemRepresentation(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                                     num_query_tokens=c["Q"], field_embedding_dim=c["E"], num_fields=c["F"], dropout=0.0)
    m = load_generated(m, R.item_qformer_shapes(cfg, c["F"]), 11).eval()
    with torch.no_grad():
        out = m(torch.zeros(0, c["F"], c["E"], device=DEV), torch.zeros(0, c["F"], dtype=torch.long, device=DEV))
    assert out["query_outputs"].shape == (0, c["Q"], c["H"]) and out["item_representation"].shape == (0, c["E"])
    assert out["reconstructed_fields"].shape == (0, c["F"], c["E"])
    # one item whose fields are ALL masked (uniform attention over the masked keys) against the oracle
    g = torch.Generator().manual_seed(3)
    x1 = torch.randn(1, c["F"], c["E"], generator=g)
    mk = torch.zeros(1, c["F"], dtype=torch.long)
    with torch.no_grad():
        got = m(x1.to(DEV), mk.to(DEV))["query_outputs"]
    P = {k: torch.from_numpy(v) for k, v in W.fill_state_dict(R.item_qformer_shapes(cfg, c["F"]), 11).items()}
    want = R.item_qformer_forward(P, cfg, x1, mk)["query_outputs"]
    assert_close(got, want.numpy(), OUT_REL, "fully masked single item")
    # user Q-Former: empty batch, and one user with a single valid key
    uc = cases.ALL["user_t8"]["cfg"]
    ucfg = R.QFormerCfg(uc["H"], uc["L"], uc["nh"], uc["I"], uc["Q"], uc["E"], 1)
    u = UserQFormer(hidden_size=uc["H"], num_hidden_layers=uc["L"], num_attention_heads=uc["nh"], intermediate_size=uc["I"],
                    num_query_tokens=uc["Q"], input_embedding_dim=uc["E"], num_item_tokens_to_predict=uc["n_pred"], dropout=0.0)
    u = load_generated(u, R.user_qformer_shapes(ucfg, uc["n_pred"]), 22).eval()
    with torch.no_grad():
        e = u(torch.zeros(0, 8, uc["E"], device=DEV), torch.zeros(0, 8, device=DEV))
    assert e.shape == (0, uc["n_pred"], uc["E"])
    xs = torch.randn(1, 8, uc["E"], generator=g) * 0.8
    ms = torch.zeros(1, 8); ms[0, 0] = 1.0
    xs = xs * ms[..., None]
    with torch.no_grad():
        got = u(xs.to(DEV), ms.to(DEV))
    PU = {k: torch.from_numpy(v) for k, v in W.fill_state_dict(R.user_qformer_shapes(ucfg, uc["n_pred"]), 22).items()}
    want, _ = R.user_qformer_forward(PU, ucfg, xs, ms, uc["n_pred"])
    assert_close(got, want.numpy(), OUT_REL, "single user, one valid key")

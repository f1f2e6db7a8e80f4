"""CPU: the round-6 reference fixture (tests/golden/make_golden_r6.py) against the oracle: BASELINE configs[3] (C4, the headline) at its
exact architecture and lengths -- 12-layer H 1024 item Q-Former on 50 history items -> 100 injected tokens -> 28 decoder layers of the
0.6B shape with LoRA r 16 at S 2048, left padding -> all-S mean pool -> InfoNCE over a pool of 1000 -> MRR rank -> gradients back into
query_embeddings, Q-Former weights and the adapters (reference: training/train_item_individual_token_joint.py:134-181,331-352,392-419)."""
import os

import numpy as np
import torch

from oracle import qformer_ref as R
from oracle import qwen3_ref as Q
from oracle import weights as W
from tests.golden import cases
from tests.test_oracle_golden import _close, _load

torch.set_num_threads(min(8, os.cpu_count() or 1))


def oracle_joint_c4(case):
    """(user embeddings, loss, ranks, PQ, PL) of the oracle on the case; the callers read gradients off PQ / PL."""
    c = case["cfg"]
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    qc = cases.qwen_cfg(case)
    qc.lora_r, qc.lora_alpha, qc.lora_dropout = case["lora_r"], case["lora_alpha"], 0.0
    PQ = {k: torch.from_numpy(v).requires_grad_(True) for k, v in W.fill_state_dict(R.item_qformer_shapes(cfg, c["F"]), case["seed"]).items()}
    sd = W.fill_state_dict(Q.qwen3_shapes(qc, lora=True), case["seed"] + 1, rules=cases.lora_weight_rules(case))
    PW = {k: torch.from_numpy(v).requires_grad_(".lora_" in k) for k, v in sd.items()}
    ids, am, hfe, ham, pos, neg, nmask = cases.joint_inputs(case)
    B, hist = case["B"], case["hist"]
    out = R.item_qformer_forward(PQ, cfg, torch.from_numpy(hfe).view(B * hist, c["F"], c["E"]), torch.from_numpy(ham).view(B * hist, c["F"]))
    toks = out["query_outputs"].view(B, hist, c["Q"], c["H"])
    user = Q.joint_forward(PW, qc, torch.from_numpy(ids), torch.from_numpy(am), toks, case["first_special_id"], fully_masked="zero",
                           checkpoint_layers=True)          # (the same arithmetic; 28 layers x 0.5 GB of saved probabilities would cost more time than the extra forward)
    loss = Q.infonce_loss(user, torch.from_numpy(pos), torch.from_numpy(neg), torch.from_numpy(nmask))
    _, rank = Q.mrr_ranks(user.detach(), torch.from_numpy(pos), torch.from_numpy(neg))
    loss.backward()
    return user.detach().numpy(), loss.detach().numpy(), rank.tolist(), PQ, PW


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_joint_c4_exact_architecture(golden_dir):
    case = cases.C4["joint_c4"]
    g = _load(golden_dir, "joint_c4")
    assert g["n_pad"].tolist()[0] == 0 and g["n_pad"].tolist()[1] > 0          # one full and one left-padded sequence
    user, loss, ranks, PQ, PW = oracle_joint_c4(case)
    _close(user, g["user_embeddings"], rtol=2e-3, atol=1e-4, what="user")
    _close(loss, g["loss"], rtol=1e-3, what="loss")
    assert ranks == g["ranks"].tolist()
    # gradients: merged (fixture) and unmerged (oracle) weights round differently in fp32 -> Frobenius norms
    assert rel(PQ["query_embeddings"].grad.numpy(), g["grad/query_embeddings"]) <= 2e-3
    for k in cases.C4_QF_KEYS:
        got = PQ[k].grad.numpy()
        assert rel(cases.c4_rows(got), g["grad/" + k]) <= 2e-3, k
        assert abs(float(np.linalg.norm(got.astype(np.float64))) - float(g["gnorm/" + k])) <= 2e-3 * float(g["gnorm/" + k]), k
    n_full = 0
    for i in range(28):
        for pj in cases.LORA_PROJ:
            for ab in ("lora_A", "lora_B"):
                k = f"layers.{i}.{pj}.{ab}.weight"
                got = PW[k].grad.numpy().astype(np.float64)
                ref = float(g["gnorm/" + k])
                assert abs(float(np.linalg.norm(got)) - ref) <= 2e-3 * ref, (k, float(np.linalg.norm(got)), ref)
                if "grad/" + k in g:
                    assert rel(got, g["grad/" + k]) <= 2e-3, k
                    n_full += 1
    assert n_full == 2 * len(cases.C4_LORA_FULL)

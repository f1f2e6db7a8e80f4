#!/usr/bin/env python3
"""Round-6 golden vector: BASELINE configs[3] (C4, the headline) at its exact architecture and lengths, produced by running the
REFERENCE's own Python on CPU (build container only; same shim as make_golden.py, which this script imports).

  joint_c4.npz   the reference's QFormerForItemRepresentation (L 12, H 1024, Q 2, F 14; models/qformer_model.py:6-50) inside the
                 reference's MultiModalQwenEmbedding.forward (training/train_item_individual_token_joint.py:134-181, built with
                 __new__ as SURVEY 8(c) prescribes) over the installed transformers Qwen3Model at the 0.6B shape (28 layers,
                 D 1024, 16 / 8 heads of 128, I 3072) carrying MERGED LoRA weights W + (alpha / r) B A (r 16; peft is absent, call
                 site :121-131), B 2 x S 2048, hist 50 (100 injected tokens per sequence), left padding, dropout off; the
                 reference's InfoNCELoss (:331-352) over 999 negatives + the positive and the MRR rank exactly as :408-419.
                 Stored: pooled user embeddings, loss, ranks, the gradient of query_embeddings, four Q-Former weight gradients
                 (every 32nd row + norm), dA / dB of the first and the last layer's adapters in full (dA = (alpha / r) B^T dW',
                 dB = (alpha / r) dW' A^T) and the norm of every adapter's gradient.

Usage:  python tests/golden/make_golden_r6.py          (~2 min of CPU, ~12 GB)
"""
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import numpy as np
import torch

from tests.golden import make_golden as mg          # installs the shim and imports the reference modules
from tests.golden import cases
from oracle import weights as W
from oracle.qformer_ref import QFormerCfg, item_qformer_shapes
from oracle.qwen3_ref import qwen3_shapes


def gen_joint_c4(case):
    c = case["cfg"]
    rj = mg.rj
    qf = mg.RefItemQFormer(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                           num_query_tokens=c["Q"], field_embedding_dim=c["E"], num_fields=c["F"], dropout=0.0)
    mg.load_generated(qf, item_qformer_shapes(QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2), c["F"]), case["seed"])
    qc = cases.qwen_cfg(case)
    sc = case["lora_alpha"] / case["lora_r"]
    gen = W.fill_state_dict(qwen3_shapes(qc, lora=True), case["seed"] + 1, rules=cases.lora_weight_rules(case))
    base = mg.build_hf_qwen3(qc, case["seed"] + 1, "sdpa")
    sd = base.state_dict()
    for i in range(qc.num_hidden_layers):
        for pj in cases.LORA_PROJ:
            n = f"layers.{i}.{pj}"
            a, b = gen[n + ".lora_A.weight"].astype(np.float64), gen[n + ".lora_B.weight"].astype(np.float64)
            assert np.array_equal(sd[n + ".weight"].numpy(), gen[n + ".weight"])
            sd[n + ".weight"] = torch.from_numpy((gen[n + ".weight"].astype(np.float64) + sc * (b @ a)).astype(np.float32))
    base.load_state_dict(sd)
    m = rj.MultiModalQwenEmbedding.__new__(rj.MultiModalQwenEmbedding)
    torch.nn.Module.__init__(m)
    m.qformer_model = qf
    m.num_history_items = case["hist"]
    m.num_query_tokens_per_item = c["Q"]
    m.hidden_size = qc.hidden_size
    m.tokenizer = mg.FakeTokenizer(case["first_special_id"], case["hist"], c["Q"])
    m.base_model = base
    m.use_lora = False
    m.train()
    rj.device = torch.device("cpu")
    ids, am, hfe, ham, pos, neg, nmask = cases.joint_inputs(case)
    t0 = time.time()
    user = m(torch.from_numpy(ids), torch.from_numpy(am), torch.from_numpy(hfe), torch.from_numpy(ham))
    loss = rj.InfoNCELoss()(user, torch.from_numpy(pos), torch.from_numpy(neg), torch.from_numpy(nmask))
    loss.backward()
    print(f"reference forward + backward: {time.time() - t0:.1f} s", flush=True)
    import torch.nn.functional as F
    un = F.normalize(user.detach(), p=2, dim=-1)
    pn = F.normalize(torch.from_numpy(pos), p=2, dim=-1)
    ranks = []
    for i in range(un.shape[0]):                        # MRR exactly as :408-419 (per-user loop)
        ne = F.normalize(torch.from_numpy(neg[i]), p=2, dim=-1)
        sims = torch.matmul(un[i], torch.cat([pn[i][None], ne], 0).t())
        si = torch.argsort(sims, descending=True)
        ranks.append((si == 0).nonzero(as_tuple=True)[0].item() + 1)
    res = {"user_embeddings": user.detach().numpy(), "loss": loss.detach().numpy(), "ranks": np.array(ranks, dtype=np.int64),
           "n_pad": (am == 0).sum(axis=1).astype(np.int64)}
    qn = dict(qf.named_parameters())
    res["grad/query_embeddings"] = qn["query_embeddings"].grad.numpy().copy()
    for k in cases.C4_QF_KEYS:
        g = qn[k].grad.numpy()
        res["grad/" + k] = cases.c4_rows(g)
        res["gnorm/" + k] = np.array(np.linalg.norm(g.astype(np.float64)))
    named = dict(base.named_parameters())
    for i in range(qc.num_hidden_layers):
        for pj in cases.LORA_PROJ:
            n = f"layers.{i}.{pj}"
            dW = named[n + ".weight"].grad.numpy().astype(np.float64)
            a, b = gen[n + ".lora_A.weight"].astype(np.float64), gen[n + ".lora_B.weight"].astype(np.float64)
            dA, dB = sc * (b.T @ dW), sc * (dW @ a.T)
            res["gnorm/" + n + ".lora_A.weight"] = np.array(np.linalg.norm(dA))
            res["gnorm/" + n + ".lora_B.weight"] = np.array(np.linalg.norm(dB))
            if n in cases.C4_LORA_FULL:
                res["grad/" + n + ".lora_A.weight"] = dA.astype(np.float32)
                res["grad/" + n + ".lora_B.weight"] = dB.astype(np.float32)
    return res


def main():
    for name, case in cases.C4.items():
        res = gen_joint_c4(case)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **res)
        print(f"{name}: {len(res)} arrays, {os.path.getsize(path) / 1024:.0f} KiB, loss {float(res['loss']):.6f}, ranks {res['ranks'].tolist()}, "
              f"padded {res['n_pad'].tolist()}")


if __name__ == "__main__":
    main()

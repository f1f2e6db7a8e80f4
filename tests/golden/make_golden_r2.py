#!/usr/bin/env python3
"""Round-2 golden vectors, produced by running the REFERENCE's own Python on CPU (build container only; same shim as
make_golden.py, which this script imports):

  qwen_mid.npz        installed transformers Qwen3Model, 4 layers D=1024 hd=128 S=512 B=4, left padding, sdpa:
                      pooled output, strided last_hidden_state and gradient w.r.t. inputs_embeds
  qwen_deep.npz       the same at the full depth of the 0.6B model: 28 layers, S=128, B=2
  user_mid.npz        the reference's default UserQFormer over T=1600 keys, B=2: prediction, MSE loss, gradients
  item_mid.npz        the reference's QFormerForItemRepresentation at C2's architecture (L12 Q32 H768 F14 E1024), B=16:
                      outputs (strided), QFormerLoss, eval metrics, gradients
  joint_mid.npz       the reference's MultiModalQwenEmbedding (item Q-Former H=1024 Q=2 F=14 -> injection -> 2 Qwen3 layers of the
                      0.6B shape, S=512, left padding, hist=10) + InfoNCELoss + MRR ranks: embeddings, loss, ranks, gradients
  use_real.npz        models/user_sequence_encoder.py UserSequenceEncoder._get_item_query_tokens_batch /
                      encode_user_sequence with a real (small) reference item Q-Former behind it
  qwen_lora.npz       LoRA (J4) pinned through the installed Qwen3Model with MERGED weights W + (alpha/r) B A: 2 layers of the
                      0.6B shape, B 2 x S 256: pooled output, gradient w.r.t. inputs_embeds, and dA / dB of every adapter
                      derived from the merged weights' gradients dW' (peft is not installed; its call site is
                      training/train_item_individual_token_joint.py:121-131)
  qwen_lora_big.npz   the same at B 4 x S 512 = 2048 tokens (cases.LORA_BIG): the size at which the product's q|k|v and gate|up launches run
                      on the persistent GEMM with the fused q/k-norm + RoPE / SwiGLU epilogues and the LoRA K tile
  state_dict_shapes.json   key -> shape of the reference modules' state_dict (item default / Q=8 duplicate / C1 / C2,
                      UserQFormer default): the checkpoint-compatibility contract of SURVEY 8(b)

Usage:  python tests/golden/make_golden_r2.py [qwen_mid qwen_deep user_mid item_mid joint_mid qwen_lora qwen_lora_big use_real shapes]
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import numpy as np
import torch

from tests.golden import make_golden as mg          # installs the shim and imports the reference modules
from tests.golden import cases
from tests.golden import data_cases as dc
from oracle import weights as W
from oracle import data_ref as D
from oracle.qformer_ref import QFormerCfg, item_qformer_shapes, user_qformer_shapes

USE_CFG = dict(H=dc.CTX_H, L=2, nh=2, I=256, Q=dc.QI, E=dc.E, seed=61)      # the item Q-Former behind UserSequenceEncoder


def gen_qwen_mid(case):
    qc = cases.qwen_cfg(case)
    x, am = cases.qwen_inputs(case)
    base = mg.build_hf_qwen3(qc, case["seed"] + 1, "sdpa")
    xt = torch.from_numpy(x).requires_grad_(True)
    h = base(inputs_embeds=xt, attention_mask=torch.from_numpy(am), output_hidden_states=True)
    last = h.hidden_states[-1]
    pooled = last.mean(dim=1)
    pooled.pow(2).sum().backward()
    g = xt.grad.numpy()
    return {"sdpa/pooled": pooled.detach().numpy(), "sdpa/last_hidden_state_s": cases.mid_sample(last.detach().numpy()),
            "sdpa/grad_inputs_embeds_s": cases.mid_sample(g),
            "sdpa/grad_inputs_embeds_norm": np.array(np.linalg.norm(g.astype(np.float64)))}


def gen_qwen_lora(case):
    """LoRA pinned to a reference-held implementation: the installed Qwen3Model runs with every projection weight set to
    W + (alpha / r) B A (what peft's LoraLayer computes with dropout off, merged); forward, input gradient and the gradient
    dW' of every merged weight come from transformers + torch autograd, and the adapters' gradients follow from dW' by the
    chain rule of the merge alone: dA = (alpha / r) B^T dW', dB = (alpha / r) dW' A^T."""
    qc = cases.qwen_cfg(case)
    sc = case["lora_alpha"] / case["lora_r"]
    x, am = cases.qwen_inputs(case)
    from oracle.qwen3_ref import qwen3_shapes
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
    for p_ in base.parameters():
        p_.requires_grad_(True)
    xt = torch.from_numpy(x).requires_grad_(True)
    h = base(inputs_embeds=xt, attention_mask=torch.from_numpy(am), output_hidden_states=True)
    pooled = h.hidden_states[-1].mean(dim=1)
    pooled.pow(2).sum().backward()
    g = xt.grad.numpy()
    res = {"pooled": pooled.detach().numpy(), "grad_inputs_embeds_s": cases.mid_sample(g),
           "grad_inputs_embeds_norm": np.array(np.linalg.norm(g.astype(np.float64)))}
    named = dict(base.named_parameters())
    for i in range(qc.num_hidden_layers):
        for pj in cases.LORA_PROJ:
            n = f"layers.{i}.{pj}"
            dW = named[n + ".weight"].grad.numpy().astype(np.float64)
            a, b = gen[n + ".lora_A.weight"].astype(np.float64), gen[n + ".lora_B.weight"].astype(np.float64)
            dA, dB = sc * (b.T @ dW), sc * (dW @ a.T)
            res["gnorm/" + n + ".lora_A.weight"] = np.array(np.linalg.norm(dA))
            res["gnorm/" + n + ".lora_B.weight"] = np.array(np.linalg.norm(dB))
            if n in cases.LORA_FULL:
                res["grad/" + n + ".lora_A.weight"] = dA.astype(np.float32)
                res["grad/" + n + ".lora_B.weight"] = dB.astype(np.float32)
    return res


def gen_user_mid(case):
    return mg.gen_user(case)


def gen_item_mid(case):
    res = mg.gen_item(case)          # the reference's QFormerForItemRepresentation + QFormerLoss + eval metrics
    full = np.asarray(res["reconstructed_fields"])
    res = cases.item_mid_sample(res)
    res["reconstructed_fields_norm"] = np.array(np.linalg.norm(full.astype(np.float64)))
    return res


def gen_joint_mid(case):
    """The reference's MultiModalQwenEmbedding.forward + InfoNCELoss + MRR ranks (train_item_individual_token_joint.py:133-212,
    326-352, 408-419) over the installed Qwen3Model; the sdpa results only (the product's semantics; fixture size)."""
    return {k: v for k, v in mg.gen_joint(case).items() if k.startswith("sdpa/")}


def gen_use_real():
    import models.user_sequence_encoder as ruse
    from models.mwne import TimestampEncoder as RefTime, GeoCoordinateEncoder as RefGeo
    from models.qformer_utils import QFormerForItemRepresentation as RefItem
    c = USE_CFG
    samples, item_dict = dc.item_samples()
    fields = sorted({k for s in samples for k in s if k != "item_id"})
    m = RefItem(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                num_query_tokens=c["Q"], field_embedding_dim=c["E"], num_fields=len(fields), dropout=0.0)
    mg.load_generated(m, item_qformer_shapes(QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2), len(fields)), c["seed"])
    H = c["H"]
    te, ge = RefTime(H), RefGeo(H)
    te.load_state_dict({k: torch.from_numpy(v) for k, v in W.fill_state_dict(D.context_mlp_shapes(H, 9), dc.CTX_SEED).items()})
    ge.load_state_dict({k: torch.from_numpy(v) for k, v in W.fill_state_dict(D.context_mlp_shapes(H, 3), dc.CTX_SEED + 1).items()})
    use = ruse.UserSequenceEncoder.__new__(ruse.UserSequenceEncoder)
    use.device = torch.device("cpu")
    use.item_encoder = dc.FakeItemEncoder()
    use.item_qformer = m.eval()
    use.item_qformer_fields = fields
    use.embedding_dim = H
    use.timestamp_encoder, use.geo_encoder = te, ge
    use.positional_encoder = ruse.PositionalEncoding(d_model=H).eval()          # dropout off for the fixture
    events = [dict(e, item_data=item_dict[e["item_id"]]) for e in dc.user_events()[0]]
    with torch.no_grad():
        toks = use._get_item_query_tokens_batch([e["item_data"] for e in events])
        seq = use.encode_user_sequence(events)
    return {"item_query_tokens": toks.numpy(), "encoded_user_sequence": seq.numpy(),
            "fields": np.array(fields)}


def gen_shapes():
    from models.qformer_utils import QFormerForItemRepresentation as RefItem
    out = {}

    def shapes(m):
        return {k: list(v.shape) for k, v in m.state_dict().items()}
    out["item_default_F14"] = shapes(RefItem(num_fields=14))
    out["item_qformer_model_default_F14"] = shapes(mg.RefItemQFormer(num_fields=14))
    out["item_c1"] = shapes(mg.RefItemQFormer(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024,
                                              num_query_tokens=4, field_embedding_dim=256, num_fields=8))
    out["item_c2"] = shapes(RefItem(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                                    num_query_tokens=32, field_embedding_dim=1024, num_fields=14))
    out["user_default"] = shapes(mg.RefUserQFormer())
    return out


def main():
    only = set(sys.argv[1:])
    want = lambda n: not only or n in only
    for name in ("qwen_mid", "qwen_deep", "user_mid", "item_mid", "joint_mid"):
        if want(name):
            case = cases.MID[name]
            res = {"qwen_mid": gen_qwen_mid, "qwen_deep": gen_qwen_mid, "user_mid": gen_user_mid, "item_mid": gen_item_mid, "joint_mid": gen_joint_mid}[name](case)
            path = os.path.join(HERE, name + ".npz")
            np.savez_compressed(path, **{k: np.asarray(v) for k, v in res.items()})
            print(f"{name}: {len(res)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")
    for lname, lcase in cases.LORA_CASES.items():
        if not want(lname):
            continue
        res = gen_qwen_lora(lcase)
        path = os.path.join(HERE, lname + ".npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in res.items()})
        print(f"{lname}: {len(res)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")
    if want("use_real"):
        res = gen_use_real()
        path = os.path.join(HERE, "use_real.npz")
        np.savez_compressed(path, **res)
        print(f"use_real: {os.path.getsize(path) / 1024:.1f} KiB", {k: v.shape for k, v in res.items()})
    if want("shapes"):
        with open(os.path.join(HERE, "state_dict_shapes.json"), "w") as f:
            json.dump(gen_shapes(), f, indent=0, sort_keys=True)
        print("state_dict_shapes.json written")


if __name__ == "__main__":
    main()

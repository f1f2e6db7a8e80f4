#!/usr/bin/env python3
"""Is the (ill-conditioned) gradient of the Q=2 self-attention query weight systematically worse with the tiny
attention kernels, or is its error one coherent rounding draw per weight seed?  Joint case of the parity tests with
several weight seeds: oracle (fp32, CPU) vs the HIP path with UR_ATTN_TINY=0 / 1."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import qformer_ref as R, qwen3_ref as Q, weights as W
from tests.golden import cases
from tests.parity_utils import rel_err
from tests.test_gpu_joint import _build_joint
from unirec_amd.joint import InfoNCELoss

KEYS = ["qformer.encoder.layer.0.attention.self.query.weight", "qformer.encoder.layer.0.crossattention.self.key.weight", "query_embeddings"]
name = [n for n, c in cases.ALL.items() if c["kind"] == "joint"][0]
for seed in (11, 22, 33, 44, 55, 66):
    case = copy.deepcopy(cases.ALL[name]); case["seed"] = seed
    c = case["cfg"]; qc = cases.qwen_cfg(case)
    ids, am, hfe, ham, pos, neg, nmask = cases.joint_inputs(case)
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    PQ = {k: torch.from_numpy(v).requires_grad_(True) for k, v in W.fill_state_dict(R.item_qformer_shapes(cfg, c["F"]), seed).items()}
    PW = {k: torch.from_numpy(v) for k, v in W.fill_state_dict(Q.qwen3_shapes(qc, lora=False), seed + 1).items()}
    B, hist = case["B"], case["hist"]
    out = R.item_qformer_forward(PQ, cfg, torch.from_numpy(hfe).view(B * hist, c["F"], c["E"]), torch.from_numpy(ham).view(B * hist, c["F"]))
    toks = out["query_outputs"].view(B, hist, c["Q"], c["H"])
    ou = Q.joint_forward(PW, qc, torch.from_numpy(ids), torch.from_numpy(am), toks, case["first_special_id"])
    Q.infonce_loss(ou, torch.from_numpy(pos), torch.from_numpy(neg), torch.from_numpy(nmask)).backward()
    t = lambda a: torch.from_numpy(a).cuda()
    line = f"seed {seed}:"
    for sw in ("0", "1"):
        from unirec_amd import hip
        hip.attn_mode(hip.ATTN_MODE_TINY, 3 if sw == "1" else 0)
        m, qf = _build_joint(case, use_lora=False)
        loss = InfoNCELoss()(m(t(ids), t(am), t(hfe), t(ham)), t(pos), t(neg), t(nmask))
        loss.backward()
        named = dict(qf.named_parameters())
        line += f"  TINY={sw}: " + " ".join(f"{rel_err(cases.trim_like(named[k].grad.float().cpu().numpy()), cases.trim_like(PQ[k].grad.numpy())):.2e}" for k in KEYS)
    print(line, flush=True)

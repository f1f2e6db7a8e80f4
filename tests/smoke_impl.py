"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the oracle.
(item Q-Former -> token injection -> Qwen3+LoRA -> mean pool -> InfoNCE, forward + backward + AdamW)."""
import numpy as np
import torch

from oracle import qformer_ref as R, qwen3_ref as Q, weights as W
from tests.golden import cases
from tests.parity_utils import GRAD_REL, OUT_REL, assert_close, load_generated


def run_smoke():
    from unirec_amd.joint import InfoNCELoss, MultiModalQwenEmbedding, mrr_ranks
    from unirec_amd.optim import FusedAdamW
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    from unirec_amd.qwen3 import Qwen3Config
    dev = "cuda:0"
    case = cases.ALL["joint_left"]
    c, qc = case["cfg"], cases.qwen_cfg(case)
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    qf = QFormerForItemRepresentation(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"],
                                      intermediate_size=c["I"], num_query_tokens=c["Q"], field_embedding_dim=c["E"],
                                      num_fields=c["F"], dropout=0.0)
    qf = load_generated(qf, R.item_qformer_shapes(cfg, c["F"]), case["seed"], dev)
    hc = Qwen3Config(vocab_size=case["first_special_id"], hidden_size=qc.hidden_size, intermediate_size=qc.intermediate_size,
                     num_hidden_layers=qc.num_hidden_layers, num_attention_heads=qc.num_attention_heads,
                     num_key_value_heads=qc.num_key_value_heads, head_dim=qc.head_dim, lora_r=qc.lora_r, lora_alpha=qc.lora_alpha, lora_dropout=0.0)
    m = MultiModalQwenEmbedding(qformer_model=qf, use_lora=True, qwen_config=hc, num_history_items=case["hist"], num_query_tokens_per_item=c["Q"])
    base = {k: torch.from_numpy(v) for k, v in W.fill_state_dict(Q.qwen3_shapes(qc, lora=False), case["seed"] + 1).items()}
    lsh = {k: s for k, s in Q.qwen3_shapes(qc, lora=True).items() if ".lora_" in k}
    lora = {k: torch.from_numpy(v) for k, v in W.fill_state_dict(lsh, case["seed"] + 2).items()}
    m.base_model.load_state_dict({**base, **lora}, strict=False)
    m = m.to(dev).train()
    ids, am, hfe, ham, pos, neg, nmask = cases.joint_inputs(case)
    t = lambda a: torch.from_numpy(a).to(dev)
    user = m(t(ids), t(am), t(hfe), t(ham))
    loss = InfoNCELoss()(user, t(pos), t(neg), t(nmask))
    loss.backward()
    _, rank = mrr_ranks(user, t(pos), t(neg))
    # ---- oracle (CPU, fp32) on the same inputs
    PQ = {k: torch.from_numpy(v).requires_grad_(True) for k, v in W.fill_state_dict(R.item_qformer_shapes(cfg, c["F"]), case["seed"]).items()}
    PW = {**base, **{k: v.clone().requires_grad_(True) for k, v in lora.items()}}
    B, hist = case["B"], case["hist"]
    out = R.item_qformer_forward(PQ, cfg, torch.from_numpy(hfe).view(B * hist, c["F"], c["E"]), torch.from_numpy(ham).view(B * hist, c["F"]))
    ou = Q.joint_forward(PW, qc, torch.from_numpy(ids), torch.from_numpy(am), out["query_outputs"].view(B, hist, c["Q"], c["H"]), case["first_special_id"])
    ol = Q.infonce_loss(ou, torch.from_numpy(pos), torch.from_numpy(neg), torch.from_numpy(nmask))
    ol.backward()
    _, orank = Q.mrr_ranks(ou.detach(), torch.from_numpy(pos), torch.from_numpy(neg))
    assert_close(user, ou.detach().numpy(), OUT_REL, "smoke user_embeddings")
    assert_close(loss, ol.detach().numpy(), OUT_REL, "smoke loss")
    assert rank.cpu().tolist() == orank.tolist()
    assert_close(qf.query_embeddings.grad, PQ["query_embeddings"].grad.numpy(), GRAD_REL * 1.5, "smoke grad/query_embeddings")
    k = "layers.1.mlp.down_proj.lora_B.weight"
    assert_close(dict(m.base_model.named_parameters())[k].grad, PW[k].grad.numpy(), GRAD_REL * 1.5, "smoke grad/" + k)
    opt = FusedAdamW([qf.pack, m.base_model.pack], lr=1e-3)
    before = qf.query_embeddings.detach().clone()
    opt.step()
    assert not torch.equal(before, qf.query_embeddings.detach())
    print("smoke ok: loss", float(loss))

#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python on CPU.

Runs only in the build container (needs /root/reference and the installed
`transformers`); its outputs (tests/golden/*.npz: seeds/config + expected outputs
and selected gradients, never weights, never reference source) are committed and
travel to the GPU box.  Weights come from oracle.weights (a pure function of
(name, shape, seed)) and are pushed into the reference via load_state_dict.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

import numpy as np
import torch

torch.manual_seed(0)
torch.set_num_threads(8)


# ----------------------------------------------------------------------------------------------
# Harness-side compatibility shim; nothing here touches the reference tree.
# ----------------------------------------------------------------------------------------------
def install_shim():
    import transformers
    from transformers import Trainer, TrainingArguments, TrainerCallback  # noqa: F401 (cache "peft unavailable")
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), None)

    st = types.ModuleType("sentence_transformers")
    st.SentenceTransformer = type("SentenceTransformer", (), {})
    sys.modules["sentence_transformers"] = st
    pf = types.ModuleType("peft")
    for n in ("LoraConfig", "get_peft_model", "TaskType", "PeftModel"):
        setattr(pf, n, type(n, (), {}))
    sys.modules["peft"] = pf
    torch.cuda.set_device = lambda *a, **k: None

    sys.path.insert(0, REF)
    import models.qformer as rq

    _orig = mu.PreTrainedModel.init_weights

    def _init_weights_once(self):
        # 1st call (from BertModel.__init__, models/qformer.py:697) -> post_init(), whose own
        # init_weights() call lands in the stock implementation.
        if getattr(self, "_shim_in_post_init", False):
            return _orig(self)
        self._shim_in_post_init = True
        try:
            self.post_init()
        finally:
            self._shim_in_post_init = False
    rq.BertPreTrainedModel.init_weights = _init_weights_once
    rq.BertModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
    return rq


rq = install_shim()
from models.qformer_model import QFormerForItemRepresentation as RefItemQFormer  # noqa: E402
from training.user_qformer_training import UserQFormer as RefUserQFormer  # noqa: E402
import training.train_item_individual_token_joint as rj  # noqa: E402
from training.item_qformer_training import QFormerLoss as RefQFormerLoss  # noqa: E402

from oracle import weights as W  # noqa: E402
from oracle.qformer_ref import QFormerCfg, item_qformer_shapes, user_qformer_shapes  # noqa: E402
from oracle.qwen3_ref import Qwen3Cfg, qwen3_shapes  # noqa: E402
from tests.golden import cases  # noqa: E402


def load_generated(module, shapes, seed):
    sd = module.state_dict()
    gen = W.fill_state_dict(shapes, seed)
    missing = [k for k in shapes if k not in sd]
    assert not missing, missing
    for k, v in gen.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v)
    module.load_state_dict(sd)


def trim(g):
    """Keep fixtures small: big 2-D gradients are stored as their first 32 rows; the full
    Frobenius norm is stored beside them (see tests/golden/cases.py:trim_like)."""
    g = np.asarray(g)
    return cases.trim_like(g)


def grads_of(module, keys, prefix="grad/"):
    named = dict(module.named_parameters())
    out = {}
    for k in keys:
        g = named[k].grad.detach().numpy()
        out[prefix + k] = trim(g)
        out[prefix.replace("grad/", "gnorm/", 1) + k] = np.array(np.linalg.norm(g.astype(np.float64)), dtype=np.float64)
    return out


# ----------------------------------------------------------------------------------------------
def gen_item(case):
    c = case["cfg"]
    m = RefItemQFormer(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"],
                       intermediate_size=c["I"], num_query_tokens=c["Q"], field_embedding_dim=c["E"],
                       num_fields=c["F"], dropout=0.0)
    cfg = QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    load_generated(m, item_qformer_shapes(cfg, c["F"]), case["seed"])
    m.train()  # dropout p=0: train == eval numerically; grads needed
    x, mask = cases.item_inputs(case)
    xt, mt = torch.from_numpy(x), torch.from_numpy(mask)
    out = m(xt, mt)
    # eval metrics exactly as evaluation/evaluate_item_qformer.py:75-92
    import torch.nn.functional as F
    rec = out["reconstructed_fields"]
    unred = F.mse_loss(rec, xt, reduction="none")
    mse = (unred * mt.unsqueeze(-1)).sum() / mt.sum()
    valid = mt.bool()
    cos = torch.sum(F.normalize(xt[valid], p=2, dim=-1) * F.normalize(rec[valid], p=2, dim=-1), dim=-1).sum()
    # training loss exactly as training/item_qformer_training.py:122-127 (pos/neg reps are constants)
    pos, neg = cases.triplet_reps(case)
    loss, rl, cl = RefQFormerLoss()(out, {"field_embeddings": xt}, torch.from_numpy(pos), torch.from_numpy(neg), mt)
    loss.backward()
    res = {"query_outputs": out["query_outputs"], "item_representation": out["item_representation"],
           "reconstructed_fields": rec, "eval_mse": mse, "eval_cos_sum": cos, "loss": loss,
           "recon_loss": rl, "cont_loss": cl}
    res = {k: v.detach().numpy() for k, v in res.items()}
    res.update(grads_of(m, cases.item_grad_keys(c)))
    return res


def gen_user(case):
    c = case["cfg"]
    m = RefUserQFormer(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"],
                       intermediate_size=c["I"], num_query_tokens=c["Q"], input_embedding_dim=c["E"],
                       num_item_tokens_to_predict=c["n_pred"], dropout=0.0)
    cfg = QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 1)
    load_generated(m, user_qformer_shapes(cfg, c["n_pred"]), case["seed"])
    m.train()
    x, mask, tgt = cases.user_inputs(case)
    pred = m(torch.from_numpy(x), torch.from_numpy(mask))
    loss = torch.nn.MSELoss()(pred, torch.from_numpy(tgt))       # training/user_qformer_training.py:193,209
    loss.backward()
    res = {"predicted_item_tokens": pred.detach().numpy(), "loss": loss.detach().numpy()}
    res.update(grads_of(m, cases.user_grad_keys(c)))
    return res


class FakeTokenizer:
    """Only what MultiModalQwenEmbedding.forward touches (:163)."""
    def __init__(self, first_special_id, hist, qi):
        self.map = {f"<|history_item_{i}_query_{j}|>": first_special_id + i * qi + j
                    for i in range(hist) for j in range(qi)}

    def convert_tokens_to_ids(self, name):
        return self.map[name]


def build_hf_qwen3(qc: Qwen3Cfg, seed, attn_impl):
    from transformers import Qwen3Config, Qwen3Model
    hc = Qwen3Config(vocab_size=qc.vocab_size, hidden_size=qc.hidden_size, intermediate_size=qc.intermediate_size,
                     num_hidden_layers=qc.num_hidden_layers, num_attention_heads=qc.num_attention_heads,
                     num_key_value_heads=qc.num_key_value_heads, head_dim=qc.head_dim, rms_norm_eps=qc.rms_norm_eps,
                     rope_parameters={"rope_type": "default", "rope_theta": qc.rope_theta},
                     max_position_embeddings=4096, attention_dropout=0.0, tie_word_embeddings=True,
                     attn_implementation=attn_impl)
    m = Qwen3Model(hc)
    load_generated(m, qwen3_shapes(qc, lora=False), seed)
    return m


def gen_joint(case):
    c, q = case["cfg"], case["qwen"]
    qf = RefItemQFormer(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"],
                        intermediate_size=c["I"], num_query_tokens=c["Q"], field_embedding_dim=c["E"],
                        num_fields=c["F"], dropout=0.0)
    load_generated(qf, item_qformer_shapes(QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2), c["F"]),
                   case["seed"])
    qc = cases.qwen_cfg(case)
    res = {}
    for impl in ("eager", "sdpa"):
        base = build_hf_qwen3(qc, case["seed"] + 1, impl)
        m = rj.MultiModalQwenEmbedding.__new__(rj.MultiModalQwenEmbedding)
        torch.nn.Module.__init__(m)
        m.qformer_model = qf
        m.num_history_items = case["hist"]
        m.num_query_tokens_per_item = c["Q"]
        m.hidden_size = qc.hidden_size
        m.tokenizer = FakeTokenizer(case["first_special_id"], case["hist"], c["Q"])
        m.base_model = base
        m.use_lora = False
        m.train()
        ids, am, hfe, ham, pos, neg, nmask = cases.joint_inputs(case)
        rj.device = torch.device("cpu")
        qf.zero_grad()
        base.zero_grad()
        user = m(torch.from_numpy(ids), torch.from_numpy(am), torch.from_numpy(hfe), torch.from_numpy(ham))
        loss = rj.InfoNCELoss()(user, torch.from_numpy(pos), torch.from_numpy(neg), torch.from_numpy(nmask))
        loss.backward()
        # MRR exactly as :408-419 (per-user loop)
        import torch.nn.functional as F
        un = F.normalize(user.detach(), p=2, dim=-1)
        pn = F.normalize(torch.from_numpy(pos), p=2, dim=-1)
        ranks = []
        for i in range(un.shape[0]):
            ne = F.normalize(torch.from_numpy(neg[i]), p=2, dim=-1)
            sims = torch.matmul(un[i], torch.cat([pn[i][None], ne], 0).t())
            si = torch.argsort(sims, descending=True)
            ranks.append((si == 0).nonzero(as_tuple=True)[0].item() + 1)
        r = {"user_embeddings": user.detach().numpy(), "loss": loss.detach().numpy(),
             "ranks": np.array(ranks, dtype=np.int64),
             "grad/embed_tokens_special_rows": base.embed_tokens.weight.grad[case["first_special_id"]:
                                                                               case["first_special_id"] + case["hist"] * c["Q"]].numpy().copy(),
             "grad/embed_tokens_norm": np.array(base.embed_tokens.weight.grad.norm().item(), dtype=np.float32)}
        r.update(grads_of(qf, cases.item_grad_keys(c, heads=False)))
        r.update(grads_of(base, cases.qwen_grad_keys(), prefix="grad/qwen/"))
        for k, v in r.items():
            res[f"{impl}/{k}"] = v
    return res


def gen_qwen_only(case):
    """Qwen3Model alone on random inputs_embeds (no injection): pins decoder math incl. padding."""
    qc = cases.qwen_cfg(case)
    res = {}
    x, am = cases.qwen_inputs(case)
    for impl in ("eager", "sdpa"):
        base = build_hf_qwen3(qc, case["seed"] + 1, impl)
        xt = torch.from_numpy(x).requires_grad_(True)
        h = base(inputs_embeds=xt, attention_mask=torch.from_numpy(am), output_hidden_states=True)
        last = h.hidden_states[-1]
        assert torch.equal(last, h.last_hidden_state)
        last.mean(dim=1).pow(2).sum().backward()
        res[f"{impl}/last_hidden_state"] = last.detach().numpy()
        res[f"{impl}/grad_inputs_embeds"] = xt.grad.numpy().copy()
    return res


def main():
    os.makedirs(HERE, exist_ok=True)
    only = set(sys.argv[1:])          # optional: regenerate the named cases only
    for name, case in cases.ALL.items():
        if only and name not in only:
            continue
        kind = case["kind"]
        res = {"item": gen_item, "user": gen_user, "joint": gen_joint, "qwen": gen_qwen_only}[kind](case)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in res.items()})
        nan = [k for k, v in res.items() if np.issubdtype(np.asarray(v).dtype, np.floating) and not np.isfinite(v).all()]
        print(f"{name}: {len(res)} arrays, {os.path.getsize(path) / 1024:.1f} KiB, non-finite: {nan}")


if __name__ == "__main__":
    main()

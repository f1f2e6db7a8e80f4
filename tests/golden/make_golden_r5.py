#!/usr/bin/env python3
"""Round-5 golden vectors: the reference's Q-Formers in TRAINING mode (dropout on), produced by running the REFERENCE's own Python
on CPU (build container only; same shim as make_golden.py, which this script imports).

nn.Dropout draws from torch's generator, which the HIP kernels cannot reproduce; the kernels' masks, however, are pure functions of
(seed, element counter) restated in numpy by oracle/dropout_ref.py.  So the masks go the other way: every nn.Dropout MODULE of the
reference model (models/qformer.py:66 embeddings, :135 attention probabilities, :283 BertSelfOutput, :369 BertOutput) gets its
forward replaced by ``x * keep / (1 - p)`` -- nn.Dropout's training-mode semantics -- with `keep` the mask the product draws at
that site.  Everything else (softmax, where the dropout sits, the LayerNorms, heads, losses, autograd) is the reference's code.

  item_c1_train.npz    QFormerForItemRepresentation at C1's configuration, dropout 0.2: outputs, QFormerLoss, gradients
  user_t96_train.npz   UserQFormer (64 queries, 96 ragged keys), dropout 0.2: prediction, MSE loss, gradients

Usage:  python tests/golden/make_golden_r5.py [item_c1_train user_t96_train]
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import numpy as np
import torch

from tests.golden import make_golden as mg          # installs the shim and imports the reference modules
from tests.golden import cases
from oracle.qformer_ref import QFormerCfg, item_qformer_shapes, user_qformer_shapes


def feed_masks(model, masks, p):
    """Replace the forward of every nn.Dropout on the query-only path by the fed mask of its site; returns the sites that ran."""
    used = []

    def site_of(name):
        # module path inside the wrapper -> mask key
        if name == "qformer.embeddings.dropout":
            return "emb"
        parts = name.split(".")
        if parts[:3] != ["qformer", "encoder", "layer"]:
            return None
        i, rest = parts[3], ".".join(parts[4:])
        return {"attention.self.dropout": f"{i}.self.probs", "attention.output.dropout": f"{i}.self.out",
                "crossattention.self.dropout": f"{i}.cross.probs", "crossattention.output.dropout": f"{i}.cross.out",
                "output_query.dropout": f"{i}.ffn.out"}.get(rest)

    for name, mod in model.named_modules():
        if not isinstance(mod, torch.nn.Dropout):
            continue
        key = site_of(name)

        def fwd(x, key=key, name=name):
            assert key is not None, f"dropout module off the query-only path ran: {name}"
            keep = torch.from_numpy(masks[key]).to(x.dtype)
            assert keep.shape == x.shape, (name, tuple(keep.shape), tuple(x.shape))
            used.append(key)
            return x * keep / (1.0 - p)
        mod.forward = fwd
    return used


def gen_item_train(case):
    c, p = case["cfg"], case["p"]
    m = mg.RefItemQFormer(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                          num_query_tokens=c["Q"], field_embedding_dim=c["E"], num_fields=c["F"], dropout=p)
    mg.load_generated(m, item_qformer_shapes(QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2), c["F"]), case["seed"])
    m.train()
    masks = cases.train_masks(case)
    used = feed_masks(m, masks, p)
    x, mask = cases.item_inputs(case)
    xt, mt = torch.from_numpy(x), torch.from_numpy(mask)
    out = m(xt, mt)
    assert sorted(used) == sorted(masks), (sorted(used), sorted(masks))
    pos, neg = cases.triplet_reps(case)
    loss, rl, cl = mg.RefQFormerLoss()(out, {"field_embeddings": xt}, torch.from_numpy(pos), torch.from_numpy(neg), mt)
    loss.backward()
    res = {k: out[k].detach().numpy() for k in ("query_outputs", "item_representation", "reconstructed_fields")}
    res.update(loss=loss.detach().numpy(), recon_loss=rl.detach().numpy(), cont_loss=cl.detach().numpy(),
               keep_fraction=np.array(np.mean([v.mean() for v in masks.values()]), dtype=np.float64))
    res.update(mg.grads_of(m, cases.item_grad_keys(c)))
    return res


def gen_user_train(case):
    c, p = case["cfg"], case["p"]
    m = mg.RefUserQFormer(hidden_size=c["H"], num_hidden_layers=c["L"], num_attention_heads=c["nh"], intermediate_size=c["I"],
                          num_query_tokens=c["Q"], input_embedding_dim=c["E"], num_item_tokens_to_predict=c["n_pred"], dropout=p)
    mg.load_generated(m, user_qformer_shapes(QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 1), c["n_pred"]), case["seed"])
    m.train()
    masks = cases.train_masks(case)
    used = feed_masks(m, masks, p)
    x, mask, tgt = cases.user_inputs(case)
    pred = m(torch.from_numpy(x), torch.from_numpy(mask))
    assert sorted(used) == sorted(masks), (sorted(used), sorted(masks))
    loss = torch.nn.MSELoss()(pred, torch.from_numpy(tgt))
    loss.backward()
    res = {"predicted_item_tokens": pred.detach().numpy(), "loss": loss.detach().numpy(),
           "keep_fraction": np.array(np.mean([v.mean() for v in masks.values()]), dtype=np.float64)}
    res.update(mg.grads_of(m, cases.user_grad_keys(c)))
    return res


def main():
    only = set(sys.argv[1:])
    for name, case in cases.TRAIN.items():
        if only and name not in only:
            continue
        res = {"item_train": gen_item_train, "user_train": gen_user_train}[case["kind"]](case)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in res.items()})
        nan = [k for k, v in res.items() if np.issubdtype(np.asarray(v).dtype, np.floating) and not np.isfinite(v).all()]
        print(f"{name}: {len(res)} arrays, {os.path.getsize(path) / 1024:.1f} KiB, keep fraction {float(res['keep_fraction']):.4f}, non-finite: {nan}")


if __name__ == "__main__":
    main()

"""Fused AdamW over flat parameter packs (one HIP kernel launch per contiguous run of live tensors; one per pack when
every tensor received a gradient).

Semantics = torch.optim.AdamW (decoupled weight decay), the optimizer of all three reference loops
(training/item_qformer_training.py:108, training/user_qformer_training.py:196, HF Trainer default at
train_item_individual_token_joint.py:755-773), including what it does with parameters a backward did not reach:
``grad is None`` => the parameter is skipped entirely -- no weight decay, no moment update, its own step count does not
advance (the item Q-Former's heads and ``UserQFormer.prediction_head`` in the joint step).  Which tensors are live is
what the backward published since the last ``zero_grad()`` (packing.ParamPack.live).  Gradients are OVERWRITTEN, not
accumulated, by every backward: call ``zero_grad()`` once per step as the reference loops do, so a tensor touched in one
step and untouched in the next is not re-stepped with a stale gradient.  ``grad_scale`` folds the 1/world_size of a
summed all-reduce into the update.  ``state_dict`` / ``load_state_dict`` carry the moments and step counts (resume).
"""
import torch

from . import hip


class FusedAdamW:
    def __init__(self, packs, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.packs = [p for p in packs if p is not None]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.state = [(torch.zeros_like(p.master), torch.zeros_like(p.master)) for p in self.packs]
        self.steps = [{n: 0 for n in p.names} for p in self.packs]          # per-tensor step counts (bias correction)

    def zero_grad(self, set_to_none=True):
        for p in self.packs:
            p.clear_grads(set_to_none)

    def step(self, grad_scale=1.0):
        self.step_count += 1
        for pack, (m, v), steps in zip(self.packs, self.state, self.steps):
            # torch.optim.AdamW's rule whoever reset the gradients: a tensor whose .grad is None now (model.zero_grad(),
            # HF Trainer, a torch optimizer's zero_grad -- none of which clears pack.live) is not stepped
            stale = [n for n in pack.live if pack.params[n].grad is None]
            for n in stale:
                pack.live.discard(n)
            if not pack.live:
                continue
            for n in pack.live:
                steps[n] += 1
            for lo, hi, t in pack.live_ranges(key=steps.__getitem__):
                hip.adamw_step(pack.master[lo:hi], pack.grad[lo:hi], m[lo:hi], v[lo:hi], self.lr, self.betas[0], self.betas[1],
                               self.eps, self.weight_decay, t, grad_scale)
            pack.mark_dirty()

    # ---- resume -----------------------------------------------------------------------------------------------------
    def state_dict(self):
        return {"step_count": self.step_count, "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                "packs": [{"names": list(p.names), "offsets": dict(p.offsets), "steps": dict(s), "exp_avg": m.detach().clone(),
                           "exp_avg_sq": v.detach().clone()} for p, (m, v), s in zip(self.packs, self.state, self.steps)]}

    def load_state_dict(self, sd):
        if len(sd["packs"]) != len(self.packs):
            raise ValueError(f"optimizer state holds {len(sd['packs'])} packs, this optimizer {len(self.packs)}")
        for p, (m, v), s, rec in zip(self.packs, self.state, self.steps, sd["packs"]):
            if list(rec["names"]) == list(p.names) and dict(rec["offsets"]) == dict(p.offsets):
                m.copy_(rec["exp_avg"].to(m.device))
                v.copy_(rec["exp_avg_sq"].to(v.device))
            elif set(rec["names"]) == set(p.names):
                # same tensors, another order (a pack layout change between versions, e.g. the hoisted cross-attention K|V weights):
                # the moments move tensor by tensor, recorded offsets -> current offsets
                rm, rv, roff = rec["exp_avg"].to(m.device), rec["exp_avg_sq"].to(v.device), dict(rec["offsets"])
                # every recorded tensor's extent = the gap to the next recorded offset (or the buffer's end) must be the 8-padded size
                # of the CURRENT tensor: the same names at another shape (a different LoRA rank, hidden size) would otherwise load
                # moments that overlap the neighbours
                order = sorted(roff.values()) + [rm.numel()]
                extent = {o: order[k + 1] - o for k, o in enumerate(order[:-1])}
                for n in p.names:
                    num = p.params[n].numel()
                    if extent[roff[n]] != (num + 7) // 8 * 8:
                        raise ValueError(f"optimizer state: {n} was recorded with {extent[roff[n]]} (padded) elements, the pack holds {num}")
                for n in p.names:
                    lo, num = p.offsets[n], p.params[n].numel()
                    m[lo:lo + num].copy_(rm[roff[n]:roff[n] + num])
                    v[lo:lo + num].copy_(rv[roff[n]:roff[n] + num])
            else:
                raise ValueError("optimizer state does not match the parameter pack (different tensor names)")
            s.update(rec["steps"])
        self.step_count = int(sd["step_count"])
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]

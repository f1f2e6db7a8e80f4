"""Fused AdamW over flat parameter packs (one HIP kernel launch per pack per step).

Semantics = torch.optim.AdamW (decoupled weight decay), the optimizer of all three reference loops
(training/item_qformer_training.py:108, training/user_qformer_training.py:196, HF Trainer default at
train_item_individual_token_joint.py:755-773).  ``grad_scale`` folds the 1/world_size of a summed
all-reduce into the update.
"""
import torch

from . import hip


class FusedAdamW:
    def __init__(self, packs, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.packs = [p for p in packs if p is not None]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.state = [(torch.zeros_like(p.master), torch.zeros_like(p.master)) for p in self.packs]

    def zero_grad(self, set_to_none=True):
        # gradients are overwritten by every backward (packing.ParamPack.publish_grads); nothing to clear
        return None

    def step(self, grad_scale=1.0):
        self.step_count += 1
        for pack, (m, v) in zip(self.packs, self.state):
            hip.adamw_step(pack.master, pack.grad, m, v, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                           self.step_count, grad_scale)
            pack.mark_dirty()

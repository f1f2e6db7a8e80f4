"""HIP versions of the reference's training losses (callers may keep using their torch losses on the
fp32 outputs; these are the fused fast path used by bench.py).

``QFormerLoss`` -- training/item_qformer_training.py:41-56: masked reconstruction MSE divided by the
number of valid FIELDS + contrastive_weight * TripletMarginLoss(margin).  ``mse_loss`` --
training/user_qformer_training.py:193,209.
"""
import torch
import torch.nn as nn

from . import hip

F32 = torch.float32


class _QFormerLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rec, item_rep, x, mask, pos, neg, recon_w, cont_w, margin):
        sums = hip.recon_stats(rec, x, mask)
        tl, d_item = hip.triplet_margin(item_rep, pos, neg, margin, cont_w, need_grad=True)
        ctx.saved = (rec, x, mask, sums, d_item, recon_w)
        # scalar combine on device (3 floats): plumbing, not compute
        recon = sums[0] / sums[1]
        return recon_w * recon + cont_w * tl[0], recon, tl[0]

    @staticmethod
    def backward(ctx, g, g_recon, g_cont):
        rec, x, mask, sums, d_item, recon_w = ctx.saved
        d_rec = hip.recon_grad(rec, x, mask, sums, recon_w)
        return d_rec * g, d_item * g, None, None, None, None, None, None, None


class QFormerLoss(nn.Module):
    def __init__(self, reconstruction_weight=1.0, contrastive_weight=0.5, margin=0.5):
        super().__init__()
        self.recon_w, self.cont_w, self.margin = reconstruction_weight, contrastive_weight, margin

    def forward(self, model_output, input_embeddings, pos_rep, neg_rep, attention_mask):
        x = input_embeddings["field_embeddings"].contiguous().to(F32)
        return _QFormerLossFn.apply(model_output["reconstructed_fields"].contiguous(), model_output["item_representation"].contiguous(),
                                    x, attention_mask.contiguous().to(F32), pos_rep.detach().contiguous().to(F32),
                                    neg_rep.detach().contiguous().to(F32), float(self.recon_w), float(self.cont_w), float(self.margin))


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        loss, da = hip.mse_loss(a, b, 1.0, need_grad=True)
        ctx.da = da
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        return ctx.da * g, None


def mse_loss(pred, target):
    return _MSEFn.apply(pred.contiguous().to(F32), target.contiguous().to(F32))

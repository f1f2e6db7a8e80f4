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
    def forward(ctx, rec, item_rep, x, mask, pos, neg, recon_w, cont_w, margin, data_parallel):
        sums = hip.recon_stats(rec, x, mask)
        recon_log = None
        if data_parallel:
            # the masked MSE divides by the GLOBAL number of valid fields.  sums[1] = sum(mask) all-reduced
            # and divided by the world size: the local loss is then world * (local numerator / global denominator), so
            # the mean over ranks -- what the sum all-reduce of gradients with its 1/world realises -- is the
            # single-process loss over the global batch.
            from . import dp
            w = dp.world_size()
            if w > 1:
                # ONE collective over (numerator, denominator): every rank must call the loss the same number of times --
                # QFormerLoss.forward only takes this branch in training mode with gradients enabled, so rank-asymmetric
                # validation passes (rank 0 only, unequal batch counts) cannot hang on it
                nd = dp.allreduce_sum_(sums[0:2].clone())
                recon_log = nd[0] / nd[1]                      # the GLOBAL reconstruction loss (what is logged)
                sums = torch.cat([sums[0:1], nd[1:2] / w, sums[2:]])
        tl, d_item = hip.triplet_margin(item_rep, pos, neg, margin, cont_w, need_grad=True)
        ctx.saved = (rec, x, mask, sums, d_item, recon_w)
        # scalar combine on device (3 floats): plumbing, not compute
        recon = sums[0] / sums[1]
        return recon_w * recon + cont_w * tl[0], (recon if recon_log is None else recon_log), tl[0]

    @staticmethod
    def backward(ctx, g, g_recon, g_cont):
        rec, x, mask, sums, d_item, recon_w = ctx.saved
        d_rec = hip.recon_grad(rec, x, mask, sums, recon_w)
        return d_rec * g, d_item * g, None, None, None, None, None, None, None, None


class QFormerLoss(nn.Module):
    def __init__(self, reconstruction_weight=1.0, contrastive_weight=0.5, margin=0.5, data_parallel=False):
        """data_parallel=True (new; the reference is single-process): under an initialised torch.distributed group the
        reconstruction term divides by the all-reduced sum of the mask, so N ranks on N shards reproduce the
        single-process loss and gradient over the global batch.  The collective runs only in training mode with gradients
        enabled (every rank must then call the loss the same number of times per step); under eval() / no_grad the loss is
        the local one.  The second return value is the GLOBAL reconstruction loss when the collective ran."""
        super().__init__()
        self.recon_w, self.cont_w, self.margin = reconstruction_weight, contrastive_weight, margin
        self.data_parallel = bool(data_parallel)

    def forward(self, model_output, input_embeddings, pos_rep, neg_rep, attention_mask):
        x = input_embeddings["field_embeddings"].contiguous().to(F32)
        # the all-reduce of the mask sum belongs to the training step only (see _QFormerLossFn.forward)
        dpar = self.data_parallel and self.training and torch.is_grad_enabled()
        return _QFormerLossFn.apply(model_output["reconstructed_fields"].contiguous(), model_output["item_representation"].contiguous(),
                                    x, attention_mask.contiguous().to(F32), pos_rep.detach().contiguous().to(F32),
                                    neg_rep.detach().contiguous().to(F32), float(self.recon_w), float(self.cont_w), float(self.margin), dpar)


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

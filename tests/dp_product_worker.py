"""One rank of the 2-rank PRODUCT-path data-parallel test (tests/test_gpu_dp_product.py starts two of these as child
processes, both on cuda:0, backend gloo -- RCCL needs one device per rank; the code path above the backend is the one
bench.py runs: HIP forward/backward, bucket hooks, sum all-reduce of the flat gradient pack, fused AdamW with 1/world).
Usage: python tests/dp_product_worker.py <outdir> <global_batch> [dropout] [joint]   (RANK / WORLD_SIZE / MASTER_* from the env)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def make_global(Bg, F, E, seed=11):
    g = torch.Generator().manual_seed(seed)

    def fields():
        x = torch.randn(Bg, F, E, generator=g)
        x = x / x.norm(dim=-1, keepdim=True)
        mk = (torch.rand(Bg, F, generator=g) < 0.7).long()
        mk[:, 0] = 1
        return x * mk[..., None], mk
    return fields(), fields(), fields()


def build(dropout):
    from unirec_amd.qformer_utils import QFormerForItemRepresentation
    torch.manual_seed(5)
    return QFormerForItemRepresentation(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024,
                                        num_query_tokens=4, field_embedding_dim=256, num_fields=8, dropout=dropout)


def run(outdir, Bg, dropout):
    from unirec_amd import dp
    from unirec_amd.losses import QFormerLoss
    from unirec_amd.optim import FusedAdamW
    rank, world, _ = dp.init_from_env()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    m = build(dropout).to(dev).train()
    dp.set_dp_rank(rank, m)          # same seeds on every rank; dropout counters offset by rank * local batch (global sample index)
    (xa, ma), (xp, mp_), (xn, mn) = make_global(Bg, 8, 256)
    lo, hi = dp.shard_range(Bg, rank, world)
    xa, ma, xp, mp_, xn, mn = [t[lo:hi].to(dev) for t in (xa, ma, xp, mp_, xn, mn)]
    pack = m._ensure_pack(dev)
    opt = FusedAdamW([pack], lr=1e-3, weight_decay=0.01)
    pre = [f"qformer.encoder.layer.{i}." for i in range(2)]
    bounds = dp.layer_boundaries(pack, pre, 1)
    bk = dp.GradBuckets(pack.grad, bounds, wire_dtype=torch.bfloat16 if os.environ.get("UNIREC_TEST_WIRE") == "bf16" else None)
    m.qformer.grad_ready_hook = dp.bucket_hook(pack, bk, pre, 1)      # layers, the hoisted K|V bucket (-2), the query table (-1)
    assert m.qformer.grad_ready_hook.hoisted_bucket == 1
    loss_fn = QFormerLoss(data_parallel=True)
    out = m(xa, ma)
    with torch.no_grad():
        pr, nr = m(xp, mp_)["item_representation"], m(xn, mn)["item_representation"]
    loss, recon, _ = loss_fn(out, {"field_embeddings": xa}, pr, nr, ma)
    loss.backward()
    bk.ready(bk.n - 1)            # heads: written last by the loss-side backward nodes, first in the autograd order
    bk.wait()
    torch.cuda.synchronize()
    grad = (pack.grad / world).cpu()
    opt.step(grad_scale=1.0 / world)
    torch.cuda.synchronize()
    torch.save({"grad": grad, "master": pack.master.cpu(), "loss": float(loss), "seed": int(m.qformer.seed), "n": hi - lo,
                "enabled": bk.enabled, "world": world}, os.path.join(outdir, f"rank{rank}.pt"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def run_joint(outdir, Bg, dropout):
    """The JOINT step (item Q-Former -> injection -> Qwen3 + LoRA -> InfoNCE) on this rank's shard of a global batch, with
    hidden / attention dropout in the Q-Former and LoRA dropout in the decoder: the reduced gradients of both packs."""
    import numpy as np
    from tests.golden import cases
    from tests.test_gpu_joint import _build_joint
    from unirec_amd import dp
    from unirec_amd.joint import InfoNCELoss
    rank, world, _ = dp.init_from_env()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    case = dict(cases.ALL["joint_left"], B=Bg, drop_one_special=False)
    m, qf = _build_joint(case, use_lora=True, lora_seed=case["seed"] + 2)
    bm = m.base_model
    bm.config.lora_dropout = dropout
    qf.qformer.config.hidden_dropout_prob = dropout
    qf.qformer.config.attention_probs_dropout_prob = dropout
    dp.set_dp_rank(rank, m)
    ids, am, hfe, ham, pos, neg, nmask = cases.joint_inputs(case)
    lo, hi = dp.shard_range(Bg, rank, world)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(dev)
    user = m(t(ids), t(am), t(hfe), t(ham))
    loss = InfoNCELoss()(user, t(pos), t(neg), t(nmask))
    loss.backward()
    grads = {}
    for name, pack in (("lora", bm._ensure_pack(dev)), ("qformer", qf._ensure_pack(dev))):
        bk = dp.GradBuckets(pack.grad, [0, pack.numel])
        bk.ready_all(); bk.wait()
        torch.cuda.synchronize()
        grads[name] = (pack.grad / world).cpu()
    torch.save({"grads": grads, "loss": float(loss), "n": hi - lo, "world": world, "lora_seed": int(bm.lora_seed), "seed": int(qf.qformer.seed)},
               os.path.join(outdir, f"rank{rank}.pt"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[4] == "joint":
        run_joint(sys.argv[1], int(sys.argv[2]), float(sys.argv[3]))
    else:
        run(sys.argv[1], int(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)

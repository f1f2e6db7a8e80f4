"""CPU: the N>1 data-parallel path with world_size 2 over gloo:
sharding, bucket boundaries, bucketed async all-reduce of the flat gradient pack, and the algebra it
relies on (sum of per-shard gradients of mean-reduced losses == world * full-batch gradient)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import qformer_ref as R
from oracle import weights as W
from unirec_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakePack:
    """Host-side stand-in with ParamPack's layout fields (no HIP needed for the bucket logic)."""

    def __init__(self, named_shapes):
        self.names, self.offsets, self.params = [], {}, {}
        off = 0
        for n, shp in named_shapes:
            self.names.append(n)
            self.offsets[n] = off
            self.params[n] = torch.zeros(shp)
            off += (self.params[n].numel() + 7) // 8 * 8
        self.numel = off
        self.grad = torch.zeros(off)


def _pack():
    shapes = [("query_embeddings", (1, 4, 16))]
    for i in range(4):
        shapes += [(f"qformer.encoder.layer.{i}.a.weight", (16, 16)), (f"qformer.encoder.layer.{i}.a.bias", (16,))]
    shapes += [("head.weight", (8, 16))]
    return _FakePack(shapes)


def test_bucket_boundaries_cover_the_pack_in_layer_groups():
    p = _pack()
    b = dp.layer_boundaries(p, [f"qformer.encoder.layer.{i}." for i in range(4)], 2)
    assert b[0] == 0 and b[-1] == p.numel and b == sorted(set(b))
    assert p.offsets["qformer.encoder.layer.0.a.weight"] in b and p.offsets["qformer.encoder.layer.2.a.weight"] in b
    assert p.offsets["qformer.encoder.layer.1.a.weight"] not in b
    assert p.offsets["head.weight"] in b            # untouched heads sit in their own trailing bucket


def test_shard_range_partitions_the_global_batch():
    got = [dp.shard_range(64 * 8, r, 8) for r in range(8)]
    assert got[0] == (0, 64) and got[-1] == (448, 512)
    assert all(got[i][1] == got[i + 1][0] for i in range(7))


def _worker(rank, world, port, tmp):
    os.environ.update(GLOO_SOCKET_IFNAME="lo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.set_num_threads(1)
    # a tiny item Q-Former on this rank's shard of a fixed global batch (oracle = host arithmetic)
    cfg = R.QFormerCfg(64, 2, 1, 128, 4, 32, 2)
    shapes = R.item_qformer_shapes(cfg, 5)
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in W.fill_state_dict(shapes, 3).items()}
    Bg = 8
    x = torch.from_numpy(W.normal("x", (Bg, 5, 32), 1, std=1.0))
    lo, hi = dp.shard_range(Bg, rank, world)
    out = R.item_qformer_forward(P, cfg, x[lo:hi], None)
    loss = out["query_outputs"].pow(2).mean()          # per-sample independent loss, mean over the local shard
    loss.backward()
    names = list(shapes.keys())
    pack = _FakePack([(n, shapes[n]) for n in names])
    for n in names:
        o = pack.offsets[n]
        if P[n].grad is not None:                       # heads are unused by this loss: stay zero
            pack.grad[o:o + P[n].numel()] = P[n].grad.reshape(-1)
    bounds = dp.layer_boundaries(pack, [f"qformer.encoder.layer.{i}." for i in range(2)], 1)
    bk = dp.GradBuckets(pack.grad, bounds)
    assert bk.enabled
    for i in reversed(range(bk.n)):                      # backward-completion order
        bk.ready(i)
    bk.wait()
    torch.save(pack.grad / world, os.path.join(tmp, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_matches_unsharded_gradient(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    assert torch.equal(g0, g1), "ranks must hold bitwise-identical reduced gradients"
    # unsharded reference
    cfg = R.QFormerCfg(64, 2, 1, 128, 4, 32, 2)
    shapes = R.item_qformer_shapes(cfg, 5)
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in W.fill_state_dict(shapes, 3).items()}
    x = torch.from_numpy(W.normal("x", (8, 5, 32), 1, std=1.0))
    R.item_qformer_forward(P, cfg, x, None)["query_outputs"].pow(2).mean().backward()
    pack = _FakePack([(n, shapes[n]) for n in shapes])
    for n in shapes:
        o = pack.offsets[n]
        want = torch.zeros(P[n].numel()) if P[n].grad is None else P[n].grad.reshape(-1)
        assert torch.allclose(g0[o:o + P[n].numel()], want, rtol=1e-4, atol=1e-7), n


def test_shard_range_keeps_the_remainder():
    got = [dp.shard_range(10, r, 4) for r in range(4)]
    assert got == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_set_sample_offset_overrides_the_rank_rule_and_resets():
    """dp.set_sample_offset: micro-batch k of mb samples on rank r (B samples per rank) starts at r * B + k * mb; None returns
    to dp_rank * B.  Every rank keeps the SAME seeds (there is no per-rank seed any more)."""
    class Leaf(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dp_rank, self.sample_offset, self.seed = 0, None, 0x5EED

    class Top(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = Leaf(), Leaf()
    t, plain = Top(), Leaf()
    dp.set_dp_rank(3, t, plain)
    dp.set_sample_offset(3 * 64 + 1 * 32, t, plain, None)
    assert [m.sample_offset for m in (t.a, t.b, plain)] == [224, 224, 224] and t.a.seed == 0x5EED and t.a.dp_rank == 3
    dp.set_sample_offset(None, t, plain)
    assert [m.sample_offset for m in (t.a, t.b, plain)] == [None, None, None]
    assert not hasattr(dp, "set_rank_seeds") and not hasattr(dp, "rank_seed")


def test_set_dp_rank_reaches_every_dropout_owner():
    """dp.set_dp_rank: same seeds on every rank, `.dp_rank` on every module (or plain object) that owns dropout counters."""
    class Leaf(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dp_rank, self.sample_offset, self.seed = 0, None, 0x5EED

    class Top(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = Leaf(), torch.nn.Sequential(Leaf())

    class Plain:
        dp_rank = 0
    t, p = Top(), Plain()
    dp.set_dp_rank(3, t, p, None)
    assert t.a.dp_rank == 3 and t.b[0].dp_rank == 3 and p.dp_rank == 3 and t.a.seed == 0x5EED and t.a.sample_offset is None


def test_bench_gpus_flag_must_match_the_world(monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class A:
        gpus = 4
    monkeypatch.delenv("UNIREC_DP_FORCE", raising=False)
    with pytest.raises(SystemExit):
        bench._check_world(A, 1)
    bench._check_world(A, 4)


def test_micro_batch_accumulation_adds_the_stash_before_the_bucket_leaves():
    """Gradient accumulation inside one optimizer step: backward OVERWRITES the flat buffer, non-final micro-batches stash
    their buckets, the final one folds the stash back in bucket by bucket."""
    flat = torch.zeros(24)
    bk = dp.GradBuckets(flat, [0, 8, 16, 24])
    bk.begin_micro_batch(last=False)
    flat.copy_(torch.arange(24.0))
    for i in reversed(range(bk.n)):
        bk.ready(i)
    bk.begin_micro_batch(last=True)
    flat.copy_(torch.ones(24))                   # the second backward overwrites
    bk.ready(2)
    assert torch.equal(flat[16:], torch.arange(16.0, 24.0) + 1) and torch.equal(flat[:16], torch.ones(16))
    bk.ready(1); bk.ready(0); bk.wait()
    assert torch.equal(flat, torch.arange(24.0) + 1)
    assert float(bk.stash.abs().sum()) == 0.0    # ready for the next step


def test_hoisted_cross_attention_kv_gets_its_own_bucket_and_hook():
    """The Q-Formers keep the cross-attention K | V weights of all layers ahead of layer 0; their gradients are final BEFORE the
    query-table reduction that closes the backward, so they leave from their own hook (signal -2) in their own bucket and the
    exposed tail is the small leading bucket."""
    shapes = [("query_embeddings", (1, 4, 16)), ("qformer.embeddings.LayerNorm.weight", (16,)), ("qformer.embeddings.LayerNorm.bias", (16,))]
    shapes += [(f"qformer.encoder.layer.{i}.crossattention.self.{n}.weight", (16, 16)) for i in (0, 2) for n in ("key", "value")]
    shapes += [(f"qformer.encoder.layer.{i}.crossattention.self.{n}.bias", (16,)) for i in (0, 2) for n in ("key", "value")]
    for i in range(4):
        shapes += [(f"qformer.encoder.layer.{i}.attention.self.query.weight", (16, 16)), (f"qformer.encoder.layer.{i}.attention.self.query.bias", (16,))]
    shapes += [("head.weight", (8, 16))]
    p = _FakePack(shapes)
    pre = [f"qformer.encoder.layer.{i}." for i in range(4)]
    b = dp.layer_boundaries(p, pre, 2)
    kv0 = p.offsets["qformer.encoder.layer.0.crossattention.self.key.weight"]
    assert b[:4] == [0, kv0, p.offsets["qformer.encoder.layer.0.attention.self.query.weight"], p.offsets["qformer.encoder.layer.2.attention.self.query.weight"]]

    class Rec:
        bounds = b
        sent = []

        def ready(self, i):
            self.sent.append(i)
    rec = Rec()
    hook = dp.bucket_hook(p, rec, pre, 2)
    assert hook.hoisted_bucket == 1
    for sig in (3, 2, 1, 0, -2, -1):                 # the backward's order
        hook(sig)
    assert rec.sent == [3, 2, 1, 0]                  # layers 3-2, layers 1-0, K | V, query table: every bucket exactly once


def _wire_worker(rank, world, port, tmp):
    os.environ.update(GLOO_SOCKET_IFNAME="lo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dp.init_from_env(backend="gloo")
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(4096, generator=g)
    mine = flat.clone()
    bk = dp.GradBuckets(flat, [0, 1000, 1000, 3000, 4096], wire_dtype=torch.bfloat16)
    for i in reversed(range(bk.n)):
        bk.ready(i)
    bk.wait_bucket(3)
    bk.wait()
    assert not bk.unpack and flat.dtype == torch.float32
    torch.save({"mine": mine, "sum": flat.clone()}, os.path.join(tmp, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_wire_buckets_under_two_gloo_ranks(tmp_path):
    """wire_dtype = bfloat16: every rank ends with the SAME f32 buffer = the bf16 sum of the ranks' bf16-rounded buckets."""
    world, port = 2, _free_port()
    mp.spawn(_wire_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    assert torch.equal(a["sum"], b["sum"])
    want = (a["mine"].bfloat16() + b["mine"].bfloat16()).float()
    assert torch.equal(a["sum"], want)
    exact = a["mine"] + b["mine"]
    assert float((a["sum"] - exact).norm() / exact.norm()) < 8e-3


def test_lora_shaped_pack_sends_every_bucket_exactly_once():
    """The LoRA pack has NOTHING ahead of `layers.0.` (offset 0) and Qwen3LoRAModel.backward fires layer signals only (no -1): the
    lead bucket must leave with layer 0 (a 28-layer pack in groups of 7 used to send 3 of its 4 buckets)."""
    shapes = []
    for i in range(28):
        shapes += [(f"layers.{i}.q.lora_A", (16, 64)), (f"layers.{i}.q.lora_B", (64, 16))]
    p = _FakePack(shapes)
    pre = [f"layers.{i}." for i in range(28)]
    for grp in (7, 1, 28, 5):
        b = dp.layer_boundaries(p, pre, grp)

        class Rec:
            bounds = b

            def __init__(self):
                self.sent = []

            def ready(self, i):
                self.sent.append(i)
        rec = Rec()
        hook = dp.bucket_hook(p, rec, pre, grp)
        assert hook.lead_by_layer
        for sig in reversed(range(28)):              # Qwen3LoRAModel.backward: layers 27 .. 0, nothing else
            hook(sig)
        n = len(b) - 1
        assert rec.sent == list(reversed(range(n))), (grp, rec.sent)
        hook(-1)                                      # a stray closing signal must not send the lead bucket twice
        assert rec.sent == list(reversed(range(n)))


def _world4_worker(rank, world, port, tmp):
    os.environ.update(GLOO_SOCKET_IFNAME="lo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dp.init_from_env(backend="gloo")
    torch.set_num_threads(1)
    # a Q-Former-shaped pack (query table | hoisted K|V | 4 layers | head) and a LoRA-shaped one, both driven through bucket_hook in the
    # backward's signal order; f32 wire for the LoRA pack, bf16 wire for the Q-Former pack (bench.py --comm-bf16)
    qshapes = [("query_embeddings", (1, 4, 16)), ("qformer.embeddings.LayerNorm.weight", (16,))]
    qshapes += [(f"qformer.encoder.layer.{i}.crossattention.self.{n}.weight", (16, 16)) for i in (0, 2) for n in ("key", "value")]
    for i in range(4):
        qshapes += [(f"qformer.encoder.layer.{i}.attention.self.query.weight", (16, 16))]
    qshapes += [("head.weight", (8, 16))]
    lshapes = [(f"layers.{i}.q.lora_A", (16, 32)) for i in range(8)]
    out = {}
    for tag, shapes, pre, grp, wire, sigs in (
            ("q", qshapes, [f"qformer.encoder.layer.{i}." for i in range(4)], 1, torch.bfloat16, [3, 2, 1, 0, -2, -1]),
            ("l", lshapes, [f"layers.{i}." for i in range(8)], 2, None, list(reversed(range(8))))):
        p = _FakePack(shapes)
        g = torch.Generator().manual_seed(7 * rank + len(shapes))
        p.grad.copy_(torch.randn(p.numel, generator=g))
        mine = p.grad.clone()
        bk = dp.GradBuckets(p.grad, dp.layer_boundaries(p, pre, grp), wire_dtype=wire)
        sent = []
        ready = bk.ready
        bk.ready = lambda i, _r=ready, _s=sent: (_s.append(i), _r(i))[1]
        hook = dp.bucket_hook(p, bk, pre, grp)
        for s in sigs:
            hook(s)
        bk.wait()
        out[tag] = {"mine": mine, "sum": p.grad.clone(), "sent": sent, "n": bk.n, "head_lo": bk.bounds[-2]}
    torch.save(out, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_world4_gloo_bucket_order_and_bf16_wire(tmp_path):
    """world 4: every layer / K|V / lead bucket leaves exactly once in the backward's order on all four
    ranks, ranks end bitwise identical, the f32 wire equals the sum and the bf16 wire stays inside world * 2^-9."""
    world, port = 4, _free_port()
    mp.spawn(_world4_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    R4 = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    for tag in ("q", "l"):
        n, head_lo = R4[0][tag]["n"], R4[0][tag]["head_lo"]
        for r in range(world):
            sent = R4[r][tag]["sent"]
            # the trailing head bucket of the Q-Former pack is never signalled (untouched heads carry no gradient)
            want = list(reversed(range(n - 1))) if tag == "q" else list(reversed(range(n)))
            assert sent == want, (tag, r, sent)
            assert torch.equal(R4[r][tag]["sum"][:head_lo] if tag == "q" else R4[r][tag]["sum"], R4[0][tag]["sum"][:head_lo] if tag == "q" else R4[0][tag]["sum"])
        exact = sum(R4[r][tag]["mine"] for r in range(world))
        got = R4[0][tag]["sum"]
        if tag == "l":
            assert torch.allclose(got, exact, rtol=1e-5, atol=1e-6)
        else:
            e = float((got[:head_lo] - exact[:head_lo]).norm() / exact[:head_lo].norm())
            assert e < world * 2.0 ** -9, e

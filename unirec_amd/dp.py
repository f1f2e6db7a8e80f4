"""Data parallelism for the hot path: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI, sum all-reduce of the flat gradient packs in a few large buckets.

The reference has no distributed code at all; this is new.  Every sample is
independent in forward and loss (per-sample negatives), so ranks take disjoint slices
of the global minibatch and the only exchange is the gradient all-reduce:
  * buckets are contiguous ranges of a ParamPack's flat fp32 gradient buffer, in the order the
    backward completes them (LoRA layers L-1..0, then Q-Former layers L-1..0, then the query table);
  * each bucket's all-reduce is issued as soon as the backward has written it, on RCCL's own stream
    (torch orders it after the producing kernels), so it overlaps the rest of the backward;
  * dead reference tensors and untouched heads are not in any bucket (no bytes, no hang);
  * the 1/world_size average is folded into the fused AdamW (grad_scale).
xGMI is point-to-point (7 links x ~153 GB/s per GPU): few large messages, not many small ones.
Works unchanged with backend "gloo" on CPU tensors (tests/test_dp_gloo.py).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # UNIREC_DP_FORCE=1: initialise the process group (and run every bucket all-reduce) even with ONE rank, so the RCCL
    # launch path -- communicator, side stream, event fences -- can be exercised on a one-GPU box
    force = os.environ.get("UNIREC_DP_FORCE") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # UNIREC_DP_BACKEND=gloo: functional rehearsal of the multi-rank path with several ranks on ONE GPU
            # (RCCL needs one device per rank); the product default on GPUs is nccl = RCCL over xGMI
            backend = os.environ.get("UNIREC_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_global, rank, world):
    """Rank r takes samples [r*B_local, (r+1)*B_local) of the global minibatch."""
    per = n_global // world
    return rank * per, (rank + 1) * per


class GradBuckets:
    """Contiguous buckets over one flat gradient buffer, reduced as they become ready."""

    def __init__(self, flat_grad, boundaries, group=None):
        """boundaries: ascending element offsets [0, ..., numel]; bucket i = [b[i], b[i+1])."""
        self.flat = flat_grad
        self.bounds = list(boundaries)
        self.group = group
        self.pending = []
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get("UNIREC_DP_FORCE") == "1")

    @property
    def n(self):
        return len(self.bounds) - 1

    def ready(self, i):
        """All gradients of bucket i have been written on the current stream: start its all-reduce."""
        if not self.enabled:
            return
        lo, hi = self.bounds[i], self.bounds[i + 1]
        if hi > lo:
            self.pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def ready_all(self):
        for i in reversed(range(self.n)):
            self.ready(i)

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []


def layer_boundaries(pack, layer_prefixes, group_size):
    """Bucket boundaries for a pack whose entries are ordered by layer: one bucket per `group_size`
    consecutive layers (plus whatever precedes the first / follows the last layer)."""
    starts = []
    for pre in layer_prefixes:
        offs = [pack.offsets[n] for n in pack.names if n.startswith(pre)]
        starts.append(min(offs))
    bounds = [0]
    for k in range(0, len(starts), group_size):
        if starts[k] > bounds[-1]:
            bounds.append(starts[k])
    # everything after the last layer group (heads) gets its own trailing bucket
    last_layer_end = max(pack.offsets[n] + ((pack.params[n].numel() + 7) // 8) * 8 for n in pack.names if n.startswith(layer_prefixes[-1]))
    if last_layer_end < pack.numel:
        bounds.append(last_layer_end)
    bounds.append(pack.numel)
    return sorted(set(bounds))

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
import atexit
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
    """Rank r takes a contiguous slice of the global minibatch; the n_global % world remainder goes one
    sample each to the first ranks, so no sample is dropped."""
    per, rem = divmod(n_global, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def set_dp_rank(rank, *modules):
    """Key every dropout mask on the GLOBAL sample index: all ranks keep the SAME seeds, and a module's forward
    over B local samples offsets its dropout counters by rank * B samples (`.dp_rank`; `set_sample_offset` overrides it for
    micro-batches, extra forwards and unequal shards).  The N-rank step then draws, sample for sample, the masks of a single-process step over the concatenation of the
    ranks' local batches -- training does not depend on the number of ranks (tests/test_gpu_dp_product.py)."""
    for m in modules:
        if m is None:
            continue
        subs = list(m.modules()) if hasattr(m, "modules") else [m]
        for sub in subs:
            if hasattr(sub, "dp_rank"):
                sub.dp_rank = int(rank)


def set_sample_offset(offset, *modules):
    """Explicit index of a forward's first sample in the GLOBAL minibatch (overrides `dp_rank * B`): needed whenever a rank
    runs more than one forward per step or shards are unequal -- micro-batch k of `mb` samples on rank r of a step over
    B samples per rank starts at r * B + k * mb.  None returns to the `dp_rank * B` rule."""
    for m in modules:
        if m is None:
            continue
        subs = list(m.modules()) if hasattr(m, "modules") else [m]
        for sub in subs:
            if hasattr(sub, "sample_offset"):
                sub.sample_offset = None if offset is None else int(offset)


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_sum_(t, group=None):
    """In-place sum all-reduce of a small tensor (loss denominators such as the item loss's sum of mask); no-op for one rank."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def launch_ranks(n, script, argv, extra_env=None):
    """Start `n` ranks of `script` on this node as CHILD processes (python -m torch.distributed.run, rendezvous on
    127.0.0.1) and return the launcher's exit code.  Must be called before the calling process touches the GPU: the
    parent only waits and relays, it never re-execs itself."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    env.update(extra_env or {})
    # the c10d rendezvous on port 0 lets torchrun pick a free port itself (no bind / close / reuse race on a busy host)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", "--local-addr", "127.0.0.1", script] + list(argv)
    return subprocess.run(cmd, env=env).returncode


class NativeComm:
    """The library's own RCCL communicator (include/unirec_hip.h: ur_comm_*): in-place SUM all-reduces queued on ONE
    library-owned side stream, fenced with events against the producing / consuming torch streams -- no host
    synchronisation, no torch.distributed call on the data path.  torch.distributed (any backend) is only the channel that
    hands rank 0's 128-byte RCCL id to the other ranks; one rank needs no channel at all."""

    _DTYPES = {torch.float32: 0, torch.bfloat16: 1}

    def __init__(self, rank, world, device, unique_id):
        import ctypes
        from . import _lib
        self._lib = _lib.load()
        self._check = _lib.check
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("NativeComm: RCCL reduces device buffers; pass a cuda device")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        if len(unique_id) != 128:
            raise ValueError("NativeComm: the RCCL unique id is 128 bytes")
        handle = ctypes.c_void_p()
        idbuf = (ctypes.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._check(self._lib.ur_comm_init(ctypes.byref(handle), self.rank, self.world, idbuf, index), "ur_comm_init")
        self._handle = handle
        self.launches = 0

    @staticmethod
    def new_unique_id():
        import ctypes
        from . import _lib
        buf = (ctypes.c_char * 128)()
        _lib.check(_lib.load().ur_comm_unique_id(buf), "ur_comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_env(cls, device, group=None):
        """Rank / world of the initialised process group (or a world of one without a group); rank 0 creates the id and
        broadcast_object_list carries it (works on gloo and nccl groups alike)."""
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        else:
            rank, world = 0, 1
        box = [cls.new_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, device, box[0])

    def all_reduce_(self, t, stream=None):
        """Queue the in-place sum of `t` (contiguous f32 / bf16 device tensor, written on `stream` -- default: torch's
        current stream) on the communicator's side stream."""
        if not t.is_contiguous() or t.device != self.device or t.dtype not in self._DTYPES:
            raise ValueError(f"NativeComm.all_reduce_: contiguous f32/bf16 tensor on {self.device} expected, got {t.dtype} on {t.device}")
        st = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        self._check(self._lib.ur_comm_allreduce_async(self._handle, t.data_ptr(), t.numel(), self._DTYPES[t.dtype], st), "ur_comm_allreduce_async")
        self.launches += 1
        return self

    def ticket(self):
        """ticket of the last all-reduce queued (0: none yet); wait(ticket=...) fences on that one and everything before it"""
        return int(self._lib.ur_comm_ticket(self._handle))

    def wait(self, stream=None, ticket=None):
        """`stream` (default: torch's current stream) waits for every all-reduce queued so far, or up to `ticket`."""
        st = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        if ticket is None:
            self._check(self._lib.ur_comm_wait(self._handle, st), "ur_comm_wait")
        else:
            self._check(self._lib.ur_comm_wait_ticket(self._handle, int(ticket), st), "ur_comm_wait_ticket")

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            h, self._handle = self._handle, None
            self._check(self._lib.ur_comm_destroy(h), "ur_comm_destroy")



_native = {}


def native_comm(device, group=None):
    """The process's NativeComm for `device` (created on first use: a collective -- every rank must reach it)."""
    key = (torch.device(device).index, id(group))
    if key not in _native:
        _native[key] = NativeComm.from_env(device, group)
    return _native[key]


def close_native_comms():
    for c in list(_native.values()):
        try:
            c.close()
        except Exception:
            pass
    _native.clear()


# communicators are torn down at interpreter exit, while HIP is still alive -- not from __del__ during module teardown
atexit.register(close_native_comms)


def use_native_comm(group=None):
    """The bucket all-reduces of a multi-rank run go through the library's own communicator (ur_comm_*: the C-ABI path SURVEY 8(b)
    names) whenever the process group runs on RCCL; UNIREC_DP_COMM=torch opts out (torch.distributed's all_reduce), =native forces
    it (also for one rank: the launch path on a one-GPU box).  gloo groups (CPU rehearsals) always use torch.distributed."""
    mode = os.environ.get("UNIREC_DP_COMM", "auto")
    if mode in ("native", "torch"):
        return mode == "native"
    try:
        return dist.get_world_size(group) > 1 and dist.get_backend(group) == "nccl"
    except Exception:
        return False


class GradBuckets:
    """Contiguous buckets over one flat gradient buffer, reduced as they become ready."""

    def __init__(self, flat_grad, boundaries, group=None, comm=None, wire_dtype=None):
        """boundaries: ascending element offsets [0, ..., numel]; bucket i = [b[i], b[i+1]).  comm: a NativeComm to reduce
        through (default: torch.distributed's group, or the process's native communicator under UNIREC_DP_COMM=native).
        wire_dtype = torch.bfloat16 (opt-in; default: the buffer's own f32): a bucket travels as bf16 -- cast into a staging buffer
        when it becomes ready, summed in bf16 by the collective (UR_COMM_BF16), cast back into the f32 gradient buffer behind its
        wait.  Half the bytes on the links for the one pack whose all-reduce cannot hide under the backward (the item Q-Former of the
        joint step is UPSTREAM of the decoder: its 714 MB of f32 gradients become final in the backward's last milliseconds, SURVEY
        5.8); the price is one bf16 rounding of every summand and partial sum (relative error <= world * 2^-9 of the largest term).
        The optimizer state and the master weights stay f32."""
        self.flat = flat_grad
        if wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("GradBuckets: wire_dtype must be None / torch.float32 / torch.bfloat16")
        self.wire = None               # bf16 staging buffer of the whole pack (allocated on first use)
        self.wire_on = wire_dtype == torch.bfloat16
        self.unpack = []               # buckets whose reduced bf16 image still has to be cast back: (lo, hi, ticket or None)
        self.bounds = list(boundaries)
        self.group = group
        self.pending = []
        self.tickets = {}          # native communicator: bucket index -> ticket of its all-reduce (wait_bucket)
        self.stash = None          # micro-batch accumulation: gradients of the earlier micro-batches of this step
        self.hold = False          # True while a non-final micro-batch runs its backward: hooks accumulate, nothing is sent
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get("UNIREC_DP_FORCE") == "1")
        self.comm = comm
        if comm is not None:
            self.enabled = True
        elif self.enabled and use_native_comm(group) and flat_grad.is_cuda:
            self.comm = native_comm(flat_grad.device, group)

    @property
    def n(self):
        return len(self.bounds) - 1

    def ready(self, i):
        """All gradients of bucket i have been written on the current stream: fold in the stashed gradients of the earlier
        micro-batches (if any) and start the bucket's all-reduce -- or, on a non-final micro-batch, only stash."""
        lo, hi = self.bounds[i], self.bounds[i + 1]
        if hi <= lo:
            return
        if self.hold:
            if self.stash is None:
                self.stash = torch.zeros_like(self.flat)
            self.stash[lo:hi].add_(self.flat[lo:hi])
            return
        if self.stash is not None:
            self.flat[lo:hi].add_(self.stash[lo:hi])
            self.stash[lo:hi].zero_()
        if not self.enabled:
            return
        buf = self.flat[lo:hi]
        if self.wire_on:
            if self.wire is None:
                self.wire = torch.empty(self.flat.numel(), dtype=torch.bfloat16, device=self.flat.device)
            buf = self.wire[lo:hi]
            _cast(self.flat[lo:hi], buf)
        if self.comm is not None:
            self.comm.all_reduce_(buf)
            self.tickets[i] = self.comm.ticket()
        else:
            self.pending.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if self.wire_on:
            self.unpack.append((i, lo, hi))

    def begin_micro_batch(self, last):
        """Gradient accumulation over micro-batches inside ONE optimizer step (every backward OVERWRITES the flat gradient
        buffer): non-final micro-batches add their buckets to a stash as the backward completes them; the final one adds
        the stash back bucket by bucket, right before that bucket's all-reduce, so the overlap with the backward stays."""
        self.hold = not last

    def ready_all(self):
        for i in reversed(range(self.n)):
            self.ready(i)

    def wait_bucket(self, i, stream=None):
        """the current stream waits for bucket i's all-reduce only (native communicator; otherwise for everything queued)"""
        if self.comm is not None and self.enabled and i in self.tickets:
            self.comm.wait(stream, ticket=self.tickets[i])
            if self.unpack:
                if stream is not None and self.flat.is_cuda:
                    with torch.cuda.stream(stream):
                        self._unpack(only=i)
                else:
                    self._unpack(only=i)
        else:
            self.wait()

    def _unpack(self, only=None):
        """reduced bf16 images -> the f32 gradient buffer (the caller has fenced the current stream behind their all-reduces)"""
        rest = []
        for (i, lo, hi) in self.unpack:
            if only is None or i == only:
                _cast(self.wire[lo:hi], self.flat[lo:hi])
            else:
                rest.append((i, lo, hi))
        self.unpack = rest

    def wait(self):
        if self.comm is not None and self.enabled:
            self.comm.wait()
            self.tickets = {}
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.unpack:
            self._unpack()


def _cast(src, dst):
    """dst <- src across f32 / bf16 (a device cast kernel for device buffers, torch's copy on the host: the gloo rehearsals)"""
    if src.is_cuda:
        from . import hip
        (hip.cast_f32_to_bf16 if src.dtype == torch.float32 else hip.cast_bf16_to_f32)(src, dst)
    else:
        dst.copy_(src)


def _hoisted(n):
    return ".crossattention.self.key." in n or ".crossattention.self.value." in n


def layer_boundaries(pack, layer_prefixes, group_size):
    """Bucket boundaries for a pack whose entries are ordered by layer: one bucket per `group_size`
    consecutive layers (plus whatever precedes the first / follows the last layer)."""
    # (the Q-Formers keep the cross-attention K | V projections of ALL layers side by side ahead of layer 0 -- one GEMM serves them,
    # qformer.py:_cross_kv_names.  Their gradients are final right after the layer loop, BEFORE the embedding LayerNorm / query-table
    # reduction that closes the backward: they get a bucket of their own, sent from its own hook (BertModel.grad_ready_hook(-2)),
    # so the exposed tail of the step is the small query-table bucket, not 50 MB of K | V gradients)
    hoisted = _hoisted
    starts = []
    for pre in layer_prefixes:
        offs = [pack.offsets[n] for n in pack.names if n.startswith(pre) and not hoisted(n)]
        starts.append(min(offs))
    bounds = [0]
    h_offs = [pack.offsets[n] for n in pack.names if hoisted(n)]
    if h_offs and 0 < min(h_offs) < min(starts):
        bounds.append(min(h_offs))
    for k in range(0, len(starts), group_size):
        if starts[k] > bounds[-1]:
            bounds.append(starts[k])
    # everything after the last layer group (heads) gets its own trailing bucket
    last_layer_end = max(pack.offsets[n] + ((pack.params[n].numel() + 7) // 8) * 8 for n in pack.names if n.startswith(layer_prefixes[-1]) and not hoisted(n))
    if last_layer_end < pack.numel:
        bounds.append(last_layer_end)
    bounds.append(pack.numel)
    return sorted(set(bounds))


def bucket_hook(pack, buckets, layer_prefixes, group_size):
    """`grad_ready_hook` of a model whose backward walks its layers last to first (BertModel, Qwen3LoRAModel) for the buckets of
    `layer_boundaries(pack, layer_prefixes, group_size)`: signal i >= 0 = layer i's gradients are final (the bucket of a layer group
    goes out when its LOWEST layer is done), -2 = the hoisted cross-attention K | V gradients, -1 = whatever precedes them (query
    table, embedding LayerNorm).  Buckets are found by OFFSET, so the mapping follows the boundaries whatever they contain."""
    import bisect
    bounds = buckets.bounds

    def at(off):
        return bisect.bisect_right(bounds, off) - 1
    first_of = {}
    for i, pre in enumerate(layer_prefixes):
        offs = [pack.offsets[n] for n in pack.names if n.startswith(pre) and not _hoisted(n)]
        if (i % group_size) == 0 and offs:
            first_of[i] = at(min(offs))
    h_offs = [pack.offsets[n] for n in pack.names if _hoisted(n)]
    h_bucket = at(min(h_offs)) if h_offs else None
    lead = 0
    if h_bucket == lead:
        h_bucket = None            # no boundary between them (nothing precedes the K | V block): one signal, the last one, sends it
    # a pack with NOTHING ahead of layer 0 (the LoRA pack: `layers.0.` sits at offset 0) has no -1 signal to wait for -- its model
    # (Qwen3LoRAModel.backward) fires layer signals only -- so the lead bucket goes out with layer 0, the lowest layer of its group
    lead_by_layer = first_of.get(0) == lead and h_bucket is None and not h_offs

    def hook(i):
        if i == -1:
            if not lead_by_layer:
                buckets.ready(lead)
        elif i == -2:
            if h_bucket is not None:
                buckets.ready(h_bucket)
        elif i == 0 and lead_by_layer:
            buckets.ready(lead)
        elif i in first_of and first_of[i] not in (lead, h_bucket):
            buckets.ready(first_of[i])
    hook.first_of, hook.hoisted_bucket, hook.lead_by_layer = first_of, h_bucket, lead_by_layer
    return hook

"""GPU: the library's RCCL communicator (include/unirec_hip.h: ur_comm_*, csrc/comm.hip) -- SURVEY.md 8(b) communicator
calls, 8(e).  The reference has no distributed code: the contract is the north-star's pure data
parallelism (in-place SUM all-reduce of gradient buckets on a library-owned side stream, event-fenced, no host sync).

A one-GPU box can hold a communicator of ONE rank: the reduction is the identity there, but communicator creation from a
unique id, the side stream, both event fences and the error paths all execute.  The two-rank cases need two GPUs (the
driver's 8-GPU node) and are skipped otherwise."""
import ctypes
import json
import os
import subprocess
import sys

import pytest
import torch

from unirec_amd import _lib, dp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda", 0)


@pytest.fixture()
def comm():
    c = dp.NativeComm(0, 1, DEV, dp.NativeComm.new_unique_id())
    yield c
    c.close()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_unique_ids_are_fresh_and_128_bytes():
    a, b = dp.NativeComm.new_unique_id(), dp.NativeComm.new_unique_id()
    assert len(a) == len(b) == 128 and a != b


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_one_rank_all_reduce_is_the_identity_and_is_ordered_after_the_producer(comm, dtype):
    """The buffer is produced by a long chain of kernels on a NON-default stream right before the call, and consumed on a
    third stream right after ur_comm_wait: only the event fences order the three streams (no host sync until the end)."""
    n = (1 << 22) + 8
    prod, cons = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    g = torch.Generator(device="cpu").manual_seed(3)
    src = torch.randn(n, generator=g).to(DEV)
    buf = torch.zeros(n, dtype=dtype, device=DEV)
    out = torch.empty(n, dtype=torch.float32, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(prod):
        acc = src.clone()
        for _ in range(40):               # ~ms of producer work the side stream must wait for
            acc = acc * 1.0001 + 0.001
        buf.copy_(acc)
        comm.all_reduce_(buf)
    with torch.cuda.stream(cons):
        comm.wait()
        out.copy_(buf.float() * 2.0)
    torch.cuda.synchronize()
    want = src.clone()
    for _ in range(40):
        want = want * 1.0001 + 0.001
    want = want.to(dtype).float() * 2.0
    assert torch.equal(out, want)
    assert comm.launches == 1


def test_bucket_slices_reduce_in_place_and_zero_counts_are_accepted(comm):
    flat = torch.arange(1000, dtype=torch.float32, device=DEV)
    ref = flat.clone()
    bk = dp.GradBuckets(flat, [0, 8, 8, 504, 1000], comm=comm)
    bk.ready_all()
    bk.wait()
    torch.cuda.synchronize()
    assert torch.equal(flat, ref) and comm.launches == 3        # the empty bucket [8, 8) sends nothing


def test_tickets_fence_single_buckets(comm):
    """every all-reduce has a ticket; a consumer stream can wait for ONE bucket (and what precedes it) while later ones are queued"""
    lib = _lib.load()
    t0 = comm.ticket()
    flat = torch.arange(4096, dtype=torch.float32, device=DEV)
    ref = flat.clone()
    bk = dp.GradBuckets(flat, [0, 1024, 2048, 4096], comm=comm)
    bk.ready_all()                                   # buckets 2, 1, 0 in backward order
    assert comm.ticket() == t0 + 3 and bk.tickets == {2: t0 + 1, 1: t0 + 2, 0: t0 + 3}
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        bk.wait_bucket(2, side)                      # fences on the first queued bucket only
        head = flat[2048:].clone()
    bk.wait()
    torch.cuda.synchronize()
    assert torch.equal(head, ref[2048:]) and torch.equal(flat, ref) and bk.tickets == {}
    st = torch.cuda.current_stream().cuda_stream
    assert lib.ur_comm_wait_ticket(comm._handle, comm.ticket() + 1, st) < 0 and b"ticket" in lib.ur_last_error()
    assert lib.ur_comm_wait_ticket(comm._handle, 0, st) == 0
    # more all-reduces than the ring holds: an old ticket still waits (for a later event of the same in-order stream)
    x = torch.ones(8, device=DEV)
    first = None
    for i in range(70):
        comm.all_reduce_(x)
        first = first or comm.ticket()
    comm.wait(ticket=first)
    comm.wait()
    torch.cuda.synchronize()
    assert torch.equal(x, torch.ones(8, device=DEV))


def test_error_paths_return_codes_and_messages(comm):
    lib = _lib.load()
    t = torch.zeros(16, dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.ur_comm_allreduce_async(comm._handle, t.data_ptr(), 16, 7, st) < 0 and b"dtype" in lib.ur_last_error()
    assert lib.ur_comm_allreduce_async(comm._handle, None, 16, 0, st) < 0 and b"null" in lib.ur_last_error()
    assert lib.ur_comm_allreduce_async(comm._handle, t.data_ptr() + 2, 4, 0, st) < 0 and b"aligned" in lib.ur_last_error()
    assert lib.ur_comm_allreduce_async(comm._handle, t.data_ptr(), -1, 0, st) < 0
    bogus = ctypes.create_string_buffer(64)
    assert lib.ur_comm_wait(bogus, st) < 0 and b"not a communicator" in lib.ur_last_error()
    h = ctypes.c_void_p()
    assert lib.ur_comm_init(ctypes.byref(h), 2, 2, ctypes.create_string_buffer(128), 0) < 0 and not h.value
    with pytest.raises(ValueError):
        comm.all_reduce_(torch.zeros(4, dtype=torch.float64, device=DEV))
    with pytest.raises(ValueError):
        comm.all_reduce_(torch.zeros(4, 4, device=DEV).t())


SMALL = ["--layers", "2", "--batch", "8", "--seq", "512", "--hist", "10", "--pool", "50", "--steps", "2", "--warmup", "1",
         "--no-cpu-baseline", "--no-stages"]


def _bench(extra_args=(), env=None, timeout=600):
    e = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("UNIREC_DP_FORCE", "UNIREC_DP_BACKEND", "UNIREC_DP_COMM"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + list(extra_args), env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_joint_step_through_the_native_communicator_equals_the_plain_step():
    """bench.py's joint step with every bucket all-reduce going through ur_comm_* (one rank): same loss and parameter
    checksums as the step without any process group, and the line says which communicator ran."""
    plain = _bench()
    native = _bench(env={"UNIREC_DP_FORCE": "1", "UNIREC_DP_COMM": "native", "MASTER_PORT": str(_free_port())})
    assert native["comm"]["backend"].startswith("rccl (native ur_comm_*") and native["comm"]["ranks"] == 1
    assert native["comm"]["allreduce_launches"] > 0
    assert native["loss"] == plain["loss"] and native["param_checksum"] == plain["param_checksum"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's 8-GPU node)")
def test_two_ranks_through_the_native_communicator_match_torch_nccl():
    """Two real ranks: the step reduced by ur_comm_* ends on the same loss / parameter checksums as the step reduced by
    torch.distributed's nccl group (both are RCCL ring sums over the same two buffers in the same order)."""
    a = _bench(["--gpus", "2"], timeout=900)
    b = _bench(["--gpus", "2"], env={"UNIREC_DP_COMM": "native"}, timeout=900)
    assert a["comm"]["ranks"] == b["comm"]["ranks"] == 2 and b["comm"]["backend"].startswith("rccl (native")
    assert b["loss"] == a["loss"] and b["param_checksum"] == a["param_checksum"]

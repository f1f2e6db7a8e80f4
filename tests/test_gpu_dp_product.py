"""GPU: the PRODUCT item Q-Former step under two data-parallel ranks.  Two child processes share cuda:0
over gloo (RCCL needs one device per rank; everything above the backend is the shipped path: HIP kernels, bucket hooks
fired from the backward, sum all-reduce of the flat gradient pack, all-reduced sum(mask) in the item loss, fused AdamW
with the folded 1/world).  Asserts (i) both ranks end the step with bit-identical parameters, (ii) the reduced gradient
equals the single-process gradient over the global batch -- also with dropout ON: masks are keyed on the global sample index."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp_product_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(outdir, world, Bg, dropout, mode=None, extra_env=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, **(extra_env or {}), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   UNIREC_DP_BACKEND="gloo", OMP_NUM_THREADS="2", GLOO_SOCKET_IFNAME="lo")      # loopback: the box's hostname may not resolve
        procs.append(subprocess.Popen([sys.executable, WORKER, str(outdir), str(Bg), str(dropout)] + ([mode] if mode else []), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240)[0])
    except subprocess.TimeoutExpired:
        # a rank that never returns (rendezvous or a collective that the other rank left) must fail the test with both ranks'
        # output, not stall the whole run: end exactly the processes started here
        for p in procs:
            if p.poll() is None:
                p.kill()
        tails = [p.communicate()[0][-2000:] for p in procs]
        raise AssertionError(f"a rank did not finish within 240 s (world {world}, port {port}):\n" + "\n-----\n".join(tails))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]


def test_two_ranks_match_each_other_and_the_single_process_step(tmp_path):
    Bg = 24
    d2 = tmp_path / "w2"; d2.mkdir()
    r0, r1 = _launch(d2, 2, Bg, 0.0)
    assert r0["enabled"] and r1["enabled"] and r0["world"] == 2
    assert r0["n"] + r1["n"] == Bg
    assert torch.equal(r0["grad"], r1["grad"]), "reduced gradients must be bit-identical across ranks"
    assert torch.equal(r0["master"], r1["master"]), "parameters after the step must be bit-identical across ranks"
    d1 = tmp_path / "w1"; d1.mkdir()
    (s0,) = _launch(d1, 1, Bg, 0.0)
    # loss: mean over ranks of the local losses == the global loss (sum(mask) all-reduced)
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - s0["loss"]) <= 2e-3 * abs(s0["loss"])
    g2, g1 = r0["grad"], s0["grad"]
    rel = float((g2 - g1).norm() / g1.norm())
    assert rel <= 2e-2, rel            # bf16 operands, different batch split: same tolerance as the parity tests
    assert float(g1.norm()) > 0


def test_dropout_masks_are_keyed_on_the_global_sample_index(tmp_path):
    """the result must not depend on the number of ranks.  With dropout ON (hidden 0.2, attention 0.2) the reduced
    gradient of the 2-rank step equals the single-process gradient over the global batch to the same tolerance as without
    dropout: every rank keeps the SAME seeds and offsets its dropout counters by rank * local batch, so sample i draws the
    masks of global sample i wherever it runs.  (With per-rank seeds the two gradients differ by the dropout noise itself.)"""
    Bg = 16
    d2 = tmp_path / "w2"; d2.mkdir()
    r0, r1 = _launch(d2, 2, Bg, 0.2)
    assert r0["seed"] == r1["seed"]
    assert torch.equal(r0["master"], r1["master"]) and torch.equal(r0["grad"], r1["grad"])
    d1 = tmp_path / "w1"; d1.mkdir()
    (s0,) = _launch(d1, 1, Bg, 0.2)
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - s0["loss"]) <= 2e-3 * abs(s0["loss"])
    rel = float((r0["grad"] - s0["grad"]).norm() / s0["grad"].norm())
    assert rel <= 2e-2, rel


def test_joint_step_under_two_ranks_equals_the_single_process_step_with_dropout_on(tmp_path):
    """The headline path under data parallelism: item Q-Former (hidden + attention dropout 0.1) -> injection -> Qwen3 + LoRA
    (lora_dropout 0.1) -> InfoNCE.  Two ranks of two sequences each reduce to the gradient of ONE process over the four
    sequences -- LoRA A / B and every Q-Former tensor -- because all masks are keyed on the global sample index."""
    d2 = tmp_path / "w2"; d2.mkdir()
    r0, r1 = _launch(d2, 2, 4, 0.1, "joint")
    assert r0["world"] == 2 and r0["n"] == 2 and r1["n"] == 2 and r0["lora_seed"] == r1["lora_seed"] and r0["seed"] == r1["seed"]
    d1 = tmp_path / "w1"; d1.mkdir()
    (s0,) = _launch(d1, 1, 4, 0.1, "joint")
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - s0["loss"]) <= 5e-3 * abs(s0["loss"])
    for k in ("lora", "qformer"):
        assert torch.equal(r0["grads"][k], r1["grads"][k])
        g2, g1 = r0["grads"][k], s0["grads"][k]
        rel = float((g2 - g1).norm() / g1.norm())
        print(k, "2-rank vs 1-rank gradient rel err", rel)
        assert float(g1.norm()) > 0 and rel <= 3e-2, (k, rel)


def test_bf16_wire_buckets_stay_within_bf16_rounding_of_the_f32_step(tmp_path):
    """dp.GradBuckets(wire_dtype=bfloat16) -- the opt-in for the item Q-Former's pack: buckets are cast to bf16 as they become ready,
    summed in bf16, cast back behind the wait.  Both ranks still end bit-identical, and the reduced gradient is the f32-wire one up to
    one bf16 rounding per summand (relative Frobenius error ~2^-9)."""
    Bg = 16
    dw = tmp_path / "wire"; dw.mkdir()
    r0, r1 = _launch(dw, 2, Bg, 0.0, extra_env={"UNIREC_TEST_WIRE": "bf16"})
    assert torch.equal(r0["grad"], r1["grad"]) and torch.equal(r0["master"], r1["master"])
    df = tmp_path / "f32"; df.mkdir()
    f0, _ = _launch(df, 2, Bg, 0.0)
    rel = float((r0["grad"] - f0["grad"]).norm() / f0["grad"].norm())
    print("bf16-wire vs f32-wire reduced gradient rel err", rel)
    assert 0 < rel <= 6e-3, rel

"""Shared helpers for the GPU parity tests (HIP path vs oracle / golden fixtures)."""
import os

import numpy as np
import torch

from oracle import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Stated tolerances of the bf16 (fp32-accumulate) HIP path against the fp32 reference/oracle:
#   relative Frobenius error of an output tensor      <= OUT_REL
#   relative Frobenius error of a parameter gradient  <= GRAD_REL
# (bf16 has an 8-bit significand: 2^-9 ~ 2e-3 per rounding; 2..12 post-LN layers and a 28-layer
#  pre-norm stack keep the accumulated error below these bounds -- measured values are printed.)
OUT_REL = 2e-2
GRAD_REL = 4e-2


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def rel_err(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    den = np.linalg.norm(want)
    return float(np.linalg.norm(got - want) / den) if den > 0 else float(np.linalg.norm(got))


def grad_scale(golden, keys):
    """Largest reference gradient norm among `keys`: the scale a gradient that is analytically zero (norm <= floor in the
    reference) is held against -- in bf16 it comes out as rounding noise of its neighbours, never as exact zeros."""
    return max(float(np.linalg.norm(np.asarray(golden["grad/" + k], dtype=np.float64))) for k in keys)


def assert_close(got, want, tol, what, floor=0.0, abs_scale=0.0, ref_scale=0.0):
    """Relative Frobenius error <= tol.  floor: reference norm at or below which the tensor counts as analytically zero;
    the HIP result must then be small as well: ||got|| <= max(floor, tol * ref_scale) (ref_scale = grad_scale of the test's
    gradient keys).  abs_scale: norm below which a gradient counts as ill-conditioned and its error is measured against
    abs_scale instead of its own norm (see test_joint_matches_reference)."""
    if isinstance(got, torch.Tensor):
        got = got.detach().float().cpu().numpy()
    e = rel_err(got, want)
    wn = float(np.linalg.norm(np.asarray(want, dtype=np.float64)))
    gn = float(np.linalg.norm(np.asarray(got, dtype=np.float64)))
    zero_ok = wn <= floor and gn <= max(floor, tol * ref_scale)
    if wn <= floor:          # analytically zero reference: only its own bound can pass it (rel_err is meaningless there)
        branch = "floor" if zero_ok else "FAIL"
    else:
        branch = "rel" if e <= tol else ("abs_scale" if (abs_scale > 0.0 and e * wn <= tol * abs_scale) else "FAIL")
    if wn <= floor:
        print(f"  {what}: reference norm {wn:.2e} <= floor {floor:.1e}; ||got|| = {gn:.3e} against max(floor, tol * ref_scale) = {max(floor, tol * ref_scale):.3e}")
    print(f"  {what}: rel_err={e:.3e} (tol {tol:.1e}) passed-by={branch}")
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    assert branch != "FAIL", f"{what}: rel err {e:.3e} > {tol:.1e} (reference norm {wn:.3e}, result norm {gn:.3e})"


def load_generated(module, shapes, seed, device="cuda"):
    """Push oracle.weights tensors into a unirec_amd module through load_state_dict (live keys only;
    the dead reference tensors keep their init) and move it to the device."""
    gen = W.fill_state_dict(shapes, seed)
    sd = {k: torch.from_numpy(v) for k, v in gen.items()}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return module.to(device)

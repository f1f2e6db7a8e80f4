"""CPU: the branch-free normal CDF behind every GELU of the library (csrc/common.hip.h: norm_cdf_f, replacing the erf-GELU of
/root/reference/models/qformer.py:386-395 `ACT2FN["gelu"]`), emulated in fp32 with the SHIPPED coefficients (parsed from the header)
over all 65 280 finite bf16 inputs against float64: gelu(x) must round to the exact bf16 wherever |gelu(x)| > 1e-6, the tails must
go to 0 / x (not to erfc(4) / 2 * x), gelu'(x) within 1e-6 absolute."""
import os
import re
import numpy as np
import torch
from scipy.special import erfc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shipped():
    src = open(os.path.join(ROOT, "unirec_amd", "csrc", "common.hip.h")).read()
    body = src[src.index("float norm_cdf_f(float x)"):src.index("float gelu_erf_f(float x)")]
    lead = float(re.search(r"float g = ([-0-9.e+]+)f;", body).group(1))
    rest = [float(v) for v in re.findall(r"g = fmaf\(g, t, ([-0-9.e+]+)f\);", body)]
    clamp = float(re.search(r"fminf\(tu, ([0-9.]+)f\)", body).group(1))
    slope = float(re.search(r"fmaf\(tu - t, ([0-9.]+)f, g \* t\)", body).group(1))
    return [lead] + rest, clamp, slope          # descending powers of t


def _emulate(xf, coef, clamp, slope):
    f = np.float32
    with np.errstate(all="ignore"):
        tu = (np.abs(xf) * f(0.70710678118654752)).astype(f)
        t = np.minimum(tu, f(clamp))
        g = np.full_like(t, f(coef[0]))
        for c in coef[1:]:
            g = (g * t + f(c)).astype(f)
        ex = ((tu - t) * f(slope) + (g * t).astype(f)).astype(f)
        e = (f(0.5) * np.exp2((f(-1.4426950408889634) * ex).astype(f))).astype(f)
        phi = np.where(xf < 0, e, f(1) - e).astype(f)
        pdf = (f(0.39894228040143268) * np.exp2((f(-0.72134752044448170) * xf * xf).astype(f))).astype(f)
    return phi, (xf * phi).astype(f), (phi + xf * pdf).astype(f)


def test_shipped_normal_cdf_rounds_to_the_exact_bf16_gelu():
    coef, clamp, slope = _shipped()
    assert len(coef) == 8 and clamp == 4.0
    allb = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float().numpy()
    x = allb[np.isfinite(allb)]
    xd = x.astype(np.float64)
    phi, gelu, grad = _emulate(x.astype(np.float32), coef, clamp, slope)
    with np.errstate(all="ignore"):
        exact = 0.5 * xd * erfc(-xd / np.sqrt(2))
        exact_grad = 0.5 * erfc(-xd / np.sqrt(2)) + xd * np.exp(-0.5 * xd * xd) / np.sqrt(2 * np.pi)
    tobf = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16)
    big = np.abs(exact) > 1e-6
    assert torch.equal(tobf(gelu[big]), tobf(exact[big]))
    # tails: exactly x on the right, vanishing on the left, monotone CDF in between
    assert np.all(gelu[x > 6] == x[x > 6])
    assert np.all(np.abs(gelu[x < -6]) < 1e-7)
    order = np.argsort(x)
    assert np.all(np.diff(phi[order]) >= -1e-7)
    assert float(np.nanmax(np.abs(grad - exact_grad))) < 1e-6

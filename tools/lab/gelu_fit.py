"""Lab (CPU): the branch-free normal CDF of csrc/common.hip.h (norm_cdf_f) -- fit of g(t) = -ln(erfc(t)) / t on [0, 4] and an
exhaustive check of gelu(x) / gelu'(x) over ALL bf16 inputs in emulated fp32 against float64, beside the 1 + erf(x / sqrt 2) form."""
import numpy as np, torch
from scipy.special import erfcx, erfc, erf
from numpy.polynomial import chebyshev as C, polynomial as P
T, DEG = 4.0, 7
h = lambda t: t * t - np.log(erfcx(t))
g = lambda t: h(np.maximum(t, 1e-9)) / np.maximum(t, 1e-9)
n = 20000
t = np.cos(np.pi * (np.arange(n) + 0.5) / n) * T / 2 + T / 2
pc = C.Chebyshev.fit(t, g(t), DEG, domain=[0, T]).convert(kind=P.Polynomial).coef
print("coefficients (ascending):", [float(np.float32(v)) for v in pc], " tail slope:", 2 * T + 1 / T - 0.03)
allb = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float().numpy()
x = allb[np.isfinite(allb)]; xf = x.astype(np.float32); xd = x.astype(np.float64)
exact = 0.5 * xd * erfc(-xd / np.sqrt(2))
exact_grad = 0.5 * erfc(-xd / np.sqrt(2)) + xd * np.exp(-0.5 * xd * xd) / np.sqrt(2 * np.pi)
tobf = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16)
ulps = lambda a, b: (tobf(a).view(torch.int16).int() - tobf(b).view(torch.int16).int()).abs()
with np.errstate(all="ignore"):
    old = (0.5 * xf * (1 + erf((xf * np.float32(0.70710678)).astype(np.float64)).astype(np.float32))).astype(np.float32)
    tu = (np.abs(xf) * np.float32(0.70710678)).astype(np.float32); tt = np.minimum(tu, np.float32(T))
    acc = np.full_like(tt, np.float32(pc[-1]))
    for k in range(DEG - 1, -1, -1):
        acc = (acc * tt + np.float32(pc[k])).astype(np.float32)
    ex = ((tu - tt) * np.float32(8.22) + acc * tt).astype(np.float32)
    e = (np.float32(0.5) * np.exp2((-ex * np.float32(1.4426950408889634)).astype(np.float32))).astype(np.float32)
    phi = np.where(xf < 0, e, np.float32(1) - e).astype(np.float32)
    gelu = (xf * phi).astype(np.float32)
    grad = (phi + xf * (np.float32(0.39894228) * np.exp2((xf * xf * np.float32(-0.72134752)).astype(np.float32)))).astype(np.float32)
u_old, u_new = ulps(old, exact), ulps(gelu, exact)
big = np.abs(exact) > 1e-6
print("gelu: bf16 results different from exact: 1 + erf form", int((u_old > 0).sum()), " this form", int((u_new > 0).sum()),
      "; among |gelu| > 1e-6:", int((u_old[big] > 0).sum()), int((u_new[big] > 0).sum()), " max ulp there", int(u_old[big].max()), int(u_new[big].max()))
print("gelu': max abs error", float(np.nanmax(np.abs(grad - exact_grad))))

"""per-workgroup vs per-iteration cost of the generated forward: dense causal launches at several S, least squares of
t * 256 CUs = a * workgroups + b * iterations (a workgroup of query block x runs 4x + 4 ring iterations + 1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from unirec_amd import hip
nq, nkv, hd = 16, 8, 128
rows = []
for S, B in ((256, 256), (512, 256), (1024, 128), (2048, 64), (4096, 16)):
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
    q = qkv[..., :nq * hd].view(B, S, nq, hd); k = qkv[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = qkv[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
    for _ in range(3):
        hip.attn_fwd(q, k, v, causal=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); hip.attn_fwd(q, k, v, causal=True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    nx = S // 256
    wgs = B * nq * nx
    its = B * nq * sum(4 * x + 5 for x in range(nx))
    rows.append((S, B, t, wgs, its))
    print(f"S={S} B={B}: {t:.3f} ms  workgroups {wgs}  iterations {its}  -> {t * 1e3 * 256 / its:.3f} us per iteration if nothing else cost time", flush=True)
A = np.array([[r[3], r[4]] for r in rows], float)
y = np.array([r[2] * 1e3 * 256 for r in rows])
(a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
print(f"fit: {a:.2f} us per workgroup + {b:.3f} us per iteration (CU time)")

"""Lab: per-block fixed cost vs per-K-tile cost of the 256x256 projection GEMM (time = a + b * K/64 per block round)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M, N = 131072, 2048
g = torch.Generator().manual_seed(0)
for f32 in (False, True):
    pts = []
    for K in (256, 512, 1024, 2048, 4096):
        R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
        S = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        t = timeit(lambda: hip.gemm(R, S, out=out))
        rounds = (M // 256) * (N // 256) / 256
        pts.append((K, t * 1e3 / rounds))
        print(f"f32out={f32} K={K}: {t:.3f} ms  {2*M*N*K/t/1e9:.0f} TFLOP/s  per-block {t*1e3/rounds:.1f} us")
    (k0, t0), (k1, t1) = pts[2], pts[4]
    b = (t1 - t0) / ((k1 - k0) / 64)
    print(f"  -> per 64-deep tile {b:.2f} us (ideal 0.86 us at 2.5 PF), fixed per block {t0 - b * k0 / 64:.1f} us")

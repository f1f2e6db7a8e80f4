"""Lab: is the fixed per-block cost of the 256x256 GEMM intrinsic to a block or a chip-level burst effect?
Time K=256/1024 GEMMs whose grids are 1, 2, 4, 16 rounds of 256 workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator().manual_seed(0)
N = 2048
for K in (256, 1024):
    for rounds in (1, 2, 4, 16):
        M = rounds * 256 * 256 // (N // 256)
        R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
        S = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        t = timeit(lambda: hip.gemm(R, S, out=out))
        print(f"K={K} rounds={rounds} (M={M}): {t*1e3:.1f} us total, {t*1e3/rounds:.1f} us per round")

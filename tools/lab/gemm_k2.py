"""Lab: cost of the LoRA second K range (K2 = 16, register-staged tail tile) vs a zero-padded K2 = 64 (LDS-DMA full tile)."""
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
M = 131072
g = torch.Generator().manual_seed(0)
for N, K in ((2048, 1024), (3072, 1024), (1024, 3072)):
    R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
    S = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    t16 = torch.randn(M, 16, generator=g).cuda().to(torch.bfloat16)
    b16 = torch.randn(N, 16, generator=g).cuda().to(torch.bfloat16)
    t64 = torch.zeros(M, 64, dtype=torch.bfloat16, device="cuda"); t64[:, :16] = t16
    b64 = torch.zeros(N, 64, dtype=torch.bfloat16, device="cuda"); b64[:, :16] = b16
    for rep in range(2):
        a = timeit(lambda: hip.gemm(R, S, out=out))
        b = timeit(lambda: hip.gemm(R, S, out=out, R2=t16, S2=b16))
        c = timeit(lambda: hip.gemm(R, S, out=out, R2=t64, S2=b64))
        print(f"N={N} K={K}: plain {a*1e3:.0f} us | K2=16 {b*1e3:.0f} us | K2=64 padded {c*1e3:.0f} us")

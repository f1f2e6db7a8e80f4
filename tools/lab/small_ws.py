import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from unirec_amd import hip
g = torch.Generator().manual_seed(0)
for (M, N, K) in [(6400, 1024, 1024), (6400, 2048, 1024), (6400, 3072, 1024), (6400, 4096, 1024), (6400, 1024, 4096), (44800, 2048, 1024)]:
    R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
    res = {}
    outs = {}
    for mode in (1, 2):
        hip.gemm_persistent_mode(mode)
        out = hip.gemm(R, W)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hip.gemm(R, W, out=out)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[mode] = statistics.median(ts); outs[mode] = out.clone()
    hip.gemm_persistent_mode(-1)
    print(f"M={M} N={N} K={K}: default path {res[1]:.1f} us | mode 2 {res[2]:.1f} us  equal {torch.equal(outs[1], outs[2])}")

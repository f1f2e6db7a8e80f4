"""Lab: token-reduction (dW) GEMM out[Mo, No] = dY^T X over `red` tokens, token-major operands, split-K sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
from tools.kernel_bench import timeit
g = torch.Generator().manual_seed(0)
for (Mo, No, red) in [(1024, 4096, 32768), (4096, 1024, 32768), (3072, 1024, 32768), (1024, 1024, 32768), (768, 3072, 8192), (768, 768, 8192), (2304, 768, 8192)]:
    dy = torch.randn(red, Mo, generator=g).cuda().to(torch.bfloat16)
    x = torch.randn(red, No, generator=g).cuda().to(torch.bfloat16)
    out = torch.empty(Mo, No, device="cuda")
    line = f"out [{Mo},{No}] red {red}:"
    for sp in (1, 2, 3, 4, 6, 8, 12, 16, 32):
        if red // sp < 256: continue
        t = timeit(lambda: hip.gemm(dy, x, r_kcontig=False, s_kcontig=False, out=out, split_k=sp), 10)
        line += f"  s{sp}: {t * 1e3:6.1f}us"
    print(line, flush=True)

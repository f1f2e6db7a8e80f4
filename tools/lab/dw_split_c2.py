"""Lab: the item stage's token-reduction (dW) GEMM shapes, every split 1..32 (the 256x256 tile is taken once tiles * splits >= 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
from tools.kernel_bench import timeit
g = torch.Generator().manual_seed(0)
for (Mo, No, red) in [(768, 3072, 8192), (3072, 768, 8192), (2304, 768, 8192), (768, 768, 8192), (1536, 1024, 3584)]:
    dy = torch.randn(red, Mo, generator=g).cuda().to(torch.bfloat16)
    x = torch.randn(red, No, generator=g).cuda().to(torch.bfloat16)
    out = torch.empty(Mo, No, device="cuda")
    line = f"out [{Mo},{No}] red {red}:"
    for sp in list(range(1, 17)) + [18, 20, 24, 28, 32]:
        if red // sp < 128: continue
        t = timeit(lambda: hip.gemm(dy, x, r_kcontig=False, s_kcontig=False, out=out, split_k=sp), 10)
        t256 = ((Mo + 255) // 256) * ((No + 255) // 256) * sp
        line += f"  s{sp}{'*' if t256 >= 224 else ''}: {t * 1e3:5.1f}"
    print(line, flush=True)

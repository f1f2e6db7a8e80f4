"""Lab: the item stage's sub-round projection launches (M = 8192 tokens, N = 768: 96 big tiles) on the 128x128 generic kernel against the
persistent 256x256 kernel (UR_PERS_MIN_TILES, lab build: tools/lab/lib_variant.sh gemm_pers lab_pers -DUR_LAB=1)."""
import os, subprocess, sys
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from unirec_amd import hip
from tools.kernel_bench import timeit
g = torch.Generator().manual_seed(0)
out = []
for (M, N, K) in [(8192, 768, 768), (8192, 768, 2304), (8192, 768, 3072), (8192, 1024, 768), (16384, 768, 768), (3584, 1536, 1024)]:
    R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16); S = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, generator=g).cuda().to(torch.bfloat16)
    t0 = timeit(lambda: hip.gemm(R, S, bias=bias), 20); t1 = timeit(lambda: hip.gemm(R, S, residual=res), 20)
    out.append(f"[{M},{N},{K}] bias {t0 * 1e3:5.1f} us  residual {t1 * 1e3:5.1f} us")
print(" | ".join(out))
'''
for mt in ("128", "96", "64", "48"):
    env = dict(os.environ, UNIREC_HIP_LIB="tools/lab/libs/lab_pers.so", UR_PERS_MIN_TILES=mt)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("min_tiles", mt, ":", (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1], flush=True)

"""Lab: in-kernel timeline of one-shot 256x256 GEMM workgroups (lib built with -DUR_GEMM_STAMPS=1)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip, _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M, N = 131072, 2048
g = torch.Generator().manual_seed(0)
R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
S = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
res = torch.randn(M, N, generator=g).cuda().to(torch.bfloat16) if len(sys.argv) > 2 and sys.argv[2] == "res" else None
for _ in range(3): hip.gemm(R, S, out=out, residual=res)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
nb = 4096
buf = (ctypes.c_longlong * (nb * 8))()
rc = lib.ur_lab_gemm_stamps(buf, nb * 8)
t = torch.tensor(list(buf), dtype=torch.int64).view(nb, 8)
t0 = t[:, 0].min()
rel = (t - t0).double()
names = ["entry", "dma issued", "first data", "8-phase done", "K done", "stores issued"]
d = rel[:, 1:6] - rel[:, :5]
print(f"K={K}: kernel span {rel.max().item():.0f} cycles (s_memtime ticks); per block (median over {nb} blocks):")
for i in range(5):
    print(f"  {names[i]:>14s} -> {names[i+1]:<14s}: median {d[:, i].median().item():9.0f}  p10 {d[:, i].quantile(0.1).item():9.0f}  p90 {d[:, i].quantile(0.9).item():9.0f}")
life = rel[:, 5] - rel[:, 0]
print(f"  block life median {life.median().item():.0f}; first-wave blocks (<256) life median {life[:256].median().item():.0f}")

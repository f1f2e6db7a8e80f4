import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
DEV = "cuda"
M, N, K = 12288, 6144, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
load = len(sys.argv) > 2 and sys.argv[2] == "load"
g = torch.Generator().manual_seed(21)
Rm = torch.randint(-2, 3, (M, K), generator=g).float()
g = torch.Generator().manual_seed(22)
Sm = torch.randint(-2, 3, (N, K), generator=g).float()
R, S = Rm.to(DEV).to(torch.bfloat16), Sm.to(DEV).to(torch.bfloat16)
ref = (Rm.to(DEV).double() @ Sm.to(DEV).double().t()).float()
big = torch.randn(64 * 1024 * 1024, device=DEV)
s2 = torch.cuda.Stream()
for it in range(4):
    if load:
        with torch.cuda.stream(s2):
            big.mul_(1.0001)
    out = hip.gemm(R, S, out_f32=True)
    bad = (out != ref)
    print("lib", os.environ.get("UNIREC_HIP_LIB"), "K", K, "load", load, "it", it, "bad fraction", bad.float().mean().item())
    if bad.any():
        idx = bad.nonzero()
        print("  bad rows mod 256 hist(16-row bins):", torch.bincount(idx[:, 0] % 256, minlength=256).view(16, 16).sum(1).tolist())
        print("  bad cols mod 256 hist(16-col bins):", torch.bincount(idx[:, 1] % 256, minlength=256).view(16, 16).sum(1).tolist())
        tiles = torch.unique((idx[:, 0] // 256) * 1000 + idx[:, 1] // 256)
        print("  bad tiles:", len(tiles), tiles[:12].tolist())
        m, n = idx[0].tolist()
        part = (Rm[m].view(-1, 64) * Sm[n].view(-1, 64)).sum(1)
        print("  first bad", m, n, out[m, n].item(), ref[m, n].item(), "diff", out[m, n].item() - ref[m, n].item(), "k-tile partials", part.tolist())
        break

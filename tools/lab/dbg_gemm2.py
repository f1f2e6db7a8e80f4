import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
DEV = "cuda"
def ints(shape, lo, hi, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(torch.float32)
def run(M, N, K):
    Rm, Sm = ints((M, K), -2, 3, 21), ints((N, K), -2, 3, 22)
    R, S = Rm.to(DEV).to(torch.bfloat16).contiguous(), Sm.to(DEV).to(torch.bfloat16).contiguous()
    ref = (Rm.to(DEV).double() @ Sm.to(DEV).double().t()).float()
    big = torch.randn(64 * 1024 * 1024, device=DEV)
    s2 = torch.cuda.Stream()
    for it in range(4):
        with torch.cuda.stream(s2):
            big.mul_(1.0001)
        out = torch.full((M, N), 12345.0, device=DEV)
        hip.gemm(R, S, out=out)
        bad = (out != ref)
        nb = int(bad.sum().item())
        print(f"M={M} N={N} K={K} it={it}: bad={nb} unwritten={(out == 12345.0).sum().item()}")
        if nb:
            idx = bad.nonzero()
            print("  rows mod 256 (16-bins):", torch.bincount(idx[:, 0] % 256, minlength=256).view(16, 16).sum(1).tolist())
            print("  cols mod 256 (16-bins):", torch.bincount(idx[:, 1] % 256, minlength=256).view(16, 16).sum(1).tolist())
            tiles = torch.unique((idx[:, 0] // 256) * 1000 + idx[:, 1] // 256)
            print("  bad tiles:", len(tiles), tiles[:12].tolist())
            m, n = idx[0].tolist()
            part = (Rm[m].view(-1, 16) * Sm[n].view(-1, 16)).sum(1)
            print("  first bad", m, n, "got", out[m, n].item(), "ref", ref[m, n].item(), "diff", out[m, n].item() - ref[m, n].item())
            print("  16-k partials", part.tolist())
            return
    torch.cuda.synchronize()
for shp in [(8192, 8192, 1024), (16384, 4096, 4096), (8192 + 256, 8192, 192 + 64), (12288, 6144, 1024 + 48)]:
    run(*shp)

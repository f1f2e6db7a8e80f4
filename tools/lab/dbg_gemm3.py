import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
DEV = "cuda"
def ints(shape, lo, hi, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(torch.float32)
def bf(x): return x.to(DEV).to(torch.bfloat16).contiguous()
def run(M, N, K):
    Rm, Sm = ints((M, K), -2, 3, 21), ints((N, K), -2, 3, 22)
    R, S = bf(Rm), bf(Sm)
    ref = (Rm.to(DEV).double() @ Sm.to(DEV).double().t()).float()
    big = torch.randn(64 * 1024 * 1024, device=DEV)
    s2 = torch.cuda.Stream()
    for it in range(4):
        with torch.cuda.stream(s2):
            big.mul_(1.0001)
        out = hip.gemm(R, S, out_f32=True)
        if not torch.equal(out, ref):
            bad = out != ref
            idx = bad.nonzero()
            ref2 = (Rm.to(DEV).double() @ Sm.to(DEV).double().t()).float()
            out2 = hip.gemm(R, S, out_f32=True)
            print(f"M={M} N={N} K={K} it={it}: bad={int(bad.sum())}; ref==ref2 {torch.equal(ref, ref2)}; out==out2 {torch.equal(out, out2)}; out2==ref2 {torch.equal(out2, ref2)}; out==ref2 {torch.equal(out, ref2)}")
            print("  rows mod 256 (16-bins):", torch.bincount(idx[:, 0] % 256, minlength=256).view(16, 16).sum(1).tolist())
            print("  cols mod 256 (16-bins):", torch.bincount(idx[:, 1] % 256, minlength=256).view(16, 16).sum(1).tolist())
            print("  row range", idx[:, 0].min().item(), idx[:, 0].max().item(), "col range", idx[:, 1].min().item(), idx[:, 1].max().item())
            return
    print(f"M={M} N={N} K={K} ok")
    torch.cuda.synchronize()
    r = 16
    R2m, S2m = ints((M, r), -1, 2, 23), ints((N, r), -1, 2, 24)
    out16 = hip.gemm(R, S, R2=bf(R2m), S2=bf(S2m), alpha=1.0 / 64)
    ref2 = ((ref.double() + R2m.to(DEV).double() @ S2m.to(DEV).double().t()) / 64).to(torch.bfloat16)
    print("   lora part equal:", torch.equal(out16, ref2))
for shp in [(8192, 8192, 1024), (16384, 4096, 4096), (8192 + 256, 8192, 192 + 64), (12288, 6144, 1024 + 48)]:
    run(*shp)

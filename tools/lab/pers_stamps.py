"""Lab: in-kernel timeline of the persistent GEMM's n-th output tile per workgroup (library built with -DUR_PERS_STAMPS=n,
tools/lab/pers_stamps.sh).  Stamps: 0 tile start, 1..5 after K tiles 0..4, 6 K loop done, 7 epilogue stores issued."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip, _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
M = 131072
g = torch.Generator().manual_seed(0)
R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
S = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
hip.gemm_persistent_mode(1)
for _ in range(3): hip.gemm(R, S, out=out)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 256 * 2 * 8
buf = (ctypes.c_longlong * n)()
lib.ur_lab_pers_stamps(buf, n)
t = torch.tensor(list(buf), dtype=torch.int64).view(256, 2, 8).double()
d = t[:, :, 1:] - t[:, :, :-1]
names = ["tile start -> K tile 0 done", "K tile 1", "K tile 2", "K tile 3", "K tile 4", f"K tiles 5..{K // 64 - 1}", "epilogue (to last store issued)"]
for grp in (0, 1):
    print(f"K={K} N={N} wave group {grp} (cycles, median / p10 / p90 over 256 workgroups):")
    for i in range(7):
        x = d[:, grp, i]
        print(f"  {names[i]:>34s}: {x.median().item():8.0f} {x.quantile(0.1).item():8.0f} {x.quantile(0.9).item():8.0f}")
    x0 = t[0::8, grp, 0]             # the workgroups of ONE XCD (ids equal mod 8): s_memtime is per XCD
    print(f"  tile total {(t[:, grp, 7] - t[:, grp, 0]).median().item():.0f}; tile start of the 32 workgroups with id % 8 == 0, relative to the earliest (sorted):")
    print("   ", " ".join(f"{v:.0f}" for v in sorted((x0 - x0.min()).tolist())))

"""Lab: in-kernel timeline of attn_bwd_dkv2_kernel's 64-query tile loop (lib built with -DUR_DKV2_STAMPS=1):
cycle counter of wave 0 at the phase boundaries of tiles 4..11 of the first 256 workgroups."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip, _lib
B, S, nq, nkv, hd = 64, 2048, 16, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
q = qkv[..., :nq * hd].view(B, S, nq, hd); k = qkv[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = qkv[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
o, ctx = hip.attn_fwd(q, k, v, causal=True)
for _ in range(2):
    hip.attn_bwd(ctx, dout)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 256 * 8 * 8
buf = (ctypes.c_longlong * n)()
assert lib.ur_lab_attn_stamps(buf, n) == 0
t = torch.tensor(list(buf), dtype=torch.int64).view(256, 8, 8).double()
names = ["top", "S/dP a", "S/dP b", "soft a", "load next", "dV/dK a + soft b", "dV/dK b", "barrier"]
d = t[:, :, 1:] - t[:, :, :-1]
ok = (t[:, :, 7] > t[:, :, 0]) & (t[:, :, 0] > 0)
print(f"tiles with stamps: {int(ok.sum())}; per tile: median {(t[:, :, 7] - t[:, :, 0])[ok].median().item():.0f} cycles; tile-to-tile (top -> next top) median "
      f"{(t[:, 1:, 0] - t[:, :-1, 0])[ok[:, 1:] & ok[:, :-1]].median().item():.0f}")
w = t[:, :, 1]
okw = ok[:, 1:] & ok[:, :-1] & (w[:, 1:] > w[:, :-1])
if okw.any():
    ghz = ((t[:, 1:, 0] - t[:, :-1, 0])[okw] / ((w[:, 1:] - w[:, :-1])[okw] * 10.0))
    print(f"shader clock during the loop (cycle counter / 100 MHz wall clock): median {ghz.median().item():.2f} GHz  p10 {ghz.quantile(0.1).item():.2f}  p90 {ghz.quantile(0.9).item():.2f}")
if os.environ.get("UR_STAMPS_V2", "1") == "1":       # the three-stream fast path stamps points 0, 2, 6, 7 only
    for a, b_, nm in ((0, 2, "top -> S/dP (streams 1+2) done"), (2, 6, "stream 3 (dV/dK)"), (6, 7, "barrier")):
        x = (t[:, :, b_] - t[:, :, a])[ok]
        print(f"  {nm:>34s}: median {x.median().item():7.0f}  p10 {x.quantile(0.1).item():7.0f}  p90 {x.quantile(0.9).item():7.0f}")
else:
    for i in range(7):
        x = d[:, :, i][ok]
        print(f"  {names[i]:>18s} -> {names[i + 1]:<18s}: median {x.median().item():7.0f}  p10 {x.quantile(0.1).item():7.0f}  p90 {x.quantile(0.9).item():7.0f}")

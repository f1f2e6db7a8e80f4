"""reads the per-wave cycle accumulators of a stamps build (tools/lab/c128_variants.sh stamps; UNIREC_HIP_LIB=tools/lab/libs/c128_stamps.so)
after one dense causal forward launch: where a wave's cycles go (ring wait, bodies by kind, C++ prologue / epilogue)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from unirec_amd import hip, _lib
B = int(os.environ.get("B", 64)); S = int(os.environ.get("S", 2048))
nq, nkv, hd = 16, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
q = qkv[..., :nq * hd].view(B, S, nq, hd); k = qkv[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = qkv[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
WHAT = os.environ.get("WHAT", "fwd")
o, ctx = hip.attn_fwd(q, k, v, causal=True)
dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
for _ in range(3):
    if WHAT == "fwd":
        hip.attn_fwd(q, k, v, causal=True)
    else:
        hip.attn_bwd(ctx, dout)
torch.cuda.synchronize()
lib = _lib.load()
n = 8192 * 4 * 32
buf = (ctypes.c_uint32 * n)()
rc = lib.ur_lab_c128_stamps(buf, n)
assert rc == 0, rc
a = np.frombuffer(buf, np.uint32).reshape(8192, 4, 32).astype(np.float64)
nwg = min(8192, B * nq * (S // 256))
a = a[:nwg]
names = ["ring wait", "PRO", "STEADY", "LAST", "EPI", "SKIP"]
print(f"{nwg} workgroups; cycles per wave (mean over all waves) and per occurrence")
tot_asm = a[:, :, 13].mean()
for i, nm in enumerate(names):
    cyc, cnt = a[:, :, 2 * i], a[:, :, 2 * i + 1]
    print(f"  {nm:10s} total {cyc.mean():10.0f}  count {cnt.mean():6.2f}  per occurrence {cyc.sum() / max(1, cnt.sum()):8.0f}")
print(f"  C++ prologue {a[:, :, 12].mean():8.0f}   asm block {tot_asm:9.0f}   C++ epilogue {a[:, :, 14].mean():8.0f}")
if WHAT == "fwd":
    print("  after the loop: advance %.0f | barrier %.0f | key state + epilogue half 0 %.0f | request (q loads + DMA statement) %.0f | epilogue half 1 %.0f" % tuple(a[:, :, 16 + i].mean() for i in range(5)))
    for w in range(4):
        print("    wave %d: barrier %.0f  half0 %.0f  request %.0f  half1 %.0f" % (w, a[:, w, 17].mean(), a[:, w, 18].mean(), a[:, w, 19].mean(), a[:, w, 20].mean()))
else:
    print("  (dQ: 'C++ prologue' = row loads + delta of this block; 'asm block' = pack + loop)")
    print("  after the loop: advance %.0f | barrier %.0f | key state + request (DMA statement) %.0f | dQ stores %.0f" % tuple(a[:, :, 16 + i].mean() for i in range(4)))
    for w in range(4):
        print("    wave %d: loads %.0f  asm %.0f  barrier %.0f  request %.0f  stores %.0f" % (w, a[:, w, 12].mean(), a[:, w, 13].mean(), a[:, w, 17].mean(), a[:, w, 18].mean(), a[:, w, 19].mean()))
for w in range(4):
    print(f"  wave {w}: STEADY per occurrence {a[:, w, 4].sum() / max(1, a[:, w, 5].sum()):7.0f}  ring wait per iteration {a[:, w, 0].sum() / max(1, a[:, w, 1].sum()):6.0f}  LAST {a[:, w, 6].sum() / max(1, a[:, w, 7].sum()):6.0f} EPI {a[:, w, 8].sum() / max(1, a[:, w, 9].sum()):6.0f}")
if WHAT != "fwd":
    print("  in-kernel clock over a block (shader cycles / 100 MHz ticks): %.0f MHz" % (100.0 * a[:, :, 21].sum() / max(1.0, a[:, :, 22].sum())))
xs = a[:, 0, 15]
for x in sorted(set(xs.astype(int))):
    m = xs == x
    print(f"  block x={x}: asm {a[m][:, :, 13].mean():9.0f}  prologue {a[m][:, :, 12].mean():7.0f}  epilogue {a[m][:, :, 14].mean():7.0f}")

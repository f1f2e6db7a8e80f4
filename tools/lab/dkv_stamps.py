"""per-phase cycle sums of the persistent dK/dV kernel (stamps build: tools/lab/c128_variants.sh stamps; UNIREC_HIP_LIB=tools/lab/libs/c128_stamps.so).
The LAST launch of ur_attn_bwd is the dK/dV kernel, so its words are what the buffer holds."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from unirec_amd import hip, _lib
B, S, nq, nkv, hd = 64, int(os.environ.get("S", 2048)), 16, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
q = qkv[..., :nq * hd].view(B, S, nq, hd); k = qkv[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = qkv[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
if os.environ.get("DKV_PERSIST") == "0":
    hip.attn_mode(hip.ATTN_MODE_DKV_PERSIST, 0)
o, ctx = hip.attn_fwd(q, k, v, causal=True)
for _ in range(3):
    hip.attn_bwd(ctx, dout)
torch.cuda.synchronize()
lib = _lib.load()
n = 8192 * 4 * 32
buf = (ctypes.c_uint32 * n)()
assert lib.ur_lab_c128_stamps(buf, n) == 0
a = np.frombuffer(buf, np.uint32).reshape(8192, 4, 32)[:256].astype(np.float64)
assert (a[:, :, 6] == 0xD0C5).all(), "not a dK/dV stamps record"
nb = a[:, :, 5]
names = ["K / V row loads + pack", "generated loop (incl. entry: 2 more tiles requested, 128 accumulators zeroed)", "draw (2 barriers, queue atomic, first-tile request)", "dV store", "dK store (+ RoPE backward when fused)"]
tot = a[:, :, :5].sum(axis=2)
print(f"blocks per workgroup {nb.mean():.1f} (min {nb.min():.0f}, max {nb.max():.0f}); cycles per wave over the launch {tot.mean():.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:90s} {a[:, :, i].mean():10.0f}  ({100 * a[:, :, i].mean() / tot.mean():4.1f} %)  per block {a[:, :, i].sum() / nb.sum():8.0f}")
for w in range(4):
    print(f"  wave {w}: loop {a[:, w, 1].mean():.0f}  draw (waits for the slowest wave) {a[:, w, 2].mean():.0f}")
if os.environ.get("DKV_PERSIST") == "0":      # (this tool's own switch: it sets ur_attn_mode(UR_ATTN_MODE_DKV_PERSIST, 0) before the launches)
    # one record per key block: the generated loop's own accumulators (words 8..13: ring wait cycles / count, body cycles / count)
    r = np.frombuffer(buf, np.uint32).reshape(8192, 4, 32).astype(np.float64)
    ok = r[:, :, 6] == 0xD0C5
    wait, nwait, body, nbody = r[:, :, 8][ok], r[:, :, 9][ok], r[:, :, 10][ok], r[:, :, 11][ok]
    loop = r[:, :, 1][ok]
    print(f"per key block (one workgroup each): loop statement {loop.mean():.0f} cycles = bodies {body.mean():.0f} ({body.sum() / nbody.sum():.0f} per iteration, "
          f"{nbody.mean():.1f} iterations) + ring wait / barrier {wait.mean():.0f} ({wait.sum() / nwait.sum():.0f} per iteration) + entry / exit {loop.mean() - body.mean() - wait.mean():.0f}")
    for w in range(4):
        m = ok[:, w]
        print(f"  wave {w}: body per iteration {r[:, w, 10][m].sum() / r[:, w, 11][m].sum():.0f}  wait per iteration {r[:, w, 8][m].sum() / r[:, w, 9][m].sum():.0f}")

"""same-box A/B of the three causal attention launches (dense B 64 S 2048): alternates library builds in child processes, N rounds"""
import os, subprocess, sys
libs = sys.argv[1:]
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from unirec_amd import hip
B, S, nq, nkv, hd = 64, 2048, 16, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
q = qkv[..., :nq * hd].view(B, S, nq, hd); k = qkv[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = qkv[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
o, ctx = hip.attn_fwd(q, k, v, causal=True)
def t(fn, n=12):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]
print("%.3f %.3f" % (t(lambda: hip.attn_fwd(q, k, v, causal=True)), t(lambda: hip.attn_bwd(ctx, dout))))
'''
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ)
        if l != "product":
            env["UNIREC_HIP_LIB"] = l
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res[l].append(tuple(float(x) for x in out.split()))
for l in libs:
    f = sorted(x[0] for x in res[l]); b = sorted(x[1] for x in res[l])
    print(f"{l:40s} fwd median {f[1]:.3f} ms  bwd median {b[1]:.3f} ms   rounds {res[l]}")

"""forward-only timing of the causal head_dim-128 attention launch (dense, B 64, S 2048 by default): min and median of N launches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
B = int(os.environ.get("B", 64)); S = int(os.environ.get("S", 2048)); N = int(os.environ.get("N", 20))
nq, nkv, hd = 16, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
q = qkv[..., :nq * hd].view(B, S, nq, hd); k = qkv[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = qkv[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
for _ in range(3):
    hip.attn_fwd(q, k, v, causal=True)
torch.cuda.synchronize()
ts = []
for _ in range(N):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.attn_fwd(q, k, v, causal=True); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
fl = 4 * B * nq * S * S * hd / 2
print(f"{os.environ.get('UNIREC_HIP_LIB', 'product'):50s} fwd min {ts[0]:.3f} ms  median {ts[len(ts)//2]:.3f} ms  ({fl / ts[len(ts)//2] / 1e9:.0f} TFLOP/s)", flush=True)

"""time of the dK/dV kernel alone (rocprofv3-free: events around ur_attn_bwd minus nothing -- prints the whole backward; compare variants)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from unirec_amd import hip
B, S, nq, nkv, hd = 64, 2048, 16, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
q = qkv[..., :nq * hd].view(B, S, nq, hd); k = qkv[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = qkv[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
o, ctx = hip.attn_fwd(q, k, v, causal=True)
ts = []
for _ in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.attn_bwd(ctx, dout); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"{os.environ.get('UNIREC_HIP_LIB', 'product'):40s} bwd median {ts[len(ts)//2]:.3f} ms")

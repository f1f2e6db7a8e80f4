"""a few forward + backward launches of the causal head_dim-128 attention (dense, B 64, S 2048) for counter passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
B, S, nq, nkv, hd = 64, 2048, 16, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
q = qkv[..., :nq * hd].view(B, S, nq, hd); k = qkv[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = qkv[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
for _ in range(int(os.environ.get("N", 3))):
    o, ctx = hip.attn_fwd(q, k, v, causal=True)
    hip.attn_bwd(ctx, dout)
torch.cuda.synchronize()
print("done")

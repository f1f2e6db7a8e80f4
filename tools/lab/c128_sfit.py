"""fixed cost per workgroup / block vs cost per tile of the three generated attention kernels: the same token count at S = 1024, 2048, 4096
(run under rocprofv3 --kernel-trace --stats: tools/lab/c128_sfit.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
S = int(os.environ.get("S", 2048)); B = 131072 // S
nq, nkv, hd = 16, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
q = qkv[..., :nq * hd].view(B, S, nq, hd); k = qkv[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = qkv[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
for _ in range(4):
    o, ctx = hip.attn_fwd(q, k, v, causal=True)
    hip.attn_bwd(ctx, dout)
torch.cuda.synchronize()

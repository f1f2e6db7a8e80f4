#!/usr/bin/env python3
"""Lab: the few-query dK/dV kernel at the C3 cross-attention shape (B 512, 16 heads, 64 queries x 1600 keys, ragged masks, dropout 0.1):
HIP-event time of ur_attn_bwd (dQ + dK/dV + column sums).  Used with the ablated libraries of tools/lab/lib_variant.sh attn ... -DUR_FEWQ_ABLATE=n."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
B, nh, Sq, Sk = 512, 16, 64, 1600
g = torch.Generator().manual_seed(0)
q = (torch.randn(B, Sq, nh, 64, generator=g) * 0.5).cuda().to(torch.bfloat16)
kv = (torch.randn(B, Sk, 2, nh, 64, generator=g) * 0.5).cuda().to(torch.bfloat16)
lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
km = (torch.arange(Sk)[None, :] < lens[:, None]).to(torch.uint8).cuda()
dout = torch.randn(B, Sq, nh, 64, generator=g).cuda().to(torch.bfloat16)
o, ctx = hip.attn_fwd(q, kv[:, :, 0], kv[:, :, 1], causal=False, key_mask=km, dropout_p=0.1, seed=3)
dkv = torch.empty_like(kv)
def run():
    hip.attn_bwd(ctx, dout, dk=dkv[:, :, 0], dv=dkv[:, :, 1])
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"attn_bwd (dQ + few-query dK/dV): {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")

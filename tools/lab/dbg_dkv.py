import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
S = int(sys.argv[1]) if len(sys.argv) > 1 else 130
B, nq, nkv, hd = 2, 4, 2, 128
g = torch.Generator().manual_seed(S)
buf = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
q = buf[..., :nq * hd].view(B, S, nq, hd); k = buf[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = buf[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
o, ctx = hip.attn_fwd(q, k, v, causal=True)
dq, dk, dv = hip.attn_bwd(ctx, dout)
torch.cuda.synchronize()
for name, t in (("dk", dk), ("dv", dv)):
    bad = ~torch.isfinite(t.float())
    print(name, "non-finite:", int(bad.sum()))
    if bad.any():
        idx = bad.nonzero()
        print("  batches", idx[:, 0].unique().tolist(), "keys", idx[:, 1].unique().tolist()[:40], "heads", idx[:, 2].unique().tolist(), "dims", idx[:, 3].unique().tolist()[:16])

"""lab: digests of the causal head_dim-128 forward / backward over seeds, lengths and key masks, each launched several times (run-to-run
determinism), for a bit-for-bit comparison of two library builds whose MFMA order is the same (a re-placed schedule must not change a bit):
    for l in a b; do UNIREC_HIP_LIB=tools/lab/libs/$l.so python tools/lab/attn_hash.py > /tmp/$l.txt; done; diff /tmp/a.txt /tmp/b.txt"""
import hashlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
from unirec_amd import hip
nq, nkv, hd = 16, 8, 128
REP = int(os.environ.get("REP", 3))


def dig(t):
    return hashlib.sha1(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:12]


for S, B in ((512, 16), (1024, 8), (2048, 16), (4096, 4)):
    for seed in range(3):
        for mask in ("none", "left", "holes"):
            g = torch.Generator().manual_seed(1000 * S + seed)
            buf = torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
            q = buf[..., :nq * hd].view(B, S, nq, hd); k = buf[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = buf[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
            dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
            km = None
            if mask == "left":
                km = torch.ones(B, S, dtype=torch.uint8)
                for b in range(1, B):
                    km[b, :int(torch.randint(1, S // 3, (1,), generator=g))] = 0
                km = km.cuda()
            elif mask == "holes":
                km = (torch.rand(B, S, generator=g) < 0.85).to(torch.uint8); km[:, 0] = 1; km = km.cuda()
            seen = set()
            for _ in range(REP):
                o, ctx = hip.attn_fwd(q, k, v, causal=True, key_mask=km)
                dq, dk, dv = hip.attn_bwd(ctx, dout)
                torch.cuda.synchronize()
                seen.add((dig(o), dig(dq), dig(dk), dig(dv)))
            print(S, B, seed, mask, "DETERMINISTIC" if len(seen) == 1 else "NONDETERMINISTIC x%d" % len(seen), sorted(seen)[0])

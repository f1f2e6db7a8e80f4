"""first contact of the generated forward kernel with the hardware: tiny shapes first, each step printed before the next starts"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from test_gpu_attn_c128 import _inputs, _ref

def one(B, S, nq, nkv, seed=0, pad=0):
    q, k, v = _inputs(B, S, nq, nkv, seed)
    km = None
    if pad:
        km = torch.ones(B, S, dtype=torch.uint8); km[:, :pad] = 0; km = km.cuda()
    print(f"launch B={B} S={S} nq={nq} nkv={nkv} pad={pad}", flush=True)
    o, ctx = hip.attn_fwd(q, k, v, causal=True, key_mask=km)
    torch.cuda.synchronize()
    ref = _ref(q, k, v, km)
    err = (o.float() - ref).abs().max().item()
    print(f"   done: max err {err:.5f} (ref max {ref.abs().max().item():.3f}) finite={bool(torch.isfinite(o.float()).all())}", flush=True)
    return err

one(1, 256, 2, 1)
one(1, 512, 2, 1)
one(2, 1024, 4, 2)
one(2, 1024, 4, 2, pad=100)
one(2, 2048, 16, 8)
print("OK", flush=True)

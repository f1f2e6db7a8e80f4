import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from unirec_amd import hip
import test_gpu_attention as T
B,S,nq,nkv,hd=1,256,2,1,128
q, k, v = T._randn((B, S, nq, hd), 0), T._randn((B, S, nkv, hd), 1), T._randn((B, S, nkv, hd), 2)
dout = T._randn((B, S, nq, hd), 3)
o, ctx = hip.attn_fwd(q, k, v, causal=True)
dq, dk, dv = hip.attn_bwd(ctx, dout)
qf, kf, vf = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
ref = T._ref(qf, kf, vf, None, True); ref.backward(dout.float())
for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
    e = (got.float() - want).abs().amax(dim=(0,2,3))
    print(name, "max err by token block of 32:", [round(e[i:i+32].max().item(),3) for i in range(0,S,32)])

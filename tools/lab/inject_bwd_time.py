import sys, os
sys.path.insert(0, os.getcwd())
import torch
from unirec_amd import hip
B, S, D, T = 64, 2048, 1024, 100
g = torch.Generator().manual_seed(0)
ids = torch.randint(0, 4096, (B, S), generator=g)
for b in range(B):
    pos = torch.randperm(S, generator=g)[:T]
    ids[b, pos] = 4096 + torch.arange(T)
ids[3, 7] = 4096 + 5     # a duplicate: the gradient sums
ids = ids.cuda()
dx = torch.randn(B, S, D, generator=g).cuda().to(torch.bfloat16)
import inspect
fn = [n for n in dir(hip) if "inject" in n]
print(fn)
out = hip.inject_bwd(dx, ids, 4096, T)
ref = torch.zeros(B, T, D, device="cuda")
m = (ids >= 4096)
bi, si = m.nonzero(as_tuple=True)
ref.index_put_((bi, ids[bi, si] - 4096), dx[bi, si].float(), accumulate=True)
print("max err", (out.float() - ref).abs().max().item())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): hip.inject_bwd(dx, ids, 4096, T)
e1.record(); torch.cuda.synchronize(); print("us per launch", e0.elapsed_time(e1) / 20 * 1e3)

"""lab: the item Q-Former of the C4 step alone (3200 items, Q 2, H 1024, 12 layers, dropout on): forward / backward ms, side stream on / off"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from unirec_amd.qformer_utils import QFormerForItemRepresentation
import unirec_amd.qformer as qformer
torch.manual_seed(0)
m = QFormerForItemRepresentation(hidden_size=1024, num_hidden_layers=12, num_attention_heads=16, intermediate_size=4096, num_query_tokens=2,
                                 field_embedding_dim=1024, num_fields=14, dropout=0.2).cuda().train()
g = torch.Generator().manual_seed(1)
x = torch.randn(3200, 14, 1024, generator=g).cuda(); mk = (torch.rand(3200, 14, generator=g) < 0.8).long(); mk[:, 0] = 1; mk = mk.cuda()
dout = torch.randn(3200, 2, 1024, generator=g).cuda()
def run(n=10):
    tf = tb = 0.0
    for i in range(n + 3):
        m.zero_grad(set_to_none=True)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); out = m(x, mk)["query_outputs"]; e[1].record(); out.backward(dout); e[2].record(); torch.cuda.synchronize()
        if i >= 3: tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    return tf / n, tb / n
for side in (True, False, True, False):
    qformer._DW_SIDE = side
    t0 = time.perf_counter(); f, b = run(); host = (time.perf_counter() - t0) / 13
    print(f"side stream {side}: forward {f:.2f} ms  backward {b:.2f} ms  (host loop {host * 1e3:.2f} ms per step)")

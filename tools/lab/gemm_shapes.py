"""Lab: every ur_gemm launch of one C4 joint step, aggregated by shape (HIP events on the launching stream)."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from unirec_amd import hip
from unirec_amd.joint import InfoNCELoss
args = argparse.Namespace(layers=28, hist=50, seq=2048, pool=1000, no_dropout=False, lora_dropout=0.1, user_tokens=False)
dev = torch.device("cuda")
model, qf, cfg, (Qi, F, E, D) = bench.build(args, dev)
b = bench.make_batch(64, 50, 2048, 1000, F, E, D, Qi, model.first_special_id, model.first_special_id, 1, dev)
loss_fn = InfoNCELoss()
def step():
    u = model(b["input_ids"], b["attention_mask"], b["history_field_embeddings"], b["history_attention_mask"])
    loss_fn(u, b["positive_item_embeddings"], b["negative_item_embeddings"]).backward()
for _ in range(2): step()
torch.cuda.synchronize()
hip.PROFILE = []
step(); torch.cuda.synchronize()
prof, hip.PROFILE = hip.PROFILE, None
agg = collections.defaultdict(lambda: [0, 0.0])
for (e0, e1, rk, sk, f32, M, N, K, split, epi) in prof:
    k = (M, N, K, rk, sk, f32, split, epi)
    agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"{len(prof)} launches, {tot:.1f} ms")
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    M, N, K, rk, sk, f32, split, epi = k
    print(f"M={M:7d} N={N:5d} K={K:7d} rk={rk} sk={sk} f32={f32} split={split:3d} epi={epi}: {n:4d} x {ms / n * 1e3:8.1f} us = {ms:7.2f} ms  {2.0 * M * N * K * n / ms / 1e9:7.1f} TFLOP/s")

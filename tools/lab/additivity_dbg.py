"""debug: per-parameter relative error of the gradient additivity property at full size (tests/test_gpu_fullsize.py)"""
import argparse, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import bench
from test_gpu_fullsize import _sub, _step
DEV = "cuda"
args = argparse.Namespace(layers=int(os.environ.get("LAYERS", 28)), hist=50, seq=2048, pool=1000, no_dropout=True, lora_dropout=0.0)
model, qf, cfg, (Qi, F, E, D) = bench.build(args, torch.device(DEV))
batch = bench.make_batch(64, args.hist, args.seq, args.pool, F, E, D, Qi, model.first_special_id, model.first_special_id, 4321, DEV)
model.train()
u, loss, g = _step(model, qf, batch)
_, la, ga = _step(model, qf, _sub(batch, slice(0, 32)), 0.5)
_, lb, gb = _step(model, qf, _sub(batch, slice(32, 64)), 0.5)
rows = []
for k in g:
    s = ga[k].float() + gb[k].float()
    rows.append((((s - g[k].float()).norm() / (g[k].float().norm() + 1e-20)).item(), k, g[k].float().norm().item(), s.norm().item()))
rows.sort(reverse=True)
for r in rows[:24]: print("%.4g  %s  |g| %.4g  |ga+gb| %.4g" % r)
print("...")
for r in rows[-4:]: print("%.4g  %s  |g| %.4g  |ga+gb| %.4g" % r)

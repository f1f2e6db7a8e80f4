"""Lab: does gloo's all-reduce of several large CUDA tensors (async, N ranks on one GPU) complete?  Isolates the
transport from the bench when rehearsing the multi-rank launch on a single-GPU box."""
import os, sys, time
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
sizes = [int(s) for s in sys.argv[1:]] or [40_000_000, 60_000_000, 60_000_000, 20_000_000]
bufs = [torch.full((n,), float(rank + 1), device=dev) for n in sizes]
t0 = time.time()
works = [dist.all_reduce(b, async_op=True) for b in bufs]
for w in works: w.wait()
torch.cuda.synchronize()
ok = all(abs(b[0].item() - world * (world + 1) / 2) < 1e-3 for b in bufs)
print(f"rank {rank}/{world}: {len(bufs)} async all-reduces of CUDA tensors done in {time.time() - t0:.1f} s, ok={ok}", flush=True)
dist.barrier()

"""Lab: column sums of wide bf16 matrices (bias gradients), GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
from tools.kernel_bench import timeit
for (M, N) in [(819200, 4096), (819200, 2048), (131072, 6144), (32768, 3072), (3584, 9216)]:
    x = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    t = timeit(lambda: hip.colsum(x), 10)
    print(f"[{M}, {N}]: {t * 1e3:8.1f} us  {M * N * 2 / t / 1e6:8.1f} GB/s", flush=True)

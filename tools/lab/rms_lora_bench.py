"""Lab: rms_lora (RMSNorm + the adapters' down projections in one pass) at the C4 shapes: us per launch and TB/s over x read + h written"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from unirec_amd import hip
M, D, r = 131072, 1024, 16
g = torch.Generator().manual_seed(0)
x = torch.randn(M, D, generator=g).cuda().to(torch.bfloat16)
w = torch.ones(D, device="cuda")
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for nad in (3, 2):
    U = [(torch.randn(r, D, generator=g) * 0.1).cuda().to(torch.bfloat16) for _ in range(nad)]
    bits = hip.lora_dropout_bits(1, 0.1, M, D, nad, "cuda")
    for b in (bits, None):
        ms = timeit(lambda: hip.rmsnorm_lora_fwd(x, w, 1e-6, U, 2.0, bits=b))
        print(f"{os.environ.get('UNIREC_HIP_LIB', 'product'):36s} rms_lora nad={nad} masked={b is not None}: {ms*1e3:7.1f} us  {2 * x.numel() * 2 / ms / 1e9:5.2f} TB/s (x + h)")

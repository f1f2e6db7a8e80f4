"""Lab: the rank-16 LoRA side kernels at every shape the C4 decoder layer launches them with (M = 64 x 2048 tokens), time and
TB/s of the activation each streams once.  Usage: python tools/lab/lora_shapes_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
M, r = 131072, 16
g = torch.Generator().manual_seed(0)
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, N, cols in (("down/o (N=1024, 1 adapter)", 1024, [(0, 1024)]), ("gate|up (N=6144, 2)", 6144, [(0, 3072), (3072, 3072)]), ("q|k|v (N=4096, 3)", 4096, [(0, 2048), (2048, 1024), (3072, 1024)])):
    dy = torch.randn(M, N, generator=g).cuda().to(torch.bfloat16)
    t = torch.randn(M, r * len(cols), generator=g).cuda().to(torch.bfloat16)
    Bt = [(torch.randn(r, n, generator=g) * 0.1).cuda().to(torch.bfloat16) for _, n in cols]
    gB = torch.empty(N, r, device="cuda")
    ms = timeit(lambda: hip.lora_bgrad(dy, t, Bt, cols, gB))
    print(f"lora_bgrad {name}: {ms*1e3:7.1f} us  {dy.numel()*2/ms/1e9:6.2f} TB/s")
for name, N, nad in (("o_proj x=att (N=2048)", 2048, 1), ("down x=act (N=3072)", 3072, 1)):
    x = torch.randn(M, N, generator=g).cuda().to(torch.bfloat16)
    A = [(torch.randn(r, N, generator=g) * 0.1).cuda().to(torch.bfloat16)]
    bits = hip.lora_dropout_bits(1, 0.1, M, N, 1, "cuda")
    tb = torch.randn(M, r, generator=g).cuda().to(torch.bfloat16)
    gA = torch.empty(r, N, device="cuda")
    ms = timeit(lambda: hip.lora_project(x, A, bits=bits))
    print(f"lora_project {name}: {ms*1e3:7.1f} us  {x.numel()*2/ms/1e9:6.2f} TB/s")
    ms = timeit(lambda: hip.lora_reduce(x, tb, gA, nad=1, bits=bits))
    print(f"lora_reduce  {name}: {ms*1e3:7.1f} us  {x.numel()*2/ms/1e9:6.2f} TB/s")
    bt = hip.lora_bits_transpose(bits, N)
    ms = timeit(lambda: hip.lora_reduce(x, tb, gA, nad=1, bits=bits, bits_t=bt))
    print(f"lora_reduce  {name} (ring, token-packed flags): {ms*1e3:7.1f} us  {x.numel()*2/ms/1e9:6.2f} TB/s")
    ms = timeit(lambda: hip.lora_bits_transpose(bits, N, out=bt))
    print(f"  lora_bits_transpose: {ms*1e3:7.1f} us")
for name, nad in (("q|k|v x=h (N=1024, 3 masks)", 3), ("gate|up x=h2 (N=1024, 2 masks)", 2)):
    x = torch.randn(M, 1024, generator=g).cuda().to(torch.bfloat16)
    bits = hip.lora_dropout_bits(1, 0.1, M, 1024, nad, "cuda")
    tb = torch.randn(M, r * nad, generator=g).cuda().to(torch.bfloat16)
    gA = torch.empty(r * nad, 1024, device="cuda")
    ms = timeit(lambda: hip.lora_reduce(x, tb, gA, nad=nad, bits=bits))
    print(f"lora_reduce  {name}: {ms*1e3:7.1f} us  {x.numel()*2/ms/1e9:6.2f} TB/s")
    bt = hip.lora_bits_transpose(bits, 1024)
    ms = timeit(lambda: hip.lora_reduce(x, tb, gA, nad=nad, bits=bits, bits_t=bt))
    print(f"lora_reduce  {name} (ring, token-packed flags): {ms*1e3:7.1f} us  {x.numel()*2/ms/1e9:6.2f} TB/s")

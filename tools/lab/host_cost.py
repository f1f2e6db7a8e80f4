"""Lab: host-side cost of one wrapper call (the item stage is ~870 launches per 17 ms step: every microsecond of Python per launch is
5 % of the step on a slow host core).  Launches are asynchronous; the loop is timed without a device sync, on tiny operands."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
dev = "cuda"
R = torch.randn(256, 64, device=dev).to(torch.bfloat16); S = torch.randn(256, 64, device=dev).to(torch.bfloat16)
bias = torch.randn(256, device=dev); out = torch.empty(256, 256, device=dev, dtype=torch.bfloat16)
w = torch.ones(64, device=dev); b = torch.zeros(64, device=dev)
q = torch.randn(4, 32, 12, 64, device=dev).to(torch.bfloat16)


def t(name, fn, n=3000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name:44s} {dt / n * 1e6:7.2f} us per call", flush=True)


t("hip._stream()", hip._stream)
t("torch.empty((256,256), bf16)", lambda: torch.empty((256, 256), dtype=torch.bfloat16, device=R.device))
t("GemmArgs()", hip.GemmArgs)
t("R.data_ptr() + R.stride(0)", lambda: (R.data_ptr(), R.stride(0)))
t("hip.gemm(R, S, out=out)", lambda: hip.gemm(R, S, out=out))
t("hip.gemm(R, S, bias=bias)  (allocates)", lambda: hip.gemm(R, S, bias=bias))
t("hip.layernorm_fwd", lambda: hip.layernorm_fwd(R, w, b, 1e-5))
t("hip.attn_fwd", lambda: hip.attn_fwd(q, q, q, causal=False))
t("hip.colsum", lambda: hip.colsum(R))
t("torch add (eager op)", lambda: torch.add(bias, bias))

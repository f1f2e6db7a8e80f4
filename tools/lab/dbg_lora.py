import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip
torch.set_printoptions(linewidth=250, precision=2, sci_mode=False)
DEV = "cuda"
g = torch.Generator().manual_seed(3)
for (M, K) in ((64, 64), (64, 128), (128, 64), (300, 128)):
    x = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    tb = torch.randn(M, 16, generator=g).to(DEV).to(torch.bfloat16)
    for masked in (False, True):
        bits = hip.lora_dropout_bits(7, 0.3, M, K, 1, DEV) if masked else None
        keep = hip.lora_bits_to_keep(bits, K).float()[0] if masked else torch.ones(M, K, device=DEV)
        gA = torch.empty(16, K, device=DEV)
        hip.lora_reduce(x, tb, gA, nad=1, bits=bits)
        want = tb.float().t() @ (x.float() * keep)
        err = (gA - want).abs()
        print(f"M={M} K={K} masked={masked}: max err {err.max().item():.3f} (ref max {want.abs().max().item():.2f}); bad cols:",
              (err.max(0).values > 0.1).nonzero().flatten().tolist()[:40])

"""GPU: ur_gemm_grouped -- the weight-gradient token reductions of one Q-Former layer as ONE launch (autograd of the nn.Linear layers
of /root/reference/models/qformer.py:56-92,238-275: dW = dY^T X per weight) -- against a plain PyTorch fp32 reference of the same
products and against one ur_gemm per product."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from unirec_amd import hip  # noqa: E402
from unirec_amd._lib import UniRecHipError  # noqa: E402

DEV = "cuda"


def _products(K, shapes, seed, pad=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for (M, N) in shapes:
        dy = torch.randn(K, M + pad, generator=g).to(DEV).to(torch.bfloat16)[:, :M]       # (pad: a row stride above the row length)
        x = torch.randn(K, N + pad, generator=g).to(DEV).to(torch.bfloat16)[:, :N]
        out.append((dy, x, torch.full((M, N), float("nan"), device=DEV)))
    return out


def _check(prods, split_k):
    hip.gemm_grouped(prods, split_k=split_k)
    torch.cuda.synchronize()
    for dy, x, o in prods:
        ref = dy.float().t() @ x.float()
        K = dy.shape[0]
        tol = 2e-3 * (K ** 0.5)          # f32 accumulation of K bf16 products of unit-variance terms
        assert torch.isfinite(o).all()
        assert (o - ref).abs().max().item() <= tol, ((o - ref).abs().max().item(), tol)
        single = torch.empty_like(o)
        hip.gemm(dy, x, r_kcontig=False, s_kcontig=False, out=single, split_k=split_k)
        assert (o - single).abs().max().item() <= tol


@pytest.mark.parametrize("split_k", [1, 2, 4])
def test_one_item_qformer_layer(split_k):
    # C2: 8192 anchor rows, H 768, I 3072 -- out-proj, cross query, q|k|v, FFN up, FFN down
    _check(_products(8192, [(768, 768), (768, 768), (2304, 768), (3072, 768), (768, 3072)], 1), split_k)


def test_one_user_qformer_layer_and_full_group():
    # C3-shaped widths at a shorter token axis; eight products = the launch's capacity
    _check(_products(4096, [(1024, 1024)] * 4 + [(3072, 1024), (4096, 1024), (1024, 4096), (1024, 1024)], 2), 1)


@pytest.mark.parametrize("split_k", [1, 3])
def test_small_and_ragged_products_take_the_small_tile(split_k):
    # outputs below one big tile, edge tiles in both directions, padded row strides
    _check(_products(1000, [(64, 136), (136, 64), (264, 520), (8, 8)], 3, pad=8), split_k)


def test_bitwise_reproducible_and_equal_to_itself_in_any_group_order():
    prods = _products(8192, [(768, 768), (2304, 768), (768, 3072)], 4)
    hip.gemm_grouped(prods, split_k=2)
    first = [o.clone() for _, _, o in prods]
    hip.gemm_grouped(prods, split_k=2)
    rev = [(dy, x, torch.empty_like(o)) for dy, x, o in reversed(prods)]
    hip.gemm_grouped(rev, split_k=2)
    torch.cuda.synchronize()
    for (_, _, o), f in zip(prods, first):
        assert torch.equal(o, f)
    for (_, _, o), f in zip(reversed(rev), first):          # a product's result does not depend on its neighbours in the grid
        assert torch.equal(o, f)


def test_refuses_what_it_cannot_group():
    prods = _products(512, [(64, 64)] * 9, 5)
    with pytest.raises(ValueError):
        hip.gemm_grouped(prods)
    with pytest.raises(ValueError):
        hip.gemm_grouped([])
    a = _products(512, [(64, 64)], 6)[0]
    b = _products(256, [(64, 64)], 7)[0]
    with pytest.raises(UniRecHipError):                       # different token counts: not one kind
        hip.gemm_grouped([a, b])

"""GPU: every A/B switch of the product path, flipped ALONE, still meets the oracle on the joint step.

The switches (README "Switches") select alternative launch sequences -- separate passes instead of fused epilogues, the
compiler-scheduled attention kernels instead of the generated ones, the generic GEMM instead of the persistent one ... -- that the
bit-identity tests compare with EACH OTHER.  Here each alternative is compared with an independent source: the CPU oracle on
`joint_mid` (item Q-Former -> injection -> 2 decoder layers of the 0.6B shape, S 512, left padding -> InfoNCE; the fixture the
reference itself produced pins the oracle for this case: tests/test_oracle_golden_r2.py) with LoRA dropout ON and the kernels' own
masks fed in -- outputs, loss, LoRA dA / dB, and the gradient that reaches the Q-Former through every masked adapter input.
The oracle runs once; each switch costs one product step."""
import functools

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import qformer_ref as R  # noqa: E402
from oracle import qwen3_ref as Q  # noqa: E402
from oracle import weights as W  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.parity_utils import GRAD_REL, OUT_REL, assert_close  # noqa: E402

DEV = "cuda"
PDROP, STEP, SEED = 0.1, 5, 77
LORA_KEYS = ("layers.0.self_attn.q_proj.lora_A.weight", "layers.0.self_attn.k_proj.lora_A.weight", "layers.0.self_attn.v_proj.lora_B.weight",
             "layers.1.self_attn.o_proj.lora_A.weight", "layers.1.self_attn.o_proj.lora_B.weight", "layers.1.mlp.gate_proj.lora_A.weight",
             "layers.0.mlp.up_proj.lora_A.weight", "layers.0.mlp.up_proj.lora_B.weight", "layers.1.mlp.down_proj.lora_A.weight",
             "layers.0.mlp.down_proj.lora_B.weight")


def _case():
    # B = 8: M = 4096 tokens = 16 row tiles -- the q|k|v and gate|up launches reach the persistent GEMM (>= 128 output tiles), so its
    # fused q/k-norm + RoPE and SwiGLU epilogues are what the default path runs and what the switches turn off
    return dict(cases.MID["joint_mid"], B=8)


def _masks(bm, qc, B, S):
    from unirec_amd import hip
    M, D, I, NQ = B * S, qc.hidden_size, qc.intermediate_size, qc.num_attention_heads * qc.head_dim
    groups = {0: (("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"), D), 1: (("self_attn.o_proj",), NQ),
              2: (("mlp.gate_proj", "mlp.up_proj"), D), 3: (("mlp.down_proj",), I)}
    masks = {}
    for i in range(qc.num_hidden_layers):
        for g, (names, width) in groups.items():
            keep = hip.lora_bits_to_keep(hip.lora_dropout_bits(bm.lora_dropout_seed(STEP, i, g), PDROP, M, width, len(names), DEV), width).cpu()
            for slot, nm in enumerate(names):
                masks[f"layers.{i}.{nm}"] = keep[slot].view(B, S, width)
    return masks


@functools.lru_cache(maxsize=1)
def _oracle():
    """(user embeddings, loss, LoRA gradients, query-table gradient) of the oracle with the kernels' masks of (SEED, STEP)."""
    from tests.test_gpu_joint import _build_joint
    case = _case()
    c, qc = case["cfg"], cases.qwen_cfg(case)
    m, _ = _build_joint(case, use_lora=True, lora_seed=case["seed"] + 2)
    m.base_model.lora_seed = SEED
    ids, am, hfe, ham, pos, neg, nmask = cases.joint_inputs(case)
    B, S = ids.shape
    masks = _masks(m.base_model, qc, B, S)
    qc.lora_dropout = PDROP
    cfg = R.QFormerCfg(c["H"], c["L"], c["nh"], c["I"], c["Q"], c["E"], 2)
    PQ = {k: torch.from_numpy(v).requires_grad_(True) for k, v in W.fill_state_dict(R.item_qformer_shapes(cfg, c["F"]), case["seed"]).items()}
    PW = {k: torch.from_numpy(v) for k, v in W.fill_state_dict(Q.qwen3_shapes(qc, lora=False), case["seed"] + 1).items()}
    lsh = {k: s_ for k, s_ in Q.qwen3_shapes(qc, lora=True).items() if ".lora_" in k}
    PL = {k: torch.from_numpy(v).requires_grad_(True) for k, v in W.fill_state_dict(lsh, case["seed"] + 2).items()}
    hist = case["hist"]
    out = R.item_qformer_forward(PQ, cfg, torch.from_numpy(hfe).view(B * hist, c["F"], c["E"]), torch.from_numpy(ham).view(B * hist, c["F"]))
    toks = out["query_outputs"].view(B, hist, c["Q"], c["H"])
    ou = Q.joint_forward({**PW, **PL}, qc, torch.from_numpy(ids), torch.from_numpy(am), toks, case["first_special_id"], lora_masks=masks)
    ol = Q.infonce_loss(ou, torch.from_numpy(pos), torch.from_numpy(neg), torch.from_numpy(nmask))
    ol.backward()
    return (ou.detach().numpy(), ol.detach().numpy(), {k: PL[k].grad.numpy() for k in LORA_KEYS}, PQ["query_embeddings"].grad.numpy())


def _product_step_meets_the_oracle(what, recompute_mlp=False):
    from tests.test_gpu_joint import _build_joint
    from unirec_amd.joint import InfoNCELoss
    ou, ol, gl, gq = _oracle()
    case = _case()
    m, qf = _build_joint(case, use_lora=True, lora_seed=case["seed"] + 2)
    bm = m.base_model
    bm.config.lora_dropout = PDROP
    bm.lora_seed, bm._lora_step = SEED, STEP
    bm.recompute_mlp = recompute_mlp
    ids, am, hfe, ham, pos, neg, nmask = cases.joint_inputs(case)
    t = lambda a: torch.from_numpy(a).to(DEV)
    user = m(t(ids), t(am), t(hfe), t(ham))
    loss = InfoNCELoss()(user, t(pos), t(neg), t(nmask))
    loss.backward()
    torch.cuda.synchronize()
    print(what)
    assert_close(user, ou, OUT_REL, f"[{what}] user_embeddings")
    assert_close(loss, ol, OUT_REL, f"[{what}] loss")
    named = dict(bm.named_parameters())
    for k in LORA_KEYS:
        assert_close(named[k].grad, gl[k], GRAD_REL * 1.5, f"[{what}] grad/{k}")
    assert_close(dict(qf.named_parameters())["query_embeddings"].grad, gq, GRAD_REL * 1.5, f"[{what}] grad/query_embeddings")


def test_default_path_meets_the_oracle():
    _product_step_meets_the_oracle("defaults")


# module-level switches of unirec_amd.qwen3 / qformer (read from the environment at import): flipped on the module
MODULE_SWITCHES = {
    "UNIREC_MERGE_PROJ=0": [("qwen3", "_MERGE_PROJ", False)],
    "UNIREC_MERGE_PROJ=0 UNIREC_SWIGLU_FWD_FUSED=1": [("qwen3", "_MERGE_PROJ", False), ("qwen3", "_FUSE_SWIGLU_FWD", True)],
    "UNIREC_FUSE_NORM_LORA=0": [("qwen3", "_FUSE_NORM_LORA", False)],
    "UNIREC_FUSE_QK_ROPE=0": [("qwen3", "_FUSE_QK_ROPE", False)],
    "UNIREC_FUSE_SWIGLU_GEMM=0": [("qwen3", "_FUSE_SWIGLU_GEMM", False)],
    "UNIREC_FUSE_SWIGLU_GEMM=0 UNIREC_FUSE_SWIGLU_LORA=0": [("qwen3", "_FUSE_SWIGLU_GEMM", False), ("qwen3", "_FUSE_SWIGLU_LORA", False)],
    "UNIREC_QF_WT=0": [("qformer", "_USE_WT", False)],
    "UNIREC_KV_COLSUM=0": [("qformer", "_KV_COLSUM", False)],
    "UNIREC_QF_DW_STREAM=0": [("qformer", "_DW_SIDE", False)],
    "UNIREC_QF_DW_GROUPED=0": [("qformer", "_DW_GROUPED", False)],
}
# switches read on every call (Python layer and library)
ENV_SWITCHES = ["UNIREC_BITS_T=0", "UNIREC_BITS_ONE_EVENT=1", "UNIREC_PAD_ATT=0", "UNIREC_ROPE_BWD_FUSED=0", "UNIREC_ROPE_K_FUSED=0",
                "UNIREC_FUSE_QK_ROPE_OFF_AND_ROPE_BWD_FUSED=1"]
# kernel-selection words of the library (ur_attn_mode; it reads no environment variable)
ATTN_MODES = {"ur_attn_mode(C128, 0)": (1, 0), "ur_attn_mode(DKV_PERSIST, 0)": (2, 0), "ur_attn_mode(TINY, 0)": (0, 0)}


@pytest.mark.parametrize("name", list(MODULE_SWITCHES))
def test_module_switch_flipped_alone(name, monkeypatch):
    import unirec_amd.qformer as qformer
    import unirec_amd.qwen3 as qwen3
    mods = {"qwen3": qwen3, "qformer": qformer}
    for mod, attr, val in MODULE_SWITCHES[name]:
        assert getattr(mods[mod], attr) != val, f"{name}: already the default?"
        monkeypatch.setattr(mods[mod], attr, val)
    _product_step_meets_the_oracle(name)


@pytest.mark.parametrize("name", ENV_SWITCHES)
def test_call_time_switch_flipped_alone(name, monkeypatch):
    if name == "UNIREC_FUSE_QK_ROPE_OFF_AND_ROPE_BWD_FUSED=1":      # the dQ kernel's q-norm + RoPE backward from the RAW projection
        import unirec_amd.qwen3 as qwen3
        monkeypatch.setattr(qwen3, "_FUSE_QK_ROPE", False)
        monkeypatch.setenv("UNIREC_ROPE_BWD_FUSED", "1")
    else:
        k, v = name.split("=")
        monkeypatch.setenv(k, v)
    _product_step_meets_the_oracle(name)


@pytest.mark.parametrize("name", list(ATTN_MODES))
def test_attention_mode_flipped_alone(name):
    from unirec_amd import hip
    key, val = ATTN_MODES[name]
    with hip.attn_mode_set(key, val):
        _product_step_meets_the_oracle(name)
    assert hip.attn_mode(key, -2) == (3 if key == 0 else 1)


def test_recompute_mlp_and_the_generic_gemm():
    from unirec_amd import _lib
    _product_step_meets_the_oracle("recompute_mlp", recompute_mlp=True)
    lib = _lib.load()
    prev = lib.ur_gemm_persistent_mode(0)
    try:
        _product_step_meets_the_oracle("ur_gemm_persistent_mode(0)")
    finally:
        lib.ur_gemm_persistent_mode(prev)

"""Golden-case definitions + deterministic input generators (numpy only).

Shared by make_golden.py (reference side, build container only) and the parity tests
(oracle side / HIP side, anywhere).  Inputs are pure functions of the case dict, so the
.npz fixtures hold expected outputs only.
"""
import numpy as np

from oracle import weights as W
from oracle.qwen3_ref import Qwen3Cfg


def trim_like(g):
    """Fixture-size rule shared by generator and tests: 2-D arrays above 16384 elements keep rows [:32]."""
    g = np.asarray(g)
    if g.ndim == 2 and g.size > 16384:
        return g[:32].copy()
    return g.copy()


def _unit_rows(name, shape, seed):
    x = W.normal(name, shape, seed, std=1.0)
    return (x / np.linalg.norm(x, axis=-1, keepdims=True)).astype(np.float32)


# ------------------------------------------------------------------ item Q-Former -------------
def item_inputs(case):
    """field_embeddings [B,F,E] unit-norm rows; masks cover: all-ones (row 0), fully masked with
    zero vectors (row 1: what padded history slots look like, train_item_individual_token_joint.py:
    568-575), fully masked with NON-zero vectors (row 2: SURVEY I3 uniform-softmax probe),
    non-contiguous partial masks (rest); masked fields of rows >=3 are exact zeros
    (models/qformer_utils.py:116)."""
    c, s = case["cfg"], case["seed"]
    B, Fn, E = case["B"], c["F"], c["E"]
    x = _unit_rows("item/x", (B, Fn, E), s)
    mask = (W.uniform("item/mask", (B, Fn), s, 0, 1) < 0.7).astype(np.int64)
    mask[:, 0] |= (mask.sum(1) == 0)
    mask[0] = 1
    if B > 2:
        mask[1] = 0
        mask[2] = 0
    for b in range(B):
        if b != 2:
            x[b][mask[b] == 0] = 0.0
    return x, mask


def triplet_reps(case):
    c, s = case["cfg"], case["seed"]
    return (W.normal("item/pos", (case["B"], c["E"]), s, std=0.05),
            W.normal("item/neg", (case["B"], c["E"]), s, std=0.05))


def item_grad_keys(c, heads=True):
    L = c["L"]
    last_cross = max(i for i in range(L) if i % 2 == 0)
    keys = ["query_embeddings",
            "qformer.embeddings.LayerNorm.weight",
            "qformer.encoder.layer.0.attention.self.query.weight",
            "qformer.encoder.layer.0.attention.self.key.bias",
            "qformer.encoder.layer.0.attention.output.LayerNorm.bias",
            "qformer.encoder.layer.0.crossattention.self.key.weight",
            f"qformer.encoder.layer.{last_cross}.crossattention.self.value.weight",
            f"qformer.encoder.layer.{last_cross}.crossattention.output.dense.weight",
            f"qformer.encoder.layer.{L - 1}.intermediate_query.dense.weight",
            f"qformer.encoder.layer.{L - 1}.output_query.dense.bias",
            f"qformer.encoder.layer.{L - 1}.output_query.LayerNorm.weight"]
    if heads:
        keys += ["item_representation_head.weight", "reconstruction_head.weight", "reconstruction_head.bias",
                 "field_projection.weight", "field_projection.bias"]
    return keys


# ------------------------------------------------------------------ user Q-Former -------------
def user_inputs(case):
    """tokens [B,T,E] ~ 0.8*N(0,1), ragged right-padded masks (float, as the
    reference's collate builds them, training/user_qformer_training.py:155-161), padded rows zero."""
    c, s = case["cfg"], case["seed"]
    B, T, E = case["B"], case["T"], c["E"]
    x = W.normal("user/x", (B, T, E), s, std=0.8)
    lens = np.maximum(1, (W.uniform("user/len", (B,), s, 0.3, 1.0) * T).astype(np.int64))
    lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.float32)
    x = x * mask[..., None]
    tgt = W.normal("user/tgt", (B, c["n_pred"], E), s, std=0.8)
    return x, mask, tgt


def user_grad_keys(c):
    L = c["L"]
    return ["query_embeddings",
            "qformer.encoder.layer.0.crossattention.self.key.weight",
            f"qformer.encoder.layer.{L - 1}.crossattention.self.query.weight",
            f"qformer.encoder.layer.{L - 1}.crossattention.self.value.bias",
            f"qformer.encoder.layer.{L - 1}.output_query.dense.weight",
            "prediction_head.0.weight", "prediction_head.2.weight", "prediction_head.3.bias"]


# ------------------------------------------------------------------ joint ----------------------
def qwen_cfg(case) -> Qwen3Cfg:
    q = case["qwen"]
    return Qwen3Cfg(hidden_size=q["D"], num_hidden_layers=q["L"], num_attention_heads=q["nq"],
                    num_key_value_heads=q["nkv"], head_dim=q["hd"], intermediate_size=q["I"],
                    vocab_size=q["vocab"], rope_theta=1e6, rms_norm_eps=1e-6)


def _padding_mask(name, B, S, seed, side):
    npad = (W.uniform(name, (B,), seed, 0.0, 0.3) * S).astype(np.int64)
    npad[0] = 0
    am = np.ones((B, S), dtype=np.int64)
    for b in range(B):
        if npad[b]:
            if side == "left":
                am[b, :npad[b]] = 0
            else:
                am[b, S - npad[b]:] = 0
    return am


def joint_inputs(case):
    """Batch layout of MultiModalDataCollator (:295-323): input_ids with each special token exactly
    once in the non-pad region, attention_mask, history field embeddings/masks (some history slots
    fully masked zeros, :568-575), positive, negatives, negative_masks."""
    c, s = case["cfg"], case["seed"]
    B, S, hist, Qi = case["B"], case["S"], case["hist"], c["Q"]
    first = case["first_special_id"]
    am = _padding_mask("joint/pad", B, S, s, case["pad_side"])
    ids = (W.uniform("joint/ids", (B, S), s, 0, 1) * first).astype(np.int64).clip(0, first - 1)
    for b in range(B):
        real = np.nonzero(am[b])[0]
        rng = np.random.Generator(np.random.Philox(key=s * 1000 + b))
        slots = rng.choice(real, size=hist * Qi, replace=False)
        for t, p in enumerate(slots):
            ids[b, p] = first + t
    if case.get("drop_one_special", False):     # truncated prompt: a special token that never appears
        ids[B - 1][ids[B - 1] == first + hist * Qi - 1] = 1
    x = _unit_rows("joint/hfe", (B, hist, c["F"], c["E"]), s)
    hmask = (W.uniform("joint/hmask", (B, hist, c["F"]), s, 0, 1) < 0.8).astype(np.int64)
    hmask[..., 0] = 1
    hmask[1:, hist - 1, :] = 0                    # padded history slot
    x = x * hmask[..., None]
    pos = _unit_rows("joint/pos", (B, case["D"]), s)
    neg = _unit_rows("joint/neg", (B, case["N"], case["D"]), s)
    nmask = np.ones((B, case["N"]), dtype=bool)
    nmask[B - 1, case["N"] - 2:] = False
    neg[B - 1, case["N"] - 2:] = 0.0
    return ids, am, x.astype(np.float32), hmask, pos, neg, nmask


def qwen_inputs(case):
    q, s = case["qwen"], case["seed"]
    x = W.normal("qwen/x", (case["B"], case["S"], q["D"]), s, std=0.05)
    am = _padding_mask("qwen/pad", case["B"], case["S"], s, case["pad_side"])
    return x, am


def qwen_grad_keys():
    return ["layers.0.self_attn.q_proj.weight", "layers.0.self_attn.k_norm.weight",
            "layers.1.mlp.down_proj.weight", "layers.1.input_layernorm.weight", "norm.weight"]


# ------------------------------------------------------------------ case table ----------------
_TINYQ = dict(D=256, L=2, nq=4, nkv=2, hd=128, I=512, vocab=96)

ALL = {
    # BASELINE configs[0] exactly (C1): L2 Q4 H256 nh4 I1024 F8 E256 B16
    "item_c1": dict(kind="item", seed=11, B=16, cfg=dict(H=256, L=2, nh=4, I=1024, Q=4, F=8, E=256)),
    # shrunken C2: cross layers {0,2}, plain layer 1, Q=32, F=14, E != H
    "item_c2s": dict(kind="item", seed=12, B=4, cfg=dict(H=128, L=3, nh=2, I=512, Q=32, F=14, E=256)),
    # joint-shaped item Q-Former: Q=2 (C4's Q_item)
    "item_q2": dict(kind="item", seed=13, B=6, cfg=dict(H=128, L=2, nh=2, I=256, Q=2, F=14, E=128)),
    # shrunken C3: cross every layer, Q=64, ragged T
    "user_t96": dict(kind="user", seed=21, B=3, T=96, cfg=dict(H=128, L=2, nh=2, I=256, Q=64, E=128, n_pred=4)),
    "user_t8": dict(kind="user", seed=22, B=2, T=8, cfg=dict(H=128, L=2, nh=2, I=256, Q=64, E=192, n_pred=2)),
    # >= 256 keys: the cross-attention backward takes the few-query dK/dV kernel (attn_bwd_dkv_fewq_kernel)
    "user_t320": dict(kind="user", seed=23, B=3, T=320, cfg=dict(H=128, L=2, nh=2, I=256, Q=64, E=128, n_pred=4)),
    # Qwen3 decoder alone, right padding (no fully-masked query rows) and left padding
    "qwen_right": dict(kind="qwen", seed=31, B=3, S=40, pad_side="right", qwen=_TINYQ),
    "qwen_left": dict(kind="qwen", seed=32, B=3, S=40, pad_side="left", qwen=_TINYQ),
    # joint: item Q-Former (H == D) -> injection -> Qwen3 -> mean-pool -> InfoNCE / MRR
    "joint_right": dict(kind="joint", seed=41, B=3, S=48, hist=3, N=7, D=256, first_special_id=90, pad_side="right",
                        cfg=dict(H=256, L=2, nh=4, I=512, Q=2, F=5, E=192), qwen=_TINYQ, drop_one_special=True),
    "joint_left": dict(kind="joint", seed=42, B=3, S=48, hist=3, N=7, D=256, first_special_id=90, pad_side="left",
                       cfg=dict(H=256, L=2, nh=4, I=512, Q=2, F=5, E=192), qwen=_TINYQ),
}


# ------------------------------------------------------------------ mid-size cases (round 2) --
# One reference-generated fixture per family at sizes where the 256x256 8-phase GEMM, the multi-tile causal
# head_dim-128 attention and the few-query dK/dV kernel actually run (tests/golden/make_golden_r2.py).
_MIDQ = dict(D=1024, L=4, nq=16, nkv=8, hd=128, I=3072, vocab=64)
MID = {
    # Qwen3-0.6B-shaped decoder slice: 4 layers, D=1024, 16 q / 8 kv heads of 128, S=512, left padding, sdpa semantics
    "qwen_mid": dict(kind="qwen_mid", seed=51, B=4, S=512, pad_side="left", qwen=_MIDQ),
    # the reference's DEFAULT UserQFormer (L4 Q64 H1024 I4096, 32 predicted tokens) over T=1600 keys (C3's shape), B=2
    "user_mid": dict(kind="user_mid", seed=52, B=2, T=1600, cfg=dict(H=1024, L=4, nh=16, I=4096, Q=64, E=1024, n_pred=32)),
    # the full depth of Qwen3-Embedding-0.6B (28 layers of D 1024, 16 / 8 heads of 128, I 3072) on a short prompt: what 28 pre-norm
    # layers of bf16 arithmetic accumulate against the installed fp32 Qwen3Model
    "qwen_deep": dict(kind="qwen_mid", seed=55, B=2, S=128, pad_side="left", qwen=dict(D=1024, L=28, nq=16, nkv=8, hd=128, I=3072, vocab=64)),
    # BASELINE configs[1]'s architecture exactly (C2: L12 Q32 H768 nh12 I3072 F14 E1024) at C1's batch: the 12-layer post-LN
    # chain, head_dim 64 x 12 heads, cross-attention over 14 fields, the Q = 32 field-projection paths
    "item_mid": dict(kind="item_mid", seed=53, B=16, cfg=dict(H=768, L=12, nh=12, I=3072, Q=32, F=14, E=1024)),
    # the joint path at the 0.6B decoder's layer shape: item Q-Former (H = D = 1024, Q 2, F 14) -> injection -> 2 Qwen3 layers
    # (16 / 8 heads of 128, I 3072), S 512, left padding (the real tokenizer's side), hist 10, 15 negatives
    "joint_mid": dict(kind="joint_mid", seed=54, B=3, S=512, hist=10, N=15, D=1024, first_special_id=200, pad_side="left",
                      cfg=dict(H=1024, L=2, nh=16, I=2048, Q=2, F=14, E=1024),
                      qwen=dict(D=1024, L=2, nq=16, nkv=8, hd=128, I=3072, vocab=220)),
}
MID_STRIDE = 16


# Training-mode cases (round 5): the reference's own classes in train() mode with dropout ON, every nn.Dropout applying the keep mask
# the HIP kernels draw for that site (oracle/dropout_ref.py restates the counter-based generator; tests/golden/make_golden_r5.py).
# p = 0.2 is the item Q-Former's default (models/qformer_utils.py:19).  item_c1_train: C1's architecture (the <= 4-query x <= 16-key
# attention kernels); user_t96_train: 64 queries x 96 keys with ragged masks (the MFMA attention kernels); drop_seed / step: the
# product's BertModel.seed and the step the forward runs as.
TRAIN = {
    "item_c1_train": dict(ALL["item_c1"], kind="item_train", p=0.2, drop_seed=0x5EED, step=1),
    "user_t96_train": dict(ALL["user_t96"], kind="user_train", p=0.2, drop_seed=0xC0FFEE, step=3),
}


def train_masks(case):
    """Every keep mask of the case's forward (numpy uint8), keyed as oracle/qformer_train_ref.py expects."""
    from oracle import dropout_ref as DR
    c = case["cfg"]
    T = c["F"] if case["kind"] == "item_train" else case["T"]
    cross_freq = 2 if case["kind"] == "item_train" else 1
    return DR.qformer_masks(case["drop_seed"], case["step"], case["p"], case["B"], c["Q"], T, c["H"], c["nh"], c["L"], cross_freq)


# LoRA pinned through the INSTALLED transformers Qwen3Model with MERGED weights W' = W + (alpha / r) B A (peft's
# merge_and_unload identity; peft itself is not installed -- call site train_item_individual_token_joint.py:121-131):
# 2 layers of the 0.6B shape, B 2 x S 256, left padding.  lora_B ~ N(0, 0.05) so the adapter term is ~40 % of |W|.
LORA = dict(kind="qwen_lora", seed=57, B=2, S=256, pad_side="left", lora_r=16, lora_alpha=32.0, lora_b_std=0.05,
            qwen=dict(D=1024, L=2, nq=16, nkv=8, hd=128, I=3072, vocab=64))
# ... and the same at 4 x 512 = 2048 tokens: the q|k|v (N 4096) and gate|up (N 6144) launches then have >= 128 output tiles of 256 x 256,
# i.e. they run on gemm_pers_kernel<3|4, 1> (fused q/k-norm + RoPE / SwiGLU epilogues with the LoRA K tile), which the 512-token case does not reach
LORA_BIG = dict(kind="qwen_lora", seed=58, B=4, S=512, pad_side="left", lora_r=16, lora_alpha=32.0, lora_b_std=0.05,
                qwen=dict(D=1024, L=2, nq=16, nkv=8, hd=128, I=3072, vocab=64))
LORA_CASES = {"qwen_lora": LORA, "qwen_lora_big": LORA_BIG}
LORA_PROJ = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")
# adapter gradients stored in full (the others by their norms): one of every projection kind, both layers touched
LORA_FULL = ("layers.0.self_attn.q_proj", "layers.0.self_attn.v_proj", "layers.1.self_attn.o_proj", "layers.1.mlp.gate_proj", "layers.0.mlp.down_proj")


# BASELINE configs[3] (C4, the headline) at its EXACT architecture and lengths, two sequences (round 6): the reference's 12-layer H 1024
# Q 2 F 14 item Q-Former on 50 history items per sequence -> 100 injected tokens -> the installed Qwen3Model at the 0.6B shape (28 layers,
# D 1024, 16 / 8 heads of 128, I 3072) with merged LoRA r 16 at S 2048, left padding -> all-S mean pool -> InfoNCE over a pool of 1000 (the
# positive + 999 negatives) -> MRR rank; gradients back into the Q-Former and into the adapters (tests/golden/make_golden_r6.py).  Only the
# embedding TABLE is smaller than the model card's (4096 + 100 rows instead of 151 669 + 100: a row lookup, no arithmetic).
C4 = {
    "joint_c4": dict(kind="joint_c4", seed=66, B=2, S=2048, hist=50, N=999, D=1024, first_special_id=4096, pad_side="left",
                     lora_r=16, lora_alpha=32.0, lora_b_std=0.05,
                     cfg=dict(H=1024, L=12, nh=16, I=4096, Q=2, F=14, E=1024),
                     qwen=dict(D=1024, L=28, nq=16, nkv=8, hd=128, I=3072, vocab=4096 + 100)),
}
# gradients kept (a [1024, 1024] weight is 4 MB: every C4_ROW_STRIDE-th row + the Frobenius norm of the whole tensor)
C4_ROW_STRIDE = 32
C4_QF_KEYS = ("qformer.encoder.layer.10.crossattention.self.key.weight", "qformer.encoder.layer.0.crossattention.self.query.weight",
              "qformer.encoder.layer.11.intermediate_query.dense.weight", "qformer.encoder.layer.5.attention.self.value.weight")
C4_LORA_FULL = ("layers.0.self_attn.q_proj", "layers.27.mlp.down_proj")


def c4_rows(g):
    g = np.asarray(g)
    return np.ascontiguousarray(g[::C4_ROW_STRIDE]) if g.ndim == 2 and g.shape[0] >= 256 else g


def lora_weight_rules(case):
    """oracle.weights rule: lora_B with the case's std (the default 0.02 makes the adapter term a few % of the base)."""
    def rule(k, shp):
        if k.endswith("lora_B.weight"):
            return W.normal(k, shp, case["seed"] + 1, std=case["lora_b_std"])
        return None
    return rule


def mid_sample(x):
    """Fixture-size rule of the mid-size cases: [B,S,D] activations keep every MID_STRIDE-th position."""
    return np.ascontiguousarray(np.asarray(x)[:, ::MID_STRIDE])


def item_mid_sample(res):
    """Fixture-size rule of item_mid: query_outputs keep every 4th query token, reconstructed_fields every 4th element."""
    out = dict(res)
    out["query_outputs"] = np.ascontiguousarray(np.asarray(res["query_outputs"])[:, ::4])
    out["reconstructed_fields"] = np.ascontiguousarray(np.asarray(res["reconstructed_fields"])[..., ::4])
    return out

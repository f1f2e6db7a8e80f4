"""Counter-based dropout keep masks, restated in numpy (TEST INFRASTRUCTURE ONLY).

The HIP kernels never store a dropout mask: every site regenerates ``keep = hash(seed, element counter) >= p * 2^32`` in the
forward and in the backward (unirec_amd/csrc/common.hip.h: ur_hash2 / ur_dropout_scale / ur_drop_threshold).  This module restates
that generator (and the pair-word form the attention kernels use: attn_keep_rows) and the counter layouts so that

  * the golden generator (tests/golden/make_golden_r5.py) can make the REFERENCE's nn.Dropout modules
    (/root/reference/models/qformer.py:66,107 embeddings; :135,258 attention probabilities; :283,287 BertSelfOutput; :369,373
    BertOutput) apply exactly the masks the kernels draw, and
  * the parity tests can feed the same masks to the training-mode oracle (oracle/qformer_train_ref.py).

`ur_dropout_keep` / `ur_attn_dropout_keep` (include/unirec_hip.h) export the device-side flags; a GPU test checks them against this
file bit for bit.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & _M32


def hash2(seed, idx):
    """32-bit decision word of element `idx` (uint64 array) of stream `seed` -- common.hip.h: ur_hash2."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    s0, s1 = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    rot = _u32((s1 << np.uint64(13)) | (s1 >> np.uint64(19)))
    k2 = _u32(s0 * np.uint64(0x7FEB352D)) ^ rot ^ np.uint64(0x5851F42D)
    idx = np.asarray(idx, dtype=np.uint64)
    lo, hi = idx & _M32, idx >> np.uint64(32)
    h = _u32(_u32((lo ^ s0) * np.uint64(0x9E3779B1)) + _u32(hi + s1))
    h ^= h >> np.uint64(16)
    h = _u32(h * np.uint64(0x85EBCA6B))
    h ^= k2
    h ^= h >> np.uint64(13)
    h = _u32(h * np.uint64(0xC2B2AE35))
    h ^= h >> np.uint64(16)
    return h


def threshold(p):
    """p * 2^32 as the kernels compute it (the float p widened to double, truncated, clamped) -- common.hip.h: ur_drop_threshold."""
    t = float(np.float32(p)) * 4294967296.0
    return np.uint64(int(min(max(t, 0.0), 4294967295.0)))


def keep_range(seed, p, idx0, n):
    """uint8 [n]: keep flags of counters idx0 .. idx0 + n - 1."""
    idx = np.uint64(int(idx0)) + np.arange(int(n), dtype=np.uint64)
    return (hash2(seed, idx) >= threshold(p)).astype(np.uint8)


def hidden_keep(seed, p, rows, H, row0=0):
    """[rows, H] keep mask of a hidden-dropout site: counter = (row0 + row) * H + column (norm.hip: make_drop)."""
    return keep_range(seed, p, int(row0) * int(H), int(rows) * int(H)).reshape(int(rows), int(H))


def threshold16(p):
    """p * 65536 rounded, at least 1 for p > 0 -- common.hip.h: ur_drop_threshold16 as attn.hip uses it."""
    if p <= 0:
        return np.uint64(0)
    t = float(np.float32(p)) * 65536.0 + 0.5
    return np.uint64(max(1, int(min(max(t, 0.0), 65535.0))))


def attn_row_keys(seed, rows):
    """the two 32-bit keys of dropout rows `rows` (uint64 array) -- common.hip.h: ur_attn_row_key"""
    rows = np.asarray(rows, dtype=np.uint64)
    return hash2(seed, rows * np.uint64(2)), hash2(seed, rows * np.uint64(2) + np.uint64(1))


def attn_pair_word(k1, k2, kp):
    """decision word of key pair kp of a row with keys (k1, k2) -- common.hip.h: ur_attn_pair_word (a two-multiply finaliser over
    kp ^ k1 with k2 added between the rounds)"""
    x = _u32(np.asarray(kp, dtype=np.uint64) ^ k1)
    x ^= x >> np.uint64(16)
    x = _u32(x * np.uint64(0x7FEB352D))
    x = _u32(x + k2)
    x ^= x >> np.uint64(15)
    x = _u32(x * np.uint64(0x846CA68B))
    x ^= x >> np.uint64(16)
    return x


def attn_keep_rows(seed, p, row0, nrows, Sk):
    """uint8 [nrows, Sk]: keep flags of dropout rows row0 .. row0 + nrows - 1 (one 32-bit word per PAIR of keys: its low / high
    16 bits decide keys 2 kp / 2 kp + 1 against p * 65536) -- what ur_attn_dropout_keep exports."""
    rows = np.uint64(int(row0)) + np.arange(int(nrows), dtype=np.uint64)
    k1, k2 = attn_row_keys(seed, rows)
    kp = np.arange((int(Sk) + 1) // 2, dtype=np.uint64)
    w = attn_pair_word(k1[:, None], k2[:, None], kp[None, :])
    fields = np.stack([w & np.uint64(0xFFFF), w >> np.uint64(16)], -1).reshape(int(nrows), -1)[:, :int(Sk)]
    return (fields >= threshold16(p)).astype(np.uint8)


def attn_keep(seed, p, B, nh, Sq, Sk, b0=0):
    """[B, nh, Sq, Sk] keep mask of an attention-probability site: dropout row of (b, h, q) = ((b0 + b) * nh + h) * Sq + q
    (attn.hip: AttnP.drow0)."""
    return attn_keep_rows(seed, p, int(b0) * int(nh) * int(Sq), int(B) * int(nh) * int(Sq), Sk).reshape(int(B), int(nh), int(Sq), int(Sk))


# dropout sites of one Q-Former layer, in the product's numbering (unirec_amd/qformer.py:_layer_forward); the embeddings site is
# (layer 1023, site 0)
SITE_SELF_PROBS, SITE_SELF_OUT, SITE_CROSS_PROBS, SITE_CROSS_OUT, SITE_FFN_OUT = 1, 2, 3, 4, 5
EMB_LAYER, EMB_SITE = 1023, 0


def site_seed(base_seed, step, layer, site):
    """Seed of one dropout site of one step (unirec_amd/qformer.py: BertModel._seed) -- restated here so that fixtures do not
    depend on the product code; a GPU test asserts the two agree."""
    return (int(base_seed) * 1000003 + int(step) * 8191 + int(layer) * 64 + int(site)) & 0x7FFFFFFFFFFFFFFF


def qformer_masks(base_seed, step, p, B, Q, T, H, nh, num_layers, cross_freq, b0=0):
    """Every keep mask of one training-mode Q-Former forward, keyed as oracle/qformer_train_ref.py expects them."""
    m = {"emb": hidden_keep(site_seed(base_seed, step, EMB_LAYER, EMB_SITE), p, B * Q, H, b0 * Q).reshape(B, Q, H)}
    for i in range(num_layers):
        s = lambda site: site_seed(base_seed, step, i, site)
        m[f"{i}.self.probs"] = attn_keep(s(SITE_SELF_PROBS), p, B, nh, Q, Q, b0)
        m[f"{i}.self.out"] = hidden_keep(s(SITE_SELF_OUT), p, B * Q, H, b0 * Q).reshape(B, Q, H)
        if i % cross_freq == 0:
            m[f"{i}.cross.probs"] = attn_keep(s(SITE_CROSS_PROBS), p, B, nh, Q, T, b0)
            m[f"{i}.cross.out"] = hidden_keep(s(SITE_CROSS_OUT), p, B * Q, H, b0 * Q).reshape(B, Q, H)
        m[f"{i}.ffn.out"] = hidden_keep(s(SITE_FFN_OUT), p, B * Q, H, b0 * Q).reshape(B, Q, H)
    return m

"""numpy mirror of the C++ prologue / epilogue around the generated forward loop (csrc/attn_c128.hip: attn_fwd_c128_kernel).

Used by the CPU tests to run ONE workgroup of the generated program in the emulator (isa.run_workgroup) against a float64
attention reference.  Everything here has a line-for-line counterpart in the kernel's C++ part; a mismatch between the two is
what the GPU parity tests (tests/test_gpu_attention.py) would then catch.
"""
import numpy as np
import isa
import attn_fwd as G

LOG2E = 1.4426950408889634


def swz(row):
    return ((row & 3) << 2) | ((row >> 2) & 3)


def off128(row, ch):
    """byte offset of 16-byte chunk ch of row `row` in a 256-byte-row tile (Cfg<128>::off)"""
    return 256 * row + 16 * (ch ^ swz(row))


def f32_to_bf16(x):
    return isa.bf16_round(np.asarray(x, np.float32)).astype(np.uint16)


def bf16_to_f32(b):
    return isa.bf16_to_f32(np.asarray(b).astype(np.uint32))


def lane_consts():
    lane = np.arange(64)
    h, l31 = lane >> 5, lane & 31
    KA = np.zeros((8, 64), np.uint32)
    for ks in range(8):
        KA[ks] = [off128(int(l31[l]), 2 * ks + int(h[l])) for l in range(64)]
    g16, i = (lane >> 4) & 1, lane & 15
    row, sub8 = 4 * h + (i >> 2), 8 * (i & 1)
    TA, TB = np.zeros((4, 64), np.uint32), np.zeros((4, 64), np.uint32)
    for dt in range(4):
        ch = 4 * dt + 2 * g16 + ((i & 3) >> 1)
        TA[dt] = [G.VBASE_LDS + off128(int(row[l]), int(ch[l])) + int(sub8[l]) for l in range(64)]
        TB[dt] = [G.VBASE_LDS + off128(int(row[l]) + 8, int(ch[l])) + int(sub8[l]) for l in range(64)]
    return KA, TA, TB


def run_block(q, k, v, kmask, x, hq, rep, scale, prog=None, check=True):
    """q [S, nq*128], k / v [S, nkv*128] bf16 bit patterns (uint16), kmask [S] bool or None.  Runs query block x (256 rows) of query
    head hq through the emulator; returns (O [256, 128] f32 normalised, m [256], l [256], per-wave instruction counts)."""
    S = q.shape[0]
    assert S % 64 == 0
    ldq, ldk, ldv = q.shape[1], k.shape[1], v.shape[1]
    kvh = hq // rep
    if prog is None:
        prog = G.build_program()[0]
    ntiles = S // 64
    valid = np.ones(S, bool) if kmask is None else np.asarray(kmask, bool)
    tend = min(ntiles, 4 * x + 4)
    words = [valid[64 * t:64 * t + 64] for t in range(tend)]
    tfirst = 0
    while tfirst < tend and not words[tfirst].any():
        tfirst += 1
    maskbits = 0
    for t in range(tend):
        if not words[t].all():
            maskbits |= 1 << t
    # global memory image: K then V
    kb, vb = k.astype(np.uint16).tobytes(), v.astype(np.uint16).tobytes()
    gmem = np.frombuffer(kb + vb, np.uint8).copy()
    kbase, vbase = kvh * 128 * 2, len(kb) + kvh * 128 * 2
    wg = isa.Workgroup(4, G.LDS_BYTES, gmem)
    bias = np.where(valid[:64 * tend], 0.0, -np.inf).astype(np.float32)
    wg.lds[G.BIAS_LDS:G.BIAS_LDS + 4 * 64 * tend] = bias.view(np.uint8)
    KA, TA, TB = lane_consts()
    lane = np.arange(64)
    h, l31 = lane >> 5, lane & 31
    c = np.float32(scale * LOG2E)
    qf32 = bf16_to_f32(q)

    def init(w):
        wave = w.wid
        q0 = 256 * x + 64 * wave
        for qb in range(2):
            rows = q0 + 32 * qb + l31
            for ks in range(8):
                for j in range(4):
                    cols = hq * 128 + 16 * ks + 8 * h + 2 * j
                    ok = rows < S
                    lo = np.where(ok, qf32[np.minimum(rows, S - 1), cols], 0.0).astype(np.float32) * c
                    hi = np.where(ok, qf32[np.minimum(rows, S - 1), cols + 1], 0.0).astype(np.float32) * c
                    w.R[G.Q_(qb, ks) + j] = f32_to_bf16(lo).astype(np.uint32) | (f32_to_bf16(hi).astype(np.uint32) << 16)
        for ks in range(8):
            w.R[G.KA(ks)] = KA[ks]
        for dt in range(4):
            w.R[G.TA(dt)] = TA[dt]
            w.R[G.TB(dt)] = TB[dt]
        row, pos = 4 * wave + (lane >> 4), lane & 15
        sw = np.array([swz(int(r)) for r in row])
        w.R[G.VOFFK0] = ((row * ldk + (pos ^ sw) * 8) * 2).astype(np.uint32)
        w.R[G.VOFFV0] = ((row * ldv + (pos ^ sw) * 8) * 2).astype(np.uint32)
        w.R[G.BIASADDR] = (G.BIAS_LDS + 16 * h).astype(np.uint32)
        w.R[G.DIAGX] = (l31 - 4 * h).astype(np.int32).view(np.uint32)
        tlast = min(4 * x + wave, ntiles - 1) if q0 < S else -1
        w.sset64(G.KBASE, kbase)
        w.sset64(G.VBASE, vbase)
        w.sset(G.K16B, 16 * ldk * 2)
        w.sset(G.V16B, 16 * ldv * 2)
        w.sset(G.TEND, tend)
        w.sset(G.TFIRST, tfirst)
        w.sset(G.TLAST, tlast & 0xFFFFFFFF)
        w.sset64(G.MASKBITS, maskbits)
        w.sset(G.WAVEB, wave * 1024)

    counts = isa.run_workgroup(prog, wg, init, check=check)
    O = np.zeros((256, 128), np.float32)
    mm, ll = np.zeros(256, np.float32), np.zeros(256, np.float32)
    for w in wg.waves:
        M = w.R[G.M_(0):G.M_(0) + 2].view(np.float32)
        for qb in range(2):
            lt = w.R[G.LA(qb, 0)].view(np.float32)          # every register of the row-sum tile holds the complete sum
            inv = np.where(lt > 0, 1.0 / np.where(lt > 0, lt, 1.0), 0.0).astype(np.float32)
            for dt in range(4):
                for r in range(16):
                    val = w.R[G.O_(qb, dt, r)].view(np.float32) * inv
                    d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * h
                    O[64 * w.wid + 32 * qb + l31, d] = val
            mm[64 * w.wid + 32 * qb + l31[:32]] = M[qb][:32]
            ll[64 * w.wid + 32 * qb + l31[:32]] = lt[:32]
    return O, mm, ll, counts


def reference(q, k, v, kmask, x, hq, rep, scale):
    """float64 causal attention of query rows 256x .. 256x+255 of head hq (SDPA semantics: a row without an allowed key gives 0)"""
    S = q.shape[0]
    kvh = hq // rep
    Q = bf16_to_f32(q)[:, hq * 128:(hq + 1) * 128].astype(np.float64)
    K = bf16_to_f32(k)[:, kvh * 128:(kvh + 1) * 128].astype(np.float64)
    V = bf16_to_f32(v)[:, kvh * 128:(kvh + 1) * 128].astype(np.float64)
    valid = np.ones(S, bool) if kmask is None else np.asarray(kmask, bool)
    out = np.zeros((256, 128))
    for i in range(256):
        qi = 256 * x + i
        if qi >= S:
            continue
        ok = valid & (np.arange(S) <= qi)
        if not ok.any():
            continue
        s = (K[ok] @ Q[qi]) * scale
        p = np.exp(s - s.max())
        out[i] = (p / p.sum()) @ V[ok]
    return out

"""numpy mirror of the C++ part of attn_bwd_dq_c128_kernel: runs ONE workgroup of the generated dQ loop (attn_dq.py) in the emulator"""
import numpy as np
import isa
import attn_dq as G
from fwd_host import swz, off128, f32_to_bf16, bf16_to_f32, LOG2E


def run_block(q, k, v, do, kmask, x, hq, rep, scale, prog=None, check=True):
    """q, do [S, nq*128], k / v [S, nkv*128] bf16 bits.  Returns dQ [256, 128] f32 (already times scale) of query block x of head hq."""
    S = q.shape[0]
    ldk, ldv = k.shape[1], v.shape[1]
    kvh = hq // rep
    if prog is None:
        prog = G.build_program()[0]
    ntiles = S // 64
    valid = np.ones(S, bool) if kmask is None else np.asarray(kmask, bool)
    tend = min(ntiles, 4 * x + 4)
    tfirst = 0
    while tfirst < tend and not valid[64 * tfirst:64 * tfirst + 64].any():
        tfirst += 1
    maskbits = 0
    words = np.zeros(64, np.uint64)
    for t in range(ntiles):
        w = 0
        for i in range(64):
            if valid[64 * t + i]:
                w |= 1 << i
        words[t] = w
        if t < tend and w != (1 << 64) - 1:
            maskbits |= 1 << t
    kb, vb = k.astype(np.uint16).tobytes(), v.astype(np.uint16).tobytes()
    gmem = np.frombuffer(kb + vb, np.uint8).copy()
    kbase, vbase = kvh * 128 * 2, len(kb) + kvh * 128 * 2
    wg = isa.Workgroup(4, G.LDS_BYTES, gmem)
    wg.lds[G.WORDS_LDS:G.WORDS_LDS + 512] = words.view(np.uint8)
    lane = np.arange(64)
    h, l31 = lane >> 5, lane & 31
    c = np.float32(scale * LOG2E)
    Qf, Kf, Vf, dOf = [bf16_to_f32(t) for t in (q, k, v, do)]
    # reference forward statistics of the rows (what the forward kernel leaves in `stats`) and delta
    Qh = Qf[:, hq * 128:(hq + 1) * 128].astype(np.float64)
    Kh, Vh = Kf[:, kvh * 128:(kvh + 1) * 128].astype(np.float64), Vf[:, kvh * 128:(kvh + 1) * 128].astype(np.float64)
    dOh = dOf[:, hq * 128:(hq + 1) * 128].astype(np.float64)
    lse2 = np.full(S, np.inf)
    delta = np.zeros(S)
    for qi in range(256 * x, min(S, 256 * x + 256)):
        ok = valid & (np.arange(S) <= qi)
        if not ok.any():
            continue
        s2 = (Kh[ok] @ Qh[qi]) * scale * LOG2E
        mx = s2.max()
        p = np.exp2(s2 - mx)
        lse2[qi] = mx + np.log2(p.sum())
        o = (p / p.sum()) @ Vh[ok]
        delta[qi] = float(f32_to_bf16_round(o) @ dOh[qi])

    g16, i16 = (lane >> 4) & 1, lane & 15
    row_t, sub8 = 4 * h + (i16 >> 2), 8 * (i16 & 1)

    def init(w):
        wave = w.wid
        q0 = 256 * x + 64 * wave
        for qb in range(2):
            rows = q0 + 32 * qb + l31
            ok = rows < S
            rc = np.minimum(rows, S - 1)
            for ks in range(8):
                for j in range(4):
                    cols = hq * 128 + 16 * ks + 8 * h + 2 * j
                    lo = np.where(ok, Qf[rc, cols], 0.0).astype(np.float32) * c
                    hi = np.where(ok, Qf[rc, cols + 1], 0.0).astype(np.float32) * c
                    w.R[G.Q_(qb, ks) + j] = f32_to_bf16(lo).astype(np.uint32) | (f32_to_bf16(hi).astype(np.uint32) << 16)
                    dl = np.where(ok, do[rc, cols], 0).astype(np.uint32)
                    dh = np.where(ok, do[rc, cols + 1], 0).astype(np.uint32)
                    w.R[G.DO_(qb, ks) + j] = dl | (dh << 16)
            w.R[G.LSE2(qb)] = np.where(ok, lse2[rc], np.inf).astype(np.float32).view(np.uint32)
            w.R[G.DELTA(qb)] = np.where(ok, delta[rc], 0.0).astype(np.float32).view(np.uint32)
        w.R[G.KA0] = np.array([off128(int(l31[l]), int(h[l])) for l in range(64)], np.uint32)
        w.R[G.VA0] = w.R[G.KA0] + np.uint32(G.VBASE_LDS)
        for dt in range(4):
            ch = 4 * dt + 2 * g16 + ((i16 & 3) >> 1)
            w.R[G.TA(dt)] = np.array([off128(int(row_t[l]), int(ch[l])) + int(sub8[l]) for l in range(64)], np.uint32)
            w.R[G.TB(dt)] = np.array([off128(int(row_t[l]) + 8, int(ch[l])) + int(sub8[l]) for l in range(64)], np.uint32)
        row, pos = 4 * wave + (lane >> 4), lane & 15
        sw = np.array([swz(int(r)) for r in row])
        w.R[G.VOFFK0] = ((row * ldk + (pos ^ sw) * 8) * 2).astype(np.uint32)
        w.R[G.VOFFV0] = ((row * ldv + (pos ^ sw) * 8) * 2).astype(np.uint32)
        w.R[G.WORDADDR] = np.full(64, G.WORDS_LDS, np.uint32)
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
    dQ = np.zeros((256, 128), np.float32)
    for w in wg.waves:
        for qb in range(2):
            for dt in range(4):
                for r in range(16):
                    d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * h
                    dQ[64 * w.wid + 32 * qb + l31, d] = w.R[G.DQ(qb, dt, r)].view(np.float32) * np.float32(scale)
    return dQ, counts


def f32_to_bf16_round(x):
    return bf16_to_f32(f32_to_bf16(np.asarray(x, np.float32))).astype(np.float64)


def reference(q, k, v, do, kmask, x, hq, rep, scale):
    S = q.shape[0]
    kvh = hq // rep
    Q = bf16_to_f32(q)[:, hq * 128:(hq + 1) * 128].astype(np.float64)
    K = bf16_to_f32(k)[:, kvh * 128:(kvh + 1) * 128].astype(np.float64)
    V = bf16_to_f32(v)[:, kvh * 128:(kvh + 1) * 128].astype(np.float64)
    dO = bf16_to_f32(do)[:, hq * 128:(hq + 1) * 128].astype(np.float64)
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
        p /= p.sum()
        o = p @ V[ok]
        dp = V[ok] @ dO[qi]
        ds = p * (dp - o @ dO[qi])
        out[i] = scale * (ds @ K[ok])
    return out

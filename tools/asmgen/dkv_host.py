"""numpy mirror of the C++ part of attn_bwd_dkv_c128_kernel: runs ONE workgroup (128 keys of one kv head) of the generated dK / dV loop"""
import numpy as np
import isa
import attn_dkv as G
from fwd_host import swz, off128, f32_to_bf16, bf16_to_f32, LOG2E


def _stats(q, k, v, do, valid, hq, kvh, scale):
    S = q.shape[0]
    Q = bf16_to_f32(q)[:, hq * 128:(hq + 1) * 128].astype(np.float64)
    K = bf16_to_f32(k)[:, kvh * 128:(kvh + 1) * 128].astype(np.float64)
    V = bf16_to_f32(v)[:, kvh * 128:(kvh + 1) * 128].astype(np.float64)
    dO = bf16_to_f32(do)[:, hq * 128:(hq + 1) * 128].astype(np.float64)
    lse2, delta = np.full(S, np.inf), np.zeros(S)
    P = np.zeros((S, S))
    for qi in range(S):
        ok = valid & (np.arange(S) <= qi)
        if not ok.any():
            continue
        s2 = (K[ok] @ Q[qi]) * scale * LOG2E
        mx = s2.max()
        e = np.exp2(s2 - mx)
        lse2[qi] = mx + np.log2(e.sum())
        P[qi, ok] = e / e.sum()
        delta[qi] = (P[qi] @ V) @ dO[qi]
    return Q, K, V, dO, P, lse2, delta


def reference(q, k, v, do, kmask, xk, kvh, rep, scale):
    """dK, dV [128, 128] of key block xk of kv head kvh (sum over the rep query heads)"""
    S = q.shape[0]
    valid = np.ones(S, bool) if kmask is None else np.asarray(kmask, bool)
    dK, dV = np.zeros((128, 128)), np.zeros((128, 128))
    ws = np.zeros((2, rep, S), np.float32)
    for hr in range(rep):
        Q, K, V, dO, P, lse2, delta = _stats(q, k, v, do, valid, kvh * rep + hr, kvh, scale)
        ws[0, hr], ws[1, hr] = -delta, -lse2
        dP = dO @ V.T
        dS = P * (dP - delta[:, None])
        ks = slice(128 * xk, 128 * xk + 128)
        dV += P[:, ks].T @ dO
        dK += scale * (dS[:, ks].T @ Q)
    kv = valid[128 * xk:128 * xk + 128]
    dK[~kv] = 0
    dV[~kv] = 0
    return dK, dV, ws


def run_block(q, k, v, do, kmask, xk, kvh, rep, scale, ws, prog=None):
    S = q.shape[0]
    assert S % 128 == 0
    ldq, ldd = q.shape[1], do.shape[1]
    if prog is None:
        prog = G.build_program()[0]
    valid = np.ones(S, bool) if kmask is None else np.asarray(kmask, bool)
    qb, dob = q.astype(np.uint16).tobytes(), do.astype(np.uint16).tobytes()
    wsb = np.ascontiguousarray(ws, np.float32).tobytes()          # [2 planes][rep heads][S]
    gmem = np.frombuffer(qb + dob + wsb, np.uint8).copy()
    qbase, dobase, wsbase = kvh * rep * 128 * 2, len(qb) + kvh * rep * 128 * 2, len(qb) + len(dob)
    nrows = rep * S
    wg = isa.Workgroup(4, G.LDS_BYTES, gmem)
    lane = np.arange(64)
    h, l31 = lane >> 5, lane & 31
    g16, i16 = (lane >> 4) & 1, lane & 15
    row_t, sub8 = 4 * h + (i16 >> 2), 8 * (i16 & 1)
    c = np.float32(scale * LOG2E)
    Kf, Vf = bf16_to_f32(k), v
    qstart = 128 * xk
    ntq = (S - qstart) // 64

    def init(w):
        wave = w.wid
        kb = 128 * xk + 32 * wave
        keys = kb + l31
        for ks in range(8):
            for j in range(4):
                cols = kvh * 128 + 16 * ks + 8 * h + 2 * j
                lo, hi = Kf[keys, cols].astype(np.float32) * c, Kf[keys, cols + 1].astype(np.float32) * c
                w.R[G.KF(ks) + j] = f32_to_bf16(lo).astype(np.uint32) | (f32_to_bf16(hi).astype(np.uint32) << 16)
                w.R[G.VF(ks) + j] = Vf[keys, cols].astype(np.uint32) | (Vf[keys, cols + 1].astype(np.uint32) << 16)
        for hi_ in range(2):
            w.R[G.RA(hi_)] = np.array([off128(int(l31[l]), int(h[l])) + hi_ * G.HIGH for l in range(64)], np.uint32)
            w.R[G.CA(hi_)] = (16 * h + hi_ * G.HIGH).astype(np.uint32)
            for dt in range(4):
                ch = 4 * dt + 2 * g16 + ((i16 & 3) >> 1)
                w.R[G.TA(hi_, dt)] = np.array([off128(int(row_t[l]), int(ch[l])) + int(sub8[l]) + hi_ * G.HIGH for l in range(64)], np.uint32)
                w.R[G.TB(hi_, dt)] = np.array([off128(int(row_t[l]) + 8, int(ch[l])) + int(sub8[l]) + hi_ * G.HIGH for l in range(64)], np.uint32)
        w.R[G.XDIAG] = (l31 - 4 * h).astype(np.int32).view(np.uint32)
        row, pos = 4 * wave + (lane >> 4), lane & 15
        sw = np.array([swz(int(r)) for r in row])
        w.R[G.VOFFQ0] = ((row * ldq + (pos ^ sw) * 8) * 2).astype(np.uint32)
        w.R[G.VOFFD0] = ((row * ldd + (pos ^ sw) * 8) * 2).astype(np.uint32)
        w.R[G.VOFFC] = (4 * lane).astype(np.uint32)
        w.sset64(G.QB, qbase)
        w.sset64(G.DOB, dobase)
        w.sset64(G.WSB, wsbase)
        w.sset(G.Q16B, 16 * ldq * 2)
        w.sset(G.D16B, 16 * ldd * 2)
        w.sset(G.NTOT, ntq * rep)
        w.sset(G.KB, kb)
        w.sset(G.QSTART, qstart)
        w.sset(G.SQ4, 4 * S)
        w.sset(G.SQ4_ROWS, S)
        w.sset(G.WSEL, 0 if (wave & 1) else 4 * nrows)
        w.sset(G.CWAVE, (wave & 1) * 256)
        w.sset(G.WAVEB, wave * 1024)

    counts = isa.run_workgroup(prog, wg, init)
    dK, dV = np.zeros((128, 128), np.float32), np.zeros((128, 128), np.float32)
    for w in wg.waves:
        for dt in range(4):
            for r in range(16):
                d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * h
                dK[32 * w.wid + l31, d] = w.R[G.DK(dt, r)].view(np.float32) * np.float32(scale)
                dV[32 * w.wid + l31, d] = w.R[G.DV(dt, r)].view(np.float32)
    kv = valid[128 * xk:128 * xk + 128]
    dK[~kv] = 0
    dV[~kv] = 0
    return dK, dV, counts

"""Generator of the hand-scheduled causal attention backward dQ main loop (head_dim 128, bf16, gfx950).

Same decomposition as the forward (tools/asmgen/attn_fwd.py): a workgroup = 4 waves = 256 query rows of one (batch, query head),
a wave owns 64 rows (two 32-row blocks qb) and the whole register file of its SIMD; the key tiles (64 keys) stream through LDS.
Per tile and wave 96 MFMAs (v_mfma_f32_32x32x16_bf16):
    S^T  = K Q~^T          (Q~ = q * scale * log2e, pre-scaled in the C++ prologue)        32
    dP^T = V dO^T                                                                            32
    dQ^T += K^T dS^T       (dS = p * (dP - delta), p = exp2(S - LSE2): no running maximum)   32
so the loop is bound by the matrix pipe (96 x 32 cycles against ~2300 cycles of issue), unlike the forward.
  a[0:127]   dQ^T accumulators [qb][dt][16]     a[128:191] Q~ fragments [qb][k-step]     a[192:255] dO fragments [qb][k-step]
  v[16:79]   S^T  [sub][qb][16]                 v[80:143]  dP^T [sub][qb][16]            v[144:175] dS fragments (bf16) [sub][qb][k16-step]
  v[176:207] K / V row-fragment rings (4 + 4)   v[208:239] K^T fragment ring (8)
MFMA order per iteration: S, dP of sub-tile 0 (tile i) | dQ of sub-tile 0 (tile i-1) | S, dP of sub-tile 1 (tile i) | dQ of sub-tile 1
(tile i-1); the vector work of a sub-tile (subtract, exp2, subtract, multiply, convert) rides in the gaps of the two blocks that
follow its S / dP block.  K tiles stay in LDS one iteration longer than in the forward (their transposed fragments feed the dQ
products of the NEXT iteration): K is loaded two tiles ahead, V three, four ring slots each.
Key padding: a tile with a masked key (bit in MASKBITS) builds per-register lane masks from the tile's key word (LDS words table).
Replaces the recomputation half of SDPA's backward (transformers modeling_qwen3.py:185-208 under autograd) for q; dK / dV: attn_bwd_dkv2_kernel.
"""
import os

from isa import *      # noqa: F401,F403
import isa
from attn_fwd import (flatten, _tag, s_m0_add, s_andn2_b64, KSLOT, VSLOT, VBASE_LDS, ACC_ROW)

LEADK, LEADV, LEADT = 6, 6, 8
PACKED = os.environ.get("UR_ASMGEN_PACKED", "0") == "1"        # round 6 LAB switch, off: packed-f32 subtracts / multiplies in the vector stream.  Correct in the emulator and on the hardware where the generator places them, but re-placed by balance.py the dQ kernel's LAST body became NONDETERMINISTIC on the MI355X (a high-half-broadcast v_pk_add_f32 directly ahead of an MFMA; docs/lab_notes.md 14.2): not shipped
BALANCE = os.environ.get("UR_ASMGEN_BALANCE", "1") == "1"      # round 6: dependency-aware re-placement of the side instructions (balance.py)
WORDS_LDS = 131072
LDS_BYTES = WORDS_LDS + 64 * 8
ASM_VGPR_FIRST = 8


def S_(sub, qb, r=0):
    return v(16 + 16 * (2 * sub + qb) + r)


def DP(sub, qb, r=0):
    return v(80 + 16 * (2 * sub + qb) + r)


def DS(sub, qb, s2, j=0):
    return v(144 + 4 * ((sub * 2 + qb) * 2 + s2) + j)


def KRF(slot):
    return v(176 + 4 * (slot & 3))


def VRF(slot):
    return v(192 + 4 * (slot & 3))


def KTF(slot, j=0):
    return v(208 + 4 * (slot & 7) + j)


KA0, VA0 = v(240), v(241)          # LDS address of this lane's row fragment, k-step 0 (k-step ks: XOR 32 ks), K ring / V ring


def TA(dt):
    return v(242 + dt)


def TB(dt):
    return v(246 + dt)


def LSE2(qb):
    return v(250 + qb)


def DELTA(qb):
    return v(252 + qb)


TMPW, TMPA = v(254), v(255)       # TMPW:TMPA doubles as the 64-bit key word read of the mask block
NEGINF = v(15)                    # (a literal cannot share an instruction's constant bus with VCC)
TMPB = TMPW
DIAGX, WORDADDR, VOFFK0, VOFFV0 = v(8), v(9), v(12), v(13)      # register tuples bound as operands start even


def VOFFK(j):
    return (v(12), v(10), v(11), v(14))[j]


VOFFV_BASE = v(13)


def DQ(qb, dt, r=0):
    return a(16 * (4 * qb + dt) + r)


def Q_(qb, ks):
    return a(128 + 4 * (8 * qb + ks))


def DO_(qb, ks):
    return a(192 + 4 * (8 * qb + ks))


KBASE, VBASE, K16B, V16B, IT, TEND, TLAST, TFIRST, MASKBITS = s(36), s(38), s(40), s(41), s(42), s(43), s(44), s(45), s(46)
K64B, V64B, PTRK, PTRV, WAVEB = s(48), s(49), s(50), s(52), s(57)
TS, TENDM1, TLASTP1, TS2, WORD, MSK = s(66), s(67), s(34), s(35), s(58), s(60)
STAMP, PREV, DBGPTR = s(68), s(70), s(72)
NACC = 12


def ACC(i):
    return s(74 + i)


ABLATE = set(x for x in os.environ.get("UR_ASMGEN_ABLATE", "").split(",") if x)
STAMPS = os.environ.get("UR_ASMGEN_STAMPS", "") == "1"

# MFMA blocks of an iteration: A = S, dP of sub-tile 0 (32) | B = dQ of sub-tile 0 of the previous tile (16) | C = S, dP sub-tile 1 (32) | D (16)
GA, GB, GC, GD, NG = 0, 32, 48, 80, 96
G0S = (36, 84)              # first gap of the vector stream of the two sub-tiles (4.5 instructions per score pair position, 3 per gap)
DMA_GAPS_K = [2, 6, 10, 14]
DMA_GAPS_V = [50, 54, 58, 62]


def dma_setup(kadd=2, vadd=3, base=IT):
    return _tag([s_add_i32(TS, base, Lit(kadd)), s_min_i32(TS, TS, TENDM1), s_mul_i32(TS2, TS, K64B),
                 s_add_u32(PTRK, KBASE, TS2), s_addc_u32(PTRK + 1, KBASE + 1, Lit(0)),
                 s_add_i32(TS, base, Lit(vadd)), s_min_i32(TS, TS, TENDM1), s_mul_i32(TS2, TS, V64B),
                 s_add_u32(PTRV, VBASE, TS2), s_addc_u32(PTRV + 1, VBASE + 1, Lit(0))], "dma")


def dma_piece(j, kslot, vslot):
    """piece j of this wave's 8 per iteration (0-3: K rows 16 j.. of tile it+2, 4-7: V rows of tile it+3)"""
    if j < 4:
        return _tag([s_m0_add(WAVEB, kslot * KSLOT + j * 4096), global_load_lds_dwordx4(VOFFK(j), PTRK)], "dma")
    jj = j - 4
    # V pieces share one per-lane offset: the piece stride goes onto the scalar base
    out = [s_m0_add(WAVEB, VBASE_LDS + vslot * VSLOT + jj * 4096)]
    if jj > 0:
        out += [s_add_u32(PTRV, PTRV, V16B), s_addc_u32(PTRV + 1, PTRV + 1, Lit(0))]
    out.append(global_load_lds_dwordx4(VOFFV_BASE, PTRV))
    return _tag(out, "dma")


def row_read(ring, slot, base, ks, lds_off):
    """row fragment of k-step ks: the swizzled address of k-step 0 XOR 32 ks"""
    out = []
    if ks == 0:
        out.append(ds_read_b128(ring(slot), base, lds_off))
    else:
        out += [valu2("v_xor_b32", TMPA, Lit(32 * ks), base, lambda p, q: p ^ q), ds_read_b128(ring(slot), TMPA, lds_off)]
    return _tag(out, "frag")


def kt_reads(f, kslot, sub):
    s2, dt = f >> 2, f & 3
    off = kslot * KSLOT + 256 * (32 * sub + 16 * s2)
    return _tag([ds_read_b64_tr_b16(KTF(f, 0), TA(dt), off), ds_read_b64_tr_b16(KTF(f, 2), TB(dt), off)], "frag")


def stamp_start():
    return [s_memtime_wait(PREV)] if STAMPS else []


def stamp_acc(i):
    if not STAMPS:
        return []
    return [s_memtime_wait(STAMP), s_sub_u32(TS2, STAMP, PREV), s_add_u32(ACC(i), ACC(i), TS2), s_add_u32(ACC(i + 1), ACC(i + 1), Lit(1)), s_mov_b32(PREV, STAMP)]


def top():
    if "bar" in ABLATE:
        return []
    return stamp_start() + [s_waitcnt(vmcnt=4), s_barrier()] + stamp_acc(0)


class Counter:
    def __init__(self):
        self.n = 0

    def new(self):
        self.n += 1
        return self.n


def mask_block(sub, skip_qb0, diag, tag, cnt):
    """masks of sub-tile `sub` applied to S before the exps: key padding (dynamic, from the tile's key word) and the causal diagonal"""
    out = []
    k = cnt.new()
    lbl = "NOMASK_%s_%d" % (tag, k)
    blk = [s_lshl_b32(TS2, IT, Lit(3)), v_add_u32(TMPA, WORDADDR, TS2), ds_read_b64(TMPW, TMPA, 0), s_waitcnt(lgkmcnt=0),
           v_readfirstlane(WORD, TMPW), v_readfirstlane(WORD + 1, TMPA)]
    for r in range(16):
        # lanes 0-31 hold key 32 sub + ACC_ROW[r], lanes 32-63 that + 4
        blk += [s_bitcmp1_b64(WORD, Lit(32 * sub + ACC_ROW[r])), s_cselect_b32(MSK, Lit(-1), Lit(0)),
                s_bitcmp1_b64(WORD, Lit(32 * sub + ACC_ROW[r] + 4)), s_cselect_b32(MSK + 1, Lit(-1), Lit(0)), s_mov_vcc(MSK)]
        for qb in range(2):
            if not (skip_qb0 and qb == 0):
                blk.append(v_cndmask_b32(S_(sub, qb, r), NEGINF, S_(sub, qb, r), VCC))
    out.append([s_bitcmp1_b64(MASKBITS, IT)] + cond_block(s_cbranch_scc(0, lbl), blk, label(lbl)))
    if diag and "diag" not in ABLATE:
        qb = sub
        for r in range(16):
            out += [v_cmp_i32("ge", VCC, DIAGX, Lit(ACC_ROW[r])), v_cndmask_b32(S_(sub, qb, r), NEGINF, S_(sub, qb, r), VCC)]
    return out


def s_mov_vcc(x):
    def fn(w):
        w.vcc = w.s64(x)
    return I("s_mov_b64 vcc, %s" % rrange(x, 2), "salu", (x, x + 1), (VCC,), fn, 1)


def ds_read_b64(d, vaddr, off):
    def fn(w):
        ad = w.R[vaddr].astype(np.int64) + off
        w.lds_read_check(ad, 8)
        for l in range(64):
            w.R[d:d + 2, l] = w.lds[ad[l]:ad[l] + 8].view(np.uint32)
    return I("ds_read_b64 %s, %s offset:%d" % (rrange(d, 2), rname(vaddr), off), "lds", (vaddr,), (d, d + 1), fn, 4)


def v_readfirstlane(d, x):
    def fn(w):
        w.sset(d, int(w.R[x, 0]))
    return I("v_readfirstlane_b32 %s, %s" % (rname(d), rname(x)), "valu", (x,), (d,), fn, 4)


import numpy as np  # noqa: E402


def vec_events(sub, G0, skip_qb0):
    """(position, order, instruction) of a sub-tile's vector stream: t = S - LSE2, u = dP - delta, p = exp2(t), dS = p u, bf16 pairs.
    Score slot m = 2 r + qb starts in gap G0 + floor(1.4 m): 32 slots over 45 gaps, the conversion of a pair three gaps behind it."""
    ev = []

    def start(m):
        return G0 + (7 * m) // 5

    for r in range(16):
        for qb in range(2):
            if skip_qb0 and qb == 0:
                continue
            s, d, pos = S_(sub, qb, r), DP(sub, qb, r), start(2 * r + qb)
            if not PACKED:
                ev += [(pos, 0, v_sub_f32(s, s, LSE2(qb))), (pos, 1, v_sub_f32(d, d, DELTA(qb))), (pos + 1, 2, v_exp_f32(s, s)), (pos + 2, 3, v_mul_f32(d, d, s))]
            elif r % 2 == 0:
                # round 6: two scores per subtract / multiply (packed f32; the row constant is one register broadcast to both halves)
                pos1 = start(2 * (r + 1) + qb)
                ev += [(pos, 0, v_pk_sub_f32_bcast(s, s, LSE2(qb))), (pos, 1, v_pk_sub_f32_bcast(d, d, DELTA(qb))), (pos + 1, 2, v_exp_f32(s, s)),
                       (pos1 + 1, 2, v_exp_f32(s + 1, s + 1)), (pos1 + 2, 3, v_pk_mul_f32(d, d, s))]
    for qb in range(2):
        if skip_qb0 and qb == 0:
            continue
        for s2 in range(2):
            for j in range(4):
                r0 = 8 * s2 + 2 * j
                ev.append((start(2 * (r0 + 1) + qb) + 3, 4, v_cvt_pk_bf16_f32(DS(sub, qb, s2, j), DP(sub, qb, r0), DP(sub, qb, r0 + 1))))
    return ev


def dq_block(sub, skip_qb0):
    """16 MFMAs: per 16-key step and head_dim block, dQ^T[qb][dt] += K^T dS for both query blocks"""
    out = []
    for s2 in range(2):
        for dt in range(4):
            for qb in range(2):
                out.append(None if (skip_qb0 and qb == 0) else v_mfma_32x32x16_bf16(DQ(qb, dt), KTF(4 * s2 + dt), DS(sub, qb, s2), DQ(qb, dt)))
    return out


def sdp_block(sub, skip_qb0):
    """32 MFMAs: per k-step, S (K row fragment) and dP (V row fragment) for both query blocks"""
    out = []
    for ks in range(8):
        for which in range(2):
            for qb in range(2):
                if skip_qb0 and qb == 0:
                    out.append(None)
                elif which == 0:
                    out.append(v_mfma_32x32x16_bf16(S_(sub, qb), KRF(ks), Q_(qb, ks), None if ks == 0 else S_(sub, qb)))
                else:
                    out.append(v_mfma_32x32x16_bf16(DP(sub, qb), VRF(ks), DO_(qb, ks), None if ks == 0 else DP(sub, qb)))
    return out


def build_body(p, kind, last, tag, cnt, with_dma=True):
    do_S, do_Q, carry = kind != "epi", kind != "pro", kind != "pro"
    prev_last = kind == "epi"
    kslot, kslot_n, kslot_prev, vslot, vslot_n = p & 3, (p + 1) & 3, (p - 1) & 3, p & 3, (p + 1) & 3
    slots = [[] for _ in range(NG)]
    pre, mf = [], [None] * NG

    def put(g, *items):
        slots[max(0, g)].extend(items)

    if do_S:
        mf[GA:GA + 32] = sdp_block(0, False)
        mf[GC:GC + 32] = sdp_block(1, last)
    if do_Q:
        mf[GB:GB + 16] = dq_block(0, False)
        mf[GD:GD + 16] = dq_block(1, prev_last)
    # K^T fragments of the PREVIOUS tile (f = (16-key step, head_dim block) first used by MFMA 8 s2 + 2 dt of its block)
    if do_Q:
        for f in range(8):
            first = 8 * (f >> 2) + 2 * (f & 3)
            put(GB + first - LEADT, *kt_reads(f, kslot_prev, 0))
            put(GD + first - LEADT, *kt_reads(f, kslot_prev, 1))
    # row fragments: k-step ks of a sub-tile's S / dP is first used by MFMA 4 ks (S) and 4 ks + 2 (dP) of the block
    if do_S:
        for ks in range(8):
            put(GC + 4 * ks - LEADK, *row_read(KRF, ks, KA0, ks, kslot * KSLOT + 8192))
            put(GC + 4 * ks + 2 - LEADV, *row_read(VRF, ks, VA0, ks, vslot * VSLOT + 8192))
        if kind == "pro":
            for ks in range(2):
                pre += row_read(KRF, ks, KA0, ks, kslot * KSLOT) + row_read(VRF, ks, VA0, ks, vslot * VSLOT)
        for ks in range(2, 8):
            put(GA + 4 * ks - LEADK, *row_read(KRF, ks, KA0, ks, kslot * KSLOT))
            put(GA + 4 * ks + 2 - LEADV, *row_read(VRF, ks, VA0, ks, vslot * VSLOT))
        if not last:
            for ks in range(2):
                put(NG - LEADK + 3 * ks, *row_read(KRF, ks, KA0, ks, kslot_n * KSLOT))
                put(NG - LEADK + 3 * ks + 1, *row_read(VRF, ks, VA0, ks, vslot_n * VSLOT))
    if with_dma:
        for j, g in enumerate(DMA_GAPS_K + DMA_GAPS_V):
            put(g, dma_piece(j, (p + 2) & 3, (p + 3) & 3))
    ev = []
    if carry:
        ev += [(pos - NG, o, i) for pos, o, i in vec_events(1, G0S[1], prev_last) if pos >= NG]
    if do_S:
        ev += [(pos, o, i) for pos, o, i in vec_events(0, G0S[0], False) if pos < NG]
        ev += [(pos, o, i) for pos, o, i in vec_events(1, G0S[1], last) if pos < NG]
    for pos, o, i in sorted(ev, key=lambda t: (t[0], t[1])):
        put(pos, _tag([i], "soft")[0])
    if do_S:
        for sub in range(2):
            put(G0S[sub] - 2, *_tag(mask_block(sub, last and sub == 1, last, tag, cnt), "max"))
    seq = list(pre)
    for g in range(NG):
        if mf[g] is not None:
            seq.append(mf[g])
        seq.extend(flatten(slots[g]))
    return [i for i in seq if not _ablated(i)]


def _ablated(i):
    if not ABLATE:
        return False
    return ("dma" in ABLATE and i.kind == "vmem") or ("soft" in ABLATE and i.tag == "soft") or ("frag" in ABLATE and i.tag == "frag") or \
        ("mfma" in ABLATE and i.kind == "mfma")


def entry_pending():
    return [tuple(range(KRF(0), KRF(0) + 4)), tuple(range(VRF(0), VRF(0) + 4)), tuple(range(KRF(1), KRF(1) + 4)), tuple(range(VRF(1), VRF(1) + 4))]


def common_scalars():
    out = [s_add_i32(TENDM1, TEND, Lit(-1)), s_lshl_b32(K64B, K16B, Lit(2)), s_lshl_b32(V64B, V16B, Lit(2))]
    for j in range(1, 4):
        out += [s_mul_i32(TS, K16B, Lit(j)), v_add_u32(VOFFK(j), VOFFK0, TS)]
    return out


PROLOGUE_TILES = (("k", 0), ("v", 0), ("k", 1), ("v", 1), ("v", 2))      # issue order (ring slot = tile - t0; K runs two tiles ahead, V three)


def prologue_pieces(tiles):
    out = []
    for which, add in tiles:
        grp = dma_setup(kadd=add, vadd=add, base=TFIRST)[:5] if which == "k" else dma_setup(kadd=add, vadd=add, base=TFIRST)[5:]
        for j in (range(4) if which == "k" else range(4, 8)):
            out.append(grp + dma_piece(j, add, add))
            grp = []
    return out


def dma_prologue_code():
    """the FIRST tile's LDS-DMA K(t0), V(t0) as a statement of its own; the other prologue tiles ride between the initialisation
    instructions of the main statement (see attn_fwd.dma_prologue_code)"""
    out = [comment("---- prologue LDS-DMA")] + common_scalars() + [s_nop(3)]
    for piece in prologue_pieces(PROLOGUE_TILES[:2]):
        out += piece
    return out


def entry_code():
    out = [comment("---- entry of the main statement")]
    if STAMPS:
        out += [s_mov_b32(ACC(i), Lit(0)) for i in range(NACC)]
    out += common_scalars() + [s_add_i32(TLASTP1, TLAST, Lit(1)), s_mov_b32(IT, TFIRST), s_nop(3)]
    init = [v_accvgpr_write(a(i), Lit(0)) for i in range(128)] + [v_mov_b32(NEGINF, Lit(0xFF800000))]
    pieces = prologue_pieces(PROLOGUE_TILES[2:])
    step = max(1, len(init) // len(pieces))
    for n, piece in enumerate(pieces):
        out += piece + init[n * step:(n + 1) * step if n + 1 < len(pieces) else len(init)]
    return out


def build_dma_program():
    P = Program()
    P.add(fix_hazards(dma_prologue_code())[0])
    P.finalize()
    return P


def build_program(with_dma_prologue=True):
    cnt = Counter()
    P = Program()
    if with_dma_prologue:
        P.add(fix_hazards(dma_prologue_code())[0])
    P.add(fix_hazards(entry_code())[0])
    P.add(s_cmp("lt", TLAST, TFIRST), s_cbranch_scc(0, "HAVE"), s_add_i32(TLAST, TFIRST, Lit(-2)), s_add_i32(TLASTP1, TFIRST, Lit(-1)),
          s_branch("D_0"), label("HAVE"), s_cmp("eq", TLAST, TFIRST), s_cbranch_scc(1, "PROL"))
    bodies = {}

    def emit_body(name, p, kind, last):
        seq = top() + dma_setup() + build_body(p, kind, last, name, cnt)
        seq += stamp_acc({"PRO": 2, "PROL": 2}.get(name, 6 if last else 4))
        if BALANCE:
            import balance
            seq = balance.balance(flatten(seq), entry_lgkm=entry_pending() if kind == "steady" else (), temps={TMPA}, name="dq_" + name.split("_")[0])
        fixed, _ = fix_hazards(seq, entry_lgkm=entry_pending() if kind == "steady" else ())
        bodies[name] = fixed
        return fixed

    P.add(label("PRO"), emit_body("PRO", 0, "pro", False), s_add_i32(IT, IT, Lit(1)), s_branch("D_1"))
    P.add(label("PROL"), emit_body("PROL", 0, "pro", True), s_add_i32(IT, IT, Lit(1)), s_branch("D_1"))
    for p in range(4):
        nxt = "D_%d" % ((p + 1) & 3)
        P.add(label("D_%d" % p),
              s_cmp("lt", IT, TLAST), s_cbranch_scc(1, "STEADY_%d" % p),
              s_cmp("eq", IT, TLAST), s_cbranch_scc(1, "LAST_%d" % p),
              s_cmp("eq", IT, TLASTP1), s_cbranch_scc(1, "EPI_%d" % p),
              s_cmp("ge", IT, TEND), s_cbranch_scc(1, "EXIT"))
        skip = top() + dma_setup()
        for j in range(8):
            skip += dma_piece(j, (p + 2) & 3, (p + 3) & 3)
        skip += stamp_acc(10)
        P.add(comment("---- SKIP_%d" % p), fix_hazards(skip)[0], s_add_i32(IT, IT, Lit(1)), s_branch(nxt))
        P.add(label("STEADY_%d" % p), emit_body("STEADY_%d" % p, p, "steady", False), s_add_i32(IT, IT, Lit(1)), s_branch(nxt))
        P.add(label("LAST_%d" % p), emit_body("LAST_%d" % p, p, "steady", True), s_add_i32(IT, IT, Lit(1)), s_branch(nxt))
        P.add(label("EPI_%d" % p), s_cmp("ge", IT, TEND), s_cbranch_scc(1, "EPIB_%d" % p))
        pre = top() + dma_setup()
        for j in range(8):
            pre += dma_piece(j, (p + 2) & 3, (p + 3) & 3)
        P.add(fix_hazards(pre)[0], label("EPIB_%d" % p))
        seq = stamp_start() + build_body(p, "epi", False, "EPI_%d" % p, cnt, with_dma=False) + stamp_acc(8)
        P.add(fix_hazards(seq)[0], s_add_i32(IT, IT, Lit(1)), s_branch(nxt))
    if STAMPS:
        P.add(label("EXIT"))
        for i in range(NACC):
            P.add(v_mov_b32(TMPA, ACC(i)), v_mov_b32(TMPB, Lit(0)), global_store_dword_s(TMPA, TMPB, DBGPTR, 4 * i))
        P.add(s_waitcnt(vmcnt=0, lgkmcnt=0), s_nop(15))
    else:
        P.add(label("EXIT"), s_waitcnt(lgkmcnt=0), s_nop(15))      # (LDS-DMA still in flight: see attn_fwd.build_program)
    P.finalize()
    return P, bodies


if __name__ == "__main__":
    import collections
    P, bodies = build_program()
    print("instructions:", len(P.ins), P.stats())
    for name in ("STEADY_1", "LAST_1", "PRO"):
        seq = bodies[name]
        c = collections.Counter()
        skip = False
        costs, cur = [], None
        for i in seq:
            if i.region == "begin":
                c["branch"] += 1
                skip = True
                continue
            if i.region == "end":
                skip = False
                continue
            if skip or i.kind == "label":
                continue
            c[i.kind] += 1
            if i.kind == "mfma":
                if cur is not None:
                    costs.append(cur)
                cur = 8
            elif cur is not None:
                cur += (i.note + 1) * 4 if i.kind == "nop" else max(4, i.cost)
        costs.append(cur)
        print(name, dict(c), "total", sum(c.values()), "sum max32", sum(max(32, x) for x in costs), "mfma*32", 32 * c["mfma"])
        print("   ", costs)

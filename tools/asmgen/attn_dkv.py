"""Generator of the hand-scheduled causal attention backward dK / dV main loop (head_dim 128, bf16, gfx950).

A workgroup = 4 waves = 128 keys of one (batch, kv head); a wave owns 32 keys (the key is the MFMA LANE, so S and dP accumulators are
directly the B operands of the dV^T / dK^T products) and the whole register file of its SIMD.  It sweeps the query tiles (64 queries =
two 32-row halves a, b) at and below its diagonal, for every query head of the GQA group; the Q / dO tiles and the row constants stream
through a 4-slot LDS ring by LDS-DMA (three tiles ahead, one barrier per tile).  Per tile 64 MFMAs (v_mfma_f32_32x32x16_bf16):
    X_a, X_b : S'[q][key] = Q K~^T - LSE2[q],  dP'[q][key] = dO V^T - delta[q]     (row constants arrive as the chains' C operand: LDS -> accumulator)
    Y_a, Y_b : dV^T[d][key] += dO^T P,  dK^T[d][key] += Q^T dS                     (P = exp2(S'), dS = P dP')
  a[0:63] dK^T   a[64:127] dV^T   a[128:159] K~ fragments (k * scale * log2e)   a[160:191] V fragments   a[192:255] Q / dO row-fragment rings
  v[16:47] S' (a, b)   v[48:79] dP' (a, b)   v[80:111] P / dS fragments   v[112:143] transposed-fragment ring
Order X_a | X_b | Y_a | Y_b inside ONE iteration (no state crosses the barrier): the vector work of half a rides in the gaps of X_b, that of
half b in the gaps of Y_a; the next tile's row constants and first row fragments are requested in Y_b.
Masks: the causal diagonal is a dynamic block (tiles with q0 < key block + 32) that sets S' = -inf where key > query; padded KEYS are
lanes, their columns never mix with valid ones, and are zeroed at the store; padded QUERY rows carry LSE2 = +inf, i.e. P = 0.
Replaces the k / v half of SDPA's backward (transformers modeling_qwen3.py:185-208 under autograd); row constants from attn_bwd_dq_c128_kernel.
"""
import os

import numpy as np

from isa import *      # noqa: F401,F403
import isa
from attn_fwd import flatten, _tag, s_m0_add, ACC_ROW
from attn_dq import s_mov_vcc, ds_read_b64, v_readfirstlane
import balance

PACKED = os.environ.get("UR_ASMGEN_PACKED", "0") == "1"        # round 6 LAB switch, off (see attn_dq.py PACKED): packed-f32 multiplies in the vector stream
BALANCE = os.environ.get("UR_ASMGEN_BALANCE", "1") == "1"      # round 6: dependency-aware re-placement of the side instructions (balance.py)

LEAD = int(os.environ.get("UR_DKV_LEAD", "6"))      # transposed fragments: MFMA slots between a read and its use (their ring has 8 slots: < 8)
LEAD_ROW = int(os.environ.get("UR_DKV_LEAD_ROW", "6"))     # row fragments (one register set per k-step: up to 14); NPRE of the next tile's come with this one
NPRE = LEAD_ROW // 2
# first gap of a half's vector stream behind the start of the MFMA block it rides in: 3 = the X block's last MFMA has retired before the first
# exp reads its accumulator.  (Lab sweep, whole backward at B 64 / S 2048: 0 and 1 run 1.5 % faster -- and are WRONG, the emulator suite fails on NaN:
# the stream reads S' under the MFMA's latency; 2: -0.5 %, 5 / 6: nothing.  UR_DKV_SOFT0 overrides for lab builds.)
SOFT0 = int(os.environ.get("UR_DKV_SOFT0", "3"))
DMA_GAPS = [int(x) for x in os.environ.get("UR_DKV_DMA_GAPS", "1,5,9,13,49,53,57,61,63").split(",")]      # lab: the MFMA gaps that carry the tile's 9 LDS-DMA pieces
SLOT = 33792                 # row constants 1 KiB (ns[64] f32 | nd[64] f32 | pad) | Q tile 16 KiB | dO tile 16 KiB  (every immediate of slot 1 < 64 KiB)
CONST_OFF, QOFF, DOFF = 0, 1024, 17408
LDS_BYTES = 4 * SLOT
HIGH = 2 * SLOT              # address registers come in two sets: slots 0-1 (immediates < 64 KiB) and slots 2-3 (+HIGH)
ASM_VGPR_FIRST = 8


def S_(half, r=0):
    return v(16 + 16 * half + r)


def DP(half, r=0):
    return v(48 + 16 * half + r)


def PF(half, s2, j=0):
    return v(80 + 4 * (2 * half + s2) + j)


def DSF(half, s2, j=0):
    return v(96 + 4 * (2 * half + s2) + j)


def TR(slot, j=0):
    return v(112 + 4 * (slot & 7) + j)


def RA(hi):
    return v(144 + hi)                 # row-fragment address (k-step 0) of this lane, slots 0-1 / 2-3


def TA(hi, dt):
    return v(146 + 4 * hi + dt)


def TB(hi, dt):
    return v(154 + 4 * hi + dt)


def CA(hi):
    return v(162 + hi)                 # row-constant address (16 h + ...)


XDIAG, NEGINF, TMPA, XH = v(164), v(165), v(166), v(167)
VOFFQ0, VOFFD0, VOFFC = v(8), v(9), v(10)


def VOFFQ(j):
    return (v(8), v(168), v(169), v(170))[j]


def VOFFD(j):
    return (v(9), v(171), v(172), v(173))[j]


def DK(dt, r=0):
    return a(16 * dt + r)


def DV(dt, r=0):
    return a(64 + 16 * dt + r)


def KF(ks):
    return a(128 + 4 * ks)


def VF(ks):
    return a(160 + 4 * ks)


def QRF(slot):
    return a(192 + 4 * (slot & 7))


def DORF(slot):
    return a(224 + 4 * (slot & 7))


QB, DOB, WSB, Q16B, D16B, IT, NTOT, NTQ, KB, Q0S, HRS = s(36), s(38), s(40), s(42), s(43), s(44), s(45), s(46), s(47), s(48), s(49)
PTRQ, PTRD, PTRC, WAVEB, TS, TS2, QSTART, NROWS4, SQ4 = s(50), s(52), s(54), s(57), s(66), s(67), s(58), s(59), s(60)
LT, LHR, LQ0, Q64B, D64B, WSEL = s(61), s(62), s(63), s(64), s(65), s(34)
STAMP, PREV, DBGPTR = s(68), s(70), s(72)
NACC = 6


def ACC(i):
    return s(74 + i)


ABLATE = set(x for x in os.environ.get("UR_ASMGEN_ABLATE", "").split(",") if x)
STAMPS = os.environ.get("UR_ASMGEN_STAMPS", "") == "1"

GXA, GXB, GYA, GYB, NG = 0, 16, 32, 48, 64


def slot_off(slot):
    """(address set, immediate) of ring slot `slot`"""
    return (slot >> 1, (slot & 1) * SLOT)


# round 6: the swizzled row-fragment addresses of k-steps 1..7 (RA ^ 32 ks) live in registers of their own (v174..v187: the statement
# clobbers v8..v255 and used 166 of them) instead of one v_xor_b32 per read: 32 vector instructions (128 issue cycles) less per tile
PRE_ADDR = os.environ.get("UR_DKV_PRE_ADDR", "1") == "1"


def RAK(hi, ks):
    return RA(hi) if ks == 0 else v(174 + 7 * hi + ks - 1)


def row_addr_setup():
    if not PRE_ADDR:
        return []
    return [valu2("v_xor_b32", RAK(hi, ks), Lit(32 * ks), RA(hi), lambda p, q: p ^ q) for hi in range(2) for ks in range(1, 8)]


def row_read(ring, rslot, slot, half, ks, is_do):
    hi, imm = slot_off(slot)
    imm += 8192 * half + (DOFF if is_do else QOFF)
    if ks == 0 or PRE_ADDR:
        return _tag([ds_read_b128(ring(rslot), RAK(hi, ks), imm)], "frag")
    return _tag([valu2("v_xor_b32", TMPA, Lit(32 * ks), RA(hi), lambda p, q: p ^ q), ds_read_b128(ring(rslot), TMPA, imm)], "frag")


def tr_reads(tslot, slot, half, s2, dt, is_do):
    hi, imm = slot_off(slot)
    imm += 256 * (32 * half + 16 * s2) + (DOFF if is_do else QOFF)
    return _tag([ds_read_b64_tr_b16(TR(tslot, 0), TA(hi, dt), imm), ds_read_b64_tr_b16(TR(tslot, 2), TB(hi, dt), imm)], "frag")


def const_reads(slot, half):
    """row constants of half `half` of the tile in `slot` straight into the accumulators: register 4 g + e <- row 8 g + 4 h + e"""
    hi, imm = slot_off(slot)
    out = []
    for g in range(4):
        out.append(ds_read_b128(S_(half, 4 * g), CA(hi), imm + CONST_OFF + 128 * half + 32 * g))
        out.append(ds_read_b128(DP(half, 4 * g), CA(hi), imm + CONST_OFF + 256 + 128 * half + 32 * g))
    return _tag(out, "frag")


def tile_ptrs():
    """64-bit sources of the tile this iteration loads (LT = min(it + 3, ntot - 1) -> head LHR, first query LQ0)"""
    return _tag([
        # Q / dO: base + head * 256 bytes + q0 * row bytes (row bytes = 16-row stride / 16)
        s_lshl_b32(TS, LHR, Lit(8)),
        s_mul_i32(TS2, LQ0, Q16B), isa.salu2("s_lshr_b32", TS2, TS2, Lit(4), lambda p, q, w: p >> (q & 31), lambda p, q, r: int(r != 0)),
        s_add_u32(TS2, TS2, TS), s_add_u32(PTRQ, QB, TS2), s_addc_u32(PTRQ + 1, QB + 1, Lit(0)),
        s_mul_i32(TS2, LQ0, D16B), isa.salu2("s_lshr_b32", TS2, TS2, Lit(4), lambda p, q, w: p >> (q & 31), lambda p, q, r: int(r != 0)),
        s_add_u32(TS2, TS2, TS), s_add_u32(PTRD, DOB, TS2), s_addc_u32(PTRD + 1, DOB + 1, Lit(0)),
        # row constants: plane (wave & 1 ? nd : ns) + (head * Sq + q0) * 4
        s_mul_i32(TS, LHR, SQ4), s_lshl_b32(TS2, LQ0, Lit(2)), s_add_u32(TS, TS, TS2), s_add_u32(TS, TS, WSEL),
        s_add_u32(PTRC, WSB, TS), s_addc_u32(PTRC + 1, WSB + 1, Lit(0))], "dma")


def advance_load_tile():
    """(LT, LHR, LQ0) -> the next tile of the sweep, clamped at the last one"""
    return _tag([s_add_i32(TS, LT, Lit(1)), s_cmp("lt", TS, NTOT), s_cbranch_scc(0, "NOADV_@"),
                 s_mov_b32(LT, TS), s_add_i32(LQ0, LQ0, Lit(64)), s_cmp("lt", LQ0, SQ4_ROWS), s_cbranch_scc(1, "NOADV_@"),
                 s_mov_b32(LQ0, QSTART), s_add_i32(LHR, LHR, Lit(1)), label("NOADV_@")], "dma")


SQ4_ROWS = s(35)            # Sq (rows)


def dma_piece(j, slot):
    base = slot * SLOT
    if j < 4:
        return _tag([s_m0_add(WAVEB, base + QOFF + j * 4096), global_load_lds_dwordx4(VOFFQ(j), PTRQ)], "dma")
    if j < 8:
        return _tag([s_m0_add(WAVEB, base + DOFF + (j - 4) * 4096), global_load_lds_dwordx4(VOFFD(j - 4), PTRD)], "dma")
    return _tag([s_m0_add(CWAVE, base + CONST_OFF), global_load_lds_dword(VOFFC, PTRC)], "dma")


CWAVE = s(56)               # lds base + (wave & 1) * 256: where this wave's row-constant piece lands (waves 2, 3 repeat those of 0, 1)


def global_load_lds_dword(voff, sbase):
    def fn(w):
        base = w.s64(sbase)
        m0 = w.m0
        w.wg.dma_issue_small(w, m0 & ~1023, 256)
        for l in range(64):
            src = base + int(w.R[voff, l])
            w.lds[m0 + 4 * l:m0 + 4 * l + 4] = w.gmem[src:src + 4]
    return I("global_load_lds_dword %s, %s" % (rname(voff), rrange(sbase, 2)), "vmem", (voff, sbase, sbase + 1, M0), (), fn, 16)


def stamp_start():
    return [s_memtime_wait(PREV)] if STAMPS else []


def stamp_acc(i):
    if not STAMPS:
        return []
    return [s_memtime_wait(STAMP), s_sub_u32(TS2, STAMP, PREV), s_add_u32(ACC(i), ACC(i), TS2), s_add_u32(ACC(i + 1), ACC(i + 1), Lit(1)), s_mov_b32(PREV, STAMP)]


def top():
    return stamp_start() + [s_waitcnt(vmcnt=9), s_barrier()] + stamp_acc(0)


def x_block(half):
    """16 MFMAs: per k-step S'(half) += Q rows x K~, dP'(half) += dO rows x V   (the accumulators start as the row constants)"""
    out = []
    for ks in range(8):
        out.append(v_mfma_32x32x16_bf16(S_(half), QRF(ks), KF(ks), S_(half)))
        out.append(v_mfma_32x32x16_bf16(DP(half), DORF(ks), VF(ks), DP(half)))
    return out


def y_block(half):
    """16 MFMAs: per 16-query step and head_dim block dV^T += dO^T P, dK^T += Q^T dS; transposed fragment (s2, which, dt) in ring slot"""
    out = []
    for s2 in range(2):
        for dt in range(4):
            out.append(v_mfma_32x32x16_bf16(DV(dt), TR(8 * 0 + (2 * dt) + 0), PF(half, s2), DV(dt)))
            out.append(v_mfma_32x32x16_bf16(DK(dt), TR(2 * dt + 1), DSF(half, s2), DK(dt)))
    return out


def soft_events(half, G0):
    """vector stream of one half: p = exp2(S'), dS = p dP', bf16 pairs of both.  Element r at gap G0 + (3 r) // 4 (16 elements over 12 gaps)."""
    ev = []
    for r in range(16):
        pos = G0 + (3 * r) // 4
        ev.append((pos, 0, v_exp_f32(S_(half, r), S_(half, r))))
        if not PACKED:
            ev.append((pos + 1, 1, v_mul_f32(DP(half, r), DP(half, r), S_(half, r))))
        elif r % 2 == 1:        # round 6: dS of two scores by one packed-f32 multiply
            ev.append((pos + 1, 1, v_pk_mul_f32(DP(half, r - 1), DP(half, r - 1), S_(half, r - 1))))
    for s2 in range(2):
        for j in range(4):
            r0 = 8 * s2 + 2 * j
            pos = G0 + (3 * (r0 + 1)) // 4
            ev.append((pos + 1, 2, v_cvt_pk_bf16_f32(PF(half, s2, j), S_(half, r0), S_(half, r0 + 1))))
            ev.append((pos + 2, 3, v_cvt_pk_bf16_f32(DSF(half, s2, j), DP(half, r0), DP(half, r0 + 1))))
    return ev


class Counter:
    def __init__(self):
        self.n = 0

    def new(self):
        self.n += 1
        return self.n


def diag_block(half, tag, cnt):
    """dynamic: only tiles that touch the diagonal (q0 + 32 half < key block + 32).  S' = -inf where key > query:
    key - query = (lane & 31) - 4 h - ACC_ROW[r] + (kb - q0 - 32 half)"""
    k = cnt.new()
    lbl = "NODIAG_%s_%d" % (tag, k)
    blk = [s_sub_i32(TS, KB, Q0S), s_add_i32(TS, TS, Lit(-32 * half)), v_add_u32(XH, XDIAG, TS)]
    for r in range(16):
        blk += [v_cmp_i32("gt", VCC, XH, Lit(ACC_ROW[r])), v_cndmask_b32(S_(half, r), S_(half, r), NEGINF, VCC)]
    head = [s_add_i32(TS, Q0S, Lit(32 * half + 1)), s_sub_i32(TS, TS, KB), s_cmp("ge", TS, Lit(32))]     # q0 + 32 half + 1 - kb >= 32: no key beyond any query
    return head + cond_block(s_cbranch_scc(1, lbl), blk, label(lbl))


def build_body(p, tag, cnt):
    """one query tile at ring phase p (slot p holds it, slot p + 1 the next one, slot p + 3 is being filled)"""
    slot, slot_n = p & 3, (p + 1) & 3
    slots = [[] for _ in range(NG)]
    mf = [None] * NG

    def put(g, *items):
        slots[max(0, min(NG - 1, g))].extend(items)

    mf[GXA:GXA + 16] = x_block(0)
    mf[GXB:GXB + 16] = x_block(1)
    mf[GYA:GYA + 16] = y_block(0)
    mf[GYB:GYB + 16] = y_block(1)
    # row fragments of half b (half a's were requested by the previous iteration / the prologue): k-step ks first used by MFMA 2 ks of X_b
    for ks in range(8):
        put(GXB + 2 * ks - LEAD_ROW, *row_read(QRF, ks, slot, 1, ks, False))
        put(GXB + 2 * ks - LEAD_ROW, *row_read(DORF, ks, slot, 1, ks, True))
    # half a, k-steps 3..7 (0..2 came with the previous iteration)
    for ks in range(NPRE, 8):
        put(GXA + 2 * ks - LEAD_ROW, *row_read(QRF, ks, slot, 0, ks, False))
        put(GXA + 2 * ks - LEAD_ROW, *row_read(DORF, ks, slot, 0, ks, True))
    # row constants of half b straight into its accumulators (free since the previous tile's Y_b consumed their fragments)
    put(GXA + 2, *const_reads(slot, 1))
    # transposed fragments: Y block MFMA 2 (4 s2 + dt) uses dO^T (ring slot 2 dt), MFMA + 1 uses Q^T (ring slot 2 dt + 1)
    for half, G in ((0, GYA), (1, GYB)):
        for s2 in range(2):
            for dt in range(4):
                first = G + 2 * (4 * s2 + dt)
                put(first - LEAD, *tr_reads(2 * dt, slot, half, s2, dt, True))
                put(first - LEAD + 1, *tr_reads(2 * dt + 1, slot, half, s2, dt, False))
    # next tile: row constants of half a into S'_a / dP'_a (their fragments were converted during X_b), row fragments 0..2
    put(GYB + 2, *const_reads(slot_n, 0))
    for ks in range(NPRE):
        put(NG - LEAD_ROW + 2 * ks, *row_read(QRF, ks, slot_n, 0, ks, False))
        put(NG - LEAD_ROW + 2 * ks, *row_read(DORF, ks, slot_n, 0, ks, True))
    # LDS-DMA of the tile three ahead
    for j, g in enumerate(DMA_GAPS):
        put(g, dma_piece(j, (p + 3) & 3))
    # vector streams
    ev = [(pos, o, i) for pos, o, i in soft_events(0, GXB + SOFT0)] + [(pos, o, i) for pos, o, i in soft_events(1, GYA + SOFT0)]
    for pos, o, i in sorted(ev, key=lambda t: (t[0], t[1])):
        put(pos, _tag([i], "soft")[0])
    if "diag" not in ABLATE:
        put(GXB + 1, _tag(diag_block(0, tag, cnt), "max"))
        put(GYA + 1, _tag(diag_block(1, tag, cnt), "max"))
    seq = []
    for g in range(NG):
        seq.append(mf[g])
        seq.extend(flatten(slots[g]))
    return [i for i in seq if not _ablated(i)]


def _ablated(i):
    if not ABLATE:
        return False
    return ("dma" in ABLATE and i.kind == "vmem") or ("soft" in ABLATE and i.tag == "soft") or ("frag" in ABLATE and i.tag == "frag") or \
        ("mfma" in ABLATE and i.kind == "mfma")


def entry_pending():
    out = []
    for g in range(4):
        out += [tuple(range(S_(0, 4 * g), S_(0, 4 * g) + 4)), tuple(range(DP(0, 4 * g), DP(0, 4 * g) + 4))]
    for ks in range(NPRE):
        out += [tuple(range(QRF(ks), QRF(ks) + 4)), tuple(range(DORF(ks), DORF(ks) + 4))]
    return out


def common_scalars():
    out = [s_lshl_b32(Q64B, Q16B, Lit(2)), s_lshl_b32(D64B, D16B, Lit(2))]
    for j in range(1, 4):
        out += [s_mul_i32(TS, Q16B, Lit(j)), v_add_u32(VOFFQ(j), VOFFQ0, TS), s_mul_i32(TS2, D16B, Lit(j)), v_add_u32(VOFFD(j), VOFFD0, TS2)]
    return out


def dma_prologue_code():
    """a statement of its own in the kernel (UR_ATTN_DKV_C128_DMA_ASM), issued as soon as the previous key block's loop has been left
    by every wave -- before the C++ part stores that block's results and loads this block's K / V rows: the first tile's LDS-DMA
    (Q, dO, row constants: 9 pieces)"""
    out = [comment("---- prologue LDS-DMA (tile 0)")] + common_scalars()
    out += [s_mov_b32(LT, Lit(0)), s_mov_b32(LHR, Lit(0)), s_mov_b32(LQ0, QSTART), s_nop(3)]
    out += tile_ptrs()
    for j in range(9):
        out += dma_piece(j, 0)
    return out


def prologue_code(with_dma_prologue=True):
    """state + the first three tiles' LDS-DMA + the first tile's row constants and fragments.  with_dma_prologue = False: tile 0 has
    been requested by the statement above; this one re-derives the sweep state and requests tiles 1 and 2."""
    out = [comment("---- entry")]
    if STAMPS:
        out += [s_mov_b32(ACC(i), Lit(0)) for i in range(NACC)]
    out += common_scalars() + row_addr_setup()
    out += [s_mov_b32(LT, Lit(0)), s_mov_b32(LHR, Lit(0)), s_mov_b32(LQ0, QSTART), s_nop(3)]
    for t in range(3):
        if t > 0 or with_dma_prologue:
            out += tile_ptrs()
            for j in range(9):
                out += dma_piece(j, t)
        out += _relabel(advance_load_tile(), "P%d" % t)
    for i in range(128):
        out.append(v_accvgpr_write(a(i), Lit(0)))
    out += [v_mov_b32(NEGINF, Lit(0xFF800000)), s_mov_b32(IT, Lit(0)), s_mov_b32(Q0S, QSTART), s_mov_b32(HRS, Lit(0))]
    # tile 0 is needed before the first barrier of the loop guarantees it: wait for it here (its 9 pieces are older than the 18 of
    # tiles 1, 2 -- and so is whatever the C++ part loaded or stored between the two statements)
    out += [s_waitcnt(vmcnt=18), s_barrier()]
    out += const_reads(0, 0)
    for ks in range(NPRE):
        out += row_read(QRF, ks, 0, 0, ks, False) + row_read(DORF, ks, 0, 0, ks, True)
    return out


def _relabel(items, suffix):
    out = []
    for it in flatten(items):
        if it.kind == "label" and it.target and "@" in it.target:
            out.append(label(it.target.replace("@", suffix)))
        elif it.kind == "branch" and "@" in it.target:
            b = s_cbranch_scc(int(it.text.split("scc")[1][0]), it.target.replace("@", suffix))
            b.tag = it.tag
            out.append(b)
        else:
            out.append(it)
    return out


def build_dma_program():
    P = Program()
    P.add(fix_hazards(dma_prologue_code())[0])
    P.finalize()
    return P


def build_program(with_dma_prologue=True):
    """with_dma_prologue = False: the kernel's main statement (tile 0 comes from build_dma_program's statement); the emulator runs the
    single-statement form"""
    cnt = Counter()
    P = Program()
    P.add(fix_hazards(prologue_code(with_dma_prologue))[0])
    P.add(s_cmp("ge", IT, NTOT), s_cbranch_scc(1, "EXIT"))
    bodies = {}
    for p in range(4):
        name = "BODY_%d" % p
        seq = top() + tile_ptrs() + build_body(p, name, cnt) + _relabel(advance_load_tile(), "B%d" % p)
        # this tile's place in the sweep -> the next one (Q0S, HRS are what the diagonal test reads)
        seq += [s_add_i32(Q0S, Q0S, Lit(64)), s_cmp("lt", Q0S, SQ4_ROWS), s_cbranch_scc(1, "SAMEHEAD_%d" % p), s_mov_b32(Q0S, QSTART),
                s_add_i32(HRS, HRS, Lit(1)), label("SAMEHEAD_%d" % p)]
        seq += stamp_acc(2)
        if BALANCE:
            seq = balance.balance(flatten(seq), entry_lgkm=entry_pending(), name="dkv")
        fixed, _ = fix_hazards(seq, entry_lgkm=entry_pending())
        bodies[name] = fixed
        P.add(label(name), fixed, s_add_i32(IT, IT, Lit(1)), s_cmp("ge", IT, NTOT), s_cbranch_scc(1, "EXIT"))
        if p == 3:
            P.add(s_branch("BODY_0"))
    if STAMPS:
        P.add(label("EXIT"))
        for i in range(NACC):
            P.add(v_mov_b32(TMPA, ACC(i)), v_mov_b32(XH, Lit(0)), global_store_dword_s(TMPA, XH, DBGPTR, 4 * i))
        P.add(s_waitcnt(vmcnt=0, lgkmcnt=0), s_nop(15))
    else:
        P.add(label("EXIT"), s_waitcnt(vmcnt=0, lgkmcnt=0), s_nop(15))
    P.finalize()
    return P, bodies


if __name__ == "__main__":
    import collections
    P, bodies = build_program()
    print("instructions:", len(P.ins), P.stats())
    seq = bodies["BODY_1"]
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
    print(dict(c), "total", sum(c.values()), "sum max32", sum(max(32, x) for x in costs), "mfma*32", 32 * c["mfma"])
    print(costs)

"""Generator of the hand-scheduled causal attention FORWARD main loop (head_dim 128, bf16, gfx950).

What is generated (one inline-asm block, `unirec_amd/csrc/gen/attn_fwd_c128.inc`, included by csrc/attn_c128.hip):
the whole key-tile loop of one wave.  A workgroup is 4 waves = 256 query rows of one (batch, query head); a wave owns 64 query
rows (two 32-row blocks qb = 0, 1) and the WHOLE register file of its SIMD (one wave per SIMD, 512 registers):

  a[0:127]    O^T accumulators  [qb][dt = head_dim/32][16]          a[128:191]  Q fragments [qb][k-step 0..7] (pre-scaled by
  a[192:223]  row sums l [qb][16] (every register = the sum)                     scale*log2e in the C++ prologue)
  a[224:255]  K row-fragment ring (8 slots)
  v[16:79]    S'^T accumulators [sub = key/32][qb][16]               v[80:111]   -m replicated 16x per qb (MFMA C operand)
  v[112:143]  P fragments (bf16) [sub][qb][k16-step]                 v[144:175]  V^T fragment ring (8 slots)

Per 64-key tile and wave: 72 MFMAs (v_mfma_f32_32x32x16_bf16): S'^T = K Q~^T - m with the running maximum as the C operand of
the chain head, so p = exp2(S') is ONE v_exp_f32 per score; O^T += V^T P with the P accumulators converted in place; the row sums
l += 1^T P ride on the matrix pipe too (8 MFMAs against an all-ones fragment: the kernel is bound by instruction ISSUE at one
wave per SIMD, an MFMA costs 8 issue cycles where the 64 v_add_f32 it replaces cost 256, and l is re-based exactly like O).
Every K / V fragment read from LDS feeds two MFMAs (both query blocks).  Software pipeline over tiles (MFMA order per iteration:
S sub0 (tile i) | P V sub0 (tile i-1) | S sub1 (tile i) | P V sub1 (tile i-1)); the vector work of a tile is spread over the 64 MFMA
gaps at one v_exp_f32 per gap.  The running maximum is DEFERRED: it moves only when a row's new maximum exceeds the current one
by more than 2^THR, through an out-of-line path that re-bases the pending scores at once and O / l at the next point in the
stream where no contribution in the old base is outstanding (ALPHA / PEND registers).
K / V tiles arrive by LDS-DMA into two 4-slot rings (K three tiles ahead, V two), one workgroup barrier per tile; the loop is
unrolled over the 4 ring phases so every LDS address is a lane-constant register plus an immediate.

tools/asmgen/isa.py emulates the result on the CPU (tests/test_asmgen_attn_fwd.py) and checks every counted wait and hazard.
Reference semantics: transformers modeling_qwen3.py:185-208 (SDPA, causal + key padding), called from
/root/reference/training/train_item_individual_token_joint.py:173-177.
"""
import os

from isa import *      # noqa: F401,F403
import isa

THR = 8.0              # log2 units: the running maximum is left alone while a row's scores stay below m + THR
LEADK, LEADV = 6, 6    # MFMA gaps between the issue of an LDS fragment read and its first use
KSLOT, VSLOT = 16384, 16384
VBASE_LDS, BIAS_LDS = 65536, 131072
LDS_BYTES = BIAS_LDS + 16384 + 64


# ---- register map -------------------------------------------------------------------------------------------------
def S_(sub, qb, r=0):
    return v(16 + 16 * (2 * sub + qb) + r)


def MNEG(qb, r=0):
    return v(80 + 16 * qb + r)


def P_(sub, qb, s2, j=0):
    return v(112 + 4 * ((sub * 2 + qb) * 2 + s2) + j)


def VF(slot, j=0):
    return v(144 + 4 * slot + j)


def KA(ks):
    return v(176 + ks)


def TA(dt):
    return v(184 + dt)


def TB(dt):
    return v(188 + dt)


def M_(qb):
    return v(192 + qb)


def ONES(j=0):
    return v(244 + j)          # 4 registers of packed bf16 1.0: the A operand of the row-sum MFMAs


def U_(qb):
    return v(196 + qb)


def T_(qb):
    return v(198 + qb)


def ALPHA(site, qb):
    return v(200 + 2 * site + qb)


VOFFK0, VOFFV0, NEGINF, THRV, BIASADDR, DIAGX, TMPA = v(204), v(205), v(206), v(207), v(208), v(209), v(210)


def VOFFK(j):
    return v(248 + j)          # per-lane source offset of K piece j (rows 16 j ..): VOFFK0 + j * K16B


def VOFFV(j):
    return v(252 + j)


def D_(qb):
    return v(211 + qb)


def TMP(i):
    return v(213 + i)          # 213..215, rare paths also use the BIAS block


def BIAS(r):
    return v(216 + r)          # 16 registers: key bias of the sub-tile being masked; scratch of the rare paths


def O_(qb, dt, r=0):
    return a(16 * (4 * qb + dt) + r)


def Q_(qb, ks):
    return a(128 + 4 * (8 * qb + ks))


def LA(qb, r=0):
    return a(192 + 16 * qb + r)


def KF(slot):
    return a(224 + 4 * (slot & 7))


KBASE, VBASE, K16B, V16B, IT, TEND, TLAST, TFIRST, MASKBITS = s(36), s(38), s(40), s(41), s(42), s(43), s(44), s(45), s(46)
K64B, V64B, PTRK, PTRV, RET, PEND0, PEND1, WAVEB = s(48), s(49), s(50), s(52), s(54), s(55), s(56), s(57)
C0, C1, NOTINIT0, NOTINIT1, TS, TENDM1, TLASTP1, TS2 = s(58), s(60), s(62), s(64), s(66), s(67), s(34), s(35)


def NOTINIT(qb):
    return NOTINIT0 if qb == 0 else NOTINIT1


def PEND(site):
    return PEND0 if site == 0 else PEND1


# lab (UR_ASMGEN_STAMPS=1): cycle stamps accumulated per wave and written to a debug buffer at the end (tools/lab/c128_stamps.py)
STAMP, PREV, DBGPTR = s(68), s(70), s(72)
NACC = 12


def ACC(i):
    return s(74 + i)


ASM_VGPR_FIRST = 16          # v0..v15 stay with the compiler
ACC_ROW = [(r & 3) + 8 * (r >> 2) for r in range(16)]


# ---- extra scalar builders --------------------------------------------------------------------------------------------
def s_andn2_b64(d, x, y):
    def fn(w):
        g = lambda z: w.vcc if z == VCC else w.s64(z)
        r = g(x) & ~g(y) & 0xFFFFFFFFFFFFFFFF
        w.scc = int(r != 0)
        if d == VCC:
            w.vcc = r
        else:
            w.sset64(d, r)
    t = lambda z: "vcc" if z == VCC else rrange(z, 2)
    rd = tuple(r for z in (x, y) for r in ((VCC,) if z == VCC else (z, z + 1)))
    return I("s_andn2_b64 %s, %s, %s" % (t(d), t(x), t(y)), "salu", rd, ((VCC,) if d == VCC else (d, d + 1)) + (SCC,), fn, 1)


def s_or_b32(d, x, y):
    return isa.salu2("s_or_b32", d, x, y, lambda p, q, w: p | q, lambda p, q, r: int((r & 0xFFFFFFFF) != 0))


def s_m0_add(x, const):
    """m0 = s[x] + const"""
    def fn(w):
        w.m0 = (w.sget(x) + const) & 0xFFFFFFFF
        w.scc = 0
    return I("s_add_u32 m0, %s, %s" % (rname(x), Lit(const).text), "salu", (x,), (M0, SCC), fn, 1)


# ---- pieces -------------------------------------------------------------------------------------------------------------
def dma_setup(kadd=3, vadd=2, base=IT):
    """64-bit source bases of the tiles this iteration loads: K tile min(it+3, tend-1), V tile min(it+2, tend-1)"""
    return _tag([s_add_i32(TS, base, Lit(kadd)), s_min_i32(TS, TS, TENDM1), s_mul_i32(TS2, TS, K64B),
                 s_add_u32(PTRK, KBASE, TS2), s_addc_u32(PTRK + 1, KBASE + 1, Lit(0)),
                 s_add_i32(TS, base, Lit(vadd)), s_min_i32(TS, TS, TENDM1), s_mul_i32(TS2, TS, V64B),
                 s_add_u32(PTRV, VBASE, TS2), s_addc_u32(PTRV + 1, VBASE + 1, Lit(0))], "dma")


def _tag(items, tag):
    for it in flatten(items):
        if it.tag is None:
            it.tag = tag
    return items


def dma_piece(j, kslot, vslot):
    """piece j of this wave's 8 per tile (0-3: K rows 16j.., 4-7: V rows 16(j-4)..); LDS destination = slot + (4*jj + wave) KiB"""
    if j < 4:
        return _tag([s_m0_add(WAVEB, kslot * KSLOT + j * 4096), global_load_lds_dwordx4(VOFFK(j), PTRK)], "dma")
    return _tag([s_m0_add(WAVEB, VBASE_LDS + vslot * VSLOT + (j - 4) * 4096), global_load_lds_dwordx4(VOFFV(j - 4), PTRV)], "dma")


def k_read(slot, ks, kslot, sub):
    return _tag([ds_read_b128(KF(slot), KA(ks), kslot * KSLOT + sub * 8192)], "frag")[0]


def v_reads(f, vslot, sub):
    s2, dt = f >> 2, f & 3
    off = vslot * VSLOT + 256 * (32 * sub + 16 * s2)
    return _tag([ds_read_b64_tr_b16(VF(f, 0), TA(dt), off), ds_read_b64_tr_b16(VF(f, 2), TB(dt), off)], "frag")


STAMPS = os.environ.get("UR_ASMGEN_STAMPS", "") == "1"


def stamp_start():
    return [s_memtime_wait(PREV)] if STAMPS else []


def stamp_acc(i, restart=True):
    """ACC(i) += cycles since the previous stamp; ACC(i + 1) += 1"""
    if not STAMPS:
        return []
    out = [s_memtime_wait(STAMP), s_sub_u32(TS2, STAMP, PREV), s_add_u32(ACC(i), ACC(i), TS2), s_add_u32(ACC(i + 1), ACC(i + 1), Lit(1))]
    if restart:
        out.append(s_mov_b32(PREV, STAMP))
    return out


def top():
    if "bar" in ABLATE:
        return []
    # stamps: ACC(0) = cycles waiting at the top of an iteration (LDS-DMA landed + barrier)
    return stamp_start() + [s_waitcnt(vmcnt=8), s_barrier()] + stamp_acc(0)


class Sites:
    """out-of-line code and the return dispatch of the two rare subroutine pairs"""
    def __init__(self):
        self.n = 0
        self.resc = {0: [], 1: []}       # site ids that call RESC<sub>
        self.oresc = {0: [], 1: []}
        self.stubs = []

    def new(self):
        self.n += 1
        return self.n


def max_phase(sub, sites, skip_qb0, diag, tag):
    """instructions of the maximum phase of sub-tile `sub` (optional key-bias / diagonal masks first), ending in the deferred-maximum check"""
    out = []
    k = sites.new()
    # key-padding bias (dynamic: only tiles whose MASKBITS bit is set)
    lbl = "NOBIAS_%s_%d" % (tag, k)
    blk = [s_lshl_b32(TS2, IT, Lit(8)), v_add_u32(TMPA, BIASADDR, TS2)]
    for g in range(4):
        blk.append(ds_read_b128(BIAS(4 * g), TMPA, sub * 128 + g * 32))
    blk.append(s_waitcnt(lgkmcnt=0))
    for qb in range(2):
        if skip_qb0 and qb == 0:
            continue
        for r in range(16):
            blk.append(v_add_f32(S_(sub, qb, r), S_(sub, qb, r), BIAS(r)))
    out.append([s_bitcmp1_b64(MASKBITS, IT)] + cond_block(s_cbranch_scc(0, lbl), blk, label(lbl)))   # one unit: never split over gaps
    if diag:
        # causal diagonal of the wave's last tile: (sub0,qb0) and (sub1,qb1) triangular, (sub0,qb1) full, (sub1,qb0) empty
        # (key row of register r in lane half h is ACC_ROW[r] + 4 h; DIAGX = query - 4 h: keep the score where key <= query)
        qb = sub
        for r in range(16):
            out += [v_cmp_i32("ge", VCC, DIAGX, Lit(ACC_ROW[r])), v_cndmask_b32(S_(sub, qb, r), NEGINF, S_(sub, qb, r), VCC)]
    qbs = [1] if skip_qb0 else [0, 1]
    if skip_qb0:
        out.append(v_mov_b32(U_(0), NEGINF))
    for kk in range(8):
        for qb in qbs:
            if kk == 0:
                out.append(v_max3_f32(U_(qb), S_(sub, qb, 0), S_(sub, qb, 1), S_(sub, qb, 2)))
            elif kk < 7:
                out.append(v_max3_f32(U_(qb), U_(qb), S_(sub, qb, 2 * kk + 1), S_(sub, qb, 2 * kk + 2)))
            else:
                out.append(v_max_f32(U_(qb), U_(qb), S_(sub, qb, 15)))
    for qb in qbs:
        out.append(v_mov_b32(T_(qb), U_(qb)))
    for qb in qbs:
        out.append(v_permlane32_swap(U_(qb), T_(qb)))
    for qb in qbs:
        out.append(v_max_f32(U_(qb), U_(qb), T_(qb)))
    chk = [v_cmp_f32("lt", C0, THRV, U_(0)), v_cmp_f32("lt", C1, THRV, U_(1)), s_or_b64(C0, C0, C1), s_or_b64(C0, C0, NOTINIT0),
           s_or_b64(C0, C0, NOTINIT1), s_cbranch_scc(1, "RARE_%d" % k), label("BACK_%d" % k)]
    out.append(chk)
    sites.resc[sub].append(k)
    sites.stubs.append([label("RARE_%d" % k), s_mov_b32(RET, Lit(k)), s_branch("RESC%d" % sub)])
    return out


def softmax_events(sub, G0, skip_qb0):
    """(position, order, instruction) of the exp / bf16-conversion stream of a sub-tile whose exps start at gap G0"""
    ev = []
    for e in range(32):
        r, qb = e >> 1, e & 1
        if skip_qb0 and qb == 0:
            continue
        ev.append((G0 + e, 2, v_exp_f32(S_(sub, qb, r), S_(sub, qb, r))))
    for qb in range(2):
        if skip_qb0 and qb == 0:
            continue
        for s2 in range(2):
            for j in range(4):
                r0 = 8 * s2 + 2 * j
                elast = 2 * (r0 + 1) + qb
                ev.append((G0 + elast + 2, 3, v_cvt_pk_bf16_f32(P_(sub, qb, s2, j), S_(sub, qb, r0), S_(sub, qb, r0 + 1))))
    return ev


def pend_test(site, sites):
    k = sites.new()
    sites.oresc[site].append(k)
    sites.stubs.append([label("ORARE_%d" % k), s_mov_b32(RET, Lit(k)), s_branch("ORESC%d" % site)])
    return [s_cmp("lg", PEND(site), Lit(0)), s_cbranch_scc(1, "ORARE_%d" % k), label("OBACK_%d" % k)]


# MFMA blocks of an iteration: A = S sub0 (16) | B = P V sub0 + row sums (20) | C = S sub1 (16) | D = P V sub1 + row sums (20)
GA, GB, GC, GD, NG = 0, 16, 36, 52, 72
G0S = (24, 60)              # first exp gap of the two sub-tiles' softmax streams (one v_exp_f32 per gap; 64 exps in 72 gaps)
MAXG = (18, 54)             # first gap of their maximum phases (6 gaps, the check closes the last)
PENDG = (8, 44)             # gaps of the two "re-base O, l" tests (no contribution in the old base outstanding there)
DMA_GAPS = [2, 10, 14, 30, 38, 46, 50, 66]


def pv_block(sub, skip_qb0):
    """the 20 MFMAs of a P V block: per 16-key step the 8 products O^T[qb][dt] += V^T P, then the two row-sum products"""
    out = []
    for s2 in range(2):
        for dt in range(4):
            for qb in range(2):
                out.append(None if (skip_qb0 and qb == 0) else v_mfma_32x32x16_bf16(O_(qb, dt), VF(4 * s2 + dt), P_(sub, qb, s2), O_(qb, dt)))
        for qb in range(2):
            out.append(None if ((skip_qb0 and qb == 0) or "rowsum" in os.environ.get("UR_ASMGEN_ABLATE", "")) else v_mfma_32x32x16_bf16(LA(qb), ONES(), P_(sub, qb, s2), LA(qb)))
    return out


def build_body(p, kind, last, sites, tag, with_dma=True):
    """one iteration at ring phase p.  kind: 'pro' (first tile of the wave: no P V of a previous tile), 'steady', 'epi' (only the
    P V of the wave's last tile).  last: the tile is the wave's last = its causal diagonal tile."""
    do_S, do_PV, carry = kind != "epi", kind != "pro", kind != "pro"
    prev_last = kind == "epi"
    kslot, kslot_n, vslot_prev = p & 3, (p + 1) & 3, (p - 1) & 3
    slots = [[] for _ in range(NG)]
    pre, mf = [], [None] * NG

    def put(g, *items):
        slots[g].extend(items)

    # ---- MFMAs
    if do_S:
        for g in range(16):
            ks, qb = g >> 1, g & 1
            mf[GA + g] = v_mfma_32x32x16_bf16(S_(0, qb), KF(ks), Q_(qb, ks), MNEG(qb) if ks == 0 else S_(0, qb))
            if not (last and qb == 0):
                mf[GC + g] = v_mfma_32x32x16_bf16(S_(1, qb), KF(ks), Q_(qb, ks), MNEG(qb) if ks == 0 else S_(1, qb))
    if do_PV:
        mf[GB:GB + 20] = pv_block(0, False)
        mf[GD:GD + 20] = pv_block(1, prev_last)
    # ---- LDS fragment reads (V fragment f = (16-key step s2, head_dim block dt) is first used by MFMA 10 s2 + 2 dt of its block)
    if do_PV:
        for f in range(8):
            first = 10 * (f >> 2) + 2 * (f & 3)
            put(GB + first - LEADV, *v_reads(f, vslot_prev, 0))
            put(GD + first - LEADV, *v_reads(f, vslot_prev, 1))
    if do_S:
        for ks in range(8):
            put(GC + 2 * ks - LEADK, k_read(ks, ks, kslot, 1))
        if kind == "pro":
            for ks in range(3):
                pre.append(k_read(ks, ks, kslot, 0))
        for ks in range(3, 8):
            put(2 * ks - LEADK, k_read(ks, ks, kslot, 0))
        if not last:
            for ks in range(3):
                put(NG - LEADK + 2 * ks, k_read(ks, ks, kslot_n, 0))
    # ---- LDS-DMA of the tiles three / two ahead
    if with_dma:
        for j, g in enumerate(DMA_GAPS):
            put(g, dma_piece(j, (p + 3) & 3, (p + 2) & 3))
    # ---- softmax streams
    ev = []
    if carry:
        ev += [(pos - NG, o, i) for pos, o, i in softmax_events(1, G0S[1], prev_last) if pos >= NG]
    if do_S:
        ev += [(pos, o, i) for pos, o, i in softmax_events(0, G0S[0], False) if pos < NG]
        ev += [(pos, o, i) for pos, o, i in softmax_events(1, G0S[1], last) if pos < NG]
    for pos, o, i in sorted(ev, key=lambda t: (t[0], t[1])):
        put(pos, _tag([i], "soft")[0])
    if do_S:
        for sub in range(2):
            items = _tag(max_phase(sub, sites, last and sub == 1, last, tag), "max")
            # units: sub-lists stay together; spread over the 6 gaps in order, the check last
            n = len(items)
            per = (n + 5) // 6
            for idx, itx in enumerate(items):
                g = MAXG[sub] + min(idx // per, 5)
                if idx == n - 1:
                    g = MAXG[sub] + 5
                put(g, itx)
    if do_PV:
        for site in range(2):
            slots[PENDG[site]] = pend_test(site, sites) + slots[PENDG[site]]
    # ---- linearise
    seq = list(pre)
    for g in range(NG):
        if mf[g] is not None:
            seq.append(mf[g])
        seq.extend(flatten(slots[g]))
    return [i for i in seq if not _ablated(i)]


ABLATE = set(x for x in os.environ.get("UR_ASMGEN_ABLATE", "").split(",") if x)   # lab: timing-only builds (results WRONG)
BALANCE = os.environ.get("UR_ASMGEN_BALANCE_FWD", os.environ.get("UR_ASMGEN_BALANCE", "1")) == "1"      # round 6: dependency-aware re-placement of the side instructions (balance.py)


def _ablated(i):
    if not ABLATE:
        return False
    return ("dma" in ABLATE and i.kind == "vmem") or ("dmaall" in ABLATE and i.tag == "dma") or ("soft" in ABLATE and i.tag == "soft") or \
        ("max" in ABLATE and i.tag == "max" and i.kind not in ("label", "branch")) or ("frag" in ABLATE and i.tag == "frag") or \
        ("mfma" in ABLATE and i.kind == "mfma") or ("exp" in ABLATE and i.kind == "trans")


def flatten(items):
    out = []
    for it in items:
        if isinstance(it, (list, tuple)):
            out.extend(flatten(it))
        elif it is not None:
            out.append(it)
    return out


def entry_pending():
    """LDS reads that may be outstanding when a steady body is entered: K sub0 fragments 0..2 of its tile"""
    return [tuple(range(KF(ks), KF(ks) + 4)) for ks in range(3)]


def resc_routine(sub, sites):
    """re-base the running maximum: entered from the check of sub-tile `sub` with U = row maxima of S'(sub) (both lane halves equal).
    Per lane d = max(u, 0) (first valid maximum of a row: d = u); m += d; S'(sub) -= d; -m replicated for the next chains;
    ALPHA(sub) = 2^-d is applied to O and l at the next clean point of the stream (PEND: oresc_routine)."""
    out = [label("RESC%d" % sub)]
    acc = C1
    out.append(s_mov_b64(acc, Lit(0)))
    for qb in range(2):
        A, DN, DI, NT = TMP(0), TMP(1), TMP(2), BIAS(0)
        out += [v_cmp_f32("lt", VCC, Lit(-3.0e38), U_(qb)),                 # valid: the row has seen a key
                v_max_f32(DN, U_(qb), Lit(0.0)),
                v_cndmask_b32(DI, Lit(0.0), U_(qb), VCC),
                v_cndmask_b32(D_(qb), DN, DI, NOTINIT(qb)),
                v_sub_f32(NT, Lit(0.0), DN), v_exp_f32(A, NT),
                v_cndmask_b32(ALPHA(sub, qb), A, Lit(1.0), NOTINIT(qb)),
                v_add_f32(M_(qb), M_(qb), D_(qb)),
                s_andn2_b64(C0, VCC, NOTINIT(qb)),                            # initialised rows that moved
                s_or_b64(acc, acc, C0),
                s_andn2_b64(NOTINIT(qb), NOTINIT(qb), VCC),
                v_sub_f32(NT, Lit(0.0), M_(qb))]
        for r in range(16):
            out.append(v_mov_b32(MNEG(qb, r), NT))
        for r in range(16):
            out.append(v_sub_f32(S_(sub, qb, r), S_(sub, qb, r), D_(qb)))
    out += [s_cmp_lg_u64(acc), s_cselect_b32(TS, Lit(1), Lit(0)), s_or_b32(PEND(sub), PEND(sub), TS), s_nop(3)]
    for k in sites.resc[sub]:
        out += [s_cmp("eq", RET, Lit(k)), s_cbranch_scc(1, "BACK_%d" % k)]
    out.append(s_branch("TRAP"))
    return out


def oresc_routine(site, sites):
    out = [label("ORESC%d" % site), s_nop(15)]
    n = 0
    for qb in range(2):
        for reg in [O_(qb, dt, r) for dt in range(4) for r in range(16)] + [LA(qb, r) for r in range(16)]:
            t = BIAS(n % 8)
            n += 1
            out += [v_accvgpr_read(t, reg), v_mul_f32(t, t, ALPHA(site, qb)), v_accvgpr_write(reg, t)]
    for qb in range(2):
        out.append(v_mov_b32(ALPHA(site, qb), Lit(1.0)))
    out += [s_mov_b32(PEND(site), Lit(0)), s_nop(3)]
    for k in sites.oresc[site]:
        out += [s_cmp("eq", RET, Lit(k)), s_cbranch_scc(1, "OBACK_%d" % k)]
    out.append(s_branch("TRAP"))
    return out


def common_scalars():
    out = [s_add_i32(TENDM1, TEND, Lit(-1)), s_lshl_b32(K64B, K16B, Lit(2)), s_lshl_b32(V64B, V16B, Lit(2)),
           v_mov_b32(VOFFK(0), VOFFK0), v_mov_b32(VOFFV(0), VOFFV0)]
    for j in range(1, 4):
        out += [s_mul_i32(TS, K16B, Lit(j)), v_add_u32(VOFFK(j), VOFFK0, TS), s_mul_i32(TS2, V16B, Lit(j)), v_add_u32(VOFFV(j), VOFFV0, TS2)]
    return out


PROLOGUE_TILES = (("k", 0), ("v", 0), ("k", 1), ("k", 2), ("v", 1))      # issue order (ring slot = tile - t0; tile indices clamped to tend-1)


def prologue_pieces(tiles):
    out = []
    for which, add in tiles:
        grp = dma_setup(kadd=add, vadd=add, base=TFIRST)[:5] if which == "k" else dma_setup(kadd=add, vadd=add, base=TFIRST)[5:]
        for j in (range(4) if which == "k" else range(4, 8)):
            out.append(grp + dma_piece(j, add, add))
            grp = []
    return out


def dma_prologue_code():
    """a statement of its own in the kernel (UR_ATTN_FWD_C128_DMA_ASM), issued before the C++ part scales q: the FIRST tile's
    LDS-DMA K(t0), V(t0).  The other three tiles of the prologue are issued by the main statement between its initialisation
    instructions (a burst of back-to-back pieces waits ~150 cycles per piece on the LDS-DMA path; vector work hides that)."""
    out = [comment("---- prologue LDS-DMA")] + common_scalars() + [s_nop(3)]
    for piece in prologue_pieces(PROLOGUE_TILES[:2]):
        out += piece
    return out


def entry_code():
    out = [comment("---- entry of the main statement: scalars and state")]
    if STAMPS:
        out += [s_mov_b32(ACC(i), Lit(0)) for i in range(NACC)]
    out += common_scalars() + [s_add_i32(TLASTP1, TLAST, Lit(1)), s_mov_b32(IT, TFIRST), s_nop(3)]
    init = [v_accvgpr_write(a(i), Lit(0)) for i in range(128)]
    for qb in range(2):
        init += [v_mov_b32(M_(qb), Lit(0)), v_mov_b32(ALPHA(0, qb), Lit(1.0)), v_mov_b32(ALPHA(1, qb), Lit(1.0))]
        for r in range(16):
            init += [v_mov_b32(MNEG(qb, r), Lit(0)), v_accvgpr_write(LA(qb, r), Lit(0))]
    for j in range(4):
        init.append(v_mov_b32(ONES(j), Lit(0x3F803F80)))
    init += [v_mov_b32(NEGINF, Lit(0xFF800000)), v_mov_b32(THRV, Lit(float(THR)))]
    # the remaining prologue tiles' LDS-DMA pieces, one every ~20 initialisation instructions
    pieces = prologue_pieces(PROLOGUE_TILES[2:])
    step = max(1, len(init) // len(pieces))
    for n, piece in enumerate(pieces):
        out += piece + init[n * step:(n + 1) * step if n + 1 < len(pieces) else len(init)]
    out += [s_mov_b32(PEND0, Lit(0)), s_mov_b32(PEND1, Lit(0)), s_mov_b64(NOTINIT0, Lit(-1)), s_mov_b64(NOTINIT1, Lit(-1))]
    return out


def build_dma_program():
    P = Program()
    P.add(fix_hazards(dma_prologue_code())[0])
    P.finalize()
    return P


def build_program(with_dma_prologue=True):
    """with_dma_prologue: the emulator runs the two statements of the kernel back to back as one program"""
    sites = Sites()
    P = Program()
    if with_dma_prologue:
        P.add(fix_hazards(dma_prologue_code())[0])
    P.add(fix_hazards(entry_code())[0])
    # waves without a tile only keep the ring protocol
    P.add(s_cmp("lt", TLAST, TFIRST), s_cbranch_scc(0, "HAVE"), s_add_i32(TLAST, TFIRST, Lit(-2)), s_add_i32(TLASTP1, TFIRST, Lit(-1)),
          s_branch("D_0"), label("HAVE"), s_cmp("eq", TLAST, TFIRST), s_cbranch_scc(1, "PROL"))
    bodies = {}

    def emit_body(name, p, kind, last, with_top=True):
        seq = []
        if with_top:
            seq += top()
        seq += dma_setup()
        seq += build_body(p, kind, last, sites, name)
        seq += stamp_acc({"PRO": 2, "PROL": 2}.get(name, 6 if last else 4))
        if BALANCE:
            import balance
            seq = balance.balance(flatten(seq), entry_lgkm=entry_pending() if kind == "steady" else (), temps={TMPA}, name="fwd")
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
        # SKIP: ring protocol only
        skip = top() + dma_setup()
        for j in range(8):
            skip += dma_piece(j, (p + 3) & 3, (p + 2) & 3)
        skip += stamp_acc(10)
        P.add(comment("---- SKIP_%d" % p), fix_hazards(skip)[0], s_add_i32(IT, IT, Lit(1)), s_branch(nxt))
        P.add(label("STEADY_%d" % p), emit_body("STEADY_%d" % p, p, "steady", False), s_add_i32(IT, IT, Lit(1)), s_branch(nxt))
        P.add(label("LAST_%d" % p), emit_body("LAST_%d" % p, p, "steady", True), s_add_i32(IT, IT, Lit(1)), s_branch(nxt))
        # EPI: the iteration after the wave's last tile; it has a ring step of its own unless it == tend
        P.add(label("EPI_%d" % p), s_cmp("ge", IT, TEND), s_cbranch_scc(1, "EPIB_%d" % p))
        pre = top() + dma_setup()
        for j in range(8):
            pre += dma_piece(j, (p + 3) & 3, (p + 2) & 3)
        P.add(fix_hazards(pre)[0], label("EPIB_%d" % p))
        seq = stamp_start() + build_body(p, "epi", False, sites, "EPI_%d" % p, with_dma=False) + stamp_acc(8)
        P.add(fix_hazards(seq)[0], s_add_i32(IT, IT, Lit(1)), s_branch(nxt))
    P.add(sites.stubs)
    for sub in range(2):
        P.add(fix_hazards(resc_routine(sub, sites))[0])
    for site in range(2):
        P.add(fix_hazards(oresc_routine(site, sites))[0])
    P.add(label("TRAP"), I("s_trap 2", "salu", (), (), lambda w: (_ for _ in ()).throw(RuntimeError("TRAP reached")), 1))
    # the compiler's code behind the statement reads O (v_accvgpr_read): MFMA results need their wait states.  LDS-DMA pieces of the
    # last iterations may still be in flight: vmcnt is in order, so the next block's first ring wait covers them, and the kernel
    # drains them before it ends (attn.hip).
    if STAMPS:
        # every wave writes its NACC accumulators: DBGPTR already points at this wave's record
        P.add(label("EXIT"))
        for i in range(NACC):
            P.add(v_mov_b32(TMP(0), ACC(i)), v_mov_b32(TMP(1), Lit(0)), global_store_dword_s(TMP(0), TMP(1), DBGPTR, 4 * i))
        P.add(s_waitcnt(vmcnt=0, lgkmcnt=0), s_nop(15))
        P.finalize()
        return P, bodies
    P.add(label("EXIT"), s_waitcnt(lgkmcnt=0), s_nop(15))
    P.finalize()
    return P, bodies


def gap_report(seq):
    """issue cost between consecutive MFMAs of a body (cycles at one wave per SIMD: MFMA 8, v_exp 8, other vector / LDS 4)"""
    costs, cur = [], None
    for it in seq:
        if it.kind == "mfma":
            if cur is not None:
                costs.append(cur)
            cur = 8
        elif cur is not None and it.kind != "label":
            cur += it.cost
    if cur is not None:
        costs.append(cur)
    return costs


if __name__ == "__main__":
    import sys
    P, bodies = build_program()
    print("instructions:", len(P.ins), P.stats())
    for name in ("PRO", "STEADY_1", "LAST_1"):
        c = gap_report(bodies[name])
        print(name, "gaps", len(c), "sum", sum(c), "max", max(c), "over32:", sum(1 for x in c if x > 32))
        print("   ", c)

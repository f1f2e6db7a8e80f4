"""gfx950 instruction model for the hand-scheduled attention kernels (tools/asmgen).

Three things live here, for the ~40 opcodes the generated kernels use:
  * text     -- the assembler line that goes into the inline-asm block of the .hip file
  * emulate  -- a 64-lane numpy semantic of the instruction, so a generated schedule is checked against a numpy
                attention reference ON THE CPU before it ever reaches an MI355X (tests/test_asmgen_*.py)
  * hazards  -- the software-visible ordering rules the hardware does not interlock (MFMA result -> vector read,
                vector write -> MFMA operand, transcendental forwarding, v_permlane operands, M0 -> LDS-DMA) and the
                counted-wait bookkeeping (every LDS read / LDS-DMA result must be covered by an s_waitcnt before use;
                LDS-DMA data additionally by a workgroup barrier for the other waves).  The rules are CONSERVATIVE
                versions of cdna_hip_programming.md §5.7 item 2 / MI355X_MICROARCH.md; `Checker` raises on a violation.

Register ids are one integer space: v0..v255 = 0..255, a0..a255 = 256..511, s0..s127 = 512..639, then VCC, SCC, M0.
"""
import numpy as np

V0, A0, S0 = 0, 256, 512
VCC, SCC, M0, EXEC = 640, 641, 642, 643


def v(i):
    assert 0 <= i < 256, i
    return V0 + i


def a(i):
    assert 0 <= i < 256, i
    return A0 + i


def s(i):
    assert 0 <= i < 104, i
    return S0 + i


def rname(r):
    if r < 256:
        return "v%d" % r
    if r < 512:
        return "a%d" % (r - 256)
    if r < 640:
        return "s%d" % (r - 512)
    return {VCC: "vcc", SCC: "scc", M0: "m0", EXEC: "exec"}[r]


def rrange(r, n):
    """assembler text of n consecutive registers starting at id r"""
    if n == 1:
        return rname(r)
    if r < 256:
        return "v[%d:%d]" % (r, r + n - 1)
    if r < 512:
        return "a[%d:%d]" % (r - 256, r - 256 + n - 1)
    assert r < 640
    return "s[%d:%d]" % (r - 512, r - 512 + n - 1)


class I:
    """one instruction.  rd / wr: register ids read / written (for the hazard and wait checks); fn(wave): emulation."""
    __slots__ = ("text", "kind", "rd", "wr", "fn", "cost", "srcc", "target", "note", "region", "tag")

    def __init__(self, text, kind, rd=(), wr=(), fn=None, cost=4, srcc=(), target=None, note=None):
        self.text, self.kind, self.rd, self.wr, self.fn = text, kind, tuple(rd), tuple(wr), fn
        self.cost, self.srcc, self.target, self.note = cost, tuple(srcc), target, note
        self.tag = None
        self.region = None         # "begin" / "end": a forward branch and its label around a conditionally executed block (cond_block)

    def __repr__(self):
        return self.text


# ---------------------------------------------------------------------------------------------- helpers
def _f32(x):
    return x.view(np.float32)


def _u32(x):
    return x.view(np.uint32)


def bf16_round(x):
    """f32 array -> bf16 bits (uint32 in the low 16), round to nearest even, NaN kept"""
    u = _u32(np.ascontiguousarray(x, dtype=np.float32)).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF
    nan = np.isnan(x)
    r = np.where(nan, 0x7FC0, r)
    return r.astype(np.uint32)


def bf16_to_f32(b):
    return _f32((b.astype(np.uint32) << 16).astype(np.uint32))


_LANE = np.arange(64)
_L31 = _LANE & 31
_LH = _LANE >> 5


def _frag_to_mat_A(regs4):
    """regs4: [4, 64] uint32 fragment -> A[32 rows][16 k] f32 (lane l: row l&31, k = 8*(l>>5)+j)"""
    m = np.zeros((32, 16), np.float32)
    for j in range(8):
        w = regs4[j >> 1]
        e = (w >> (16 * (j & 1))) & 0xFFFF
        m[_L31, 8 * _LH + j] = bf16_to_f32(e)
    return m


def _acc_rows():
    rows = np.zeros((16, 64), np.int64)
    for r in range(16):
        rows[r] = (r & 3) + 8 * (r >> 2) + 4 * _LH
    return rows


_ACC_ROWS = _acc_rows()


# ---------------------------------------------------------------------------------------------- builders
def _src_text(x):
    """operand: register id, python float/int literal"""
    if isinstance(x, (int, np.integer)) and not isinstance(x, bool) and x >= 0 and x < 700:
        return rname(int(x))
    raise TypeError(x)


class Lit:
    """literal operand (32-bit pattern given as python int, or float)"""
    def __init__(self, val):
        if isinstance(val, float):
            self.bits = int(np.array([val], np.float32).view(np.uint32)[0])
            self.text = {0.0: "0", 1.0: "1.0", -1.0: "-1.0", 0.5: "0.5", 2.0: "2.0"}.get(val, "0x%08x" % self.bits)
        else:
            self.bits = int(val) & 0xFFFFFFFF
            self.text = str(int(val)) if -16 <= int(val) <= 64 else "0x%08x" % self.bits


def _opd(w, x):
    """value of a VALU source operand as a [64] uint32 array"""
    if isinstance(x, Lit):
        return np.full(64, x.bits, np.uint32)
    if x >= S0:
        return np.full(64, w.sget(x), np.uint32)
    return w.R[x]


def _otext(x):
    return x.text if isinstance(x, Lit) else rname(x)


def _ords(*xs):
    return tuple(x for x in xs if not isinstance(x, Lit))


def valu2(op, d, x, y, fnp, kind="valu", cost=4):
    def fn(w):
        w.R[d] = _u32(fnp(_opd(w, x), _opd(w, y)))
    return I("%s %s, %s, %s" % (op, rname(d), _otext(x), _otext(y)), kind, _ords(x, y), (d,), fn, cost)


def v_add_f32(d, x, y):
    return valu2("v_add_f32", d, x, y, lambda p, q: (_f32(p) + _f32(q)).astype(np.float32))


def v_sub_f32(d, x, y):
    return valu2("v_sub_f32", d, x, y, lambda p, q: (_f32(p) - _f32(q)).astype(np.float32))


def v_mul_f32(d, x, y):
    return valu2("v_mul_f32", d, x, y, lambda p, q: (_f32(p) * _f32(q)).astype(np.float32))


def v_max_f32(d, x, y):
    return valu2("v_max_f32", d, x, y, lambda p, q: np.fmax(_f32(p), _f32(q)).astype(np.float32))


def v_add_u32(d, x, y):
    return valu2("v_add_u32", d, x, y, lambda p, q: (p.astype(np.uint64) + q.astype(np.uint64)).astype(np.uint32))


def v_lshlrev_b32(d, sh, x):
    return valu2("v_lshlrev_b32", d, sh, x, lambda p, q: (q.astype(np.uint64) << (p & 31).astype(np.uint64)).astype(np.uint32))


def v_and_b32(d, x, y):
    return valu2("v_and_b32", d, x, y, lambda p, q: p & q)


def v_pk_mul_f32(d, x, y):
    """packed f32: (d, d+1) = (x, x+1) * (y, y+1) -- two products per lane at the issue cost of one (round 6: halves the multiply /
    subtract instructions of the backward loops' vector streams).  d, x, y: EVEN register ids (64-bit pairs)."""
    assert d % 2 == 0 and x % 2 == 0 and y % 2 == 0, (d, x, y)

    def fn(w):
        lo = (_f32(w.R[x]) * _f32(w.R[y])).astype(np.float32)
        hi = (_f32(w.R[x + 1]) * _f32(w.R[y + 1])).astype(np.float32)
        w.R[d], w.R[d + 1] = _u32(lo), _u32(hi)
    return I("v_pk_mul_f32 %s, %s, %s" % (rrange(d, 2), rrange(x, 2), rrange(y, 2)), "valu", (x, x + 1, y, y + 1), (d, d + 1), fn, 4)


def v_pk_sub_f32_bcast(d, x, y):
    """packed f32: (d, d+1) = (x, x+1) - y broadcast: y is ONE register (either half of an even-aligned pair), subtracted from both
    halves (v_pk_add_f32 with neg_lo / neg_hi on the second source and op_sel / op_sel_hi picking y's half for both results)"""
    assert d % 2 == 0 and x % 2 == 0, (d, x)
    base, half = y & ~1, y & 1

    def fn(w):
        lo = (_f32(w.R[x]) - _f32(w.R[y])).astype(np.float32)
        hi = (_f32(w.R[x + 1]) - _f32(w.R[y])).astype(np.float32)
        w.R[d], w.R[d + 1] = _u32(lo), _u32(hi)
    mods = "op_sel:[0,1] op_sel_hi:[1,1]" if half else "op_sel_hi:[1,0]"
    return I("v_pk_add_f32 %s, %s, %s %s neg_lo:[0,1] neg_hi:[0,1]" % (rrange(d, 2), rrange(x, 2), rrange(base, 2), mods), "valu",
             (x, x + 1, y), (d, d + 1), fn, 4)


def v_max3_f32(d, x, y, z):
    def fn(w):
        w.R[d] = _u32(np.fmax(np.fmax(_f32(_opd(w, x)), _f32(_opd(w, y))), _f32(_opd(w, z))).astype(np.float32))
    return I("v_max3_f32 %s, %s, %s, %s" % (rname(d), _otext(x), _otext(y), _otext(z)), "valu", _ords(x, y, z), (d,), fn, 4)


def v_mov_b32(d, x):
    def fn(w):
        w.R[d] = _opd(w, x).copy()
    return I("v_mov_b32 %s, %s" % (rname(d), _otext(x)), "valu", _ords(x), (d,), fn, 4)


def v_exp_f32(d, x):
    def fn(w):
        with np.errstate(over="ignore", under="ignore", invalid="ignore"):
            w.R[d] = _u32(np.exp2(_f32(_opd(w, x)).astype(np.float64)).astype(np.float32))
    return I("v_exp_f32 %s, %s" % (rname(d), _otext(x)), "trans", _ords(x), (d,), fn, 8)


def v_cvt_pk_bf16_f32(d, x, y):
    def fn(w):
        lo, hi = bf16_round(_f32(_opd(w, x))), bf16_round(_f32(_opd(w, y)))
        w.R[d] = (lo | (hi << 16)).astype(np.uint32)
    return I("v_cvt_pk_bf16_f32 %s, %s, %s" % (rname(d), _otext(x), _otext(y)), "valu", _ords(x, y), (d,), fn, 4)


def v_cndmask_b32(d, x, y, m):
    """d = mask bit ? y : x   (mask: SGPR pair id, or VCC)"""
    def fn(w):
        mask = w.vcc if m == VCC else w.s64(m)
        bits = ((mask >> _LANE.astype(np.uint64)) & 1).astype(bool) if isinstance(mask, np.ndarray) else \
            np.array([(mask >> int(l)) & 1 for l in range(64)], bool)
        w.R[d] = np.where(bits, _opd(w, y), _opd(w, x)).astype(np.uint32)
    mt = "vcc" if m == VCC else rrange(m, 2)
    rd = _ords(x, y) + ((VCC,) if m == VCC else (m, m + 1))
    return I("v_cndmask_b32 %s, %s, %s, %s" % (rname(d), _otext(x), _otext(y), mt), "valu", rd, (d,), fn, 4)


def v_cmp_f32(cmp, dst, x, y):
    """dst (VCC or SGPR pair id) = per-lane x <cmp> y; cmp in lt, gt, le, ge"""
    op = {"lt": np.less, "gt": np.greater, "le": np.less_equal, "ge": np.greater_equal}[cmp]

    def fn(w):
        b = op(_f32(_opd(w, x)), _f32(_opd(w, y)))
        mask = 0
        for l in range(64):
            if b[l]:
                mask |= 1 << l
        if dst == VCC:
            w.vcc = mask
        else:
            w.sset64(dst, mask)
    dt = "vcc" if dst == VCC else rrange(dst, 2)
    wr = (VCC,) if dst == VCC else (dst, dst + 1)
    return I("v_cmp_%s_f32 %s, %s, %s" % (cmp, dt, _otext(x), _otext(y)), "valu", _ords(x, y), wr, fn, 4)


def v_cmp_i32(cmp, dst, x, y):
    op = {"lt": np.less, "gt": np.greater, "le": np.less_equal, "ge": np.greater_equal, "eq": np.equal}[cmp]

    def fn(w):
        b = op(_opd(w, x).view(np.int32), _opd(w, y).view(np.int32))
        mask = 0
        for l in range(64):
            if b[l]:
                mask |= 1 << l
        if dst == VCC:
            w.vcc = mask
        else:
            w.sset64(dst, mask)
    dt = "vcc" if dst == VCC else rrange(dst, 2)
    wr = (VCC,) if dst == VCC else (dst, dst + 1)
    return I("v_cmp_%s_i32 %s, %s, %s" % (cmp, dt, _otext(x), _otext(y)), "valu", _ords(x, y), wr, fn, 4)


def v_permlane32_swap(d, x):
    """lanes 32-63 of d swap with lanes 0-31 of x"""
    def fn(w):
        dd, xx = w.R[d].copy(), w.R[x].copy()
        w.R[d][32:] = xx[:32]
        w.R[x][:32] = dd[32:]
    return I("v_permlane32_swap_b32 %s, %s" % (rname(d), rname(x)), "permlane", (d, x), (d, x), fn, 4)


def v_accvgpr_read(d, src):
    def fn(w):
        w.R[d] = w.R[src].copy()
    return I("v_accvgpr_read_b32 %s, %s" % (rname(d), rname(src)), "valu", (src,), (d,), fn, 4)


def v_accvgpr_write(d, x):
    def fn(w):
        w.R[d] = _opd(w, x).copy()
    return I("v_accvgpr_write_b32 %s, %s" % (rname(d), _otext(x)), "valu", _ords(x), (d,), fn, 4)


def v_mfma_32x32x16_bf16(d, fa, fb, c):
    """D[32x32] = A[32x16] B[16x32] + C.  d, c: base id of 16 registers (c None -> 0); fa, fb: base id of 4 registers"""
    def fn(w):
        Am = _frag_to_mat_A(w.R[fa:fa + 4])
        Bm = _frag_to_mat_A(w.R[fb:fb + 4]).T            # B[k][col]: lane l holds col l&31, k = 8*(l>>5)+j: same packing
        P = (Am.astype(np.float64) @ Bm.astype(np.float64))
        with np.errstate(invalid="ignore", over="ignore"):
            for r in range(16):
                cin = _f32(w.R[c + r]).astype(np.float64) if c is not None else 0.0
                w.R[d + r] = _u32((P[_ACC_ROWS[r], _L31] + cin).astype(np.float32))
    ct = rrange(c, 16) if c is not None else "0"
    rd = tuple(range(fa, fa + 4)) + tuple(range(fb, fb + 4)) + (tuple(range(c, c + 16)) if c is not None else ())
    text = "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (rrange(d, 16), rrange(fa, 4), rrange(fb, 4), ct)
    import os
    if os.environ.get("UR_ASMGEN_MFMA16") == "1":
        # lab, timing only (results WRONG): the same matrix-pipe cycles as two v_mfma_f32_16x16x32_bf16 -- does the chip hold a higher clock on
        # that shape under these loops' load (MI355X_MICROARCH.md 'DVFS give-back' item 7: 1.12-1.15 x in bare loops)?
        text = "\n".join("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (rrange(d + 4 * h, 4), rrange(fa, 4), rrange(fb, 4), rrange(c + 4 * h, 4) if c is not None else "0")
                         for h in range(2))
    return I(text, "mfma", rd, tuple(range(d, d + 16)), fn, 8, srcc=(tuple(range(c, c + 16)) if c is not None else ()))


def ds_read_b128(d, vaddr, off):
    assert 0 <= off < 65536 and off % 16 == 0, off

    def fn(w):
        ad = w.R[vaddr].astype(np.int64) + off
        w.lds_read_check(ad, 16)
        for l in range(64):
            w.R[d:d + 4, l] = w.lds[ad[l]:ad[l] + 16].view(np.uint32)
    return I("ds_read_b128 %s, %s offset:%d" % (rrange(d, 4), rname(vaddr), off), "lds", (vaddr,), tuple(range(d, d + 4)), fn, 4)


def ds_read_b64_tr_b16(d, vaddr, off):
    assert 0 <= off < 65536 and off % 8 == 0, off

    def fn(w):
        ad = w.R[vaddr].astype(np.int64) + off
        w.lds_read_check(ad, 8)
        out = np.zeros((64, 4), np.uint16)
        for l in range(64):
            g, i = l & ~15, l & 15
            for q in range(4):
                src = ad[g + 4 * q + (i >> 2)] + 2 * (i & 3)
                out[l, q] = w.lds[src:src + 2].view(np.uint16)[0]
        w.R[d] = out[:, 0].astype(np.uint32) | (out[:, 1].astype(np.uint32) << 16)
        w.R[d + 1] = out[:, 2].astype(np.uint32) | (out[:, 3].astype(np.uint32) << 16)
    return I("ds_read_b64_tr_b16 %s, %s offset:%d" % (rrange(d, 2), rname(vaddr), off), "lds", (vaddr,), (d, d + 1), fn, 4)


def global_load_lds_dwordx4(voff, sbase):
    """LDS[M0 + 16*lane .. +16] = mem[s[sbase:sbase+1] + zext(v[voff])]   (one 1-KiB piece per wave instruction)"""
    def fn(w):
        base = w.s64(sbase)
        m0 = w.m0
        w.dma_issue(m0, 1024)
        for l in range(64):
            src = base + int(w.R[voff, l])
            w.lds[m0 + 16 * l:m0 + 16 * l + 16] = w.gmem[src:src + 16]
    return I("global_load_lds_dwordx4 %s, %s" % (rname(voff), rrange(sbase, 2)), "vmem", (voff, sbase, sbase + 1, M0), (), fn, 16)


# ---- scalar
def _sval(w, x):
    if isinstance(x, Lit):
        return x.bits
    return w.sget(x)


def salu2(op, d, x, y, fnp, setscc=None):
    def fn(w):
        r = fnp(_sval(w, x), _sval(w, y), w)
        if setscc is not None:
            w.scc = setscc(_sval(w, x), _sval(w, y), r)
        w.sset(d, r & 0xFFFFFFFF)
    wr = (d,) + ((SCC,) if setscc is not None else ())
    return I("%s %s, %s, %s" % (op, rname(d), _otext(x), _otext(y)), "salu", _ords(x, y), wr, fn, 1)


def _s32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def s_add_u32(d, x, y):
    return salu2("s_add_u32", d, x, y, lambda p, q, w: p + q, lambda p, q, r: int(r > 0xFFFFFFFF))


def s_addc_u32(d, x, y):
    def fn(w):
        r = _sval(w, x) + _sval(w, y) + w.scc
        w.scc = int(r > 0xFFFFFFFF)
        w.sset(d, r & 0xFFFFFFFF)
    return I("s_addc_u32 %s, %s, %s" % (rname(d), _otext(x), _otext(y)), "salu", _ords(x, y) + (SCC,), (d, SCC), fn, 1)


def s_add_i32(d, x, y):
    return salu2("s_add_i32", d, x, y, lambda p, q, w: p + q, lambda p, q, r: 0)


def s_sub_i32(d, x, y):
    return salu2("s_sub_i32", d, x, y, lambda p, q, w: p - q, lambda p, q, r: 0)


def s_mul_i32(d, x, y):
    return salu2("s_mul_i32", d, x, y, lambda p, q, w: _s32(p) * _s32(q))


def s_lshl_b32(d, x, y):
    return salu2("s_lshl_b32", d, x, y, lambda p, q, w: p << (q & 31), lambda p, q, r: int((r & 0xFFFFFFFF) != 0))


def s_min_i32(d, x, y):
    return salu2("s_min_i32", d, x, y, lambda p, q, w: min(_s32(p), _s32(q)), lambda p, q, r: int(_s32(p) <= _s32(q)))


def s_and_b32(d, x, y):
    return salu2("s_and_b32", d, x, y, lambda p, q, w: p & q, lambda p, q, r: int((r & 0xFFFFFFFF) != 0))


def s_mov_b32(d, x):
    def fn(w):
        val = _sval(w, x)
        if d == M0:
            w.m0 = val
        else:
            w.sset(d, val)
    return I("s_mov_b32 %s, %s" % (rname(d), _otext(x)), "salu", _ords(x), (d,), fn, 1)


def s_mov_b64(d, x):
    def fn(w):
        val = x.bits if isinstance(x, Lit) else w.s64(x)
        if isinstance(x, Lit) and x.bits & 0x80000000:
            val |= 0xFFFFFFFF00000000           # 32-bit literals sign-extend
        w.sset64(d, val)
    rd = () if isinstance(x, Lit) else (x, x + 1)
    return I("s_mov_b64 %s, %s" % (rrange(d, 2), x.text if isinstance(x, Lit) else rrange(x, 2)), "salu", rd, (d, d + 1), fn, 1)


def s_or_b64(d, x, y):
    def fn(w):
        r = (w.vcc if x == VCC else w.s64(x)) | (w.vcc if y == VCC else w.s64(y))
        w.scc = int(r != 0)
        if d == VCC:
            w.vcc = r
        else:
            w.sset64(d, r)
    t = lambda z: "vcc" if z == VCC else rrange(z, 2)
    rd = tuple(r for z in (x, y) for r in ((VCC,) if z == VCC else (z, z + 1)))
    wr = ((VCC,) if d == VCC else (d, d + 1)) + (SCC,)
    return I("s_or_b64 %s, %s, %s" % (t(d), t(x), t(y)), "salu", rd, wr, fn, 1)


def s_cmp(cmp, x, y, unsigned=False):
    """SCC = x <cmp> y (32-bit)"""
    ops = {"lt": lambda p, q: p < q, "le": lambda p, q: p <= q, "gt": lambda p, q: p > q, "ge": lambda p, q: p >= q,
           "eq": lambda p, q: p == q, "lg": lambda p, q: p != q}

    def fn(w):
        p, q = _sval(w, x), _sval(w, y)
        if not unsigned:
            p, q = _s32(p), _s32(q)
        w.scc = int(ops[cmp](p, q))
    return I("s_cmp_%s_%s %s, %s" % (cmp, "u32" if unsigned else "i32", _otext(x), _otext(y)), "salu", _ords(x, y), (SCC,), fn, 1)


def s_cmp_lg_u64(x, lit0=True):
    def fn(w):
        w.scc = int((w.vcc if x == VCC else w.s64(x)) != 0)
    t = "vcc" if x == VCC else rrange(x, 2)
    return I("s_cmp_lg_u64 %s, 0" % t, "salu", ((VCC,) if x == VCC else (x, x + 1)), (SCC,), fn, 1)


def s_bitcmp1_b64(x, bit):
    """SCC = bit `bit` (SGPR id or Lit) of the 64-bit value s[x:x+1]"""
    def fn(w):
        w.scc = int((w.s64(x) >> (_sval(w, bit) & 63)) & 1)
    return I("s_bitcmp1_b64 %s, %s" % (rrange(x, 2), _otext(bit)), "salu", (x, x + 1) + _ords(bit), (SCC,), fn, 1)


def s_cselect_b32(d, x, y):
    def fn(w):
        w.sset(d, _sval(w, x) if w.scc else _sval(w, y))
    return I("s_cselect_b32 %s, %s, %s" % (rname(d), _otext(x), _otext(y)), "salu", _ords(x, y) + (SCC,), (d,), fn, 1)


def s_cselect_b64(d, x, y):
    def fn(w):
        gv = lambda z: (z.bits | (0xFFFFFFFF00000000 if z.bits & 0x80000000 else 0)) if isinstance(z, Lit) else w.s64(z)
        w.sset64(d, gv(x) if w.scc else gv(y))
    t = lambda z: z.text if isinstance(z, Lit) else rrange(z, 2)
    rd = tuple(r for z in (x, y) if not isinstance(z, Lit) for r in (z, z + 1)) + (SCC,)
    return I("s_cselect_b64 %s, %s, %s" % (rrange(d, 2), t(x), t(y)), "salu", rd, (d, d + 1), fn, 1)


def s_branch(label):
    return I("s_branch %s" % label, "branch", (), (), lambda w: True, 1, target=label)


def s_cbranch_scc(val, label):
    return I("s_cbranch_scc%d %s" % (val, label), "branch", (SCC,), (), lambda w: w.scc == val, 1, target=label)


def label(name):
    return I("%s:" % name, "label", target=name, cost=0)


def s_waitcnt(vmcnt=None, lgkmcnt=None):
    parts = []
    if vmcnt is not None:
        assert 0 <= vmcnt < 64
        parts.append("vmcnt(%d)" % vmcnt)
    if lgkmcnt is not None:
        assert 0 <= lgkmcnt < 16
        parts.append("lgkmcnt(%d)" % lgkmcnt)

    def fn(w):
        w.wait(vmcnt, lgkmcnt)
    return I("s_waitcnt " + " ".join(parts), "wait", (), (), fn, 1, note=(vmcnt, lgkmcnt))


def s_memtime_wait(d):
    """lab: s_memtime into s[d:d+1] and the wait that makes it usable (drains the LDS reads in flight as well)"""
    def fn(w):
        w.sset64(d, w.step * 4)
        w.wait(None, 0)
    return I("s_memtime %s\ns_waitcnt lgkmcnt(0)" % rrange(d, 2), "wait", (), (d, d + 1), fn, 8, note=(None, 0))


def s_sub_u32(d, x, y):
    return salu2("s_sub_u32", d, x, y, lambda p, q, w: p - q, lambda p, q, r: int(p < q))


def global_store_dword_s(vdata, voff, sbase, imm):
    """mem[s[sbase:sbase+1] + v[voff] + imm] = v[vdata]  (emulator: recorded in wave.stores)"""
    def fn(w):
        w.stores = getattr(w, "stores", [])
        w.stores.append((w.s64(sbase) + int(w.R[voff, 0]) + imm, int(w.R[vdata, 0])))
    return I("global_store_dword %s, %s, %s offset:%d" % (rname(voff), rname(vdata), rrange(sbase, 2), imm), "vmem_st", (vdata, voff, sbase, sbase + 1), (), fn, 4)


def s_barrier():
    return I("s_barrier", "barrier", (), (), None, 1)


def s_nop(n):
    assert 0 <= n <= 15
    return I("s_nop %d" % n, "nop", (), (), lambda w: None, 4 * (n + 1), note=n)


def comment(text):
    return I("; " + text, "label", cost=0)


def cond_block(branch, body, lbl):
    """`branch` jumps over `body` to `lbl`.  Contract (what lets fix_hazards treat the join like the not-taken path): the body
    drains every LDS read it issues (ends its LDS use with s_waitcnt lgkmcnt(0)), issues no MFMA, and ends with s_nop 3, so no
    vector write of the body is closer than 4 wait states to the code behind the label."""
    assert branch.kind == "branch" and lbl.kind == "label" and branch.target == lbl.target
    branch.region, lbl.region = "begin", "end"
    assert all(i.kind != "mfma" for i in body)
    return [branch] + list(body) + [s_nop(3), lbl]


# ---------------------------------------------------------------------------------------------- emulator
class HazardError(Exception):
    pass


MFMA_RESULT_WAIT = 14      # wait states between an MFMA and a non-MFMA read/write of its result (8-pass XDL: 12 are required)
VALU_TO_MFMA_WAIT = 3      # vector write -> MFMA operand read (2 required)
TRANS_TO_VALU_WAIT = 1     # v_exp_f32 result -> non-transcendental vector read
PERMLANE_WAIT = 3          # vector write -> v_permlane32_swap operand (2 required), and back
M0_TO_DMA_WAIT = 1
SRCC_WAR_WAIT = 20         # MFMA srcC read -> vector write of those registers


class Wave:
    def __init__(self, wg, wid):
        self.wg, self.wid = wg, wid
        self.R = np.zeros((512, 64), np.uint32)
        # poison: uninitialised registers hold a NaN pattern so that a read-before-write shows in the result
        self.R[:] = 0x7FC0DEAD
        self.S = [0xDEADBEEF] * 128
        self.vcc, self.scc, self.m0 = 0, 0, 0
        self.lds, self.gmem = wg.lds, wg.gmem
        self.pc, self.done, self.at_barrier = 0, False, False
        self.step = 0                      # wait-state clock
        self.lastw = {}                    # reg -> (step, kind) of the last write
        self.lastc = {}                    # reg -> step of the last MFMA srcC read
        self.lgkm = []                     # outstanding LDS reads, oldest first: tuple of destination registers
        self.vm = []                       # outstanding LDS-DMA pieces: (lds_addr, size)
        self.pend = {}                     # register -> outstanding LDS read count
        self.nbar = 0
        self.m0_step = -100
        self.trace_n = {}

    # scalar access
    def sget(self, r):
        if r == M0:
            return self.m0
        return self.S[r - S0]

    def sset(self, r, val):
        self.S[r - S0] = val & 0xFFFFFFFF

    def s64(self, r):
        return self.S[r - S0] | (self.S[r - S0 + 1] << 32)

    def sset64(self, r, val):
        self.S[r - S0] = val & 0xFFFFFFFF
        self.S[r - S0 + 1] = (val >> 32) & 0xFFFFFFFF

    # counters
    def wait(self, vmcnt, lgkmcnt):
        if lgkmcnt is not None:
            while len(self.lgkm) > lgkmcnt:
                for r in self.lgkm.pop(0):
                    self.pend[r] -= 1
                    if self.pend[r] == 0:
                        del self.pend[r]
        if vmcnt is not None:
            while len(self.vm) > vmcnt:
                ad, n = self.vm.pop(0)
                self.wg.dma_landed(self, ad, n)

    def dma_issue(self, ad, n):
        self.wg.dma_issue(self, ad, n)
        self.vm.append((ad, n))
        if len(self.vm) > 63:
            raise HazardError("vmcnt overflow")

    def lds_read_check(self, ad, n):
        self.wg.lds_read(self, ad, n)


class Workgroup:
    """NW waves sharing one LDS; waves run round-robin from barrier to barrier"""
    BLK = 1024

    def __init__(self, nwaves, lds_bytes, gmem):
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.lds[:] = 0xFF                          # poison (bf16 0xFFFF = NaN)
        self.gmem = gmem
        nb = (lds_bytes + self.BLK - 1) // self.BLK
        # LDS-DMA block state: None = plain memory; ('fly', wave) in flight; ('landed', barrier index at which it becomes visible)
        self.blk = [None] * nb
        self.lastread = [[-1] * nwaves for _ in range(nb)]     # per block and wave: barrier epoch of the last read
        self.waves = [Wave(self, i) for i in range(nwaves)]

    def dma_issue(self, w, ad, n):
        assert ad % self.BLK == 0 and n == self.BLK, (ad, n)
        b = ad // self.BLK
        if self.blk[b] is not None and self.blk[b][0] == "fly":
            raise HazardError("LDS-DMA into block %d that is still in flight" % b)
        for ww in self.waves:
            if self.lastread[b][ww.wid] >= w.nbar:
                raise HazardError("wave %d issues LDS-DMA into block %d (epoch %d) that wave %d read in epoch %d: no barrier between"
                                  % (w.wid, b, w.nbar, ww.wid, self.lastread[b][ww.wid]))
        self.blk[b] = ("fly", w.wid)

    def dma_issue_small(self, w, ad, n):
        """a piece smaller than a block (row constants: several waves fill parts of one block)"""
        b = ad // self.BLK
        for ww in self.waves:
            if self.lastread[b][ww.wid] >= w.nbar:
                raise HazardError("wave %d issues LDS-DMA into block %d (epoch %d) that wave %d read in epoch %d: no barrier between"
                                  % (w.wid, b, w.nbar, ww.wid, self.lastread[b][ww.wid]))
        st = self.blk[b]
        self.blk[b] = ("fly", w.wid, (st[2] if st is not None and st[0] == "fly" and len(st) > 2 else 0) + 1)
        w.vm.append((ad, n))

    def dma_landed(self, w, ad, n):
        b = ad // self.BLK
        st = self.blk[b]
        if n < self.BLK and st is not None and st[0] == "fly" and len(st) > 2 and st[2] > 1:
            self.blk[b] = ("fly", st[1], st[2] - 1)       # other pieces of the block are still in flight
            return
        self.blk[b] = ("landed", w.nbar, w.wid)    # visible to other waves after the barrier the issuer reaches next

    def lds_read(self, w, ad, n):
        for b in set(int(x) // self.BLK for x in ad) | set(int(x + n - 1) // self.BLK for x in ad):
            st = self.blk[b]
            if st is not None:
                if st[0] == "fly":
                    raise HazardError("wave %d pc %d reads LDS block %d while its LDS-DMA (wave %d) is in flight" % (w.wid, w.pc, b, st[1]))
                if w.nbar <= st[1]:
                    raise HazardError("wave %d pc %d reads LDS block %d landed in epoch %d by wave %d without a barrier (reader epoch %d)"
                                      % (w.wid, w.pc, b, st[1], st[2], w.nbar))
            self.lastread[b][w.wid] = w.nbar


class Program:
    def __init__(self):
        self.ins = []

    def add(self, *items):
        for it in items:
            if isinstance(it, (list, tuple)):
                self.add(*it)
            elif it is not None:
                self.ins.append(it)
        return self

    def finalize(self):
        self.labels = {}
        for k, it in enumerate(self.ins):
            if it.kind == "label" and it.target is not None:
                assert it.target not in self.labels, "duplicate label " + it.target
                self.labels[it.target] = k
        for it in self.ins:
            if it.kind == "branch":
                assert it.target in self.labels, "unknown label " + it.target
        return self

    def text(self, suffix="%="):
        """assembler text; labels get `suffix` appended (inline asm: %= makes them unique per statement)"""
        out = []
        for it in self.ins:
            t = it.text
            if it.kind == "label" and it.target is not None:
                t = "%s%s:" % (it.target, suffix)
            elif it.kind == "branch":
                t = t.replace(it.target, it.target + suffix)
            out.extend(x.strip() for x in t.split("\n"))
        return out

    def stats(self):
        c = {}
        for it in self.ins:
            c[it.kind] = c.get(it.kind, 0) + 1
        return c


def missing_wait_states(w, it):
    """(n, why): wait states still missing before `it` may issue on tracker `w` (0 = none).  Shared by the dynamic checker
    (raises) and the generator's static fixer (pads with s_nop)."""
    k = it.kind
    need, why = 0, ""

    def want(n, msg):
        nonlocal need, why
        if n > need:
            need, why = n, msg
    srcc = set(it.srcc)
    for r in it.rd:
        lw = w.lastw.get(r)
        if lw is None:
            continue
        st, wk, wregs = lw
        dist = w.step - st - 1            # wait states strictly between the two instructions
        if wk == "mfma":
            if k == "mfma" and r in srcc and wregs == it.srcc and set(it.wr) == srcc:
                continue                  # accumulate chain: vdst == srcC of the same registers
            want(MFMA_RESULT_WAIT - dist, "%s read after the MFMA that wrote it" % rname(r))
            continue
        if k == "mfma" and wk in ("valu", "trans", "permlane"):
            want(VALU_TO_MFMA_WAIT - dist, "MFMA operand %s after a vector write" % rname(r))
        if wk == "trans" and k in ("valu", "permlane", "mfma"):
            want(TRANS_TO_VALU_WAIT - dist, "%s read after v_exp" % rname(r))
        if k == "permlane" and wk in ("valu", "trans"):
            want(PERMLANE_WAIT - dist, "v_permlane operand %s after a vector write" % rname(r))
        if wk == "permlane" and k in ("valu", "trans", "mfma"):
            want(PERMLANE_WAIT - 1 - dist, "%s read after v_permlane32_swap" % rname(r))
    for r in it.wr:
        lw = w.lastw.get(r)
        if lw is not None and lw[1] == "mfma" and k != "mfma":
            want(MFMA_RESULT_WAIT - (w.step - lw[0] - 1), "%s overwritten after the MFMA that wrote it" % rname(r))
        lc = w.lastc.get(r)
        if lc is not None and k != "mfma":
            want(SRCC_WAR_WAIT - (w.step - lc - 1), "%s overwritten after an MFMA read it as C" % rname(r))
    if k == "vmem" and w.step - w.m0_step - 1 < M0_TO_DMA_WAIT:
        want(M0_TO_DMA_WAIT - (w.step - w.m0_step - 1), "LDS-DMA right after the M0 write")
    return need, why


def track(w, it):
    """bookkeeping after `it` issued"""
    k = it.kind
    for r in it.wr:
        w.lastw[r] = (w.step, k, it.wr if k == "mfma" else None)
    if k == "mfma":
        for r in it.srcc:
            if r not in it.wr:
                w.lastc[r] = w.step
    if M0 in it.wr:
        w.m0_step = w.step
    w.step += (it.note + 1) if k == "nop" else 1


def check_and_step(w, it):
    """dynamic hazard / wait checks for instruction `it` about to issue on wave w, then bookkeeping"""
    k = it.kind
    if k == "label":
        return
    if w.pend:
        bad = [r for r in set(it.rd) | set(it.wr) if r in w.pend]
        if bad:
            raise HazardError("wave %d pc %d `%s`: %s still has an LDS read outstanding" % (w.wid, w.pc, it.text, rname(bad[0])))
    n, why = missing_wait_states(w, it)
    if n > 0:
        raise HazardError("wave %d pc %d `%s`: %d wait state(s) missing: %s" % (w.wid, w.pc, it.text, n, why))
    if k == "lds":
        w.lgkm.append(tuple(it.wr))
        for r in it.wr:
            w.pend[r] = w.pend.get(r, 0) + 1
        if len(w.lgkm) > 16:
            raise HazardError("wave %d pc %d: more than 16 LDS operations outstanding" % (w.wid, w.pc))
    track(w, it)


class Tracker:
    """register-write history without the machine state: what the static fixer walks a straight-line body with"""
    def __init__(self):
        self.step, self.lastw, self.lastc, self.m0_step = 1000, {}, {}, -100


LANDED_AFTER = 16          # wait states after which an LDS read is taken to have landed: a counted wait placed for one fragment also covers such older-than-that successors (fewer s_waitcnt in the stream)


def fix_hazards(seq, entry_lgkm=()):
    """straight-line instruction list -> the same list with s_nop / s_waitcnt lgkmcnt(n) inserted where the rules ask for them.
    entry_lgkm: destination-register tuples of the LDS reads that may still be outstanding on entry (oldest first).
    Conditionally executed blocks (cond_block) are fixed as if taken, then the tracker returns to its state at the branch."""
    import copy
    t = Tracker()
    q = [(tuple(x), t.step - 1000) for x in entry_lgkm]
    out = []
    saved = None

    import os
    _abl = set(os.environ.get("UR_ASMGEN_ABLATE", "").split(","))       # lab, timing only (results WRONG): "lgkm" drops the counted LDS waits, "nop" the s_nops

    def wait_keep(keep):
        wi = s_waitcnt(lgkmcnt=min(keep, 15))
        if "lgkm" not in _abl:
            out.append(wi)
        track(t, wi)
        del q[:len(q) - min(keep, 15)]

    for it in seq:
        k = it.kind
        if it.region == "begin":
            out.append(it)
            track(t, it)
            saved = (copy.deepcopy(t), list(q))
            continue
        if it.region == "end":
            assert saved is not None
            t, q = saved
            saved = None
            out.append(it)
            continue
        if k == "label":
            out.append(it)
            continue
        if k == "wait" and it.note[1] is not None:
            del q[:max(0, len(q) - it.note[1])]
        regs = set(it.rd) | set(it.wr)
        idx = [i for i, (dst, _) in enumerate(q) if regs & set(dst)]
        if idx:
            last = max(idx)
            while last + 1 < len(q) and t.step - q[last + 1][1] >= LANDED_AFTER:
                last += 1
            wait_keep(len(q) - 1 - last)
        n, _ = missing_wait_states(t, it)
        while n > 0:
            m = min(n, 8)
            ni = s_nop(m - 1)
            if "nop" not in _abl:
                out.append(ni)
            track(t, ni)
            n -= m
        if k == "lds":
            if len(q) >= 15:
                wait_keep(14)            # the 4-bit counter cannot express more: retire the oldest first
            q.append((tuple(it.wr), t.step))
        out.append(it)
        track(t, it)
    return out, [d for d, _ in q]


def run_workgroup(prog, wg, init, max_steps=5_000_000, check=True):
    """run all waves of `wg` through `prog` (finalized).  init(wave) sets the entry state.  Returns per-wave instruction counts."""
    for w in wg.waves:
        init(w)
    ins, labels = prog.ins, prog.labels
    nsteps = 0
    while True:
        progressed = False
        for w in wg.waves:
            if w.done or w.at_barrier:
                continue
            while True:
                if w.pc >= len(ins):
                    w.done = True
                    break
                it = ins[w.pc]
                if check:
                    check_and_step(w, it)
                w.trace_n[it.kind] = w.trace_n.get(it.kind, 0) + 1
                if it.kind == "label" and it.target is not None:
                    w.trace_n["@" + it.target] = w.trace_n.get("@" + it.target, 0) + 1
                nsteps += 1
                if nsteps > max_steps:
                    raise RuntimeError("emulation does not terminate")
                if it.kind == "barrier":
                    w.at_barrier = True
                    w.pc += 1
                    progressed = True
                    break
                if it.kind == "branch":
                    if it.fn(w):
                        w.pc = labels[it.target]
                    else:
                        w.pc += 1
                    continue
                if it.fn is not None:
                    it.fn(w)
                w.pc += 1
                progressed = True
        live = [w for w in wg.waves if not w.done]
        if not live:
            break
        if all(w.at_barrier for w in live):
            if len(live) != len(wg.waves):
                raise HazardError("barrier reached by %d of %d waves (the others ended): deadlock" % (len(live), len(wg.waves)))
            for w in live:
                w.at_barrier = False
                w.nbar += 1
            continue
        if not progressed:
            raise HazardError("deadlock: some waves ended while others wait at a barrier")
    return [dict(w.trace_n) for w in wg.waves]

"""Dependency-aware re-placement of the non-MFMA instructions of a generated loop body (round 6).

The generators (attn_fwd.py, attn_dq.py, attn_dkv.py) place every LDS read, LDS-DMA piece and vector instruction in a FIXED gap of the
MFMA stream (`put(gap, ...)`): positions derived from the data flow by hand.  Their own static issue model -- a gap costs
max(32, 8 + the issue cycles of what rides in it) -- shows the price of that: the dK/dV body sums to 1890 issue cycles over 64 gaps
(floor 2048) yet models at 2304 because some gaps carry 40-88 cycles while others carry 8-16.  `balance()` is a greedy list re-scheduler
over the straight-line body: it takes instructions out of gaps above the budget and moves them into the nearest gap with room, earlier or
later, when
  * no instruction in between reads or writes what the moved one writes, or writes what it reads (RAW / WAR / WAW over the register ids
    of isa.I.rd / .wr, which carry VCC, SCC and M0), and nothing crosses a branch, label, barrier, counted wait or conditional block;
    LDS-DMA pieces keep their order (the counted `vmcnt` waits at the loop top count them);
  * an LDS read keeps at least `min_lead` MFMA slots ahead of its first consumer (its latency), and is never moved later when its
    consumer lies in the next iteration;
  * the result, after isa.fix_hazards has re-inserted every counted wait and s_nop the rules ask for, models strictly cheaper.
MFMAs never move.  The emulator suites (tests/test_asmgen_*.py) run every balanced body against the fp64 reference with the hazard
checker on, exactly as they did the hand-placed ones.
"""
import isa

MOVABLE = ("lds", "valu", "trans", "salu", "vmem")
FENCE = ("branch", "label", "barrier", "wait", "vmem_st")


def icost(i):
    if i.kind == "nop":
        return (i.note + 1) * 4
    return max(4, i.cost)


def gap_costs(fixed):
    """issue cycles per MFMA gap of a fixed (hazard-complete) body; conditionally executed blocks are not counted (as the generators'
    own model); what precedes the first MFMA is returned separately"""
    costs, cur, pre, skip = [], None, 0, False
    for i in fixed:
        if i.region == "begin":
            skip = True
            continue
        if i.region == "end":
            skip = False
            continue
        if skip or i.kind == "label":
            continue
        if i.kind == "mfma":
            if cur is not None:
                costs.append(cur)
            cur = 8
        elif cur is not None:
            cur += icost(i)
        else:
            pre += icost(i)
    if cur is not None:
        costs.append(cur)
    return costs, pre


def model(fixed, budget=32):
    costs, pre = gap_costs(fixed)
    return sum(max(budget, c) for c in costs) + pre


def _conflict(x, y):
    xr, xw = set(x.rd), set(x.wr)
    yr, yw = set(y.rd), set(y.wr)
    return bool((yw & xr) or (yr & xw) or (yw & xw))


def _is_fence(i):
    return i.kind in FENCE or i.region is not None


def _units(seq, temps):
    """index -> (start, end) of the unit an instruction moves with: an instruction and the one that consumes the scratch register it
    writes (M0 before an LDS-DMA piece, an address temporary before its read)"""
    unit = {}
    k = 0
    n = len(seq)
    while k < n:
        e = k
        while e + 1 < n and not _is_fence(seq[e]) and not _is_fence(seq[e + 1]) and seq[e].kind != "mfma" and seq[e + 1].kind != "mfma" and \
                (set(seq[e].wr) & set(seq[e + 1].rd) & temps):
            e += 1
        for j in range(k, e + 1):
            unit[j] = (k, e)
        k = e + 1
    return unit


def balance(seq, entry_lgkm=(), budget=32, min_lead=5, reach=12, temps=(), max_moves=400, verbose=False, frozen_tags=("max",), name=""):
    """seq: straight-line body (MFMAs + side instructions, before fix_hazards).  Returns the re-placed list (same instructions)."""
    import os
    # lab: bisecting a schedule on the hardware (name = generator, or generator_BODY: the longest match wins)
    for key in ("UR_BALANCE_MAX_MOVES_" + name.upper(), "UR_BALANCE_MAX_MOVES_" + name.upper().split("_")[0], "UR_BALANCE_MAX_MOVES"):
        if key in os.environ:
            max_moves = int(os.environ[key])
            break
    fill_m0 = os.environ.get("UR_BALANCE_FILL_M0", "1") == "1"
    budget = int(os.environ.get("UR_BALANCE_BUDGET", budget))              # lab
    min_lead = int(os.environ.get("UR_BALANCE_MIN_LEAD", min_lead))        # lab
    reach = int(os.environ.get("UR_BALANCE_REACH", reach))                 # lab
    temps = set(temps) | {isa.M0}
    seq = list(seq)

    def evaluate(s):
        fixed, _ = isa.fix_hazards(s, entry_lgkm=entry_lgkm)
        return model(fixed, budget), fixed

    best, fixed = evaluate(seq)
    start = best
    moves = 0
    while moves < max_moves:
        # gap index of every instruction of seq (gap g = behind MFMA g; -1 = ahead of the first MFMA)
        gap_of, g = [], -1
        mf_pos = []
        for k, i in enumerate(seq):
            if i.kind == "mfma":
                g += 1
                mf_pos.append(k)
            gap_of.append(g)
        ng = g + 1
        # cost per gap from the FIXED body (waits and nops included), mapped back by walking both lists
        costs, _ = gap_costs(fixed)
        if len(costs) != ng:
            break
        unit = _units(seq, temps)
        order = sorted((c, gi) for gi, c in enumerate(costs) if c > budget)
        order.reverse()
        if mf_pos and mf_pos[0] > 0:
            order.append((0, -1))               # what sits ahead of the first MFMA delays the whole body: try to hide it in the gaps
        done = False
        for c, gi in order:
            lo = mf_pos[gi] + 1 if gi >= 0 else 0
            hi = mf_pos[gi + 1] if gi + 1 < ng else len(seq)
            cands = [k for k in range(lo, hi) if seq[k].kind in MOVABLE and not _is_fence(seq[k]) and seq[k].tag not in frozen_tags and unit[k][0] == k]
            for k in cands:
                us, ue = unit[k]
                if ue >= hi:
                    continue
                ucost = sum(icost(seq[j]) for j in range(us, ue + 1))
                # nearest gaps first; a target may end above the budget as long as the evaluated model gets cheaper (it takes less
                # over the budget than the source gives up)
                src_over = (c - budget) if gi >= 0 else ucost
                targets = []
                for d in range(1, reach + 1):
                    for t in (gi - d, gi + d):
                        if 0 <= t < ng and max(0, costs[t] + ucost - budget) - max(0, costs[t] - budget) < min(ucost, src_over):
                            targets.append(t)
                for t in targets:
                    cand = _try_move(seq, mf_pos, us, ue, gi, t, ng, min_lead)
                    if cand is None:
                        continue
                    val, fx = evaluate(cand)
                    if val < best:
                        if verbose:
                            print("  move %-40s gap %2d -> %2d   model %d -> %d" % (seq[us].text[:40], gi, t, best, val))
                        seq, best, fixed = cand, val, fx
                        moves += 1
                        done = True
                        break
                if done:
                    break
            if done:
                break
        if not done:
            break
    # an LDS-DMA piece needs one wait state behind its M0 write (fix_hazards pads with s_nop 0): put an independent neighbour between
    k = 0
    while k + 1 < len(seq):
        x, y = seq[k], seq[k + 1]
        if fill_m0 and isa.M0 in x.wr and y.kind == "vmem" and isa.M0 in y.rd:
            nxt = seq[k + 2] if k + 2 < len(seq) else None
            prv = seq[k - 1] if k > 0 else None
            ok = lambda z, *others: z is not None and z.kind in ("lds", "valu", "trans", "salu") and not _is_fence(z) and z.tag not in frozen_tags and \
                isa.M0 not in z.wr and not any(_conflict(z, o) for o in others)
            cand = None
            if ok(nxt, y):
                cand = seq[:k + 1] + [nxt, y] + seq[k + 3:]
            elif ok(prv, x):
                cand = seq[:k - 1] + [x, prv, y] + seq[k + 2:]
            if cand is not None:
                val, fx = evaluate(cand)
                if val <= best:
                    seq, best, fixed = cand, val, fx
        k += 1
    if verbose:
        print("balance: model %d -> %d in %d moves" % (start, best, moves))
    return seq


def _try_move(seq, mf_pos, us, ue, gi, t, ng, min_lead):
    """the list with unit [us, ue] moved from gap gi to gap t, or None when a dependency, a fence or a latency rule forbids it"""
    grp = seq[us:ue + 1]
    if t < gi:
        dst = mf_pos[t + 1]                     # end of gap t: right ahead of MFMA t + 1
        between = seq[dst:us]
    else:
        dst = mf_pos[t] + 1                     # start of gap t: right behind MFMA t
        between = seq[ue + 1:dst]
    for y in between:
        if _is_fence(y):
            return None
        for x in grp:
            if _conflict(x, y) or (x.kind == "vmem" and y.kind == "vmem"):
                return None
    if t > gi:
        # an LDS read moved later keeps min_lead MFMA slots ahead of its first consumer; without a consumer in this body it stays
        for x in grp:
            if x.kind != "lds":
                continue
            cons = None
            slot = gi
            for y in seq[ue + 1:]:
                if y.kind == "mfma":
                    slot += 1
                if set(y.rd) & set(x.wr):
                    cons = slot
                    break
            if cons is None or cons - t < min_lead:
                return None
    if t < gi:
        return seq[:dst] + grp + seq[dst:us] + seq[ue + 1:]
    return seq[:us] + seq[ue + 1:dst] + grp + seq[dst:]

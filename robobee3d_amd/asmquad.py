"""ONE ROBOT PER LANE QUAD: ADMM iterations 2..maxIter of the all-assembly fp32 step stream (asmstep.py) for the
latency-bound shapes -- the B = 1 drop-in and every batch that cannot give each SIMD a wave of its own (B <= 16 384).

Why. One lane per robot executes ~800 instructions per iteration however few robots there are: a
lone wave issues one instruction every ~5 cycles, so a single umpcUpdate is 0.12 ms of a serial chain with 63 lanes of
its wave and 1 023 SIMDs idle. Here the wavefront carries 16 robots, four lanes each. The phases around the loop run
REDUNDANTLY in the four lanes of a quad (same instruction stream, the same robot's rows: v0 = 4 * (lane >> 2 + 16 * wave)),
so at the loop every lane already holds the whole problem and nothing has to be communicated to split it; the loop then
keeps a third of the unknowns in each of lanes 0..2 (lane 3 idles on zeros):

  * the x / y / z members of every position / orientation triple of the QP (the KKT pattern is invariant under their
    cyclic relabelling, asmgen.slot_maps) sit in lanes 0 / 1 / 2 of ONE register; thrust, first and second moment of the
    three horizon steps likewise (lane = step), the three thrust rows in one register. 84 unknowns = 29 registers.
  * a triangular-solve operation W[d] += (-L_j) W[s] runs in the lane that owns d; its source comes through the DPP
    quad_perm operand of v_fmac_f32 (any lane of the quad, no extra instruction). Operations with the same destination
    register and the same source register share ONE instruction -- the coefficient register holds a different entry of
    L per lane, ZERO where a lane has nothing to do. EXEC stays full inside the solves: on gfx9 a DPP source lane that is
    masked off is an invalid lane (its reader's write is dropped), so masking is used only by the plain moves of the
    entry transposition. 94 + 100 instructions for the 213 + 213 entries (one lane per robot: 150 + 150 packed ones plus
    ~270 operand fetches).
  * every element-wise phase (right-hand side, 1/D, x / y updates) is 15 + 12 registers wide instead of 45 + 39 + 39,
    packed two registers per instruction; q, l, 1/D and 116 coefficient words are VGPR-resident (a lane owns a third of
    the 561-word working set), the remaining 78 coefficient words come four per ds_read_b128 from the lane's LDS words.

318 instructions per iteration instead of 804. Entry (after the first iteration, which runs in the one-lane form on
the one-lane homes phase A filled): each lane copies ITS third of x, y, z, q, l, 1/D and L into the quad registers with
plain moves under a lane-class mask (~1 000 instructions, two iterations' worth). Exit: every word of x, y, z, delta_x,
delta_y is broadcast to the four lanes' one-lane homes with one v_mov_b32_dpp each (171), and phase C goes on as ever.

The accumulation order inside one unknown follows THIS schedule (rounding only, as asmgen's list schedule already
differs from QDLDL's column sweep); every other operation is the one-lane iteration's, operand for operand.
`simulate()` interprets the section on the four lanes of a quad (DPP selects, EXEC masks, AGPRs, LDS) for the CPU tests.

Reference mapping: template/uprightmpc2/osqp.c:354-370, auxil.c:164-228, qdldl.c:250-293 (through asmgen.body).
"""
import numpy as np

from . import symbolic
from .asmgen import (A_D, A_L, A_LO, A_M, A_Q, NLDS, S_ALPHA, S_CNT, S_ITERS, S_OMA, S_RINV, S_SIGMA, V_W, V_WZ, V_X,
                     V_Y, V_Z, _sb, _vp, pk)

# ---- quad register map (disjoint from the one-lane homes of the loop's outputs: v2..v173 and the thrust-row z, v210..v212)
QDI = 2       # 29: 1/D by unknown (same index as QW)
QQ = 32       # 16: q by x index
QL = 48       # 12: l (= u) of the dynamics rows
QC = 60       # coefficient words resident in VGPRs: v60 .. v175
NCV = 116
QX = 176      # 16: x
QYT, QZT, QLO3, QUP3, QRHO3, QRINV3 = 192, 193, 194, 195, 196, 197    # thrust rows: lanes 0..2 = rows 36..38
QT = 198      # 12 temporaries, v198 .. v209
QW = 214      # 29: W (x part 0..14 + pad, dynamics rows 16..27, thrust rows 28)
QY = 244      # 12: y of the dynamics rows
IX_EQ, IX_T = 16, 28
NQW = 29
S_L0, S_L1, S_L2, S_EXEC = 30, 32, 34, 36      # lane-class masks (lanes 0 / 1 / 2 of every quad) and the entry EXEC
A_OVF = A_D                                     # AGPR homes of the coefficient words beyond NCV (1/D and q are consumed first)
DPP_TAIL = " row_mask:0xf bank_mask:0xf"


def qperm(p):
    return "quad_perm:[%d,%d,%d,%d]" % tuple(p) + DPP_TAIL


class QuadPlan:
    def __init__(self, st):
        """st: asmstep.Struct (symbolic structure, one-lane slot maps and L storage positions)"""
        self.st, s = st, st.s
        self.s = s
        nx, nk, neq = s.nx, s.nk, st.neq
        assert s.N == 3 and symbolic.NY == 6
        ny2 = 2 * s.N * symbolic.NY

        def home(k):
            o = s.perm[k]
            if o < nx:
                if o < ny2:
                    return o % 3, o // 3
                u = o - ny2
                return u // 3, 12 + u % 3
            i = o - nx
            if i < neq:
                return i % 3, IX_EQ + i // 3
            return i - neq, IX_T
        self.home = [home(k) for k in range(nk)]
        assert len(set(self.home)) == nk
        self.xhome = [self.home[s.pinv[j]] for j in range(nx)]
        self.zhome = [self.home[s.pinv[nx + i]] for i in range(s.nc)]
        ents = [(s.L_i[j], c, j) for c in range(nk) for j in range(s.L_p[c], s.L_p[c + 1])]
        self.fwd = self.schedule([(r, c, j) for (r, c, j) in ents])
        self.bwd = self.schedule([(c, r, j) for (r, c, j) in ents])
        self.ncoef = len(self.fwd) + len(self.bwd)
        # coefficient word of instruction q (forward instructions first): VGPR-resident, or an AGPR spread evenly over the
        # sequence so that the reads never bunch up
        novf = max(0, self.ncoef - NCV)
        self.cloc, nv, na = [], 0, 0
        for q in range(self.ncoef):
            if ((q + 1) * novf) // self.ncoef != (q * novf) // self.ncoef:
                self.cloc.append(("a", A_OVF + na))
                na += 1
            else:
                self.cloc.append(("v", QC + nv))
                nv += 1
        assert nv <= NCV and na == novf and A_OVF + na <= A_Q, (nv, na)

    # ---- static matching + list schedule ---------------------------------------------------------------------------
    def schedule(self, ops):
        """ops (d, s, j): W[d] += (-L_j) W[s], legal once W[s] is final. Ops with the same destination register and the
        same source register, one per lane, are matched into one instruction up front; a matched group that would have
        to wait for itself through other ops sheds one member. Then a list schedule: among the ready instructions one
        that does not touch what the last two wrote (DPP source hazard, dependent-issue stall), then the longest chain."""
        home = self.home
        byreg = {}
        for o in ops:
            (ld, rd), (_, rs) = home[o[0]], home[o[1]]
            byreg.setdefault((rd, rs), {}).setdefault(ld, []).append(o)
        groups = []
        for lanes in byreg.values():
            while any(lanes.values()):
                groups.append({ln: lst.pop(0) for ln, lst in lanes.items() if lst})
        # every lane of an instruction reads its source BEFORE any lane writes: an op whose source unknown is the
        # destination of another op of its own group (x and y member of one triple, same register) cannot ride along
        gi = 0
        while gi < len(groups):
            g = groups[gi]
            dsts = {o[0] for o in g.values()}
            bad = [ln for ln, o in g.items() if o[1] in dsts]
            if len(g) > 1 and bad:
                groups.append({bad[0]: g.pop(bad[0])})
                continue
            gi += 1
        while True:
            writers = {}
            for gi, g in enumerate(groups):
                for o in g.values():
                    writers.setdefault(o[0], set()).add(gi)
            opdeps = lambda o: writers.get(o[1], set())
            deps = [set().union(*[opdeps(o) for o in g.values()]) - {gi} for gi, g in enumerate(groups)]
            done, order, progress = set(), [], True
            while progress:
                progress = False
                for gi in range(len(groups)):
                    if gi not in done and deps[gi] <= done:
                        done.add(gi)
                        order.append(gi)
                        progress = True
            if len(order) == len(groups):
                break
            best = None
            for gi in range(len(groups)):
                if gi in done or len(groups[gi]) < 2:
                    continue
                for ln, o in groups[gi].items():
                    sc = len(opdeps(o) - {gi} - done)
                    if best is None or sc < best[0]:
                        best = (sc, gi, ln)
            assert best is not None, "dependency cycle among single operations"
            _, gi, ln = best
            groups.append({ln: groups[gi].pop(ln)})
        succ = [[] for _ in groups]
        for gi, dd in enumerate(deps):
            for x in dd:
                succ[x].append(gi)
        height = [0] * len(groups)
        for gi in reversed(order):
            height[gi] = 1 + max([height[x] for x in succ[gi]] or [0])
        regs = lambda gi: (home[next(iter(groups[gi].values()))[0]][1], home[next(iter(groups[gi].values()))[1]][1])
        npend = [len(dd) for dd in deps]
        ready = {gi for gi in range(len(groups)) if npend[gi] == 0}
        out, last = [], [None, None]
        while ready:
            def score(gi):
                rd, rs = regs(gi)
                return (2 * (last[0] in (rd, rs)) + (last[1] in (rd, rs)), -height[gi], gi)
            gi = min(ready, key=score)
            ready.discard(gi)
            rd, rs = regs(gi)
            perm = [0, 1, 2, 3]
            for ln, o in groups[gi].items():
                perm[ln] = home[o[1]][0]
            out.append(dict(d=rd, s=rs, ops=dict(groups[gi]), perm=perm))
            last = [rd, last[0]]
            for x in succ[gi]:
                npend[x] -= 1
                if npend[x] == 0:
                    ready.add(x)
        assert len(out) == len(groups)
        return out


# ---------------------------------------------------------------------------------------------------------------------
# the iteration body
# ---------------------------------------------------------------------------------------------------------------------
def body(e, plan, capture):
    """one quad iteration (z == l on the dynamics rows: every iteration but the first). capture: leave delta_x in the x
    part of QW and delta_y in its row part, as asmgen.body(delta_in_w=True) does for phase C."""
    v = lambda n: "v%d" % n
    sA, sO = "s%d" % S_ALPHA, "s%d" % S_OMA
    # ---- rhs  W = [sigma x - q ; l - y / rho]  (auxil.c:164-178)
    for p in range(0, 16, 2):
        pk(e, "v_pk_fma_f32", QW + p, [_sb(S_SIGMA), _vp(QX + p), _vp(QQ + p)], [0, 0, 1])
    for p in range(0, 12, 2):
        pk(e, "v_pk_fma_f32", QW + IX_EQ + p, [_sb(S_RINV), _vp(QY + p), _vp(QL + p)], [1, 0, 0])
    e("v_fma_f32", v(QW + IX_T), "-" + v(QRINV3), v(QYT), v(QZT))
    e("s_nop", 1)
    # ---- solves: operand fetches (AGPR-homed coefficient words) run AHEAD of their consumers in a ring of four temporaries
    seq = [("op", q, ins) for q, ins in enumerate(plan.fwd)]
    seq.append(("diag",))
    seq += [("op", len(plan.fwd) + q, ins) for q, ins in enumerate(plan.bwd)]
    # coefficient words beyond the VGPR-resident ones: four per ds_read_b128 from this lane's LDS words 0.. (written there
    # by the entry), three quads in flight in the twelve temporaries (idle during the solves)
    ovf = [q for q in range(plan.ncoef) if plan.cloc[q][0] == "a"]
    oidx = {q: n for n, q in enumerate(ovf)}
    first_use = {}                  # quad -> sequence position of its first consumer
    for k, item in enumerate(seq):
        if item[0] == "op" and item[1] in oidx:
            first_use.setdefault(oidx[item[1]] // 4, k)
    nquads = (len(ovf) + 3) // 4
    AHQ = 12                        # sequence positions between a quad's read and its first consumer
    issued, waited = [0], [0]

    def issue_upto(k):
        while issued[0] < nquads and first_use[issued[0]] <= k + AHQ and issued[0] < waited[0] + 3:
            gq = issued[0]
            e("ds_read_b128", "v[%d:%d]" % (QT + 4 * (gq % 3), QT + 4 * (gq % 3) + 3), "v1", gq * 1024)
            issued[0] += 1
    lastw = [set(), set()]
    issue_upto(0)
    for k, item in enumerate(seq):
        issue_upto(k)
        if item[0] == "diag":   # qdldl.c:289
            for p in range(0, 28, 2):
                pk(e, "v_pk_mul_f32", QW + p, [_vp(QW + p), _vp(QDI + p)])
            e("v_mul_f32", v(QW + IX_T), v(QW + IX_T), v(QDI + IX_T))
            e("s_nop", 1)
            lastw = [set(), set()]
            continue
        _, q, ins = item
        if q in oidx:
            gq = oidx[q] // 4
            if gq >= waited[0]:         # first consumer of this quad: reads return in order
                assert gq < issued[0]
                e("s_waitcnt", "lgkmcnt(%d)" % (issued[0] - 1 - gq))
                waited[0] = gq + 1
                issue_upto(k)
            c = QT + 4 * (gq % 3) + oidx[q] % 4
        else:
            c = plan.cloc[q][1]
        d, sr = QW + ins["d"], QW + ins["s"]
        if ins["perm"] == [0, 1, 2, 3]:
            e("v_fmac_f32", v(d), v(sr), v(c))
        else:
            if sr in lastw[0]:
                e("s_nop", 1)
            elif sr in lastw[1]:
                e("s_nop", 0)
            e("v_fmac_f32_dpp", v(d), v(sr), v(c), qperm(ins["perm"]))
        lastw = [{d}, lastw[0]]
    # ---- x <- alpha x~ + (1 - alpha) x   (auxil.c:188-201), software-pipelined over the temporaries
    first, second = [], []
    for p in range(0, 16, 2):
        t = QT + 4 + 2 * ((p // 2) % 4)
        first.append(lambda p=p, t=t: pk(e, "v_pk_mul_f32", t, [_sb(S_OMA), _vp(QX + p)]))
        if capture:
            def upd(p=p, t=t):
                pk(e, "v_pk_fma_f32", t, [_sb(S_ALPHA), _vp(QW + p), _vp(t)])
                pk(e, "v_pk_add_f32", QW + p, [_vp(t), _vp(QX + p)], [0, 1])
                e("v_pk_mov_b32", "v[%d:%d]" % (QX + p, QX + p + 1), "v[%d:%d]" % (t, t + 1), "v[%d:%d]" % (t, t + 1),
                  dict(op_sel=[0, 1], op_sel_hi=[0, 0], neg_lo=[0, 0], neg_hi=[0, 0]))
            second.append(upd)
        else:
            second.append(lambda p=p, t=t: pk(e, "v_pk_fma_f32", QX + p, [_sb(S_ALPHA), _vp(QW + p), _vp(t)]))
    # ---- dynamics rows: delta_y = alpha (nu - y)   (asmgen.body, z == l)
    for p in range(0, 12, 2):
        t = QT + 4 + 2 * (((16 + p) // 2) % 4)
        first.append(lambda p=p, t=t: pk(e, "v_pk_add_f32", t, [_vp(QW + IX_EQ + p), _vp(QY + p)], [0, 1]))
        if capture:
            def updy(p=p, t=t):
                pk(e, "v_pk_mul_f32", QW + IX_EQ + p, [_sb(S_ALPHA), _vp(t)])
                pk(e, "v_pk_fma_f32", QY + p, [_sb(S_ALPHA), _vp(t), _vp(QY + p)])
            second.append(updy)
        else:
            second.append(lambda p=p, t=t: pk(e, "v_pk_fma_f32", QY + p, [_sb(S_ALPHA), _vp(t), _vp(QY + p)]))
    lag = 3
    for k in range(len(first) + lag):
        if k < len(first):
            first[k]()
        if k >= lag:
            second[k - lag]()
    # ---- thrust rows (auxil.c:203-228, proj.c:4-14): the one-lane chain on one register, lanes 0..2
    t1, t2, t3 = v(QT), v(QT + 1), v(QT + 2)
    nu, y, z = v(QW + IX_T), v(QYT), v(QZT)
    e("v_fma_f32", t1, "-" + v(QRINV3), y, z)
    e("v_mul_f32", t2, sO, z)
    e("v_fma_f32", t1, v(QRINV3), nu, t1)
    e("v_fma_f32", t1, sA, t1, t2)
    e("v_fma_f32", t3, v(QRINV3), y, t1)
    e("v_max_f32", t3, t3, v(QLO3))
    e("v_min_f32", z, t3, v(QUP3))
    e("v_sub_f32", t2, t1, z)
    e("v_mul_f32", t2, v(QRHO3), t2)
    e("v_add_f32", y, y, t2)
    if capture:
        e("v_mov_b32", nu, t2)


# ---------------------------------------------------------------------------------------------------------------------
# entry / exit transposition
# ---------------------------------------------------------------------------------------------------------------------
def entry(e, plan):
    """one-lane homes (every lane of the quad holds all of them) -> quad registers. Plain moves under lane-class masks."""
    st, s = plan.st, plan.s
    nx, nc, nk, neq = s.nx, s.nc, s.nk, st.neq
    v = lambda n: "v%d" % n
    masks = (S_L0, S_L1, S_L2)
    sp_ = lambda n: "s[%d:%d]" % (n, n + 1)
    e("s_mov_b64", sp_(S_EXEC), "exec")
    for ln, m in enumerate(masks):
        e("s_mov_b32", "s%d" % m, 0x11111111 << ln)
        e("s_mov_b32", "s%d" % (m + 1), 0x11111111 << ln)
        e("s_and_b64", sp_(m), sp_(m), sp_(S_EXEC))
    # ---- 1. state and per-row / per-column data (targets: v176..v197, v244..v255, v2..v59 -- no source lives there)
    zero = [QX + k for k in range(16)] + [QY + k for k in range(12)] + [QYT, QZT, QLO3, QUP3, QRHO3, QRINV3] + \
           [QDI + k for k in range(NQW)] + [QQ + k for k in range(16)] + [QL + k for k in range(12)]
    for r in zero:
        e("v_mov_b32", v(r), 0)
    moves = {0: [], 1: [], 2: []}      # lane -> [(kind, dst, src)]
    for j in range(nx):
        ln, ix = plan.xhome[j]
        moves[ln].append(("v", QX + ix, V_X + st.xs[j]))
        moves[ln].append(("a", QQ + ix, A_Q + j))
    for i in range(nc):
        ln, ix = plan.zhome[i]
        if i < neq:
            moves[ln].append(("v", QY + ix - IX_EQ, V_Y + st.zs[i]))
            moves[ln].append(("a", QL + ix - IX_EQ, A_LO + i))
        else:
            k = i - neq
            assert ln == k and ix == IX_T
            moves[ln].append(("v", QYT, V_Y + st.zs[i]))
            moves[ln].append(("v", QZT, V_Z + st.zs[i]))
            for dst, a0 in ((QLO3, A_M), (QUP3, A_M + 3), (QRHO3, A_M + 6), (QRINV3, A_M + 9)):
                moves[ln].append(("a", dst, a0 + k))
    for k in range(nk):
        ln, ix = plan.home[k]
        moves[ln].append(("a", QDI + ix, A_D + k))
    for ln in range(3):
        e("s_mov_b64", "exec", sp_(masks[ln]))
        for kind, dst, src in moves[ln]:
            if kind == "v":
                e("v_mov_b32", v(dst), v(src))
            else:
                e("v_accvgpr_read_b32", v(dst), "a%d" % src)
    e("s_mov_b64", "exec", sp_(S_EXEC))
    # ---- 2. coefficient words: zero everywhere, then every entry of L to its forward word (lane of its row unknown) and its
    # backward word (lane of its column unknown)
    for kind, r in plan.cloc:
        if kind == "v":
            e("v_mov_b32", v(r), 0)
    e("v_mov_b32", v(QT), 0)
    for kind, r in plan.cloc:
        if kind == "a":
            e("v_accvgpr_write_b32", "a%d" % r, v(QT))
    dests = {}      # entry j -> [(lane, word location)]
    for q, ins in enumerate(plan.fwd + plan.bwd):
        for ln, o in ins["ops"].items():
            dests.setdefault(o[2], []).append((ln, plan.cloc[q]))
    assert all(len(dests[j]) == 2 for j in range(len(s.L_i)))
    lpos = st.lpos
    by_src = {"a": [], "l": {}}
    for j in range(len(s.L_i)):
        if lpos[j] < NLDS:
            by_src["l"].setdefault(lpos[j] // 4, []).append(j)
        else:
            by_src["a"].append(j)
    # 2a. entries parked in AGPRs (a0 .. a52): through a temporary (full EXEC), then the masked writes
    tmp = [QT + 1 + k for k in range(8)]
    for base in range(0, len(by_src["a"]), 8):
        chunk = by_src["a"][base:base + 8]
        e("s_mov_b64", "exec", sp_(S_EXEC))
        for k, j in enumerate(chunk):
            e("v_accvgpr_read_b32", v(tmp[k]), "a%d" % (A_L + lpos[j] - NLDS))
        for ln in range(3):
            todo = [(k, loc) for k, j in enumerate(chunk) for (l2, loc) in dests[j] if l2 == ln]
            if not todo:
                continue
            e("s_mov_b64", "exec", sp_(masks[ln]))
            for k, loc in todo:
                if loc[0] == "v":
                    e("v_mov_b32", v(loc[1]), v(tmp[k]))
                else:
                    e("v_accvgpr_write_b32", "a%d" % loc[1], v(tmp[k]))
    # 2b. entries in LDS: seven float4 per round through the (idle) W registers
    quads = sorted(by_src["l"])
    for base in range(0, len(quads), 7):
        chunk = quads[base:base + 7]
        e("s_mov_b64", "exec", sp_(S_EXEC))
        for k, qd in enumerate(chunk):
            e("ds_read_b128", "v[%d:%d]" % (QW + 4 * k, QW + 4 * k + 3), "v1", qd * 1024)
        e("s_waitcnt", "lgkmcnt(0)")
        for ln in range(3):
            todo = []
            for k, qd in enumerate(chunk):
                for j in by_src["l"][qd]:
                    for (l2, loc) in dests[j]:
                        if l2 == ln:
                            todo.append((QW + 4 * k + lpos[j] % 4, loc))
            if not todo:
                continue
            e("s_mov_b64", "exec", sp_(masks[ln]))
            for src, loc in todo:
                if loc[0] == "v":
                    e("v_mov_b32", v(loc[1]), v(src))
                else:
                    e("v_accvgpr_write_b32", "a%d" % loc[1], v(src))
    e("s_mov_b64", "exec", sp_(S_EXEC))
    # the composed AGPR words move to this lane's LDS words 0.. (the factor there has been consumed): four per float4,
    # read back with one ds_read_b128 per four coefficients in every iteration instead of one v_accvgpr_read each
    ovf = [r for kind, r in plan.cloc if kind == "a"]
    for base in range(0, len(ovf), 4):
        t = QT + 4 * ((base // 4) % 3)
        for k in range(4):
            if base + k < len(ovf):
                e("v_accvgpr_read_b32", v(t + k), "a%d" % ovf[base + k])
        e("ds_write_b128", "v1", "v[%d:%d]" % (t, t + 3), (base // 4) * 1024)
    e("s_waitcnt", "lgkmcnt(0)")
    e("s_nop", 4)


def exit_(e, plan):
    """quad registers -> the one-lane homes phase C reads, in all four lanes of the quad (one DPP broadcast per word)"""
    st, s = plan.st, plan.s
    nx, nc, neq = s.nx, s.nc, st.neq
    v = lambda n: "v%d" % n
    e("s_nop", 1)

    def bcast(dst, src, ln):
        e("v_mov_b32_dpp", v(dst), v(src), qperm([ln] * 4))
    for j in range(nx):
        ln, ix = plan.xhome[j]
        bcast(V_X + st.xs[j], QX + ix, ln)
        bcast(V_W + st.xs[j], QW + ix, ln)
    for i in range(nc):
        ln, ix = plan.zhome[i]
        if i < neq:
            bcast(V_Y + st.zs[i], QY + ix - IX_EQ, ln)
            bcast(V_WZ + st.zs[i], QW + ix, ln)
        else:
            bcast(V_Y + st.zs[i], QYT, ln)
            bcast(V_WZ + st.zs[i], QW + IX_T, ln)
            bcast(V_Z + st.zs[i], QZT, ln)
    for pad in (V_X + nx, V_Y + nc, V_Z + nc, V_W + nx, V_WZ + nc):
        e("v_mov_b32", v(pad), 0)
    e("s_nop", 1)


def section(e, plan, label):
    """everything between the first (one-lane) iteration and phase C; label() hands out fresh numeric labels"""
    sg = lambda n: "s%d" % n
    lab7, lab8 = label(), label()
    e("quad_begin",)
    entry(e, plan)
    e("s_sub_i32", sg(S_CNT), sg(S_ITERS), 2)
    e("s_cmp_lt_i32", sg(S_CNT), 1)
    e("s_cbranch_scc1", lab8 + "f")
    e("label", lab7)
    body(e, plan, capture=False)
    e("s_sub_i32", sg(S_CNT), sg(S_CNT), 1)
    e("s_cmp_gt_i32", sg(S_CNT), 0)
    e("s_cbranch_scc1", lab7 + "b")
    e("label", lab8)
    body(e, plan, capture=True)
    exit_(e, plan)
    e("quad_end",)


# ---------------------------------------------------------------------------------------------------------------------
# CPU interpreter of the section on the four lanes of ONE quad
# ---------------------------------------------------------------------------------------------------------------------
def simulate(ins, pc, V, A, lds, S, max_exec=400000):
    """ins[pc] is ("quad_begin",). V, A: uint32 [4][256]; lds: uint32 [4][NLDS] (each lane's own slice); S: SGPR dict of
    the calling interpreter (scalar constants, S_ITERS). Runs to ("quad_end",) and returns (pc behind it, executed count).
    EXEC is modelled per lane of the quad; a DPP read of a lane that is masked off raises (gfx9: the write would be
    dropped -- the generator must never rely on it)."""
    f32, u32 = np.float32, np.uint32
    asf = lambda b: np.array(b, u32).view(f32)
    bits = lambda x: np.array(x, f32).view(u32)
    exec_ = np.ones(4, bool)
    entry_exec = exec_.copy()
    masks = {}
    scc = 0
    labels = {}
    for k, t in enumerate(ins):
        if t[0] == "label":
            labels.setdefault(t[1], []).append(k)
    vi = lambda x: int(x[1:])

    def sval(x):
        if isinstance(x, float):
            return f32(x)
        if isinstance(x, int):
            assert x == 0
            return f32(0)
        return asf(u32(S.get(int(x[1:]), 0) & 0xFFFFFFFF))

    def fsrc(x):
        """float32 [4] of a VALU source operand"""
        if isinstance(x, (int, float)):
            return np.full(4, sval(x), f32)
        neg = x.startswith("-")
        if neg:
            x = x[1:]
        ab = x.startswith("|")
        if ab:
            x = x[1:-1]
        val = asf(V[:, vi(x)]) if x[0] == "v" else np.full(4, sval(x), f32)
        if ab:
            val = np.abs(val)
        return -val if neg else val

    def setv(x, val):
        r = vi(x)
        V[exec_, r] = bits(np.asarray(val, f32))[exec_]

    def half(x, sel):
        lo = int(x[2:x.index(":")])
        if x[0] == "v":
            return asf(V[:, lo + sel]).astype(np.float64)
        return np.full(4, np.float64(asf(u32(S.get(lo + sel, 0)))))

    def dpp_src(reg, mod):
        """reg: a register number, or an operand string with modifiers ("|v12|")"""
        neg = ab = False
        if isinstance(reg, str):
            neg = reg.startswith("-")
            reg = reg[1:] if neg else reg
            ab = reg.startswith("|")
            reg = vi(reg[1:-1] if ab else reg)
        qp = [int(c) for c in mod[mod.index("[") + 1:mod.index("]")].split(",")]
        for ln in range(4):
            if exec_[ln] and not exec_[qp[ln]]:
                raise AssertionError("DPP read of a masked-off lane: %r" % (mod,))
        val = asf(V[qp, reg])
        if ab:
            val = np.abs(val)
        return -val if neg else val
    vcc = np.zeros(4, bool)
    nexec = 0
    pend = []           # outstanding LDS operations in issue order: the registers a read will write (writes: empty)
    import re as _re

    def regs_of(x):
        if not isinstance(x, str):
            return set()
        x = x.lstrip("-")
        mm = _re.fullmatch(r"v\[(\d+):(\d+)\]", x)
        if mm:
            return set(range(int(mm.group(1)), int(mm.group(2)) + 1))
        return {int(x[1:])} if _re.fullmatch(r"v\d+", x) else set()
    assert ins[pc][0] == "quad_begin"
    pc += 1
    with np.errstate(all="ignore"):
        while ins[pc][0] != "quad_end":
            t = ins[pc]
            m = t[0]
            if m in ("label", "kill"):
                pc += 1
                continue
            nexec += 1
            assert nexec < max_exec, "runaway quad section"
            if m[0] == "v" or m.startswith("ds_"):
                used = set().union(*[regs_of(x) for x in t[1:]])
                for dst in pend:
                    assert not (dst & used), ("register used before its LDS read was waited for", t)
                if m.startswith("ds_read"):
                    pend.append(regs_of(t[1]))
                elif m.startswith("ds_write"):
                    pend.append(set())
            if m == "s_waitcnt":
                for part in t[1].split():
                    if part.startswith("lgkmcnt("):
                        del pend[:max(0, len(pend) - int(part[8:-1]))]
            elif m == "s_nop":
                pass
            elif m == "s_mov_b32":
                S[int(t[1][1:])] = (t[2] & 0xFFFFFFFF) if isinstance(t[2], int) else S.get(int(t[2][1:]), 0)
            elif m == "s_mov_b64":
                if t[2] == "exec":
                    masks[t[1]] = exec_.copy()
                else:
                    assert t[1] == "exec"
                    exec_ = masks[t[2]].copy()
            elif m == "s_and_b64":
                # lane-class mask & entry EXEC: the low word of the first source was set by s_mov_b32 just before
                lo = int(t[2][2:t[2].index(":")])
                word = S[lo]
                masks[t[1]] = np.array([(word >> ln) & 1 for ln in range(4)], bool) & masks[t[3]]
            elif m == "s_or_b64":
                masks[t[1]] = masks[t[2]] | masks[t[3]]
            elif m in ("s_sub_i32", "s_add_i32"):
                a = S.get(int(t[2][1:]), 0) if isinstance(t[2], str) else t[2]
                b = S.get(int(t[3][1:]), 0) if isinstance(t[3], str) else t[3]
                S[int(t[1][1:])] = (a - b if m == "s_sub_i32" else a + b) & 0xFFFFFFFF
            elif m in ("s_cmp_lt_i32", "s_cmp_gt_i32"):
                sx = lambda x: (lambda w: w - (1 << 32) if w & 0x80000000 else w)(S.get(int(x[1:]), 0) if isinstance(x, str) else x & 0xFFFFFFFF)
                a, b = sx(t[1]), sx(t[2])
                scc = int(a < b) if m == "s_cmp_lt_i32" else int(a > b)
            elif m == "s_cbranch_scc1":
                if scc:
                    lab, d = t[1][:-1], t[1][-1]
                    c = labels[lab]
                    pc = min(x for x in c if x > pc) if d == "f" else max(x for x in c if x < pc)
            elif m == "ds_read_b128":
                lo = int(t[1][2:t[1].index(":")])
                w0 = t[3] // 1024 * 4
                for ln in range(4):
                    if exec_[ln]:
                        V[ln, lo:lo + 4] = lds[ln, w0:w0 + 4]
            elif m == "ds_write_b128":
                lo = int(t[2][2:t[2].index(":")])
                w0 = t[3] // 1024 * 4
                for ln in range(4):
                    if exec_[ln]:
                        lds[ln, w0:w0 + 4] = V[ln, lo:lo + 4]
            elif m == "v_accvgpr_read_b32":
                V[exec_, vi(t[1])] = A[exec_, int(t[2][1:])]
            elif m == "v_accvgpr_write_b32":
                A[exec_, int(t[1][1:])] = V[exec_, vi(t[2])]
            elif m == "v_mov_b32":
                if isinstance(t[2], str) and t[2][0] == "v":
                    V[exec_, vi(t[1])] = V[exec_, vi(t[2])]
                else:
                    setv(t[1], fsrc(t[2]))
            elif m == "v_mov_b32_dpp":
                setv(t[1], dpp_src(vi(t[2]), t[3]))
            elif m == "v_fmac_f32":
                setv(t[1], (fsrc(t[2]).astype(np.float64) * fsrc(t[3]).astype(np.float64) + fsrc(t[1]).astype(np.float64)).astype(f32))
            elif m == "v_fmac_f32_dpp":
                a = dpp_src(vi(t[2]), t[4]).astype(np.float64)
                setv(t[1], (a * fsrc(t[3]).astype(np.float64) + fsrc(t[1]).astype(np.float64)).astype(f32))
            elif m == "v_fma_f32":
                setv(t[1], (fsrc(t[2]).astype(np.float64) * fsrc(t[3]).astype(np.float64) + fsrc(t[4]).astype(np.float64)).astype(f32))
            elif m == "v_mul_f32":
                setv(t[1], fsrc(t[2]) * fsrc(t[3]))
            elif m == "v_add_f32":
                setv(t[1], fsrc(t[2]) + fsrc(t[3]))
            elif m == "v_sub_f32":
                setv(t[1], fsrc(t[2]) - fsrc(t[3]))
            elif m in ("v_max_f32", "v_min_f32", "v_max_f32_dpp"):
                a, b = (dpp_src(t[2], t[4]) if m.endswith("_dpp") else fsrc(t[2])), fsrc(t[3])
                r = np.where(b != b, a, np.where(a != a, b, np.minimum(a, b) if m == "v_min_f32" else np.maximum(a, b)))
                setv(t[1], r)
            elif m == "v_max3_f32":
                a, b, c = fsrc(t[2]), fsrc(t[3]), fsrc(t[4])
                setv(t[1], np.fmax(np.fmax(a, b), c))
            elif m == "v_and_b32":
                assert t[2] == 0x7fffffff
                V[exec_, vi(t[1])] = (V[:, vi(t[3])] & u32(0x7fffffff))[exec_]
            elif m in ("v_mul_f32_dpp", "v_add_f32_dpp"):
                a, b = dpp_src(t[2], t[4]), fsrc(t[3])
                setv(t[1], a * b if m == "v_mul_f32_dpp" else a + b)
            elif m == "v_cmp_lt_f32":
                assert t[1] == "vcc"
                vcc[exec_] = (fsrc(t[2]) < fsrc(t[3]))[exec_]
            elif m == "v_cndmask_b32":
                assert t[4] == "vcc"
                a, b = fsrc(t[2]), fsrc(t[3])
                setv(t[1], np.where(vcc, b, a))
            elif m == "v_rsq_f32":
                setv(t[1], (1.0 / np.sqrt(fsrc(t[2]).astype(np.float64))).astype(f32))
            elif m == "v_rcp_f32":
                setv(t[1], f32(1.0) / fsrc(t[2]))
            elif m == "v_pk_mov_b32":
                d = t[-1]
                lo = int(t[1][2:t[1].index(":")])
                r0, r1 = half(t[2], d["op_sel"][0]), half(t[3], d["op_sel"][1])
                V[exec_, lo] = bits(r0.astype(f32))[exec_]
                V[exec_, lo + 1] = bits(r1.astype(f32))[exec_]
            elif m in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"):
                d = t[-1]
                srcs = t[2:-1]
                dlo = int(t[1][2:t[1].index(":")])
                res = []
                for hi in (0, 1):
                    sel = d["op_sel_hi"] if hi else d["op_sel"]
                    ng = d["neg_hi"] if hi else d["neg_lo"]
                    vals = [half(x, sel[q]) * (-1 if ng[q] else 1) for q, x in enumerate(srcs)]
                    if m == "v_pk_fma_f32":
                        res.append((vals[0] * vals[1] + vals[2]).astype(f32))
                    elif m == "v_pk_mul_f32":
                        res.append(vals[0].astype(f32) * vals[1].astype(f32))
                    else:
                        res.append(vals[0].astype(f32) + vals[1].astype(f32))
                V[exec_, dlo] = bits(res[0])[exec_]
                V[exec_, dlo + 1] = bits(res[1])[exec_]
            else:
                raise ValueError("unknown instruction in the quad section: %r" % (t,))
            pc += 1
    assert exec_.all() and not pend, "EXEC not restored / LDS operations outstanding at the end of the quad section"
    return pc + 1, nexec

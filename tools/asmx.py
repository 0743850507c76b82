"""EXPERIMENT, CLOSED in round 5 (DESIGN.md "Two lanes per robot": measured, not built; this file lives in tools/, outside the
product package): the ADMM iteration with TWO LANES PER ROBOT, so that two
wavefronts fit on a SIMD (256 unified registers + 80 LDS words per lane; a robot's 561-word working set is split over
a lane pair, no AGPR is needed) and the SIMD issues from two waves: measured 2.6 cycles per VOP2 SIMD-instruction
instead of 5.2 for a lone wave, 3.5 per packed one, 3.7 per DPP-operand one (tools/microbench2.hip).

Placement: every array is stored by the slot maps of asmgen (x/y members of a triple are slot partners); slot s lives in
lane s & 1 of the pair, register base + (s >> 1). Element-wise phases therefore run on HALF the registers, packed two
registers per instruction. A triangular-solve op  W[d] += (-L_j) W[s]  executes in the lane that owns d; the source
is read own-lane (v_fmac_f32) or through DPP quad_perm (swap [1,0,3,2], broadcast [0,0,2,2] / [1,1,3,3]); an op of
lane 0 and an op of lane 1 with the same destination register and the same source register share ONE instruction
(the coefficient register holds a different L entry in each lane); ops without a partner run under an exec mask of
their lane.

This module generates the iteration body, interprets it on a lane pair (simulate) against a numpy statement of the same
iteration, and emits tools/microbench_x.hip, which times it on the MI355X next to the shipped one-lane body."""
import os

import numpy as np

import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from robobee3d_amd import asmgen, symbolic              # noqa: E402
from robobee3d_amd.asmgen import Emit, pk, _sb, _vp, f32bits     # noqa: E402

# per-lane register map (v0 = robot offset, v1 = lane LDS address)
VWX, VWZ, VX, VY, VZT, VDX, VDZ, VT3, VTMP, VL = 2, 26, 46, 70, 90, 92, 116, 136, 144, 160
NLREG = 256 - VL            # L coefficients resident in registers; the rest stream from LDS
# per-lane LDS quads (80 words = 20 quads at two waves per SIMD): q (6), l of the dynamics rows (5), L stream (9)
LQ_Q, LQ_LO, LQ_L, LQ_END = 0, 6, 11, 20
DESPACE = int(os.environ.get("UMPC_X_DESPACE", "1"))
S_ALPHA, S_OMA, S_SIGMA, S_RINV, S_RHO = asmgen.S_ALPHA, asmgen.S_OMA, asmgen.S_SIGMA, asmgen.S_RINV, asmgen.S_RHO
S_MA, S_MB, S_ALL, S_CNT = 36, 38, 40, 14      # exec masks of lane 0 / lane 1 of every pair, all lanes; loop counter


def quad_perm(la, lb):
    """lane 0 of a pair reads the source from lane la, lane 1 from lane lb (0 / 1 = which lane of the pair)"""
    return "quad_perm:[%d,%d,%d,%d]" % (la, lb, 2 + la, 2 + lb)


class Plan:
    def __init__(self, N=3, perm=None):
        s = symbolic.analyse(N, perm)
        self.s = s
        self.xs, self.zs, self.xinv, self.zinv = asmgen.slot_maps(s)
        self.neq = 2 * s.N * symbolic.NY
        nk = s.nk

        def home(k):
            o = s.perm[k]
            slot, base = (self.xs[o], VWX) if o < s.nx else (self.zs[o - s.nx], VWZ)
            return slot & 1, base + (slot >> 1)
        self.home = [home(k) for k in range(nk)]
        ents = [(s.L_i[j], c, j) for c in range(nk) for j in range(s.L_p[c], s.L_p[c + 1])]
        self.fwd = self.schedule([(r, c, j) for (r, c, j) in ents])
        mate = {}
        for ins in self.fwd:
            if len(ins["ops"]) == 2:
                mate[ins["ops"][0][2]], mate[ins["ops"][1][2]] = ins["ops"][1][2], ins["ops"][0][2]
        self.bwd = self.schedule([(c, r, j) for (r, c, j) in ents], prefer=mate)
        # coefficient storage: a word index holds one L entry per lane; instructions share an index when their lanes'
        # entries agree (a single may sit in the free lane of another instruction's word)
        self.coef = []              # per index: {lane: entry}
        allins = sorted(self.fwd + self.bwd, key=lambda ins: -len(ins["ops"]))
        for ins in allins:
            want = {ln: o[2] for ln, o in ins["ops"].items()}
            fit = [ci for ci, c in enumerate(self.coef) if all(c.get(ln, j) == j for ln, j in want.items())]
            exact = [ci for ci in fit if all(self.coef[ci].get(ln) == j for ln, j in want.items())]
            if exact:
                ci = exact[0]
            elif fit:
                ci = fit[0]
                self.coef[ci].update(want)
            else:
                ci = len(self.coef)
                self.coef.append(dict(want))
            ins["ci"] = ci
        # renumber: words used by both solves first (they stay in registers), the others in order of first use (they
        # stream from LDS in quads)
        seq = self.fwd + self.bwd
        nf = len(self.fwd)
        uses = {}
        for q, ins in enumerate(seq):
            uses.setdefault(ins["ci"], []).append(q)
        both = lambda ci: uses[ci][0] < nf <= uses[ci][-1]
        order = sorted(uses, key=lambda ci: (not both(ci), uses[ci][0]))
        ren = {ci: k for k, ci in enumerate(order)}
        self.coef = [self.coef[ci] for ci in order]
        for ins in seq:
            ins["ci"] = ren[ins["ci"]]
        self.nboth = sum(both(ci) for ci in uses)

    def schedule(self, ops, prefer=None):
        """static pairing, then a topological order. Ops (d, s, j) that share destination register and source register
        and execute in different lanes are matched up front (entry pairs in `prefer`, the matching of the other direction,
        first: a pair kept in both directions needs ONE coefficient word per lane); a pair whose two ops depend on each
        other through other ops is split again. Returns instructions dict(dreg, sreg, ops={lane: (d, s, j)},
        src={lane: source lane}) in issue order, runs of the same exec state kept together."""
        prefer = prefer or {}
        byreg = {}
        for o in ops:
            (ld, rd), (ls, rs) = self.home[o[0]], self.home[o[1]]
            byreg.setdefault((rd, rs), {0: [], 1: []})[ld].append(o)
        groups = []
        for v in byreg.values():
            a, b = list(v[0]), list(v[1])
            for o in list(a):
                m = [q for q in b if prefer.get(o[2]) == q[2]]
                if m:
                    groups.append({0: o, 1: m[0]}); a.remove(o); b.remove(m[0])
            while a and b:
                groups.append({0: a.pop(0), 1: b.pop(0)})
            groups += [{0: o} for o in a] + [{1: o} for o in b]
        while True:
            writers = {}
            for gi, g in enumerate(groups):
                for o in g.values():
                    writers.setdefault(o[0], set()).add(gi)
            deps = [set().union(*[writers.get(o[1], set()) for o in g.values()]) - {gi} for gi, g in enumerate(groups)]
            done, order, state = set(), [], 2
            while len(order) < len(groups):
                ready = [gi for gi in range(len(groups)) if gi not in done and deps[gi] <= done]
                if not ready:
                    break
                st = lambda gi: 2 if len(groups[gi]) == 2 else next(iter(groups[gi]))
                same = [gi for gi in ready if st(gi) == state]
                if not same:
                    cnt = {k: sum(st(gi) == k for gi in ready) for k in (0, 1, 2)}
                    state = 2 if cnt[2] else max((0, 1), key=lambda k: cnt[k])
                    same = [gi for gi in ready if st(gi) == state]
                for gi in same:
                    order.append(gi); done.add(gi)
            if len(order) == len(groups):
                break
            def on_cycle(g0):
                seen, todo = set(), list(deps[g0] - done)
                while todo:
                    x = todo.pop()
                    if x == g0:
                        return True
                    if x not in seen:
                        seen.add(x)
                        todo += list(deps[x] - done)
                return False
            stuck = [gi for gi in range(len(groups)) if gi not in done and len(groups[gi]) == 2 and on_cycle(gi)]
            assert stuck
            g = groups.pop(stuck[0])
            groups += [{0: g[0]}, {1: g[1]}]
        # issue order: a dependent VALU instruction issued right behind its producer waits ~8.5 cycles, and with two waves
        # on the SIMD a wave's consecutive instructions are only ~5 apart (tools/microbench_x.hip: the naive order gained
        # nothing over one lane per robot), so among the ready instructions one that does not read what the previous
        # one wrote goes first, then one that keeps the exec mask, then the longest remaining chain.
        regs = lambda gi: (self.home[next(iter(groups[gi].values()))[0]][1], self.home[next(iter(groups[gi].values()))[1]][1])
        succ = [[] for _ in groups]
        for gi, dd in enumerate(deps):
            for x in dd:
                succ[x].append(gi)
        height = [0] * len(groups)
        for gi in reversed(order):
            height[gi] = 1 + max([height[x] for x in succ[gi]] or [0])
        if DESPACE:
            done, order, state, last = set(), [], 2, [None, None]
            npend = [len(dd) for dd in deps]
            ready = {gi for gi in range(len(groups)) if npend[gi] == 0}
            st = lambda gi: 2 if len(groups[gi]) == 2 else next(iter(groups[gi]))
            while ready:
                def score(gi):
                    rd, rs = regs(gi)
                    return (4 * (last[0] in (rd, rs)) + 2 * (st(gi) != state) + (last[1] in (rd, rs)), -height[gi])
                gi = min(ready, key=score)
                ready.discard(gi)
                order.append(gi)
                if st(gi) != state:
                    last = [None, last[0]]      # the s_mov of exec sits between
                    state = st(gi)
                last = [regs(gi)[0], last[0]]
                for x in succ[gi]:
                    npend[x] -= 1
                    if npend[x] == 0:
                        ready.add(x)
            assert len(order) == len(groups)
        out = []
        for gi in order:
            grp = groups[gi]
            any_op = next(iter(grp.values()))
            rd, rs = self.home[any_op[0]][1], self.home[any_op[1]][1]
            out.append(dict(dreg=rd, sreg=rs, ops=grp, src={ln: self.home[o[1]][0] for ln, o in grp.items()}))
        return out


def body(e, plan, variant=()):
    """one ADMM iteration (middle iterations of asmgen.body: z == l on the dynamics rows) for a lane pair"""
    s = plan.s
    nx, nc, neq = s.nx, s.nc, plan.neq
    v = lambda n: "v%d" % n
    nxr, neqr = 2 * ((nx + 3) // 4), neq // 2      # registers per lane: x part (24 with pad), dynamics rows (18)
    # ---- rhs: q and l stream from LDS straight into the work registers
    for k in range(nxr // 4):
        e("ds_read_b128", "v[%d:%d]" % (VWX + 4 * k, VWX + 4 * k + 3), "v1", (LQ_Q + k) * 1024)
    for k in range((neqr + 3) // 4):
        e("ds_read_b128", "v[%d:%d]" % (VWZ + 4 * k, VWZ + 4 * k + 3), "v1", (LQ_LO + k) * 1024)
    nq = nxr // 4 + (neqr + 3) // 4
    for r in range(0, nxr, 2):
        if r % 4 == 0:
            e("s_waitcnt", "lgkmcnt(%d)" % (nq - 1 - r // 4))
        pk(e, "v_pk_fma_f32", VWX + r, [_sb(S_SIGMA), _vp(VX + r), _vp(VWX + r)], [0, 0, 1])
    for r in range(0, neqr, 2):
        if r % 4 == 0:
            e("s_waitcnt", "lgkmcnt(%d)" % ((neqr + 3) // 4 - 1 - r // 4))
        pk(e, "v_pk_fma_f32", VWZ + r, [_sb(S_RINV), _vp(VY + r), _vp(VWZ + r)], [1, 0, 0])
    for k in range(2):      # thrust rows: slots 36..39 -> registers 18, 19 of the z arrays
        e("v_fma_f32", v(VWZ + neqr + k), "-" + v(VT3 + 6 + k), v(VY + neqr + k), v(VZT + k))
    # ---- solves
    state = [2]

    def set_exec(want):
        if want != state[0] and "noexec" in variant:
            state[0] = want
        if want != state[0]:
            e("s_mov_b64", "exec", "s[%d:%d]" % ({0: S_MA, 1: S_MB, 2: S_ALL}[want], {0: S_MA, 1: S_MB, 2: S_ALL}[want] + 1))
            state[0] = want
    # streamed coefficient quads: FIFO over the instruction sequence, two quads usable, two more in flight (4 ring slots
    # in the 16 temporaries, which the solves do not need otherwise)
    seq = plan.fwd + plan.bwd
    ring = [VTMP + 4 * k for k in range(4)]
    events, where = [], {}           # events: (first instruction index, quad); where[instruction index] = (event, word)
    for i, ins in enumerate(seq):
        if ins["ci"] >= NLREG:
            w = ins["ci"] - NLREG
            live = [k for k in range(max(0, len(events) - 2), len(events)) if events[k][1] == w // 4]
            if not live:
                events.append((i, w // 4))
                live = [len(events) - 1]
            where[i] = (live[-1], w % 4)
    need = {ev[0]: k for k, ev in enumerate(events)}

    def issue(k):
        assert LQ_L + events[k][1] < LQ_END
        slot = ring[k % 4]
        e("ds_read_b128", "v[%d:%d]" % (slot, slot + 3), "v1", (LQ_L + events[k][1]) * 1024)
    lastw = [set(), set()]          # VGPRs written by the previous two instructions (DPP source hazard: 2 wait states)

    def emit_solve(lo_, hi_):
        for i in range(lo_, hi_):
            ins = seq[i]
            if i in need:
                k = need[i]
                set_exec(2)
                e("s_waitcnt", "lgkmcnt(%d)" % min(1, len(events) - 1 - k))
                if k + 2 < len(events):
                    issue(k + 2)
                lastw[0], lastw[1] = set(), set()
            lr = VL + ins["ci"] if ins["ci"] < NLREG else ring[where[i][0] % 4] + where[i][1]
            lanes = sorted(ins["ops"])
            if state[0] != (2 if len(lanes) == 2 else lanes[0]):
                lastw[0], lastw[1] = set(), lastw[0]
            set_exec(2 if len(lanes) == 2 else lanes[0])
            la = ins["src"].get(0, 0)
            lb = ins["src"].get(1, 1)
            if (la == 0 and lb == 1) or "nodpp" in variant:
                e("v_fmac_f32", v(ins["dreg"]), v(lr), v(ins["sreg"]))
            else:
                if ins["sreg"] in lastw[0]:
                    e("s_nop", 1)
                elif ins["sreg"] in lastw[1]:
                    e("s_nop", 0)
                e("v_fmac_f32_dpp", v(ins["dreg"]), v(ins["sreg"]), v(lr), quad_perm(la, lb) + " row_mask:0xf bank_mask:0xf")
            lastw[0], lastw[1] = {ins["dreg"]}, lastw[0]
    for k in range(min(2, len(events))):
        issue(k)
    emit_solve(0, len(plan.fwd))
    set_exec(2)
    for r in range(0, nxr, 2):
        pk(e, "v_pk_mul_f32", VWX + r, [_vp(VWX + r), _vp(VDX + r)])
    for r in range(0, 20, 2):
        pk(e, "v_pk_mul_f32", VWZ + r, [_vp(VWZ + r), _vp(VDZ + r)])
    lastw[0], lastw[1] = {VWZ + 18, VWZ + 19}, {VWZ + 16, VWZ + 17}
    emit_solve(len(plan.fwd), len(seq))
    set_exec(2)
    # ---- x, y, z updates, software-pipelined over four temporary pairs so that no instruction reads its predecessor
    t = VTMP
    first, second = [], []
    for r in range(0, nxr, 2):
        tp = t + 2 * ((r // 2) % 4)
        first.append(("v_pk_mul_f32", tp, [_sb(S_OMA), _vp(VX + r)], None))
        second.append(("v_pk_fma_f32", VX + r, [_sb(S_ALPHA), _vp(VWX + r), _vp(tp)], None))
    for r in range(0, neqr, 2):
        tp = t + 2 * (((nxr + r) // 2) % 4)
        first.append(("v_pk_add_f32", tp, [_vp(VWZ + r), _vp(VY + r)], [0, 1]))
        second.append(("v_pk_fma_f32", VY + r, [_sb(S_ALPHA), _vp(tp), _vp(VY + r)], None))
    lag = 3
    for k in range(len(first) + lag):
        if k < len(first):
            pk(e, first[k][0], first[k][1], first[k][2], first[k][3])
        if k >= lag:
            pk(e, second[k - lag][0], second[k - lag][1], second[k - lag][2], second[k - lag][3])
    chains = []
    for k in range(2):
        nu, y, z = v(VWZ + neqr + k), v(VY + neqr + k), v(VZT + k)
        lo, up, rho, rinv = v(VT3 + k), v(VT3 + 2 + k), v(VT3 + 4 + k), v(VT3 + 6 + k)
        t1, t2, t3 = v(t + 8 + 4 * k), v(t + 8 + 4 * k + 1), v(t + 8 + 4 * k + 2)
        chains.append([("v_fma_f32", t1, "-" + rinv, y, z), ("v_mul_f32", t2, "s%d" % S_OMA, z),
                       ("v_fma_f32", t1, rinv, nu, t1), ("v_fma_f32", t1, "s%d" % S_ALPHA, t1, t2),
                       ("v_fma_f32", t3, rinv, y, t1), ("v_max_f32", t3, t3, lo), ("v_min_f32", z, t3, up),
                       ("v_sub_f32", t2, t1, z), ("v_mul_f32", t2, rho, t2), ("v_add_f32", y, y, t2)])
    for a_, b_ in zip(*chains):
        e(*a_)
        e(*b_)
    return len(events)


def reference_iteration(plan, d):
    """numpy (float64) statement of the same iteration on the unsplit arrays in d: x, y, z, q, lo (dynamics rows), lo3,
    up3, rho3, rinv3, L (by CSC entry, NOT negated), Dinv (permuted order)"""
    s = plan.s
    nx, nc, neq, nk = s.nx, s.nc, plan.neq, s.nk
    sigma, alpha = 1e-6, 1.6
    rinv = np.concatenate((np.full(neq, 0.01), d["rinv3"]))
    rho = np.concatenate((np.full(neq, 100.0), d["rho3"]))
    z = np.concatenate((d["lo"], d["z3"]))
    rhs = np.concatenate((sigma * d["x"] - d["q"], z - rinv * d["y"]))
    w = rhs[s.perm].copy()
    for c in range(nk):
        for j in range(s.L_p[c], s.L_p[c + 1]):
            w[s.L_i[j]] -= d["L"][j] * w[c]
    w *= d["Dinv"]
    for c in range(nk - 1, -1, -1):
        for j in range(s.L_p[c], s.L_p[c + 1]):
            w[c] -= d["L"][j] * w[s.L_i[j]]
    sol = np.empty(nk)
    sol[s.perm] = w
    xt, nu = sol[:nx], sol[nx:]
    x = alpha * xt + (1 - alpha) * d["x"]
    zt = z - rinv * d["y"] + rinv * nu
    tt = alpha * zt + (1 - alpha) * z
    lo = np.concatenate((d["lo"], d["lo3"]))
    up = np.concatenate((d["lo"], d["up3"]))
    zn = np.minimum(np.maximum(tt + rinv * d["y"], lo), up)
    y = d["y"] + rho * (tt - zn)
    return x, y, zn[neq:]


def load_pair(plan, d):
    """register images V[lane][256] (float32) and the per-lane LDS words of the streamed coefficients for data d"""
    s = plan.s
    V = np.zeros((2, 256), np.float32)
    lds = np.zeros((2, 4 * LQ_END), np.float32)

    def put(base, slot, val):
        V[slot & 1, base + (slot >> 1)] = val
    for j in range(s.nx):
        put(VX, plan.xs[j], d["x"][j])
        lds[plan.xs[j] & 1, 4 * LQ_Q + (plan.xs[j] >> 1)] = d["q"][j]
    for i in range(s.nc):
        put(VY, plan.zs[i], d["y"][i])
    for i in range(plan.neq):
        lds[plan.zs[i] & 1, 4 * LQ_LO + (plan.zs[i] >> 1)] = d["lo"][i]
    for k in range(s.N):
        sl = plan.zs[plan.neq + k]
        lane, r = sl & 1, (sl >> 1) - plan.neq // 2
        V[lane, VZT + r] = d["z3"][k]
        V[lane, VT3 + r], V[lane, VT3 + 2 + r] = d["lo3"][k], d["up3"][k]
        V[lane, VT3 + 4 + r], V[lane, VT3 + 6 + r] = d["rho3"][k], d["rinv3"][k]
    for k in range(s.nk):
        lane, reg = plan.home[k]
        V[lane, (VDX if reg < VWZ else VDZ) + reg - (VWX if reg < VWZ else VWZ)] = d["Dinv"][k]
    for ci, owners in enumerate(plan.coef):
        for lane, j in owners.items():
            if ci < NLREG:
                V[lane, VL + ci] = -d["L"][j]
            else:
                lds[lane, 4 * LQ_L + ci - NLREG] = -d["L"][j]
    return V, lds


def simulate(ins, V, lds):
    """interprets the body on ONE lane pair; V [2][256] float32 and lds [2][words] are updated in place"""
    f32 = np.float32
    S = {S_ALPHA: f32(1.6), S_OMA: f32(f32(1.0) - f32(1.6)), S_SIGMA: f32(1e-6), S_RINV: f32(0.01), S_RHO: f32(100.0)}
    exec_ = [1, 1]
    for t in ins:
        m = t[0]
        lanes = [ln for ln in (0, 1) if exec_[ln]]
        if m == "s_mov_b64":
            exec_ = {S_MA: [1, 0], S_MB: [0, 1], S_ALL: [1, 1]}[int(t[2][2:t[2].index(":")])]
        elif m in ("s_waitcnt", "s_nop"):
            pass
        elif m == "ds_read_b128":
            lo = int(t[1][2:t[1].index(":")])
            w0 = t[3] // 1024 * 4
            for ln in lanes:
                V[ln, lo:lo + 4] = lds[ln, w0:w0 + 4]
        elif m in ("v_fmac_f32", "v_fmac_f32_dpp"):
            d_ = int(t[1][1:])
            if m == "v_fmac_f32":
                a_, b_ = int(t[2][1:]), int(t[3][1:])
                for ln in lanes:
                    V[ln, d_] = f32(np.float64(V[ln, a_]) * np.float64(V[ln, b_]) + np.float64(V[ln, d_]))
            else:
                sreg, lreg = int(t[2][1:]), int(t[3][1:])
                qp = [int(c) for c in t[4][t[4].index("[") + 1:t[4].index("]")].split(",")]
                old = V[:, sreg].copy()
                for ln in lanes:
                    V[ln, d_] = f32(np.float64(old[qp[ln]]) * np.float64(V[ln, lreg]) + np.float64(V[ln, d_]))
        elif m in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"):
            d_ = t[-1]
            srcs = t[2:-1]
            dlo = int(t[1][2:t[1].index(":")])
            for ln in lanes:
                res = []
                for hi in (0, 1):
                    sel = d_["op_sel_hi"] if hi else d_["op_sel"]
                    ng = d_["neg_hi"] if hi else d_["neg_lo"]
                    vals = []
                    for q, x in enumerate(srcs):
                        lo = int(x[2:x.index(":")])
                        val = V[ln, lo + sel[q]] if x[0] == "v" else S[lo]
                        vals.append(np.float64(val) * (-1 if ng[q] else 1))
                    res.append(f32(vals[0] * vals[1] + vals[2]) if m == "v_pk_fma_f32" else
                               f32(f32(vals[0]) * f32(vals[1])) if m == "v_pk_mul_f32" else f32(f32(vals[0]) + f32(vals[1])))
                V[ln, dlo], V[ln, dlo + 1] = res
        else:
            def val(x, ln):
                neg = x.startswith("-")
                x = x[1:] if neg else x
                r = V[ln, int(x[1:])] if x[0] == "v" else S[int(x[1:])]
                return -np.float64(r) if neg else np.float64(r)
            for ln in lanes:
                d_ = int(t[1][1:])
                if m == "v_fma_f32":
                    V[ln, d_] = f32(val(t[2], ln) * val(t[3], ln) + val(t[4], ln))
                elif m == "v_mul_f32":
                    V[ln, d_] = f32(f32(val(t[2], ln)) * f32(val(t[3], ln)))
                elif m == "v_add_f32":
                    V[ln, d_] = f32(f32(val(t[2], ln)) + f32(val(t[3], ln)))
                elif m == "v_sub_f32":
                    V[ln, d_] = f32(f32(val(t[2], ln)) - f32(val(t[3], ln)))
                elif m == "v_max_f32":
                    V[ln, d_] = max(f32(val(t[2], ln)), f32(val(t[3], ln)))
                elif m == "v_min_f32":
                    V[ln, d_] = min(f32(val(t[2], ln)), f32(val(t[3], ln)))
                else:
                    raise ValueError(t)


def fmt(t):
    if t[0] == "v_fmac_f32_dpp":
        return "v_fmac_f32_dpp %s, %s, %s %s" % (t[1], t[2], t[3], t[4])
    return asmgen.fmt(t)


def write_microbench(path=None):
    """tools/microbench_x.hip: the two-lane body at 8 waves per CU (2 per SIMD) against the shipped one-lane middle
    iteration at 4 waves per CU, same harness, registers filled with finite values, iterations timed by HIP events.
    Two more kernels bound what the exec-mask switches and the DPP operands cost (their results are NOT an ADMM
    iteration): X without the s_mov of exec, X with plain operands instead of DPP ones."""
    path = path or os.path.join(HERE, "microbench_x.hip")
    plan = Plan()
    s = plan.s
    streams = {}
    for name, variant in (("kx", ()), ("kx_noexec", ("noexec",)), ("kx_nodpp", ("nodpp",)), ("kx_neither", ("noexec", "nodpp"))):
        e = Emit()
        nl = body(e, plan, variant)
        streams[name] = e.ins
    xins = streams["kx"]
    e1 = Emit()
    asmgen.body(e1, s, first=False, capture=False, plan=asmgen.solve_plan(s), lv=True)
    oins = e1.ins

    def block(ins, fm):
        return "\n".join('    "%s\\n"' % fm(t) for t in ins)
    clobx = ", ".join(['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in range(2, 256)] + ['"s14"'] +
                      ['"s%d"' % i for i in list(range(20, 30)) + list(range(36, 42))])
    clobo = ", ".join(['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in range(2, 246)] + ['"a%d"' % i for i in range(256)] +
                      ['"s14"'] + ['"s%d"' % i for i in range(20, 30)])
    consts = "\n".join('    "s_mov_b32 s%d, 0x%x\\n"\n    "s_mov_b32 s%d, 0x%x\\n"' % (r, f32bits(val), r + 1, f32bits(val))
                       for r, val in ((S_ALPHA, 1.6), (S_OMA, float(np.float32(1.0) - np.float32(1.6))), (S_SIGMA, 1e-6),
                                      (S_RINV, 0.01), (S_RHO, 100.0)))
    initv = lambda hi, acc=False: "\n".join('    "v_mov_b32 v%d, 0x3c23d70a\\n"' % r for r in range(2, hi)) + \
        ("\n" + "\n".join('    "v_accvgpr_write_b32 a%d, v2\\n"' % r for r in range(256)) if acc else "")
    kx = '''__global__ __launch_bounds__(64) void %s(float *out, int iters) {
  __shared__ float4 lds[20 * 64];
  const unsigned ldsaddr = (unsigned)(size_t)(&lds[threadIdx.x]);
  for (int q = 0; q < 20; ++q) lds[q * 64 + threadIdx.x] = make_float4(0.01f, 0.01f, 0.01f, 0.01f);
  __syncthreads();
  asm volatile(
    "s_mov_b32 s36, 0x55555555\\n" "s_mov_b32 s37, 0x55555555\\n" "s_mov_b32 s38, 0xaaaaaaaa\\n" "s_mov_b32 s39, 0xaaaaaaaa\\n"
    "s_mov_b64 s[40:41], exec\\n"
%s
%s
    "s_mov_b32 s14, %%1\\n"
    "1:\\n"
%s
    "s_mov_b64 exec, s[40:41]\\n"
    "s_sub_i32 s14, s14, 1\\n" "s_cmp_gt_i32 s14, 0\\n" "s_cbranch_scc1 1b\\n"
    : : "{v1}"(ldsaddr), "s"(iters) : %s);
  if (iters < 0) out[threadIdx.x] = 0;
}
'''
    txt = '''// GENERATED by tools/asmx.py (experiment, closed in round 5) -- do not edit.
// Times one ADMM iteration: X = two lanes per robot, 8 waves per CU (two per SIMD), %d instructions per wave-iteration
// of 32 robots (%d forward + %d backward solve instructions, %d LDS quads); O = the shipped one-lane middle iteration,
// 4 waves per CU, %d instructions per wave-iteration of 64 robots. Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_x tools/microbench_x.hip
#include <hip/hip_runtime.h>
#include <cstdio>
''' % (len(xins), len(plan.fwd), len(plan.bwd), nl, len(oins))
    for name in streams:
        txt += kx % (name, consts, initv(256), block(streams[name], fmt), clobx)
    txt += '''__global__ __launch_bounds__(64) void ko(float *out, int iters) {
  __shared__ float4 lds[40 * 64];
  const unsigned ldsaddr = (unsigned)(size_t)(&lds[threadIdx.x]);
  for (int q = 0; q < 40; ++q) lds[q * 64 + threadIdx.x] = make_float4(0.01f, 0.01f, 0.01f, 0.01f);
  __syncthreads();
  asm volatile(
%s
%s
    "s_mov_b32 s14, %%1\\n"
    "1:\\n"
%s
    "s_sub_i32 s14, s14, 1\\n" "s_cmp_gt_i32 s14, 0\\n" "s_cbranch_scc1 1b\\n"
    : : "{v1}"(ldsaddr), "s"(iters) : %s);
  if (iters < 0) out[threadIdx.x] = 0;
}
typedef void (*kern_t)(float *, int);
int main() {
  float *d;
  if (hipMalloc(&d, 4096) != hipSuccess) return 1;
  const int iters = 20000;
  const char *names[5] = {"X  two lanes per robot, 2 waves/SIMD      ", "X  without the exec switches (timing only)", "X  without DPP operands (timing only)      ",
                          "X  without either (timing only)            ", "O  one lane per robot, 1 wave/SIMD         "};
  kern_t ks[5] = {kx, kx_noexec, kx_nodpp, kx_neither, ko};
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    for (int which = 0; which < 5; ++which) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(ks[which], dim3(which == 4 ? 256 * 4 : 256 * 8), dim3(64), 0, 0, d, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      // every launch iterates 65536 robots: X 2048 waves x 32 robots, O 1024 waves x 64 robots
      printf("%%s: %%.3f us per iteration of 65536 robots (%%d iterations, %%.2f ms)\\n", names[which], ms * 1e3 / iters, iters, ms);
    }
  return 0;
}
''' % (consts, initv(246, True), block(oins, asmgen.fmt), clobo)
    with open(path, "w") as f:
        f.write(txt)
    return path, len(xins), len(oins), plan


if __name__ == "__main__":
    p, nx_, no_, plan = write_microbench()
    print("wrote", p, "X body", nx_, "instructions (fwd %d, bwd %d, coefficient words per lane %d), shipped body %d"
          % (len(plan.fwd), len(plan.bwd), len(plan.coef), no_))

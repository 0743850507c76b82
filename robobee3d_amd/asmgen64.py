"""fp64 ADMM phase of the small-batch step kernel (BASELINE config 2: B = 4 096, fp64) as generated gfx950 assembly
(robobee3d_amd/csrc/umpc_admm_asm64.h, one `asm volatile` block inside umpc_rollout_kernel<double, LDSF>).

Why: the fp64 loop's working set is 561 eight-byte words per robot; hipcc keeps ~130 of them in VGPRs and streams the
rest through scratch, one wave per CU with nothing to hide the latency behind: 37 us per ADMM iteration, 1.85 of the
2.32 ms step (tools/f64_breakdown.sh). A lane owns 128 words of VGPR, 128 of AGPR and -- with ONE wave per CU, which is
what a 4 096-robot batch gives anyway -- 320 words of LDS. That is 576 words, and q and l need no home at all:

    v0            robot byte offset (8*b)                 input
    v1            lane LDS address (base + 16*lane)       input;  v2 = v1 + 64 KiB, v3 = v1 + 128 KiB (DS offsets are 16 bit)
    v4..v171      W      KKT rhs / solution, 84 words by ORIGINAL index (x part, then constraint rows)
    v172..v201    z, lo, up, rho, 1/rho of the three thrust rows
    v202..v229    LDS read ring (7 x 16 bytes: a quad carries two words, and ~100 cycles of LDS latency at one
                  fp64 FMA per ~5 cycles need that many quads in flight)
    v230..v237    AGPR read temporaries     v238..v245  arithmetic temporaries      v246..v255  left to the compiler
                  (it needs a few for values that live across the block and for its SGPR spills)
    a0..a167      1/D (permuted order), two v_accvgpr_read per word
    a168..a239    z of the dynamics rows during the FIRST iteration (afterwards z == l there)
    LDS           words 0..212 L (CSC order, written there by phase A), 213..257 x, 258..296 y;
                  word w = byte (w >> 1) * 1024 + 16 * lane + 8 * (w & 1): a ds_read_b128 returns two words
    q, l          never stored on chip: their 81 words are global_load_dwordx2'ed (L2 hits, 8 B x 64 lanes coalesced)
                  straight INTO the W registers they are about to be combined with -- W_x <- q right after the x update
                  freed it, W_z <- l after the y update -- and the rhs is formed in place (W_x = sigma x - W_x).
                  One VMEM instruction per word where an AGPR home costs two v_accvgpr_read.

Per middle iteration ~1 500 instructions (426 v_fma_f64 of the solves, 214 + 86 ds_read_b128, 168 v_accvgpr_read, 81
global loads, 43 ds_write). Arithmetic = UMPC_GEN_ADMM_ITER of the C++ statement (csrc/umpc_gen.h) except that the
backward solve walks each column's entries in descending order and the dynamics rows use delta_y = alpha (nu - y) after
the first iteration (as asmgen.py; rounding only, parity band 1e-9).

Reference mapping: auxil.c:164-228 (compute_rhs, update_x, update_z, update_y), qdldl_interface.c:322-369,
qdldl.c:250-293, proj.c:4-14.

`simulate()` interprets the emitted list in exact-rounded float64 for the CPU tests."""
import os
import struct

from . import asmgen, symbolic
from .asmgen import (Emit, _row_ptr, _adv, FAC_Q, FAC_LOEQ, FAC_M, WS_DS, WS_ES, WS_ROWS,
                     S_WS, S_CTRL, S_STRIDE, S_ITERS, S_P, S_CNT, S_P2, S_ALPHA, S_OMA, S_SIGMA, S_RINV, S_RHO)

HERE = os.path.dirname(os.path.abspath(__file__))

V_B1, V_B2, V_W, V_C, V_RING, V_AT, V_TT, V_END = 2, 3, 4, 172, 202, 230, 238, 246
NSLOT, N_AT, N_TT = 7, 4, 4
A_D, A_Z = 0, 168
NNZL = 213
LW_X, LW_Y, LW_END = 213, 258, 297
# what phase C reads, staged in the (by then dead) L words by the last iteration and the epilogue: the scalings D, E of
# phase A, z, x_prev and delta_y of the last iteration, the thrust-row bounds
PC_DS, PC_ES, PC_Z, PC_XP, PC_DY, PC_LO3, PC_UP3 = 0, 45, 84, 123, 168, 207, 210
LDS_BYTES_PER_LANE = 2560
# Round 3: from the second iteration on q and l come from AGPR homes instead of L2 (one VMEM instruction costs a lone
# wave ~14 ns, two v_accvgpr_read ~4.4: tools/microbench_vmem.hip), and their structural zeros cost nothing:
#   a168..a253   l of the 16 dynamics rows that can be non-zero, then the 27 entries of q that can be (the z words of the
#                first iteration are dead by then; filled at the end of the first iteration)
A_H = 168
HOMES = os.environ.get("UMPC_ASM64_HOMES", "1") == "1"
FUSE = os.environ.get("UMPC_ASM64_FUSE", "1") == "1"       # the x / y updates leave the NEXT iteration's right-hand side in W


def rhs_structure(s):
    """(q entries, dynamics rows) whose right-hand-side data can be non-zero (uprightmpc2.c:126-207: q on the y part and
    the dp triples; l = u on rows 0..5 (-y1), 18..23, 24..26 and 32 (dt g)); everything else is +-0 after scaling too.
    The same facts as asmstep.Struct.qzero / lzero (tests/test_asm64_schedule.py compares them)."""
    assert s.N == 3
    NY = symbolic.NY
    qnz = list(range(s.N * NY)) + [s.N * NY + k * NY + i for k in range(s.N) for i in range(3)]
    lnz = sorted(set(range(6)) | set(range(18, 24)) | set(range(24, 27)) | {32})
    return qnz, lnz


def homes(s):
    """{('l', row) | ('q', column): first AGPR of its home}"""
    qnz, lnz = rhs_structure(s)
    h = {}
    for i in lnz:
        h[("l", i)] = A_H + 2 * len(h)
    for j in qnz:
        h[("q", j)] = A_H + 2 * len(h)
    assert A_H + 2 * len(h) <= 256
    return h


def f64bits(v):
    return struct.unpack("<Q", struct.pack("<d", v))[0]


def vp(n):
    return "v[%d:%d]" % (n, n + 1)


def sp(n):
    return "s[%d:%d]" % (n, n + 1)


def lds_addr(word):
    """(base VGPR, 16-bit offset) of the quad that holds LDS word `word`, and the word's half (0 / 1)"""
    byte = (word >> 1) * 1024
    return "v%d" % (1, V_B1, V_B2)[byte >> 16], byte & 0xFFFF, word & 1


class Fetch:
    """Issues LDS quad reads and AGPR reads ahead of the ops that consume them. An op is dict(srcs=[src...], emit=fn)
    with src = ('L', lds word) | ('A', first AGPR of the word) | ('V', first VGPR of the word); emit receives the first
    VGPR of every source word. LDS quads live in NSLOT ring slots, least recently used replaced (static analysis)."""

    def __init__(self, e, ahead=None, la=2):
        ahead = int(os.environ.get("UMPC_ASM64_AHEAD", "14")) if ahead is None else ahead
        self.e, self.ahead, self.la = e, ahead, la
        self.merge = int(os.environ.get("UMPC_ASM64_MERGE", "1"))     # a wait also covers this many later reads in flight

    def run(self, ops):
        e = self.e
        n = len(ops)
        insts, slots = [], [None] * NSLOT
        inst_of = {}
        for i, op in enumerate(ops):
            for q, src in enumerate(op["srcs"]):
                if src[0] != "L":
                    continue
                qd = src[1] >> 1
                hit = [k for k in slots if k is not None and insts[k]["quad"] == qd]
                if hit:
                    insts[hit[0]]["last"] = i
                    inst_of[(i, q)] = hit[0]
                    continue
                free = [sl for sl in range(NSLOT) if slots[sl] is None]
                sl = free[0] if free else min(range(NSLOT), key=lambda z: insts[slots[z]]["last"])
                insts.append(dict(quad=qd, first=i, last=i, slot=sl, prev=slots[sl], issued=None))
                slots[sl] = len(insts) - 1
                inst_of[(i, q)] = slots[sl]
        nds, waited, next_inst, next_acc, acc_rr = 0, -1, 0, 0, 0
        atemp = {}

        def issue(it):
            nonlocal nds
            base, off, _ = lds_addr(2 * it["quad"])
            r = V_RING + 4 * it["slot"]
            e("ds_read_b128", "v[%d:%d]" % (r, r + 3), base, off)
            it["issued"] = nds
            nds += 1
        for i in range(n):
            while next_acc < n and next_acc < i + self.la:
                for q, src in enumerate(ops[next_acc]["srcs"]):
                    if src[0] == "A":
                        t = V_AT + 2 * (acc_rr % N_AT)
                        acc_rr += 1
                        e("v_accvgpr_read_b32", "v%d" % t, "a%d" % src[1])
                        e("v_accvgpr_read_b32", "v%d" % (t + 1), "a%d" % (src[1] + 1))
                        atemp[(next_acc, q)] = t
                next_acc += 1
            while next_inst < len(insts):
                it = insts[next_inst]
                prev = insts[it["prev"]] if it["prev"] is not None else None
                if it["first"] <= i + self.ahead and (prev is None or prev["last"] < i):
                    issue(it)
                    next_inst += 1
                else:
                    break
            regs = []
            for q, src in enumerate(ops[i]["srcs"]):
                if src[0] == "V":
                    regs.append(src[1])
                elif src[0] == "A":
                    regs.append(atemp.pop((i, q)))
                else:
                    it = insts[inst_of[(i, q)]]
                    if it["issued"] is None:
                        assert inst_of[(i, q)] == next_inst and (it["prev"] is None or insts[it["prev"]]["last"] < i)
                        issue(it)
                        next_inst += 1
                    if it["issued"] > waited:
                        upto = min(nds - 1, it["issued"] + self.merge)
                        e("s_waitcnt", "lgkmcnt(%d)" % min(15, nds - 1 - upto))
                        waited = upto if nds - 1 - upto <= 15 else it["issued"]
                    regs.append(V_RING + 4 * it["slot"] + 2 * (src[1] & 1))
            ops[i]["emit"](regs)


SCHED_LAT = int(os.environ.get("UMPC_ASM64_SCHED_LAT", "0"))
SCHED_SLACK = int(os.environ.get("UMPC_ASM64_SCHED_SLACK", "2"))


def list_schedule(ops, lat=None):
    """Reorders the rhs / solve operations so that an operation follows the one that produced its operand by at least
    `lat` issue slots where the dependence graph allows (one wave per SIMD: nothing else hides a dependent fp64 FMA's
    latency). Every word keeps the order of ITS updates (read-after-write, write-after-read and write-after-write
    edges on the W words), so each value is rounded exactly as in program order. Priority: longest latency-weighted path
    to the end; ties in program order (= storage order of L, which keeps the two words of an LDS quad together)."""
    lat = SCHED_LAT if lat is None else lat
    if lat <= 0:
        return ops
    n = len(ops)
    preds = [[] for _ in range(n)]      # (op, is_raw)
    last_w, readers = {}, {}
    for i, o in enumerate(ops):
        for w in o["rd"]:
            if w in last_w:
                preds[i].append((last_w[w], True))
            readers.setdefault(w, []).append(i)
        w = o["wr"]
        if w in last_w:
            preds[i].append((last_w[w], True))      # read-modify-write of the word
        for r in readers.get(w, ()):
            if r != i:
                preds[i].append((r, False))
        last_w[w] = i
        readers[w] = []
    succs = [[] for _ in range(n)]
    for i in range(n):
        for (p_, raw) in preds[i]:
            succs[p_].append((i, raw))
    prio = [0] * n
    for i in range(n - 1, -1, -1):
        prio[i] = max([prio[j] + (lat if raw else 1) for (j, raw) in succs[i]] or [0])
    npred = [len(set(p_ for p_, _ in preds[i])) for i in range(n)]
    pset = [set(p_ for p_, _ in preds[i]) for i in range(n)]
    rawset = [set(p_ for p_, raw in preds[i] if raw) for i in range(n)]
    issued_at = {}
    quad_of = [next((src[1] >> 1 for src in o["srcs"] if src[0] == "L"), None) for o in ops]
    recent = []
    ready = [i for i in range(n) if npred[i] == 0]
    out, t = [], 0
    remaining = [len(ps) for ps in pset]
    users = [sorted(set(j for j, _ in succs[i])) for i in range(n)]
    while ready:
        def avail(i):
            return max([issued_at[p_] + lat for p_ in rawset[i]] or [0])
        ok = [i for i in ready if avail(i) <= t]
        if ok:
            # an operation whose LDS quad is still in the ring (its partner word was consumed a moment ago) first
            best = max(prio[i] for i in ok)
            near = [i for i in ok if quad_of[i] in recent and prio[i] >= best - SCHED_SLACK * lat]
            pick = min(near) if near else max(ok, key=lambda i: (prio[i], -i))
        else:
            pick = min(ready, key=lambda i: (avail(i), -prio[i], i))
        ready.remove(pick)
        if quad_of[pick] is not None:
            if quad_of[pick] in recent:
                recent.remove(quad_of[pick])
            recent.append(quad_of[pick])
            del recent[:-(NSLOT - 3)]
        issued_at[pick] = t
        out.append(ops[pick])
        t += 1
        for j in users[pick]:
            remaining[j] -= 1
            if remaining[j] == 0:
                ready.append(j)
    assert len(out) == n
    return out


def _words(lo, hi):
    """LDS quads covering words [lo, hi): list of (quad, [words of the range in it])"""
    out = []
    for qd in range(lo >> 1, (hi + 1) >> 1):
        out.append((qd, [w for w in (2 * qd, 2 * qd + 1) if lo <= w < hi]))
    return out


def _read_group(e, quads):
    """issues the reads of up to NSLOT quads into ring slots 0.., returns {word: first VGPR}"""
    where = {}
    for k, (qd, ws) in enumerate(quads):
        base, off, _ = lds_addr(2 * qd)
        r = V_RING + 4 * k
        e("ds_read_b128", "v[%d:%d]" % (r, r + 3), base, off)
        for w in ws:
            where[w] = r + 2 * (w & 1)
    return where


def _write_quad(e, qd, ws, reg_of):
    """writes the words ws of quad qd back from registers reg_of[w] (adjacent registers when both halves are written);
    returns the number of LDS instructions issued"""
    base, off, _ = lds_addr(2 * qd)
    if len(ws) == 2 and reg_of[ws[1]] == reg_of[ws[0]] + 2:
        e("ds_write_b128", base, "v[%d:%d]" % (reg_of[ws[0]], reg_of[ws[0]] + 3), off)
        return 1
    for w in ws:
        e("ds_write_b64", base, vp(reg_of[w]), off + 8 * (w & 1))
    return len(ws)


def consts(e):
    for reg, val in ((S_ALPHA, 1.6), (S_OMA, 1.0 - 1.6), (S_SIGMA, 1e-6), (S_RINV, 1.0 / 100.0), (S_RHO, 100.0)):
        b = f64bits(val)
        e("s_mov_b32", "s%d" % reg, b & 0xFFFFFFFF)
        e("s_mov_b32", "s%d" % (reg + 1), b >> 32)


def preload_q(e, s):
    """W_x <- q (rows FAC_Q..), in place operands of the next rhs"""
    _row_ptr(e, S_P, S_WS, FAC_Q)
    for j in range(s.nx):
        e("global_load_dwordx2", vp(V_W + 2 * j), "v0", sp(S_P))
        _adv(e, S_P)


def preload_l(e, s, neq):
    _row_ptr(e, S_P, S_WS, FAC_LOEQ)
    for i in range(neq):
        e("global_load_dwordx2", vp(V_W + 2 * (s.nx + i)), "v0", sp(S_P))
        _adv(e, S_P)


def prologue(e, s):
    nx, nc, nk = s.nx, s.nc, s.nk
    neq = 2 * s.N * symbolic.NY
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")     # the C++ side's hand-off stores have left the wave
    consts(e)
    e("v_add_u32", "v%d" % V_B1, 0x10000, "v1")
    e("v_add_u32", "v%d" % V_B2, 0x20000, "v1")
    # 1/D: phase A left it in LDS words 213.. (permuted order) -> AGPRs; then x, y take those words
    quads = _words(NNZL, NNZL + nk)
    for g in range(0, len(quads), NSLOT):
        grp = quads[g:g + NSLOT]
        where = _read_group(e, grp)
        e("s_waitcnt", "lgkmcnt(0)")
        for w, r in sorted(where.items()):
            e("v_accvgpr_write_b32", "a%d" % (A_D + 2 * (w - NNZL)), "v%d" % r)
            e("v_accvgpr_write_b32", "a%d" % (A_D + 2 * (w - NNZL) + 1), "v%d" % (r + 1))
    # x, y of the warm start: ctrl rows 0..83 -> W registers (landing zone) -> LDS
    e("s_mov_b64", sp(S_P), sp(S_CTRL))
    for o in range(nx + nc):
        e("global_load_dwordx2", vp(V_W + 2 * o), "v0", sp(S_P))
        _adv(e, S_P)
    # z of the dynamics rows (first iteration) -> AGPRs; z of the thrust rows -> registers
    for i in range(nc):
        dst = "a[%d:%d]" % (A_Z + 2 * i, A_Z + 2 * i + 1) if i < neq else vp(V_C + 2 * (i - neq))
        e("global_load_dwordx2", dst, "v0", sp(S_P))
        _adv(e, S_P)
    # thrust-row words lo3 up3 rho3 rinv3 (rows FAC_M..)
    _row_ptr(e, S_P, S_WS, FAC_M)
    for k in range(4 * s.N):
        e("global_load_dwordx2", vp(V_C + 2 * s.N + 2 * k), "v0", sp(S_P))
        _adv(e, S_P)
    e("s_waitcnt", "vmcnt(0)")
    reg_of = {LW_X + o: V_W + 2 * o for o in range(nx + nc)}
    for qd, ws in _words(LW_X, LW_X + nx + nc):
        _write_quad(e, qd, ws, reg_of)
    preload_q(e, s)


def body(e, s, first, capture, handoff=True):
    """handoff=False (the quad form, asmquad64.py): the first iteration does not form the next iteration's right-hand side"""
    nx, nc, nk = s.nx, s.nc, s.nk
    N = s.N
    neq = 2 * N * symbolic.NY
    W = lambda o: V_W + 2 * o
    WK = lambda k: W(s.perm[k])
    sA, sO, sS, sRi, sRh = (sp(r) for r in (S_ALPHA, S_OMA, S_SIGMA, S_RINV, S_RHO))
    Z3 = lambda k: V_C + 2 * k
    LO3 = lambda k: V_C + 2 * N + 2 * k
    UP3 = lambda k: V_C + 4 * N + 2 * k
    RHO3 = lambda k: V_C + 6 * N + 2 * k
    RINV3 = lambda k: V_C + 8 * N + 2 * k
    ptr = sp(S_P2)
    # q (and, unless first, l of the dynamics rows) are on their way into the W registers: loads issued in the order q_0..q_44,
    # l_0..l_35 behind the previous iteration's updates (or by the prologue), no other VMEM operation since. Each rhs
    # operation waits for ITS word only (loads return in order), so the tail of the loads hides behind the head of the rhs.
    npre = nx if first else nx + neq
    waited = [-1]

    def wait_pre(pos):
        if pos > waited[0]:
            e("s_waitcnt", "vmcnt(%d)" % min(63, npre - 1 - pos))
            waited[0] = pos
    ops = []

    def op(srcs, fn, wr, rd=()):
        """wr / rd: the W word (original index) the operation updates / the other W words it reads"""
        ops.append(dict(srcs=srcs, emit=fn, wr=wr, rd=tuple(rd)))
    # ---- rhs: W = [sigma x - q ; z - y / rho]  (auxil.c:164-178). First iteration (and HOMES off): q / l already in W;
    # later iterations with HOMES: q, l from their AGPR homes, structural zeros dropped. FUSE: the middle iterations have no
    # rhs phase at all -- the x and y updates of the iteration before leave the next right-hand side in the W registers (the
    # updated word is in a register there: no second LDS read of x and y; same operations on the same values)
    hm = homes(s) if HOMES else None
    fuse = HOMES and FUSE
    fused_here = fuse and not first and not capture          # this body's updates form the next rhs

    def add_rhs(first_style):
        use_h = HOMES and not first_style
        for j in range(nx):
            if not use_h:
                op([("L", LW_X + j)], lambda r, j=j: (wait_pre(j), e("v_fma_f64", vp(W(j)), sS, vp(r[0]), "-" + vp(W(j)))), j)
            elif ("q", j) in hm:
                op([("L", LW_X + j), ("A", hm[("q", j)])], lambda r, j=j: e("v_fma_f64", vp(W(j)), sS, vp(r[0]), "-" + vp(r[1])), j)
            else:
                op([("L", LW_X + j)], lambda r, j=j: e("v_mul_f64", vp(W(j)), sS, vp(r[0])), j)
        for i in range(neq):
            if first_style:
                op([("L", LW_Y + i), ("A", A_Z + 2 * i)],
                   lambda r, i=i: e("v_fma_f64", vp(W(nx + i)), "-" + vp(r[0]), sRi, vp(r[1])), nx + i)
            elif not use_h:
                op([("L", LW_Y + i)], lambda r, i=i: (wait_pre(nx + i),
                                                      e("v_fma_f64", vp(W(nx + i)), "-" + vp(r[0]), sRi, vp(W(nx + i)))), nx + i)
            elif ("l", i) in hm:
                op([("L", LW_Y + i), ("A", hm[("l", i)])],
                   lambda r, i=i: e("v_fma_f64", vp(W(nx + i)), "-" + vp(r[0]), sRi, vp(r[1])), nx + i)
            else:
                op([("L", LW_Y + i)], lambda r, i=i: e("v_mul_f64", vp(W(nx + i)), "-" + vp(r[0]), sRi), nx + i)
        for k in range(N):
            i = neq + k
            op([("L", LW_Y + i)], lambda r, i=i, k=k: e("v_fma_f64", vp(W(nx + i)), "-" + vp(RINV3(k)), vp(r[0]), vp(Z3(k))), nx + i)
    if first or not fuse:
        add_rhs(first)
    # ---- forward solve (qdldl.c:250-262), columns ascending, entries ascending: W[r] -= L_j W[c]
    for c in range(nk):
        for j in range(s.L_p[c], s.L_p[c + 1]):
            r_ = s.L_i[j]
            op([("L", j)], lambda r, r_=r_, c=c: e("v_fma_f64", vp(WK(r_)), "-" + vp(r[0]), vp(WK(c)), vp(WK(r_))),
               s.perm[r_], [s.perm[c]])
    # ---- diagonal (qdldl.c:289)
    for k in range(nk):
        op([("A", A_D + 2 * k)], lambda r, k=k: e("v_mul_f64", vp(WK(k)), vp(WK(k)), vp(r[0])), s.perm[k])
    # ---- backward solve (qdldl.c:264-277), the storage walked backwards: W[c] -= L_j W[r]
    for c in range(nk - 1, -1, -1):
        for j in range(s.L_p[c + 1] - 1, s.L_p[c] - 1, -1):
            r_ = s.L_i[j]
            op([("L", j)], lambda r, r_=r_, c=c: e("v_fma_f64", vp(WK(c)), "-" + vp(r[0]), vp(WK(r_)), vp(WK(c))),
               s.perm[c], [s.perm[r_]])
    Fetch(e).run(list_schedule(ops))
    # ---- x <- alpha x~ + (1 - alpha) x   (auxil.c:188-201); x_prev of a capturing iteration -> workspace
    def lds_store(word, reg):
        """one word of a capture -> LDS (the capturing iteration is the last one: the L words are dead)"""
        base, off, half = lds_addr(word)
        e("ds_write_b64", base, vp(reg), off + 8 * half)
        return 1
    quads = _words(LW_X, LW_X + nx)
    if not first and not HOMES:
        _row_ptr(e, S_P, S_WS, FAC_Q)      # the next iteration's q follows the update into each group of W_x registers
    for g in range(0, len(quads), NSLOT):
        grp = quads[g:g + NSLOT]
        where = _read_group(e, grp)
        nw = 0      # LDS writes issued behind the group's reads (LDS operations complete in order)
        for k, (qd, ws) in enumerate(grp):
            e("s_waitcnt", "lgkmcnt(%d)" % min(15, len(grp) - 1 - k + nw))
            for w in ws:
                j, r, t = w - LW_X, where[w], V_TT + 2 * ((w - LW_X) % N_TT)
                if capture:
                    nw += lds_store(PC_XP + j, r)
                e("v_mul_f64", vp(t), sO, vp(r))
                e("v_fma_f64", vp(t if capture else r), sA, vp(W(j)), vp(t))
                if fused_here:          # the next iteration's W_x = sigma x_new - q
                    if ("q", j) in hm:
                        ah = V_AT + 2 * (j % N_AT)
                        e("v_accvgpr_read_b32", "v%d" % ah, "a%d" % hm[("q", j)])
                        e("v_accvgpr_read_b32", "v%d" % (ah + 1), "a%d" % (hm[("q", j)] + 1))
                        e("v_fma_f64", vp(W(j)), sS, vp(r), "-" + vp(ah))
                    else:
                        e("v_mul_f64", vp(W(j)), sS, vp(r))
            nw += _write_quad(e, qd, ws, {w: (V_TT + 2 * ((w - LW_X) % N_TT) if capture else where[w]) for w in ws})
        if not first and not HOMES:
            for qd, ws in grp:
                for w in ws:
                    e("global_load_dwordx2", vp(W(w - LW_X)), "v0", sp(S_P))
                    _adv(e, S_P)
    lrows = [i for i in range(neq) if not HOMES or ("l", i) in hm]
    if first:
        # l of the dynamics rows (the new z there) -> the W_x registers the x update has just freed, one round trip for
        # all rows (HOMES: for the 16 that can be non-zero); q follows after the row update
        for i in lrows:
            _row_ptr(e, S_P, S_WS, FAC_LOEQ + i)
            e("global_load_dwordx2", vp(W(i)), "v0", sp(S_P))
        e("s_waitcnt", "vmcnt(0)")
    # ---- z, y  (auxil.c:203-228, qdldl_interface.c:364-366, proj.c:4-14)
    quads = _words(LW_Y, LW_Y + nc)
    if not first and not HOMES:
        _row_ptr(e, S_P, S_WS, FAC_LOEQ)   # ... and l into each group of W_z registers
    for g in range(0, len(quads), NSLOT):
        grp = quads[g:g + NSLOT]
        where = _read_group(e, grp)
        nw = 0
        for k, (qd, ws) in enumerate(grp):
            e("s_waitcnt", "lgkmcnt(%d)" % min(15, len(grp) - 1 - k + nw))
            newreg = {}
            for w in ws:
                i, r = w - LW_Y, where[w]
                nu = W(nx + i)
                t1, t2, t3 = V_TT + 2 * ((2 * i) % N_TT), V_TT + 2 * ((2 * i + 1) % N_TT), V_AT + 2 * (i % 2)
                if i < neq and not first:
                    # dynamics rows, z == l == u: delta_y = alpha (nu - y)
                    e("v_add_f64", vp(t1), vp(nu), "-" + vp(r))
                    if capture:
                        e("v_mul_f64", vp(t2), sA, vp(t1))
                        nw += lds_store(PC_DY + i, t2)
                    e("v_fma_f64", vp(r), sA, vp(t1), vp(r))
                    if fused_here:      # the next iteration's W_z = l - y_new / rho
                        if ("l", i) in hm:
                            ah = V_AT + 2 * (i % N_AT)
                            e("v_accvgpr_read_b32", "v%d" % ah, "a%d" % hm[("l", i)])
                            e("v_accvgpr_read_b32", "v%d" % (ah + 1), "a%d" % (hm[("l", i)] + 1))
                            e("v_fma_f64", vp(nu), "-" + vp(r), sRi, vp(ah))
                        else:
                            e("v_mul_f64", vp(nu), "-" + vp(r), sRi)
                    newreg[w] = r
                    continue
                if i < neq:     # first iteration, dynamics row: z_prev from its AGPR, l (= new z) from W_x[i]
                    zr, lr = V_AT + 4, W(i)
                    e("v_accvgpr_read_b32", "v%d" % zr, "a%d" % (A_Z + 2 * i))
                    e("v_accvgpr_read_b32", "v%d" % (zr + 1), "a%d" % (A_Z + 2 * i + 1))
                    rinv, rho = sRi, sRh
                else:
                    k3 = i - neq
                    zr, rinv, rho = Z3(k3), vp(RINV3(k3)), vp(RHO3(k3))
                e("v_fma_f64", vp(t1), "-" + vp(r), rinv, vp(zr))              # z - y/rho (the rhs again)
                e("v_fma_f64", vp(t1), vp(nu), rinv, vp(t1))                   # z~
                e("v_mul_f64", vp(t2), sO, vp(zr))
                e("v_fma_f64", vp(t1), sA, vp(t1), vp(t2))                     # t = alpha z~ + (1-alpha) z
                if i < neq and i in lrows:
                    e("v_add_f64", vp(t2), vp(t1), "-" + vp(lr))               # z <- l
                    e("v_mul_f64", vp(t2), vp(t2), rho)
                elif i < neq:
                    e("v_mul_f64", vp(t2), vp(t1), rho)                        # z <- l == 0
                else:
                    e("v_fma_f64", vp(t3), vp(r), rinv, vp(t1))
                    e("v_max_f64", vp(t3), vp(t3), vp(LO3(k3)))
                    e("v_min_f64", vp(zr), vp(t3), vp(UP3(k3)))
                    e("v_add_f64", vp(t2), vp(t1), "-" + vp(zr))
                    e("v_mul_f64", vp(t2), vp(t2), rho)                        # delta_y
                if capture:
                    nw += lds_store(PC_DY + i, t2)
                e("v_add_f64", vp(r), vp(r), vp(t2))
                if fused_here and i >= neq:     # thrust rows: the next iteration's W_z = z_new - y_new / rho
                    e("v_fma_f64", vp(nu), "-" + rinv, vp(r), vp(zr))
                newreg[w] = r
            nw += _write_quad(e, qd, ws, newreg)
        if not first and not HOMES:
            for qd, ws in grp:
                for w in ws:
                    if w - LW_Y < neq:
                        e("global_load_dwordx2", vp(W(nx + w - LW_Y)), "v0", sp(S_P))
                        _adv(e, S_P)
    if first and not HOMES:
        preload_q(e, s)
        preload_l(e, s, neq)
    elif first:
        # the z words of the first iteration (a168..) are dead: l (still in the W_x registers) and q take their homes there
        for i in lrows:
            e("v_accvgpr_write_b32", "a%d" % hm[("l", i)], "v%d" % W(i))
            e("v_accvgpr_write_b32", "a%d" % (hm[("l", i)] + 1), "v%d" % (W(i) + 1))
        for (kind, j), a_ in sorted(hm.items(), key=lambda kv: kv[1]):
            if kind == "q":
                _row_ptr(e, S_P, S_WS, FAC_Q + j)
                e("global_load_dwordx2", "a[%d:%d]" % (a_, a_ + 1), "v0", sp(S_P))
        e("s_waitcnt", "vmcnt(0)")
        if fuse and not capture and handoff:
            # FUSE: the first iteration hands the second its right-hand side the classic way (x, y from LDS, q and l from the
            # homes just filled); from then on every iteration's updates do it for the next one
            ops = []
            add_rhs(False)
            Fetch(e).run(ops)


def epilogue(e, s):
    """x, y, z -> ctrl rows (warm start of the next step); everything phase C reads -> LDS words PC_* (z; the thrust-row
    bounds; D and E of phase A, fetched here in ONE round trip -- hipcc fetches them one exposed load at a time)."""
    nx, nc = s.nx, s.nc
    N = s.N
    neq = 2 * N * symbolic.NY
    e("s_waitcnt", "vmcnt(0)")      # HOMES off: the last preloads, W_z of the dynamics rows holds l (= z there)
    if HOMES:
        hm = homes(s)
        for i in range(neq):
            r = V_W + 2 * (nx + i)
            if ("l", i) in hm:
                e("v_accvgpr_read_b32", "v%d" % r, "a%d" % hm[("l", i)])
                e("v_accvgpr_read_b32", "v%d" % (r + 1), "a%d" % (hm[("l", i)] + 1))
            else:
                e("v_mov_b32", "v%d" % r, 0)
                e("v_mov_b32", "v%d" % (r + 1), 0)
    e("s_mov_b64", sp(S_P), sp(S_CTRL))
    quads = _words(LW_X, LW_X + nx + nc)
    for g in range(0, len(quads), NSLOT):
        grp = quads[g:g + NSLOT]
        where = _read_group(e, grp)
        e("s_waitcnt", "lgkmcnt(0)")
        for qd, ws in grp:
            for w in ws:
                e("global_store_dwordx2", "v0", vp(where[w]), sp(S_P))
                _adv(e, S_P)
    zreg = lambda i: V_W + 2 * (nx + i) if i < neq else V_C + 2 * (i - neq)
    for i in range(nc):
        e("global_store_dwordx2", "v0", vp(zreg(i)), sp(S_P))
        _adv(e, S_P)
    for qd, ws in _words(PC_Z, PC_Z + nc):
        _write_quad(e, qd, ws, {w: zreg(w - PC_Z) for w in ws})
    for qd, ws in _words(PC_LO3, PC_LO3 + 2 * N):      # lo3 up3 are adjacent register pairs (V_C + 2N ..)
        _write_quad(e, qd, ws, {w: V_C + 2 * N + 2 * (w - PC_LO3) for w in ws})
    # D (45) -> the W_x registers, E (39) -> the W_z registers (all dead now), then LDS
    _row_ptr(e, S_P, S_WS, WS_DS)
    for o in range(nx + nc):
        e("global_load_dwordx2", vp(V_W + 2 * o), "v0", sp(S_P))
        _adv(e, S_P)
    e("s_waitcnt", "vmcnt(0)")
    for qd, ws in _words(PC_DS, PC_DS + nx + nc):
        _write_quad(e, qd, ws, {w: V_W + 2 * (w - PC_DS) for w in ws})
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")


def program(N=3, perm=None, quad=False):
    """maxIter >= 1 (the caller takes the C++ loop otherwise). quad: iterations 2.. on the lane quad (asmquad64.py)"""
    s = symbolic.analyse(N, perm)
    e = Emit()
    timing = os.environ.get("UMPC_ASM64_TIMING") == "1"      # diagnostic builds: 100 MHz stamps -> spare LDS words 297..

    def stamp(k):
        if timing:
            e("s_memrealtime", sp(36 + 2 * k))
    stamp(0)
    prologue(e, s)
    stamp(1)
    # the captures overwrite L in LDS, so only the LAST iteration captures: a single iteration is its own variant
    e("s_cmp_lt_i32", "s%d" % S_ITERS, 2)
    e("s_cbranch_scc1", "4f")
    if quad:
        from . import asmquad64
        body(e, s, first=True, capture=False, handoff=False)
        stamp(2)
        asmquad64.section(e, asmquad64.plan_for(s), s)
        stamp(3)
        e("s_branch", "6f")
    else:
        body(e, s, first=True, capture=False)
    stamp(2)
    e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_ITERS, 2)
    e("s_cmp_lt_i32", "s%d" % S_CNT, 1)
    e("s_cbranch_scc1", "8f")
    e("label", "7")
    body(e, s, first=False, capture=False)
    e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_CNT, 1)
    e("s_cmp_gt_i32", "s%d" % S_CNT, 0)
    e("s_cbranch_scc1", "7b")
    e("label", "8")
    stamp(3)
    body(e, s, first=False, capture=True)
    e("s_branch", "6f")
    e("label", "4")
    body(e, s, first=True, capture=True)
    e("label", "6")
    stamp(4)
    epilogue(e, s)
    if timing:
        stamp(5)
        e("s_waitcnt", "lgkmcnt(0)")
        for k in range(6):
            e("v_mov_b32", "v%d" % V_TT, "s%d" % (36 + 2 * k))
            e("v_mov_b32", "v%d" % (V_TT + 1), "s%d" % (37 + 2 * k))
            base, off, half = lds_addr(LW_END + k)
            e("ds_write_b64", base, vp(V_TT), off + 8 * half)
        e("s_waitcnt", "lgkmcnt(0)")
    return e.ins, s


def fmt(t):
    m = t[0]
    if m == "label":
        return "%s:" % t[1]
    a = [("0x%x" % x if isinstance(x, int) and m in ("s_mov_b32", "v_add_u32", "v_mov_b32", "v_cndmask_b32") else str(x)) for x in t[1:]]
    if m.startswith("ds_read") or m.startswith("ds_write"):
        return "%s %s, %s offset:%s" % (m, a[0], a[1], a[2])
    if m == "s_waitcnt":
        return "s_waitcnt " + " ".join(a)
    if m.endswith("_dpp") or (isinstance(t[-1], str) and t[-1].startswith("offset:")):     # trailing control / offset: no comma
        return "%s %s %s" % (m, ", ".join(a[:-1]), t[-1])
    return "%s %s" % (m, ", ".join(a))


PSEUDO = ("quad_begin", "quad_end")


def used_registers(ins):
    """(AGPR numbers, VGPR numbers) that appear in ANY operand of the instruction list: an over-approximation of what an
    `asm volatile` block of it may write, for exact clobber lists"""
    import re
    A, V = set(), set()
    for t in ins:
        for x in t[1:]:
            if not isinstance(x, str):
                continue
            for m_ in re.finditer(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b", x):
                if m_.group(1):
                    (A if m_.group(1) == "a" else V).update(range(int(m_.group(2)), int(m_.group(3)) + 1))
                else:
                    (A if m_.group(4) == "a" else V).add(int(m_.group(5)))
    return A, V


def write_quad(path=None, N=3, perm=None):
    """csrc/umpc_admm_asm64_quad.h: the ADMM phase with one robot per lane quad (asmquad64.py) as UMPC_ADMM_ASM64_QUAD, and
    csrc/umpc_quad64_tab.h, the constant table of per-lane coefficient addresses it reads through s[8:9]."""
    from . import asmquad64
    path = path or os.path.join(HERE, "csrc", "umpc_admm_asm64_quad.h")
    ins, s = program(N, perm, quad=True)
    ins = [t for t in ins if t[0] not in PSEUDO]
    used_s = [S_P, S_P + 1, S_CNT, S_P2, S_P2 + 1] + list(range(S_ALPHA, S_RHO + 2)) + list(range(30, 38))
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in range(2, V_END)] + \
           ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in used_s]
    l17 = [k for k, t in enumerate(ins) if t == ("label", "17")][0]
    l18 = [k for k, t in enumerate(ins) if t == ("label", "18")][0]
    out = ["// GENERATED by robobee3d_amd/asmgen64.py (quad=True) + asmquad64.py -- do not edit.", asmgen.switch_banner(),
           "// ADMM phase of the fp64 small-batch step kernel, ONE ROBOT PER LANE QUAD: %d instructions, middle-iteration body %d."
           % (len(ins), l18 - l17),
           "#pragma once",
           "// inputs: v0 = 8*robot (the same in the four lanes of a quad), v1 = lane LDS address (16*lane), s[4:5] = workspace,",
           "// s[6:7] = ctrl, s[8:9] = umpcquad64::kTab, s10 = 8*B, s11 = maxIter >= 1",
           "#define UMPC_ADMM_ASM64_QUAD(voff, ldsaddr, ws, ctrl, tab, stride, iters) asm volatile( \\"]
    for t in ins:
        out.append('  "%s\\n" \\' % fmt(t))
    out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{s[4:5]}"(ws), "{s[6:7]}"(ctrl), "{s[8:9]}"(tab), "{s10}"(stride), "{s11}"(iters) \\')
    out.append("  : " + ", ".join(clob) + ")")
    # the Ruiz passes on the quad
    rins, _ = asmquad64.ruiz_program(N, perm)
    rins = [t for t in rins if t[0] not in PSEUDO]
    r27 = [k for k, t_ in enumerate(rins) if t_ == ("label", "27")][0]
    rend = [k for k, t_ in enumerate(rins) if t_[0] == "s_cbranch_scc1"][-1]
    # exact VGPR clobbers (round 5: the quad passes touch v2..v187 and no AGPR; what lives across them stays in registers)
    rclob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in sorted(used_registers(rins)[1] - {0, 1})] + \
            ['"s%d"' % i for i in [S_CNT] + list(range(30, 42))]
    assert not used_registers(rins)[0]
    out += ["// The Ruiz passes of the fp64 step on the lane quad (asmquad64.ruiz_program): %d instructions, %d per pass (one-lane"
            % (len(rins), rend - r27),
            "// block: 2 546 per pass). LDS words in and out as UMPC_RUIZ_ASM64; every lane's slice ends with the whole result.",
            "// inputs: v1 = lane LDS address (16*lane), s11 = passes >= 1",
            "#define UMPC_RUIZ_ASM64_QUAD(ldsaddr, passes) asm volatile( \\"]
    for t_ in rins:
        out.append('  "%s\\n" \\' % fmt(t_))
    out.append('  : : "{v1}"(ldsaddr), "{s11}"(passes) \\')
    out.append("  : " + ", ".join(rclob) + ")")
    txt = "\n".join(out) + "\n"
    if not os.path.exists(path) or open(path).read() != txt:
        with open(path, "w") as fh:
            fh.write(txt)
    tab = asmquad64.table(asmquad64.plan_for(s))
    tpath = os.path.join(os.path.dirname(path), "umpc_quad64_tab.h")
    ttxt = "// GENERATED by robobee3d_amd/asmgen64.py (write_quad) -- do not edit.\n#pragma once\n#include <stdint.h>\n" \
           "// byte offset, from the lane's LDS base, of the entry of L that lane class (lane & 3) multiplies with in solve\n" \
           "// instruction q of the quad loop (asmquad64.table); the zero word where it has none\n" \
           "namespace umpcquad64 {\n__device__ const uint32_t kTab[4][%d] = {\n%s\n};\n}  // namespace umpcquad64\n" % (
               tab.shape[1], ",\n".join("  {" + ", ".join(str(int(x)) for x in row) + "}" for row in tab))
    if not os.path.exists(tpath) or open(tpath).read() != ttxt:
        with open(tpath, "w") as fh:
            fh.write(ttxt)
    return path, len(ins), l18 - l17


def write(path=None, N=3, perm=None):
    path = path or os.path.join(HERE, "csrc", "umpc_admm_asm64.h")
    ins, s = program(N, perm)
    used_s = [S_P, S_P + 1, S_CNT, S_P2, S_P2 + 1] + list(range(S_ALPHA, S_RHO + 2)) + \
             (list(range(36, 48)) if os.environ.get("UMPC_ASM64_TIMING") == "1" else [])
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in range(2, V_END)] + \
           ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in used_s]
    lab7 = [k for k, t in enumerate(ins) if t == ("label", "7")][0]
    lab8 = [k for k, t in enumerate(ins) if t == ("label", "8")][0]
    out = ["// GENERATED by robobee3d_amd/asmgen64.py -- do not edit.", asmgen.switch_banner(),
           "// ADMM phase of the fp64 small-batch step kernel: %d instructions, middle-iteration body %d." % (len(ins), lab8 - lab7),
           "#pragma once",
           "namespace umpcasm64 {",
           "constexpr int LDS_BYTES_PER_LANE = %d, LW_X = %d, LW_Y = %d;" % (LDS_BYTES_PER_LANE, LW_X, LW_Y),
           "// LDS words phase C reads after the loop: D, E of phase A, z, x_prev and delta_y of the last iteration, lo3, up3",
           "constexpr int PC_DS = %d, PC_ES = %d, PC_Z = %d, PC_XP = %d, PC_DY = %d, PC_LO3 = %d, PC_UP3 = %d;"
           % (PC_DS, PC_ES, PC_Z, PC_XP, PC_DY, PC_LO3, PC_UP3),
           "}  // namespace umpcasm64",
           "// inputs: v0 = 8*robot, v1 = lane LDS address (16*lane), s[4:5] = workspace, s[6:7] = ctrl, s10 = 8*B, s11 = maxIter >= 1",
           "#define UMPC_ADMM_ASM64(voff, ldsaddr, ws, ctrl, stride, iters) asm volatile( \\"]
    for t in ins:
        out.append('  "%s\\n" \\' % fmt(t))
    out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{s[4:5]}"(ws), "{s[6:7]}"(ctrl), "{s10}"(stride), "{s11}"(iters) \\')
    out.append("  : " + ", ".join(clob) + ")")
    # the Ruiz passes
    rins, _ = ruiz_program(N, perm)
    rclob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in range(2, V_END)] + ['"a%d"' % i for i in range(256)] + \
            ['"s%d"' % i for i in [S_CNT, S_MINS, S_MINS + 1, S_MAXS, S_MAXS + 1]]
    rl7 = [k for k, t_ in enumerate(rins) if t_ == ("label", "7")][0]
    out += ["// The Ruiz passes of the fp64 step (scaling.c:44-156): %d instructions, %d per pass. LDS words (LDSF_W layout):"
            % (len(rins), sum(1 for t_ in rins[rl7:] if t_[0] in ("s_cbranch_scc1",)) and
               [k for k, t_ in enumerate(rins) if t_[0] == "s_cbranch_scc1"][0] - rl7),
            "// P at RZ_P.., q at RZ_Q.., A at RZ_A.. on entry and exit, the accumulated cost scaling c at RZ_C on exit.",
            "namespace umpcasm64 { constexpr int RZ_P = %d, RZ_C = %d, RZ_Q = %d, RZ_A = %d; }" % (RZ_P, RZ_C, RZ_Q, RZ_A),
            "// inputs: v1 = lane LDS address (16*lane), s11 = passes >= 1",
            "#define UMPC_RUIZ_ASM64(ldsaddr, passes) asm volatile( \\"]
    for t_ in rins:
        out.append('  "%s\\n" \\' % fmt(t_))
    out.append('  : : "{v1}"(ldsaddr), "{s11}"(passes) \\')
    out.append("  : " + ", ".join(rclob) + ")")
    # the residual norms
    sins, _ = resid_program(N, perm)
    # exact AGPR clobbers (round 5): the block touches 70 AGPRs; the compiler may keep what lives across it (state, weights,
    # row pointers) in the other 186 -- v_accvgpr_write / read instead of scratch stores / loads around the block
    sclob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in sorted(used_registers(sins)[1] - {0, 1})] + ['"a%d"' % i for i in sorted(used_registers(sins)[0])] + \
            ['"s%d"' % i for i in [S_P, S_P + 1]]
    out += ["// The residual norms of the fp64 step (auxil.c:243-307): %d instructions. LDS words in: x, y, z, D, E where the ADMM"
            % len(sins),
            "// block left them, RS_PAR.. (T0 dt, s0 dt[3], Btau dt[6]), RS_C c, RS_W.. the eight weights, RS_DT dt; out: RS_OUT..",
            "// = pri_res, dua_res before the division by c, |z|, |Ax|, |q|, |A'y|, |Px| norms, NaN accumulator.",
            "namespace umpcasm64 { constexpr int RS_PAR = %d, RS_C = %d, RS_W = %d, RS_DT = %d, RS_OUT = %d; }"
            % (RS_PAR, RS_C, RS_W, RS_DT, RS_OUT),
            "// inputs: v0 = 8*robot, v1 = lane LDS address (16*lane), s[4:5] = workspace, s10 = 8*B",
            "#define UMPC_RESID_ASM64(voff, ldsaddr, ws, stride) asm volatile( \\"]
    for t_ in sins:
        out.append('  "%s\\n" \\' % fmt(t_))
    out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{s[4:5]}"(ws), "{s10}"(stride) \\')
    out.append("  : " + ", ".join(sclob) + ")")
    txt = "\n".join(out) + "\n"
    old = open(path).read() if os.path.exists(path) else None
    if old != txt:
        with open(path, "w") as fh:
            fh.write(txt)
    return path, len(ins), lab8 - lab7


# ---------------------------------------------------------------------------
# CPU interpreter (one lane), exact-rounded float64
# ---------------------------------------------------------------------------
def simulate(ins, mem_ws, mem_ctrl, iters, lds, perm=None):
    """mem_ws: float64[WS_ROWS], mem_ctrl: float64[123], lds: float64[320] (L in words 0..212, 1/D in 213..296 on entry).
    Updates the memories in place; returns the executed instruction count. perm: the KKT permutation of the program (only
    the quad form needs it, for its plan)."""
    quad_perm_ = perm
    import numpy as np
    from fractions import Fraction
    V = np.zeros(256, np.uint32)
    A = np.zeros(256, np.uint32)
    S = {}
    scc = 0
    labels = {}
    for k, t in enumerate(ins):
        if t[0] == "label":
            labels.setdefault(t[1], []).append(k)
    STRIDE = 4096
    S[S_WS], S[S_WS + 1] = 1 << 20, 0
    S[S_CTRL], S[S_CTRL + 1] = 1 << 30, 0
    S[S_STRIDE], S[S_ITERS] = STRIDE, iters
    V[1] = 0

    def lohi(x):
        return int(x[2:x.index(":")])

    def sval(x):
        if isinstance(x, int):
            return x
        if x.startswith("s["):
            lo = lohi(x)
            return S.get(lo, 0) | (S.get(lo + 1, 0) << 32)
        return S.get(int(x[1:]), 0)

    def getd(x):
        if isinstance(x, float):
            return x                      # inline constant (1.0, 0.5)
        neg = x.startswith("-")
        if neg:
            x = x[1:]
        if x.startswith("|"):
            return (-1.0 if neg else 1.0) * abs(getd(x[1:-1]))
        lo = lohi(x)
        bits = (int(V[lo]) | (int(V[lo + 1]) << 32)) if x[0] == "v" else (S[lo] | (S[lo + 1] << 32))
        val = struct.unpack("<d", struct.pack("<Q", bits))[0]
        return -val if neg else val

    def setd(x, val, file=None):
        file = V if file is None else file
        lo = lohi(x)
        b = f64bits(float(val))
        file[lo], file[lo + 1] = b & 0xFFFFFFFF, b >> 32

    def fma(a, b, c):
        if not (np.isfinite(a) and np.isfinite(b) and np.isfinite(c)):
            return a * b + c
        return float(Fraction(a) * Fraction(b) + Fraction(c))

    def mem(addr):
        if addr >= (1 << 30):
            return mem_ctrl, (addr - (1 << 30)) // STRIDE
        return mem_ws, (addr - (1 << 20)) // STRIDE

    def ldsword(basereg, off):
        byte = int(V[int(basereg[1:])]) + off
        return (byte // 1024) * 2 + (byte % 1024) // 8

    # completion model (as asmqp.simulate): LDS operations and VMEM loads complete in issue order; a register that an
    # outstanding load will write must not be read or written before an s_waitcnt has retired that load
    pend = {"lgkmcnt": [], "vmcnt": []}

    def regs_of(x):
        if not isinstance(x, str):
            return set()
        x = x.lstrip("-").strip("|")
        if x[:2] in ("v[", "a["):
            lo_, hi_ = x[2:-1].split(":")
            return {(x[0], r) for r in range(int(lo_), int(hi_) + 1)}
        if x[0] in "va" and x[1:].isdigit():
            return {(x[0], int(x[1:]))}
        return set()

    pc = nexec = 0
    simulate.last_quad_instructions = 0
    while pc < len(ins):
        t = ins[pc]
        m = t[0]
        if m == "quad_begin":
            # the one-robot-per-quad section (asmquad64.py): the four lanes of a quad ran everything so far redundantly, so
            # each starts from this lane's registers, AGPRs and LDS slice; afterwards the four slices and the thrust-row
            # words must agree, and the other registers are poisoned (the epilogue may only read what the exit restored)
            from . import asmquad64
            assert ins[pc + 1] == ("s_waitcnt", "vmcnt(0) lgkmcnt(0)")      # the section starts by draining the one-lane body's LDS writes
            pend["lgkmcnt"].clear(); pend["vmcnt"].clear()
            s_ = symbolic.analyse(3, quad_perm_)
            V4, A4 = np.tile(V, (4, 1)), np.tile(A, (4, 1))
            L4 = np.tile(np.asarray(lds, np.float64), (4, 1))
            pc, nq = asmquad64.simulate(ins, pc, V4, A4, L4, S, asmquad64.table(asmquad64.plan_for(s_)))
            nexec += nq
            simulate.last_quad_instructions = nq
            # (words LW_END.. are scratch: the quad loop's compact coefficient array ends there; the C++ side writes what the
            # residual block reads from them before it runs. The array's first words overlay the last words of L: whatever the
            # exit does not rewrite there is per-lane data, dead -- the epilogue writes the thrust-row bounds over it --
            # and poisoned here so that a read before that write shows up)
            differ = np.zeros(LW_END, bool)
            for ln in range(1, 4):
                differ |= ~((L4[ln, :LW_END] == L4[0, :LW_END]) | (np.isnan(L4[ln, :LW_END]) & np.isnan(L4[0, :LW_END])))
            assert set(np.nonzero(differ)[0]) <= set(range(asmquad64.CW0, NNZL)), "LDS slices of the quad disagree"
            lds[:LW_END] = L4[0, :LW_END]
            lds[:LW_END][differ] = np.nan
            lds[LW_END:] = np.nan
            keep = {0, 1, V_B1, V_B2} | set(range(V_C, V_RING))
            for r in range(256):
                if r in keep:
                    assert (V4[1:, r] == V4[0, r]).all(), r
                    V[r] = V4[0, r]
                else:
                    V[r] = 0x7ff8dead if r % 2 else 0xdeadbeef      # a NaN pattern in every pair
            A[:] = A4[0]
            A[:A_H] = 0x7ff8dead
            continue
        nexec += 1
        assert nexec < 2000000, "runaway program"
        if m == "s_waitcnt":
            for part in t[1].split():
                name, val = part[:-1].split("(")
                del pend[name][:max(0, len(pend[name]) - int(val))]
        elif m[0] == "v" or m.startswith("ds_") or m.startswith("global_"):
            used = set().union(*[regs_of(x) for x in t[1:]])
            for q_ in pend.values():
                for dst in q_:
                    assert not (dst & used), ("register used before its load was waited for", pc, t, sorted(dst & used))
            if m.startswith("ds_read"):
                pend["lgkmcnt"].append(regs_of(t[1]))
            elif m.startswith("ds_write"):
                pend["lgkmcnt"].append(set())
            elif m.startswith("global_load"):
                pend["vmcnt"].append(regs_of(t[1]))
            elif m.startswith("global_store"):
                pend["vmcnt"].append(set())
        if m in ("label", "s_waitcnt", "s_nop"):
            pass
        elif m == "v_mov_b32":
            V[int(t[1][1:])] = t[2] if isinstance(t[2], int) else V[int(t[2][1:])]
        elif m == "v_cmp_nlt_f64":
            S["vcc"] = int(not (getd(t[2]) < getd(t[3])))
        elif m == "v_cndmask_b32":
            src0 = t[2] if isinstance(t[2], int) else int(V[int(t[2][1:])])
            V[int(t[1][1:])] = int(V[int(t[3][1:])]) if S["vcc"] else src0
        elif m == "v_rsq_f64":
            setd(t[1], 1.0 / np.sqrt(getd(t[2])))
        elif m == "v_rcp_f64":
            setd(t[1], 1.0 / getd(t[2]))
        elif m == "ds_read_b64":
            lo = lohi(t[1])
            b = f64bits(float(lds[ldsword(t[2], t[3])]))
            V[lo], V[lo + 1] = b & 0xFFFFFFFF, b >> 32
        elif m == "s_mov_b32":
            S[int(t[1][1:])] = t[2] if isinstance(t[2], int) else sval(t[2])
        elif m == "s_mov_b64":
            lo = lohi(t[1])
            val = sval(t[2])
            S[lo], S[lo + 1] = val & 0xFFFFFFFF, val >> 32
        elif m == "s_mul_i32":
            S[int(t[1][1:])] = (sval(t[2]) * sval(t[3])) & 0xFFFFFFFF
        elif m == "s_mul_hi_u32":
            S[int(t[1][1:])] = ((sval(t[2]) * sval(t[3])) >> 32) & 0xFFFFFFFF
        elif m == "s_add_u32":
            r = sval(t[2]) + sval(t[3])
            S[int(t[1][1:])] = r & 0xFFFFFFFF
            scc = r >> 32
        elif m == "s_addc_u32":
            r = sval(t[2]) + sval(t[3]) + scc
            S[int(t[1][1:])] = r & 0xFFFFFFFF
            scc = r >> 32
        elif m == "s_sub_i32":
            S[int(t[1][1:])] = (sval(t[2]) - sval(t[3])) & 0xFFFFFFFF
        elif m in ("s_cmp_lt_i32", "s_cmp_gt_i32"):
            a, b = sval(t[1]), sval(t[2])
            a = a - (1 << 32) if a & 0x80000000 else a
            scc = int(a < b) if m == "s_cmp_lt_i32" else int(a > b)
        elif m in ("s_branch", "s_cbranch_scc1"):
            if m == "s_branch" or scc:
                lab, d = t[1][:-1], t[1][-1]
                cands = labels[lab]
                pc = min(c for c in cands if c > pc) if d == "f" else max(c for c in cands if c < pc)
        elif m == "v_add_u32":
            V[int(t[1][1:])] = (t[2] + int(V[int(t[3][1:])])) & 0xFFFFFFFF
        elif m == "global_load_dwordx2":
            arr, row = mem(sval(t[3]))
            setd(t[1], arr[row], A if t[1][0] == "a" else V)
        elif m == "global_store_dwordx2":
            arr, row = mem(sval(t[3]))
            arr[row] = getd(t[2])
        elif m == "ds_read_b128":
            lo = lohi(t[1])
            w = ldsword(t[2], t[3])
            for h in range(2):
                b = f64bits(float(lds[w + h]))
                V[lo + 2 * h], V[lo + 2 * h + 1] = b & 0xFFFFFFFF, b >> 32
        elif m == "ds_write_b128":
            w = ldsword(t[1], t[3])
            lo = lohi(t[2])
            for h in range(2):
                lds[w + h] = getd(vp(lo + 2 * h))
        elif m == "ds_write_b64":
            lds[ldsword(t[1], t[3])] = getd(t[2])
        elif m == "v_accvgpr_read_b32":
            V[int(t[1][1:])] = A[int(t[2][1:])]
        elif m == "v_accvgpr_write_b32":
            A[int(t[1][1:])] = V[int(t[2][1:])]
        elif m == "v_fma_f64":
            setd(t[1], fma(getd(t[2]), getd(t[3]), getd(t[4])))
        elif m == "v_mul_f64":
            setd(t[1], getd(t[2]) * getd(t[3]))
        elif m == "v_add_f64":
            setd(t[1], getd(t[2]) + getd(t[3]))
        elif m == "v_max_f64":
            setd(t[1], max(getd(t[2]), getd(t[3])))
        elif m == "v_min_f64":
            setd(t[1], min(getd(t[2]), getd(t[3])))
        else:
            raise ValueError("unknown instruction %r" % (t,))
        pc += 1
    return nexec


if __name__ == "__main__":
    print(write())


# ---------------------------------------------------------------------------
# The ten Ruiz passes of the fp64 step (scaling.c:44-156) as one assembly block
# ---------------------------------------------------------------------------
# hipcc's fp64 pass is ~3 700 instructions with ~400 exposed LDS / scratch waits: 22 us, ten times per step, 0.22 of the
# 0.53 ms (tools/f64_phases.sh). Here nothing leaves the register files for the ten passes:
#
#   v4..v225     A (CSC order), 111 words            v226:227  q[44]      v228:229  c (accumulated cost scaling)
#   v230..v243   temporaries (7 words)              v245  high word of 1.0 (limit_scaling's select)
#   a0..a77      Et (39)      a78..a167  P (45)       a168..a255  q[0..44)
#   LDS words (quad layout of LDSF_W) on entry and exit: RZ_P.. P (45), RZ_C c (exit only), RZ_Q.. q (45), RZ_A.. A (111)
#
# The column scalings Dt are never stored: a pass first forms all row scalings Et (they need every column), then walks
# the columns, forming Dt_j from the still untouched column j and applying it to P_j, column j of A and q_j at once.
# Arithmetic = UMPC_GEN_RUIZ_NORMS / APPLY_A and the loop around them in csrc/umpc_step.h, operation by operation
# (max is exact; 1/sqrt and the two divisions are v_rsq_f64 / v_rcp_f64 + two Newton steps like umpc_rsqrt / umpc_recip;
# pmean / nx is a multiplication by the refined reciprocal with one correction step: within an ulp of the divide).
RZ_P, RZ_C, RZ_Q, RZ_A = 0, 45, 84, 168
RV_A, RV_QL, RV_C, RV_T = 4, 226, 228, 230
RV_ONEHI = 245                               # high word of 1.0 (set after the prologue, which lands quads in v230..v245)
RA_ET, RA_P, RA_Q = 0, 78, 168
S_MINS, S_MAXS = 30, 32                      # 1e-4, 1e4 (doubles in SGPR pairs)
# Round 3: the cost scaling c is carried as a scalar through the passes (the norms see c * P_j, c * sum|P|, c * max|q|) and
# applied to P and q once on the way out, instead of 90 AGPR read / multiply / write triples per pass (rounding only)
CARRY_C = os.environ.get("UMPC_ASM64_RUIZ_CARRY_C", "1") == "1"


def ruiz_program(N=3, perm=None):
    """s11 = number of passes (>= 1)"""
    s = symbolic.analyse(N, perm)
    nx, nc = s.nx, s.nc
    nnz = len(s.A_i)
    assert RV_A + 2 * nnz <= RV_QL and RA_ET + 2 * nc <= RA_P and RA_P + 2 * nx <= RA_Q and RA_Q + 2 * (nx - 1) <= 256
    e = Emit()
    T = lambda q: RV_T + 2 * q
    A = lambda p_: RV_A + 2 * p_
    sMIN, sMAX = sp(S_MINS), sp(S_MAXS)
    ONE_HI = f64bits(1.0) >> 32

    def setc(reg, val):
        b = f64bits(val)
        e("s_mov_b32", "s%d" % reg, b & 0xFFFFFFFF)
        e("s_mov_b32", "s%d" % (reg + 1), b >> 32)

    def limit(t, t2):
        """t <- limit_scaling(t): t < 1e-4 ? 1 : min(t, 1e4)   (t2: scratch pair)"""
        e("v_cmp_nlt_f64", "vcc", vp(t), sMIN)
        e("v_min_f64", vp(t2), vp(t), sMAX)
        e("v_cndmask_b32", "v%d" % t, 0, "v%d" % t2, "vcc")
        e("v_cndmask_b32", "v%d" % (t + 1), "v%d" % RV_ONEHI, "v%d" % (t2 + 1), "vcc")   # (a literal next to vcc exceeds the constant bus)

    def rsqrt(y, t, a_, h_):
        """y <- 1/sqrt(t): v_rsq_f64 + two Newton steps (umpc_rsqrt)"""
        e("v_rsq_f64", vp(y), vp(t))
        e("s_nop", 0)
        for _ in range(2):
            e("v_mul_f64", vp(a_), vp(t), vp(y))
            e("v_fma_f64", vp(a_), "-" + vp(a_), vp(y), 1.0)
            e("v_mul_f64", vp(h_), 0.5, vp(y))
            e("v_fma_f64", vp(y), vp(h_), vp(a_), vp(y))

    def recip(y, t, a_):
        """y <- 1/t: v_rcp_f64 + two Newton steps (umpc_recip)"""
        e("v_rcp_f64", vp(y), vp(t))
        e("s_nop", 0)
        for _ in range(2):
            e("v_fma_f64", vp(a_), "-" + vp(t), vp(y), 1.0)
            e("v_fma_f64", vp(y), vp(y), vp(a_), vp(y))

    def acc_write(areg, v_):
        e("v_accvgpr_write_b32", "a%d" % areg, "v%d" % v_)
        e("v_accvgpr_write_b32", "a%d" % (areg + 1), "v%d" % (v_ + 1))

    def acc_read(v_, areg):
        e("v_accvgpr_read_b32", "v%d" % v_, "a%d" % areg)
        e("v_accvgpr_read_b32", "v%d" % (v_ + 1), "a%d" % (areg + 1))

    def qhome(j):
        return ("a", RA_Q + 2 * j) if j < nx - 1 else ("v", RV_QL)

    # ---- prologue: constants; A -> VGPRs, P and q -> AGPRs (through the temporaries, four quads at a time)
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    setc(S_MINS, 1e-4)
    setc(S_MAXS, 1e4)
    e("v_add_u32", "v%d" % V_B1, 0x10000, "v1")
    e("v_add_u32", "v%d" % V_B2, 0x20000, "v1")
    b = f64bits(1.0)
    e("v_mov_b32", "v%d" % RV_C, b & 0xFFFFFFFF)
    e("v_mov_b32", "v%d" % (RV_C + 1), b >> 32)
    aquads = _words(RZ_A, RZ_A + nnz)
    for qd, ws in aquads:
        base, off, _ = lds_addr(2 * qd)
        if len(ws) == 2:
            e("ds_read_b128", "v[%d:%d]" % (A(ws[0] - RZ_A), A(ws[0] - RZ_A) + 3), base, off)
        else:
            e("ds_read_b64", vp(A(ws[0] - RZ_A)), base, off + 8 * (ws[0] & 1))
    for lo, home in ((RZ_P, lambda j: ("a", RA_P + 2 * j)), (RZ_Q, qhome)):
        quads = _words(lo, lo + nx)
        for g in range(0, len(quads), 4):
            grp = quads[g:g + 4]
            for k, (qd, ws) in enumerate(grp):
                base, off, _ = lds_addr(2 * qd)
                e("ds_read_b128", "v[%d:%d]" % (T(2 * k), T(2 * k) + 3), base, off)
            e("s_waitcnt", "lgkmcnt(0)")
            for k, (qd, ws) in enumerate(grp):
                for w_ in ws:
                    kind, reg = home(w_ - lo)
                    r = T(2 * k) + 2 * (w_ & 1)
                    if kind == "a":
                        acc_write(reg, r)
                    else:
                        e("v_mov_b32", "v%d" % reg, "v%d" % r)
                        e("v_mov_b32", "v%d" % (reg + 1), "v%d" % (r + 1))
    e("s_waitcnt", "lgkmcnt(0)")
    e("v_mov_b32", "v%d" % RV_ONEHI, ONE_HI)
    e("s_mov_b32", "s%d" % S_CNT, "s%d" % S_ITERS)
    e("label", "7")
    # ---- row scalings Et_i = 1/sqrt(limit(max_j |A_ij|)) -> AGPRs
    rows = [[] for _ in range(nc)]
    for j in range(nx):
        for p_ in range(s.A_p[j], s.A_p[j + 1]):
            rows[s.A_i[p_]].append(p_)
    for i in range(nc):
        t = T(0)
        ps = rows[i]
        if len(ps) == 1:
            e("v_max_f64", vp(t), "|" + vp(A(ps[0])) + "|", "|" + vp(A(ps[0])) + "|")
        else:
            e("v_max_f64", vp(t), "|" + vp(A(ps[0])) + "|", "|" + vp(A(ps[1])) + "|")
            for p_ in ps[2:]:
                e("v_max_f64", vp(t), vp(t), "|" + vp(A(p_)) + "|")
        limit(t, T(1))
        rsqrt(T(1), t, T(2), T(3))
        acc_write(RA_ET + 2 * i, T(1))
    # ---- columns: Dt_j from the untouched column, then P_j, column j of A, q_j; pmean in T(4), qn in T(5)
    for r in (T(4), T(5)):
        e("v_mov_b32", "v%d" % r, 0)
        e("v_mov_b32", "v%d" % (r + 1), 0)
    for j in range(nx):
        pj, t, dt = T(6), T(0), T(1)
        acc_read(pj, RA_P + 2 * j)
        first = True
        pn = pj
        if CARRY_C:      # P and q are kept WITHOUT the accumulated cost scaling c; the norms see c * P_j
            pn = T(3)
            e("v_mul_f64", vp(pn), vp(pj), vp(RV_C))
        for p_ in range(s.A_p[j], s.A_p[j + 1]):
            e("v_max_f64", vp(t), "|" + vp(pn if first else t) + "|", "|" + vp(A(p_)) + "|")
            first = False
        if first:
            e("v_max_f64", vp(t), "|" + vp(pn) + "|", "|" + vp(pn) + "|")
        limit(t, dt)
        rsqrt(dt, t, T(2), T(3))
        e("v_mul_f64", vp(pj), vp(pj), vp(dt))
        e("v_mul_f64", vp(pj), vp(pj), vp(dt))
        acc_write(RA_P + 2 * j, pj)
        e("v_add_f64", vp(T(4)), vp(T(4)), "|" + vp(pj) + "|")
        for p_ in range(s.A_p[j], s.A_p[j + 1]):
            acc_read(T(2), RA_ET + 2 * s.A_i[p_])
            e("v_mul_f64", vp(A(p_)), vp(A(p_)), vp(T(2)))
            e("v_mul_f64", vp(A(p_)), vp(A(p_)), vp(dt))
        kind, reg = qhome(j)
        if kind == "a":
            acc_read(T(2), reg)
            e("v_mul_f64", vp(T(2)), vp(T(2)), vp(dt))
            acc_write(reg, T(2))
            qr = T(2)
        else:
            e("v_mul_f64", vp(reg), vp(reg), vp(dt))
            qr = reg
        e("v_max_f64", vp(T(5)), vp(T(5)), "|" + vp(qr) + "|")
    # ---- cost scaling: ct = 1 / limit(max(pmean / nx, limit(qn)))
    b = f64bits(float(nx))
    e("v_mov_b32", "v%d" % T(0), b & 0xFFFFFFFF)
    e("v_mov_b32", "v%d" % (T(0) + 1), b >> 32)
    recip(T(1), T(0), T(2))
    if CARRY_C:          # sum |P_j| and max |q_j| were formed without c (both commute with a positive factor up to rounding)
        e("v_mul_f64", vp(T(4)), vp(T(4)), vp(RV_C))
        e("v_mul_f64", vp(T(5)), vp(T(5)), vp(RV_C))
    e("v_mul_f64", vp(T(2)), vp(T(4)), vp(T(1)))
    e("v_fma_f64", vp(T(3)), "-" + vp(T(0)), vp(T(2)), vp(T(4)))
    e("v_fma_f64", vp(T(4)), vp(T(3)), vp(T(1)), vp(T(2)))              # pmean / nx
    limit(T(5), T(0))
    e("v_max_f64", vp(T(4)), vp(T(4)), vp(T(5)))
    limit(T(4), T(0))
    recip(T(5), T(4), T(0))                                              # ct
    e("v_mul_f64", vp(RV_C), vp(RV_C), vp(T(5)))
    for j in range(nx if not CARRY_C else 0):
        acc_read(T(j % 2), RA_P + 2 * j)
        e("v_mul_f64", vp(T(j % 2)), vp(T(j % 2)), vp(T(5)))
        acc_write(RA_P + 2 * j, T(j % 2))
    for j in range(nx if not CARRY_C else 0):
        kind, reg = qhome(j)
        if kind == "a":
            acc_read(T(2 + j % 2), reg)
            e("v_mul_f64", vp(T(2 + j % 2)), vp(T(2 + j % 2)), vp(T(5)))
            acc_write(reg, T(2 + j % 2))
        else:
            e("v_mul_f64", vp(reg), vp(reg), vp(T(5)))
    e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_CNT, 1)
    e("s_cmp_gt_i32", "s%d" % S_CNT, 0)
    e("s_cbranch_scc1", "7b")
    # ---- epilogue: A, P, q, c back to LDS
    for qd, ws in aquads:
        _write_quad(e, qd, ws, {w_: A(w_ - RZ_A) for w_ in ws})
    for lo, home in ((RZ_P, lambda j: ("a", RA_P + 2 * j)), (RZ_Q, qhome)):
        quads = _words(lo, lo + nx)
        for g in range(0, len(quads), 4):
            grp = quads[g:g + 4]
            for k, (qd, ws) in enumerate(grp):
                regs = {}
                for w_ in ws:
                    kind, reg = home(w_ - lo)
                    r = T(2 * k) + (w_ & 1) * 2
                    if kind == "a":
                        acc_read(r, reg)
                    else:
                        e("v_mov_b32", "v%d" % r, "v%d" % reg)
                        e("v_mov_b32", "v%d" % (r + 1), "v%d" % (reg + 1))
                    if CARRY_C:      # the accumulated cost scaling, once
                        e("v_mul_f64", vp(r), vp(r), vp(RV_C))
                    regs[w_] = r
                _write_quad(e, qd, ws, regs)
    base, off, _ = lds_addr(RZ_C)
    e("ds_write_b64", base, vp(RV_C), off + 8 * (RZ_C & 1))
    e("s_waitcnt", "lgkmcnt(0)")
    return e.ins, s


# ---------------------------------------------------------------------------
# Residual norms of the fp64 step (update_info, auxil.c:243-307) as one assembly block  (round 3)
# ---------------------------------------------------------------------------
# hipcc's fp64 phase C keeps D, E (84 doubles), the 39 / 45 accumulators and the 111 scaled entries of A it has
# decided to reuse alive at once, parks ~450 words in scratch and AGPRs and fetches them back one exposed load at a
# time: 38.7 of the 341 us step (tools/f64_timing.py). Here:
#
#   v4..v81     E (39)             v82..v159   A x accumulators (pass 1), then y (pass 2)
#   v160..v179  T0 dt, s0 dt[3], Btau dt[6] (the raw per-robot entries of A)     v180:181 dt    v182:183 c
#   v184..v199  pri, dua (before 1/c), |z|, |Ax|, |q|, |A'y|, |Px| norms and the NaN accumulator
#   v200..v215  D_j, x_j / z_i stream (double-buffered ds_read_b64)              v216..v241 temporaries
#   a0..a89     q (the 27 entries that can be non-zero, global loads issued first)   a90..a105  the eight weights
#
# Pass 1 walks the columns: A x with the entry re-derived as (raw * E_i) * D_j (the factorisation consumed the
# equilibrated copy), then the row norms. Pass 2 walks them again for A' y and the dual terms. Arithmetic =
# UMPC_GEN_A_MUL_SCALED / UMPC_GEN_AT_MUL_SCALED and the two norm loops of csrc/umpc_step.h operation by operation
# (umpc_recip = v_rcp_f64 + two Newton steps; fmax = v_max_f64; x + 0 and max(n, |t * 0|) dropped for the 18
# structural zeros of q), so the C++ status logic that follows sees the values it computed itself before.
#
# LDS words on entry: x, y (LW_X, LW_Y), z, D, E (PC_Z, PC_DS, PC_ES) from the ADMM block; RS_PAR.. the ten raw
# entries, RS_C c, RS_W.. the eight weights (ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom), RS_DT dt -- written by the
# C++ side just before. On exit RS_OUT.. = pri_res, dua_res * c (before the division), |z|, |Ax|, |q|, |A'y|, |Px|,
# NaN accumulator.
RS_PAR, RS_C, RS_W, RS_DT, RS_OUT = 297, 307, 308, 316, 297
SV_E, SV_R, SV_PAR, SV_DT, SV_C, SV_N, SV_S, SV_T = 4, 82, 160, 180, 182, 184, 200, 216
SA_Q, SA_W = 0, 90
N_PRI, N_DUA, N_Z, N_AX, N_Q, N_ATY, N_PX, N_NAN = range(8)


def weight_class(s, j):
    """index into (ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom) of column j's raw P entry: PXRAW_OF, csrc/umpc_step.h"""
    N, NY, NU = s.N, symbolic.NY, 3
    if j < 2 * N * NY:
        if j % NY < 3:
            return (3 if j // NY == N - 1 else 2) if j < N * NY else (5 if (j - N * NY) // NY == N - 1 else 4)
        return 0 if j < N * NY else 1
    return 6 if (j - 2 * N * NY) % NU == 0 else 7


def resid_program(N=3, perm=None):
    s = symbolic.analyse(N, perm)
    nx, nc = s.nx, s.nc
    qnz, _ = rhs_structure(s)
    e = Emit()
    E = lambda i: SV_E + 2 * i
    R = lambda i: SV_R + 2 * i
    T = lambda k: SV_T + 2 * k
    NRM = lambda k: SV_N + 2 * k
    PAR = lambda k: SV_PAR + 2 * k

    def rd64(dst, word):
        base, off, half = lds_addr(word)
        e("ds_read_b64", vp(dst), base, off + 8 * half)

    def load_words(lo, n, reg_of):
        """LDS words lo..lo+n-1 -> consecutive register pairs reg_of(k) (k = word - lo); returns the LDS instructions issued"""
        cnt = 0
        for qd, ws in _words(lo, lo + n):
            base, off, _ = lds_addr(2 * qd)
            if len(ws) == 2:
                assert reg_of(ws[1] - lo) == reg_of(ws[0] - lo) + 2
                e("ds_read_b128", "v[%d:%d]" % (reg_of(ws[0] - lo), reg_of(ws[0] - lo) + 3), base, off)
            else:
                e("ds_read_b64", vp(reg_of(ws[0] - lo)), base, off + 8 * (ws[0] & 1))
            cnt += 1
        return cnt

    def recip(y, v_, a_):
        e("v_rcp_f64", vp(y), vp(v_))
        e("s_nop", 0)
        for _ in range(2):
            e("v_fma_f64", vp(a_), "-" + vp(v_), vp(y), 1.0)
            e("v_fma_f64", vp(y), vp(y), vp(a_), vp(y))

    def raw_of(tag):
        if tag[0] == "c":
            assert tag[1] in (1.0, -1.0)
            return ("c", tag[1])
        if tag[0] == "dt":
            return ("v", SV_DT)
        if tag[0] == "T0dt":
            return ("v", PAR(0))
        if tag[0] == "s0":
            return ("v", PAR(1 + tag[1]))
        assert tag[0] == "Btau"
        return ("v", PAR(4 + tag[1]))

    def scaled_entry(dst, tmp, p_, i, dreg):
        """dst <- (raw * E_i) * D_j"""
        kind, val = raw_of(s.A_tag[p_])
        if kind == "c":
            e("v_mul_f64", vp(dst), ("-" if val < 0 else "") + vp(E(i)), vp(dreg))     # (+-1 * E) * D: the sign is exact
        else:
            e("v_mul_f64", vp(tmp), vp(val), vp(E(i)))
            e("v_mul_f64", vp(dst), vp(tmp), vp(dreg))

    # ---- prologue
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    e("v_add_u32", "v%d" % V_B1, 0x10000, "v1")
    e("v_add_u32", "v%d" % V_B2, 0x20000, "v1")
    for j in qnz:
        _row_ptr(e, S_P, S_WS, FAC_Q + j)
        e("global_load_dwordx2", "a[%d:%d]" % (SA_Q + 2 * j, SA_Q + 2 * j + 1), "v0", sp(S_P))
    load_words(PC_ES, nc, lambda k: E(k))
    load_words(RS_PAR, 10, lambda k: PAR(k))
    rd64(SV_C, RS_C)
    rd64(SV_DT, RS_DT)
    load_words(RS_W, 8, lambda k: T(k))
    e("s_waitcnt", "lgkmcnt(0)")
    for k in range(8):
        e("v_accvgpr_write_b32", "a%d" % (SA_W + 2 * k), "v%d" % T(k))
        e("v_accvgpr_write_b32", "a%d" % (SA_W + 2 * k + 1), "v%d" % (T(k) + 1))
    for r in list(range(SV_R, SV_R + 2 * nc)) + list(range(SV_N, SV_N + 16)):
        e("v_mov_b32", "v%d" % r, 0)
    DB = lambda j: SV_S + 4 * (j & 1)            # D_j
    XB = lambda j: SV_S + 4 * (j & 1) + 2        # x_j (pass 1, pass 2) / z_i (row norms share the x buffers)
    ZB = lambda i: SV_S + 8 + 2 * (i & 1)

    # ---- pass 1: A x, column by column (entries of a column ascending, as the macro)
    rd64(DB(0), PC_DS + 0)
    rd64(XB(0), LW_X + 0)
    for j in range(nx):
        if j + 1 < nx:
            rd64(DB(j + 1), PC_DS + j + 1)
            rd64(XB(j + 1), LW_X + j + 1)
        e("s_waitcnt", "lgkmcnt(%d)" % (2 if j + 1 < nx else 0))
        ent = list(range(s.A_p[j], s.A_p[j + 1]))
        # the product of entry k + 1 is formed before the accumulation of entry k (no back-to-back dependence)
        for k, p_ in enumerate(ent):
            if k == 0:
                scaled_entry(T(0), T(2), p_, s.A_i[p_], DB(j))
            if k + 1 < len(ent):
                scaled_entry(T((k + 1) & 1), T(2), ent[k + 1], s.A_i[ent[k + 1]], DB(j))
            i = s.A_i[p_]
            e("v_fma_f64", vp(R(i)), vp(T(k & 1)), vp(XB(j)), vp(R(i)))
    # ---- row norms
    rd64(ZB(0), PC_Z + 0)
    for i in range(nc):
        if i + 1 < nc:
            rd64(ZB(i + 1), PC_Z + i + 1)
        einv, a_, d_, t_ = T(0), T(1), T(2), T(3)
        recip(einv, E(i), a_)
        e("s_waitcnt", "lgkmcnt(%d)" % (1 if i + 1 < nc else 0))
        e("v_add_f64", vp(d_), vp(R(i)), "-" + vp(ZB(i)))
        e("v_mul_f64", vp(t_), vp(einv), vp(d_))
        e("v_mul_f64", vp(T(4)), vp(einv), vp(ZB(i)))
        e("v_mul_f64", vp(T(5)), vp(einv), vp(R(i)))
        e("v_max_f64", vp(NRM(N_PRI)), vp(NRM(N_PRI)), "|" + vp(t_) + "|")
        e("v_fma_f64", vp(NRM(N_NAN)), 0.0, vp(d_), vp(NRM(N_NAN)))
        e("v_max_f64", vp(NRM(N_Z)), vp(NRM(N_Z)), "|" + vp(T(4)) + "|")
        e("v_max_f64", vp(NRM(N_AX)), vp(NRM(N_AX)), "|" + vp(T(5)) + "|")
    # ---- pass 2: y -> the accumulator registers; A' y and the dual terms column by column
    load_words(LW_Y, nc, lambda k: R(k))
    rd64(DB(0), PC_DS + 0)
    rd64(XB(0), LW_X + 0)
    e("s_waitcnt", "vmcnt(0)")              # q has long arrived
    for j in range(nx):
        if j + 1 < nx:
            rd64(DB(j + 1), PC_DS + j + 1)
            rd64(XB(j + 1), LW_X + j + 1)
        e("s_waitcnt", "lgkmcnt(%d)" % (2 if j + 1 < nx else 0))
        aty, dinv, a_, px, qj, r_ = T(3), T(4), T(5), T(6), T(7), T(8)
        ent = list(range(s.A_p[j], s.A_p[j + 1]))
        recip(dinv, DB(j), a_)
        e("v_accvgpr_read_b32", "v%d" % px, "a%d" % (SA_W + 2 * weight_class(s, j)))
        e("v_accvgpr_read_b32", "v%d" % (px + 1), "a%d" % (SA_W + 2 * weight_class(s, j) + 1))
        if j in qnz:
            e("v_accvgpr_read_b32", "v%d" % qj, "a%d" % (SA_Q + 2 * j))
            e("v_accvgpr_read_b32", "v%d" % (qj + 1), "a%d" % (SA_Q + 2 * j + 1))
        for k, p_ in enumerate(ent):
            if k == 0:
                scaled_entry(T(0), T(2), p_, s.A_i[p_], DB(j))
            if k + 1 < len(ent):
                scaled_entry(T((k + 1) & 1), T(2), ent[k + 1], s.A_i[ent[k + 1]], DB(j))
            i = s.A_i[p_]
            e("v_fma_f64", vp(aty), vp(T(k & 1)), vp(R(i)), 0.0 if k == 0 else vp(aty))
        if not ent:
            e("v_mov_b32", "v%d" % aty, 0)
            e("v_mov_b32", "v%d" % (aty + 1), 0)
        # Pxj = (((PXRAW * D) * D) * c) * x
        e("v_mul_f64", vp(px), vp(px), vp(DB(j)))
        e("v_mul_f64", vp(px), vp(px), vp(DB(j)))
        e("v_mul_f64", vp(px), vp(px), vp(SV_C))
        e("v_mul_f64", vp(px), vp(px), vp(XB(j)))
        if j in qnz:
            e("v_add_f64", vp(r_), vp(qj), vp(px))
            e("v_add_f64", vp(r_), vp(r_), vp(aty))
            e("v_mul_f64", vp(T(9)), vp(dinv), vp(qj))
        else:
            e("v_add_f64", vp(r_), vp(px), vp(aty))          # (0 + Pxj) + A'y_j
        e("v_mul_f64", vp(T(10)), vp(dinv), vp(r_))
        e("v_mul_f64", vp(T(11)), vp(dinv), vp(aty))
        e("v_mul_f64", vp(T(12)), vp(dinv), vp(px))
        e("v_max_f64", vp(NRM(N_DUA)), vp(NRM(N_DUA)), "|" + vp(T(10)) + "|")
        e("v_fma_f64", vp(NRM(N_NAN)), 0.0, vp(r_), vp(NRM(N_NAN)))
        if j in qnz:
            e("v_max_f64", vp(NRM(N_Q)), vp(NRM(N_Q)), "|" + vp(T(9)) + "|")
        e("v_max_f64", vp(NRM(N_ATY)), vp(NRM(N_ATY)), "|" + vp(T(11)) + "|")
        e("v_max_f64", vp(NRM(N_PX)), vp(NRM(N_PX)), "|" + vp(T(12)) + "|")
    # ---- the eight results -> LDS
    for qd, ws in _words(RS_OUT, RS_OUT + 8):
        _write_quad(e, qd, ws, {w: NRM(w - RS_OUT) for w in ws})
    e("s_waitcnt", "lgkmcnt(0)")
    assert SV_T + 2 * 13 <= V_END
    return e.ins, s

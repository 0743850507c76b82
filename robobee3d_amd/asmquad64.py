"""ONE ROBOT PER LANE QUAD, fp64: ADMM iterations 2..maxIter of the small-batch fp64 step (asmgen64.py, BASELINE config 2:
B = 4 096) -- the fp64 counterpart of asmquad.py. 4 096 robots are 64 one-lane waves on 64 of the chip's 256 CUs; with
four lanes per robot they are 256 waves, one per CU (the fp64 loop owns its CU's LDS), and each iteration is 819
instructions with 314 fp64 operations instead of 1 483 with 786 (measured: 0.295 -> 0.224 ms per step; 0.185 with the Ruiz
passes on the quad too, ruiz_program below; 0.152 with the coefficients gathered once per step).

Same idea as asmquad.py (its docstring has the reasoning): the phases around the loop run redundantly in the four lanes
of a quad, the loop keeps a third of the unknowns in each of lanes 0..2, a triangular-solve operation runs in the lane
of its destination, operations with one destination and one source register share an instruction (asmquad.QuadPlan is
used unchanged). What differs for eight-byte words:

  * v_fma_f64 is VOP3: no DPP operand. A source that lives in another lane is fetched with two v_mov_b32_dpp (low, high
    word) into a temporary pair right before its consumer; consecutive consumers of one (register, selection) reuse it.
  * a lane owns 121 words of VGPRs: W, x, y, 1/D and the thrust-row words (92 words) leave no room for the 194
    coefficient words. Every lane already holds the whole factor in its own LDS slice (phase A wrote it there for the
    one-lane first iteration) and instruction q needs, in lane l, ONE of its words -- a per-lane choice, a constant table
    (`table()`, csrc/umpc_quad64_tab.h); a lane with nothing to do in an instruction gets a word that holds 0.0.
    Rounds 3-4 read that word with a per-lane address EVERY iteration; since round 5 the entry gathers each lane's 194
    words once per step into AGPR pairs and a compact LDS array read at uniform addresses (see NAC / CW0 below):
    0.173 -> 0.152 ms per step (profiles/r05_quad64_gather_ab.txt).
  * q and l of the lane's unknowns sit in AGPR pairs (read once per iteration), composed from the one-lane homes.

Exit: x, y, x_prev and delta_y of the capturing iteration go back to EVERY lane's LDS slice (two DPP moves + one
ds_write_b64 per word), the thrust-row z to its registers: the epilogue and phase C read what they always read.
Arithmetic = asmgen64.body's, operation for operation, except the order in which an unknown's updates are added up.

Reference mapping: template/uprightmpc2/osqp.c:354-370, auxil.c:164-228, qdldl.c:250-293 (through asmgen64.body).
"""
import struct

import numpy as np

from . import asmquad, symbolic
from .asmgen import S_ALPHA, S_CNT, S_ITERS, S_OMA, S_RINV, S_SIGMA
from .asmquad import IX_EQ, IX_T, qperm

# ---- VGPR words (first register of the pair); the block may use v2..v245, v172..v201 are the one-lane thrust-row words
QYT, QZT, QLO3, QUP3, QRHO3, QRINV3 = 118, 120, 122, 124, 126, 128
T_CQ = [218, 222, 226, 230, 234, 238]  # coefficient float4 in flight: two coefficients per ds_read_b128 (ring of six)
T_COEF = [230, 232, 234, 236, 238, 240]    # coefficient words read from AGPRs in flight (ring of NRING; the first NAC instructions)
T_SRC = [242, 244]                     # sources fetched from another lane
T_A = [230, 232, 234, 236]             # AGPR read / update temporaries (outside the solves: the rings are idle there)
T_X = [238, 240]
NRING = 6
RD_AHEAD = 5                           # coefficient reads run this many solve instructions ahead
STAGE = 4                              # entry staging: the (idle) W words, v4..v59 = 14 float4
AQ, AL = 0, 32                         # AGPRs: q (16 words), l (12 words)
NTAB = 196                             # table row length (dwords): the 194 solve instructions + padding to whole dwordx4 loads
ZERO_WORD = 319                        # this lane's LDS word that holds 0.0
S_L0, S_L1, S_L2, S_EXEC, S_TAB = 30, 32, 34, 36, 8
# Where a lane's coefficient of solve instruction n lives during the loop (round 5). Every lane holds the whole factor in its
# own LDS slice, and instruction n needs, in lane l, ONE of its words -- a per-lane choice (`table()`). Rounds 4 read it with a
# per-lane address every iteration (ds_read_b64 + a quarter of an address float4: two LDS instructions per coefficient pair and
# then some, and an LDS instruction costs a lone wave ~6 ns whatever its width). Now the entry GATHERS each lane's 194 words
# ONCE per step -- addresses straight from the constant table, per-lane reads as before -- into storage that needs no
# address in the loop:
#   n < NAC       AGPR pairs (per-lane storage by nature): a56..a167 (dead 1/D copies of the one-lane block) and a200..a251
#                 (the one-lane homes of q: composed into a0..a31 by the entry; the epilogue reads the l homes a168..a199 only)
#   n >= NAC      a COMPACT array in the lane's LDS words CW0.., instruction n at word CW0 + n - NAC: ONE ds_read_b128 at a
#                 uniform address fetches the coefficients of two instructions. The array overlays the words of x and y
#                 (they live in registers during the loop; the exit rewrites them) and the last seven words of L, which the
#                 gather therefore overwrites in its LAST round, after every source has been read.
NAC1, NAC = 56, 82
AC0, AC1 = 56, 200
CW0 = 206
assert NAC % 2 == 0 and CW0 % 2 == 0 and AC0 + 2 * NAC1 <= 168 and AC1 + 2 * (NAC - NAC1) <= 254 and CW0 + (194 - NAC) <= ZERO_WORD - 1


def coef_agpr(n):
    return AC0 + 2 * n if n < NAC1 else AC1 + 2 * (n - NAC1)


def coef_word(n):
    return CW0 + n - NAC


def QW(ix):
    return 4 + 2 * ix


def QX(ix):
    return 62 + 2 * ix


def QY(ix):
    return 94 + 2 * ix


def QDI(ix):
    return 130 + 2 * ix if ix < 21 else 202 + 2 * (ix - 21)


def vp(n):
    return "v[%d:%d]" % (n, n + 1)


def sp(n):
    return "s[%d:%d]" % (n, n + 1)


def rel_addr(word):
    """byte offset of LDS word `word` from the lane's LDS base (asmgen64's [quad of two words][lane] layout)"""
    return (word >> 1) * 1024 + 8 * (word & 1)


def plan_for(s):
    class _St:      # the two facts QuadPlan needs of asmstep.Struct
        pass
    st = _St()
    st.s, st.neq = s, 2 * s.N * symbolic.NY
    return asmquad.QuadPlan(st)


def table(plan):
    """uint32 [4][NTAB]: for lane class l (lane & 3) and solve instruction q (forward instructions first) the byte offset,
    from the lane's LDS base, of the entry of L that lane multiplies with -- the zero word where it has none."""
    tab = np.full((4, NTAB), rel_addr(ZERO_WORD), np.uint32)
    for q, ins in enumerate(plan.fwd + plan.bwd):
        for ln, o in ins["ops"].items():
            tab[ln, q] = rel_addr(o[2])
    return tab


# ---------------------------------------------------------------------------------------------------------------------
def entry(e, plan, s):
    from . import asmgen64 as g
    nx, nc, nk, neq = s.nx, s.nc, s.nk, 2 * s.N * symbolic.NY
    v = lambda n: "v%d" % n
    masks = (S_L0, S_L1, S_L2)
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    e("s_mov_b64", sp(S_EXEC), "exec")
    for ln, m in enumerate(masks):
        e("s_mov_b32", "s%d" % m, 0x11111111 << ln)
        e("s_mov_b32", "s%d" % (m + 1), 0x11111111 << ln)
        e("s_and_b64", sp(m), sp(m), sp(S_EXEC))
    # the coefficient-address table first: its loads land while everything else is moved
    tb = T_X[0]
    e("v_bfe_u32", v(tb), "v1", 4, 2)
    e("v_mul_u32_u24", v(tb), NTAB * 4, v(tb))
    # zero: the quad words of lanes that own nothing (lane 3, pads) and the AGPR words of q / l
    zero = [QX(k) for k in range(16)] + [QY(k) for k in range(12)] + [QYT, QZT, QLO3, QUP3, QRHO3, QRINV3] + \
           [QDI(k) for k in range(asmquad.NQW)]
    for r in zero:
        e("v_mov_b32", v(r), 0)
        e("v_mov_b32", v(r + 1), 0)
    # ---- 1/D (AGPR pairs, permuted order) -> QDI; thrust-row words
    mv = {0: [], 1: [], 2: []}
    for k in range(nk):
        ln, ix = plan.home[k]
        mv[ln].append(("a", QDI(ix), g.A_D + 2 * k))
    for k in range(s.N):
        Z3, LO3, UP3, RHO3, RINV3 = (g.V_C + 2 * (q * s.N + k) for q in range(5))
        for dst, src in ((QZT, Z3), (QLO3, LO3), (QUP3, UP3), (QRHO3, RHO3), (QRINV3, RINV3)):
            mv[k].append(("v", dst, src))
    for ln in range(3):
        e("s_mov_b64", "exec", sp(masks[ln]))
        for kind, dst, src in mv[ln]:
            for h in range(2):
                if kind == "a":
                    e("v_accvgpr_read_b32", v(dst + h), "a%d" % (src + h))
                else:
                    e("v_mov_b32", v(dst + h), v(src + h))
    e("s_mov_b64", "exec", sp(S_EXEC))
    # ---- q, l: one-lane homes (a168..) -> the lane's AGPR words a0..a55 (zero where the entry is structurally zero)
    e("v_mov_b32", v(T_A[0]), 0)
    for a_ in range(AQ, AL + 24):
        e("v_accvgpr_write_b32", "a%d" % a_, v(T_A[0]))
    hm = g.homes(s)
    byl = {0: [], 1: [], 2: []}
    for (kind, idx), a_ in hm.items():
        if kind == "q":
            ln, ix = plan.xhome[idx]
            byl[ln].append((AQ + 2 * ix, a_))
        else:
            ln, ix = plan.zhome[idx]
            byl[ln].append((AL + 2 * (ix - IX_EQ), a_))
    for ln in range(3):
        e("s_mov_b64", "exec", sp(masks[ln]))
        for k, (dst, src) in enumerate(sorted(byl[ln])):
            t = T_A[k % 4]
            for h in range(2):
                e("v_accvgpr_read_b32", v(t + h), "a%d" % (src + h))
            for h in range(2):
                e("v_accvgpr_write_b32", "a%d" % (dst + h), v(t + h))
    e("s_mov_b64", "exec", sp(S_EXEC))
    # ---- x, y: LDS words -> quad words, 14 float4 per round through the W words
    words = [(g.LW_X + j, plan.xhome[j][0], QX(plan.xhome[j][1])) for j in range(nx)]
    for i in range(nc):
        ln, ix = plan.zhome[i]
        words.append((g.LW_Y + i, ln, QY(ix - IX_EQ) if i < neq else QYT))
    byquad = {}
    for w, ln, dst in words:
        byquad.setdefault(w >> 1, []).append((w, ln, dst))
    quads = sorted(byquad)
    for base in range(0, len(quads), 14):
        chunk = quads[base:base + 14]
        for k, qd in enumerate(chunk):
            b_, off, _ = g.lds_addr(2 * qd)
            e("ds_read_b128", "v[%d:%d]" % (STAGE + 4 * k, STAGE + 4 * k + 3), b_, off)
        e("s_waitcnt", "lgkmcnt(0)")
        for ln in range(3):
            todo = [(STAGE + 4 * k + 2 * (w & 1), dst) for k, qd in enumerate(chunk) for (w, l2, dst) in byquad[qd] if l2 == ln]
            if not todo:
                continue
            e("s_mov_b64", "exec", sp(masks[ln]))
            for src, dst in todo:
                e("v_mov_b32", v(dst), v(src))
                e("v_mov_b32", v(dst + 1), v(src + 1))
        e("s_mov_b64", "exec", sp(S_EXEC))
    # ---- the zero word (lanes that have nothing to do in an instruction read it during the gather)
    e("v_mov_b32", v(T_A[0]), 0)
    e("v_mov_b32", v(T_A[0] + 1), 0)
    b_, off, half = g.lds_addr(ZERO_WORD)
    e("ds_write_b64", b_, vp(T_A[0]), off + 8 * half)
    e("s_waitcnt", "lgkmcnt(0)")
    # ---- the gather: the lane's coefficient of every solve instruction -> its AGPR pair / its word of the compact array.
    # Table groups of four instructions, up to four groups per round through the idle W words: addresses from the constant
    # table (+ the lane's LDS base), one per-lane ds_read_b64 each, then the destination. The compact array shares its first
    # words with the last words of L: the rounds that write BELOW word NNZL run last, in one round, behind all other reads.
    nops = len(plan.fwd) + len(plan.bwd)
    assert nops <= NTAB and nops % 2 == 0 and coef_word(nops - 1) < ZERO_WORD
    ngrp = (nops + 3) // 4
    last = [gq for gq in range(ngrp) if any(NAC <= n < nops and coef_word(n) < g.NNZL for n in range(4 * gq, 4 * gq + 4))]
    first = [gq for gq in range(ngrp) if gq not in last and all(n < NAC for n in range(4 * gq, 4 * gq + 4))]
    mid = [gq for gq in range(ngrp) if gq not in last and gq not in first]
    assert len(last) <= 4 and all(coef_word(n) >= g.NNZL for gq in mid for n in range(4 * gq, 4 * gq + 4) if NAC <= n < nops)
    rounds = [first[k:k + 4] for k in range(0, len(first), 4)] + [mid[k:k + 4] for k in range(0, len(mid), 4)] + [last]
    for gqs in rounds:
        for k, gq in enumerate(gqs):
            e("global_load_dwordx4", "v[%d:%d]" % (STAGE + 4 * k, STAGE + 4 * k + 3), v(tb), sp(S_TAB), "offset:%d" % (16 * gq))
        e("s_waitcnt", "vmcnt(0)")
        for k, gq in enumerate(gqs):
            for h in range(4):
                e("v_add_u32", v(STAGE + 4 * k + h), v(STAGE + 4 * k + h), "v1")
        for k, gq in enumerate(gqs):
            for h in range(4):
                e("ds_read_b64", vp(STAGE + 16 + 8 * k + 2 * h), v(STAGE + 4 * k + h), 0)
        e("s_waitcnt", "lgkmcnt(0)")
        for k, gq in enumerate(gqs):
            for h in range(0, 4, 2):
                n_ = 4 * gq + h
                r = STAGE + 16 + 8 * k + 2 * h
                if n_ >= nops:
                    continue
                if n_ < NAC:
                    for d in range(2):
                        e("v_accvgpr_write_b32", "a%d" % coef_agpr(n_ + d), v(r + 2 * d))
                        e("v_accvgpr_write_b32", "a%d" % (coef_agpr(n_ + d) + 1), v(r + 2 * d + 1))
                else:
                    b_, off, half = g.lds_addr(coef_word(n_))
                    assert half == 0
                    e("ds_write_b128", b_, "v[%d:%d]" % (r, r + 3), off)
    e("s_waitcnt", "lgkmcnt(0)")
    e("s_nop", 4)


def body(e, plan, s, capture):
    v = lambda n: "v%d" % n
    sA, sO, sS, sRi = (sp(r) for r in (S_ALPHA, S_OMA, S_SIGMA, S_RINV))
    # ---- rhs
    for ix in range(15):
        t = T_A[ix % 4]
        e("v_accvgpr_read_b32", v(t), "a%d" % (AQ + 2 * ix))
        e("v_accvgpr_read_b32", v(t + 1), "a%d" % (AQ + 2 * ix + 1))
        e("v_fma_f64", vp(QW(ix)), sS, vp(QX(ix)), "-" + vp(t))
    for ix in range(12):
        t = T_A[(ix + 3) % 4]
        e("v_accvgpr_read_b32", v(t), "a%d" % (AL + 2 * ix))
        e("v_accvgpr_read_b32", v(t + 1), "a%d" % (AL + 2 * ix + 1))
        e("v_fma_f64", vp(QW(IX_EQ + ix)), "-" + vp(QY(ix)), sRi, vp(t))
    e("v_fma_f64", vp(QW(IX_T)), "-" + vp(QRINV3), vp(QYT), vp(QZT))
    e("s_nop", 1)
    # ---- solves: address (AGPR) three instructions ahead, coefficient read two ahead, the other lane's source right
    # before its consumer
    seq = [("op", q, ins) for q, ins in enumerate(plan.fwd)] + [("diag",)] + \
          [("op", len(plan.fwd) + q, ins) for q, ins in enumerate(plan.bwd)]
    opsidx = [k for k, it in enumerate(seq) if it[0] == "op"]
    pos = {k: n for n, k in enumerate(opsidx)}          # sequence index -> running op number
    nops = len(opsidx)

    from . import asmgen64 as g
    nlds = [0]                      # LDS reads issued so far in this body (they complete in order)
    co_seq = {}                     # solve instruction whose coefficient comes from LDS -> issue number of its read

    def wait_for(seq):
        e("s_waitcnt", "lgkmcnt(%d)" % min(15, nlds[0] - 1 - seq))

    def creg(n):
        """first register of instruction n's coefficient word"""
        if n < NAC:
            return T_COEF[n % NRING]
        return T_CQ[((n - NAC) // 2) % len(T_CQ)] + 2 * ((n - NAC) & 1)

    def read(n):
        if n >= nops:
            return
        if n < NAC:
            e("v_accvgpr_read_b32", v(T_COEF[n % NRING]), "a%d" % coef_agpr(n))
            e("v_accvgpr_read_b32", v(T_COEF[n % NRING] + 1), "a%d" % (coef_agpr(n) + 1))
        elif (n - NAC) % 2 == 0:
            b_, off, half = g.lds_addr(coef_word(n))
            r = creg(n)
            e("ds_read_b128", "v[%d:%d]" % (r, r + 3), b_, off)      # the coefficients of instructions n and n + 1
            co_seq[n] = co_seq[n + 1] = nlds[0]
            nlds[0] += 1
    assert all(seq[opsidx[n]][1] == n for n in range(nops))     # running op number == coefficient index
    for n in range(RD_AHEAD):
        read(n)
    waited_co = [-1]
    lastw = [None, None]
    cached = [None, None]           # (sreg, perm) held by T_SRC[k]
    nsrc = 0
    for k, it in enumerate(seq):
        if it[0] == "diag":
            for ix in range(asmquad.NQW):
                if ix != 15:
                    e("v_mul_f64", vp(QW(ix)), vp(QW(ix)), vp(QDI(ix)))
            lastw, cached = [None, None], [None, None]
            continue
        n = pos[k]
        read(n + RD_AHEAD)
        _, q, ins = it
        d, sr = QW(ins["d"]), QW(ins["s"])
        if ins["perm"] == [0, 1, 2, 3]:
            src = sr
        else:
            key = (sr, tuple(ins["perm"]))
            if key in cached:
                src = T_SRC[cached.index(key)]
            else:
                slot = nsrc % 2
                nsrc += 1
                if sr == lastw[0]:
                    e("s_nop", 1)
                elif sr == lastw[1]:
                    e("s_nop", 0)
                e("v_mov_b32_dpp", v(T_SRC[slot]), v(sr), qperm(ins["perm"]))
                e("v_mov_b32_dpp", v(T_SRC[slot] + 1), v(sr + 1), qperm(ins["perm"]))
                cached[slot] = key
                src = T_SRC[slot]
        if n in co_seq and waited_co[0] < co_seq[n]:            # one wait per float4 = two coefficients (reads return in order)
            wait_for(co_seq[n])
            waited_co[0] = co_seq[n]
        e("v_fma_f64", vp(d), "-" + vp(creg(n)), vp(src), vp(d))
        lastw = [d, lastw[0]]
        cached = [None if (c is not None and c[0] == d) else c for c in cached]     # a fetched copy of d is stale now
    # ---- x <- alpha x~ + (1 - alpha) x; capturing: x_new into W (x stays x_prev)
    for ix in range(15):
        t = T_A[ix % 4]
        e("v_mul_f64", vp(t), sO, vp(QX(ix)))
        e("v_fma_f64", vp(QW(ix) if capture else QX(ix)), sA, vp(QW(ix)), vp(t))
    # ---- dynamics rows: delta_y = alpha (nu - y)
    for ix in range(12):
        t = T_A[(ix + 3) % 4]
        e("v_add_f64", vp(t), vp(QW(IX_EQ + ix)), "-" + vp(QY(ix)))
        if capture:
            e("v_mul_f64", vp(QW(IX_EQ + ix)), sA, vp(t))
        e("v_fma_f64", vp(QY(ix)), sA, vp(t), vp(QY(ix)))
    # ---- thrust rows (lanes 0..2 of one word)
    t1, t2, t3 = T_X[0], T_X[1], T_A[0]
    nu, y, z = QW(IX_T), QYT, QZT
    e("v_fma_f64", vp(t1), "-" + vp(y), vp(QRINV3), vp(z))
    e("v_fma_f64", vp(t1), vp(nu), vp(QRINV3), vp(t1))
    e("v_mul_f64", vp(t2), sO, vp(z))
    e("v_fma_f64", vp(t1), sA, vp(t1), vp(t2))
    e("v_fma_f64", vp(t3), vp(y), vp(QRINV3), vp(t1))
    e("v_max_f64", vp(t3), vp(t3), vp(QLO3))
    e("v_min_f64", vp(z), vp(t3), vp(QUP3))
    e("v_add_f64", vp(t2), vp(t1), "-" + vp(z))
    e("v_mul_f64", vp(t2), vp(t2), vp(QRHO3))
    e("v_add_f64", vp(y), vp(y), vp(t2))
    if capture:
        e("v_mov_b32", v(nu), v(t2))
        e("v_mov_b32", v(nu + 1), v(t2 + 1))


def exit_(e, plan, s):
    """x (new, in W), x_prev (in x), y, delta_y (in W) -> every lane's LDS slice; thrust-row z -> its one-lane registers"""
    from . import asmgen64 as g
    nx, nc, neq = s.nx, s.nc, 2 * s.N * symbolic.NY
    v = lambda n: "v%d" % n
    e("s_nop", 1)
    temps = list(T_COEF)            # eight distinct words (the solves are over: the ring is idle)
    cnt = [0]

    def put(word, src, ln):
        t = temps[cnt[0] % len(temps)]
        cnt[0] += 1
        e("v_mov_b32_dpp", v(t), v(src), qperm([ln] * 4))
        e("v_mov_b32_dpp", v(t + 1), v(src + 1), qperm([ln] * 4))
        b_, off, half = g.lds_addr(word)
        e("ds_write_b64", b_, vp(t), off + 8 * half)
    for j in range(nx):
        ln, ix = plan.xhome[j]
        put(g.LW_X + j, QW(ix), ln)
        put(g.PC_XP + j, QX(ix), ln)
    for i in range(nc):
        ln, ix = plan.zhome[i]
        if i < neq:
            put(g.LW_Y + i, QY(ix - IX_EQ), ln)
            put(g.PC_DY + i, QW(ix), ln)
        else:
            k = i - neq
            put(g.LW_Y + i, QYT, ln)
            put(g.PC_DY + i, QW(IX_T), ln)
            e("v_mov_b32_dpp", v(g.V_C + 2 * k), v(QZT), qperm([ln] * 4))
            e("v_mov_b32_dpp", v(g.V_C + 2 * k + 1), v(QZT + 1), qperm([ln] * 4))
    e("s_waitcnt", "lgkmcnt(0)")


def section(e, plan, s):
    sg = lambda n: "s%d" % n
    e("quad_begin",)
    entry(e, plan, s)
    e("s_sub_i32", sg(S_CNT), sg(S_ITERS), 2)
    e("s_cmp_lt_i32", sg(S_CNT), 1)
    e("s_cbranch_scc1", "18f")
    e("label", "17")
    body(e, plan, s, capture=False)
    e("s_sub_i32", sg(S_CNT), sg(S_CNT), 1)
    e("s_cmp_gt_i32", sg(S_CNT), 0)
    e("s_cbranch_scc1", "17b")
    e("label", "18")
    body(e, plan, s, capture=True)
    exit_(e, plan, s)
    e("quad_end",)


# ---------------------------------------------------------------------------------------------------------------------
# CPU interpreter of the section on the four lanes of a quad, exact-rounded float64
# ---------------------------------------------------------------------------------------------------------------------
def simulate(ins, pc, V, A, lds, S, tab, max_exec=400000):
    """ins[pc] == ("quad_begin",). V, A: uint32 [4][256]; lds: float64 [4][320], each lane's own slice by WORD; S: the
    calling interpreter's SGPR dict (constants as 64-bit pairs, S_ITERS); tab: table() of the plan. LDS addresses are
    taken relative to the lane's base (v1 must be equal in the four images: the caller runs with v1 = 0). Returns
    (pc behind ("quad_end",), executed instructions)."""
    from fractions import Fraction
    u32 = np.uint32
    exec_ = np.ones(4, bool)
    vcc = np.zeros(4, bool)
    masks, labels, scc = {}, {}, 0
    for k, t in enumerate(ins):
        if t[0] == "label":
            labels.setdefault(t[1], []).append(k)
    TAB_BASE = 1 << 44
    S[S_TAB], S[S_TAB + 1] = TAB_BASE & 0xFFFFFFFF, TAB_BASE >> 32
    vi = lambda x: int(x[1:])
    lohi = lambda x: int(x[2:x.index(":")])

    def f64(lo, hi):
        return struct.unpack("<d", struct.pack("<Q", int(lo) | (int(hi) << 32)))[0]

    def getd(x, ln):
        if isinstance(x, float):
            return x
        neg = x.startswith("-")
        if neg:
            x = x[1:]
        if x.startswith("|"):
            return (-1.0 if neg else 1.0) * abs(getd(x[1:-1], ln))
        lo = lohi(x)
        val = f64(V[ln, lo], V[ln, lo + 1]) if x[0] == "v" else f64(S[lo], S[lo + 1])
        return -val if neg else val

    def setd(x, ln, val):
        lo = lohi(x)
        b = struct.unpack("<Q", struct.pack("<d", float(val)))[0]
        V[ln, lo], V[ln, lo + 1] = b & 0xFFFFFFFF, b >> 32

    def fma(a, b, c):
        if not (np.isfinite(a) and np.isfinite(b) and np.isfinite(c)):
            return a * b + c
        return float(Fraction(a) * Fraction(b) + Fraction(c))

    def word_of(byte):
        assert byte % 8 == 0 and 0 <= byte < 160 * 1024
        return (byte // 1024) * 2 + (byte % 1024) // 8
    pend = []        # outstanding LDS reads: sets of (lane-independent) destination registers, in issue order
    nexec = 0
    assert ins[pc] == ("quad_begin",)
    pc += 1
    while ins[pc] != ("quad_end",):
        t = ins[pc]
        m = t[0]
        if m == "label":
            pc += 1
            continue
        nexec += 1
        assert nexec < max_exec, "runaway quad section"
        if m[0] == "v" or m.startswith("ds_"):
            used = set()
            for x in t[1:]:
                if isinstance(x, str):
                    y = x.lstrip("-")
                    if y.startswith("v["):
                        used |= set(range(lohi(y), int(y[y.index(":") + 1:-1]) + 1))
                    elif y[0] == "v" and y[1:].isdigit():
                        used.add(int(y[1:]))
            for dst in pend:
                assert not (dst & used), ("register used before its LDS read was waited for", t)
        if m == "s_waitcnt":
            for part in t[1].split():
                name, val = part[:-1].split("(")
                if name == "lgkmcnt":
                    del pend[:max(0, len(pend) - int(val))]
        elif m == "s_nop":
            pass
        elif m == "s_mov_b32":
            S[int(t[1][1:])] = (t[2] & 0xFFFFFFFF) if isinstance(t[2], int) else S.get(int(t[2][1:]), 0)
        elif m == "v_cmp_nlt_f64":
            assert t[1] == "vcc"
            for ln in range(4):
                if exec_[ln]:
                    vcc[ln] = not (getd(t[2], ln) < getd(t[3], ln))
        elif m == "v_cndmask_b32":
            assert t[4] == "vcc"
            for ln in range(4):
                if exec_[ln]:
                    a = u32(t[2]) if isinstance(t[2], int) else V[ln, vi(t[2])]
                    V[ln, vi(t[1])] = V[ln, vi(t[3])] if vcc[ln] else a
        elif m in ("v_rsq_f64", "v_rcp_f64"):
            for ln in range(4):
                if exec_[ln]:
                    a = getd(t[2], ln)
                    with np.errstate(all="ignore"):
                        setd(t[1], ln, (1.0 / np.sqrt(a)) if m == "v_rsq_f64" else (np.float64(1.0) / np.float64(a)))
        elif m == "s_mov_b64":
            if t[2] == "exec":
                masks[t[1]] = exec_.copy()
            else:
                assert t[1] == "exec"
                exec_ = masks[t[2]].copy()
        elif m == "s_and_b64":
            word = S[lohi(t[2])]
            masks[t[1]] = np.array([(word >> ln) & 1 for ln in range(4)], bool) & masks[t[3]]
        elif m == "s_sub_i32":
            a = S.get(int(t[2][1:]), 0) if isinstance(t[2], str) else t[2]
            b = S.get(int(t[3][1:]), 0) if isinstance(t[3], str) else t[3]
            S[int(t[1][1:])] = (a - b) & 0xFFFFFFFF
        elif m in ("s_cmp_lt_i32", "s_cmp_gt_i32"):
            sx = lambda x: (lambda w: w - (1 << 32) if w & 0x80000000 else w)(S.get(int(x[1:]), 0) if isinstance(x, str) else x & 0xFFFFFFFF)
            scc = int(sx(t[1]) < sx(t[2])) if m == "s_cmp_lt_i32" else int(sx(t[1]) > sx(t[2]))
        elif m == "s_cbranch_scc1":
            if scc:
                lab, d = t[1][:-1], t[1][-1]
                c = labels[lab]
                pc = min(x for x in c if x > pc) if d == "f" else max(x for x in c if x < pc)
        elif m == "v_bfe_u32":
            assert (t[3], t[4]) == (4, 2) and t[2] == "v1"
            for ln in range(4):
                if exec_[ln]:
                    V[ln, vi(t[1])] = ln            # (lane & 3): the images are the four lanes of one quad
        elif m == "v_mul_u32_u24":
            for ln in range(4):
                if exec_[ln]:
                    V[ln, vi(t[1])] = (t[2] * int(V[ln, vi(t[3])])) & 0xFFFFFFFF
        elif m == "v_add_u32":
            for ln in range(4):
                if exec_[ln]:
                    a = t[2] if isinstance(t[2], int) else int(V[ln, vi(t[2])])
                    V[ln, vi(t[1])] = (a + int(V[ln, vi(t[3])])) & 0xFFFFFFFF
        elif m == "v_mov_b32":
            for ln in range(4):
                if exec_[ln]:
                    V[ln, vi(t[1])] = u32(t[2]) if isinstance(t[2], int) else V[ln, vi(t[2])]
        elif m == "v_mov_b32_dpp":
            mod = t[3]
            qp = [int(c) for c in mod[mod.index("[") + 1:mod.index("]")].split(",")]
            old = V[:, vi(t[2])].copy()
            for ln in range(4):
                if exec_[ln]:
                    assert exec_[qp[ln]], "DPP read of a masked-off lane"
                    V[ln, vi(t[1])] = old[qp[ln]]
        elif m == "v_accvgpr_read_b32":
            V[exec_, vi(t[1])] = A[exec_, int(t[2][1:])]
        elif m == "v_accvgpr_write_b32":
            A[exec_, int(t[1][1:])] = V[exec_, vi(t[2])]
        elif m == "global_load_dwordx4":
            lo = lohi(t[1])
            off = int(t[4].split(":")[1])
            assert (S[S_TAB] | (S[S_TAB + 1] << 32)) == TAB_BASE
            for ln in range(4):
                if exec_[ln]:
                    byte = int(V[ln, vi(t[2])]) + off
                    assert byte % 4 == 0 and 0 <= byte // 4 + 3 < tab.size
                    V[ln, lo:lo + 4] = tab.ravel()[byte // 4:byte // 4 + 4]
        elif m == "ds_read_b128":
            lo = lohi(t[1])
            for ln in range(4):
                if exec_[ln]:
                    w = word_of(int(V[ln, vi(t[2])]) + t[3])
                    for h in range(2):
                        b = struct.unpack("<Q", struct.pack("<d", float(lds[ln, w + h])))[0]
                        V[ln, lo + 2 * h], V[ln, lo + 2 * h + 1] = b & 0xFFFFFFFF, b >> 32
            pend.append(set(range(lo, lo + 4)))
        elif m == "ds_read_b64":
            lo = lohi(t[1])
            for ln in range(4):
                if exec_[ln]:
                    w = word_of(int(V[ln, vi(t[2])]) + t[3])
                    b = struct.unpack("<Q", struct.pack("<d", float(lds[ln, w])))[0]
                    V[ln, lo], V[ln, lo + 1] = b & 0xFFFFFFFF, b >> 32
            pend.append({lo, lo + 1})
        elif m == "ds_write_b64":
            for ln in range(4):
                if exec_[ln]:
                    lds[ln, word_of(int(V[ln, vi(t[1])]) + t[3])] = getd(t[2], ln)
            pend.append(set())
        elif m == "ds_write_b128":
            lo = lohi(t[2])
            for ln in range(4):
                if exec_[ln]:
                    w = word_of(int(V[ln, vi(t[1])]) + t[3])
                    for h in range(2):
                        lds[ln, w + h] = f64(V[ln, lo + 2 * h], V[ln, lo + 2 * h + 1])
            pend.append(set())
        elif m in ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64"):
            for ln in range(4):
                if not exec_[ln]:
                    continue
                a, b = getd(t[2], ln), getd(t[3], ln)
                if m == "v_fma_f64":
                    r = fma(a, b, getd(t[4], ln))
                elif m == "v_mul_f64":
                    r = a * b
                elif m == "v_add_f64":
                    r = a + b
                else:
                    r = max(a, b) if m == "v_max_f64" else min(a, b)
                setd(t[1], ln, r)
        else:
            raise ValueError("unknown instruction in the fp64 quad section: %r" % (t,))
        pc += 1
    assert exec_.all() and not pend, "EXEC not restored / LDS reads outstanding at the end of the quad section"
    return pc + 1, nexec


# =====================================================================================================================
# The ten Ruiz passes of the fp64 step on the lane quad (the quad counterpart of asmgen64.ruiz_program)
# =====================================================================================================================
# The one-lane block spends 2 546 instructions per pass, most of them the 84 refined 1/sqrt and limit_scaling sequences --
# one per row and per column. On the quad a register holds the x / y / z members of a triple (or the three horizon steps of
# an input), so the same sequences run on 13 row registers and 15 column registers: a third of the work, and nothing
# crosses lanes except the nine input columns (thrust and the two moments of the three steps), whose entries sit in the
# lanes of their ROWS: their column norms are reduced over the quad with DPP moves and their scalings are fetched from the
# lane of the step. Every lane reads ITS entries of P, q and A straight from its own LDS slice (masked reads: all four
# slices hold the whole problem) and the scaled data go back to every slice at the end.
#   LDS words on entry and exit: as asmgen64.ruiz_program (RZ_P, RZ_Q, RZ_A; RZ_C on exit).
RQ_T = 4                   # 10 temporaries (words) v4..v23
RQ_C = 24                  # c (accumulated cost scaling)
RQ_ONEHI = 26              # high word of 1.0
RQ_ET = 28                 # 13 words: row scalings of this pass
RQ_P = 54                  # 15 words
RQ_Q = 84                  # 15 words
RQ_A = 114                 # the entry slots (<= 66 words to v245)
S_RMINS, S_RMAXS = 30, 32
S_RL0, S_RL1, S_RL2, S_REXEC = 34, 36, 38, 40
ROT1, ROT2 = [1, 2, 0, 3], [2, 0, 1, 3]


class RuizQuadPlan:
    def __init__(self, s, plan):
        self.s, self.plan = s, plan
        self.slots = {}          # (row register 0..12, column register 0..14) -> {row lane: entry}
        collane = {}             # slot -> {row lane: lane of the entry's column}
        for j in range(s.nx):
            lc, C = plan.xhome[j]
            for p in range(s.A_p[j], s.A_p[j + 1]):
                lr, R = plan.zhome[s.A_i[p]]
                key = (R - IX_EQ, C)
                d = self.slots.setdefault(key, {})
                assert lr not in d
                d[lr] = p
                collane.setdefault(key, {})[lr] = lc
        # "own": every entry's column lives in the entry's own lane; otherwise ONE lane k holds the column of every entry
        # of the slot (an input of horizon step k against the three members of a row triple of that step)
        self.kind = {}
        for key, cl in collane.items():
            if all(lc == lr for lr, lc in cl.items()):
                self.kind[key] = "own"
            else:
                ks = set(cl.values())
                assert len(ks) == 1, (key, cl)
                self.kind[key] = ks.pop()
        self.order = sorted(self.slots)
        self.reg = {key: RQ_A + 2 * n for n, key in enumerate(self.order)}
        assert RQ_A + 2 * len(self.order) <= 246, len(self.order)
        assert all((kd == "own") or C >= 12 for (R, C), kd in self.kind.items())


def ruiz_program(N=3, perm=None):
    """s11 = number of passes (>= 1); v1 = lane LDS address. Standalone block: the whole list runs on the quad."""
    from . import asmgen64 as g
    from .asmgen64 import RZ_A, RZ_C, RZ_P, RZ_Q, f64bits
    s = symbolic.analyse(N, perm)
    plan = plan_for(s)
    rp = RuizQuadPlan(s, plan)
    nx, nc = s.nx, s.nc
    e = g.Emit()
    v = lambda n: "v%d" % n
    T = lambda q: RQ_T + 2 * q
    ET = lambda R: RQ_ET + 2 * R
    P = lambda C: RQ_P + 2 * C
    Q = lambda C: RQ_Q + 2 * C
    sMIN, sMAX = sp(S_RMINS), sp(S_RMAXS)
    ab = lambda r: "|" + vp(r) + "|"
    masks = (S_RL0, S_RL1, S_RL2)

    def setc(reg, val):
        b = f64bits(val)
        e("s_mov_b32", "s%d" % reg, b & 0xFFFFFFFF)
        e("s_mov_b32", "s%d" % (reg + 1), b >> 32)

    def limit(t, t2):
        e("v_cmp_nlt_f64", "vcc", vp(t), sMIN)
        e("v_min_f64", vp(t2), vp(t), sMAX)
        e("v_cndmask_b32", v(t), 0, v(t2), "vcc")
        e("v_cndmask_b32", v(t + 1), v(RQ_ONEHI), v(t2 + 1), "vcc")

    def rsqrt(y, t, a_, h_):
        e("v_rsq_f64", vp(y), vp(t))
        e("s_nop", 0)
        for _ in range(2):
            e("v_mul_f64", vp(a_), vp(t), vp(y))
            e("v_fma_f64", vp(a_), "-" + vp(a_), vp(y), 1.0)
            e("v_mul_f64", vp(h_), 0.5, vp(y))
            e("v_fma_f64", vp(y), vp(h_), vp(a_), vp(y))

    def recip(y, t, a_):
        e("v_rcp_f64", vp(y), vp(t))
        e("s_nop", 0)
        for _ in range(2):
            e("v_fma_f64", vp(a_), "-" + vp(t), vp(y), 1.0)
            e("v_fma_f64", vp(y), vp(y), vp(a_), vp(y))

    def fetch(dst, src, perm):
        """dst <- src of the lanes `perm` selects (two 32-bit DPP moves)"""
        e("v_mov_b32_dpp", v(dst), v(src), qperm(perm))
        e("v_mov_b32_dpp", v(dst + 1), v(src + 1), qperm(perm))

    # ---- prologue: constants, masks, zero, then every lane's own entries from its LDS slice
    e("quad_begin",)
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    setc(S_RMINS, 1e-4)
    setc(S_RMAXS, 1e4)
    e("v_add_u32", v(g.V_B1), 0x10000, "v1")
    e("v_add_u32", v(g.V_B2), 0x20000, "v1")
    e("s_mov_b64", sp(S_REXEC), "exec")
    for ln, m in enumerate(masks):
        e("s_mov_b32", "s%d" % m, 0x11111111 << ln)
        e("s_mov_b32", "s%d" % (m + 1), 0x11111111 << ln)
        e("s_and_b64", sp(m), sp(m), sp(S_REXEC))
    words = [P(C) for C in range(15)] + [Q(C) for C in range(15)] + [rp.reg[k] for k in rp.order]
    for r in words:
        e("v_mov_b32", v(r), 0)
        e("v_mov_b32", v(r + 1), 0)
    b1 = f64bits(1.0)
    e("v_mov_b32", v(RQ_C), b1 & 0xFFFFFFFF)
    e("v_mov_b32", v(RQ_C + 1), b1 >> 32)
    e("v_mov_b32", v(RQ_ONEHI), b1 >> 32)
    loads = {0: [], 1: [], 2: []}
    for j in range(nx):
        ln, C = plan.xhome[j]
        loads[ln].append((P(C), RZ_P + j))
        loads[ln].append((Q(C), RZ_Q + j))
    for key in rp.order:
        for ln, p in rp.slots[key].items():
            loads[ln].append((rp.reg[key], RZ_A + p))
    for ln in range(3):
        e("s_mov_b64", "exec", sp(masks[ln]))
        for k, (dst, word) in enumerate(loads[ln]):
            b_, off, half = g.lds_addr(word)
            e("ds_read_b64", vp(dst), b_, off + 8 * half)
            if k % 12 == 11:
                e("s_waitcnt", "lgkmcnt(0)")
        e("s_waitcnt", "lgkmcnt(0)")
    e("s_mov_b64", "exec", sp(S_REXEC))
    e("s_mov_b32", "s%d" % S_CNT, "s%d" % S_ITERS)
    e("label", "27")
    # ---- row scalings
    byrow = {}
    for key in rp.order:
        byrow.setdefault(key[0], []).append(rp.reg[key])
    for R in range(13):
        t, regs = T(0), byrow[R]
        if len(regs) == 1:
            e("v_max_f64", vp(t), ab(regs[0]), ab(regs[0]))
        else:
            e("v_max_f64", vp(t), ab(regs[0]), ab(regs[1]))
            for r in regs[2:]:
                e("v_max_f64", vp(t), vp(t), ab(r))
        limit(t, T(1))
        rsqrt(ET(R), t, T(2), T(3))
    # ---- columns
    for r in (T(4), T(5)):
        e("v_mov_b32", v(r), 0)
        e("v_mov_b32", v(r + 1), 0)
    bycol = {}
    for key in rp.order:
        bycol.setdefault(key[1], []).append(key)
    for C in range(15):
        t, dt, pn = T(0), T(1), T(6)
        own = [k for k in bycol.get(C, []) if rp.kind[k] == "own"]
        far = [k for k in bycol.get(C, []) if rp.kind[k] != "own"]
        e("v_mul_f64", vp(pn), vp(P(C)), vp(RQ_C))            # the norms see c * P_j (the cost scaling is carried as a scalar)
        first = True
        for k in own:
            e("v_max_f64", vp(t), ab(pn if first else t), ab(rp.reg[k]))
            first = False
        if first:
            e("v_max_f64", vp(t), ab(pn), ab(pn))
        for k in far:          # the column of step rp.kind[k]: its entries sit in the three ROW lanes of slot k
            r = rp.reg[k]
            fetch(T(7), r, ROT1)
            fetch(T(8), r, ROT2)
            e("v_max_f64", vp(T(7)), ab(T(7)), ab(T(8)))
            e("v_max_f64", vp(T(7)), vp(T(7)), ab(r))          # max over lanes 0..2, in each of them
            e("s_mov_b64", "exec", sp(masks[rp.kind[k]]))
            e("v_max_f64", vp(t), vp(t), vp(T(7)))
            e("s_mov_b64", "exec", sp(S_REXEC))
        limit(t, dt)
        rsqrt(dt, t, T(2), T(3))
        e("v_mul_f64", vp(P(C)), vp(P(C)), vp(dt))
        e("v_mul_f64", vp(P(C)), vp(P(C)), vp(dt))
        e("v_add_f64", vp(T(4)), vp(T(4)), ab(P(C)))
        for k in own:
            e("v_mul_f64", vp(rp.reg[k]), vp(rp.reg[k]), vp(ET(k[0])))
            e("v_mul_f64", vp(rp.reg[k]), vp(rp.reg[k]), vp(dt))
        for k in far:
            e("s_nop", 1)
            fetch(T(7), dt, [rp.kind[k]] * 4)
            e("v_mul_f64", vp(rp.reg[k]), vp(rp.reg[k]), vp(ET(k[0])))
            e("v_mul_f64", vp(rp.reg[k]), vp(rp.reg[k]), vp(T(7)))
        e("v_mul_f64", vp(Q(C)), vp(Q(C)), vp(dt))
        e("v_max_f64", vp(T(5)), vp(T(5)), ab(Q(C)))
    # ---- sum |P_j| and max |q_j| over the quad: a butterfly, so that ALL FOUR lanes end with the same value (lane 3 owns
    # nothing but runs the phases around this block like the others and must read the same c)
    e("s_nop", 1)
    for perm_ in ([1, 0, 3, 2], [2, 3, 0, 1]):
        fetch(T(7), T(4), perm_)
        fetch(T(8), T(5), perm_)
        e("v_add_f64", vp(T(4)), vp(T(4)), vp(T(7)))
        e("v_max_f64", vp(T(5)), vp(T(5)), vp(T(8)))
        e("s_nop", 1)
    # ---- cost scaling: ct = 1 / limit(max(pmean / nx, limit(qn))), as the one-lane block
    b = f64bits(float(nx))
    e("v_mov_b32", v(T(0)), b & 0xFFFFFFFF)
    e("v_mov_b32", v(T(0) + 1), b >> 32)
    recip(T(1), T(0), T(2))
    e("v_mul_f64", vp(T(4)), vp(T(4)), vp(RQ_C))
    e("v_mul_f64", vp(T(5)), vp(T(5)), vp(RQ_C))
    e("v_mul_f64", vp(T(2)), vp(T(4)), vp(T(1)))
    e("v_fma_f64", vp(T(3)), "-" + vp(T(0)), vp(T(2)), vp(T(4)))
    e("v_fma_f64", vp(T(4)), vp(T(3)), vp(T(1)), vp(T(2)))
    limit(T(5), T(0))
    e("v_max_f64", vp(T(4)), vp(T(4)), vp(T(5)))
    limit(T(4), T(0))
    recip(T(5), T(4), T(0))
    e("v_mul_f64", vp(RQ_C), vp(RQ_C), vp(T(5)))
    e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_CNT, 1)
    e("s_cmp_gt_i32", "s%d" % S_CNT, 0)
    e("s_cbranch_scc1", "27b")
    # ---- epilogue: c once into P and q; every word back to every lane's slice
    for C in range(15):
        e("v_mul_f64", vp(P(C)), vp(P(C)), vp(RQ_C))
        e("v_mul_f64", vp(Q(C)), vp(Q(C)), vp(RQ_C))
    e("s_nop", 1)
    cnt = [0]

    def put(word, src, ln):
        t = T(cnt[0] % 10)
        cnt[0] += 1
        fetch(t, src, [ln] * 4)
        b_, off, half = g.lds_addr(word)
        e("ds_write_b64", b_, vp(t), off + 8 * half)
    for j in range(nx):
        ln, C = plan.xhome[j]
        put(RZ_P + j, P(C), ln)
        put(RZ_Q + j, Q(C), ln)
    for key in rp.order:
        for ln, p in rp.slots[key].items():
            put(RZ_A + p, rp.reg[key], ln)
    b_, off, half = g.lds_addr(RZ_C)
    e("ds_write_b64", b_, vp(RQ_C), off + 8 * half)
    e("s_waitcnt", "lgkmcnt(0)")
    e("quad_end",)
    return e.ins, s

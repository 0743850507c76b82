"""Generated gfx950 assembly for the MIDDLE ADMM iterations of a build-time-known, chain-structured batch QP (BASELINE
config 4: planar p5f, n = 87, m = 164, nnz(L) = 264), fp32, one lane per robot, one workgroup per CU (B <= 16 384).

Why: the straight-line C++ specialisation (codegen_qp.py) keeps ~3 000 words per robot in compiler-managed storage; a
lone wave per CU walks them as a chain of exposed scratch round trips -- 0.21 ms per iteration, 10.5 of the 12.4 ms tick
(tools/p5f_split.sh). A lane owns 256 VGPRs + 256 AGPRs + 640 LDS words when its wave has the CU to itself:

    W (KKT rhs / solution)   VGPRs, only the NON-LEAF unknowns (162 of 251): a constraint row whose KKT unknown is a leaf
                             of the elimination tree (one L entry, nothing eliminated into it: the 84 box rows and 5
                             single-entry dynamics rows of p5f) never gets a register -- its rhs is formed when its row is
                             visited, pushed into its variable's unknown, and its multiplier is re-formed in the row
                             update from the final value of that unknown (same operations, different moment)
    1/D                      AGPRs (251), one v_accvgpr_read per use
    L, x, y, z(inequality)   LDS, float4-interleaved (264 + 87 + 164 + 87 = 602 words); L stored NEGATED in order of use
    q, l(dynamics rows)      never stored: loaded from the stream buffer straight INTO the W registers they are combined
                             with (W_x <- q, W_z <- l), as asmgen64.py
    l, u, 1/rho, rho of the inequality rows: streamed too (4 words per row per iteration, plus 1/rho once more for the
                             rhs), through 36 landing registers filled ~8 rows ahead
    stream buffer            [wave][item][lane] floats: one iteration's read-only words in the exact order the loop
                             consumes them, 256 B per item, so one 64-bit base + the 12-bit instruction offset address 16
                             items (two SALU instructions per 16 loads)

Rows in `eq_rows` (p5f: the 77 dynamics rows) are taken to be equalities (l == u, rho = rho_eq): after the first iteration
z == l there and the row update is delta_y = alpha (nu - y) (as asmgen.py). The caller (codegen_qp.py's kernel) runs the
FIRST and the LAST iteration in C++ (warm-start z, capture of x_prev / delta_y) and checks the equality assumption on
device, taking the C++ loop for a wave where it does not hold.

Arithmetic vs the C++ statement: fused multiply-adds in the solves, leaf contributions applied before the other
columns, the equality-row shortcut: rounding only. `simulate()` interprets the stream on numpy float32 (tests/test_asmqp.py).
Reference mapping: osqp 0.6.0 auxil.c:164-228, qdldl.c:250-293, proj.c:4-14 (planar/mpc_osqp_p5f.py never calls
solve(); the iteration is the build's).

Round 4: the kernel's workgroup is FOUR wavefronts (one per SIMD of the CU) that own the same 64 robots and LDS slots and
divide every block's instructions -- RuizSplit / ruiz_group_program (stretches of columns), glue_group_program (rows),
LoopSplit / loop_group_program (the QP's connected components: independent QPs) and res_group_program (quarters of the rows
and columns). Same layouts, every word bit-identical to the one-wavefront blocks; simulate_group() runs the wavefronts barrier
phase by barrier phase on one LDS image and rejects a word written by one and touched by another between two barriers.

Round 5: two of the four wavefronts were nearly idle in the loop (the QP is two long chains and five crumbs). A chain is now CUT
IN TWO (LoopSplit; the elimination order comes bisected from qpstruct.bisect_ordering): each half's subtree of the elimination
tree is one wavefront's, the separator (one or two unknowns) goes with half A, the solves meet at two barriers per iteration.
The W registers a wavefront no longer needs hold its entries of L and its other loop constants (Own.lhome / chome / yhome): with
four wavefronts at work the CU's one LDS pipe is the second bottleneck. Still bit-identical to the one-wavefront block. And
asmgen.Emit keeps the one hazard nobody checks for inline assembly out of every stream (a VALU write to the data registers of a
ds_write_b128 within two wait states)."""
import os
import struct

import numpy as np

from .asmgen import Emit, f32bits, pk as _pk

HERE = os.path.dirname(os.path.abspath(__file__))

# SGPR inputs / temporaries
S_W, S_S, S_STRIDE, S_ITERS = 4, 6, 10, 11          # s[4:5] row workspace, s[6:7] this wave's stream block, 4*B, iterations
S_P, S_CNT, S_SP = 12, 14, 16                       # row pointer, loop counter, stream pointer
S_ALPHA, S_OMA, S_SIGMA, S_RINVEQ = 20, 21, 22, 23  # floats (inputs)
S_RHO0, S_RINV0, S_RHOEQ = 31, 34, 35               # rho, 1/rho, rho_eq (float bits, inputs): the classes update_rho_vec assigns
RHO_MIN_F32 = 1e-6                                  # (auxil.c:103-145; the third class, both bounds infinite, is a constant)
V_B1, V_B2, V_LANE, V_W = 2, 3, 4, 5     # inputs: v0 = 4*robot, v1 = lane LDS address (16*lane), v4 = 4*lane
NRING, NLAND, N_AT, N_TT = 6, 36, 4, 8
V_END = 246
BLOCK = 16                                          # stream items per pointer bump (16 x 256 B = the offset field's reach)
LW_FLAGS = 636                                      # LDS words 636..639: LOOSE_FLAG, GLUE_FLAG, FAC_MIN, RES_FLAG (between the blocks)
# The LOOSE variant of the loop (program(..., loose=True)): every inequality row of the wave is in the third class of
# update_rho_vec (auxil.c:103-145: both scaled bounds beyond +-1e30 * 1e-4, rho = RHO_MIN) -- the reference's planar p5f
# problem has xmin = umin = -inf, xmax = umax = +inf (planar/mpc_osqp_p5f.py:94-97), so ALL 87 of its inequality rows are.
# Their rho and 1/rho are then two constants and the projection onto [l, u] never clips a finite value: the variant takes
# them from SGPRs, drops v_max / v_min and streams NOTHING per inequality row (the general loop streams l, u, 1/rho, rho:
# 4-5 loads per row and iteration). Same operations on the same values: bit-identical results. The glue block decides
# per wave (LDS word LOOSE_FLAG).
S_RIMIN, S_RHOMIN = 36, 37                            # 1 / RHO_MIN, RHO_MIN as float bits (set by the loose program itself)
Y0_VARIANT = os.environ.get("UMPC_QP_Y0", "1") == "1"   # the loose program carries the y == 0 variant of its loop (program())
Y0_DLEAF = os.environ.get("UMPC_QP_Y0_DLEAF", "1") == "1"    # ... with 1/d of the leaf loose rows in an SGPR (S_DLEAF)
Y0_FUSE = os.environ.get("UMPC_QP_Y0_FUSE", "1") == "1"      # ... and the updates forming the next iteration's right-hand side
S_DLEAF = 38


def lds_addr(word):
    byte = (word >> 2) * 1024
    return "v%d" % (1, V_B1, V_B2)[byte >> 16], (byte & 0xFFFF) + 4 * (word & 3)


# generation-time switches (experiments; the defaults are what ships): where the bounds of the leaf equality rows live
# ('V': registers, 'A': the AGPRs behind 1/D then LDS), 1/rho of some inequality rows resident in the free LDS words,
# rho selected instead of streamed
PACK_PARTS = os.environ.get("UMPC_QP_PACK_PARTS", "1234")     # (diagnostics) 1: rhs of x, 2: 1/D scaling, 3: x update, 4: equality rows
PACK_LOOSE = os.environ.get("UMPC_QP_PACK_LOOSE", "1") == "1"      # packed row updates in the loose variant (A/B switch)
KNOB = dict(leafeq=os.environ.get("UMPC_QP_LEAFEQ", "V"), rinv_lds=os.environ.get("UMPC_QP_RINV_LDS", "1") == "1",
            rho_select=os.environ.get("UMPC_QP_RHO_SELECT", "0") == "1")


class Plan:
    def __init__(self, s, eq_rows, pad=None):
        if pad is None:
            # one unused word between L and x or none: whichever parity of the x / y words lets more row pairs share packed
            # instructions (a box row pair needs its y words AND its variables' W registers on aligned pairs)
            best = max((Plan(s, eq_rows, q) for q in (0, 1)), key=lambda c: (len(c.pairs), len(c.eqpairs), -c.pad))
            self.__dict__.update(best.__dict__)
            return
        self.pad = pad
        t = s.tables
        n, m, nk = s.n, s.m, s.nk
        self.s, self.n, self.m, self.nk = s, n, m, nk
        pinv, L_p, L_i, Lr_p = list(t["pinv"]), list(t["L_p"]), list(t["L_i"]), list(t["Lr_p"])
        self.pinv, self.L_p, self.L_i = pinv, L_p, L_i
        eq = set(int(i) for i in eq_rows)
        self.eq_rows_set = eq
        self.rows = []
        leafk = set()
        for i in range(m):
            k = pinv[n + i]
            leaf = (L_p[k + 1] - L_p[k] == 1) and (Lr_p[k + 1] - Lr_p[k] == 0)
            r = dict(i=i, k=k, eq=i in eq, leaf=leaf)
            if leaf:
                r["j"] = L_p[k]
                r["r"] = L_i[L_p[k]]
                leafk.add(k)
            self.rows.append(r)
        for j in range(n):
            assert pinv[j] not in leafk
        self.nonleaf = [k for k in range(nk) if k not in leafk]
        # W registers: the unknowns of the variables x_0 .. x_{n-1} on consecutive registers from an EVEN base (x_j, x_j+1 an
        # aligned pair for even j: the packed row updates of the loose loop variant take two box rows per instruction),
        # the other non-leaf unknowns around them
        xk = [pinv[j] for j in range(n)]
        xbase = V_W + (V_W % 2)
        self.wreg = {k: xbase + j for j, k in enumerate(xk)}
        # ... and the unknowns of consecutive non-leaf EQUALITY rows (i, i + 1) whose y words are an aligned pair (LW_Y + i
        # even) on aligned register pairs too: their row updates and right-hand sides pack the same way
        LW_Y_ = len(L_i) + pad + n
        nl_eq = {r["i"]: r["k"] for r in self.rows if r["eq"] and not r["leaf"]}
        self.eqpairs, usede = [], set()
        for i in sorted(nl_eq):
            if i not in usede and i + 1 in nl_eq and (LW_Y_ + i) % 2 == 0:
                self.eqpairs.append((i, i + 1))
                usede.update((i, i + 1))
        paired_k = [nl_eq[i] for ab in self.eqpairs for i in ab]
        rest = [k for k in self.nonleaf if k not in set(xk) and k not in set(paired_k)]
        free = [r_ for r_ in range(V_W, V_W + len(self.nonleaf)) if r_ not in set(self.wreg.values())]
        evens = [r_ for r_ in free if r_ % 2 == 0 and r_ + 1 in free]
        taken, kept = set(), []
        for q in range(0, len(paired_k), 2):
            base = next((r_ for r_ in evens if r_ not in taken and r_ + 1 not in taken), None)
            if base is None:                                  # no aligned pair left: these two rows stay unpaired
                rest += [paired_k[q], paired_k[q + 1]]
                continue
            taken.update((base, base + 1))
            self.wreg[paired_k[q]], self.wreg[paired_k[q + 1]] = base, base + 1
            kept.append(self.eqpairs[q // 2])
        self.eqpairs = kept
        for k, r_ in zip(rest, [r_ for r_ in free if r_ not in taken]):
            self.wreg[k] = r_
        assert sorted(self.wreg.values()) == list(range(V_W, V_W + len(self.nonleaf)))
        for r in self.rows:
            if r["leaf"]:
                assert r["r"] in self.wreg, "a leaf row must hang off a non-leaf unknown"
        # (generation-time experiment UMPC_QP_RHO_SELECT=1: rho of a row selected from the three class constants by comparing
        # its streamed 1/rho instead of streaming it -- 20 % fewer stream items, 4 more VALU per row: slower, the loop is
        # bound by instruction issue; off by default)
        self.V_RHO0 = V_W + len(self.nonleaf)
        self.V_RHOEQ, self.V_RHOMIN = self.V_RHO0 + 1, self.V_RHO0 + 2
        self.leafeq = [r["i"] for r in self.rows if r["leaf"] and r["eq"]]
        self.V_LEQ = self.V_RHO0 + (3 if KNOB["rho_select"] else 0)
        self.V_RING = self.V_LEQ + (len(self.leafeq) if KNOB["leafeq"] == "V" else 0)
        self.V_LAND = self.V_RING + 4 * NRING
        self.NLAND = V_END - N_TT - N_AT - self.V_LAND        # every register left over lands stream items
        assert self.NLAND >= 16
        self.V_AT = self.V_LAND + self.NLAND
        self.V_TT = self.V_AT + N_AT
        assert self.V_TT + N_TT <= V_END, (self.V_TT + N_TT, V_END)
        # L storage: leaf entries in row order, then the solve entries in forward order (walked backwards by the
        # backward solve)
        self.solve_entries = [(L_i[j], c, j) for c in self.nonleaf for j in range(L_p[c], L_p[c + 1])]
        # Pairs of leaf inequality rows (i, i + 1) for the packed row updates of the loose variant: their y words are an
        # aligned pair when LW_Y + i is even, their variables' W registers (x_j, x_j+1 on consecutive registers) likewise;
        # their z words and L entries are PLACED on aligned pairs below (paired rows first in both orders).
        self.LW_L, self.LW_X = 0, len(L_i) + pad
        self.LW_Y = self.LW_X + n
        cand = {r["i"]: r for r in self.rows if r["leaf"] and not r["eq"]}
        self.pairs, used = [], set()
        for i in sorted(cand):
            if i in used or i + 1 not in cand or (self.LW_Y + i) % 2:
                continue
            a_, b_ = cand[i], cand[i + 1]
            wa, wb = self.wreg[a_["r"]], self.wreg[b_["r"]]
            if wa % 2 == 0 and wb == wa + 1:
                self.pairs.append((a_, b_))
                used.update((i, i + 1))
        pflat = [r for ab in self.pairs for r in ab]
        order = [r["j"] for r in pflat] + [r["j"] for r in self.rows if r["leaf"] and not r["eq"] and r["i"] not in used] + \
                [r["j"] for r in self.rows if r["leaf"] and r["eq"]] + [j for (_, _, j) in self.solve_entries]
        assert sorted(order) == list(range(len(L_i)))
        self.lpos = {j: p for p, j in enumerate(order)}
        self.LW_Z = self.LW_Y + m + (self.LW_Y + m) % 2          # even: (z_i, z_i+1) of a row pair share an aligned pair
        gen = [r["i"] for r in pflat] + [r["i"] for r in self.rows if not r["eq"] and r["i"] not in used]
        self.zpos = {i: q for q, i in enumerate(gen)}
        gen = sorted(gen)                                     # (the stream below is consumed in row order)
        self.LW_END = self.LW_Z + len(gen)
        assert self.LW_END <= 640
        # Words that are constant over the iterations and find a home on chip are loaded ONCE (prologue) instead of streamed
        # every iteration: the scaled bounds of the leaf equality rows -> the AGPRs behind 1/D, then LDS; 1/rho of as many
        # inequality rows as LDS words are left (each is read twice per iteration: rhs and row update).
        self.once = {}                                         # item -> ('A', agpr) | ('L', lds word)
        free_a = list(range(nk, 256))
        free_l = list(range(self.LW_END, LW_XCH))
        for q, i in enumerate(self.leafeq):
            if KNOB["leafeq"] == "V":
                self.once[("l", i)] = ("V", self.V_LEQ + q)
            else:
                self.once[("l", i)] = ("A", free_a.pop(0)) if free_a else ("L", free_l.pop(0))
        for i in gen:
            if free_l and KNOB["rinv_lds"]:
                self.once[("rinv", i)] = ("L", free_l.pop(0))
        # stream items of one iteration, in consumption order (rho is not streamed: see V_RHO0)
        self.stream = [("rinv", i) for i in gen if ("rinv", i) not in self.once]
        for i in gen:
            self.stream += [("l", i), ("u", i)] + ([("rinv", i)] if ("rinv", i) not in self.once else []) + \
                           ([("rho", i)] if not KNOB["rho_select"] else [])
        self.n_land = len(self.stream)
        self.stream += [("q", j) for j in range(n)]
        self.stream += [("l", r["i"]) for r in self.rows if r["eq"] and not r["leaf"]]
        self.n_stream = len(self.stream)
        self.extra = list(self.once)                           # loaded once (prologue), after the per-iteration items
        # The y0 variant of the loose loop (see program()): the multipliers of the loose rows are exactly zero and need no
        # LDS word, so q takes the y words of the inequality rows (in row order: q_j, q_j+1 stay an aligned pair where the
        # rows' y words are), and l of non-leaf equality rows takes the words behind z that the general loop keeps 1/rho
        # in (aligned pairs for the packed row pairs first). Items without a home stay in the per-iteration preloads.
        self.ineq = gen
        self.y0_home = {}
        for j, i in zip(range(n), gen):
            self.y0_home[("q", j)] = self.LW_Y + i
        spare = list(range(self.LW_END + self.LW_END % 2, LW_XCH))
        for (a_, b_) in self.eqpairs:
            if len(spare) >= 2:
                self.y0_home[("l", a_)], self.y0_home[("l", b_)] = spare[0], spare[1]
                spare = spare[2:]
        for item in self.stream[self.n_land:]:
            if spare and item not in self.y0_home:
                self.y0_home[item] = spare.pop(0)
        # ... and 1/d of a leaf loose row is ONE number (its unknown is eliminated first: d = -1/rho_min, nothing folded in),
        # which the y0 loop keeps in an SGPR; the AGPRs of those rows take the items that are still homeless
        self.y0_dleaf = [r["k"] for r in self.rows if r["leaf"] and not r["eq"]]
        apool = list(self.y0_dleaf) if Y0_DLEAF else []
        for (a_, b_) in self.eqpairs:
            if ("l", a_) not in self.y0_home and len(apool) >= 2:
                self.y0_home[("l", a_)], self.y0_home[("l", b_)] = ("A", apool.pop(0)), ("A", apool.pop(0))
        for item in self.stream[self.n_land:]:
            if apool and item not in self.y0_home:
                self.y0_home[item] = ("A", apool.pop(0))
        # row-major hand-off rows (floats, [row][B])
        self.R_L, self.R_DI = 0, len(L_i)
        self.R_X = self.R_DI + nk
        self.R_Y = self.R_X + n
        self.R_Z = self.R_Y + m
        self.R_END = self.R_Z + len(gen)
        # x_prev and delta_y of the last (capturing) iteration go to LDS words of L that are dead by then: delta_y over
        # the solve block (dead after the backward solve), x_prev over the leaf block (dead after the row updates)
        self.n_leaf = sum(r["leaf"] for r in self.rows)
        self.LW_DY, self.LW_XP = self.n_leaf, 0
        assert self.LW_DY + m <= self.LW_X and self.LW_XP + n <= self.n_leaf + m


# distance (instructions) after which an LDS / stream access is taken to have completed when a wait is placed: the wait
# then covers it too and the later wait for it is not emitted (0: every wait covers exactly what its op needs)
MERGE_LDS = int(os.environ.get("UMPC_QP_MERGE_LDS", "16"))
MERGE_VM = int(os.environ.get("UMPC_QP_MERGE_VM", "150"))
# generation-time experiment: the first NT_ITEMS landing items of an iteration are loaded with the non-temporal hint, so that
# the rest of the stream (re-read every iteration) can stay in the XCD's L2 instead of the whole of it cycling through
NT_ITEMS = int(os.environ.get("UMPC_QP_NT", "0"))
NT_KINDS = tuple(k for k in os.environ.get("UMPC_QP_NT_KINDS", "").split(",") if k)     # "", "rows", "stream", "rows,stream"
RING_AHEAD = int(os.environ.get("UMPC_QP_RING_AHEAD", "20"))   # ops of look-ahead for the LDS ring reads


class Own:
    """Which variables, rows and KKT unknowns a wavefront's copy of the loop block works on (LoopSplit); ALL = one wave, all."""

    def __init__(self, varw=None, roww=None, kw=None, wave=0, lhome=None, fkw=None, split=None, fvarw=None):
        self.varw, self.roww, self.kw, self.wave = varw, roww, kw, wave
        self.all = varw is None
        self.lhome = lhome or {}          # L entry (CSC index) -> VGPR that holds -L during the iterations (LoopSplit.own)
        self.chome, self.yhome = {}, {}   # LDS word of a loop constant -> VGPR: leaf entries of L (every variant); q, l (y0 bodies)
        self.ahome, self.shome = {}, {}   # fused y0 bodies (S_HOMES): LDS word of a constant -> AGPR; of the wave's x / y / z -> VGPR
        self.dhome = {}                   # ... and 1/D of own unknowns -> VGPR (registers that are idle in those bodies)
        self.ghome = {}                   # bodies that are not y0 (G_HOMES): stream item -> AGPR (what they loaded every iteration)
        self.fkw = fkw if fkw is not None else kw          # who FACTORISES unknown k (a component cut in two: the wave of half A)
        self.fvarw = fvarw if fvarw is not None else varw
        # a component cut in two (LoopSplit): the solves of every wave of the workgroup meet at two barriers
        self.xbar = split is not None
        sp = split or {}
        self.top = sp.get("top", set())          # unknowns of the separators this wave owns
        self.cross = sp.get("cross", [])         # solve entries (r, c, j), c the partner's, r in `top`: this wave's forward solve applies them
        self.xrecv_f = sp.get("xrecv_f", {})     # ... reading the partner's forward value of c from this LDS word
        self.xsend_f = sp.get("xsend_f", {})     # unknown c -> LDS word this wave leaves its forward value in
        self.xsend_b = sp.get("xsend_b", {})     # separator unknown r -> LDS word this wave leaves its solution in
        self.xrecv_b = sp.get("xrecv_b", {})     # separator unknown r (the partner's) -> LDS word
        self.hand_out = sp.get("hand_out", {})   # unknown k this wave factorises for its partner -> LDS word 1/D_k is handed over in
        self.hand_in = sp.get("hand_in", {})     # unknown k of this wave that the partner factorises -> that word
        self.ftop = sp.get("ftop", set())        # separator unknowns this wave factorises AFTER its partner's subtree (FACTOR_SPLIT)

    def var(self, j):
        return self.all or self.varw[j] == self.wave

    def row(self, i):
        return self.all or self.roww[i] == self.wave

    def k(self, k):
        return self.all or self.kw[k] == self.wave

    def fk(self, k):
        return self.all or self.fkw[k] == self.wave

    def fvar(self, j):
        return self.all or self.fvarw[j] == self.wave

    def item(self, item):
        what, idx = item
        return self.var(idx) if what == "q" else self.row(idx)


ALL = Own()
L_HOMES = os.environ.get("UMPC_QP_L_HOMES", "1") == "1"        # (A/B switch: LoopSplit.own)
FACTOR_SPLIT = os.environ.get("UMPC_QP_FACTOR_SPLIT", "1") == "1"   # (A/B switch: a cut component's halves factorise their own subtrees)
G_HOMES = os.environ.get("UMPC_QP_G_HOMES", "1") == "1"        # (A/B switch: LoopSplit.own, the general loop's stream items in AGPRs)
D_HOMES = os.environ.get("UMPC_QP_D_HOMES", "1") == "1"        # (A/B switch: LoopSplit.own, 1/D of own unknowns in idle VGPRs)
Y_NRING = 2                                                     # ring slots the fused y0 bodies of such a wave may use
S_HOMES = os.environ.get("UMPC_QP_S_HOMES", "1") == "1"        # (A/B switch: LoopSplit.own, the iterates in VGPRs, the constants in AGPRs)
C_HOMES = os.environ.get("UMPC_QP_C_HOMES", "1") == "1"        # (A/B switch: LoopSplit.own, the loop's other constants in VGPRs)
TREE_SPLIT = os.environ.get("UMPC_QP_TREE_SPLIT", "1") == "1"  # (A/B switch: LoopSplit cuts large components in two)
SCALE_LATE = os.environ.get("UMPC_QP_SCALE_LATE", "1") == "1"  # (A/B switch: where a cut component's W / D sits between the barriers)
N_XCH = 16                                                      # LDS words LW_XCH.. below the flags: the halves' exchange words
LW_XCH = LW_FLAGS - N_XCH


class LoopSplit:
    """A QP whose constraint graph falls into several connected components (qpstruct.qp_components) is several independent
    QPs: right-hand sides, LDL' factor, triangular solves, row and x updates of one component never touch another's words.
    The loop block deals the components out to the wavefronts of the workgroup, largest first, each to the least loaded; they
    run their 50 iterations side by side on disjoint LDS words of the SAME layout (Plan). Words come out bit-identical to the
    one-wave block.

    Round 5: a component larger than a wavefront's fair share whose elimination tree is two subtrees under a short top chain
    (qpstruct.bisect_ordering makes the horizon chains of an MPC such trees) is cut in TWO units: half A with the top chain
    (the separator), half B. Everything elementwise belongs to the owner of its unknown; the triangular solves meet twice:
        forward   both halves eliminate their subtree; B leaves the forward values of its columns that reach into the separator
                  in LDS -- barrier -- A applies those entries (the same fmac on the same operands, in the one-wave order: the
                  ordering puts A's columns before B's), eliminates the separator columns, scales, solves them back
        backward  A leaves the separator's solution in LDS -- barrier -- both halves walk their subtree back
    Two barriers per iteration for every wavefront of the workgroup (a wavefront without a cut component just meets them).
    The factorisation of a cut component stays with ONE wavefront (A's; it hands 1/D of B's unknowns over through LDS once).
    planar p5f: 118 + 112 + 12 + 9 unknowns on four wavefronts became 63 + 60 + 64 + 64."""

    def __init__(self, p, nw=4, tree=None):
        from . import qpstruct
        s = p.s
        self.p = p
        tree = TREE_SPLIT if tree is None else tree
        vc, rc = qpstruct.qp_components(s.n, s.m, list(s.tables["A_p"]), list(s.tables["A_i"]))
        self.nw = nw
        ncomp = max(vc) + 1
        comp = [0] * s.nk
        for j in range(s.n):
            comp[p.pinv[j]] = vc[j]
        for i in range(s.m):
            comp[p.pinv[s.n + i]] = rc[i]
        size = [sum(1 for x in comp if x == c) for c in range(ncomp)]
        # units: whole components, or the two halves of a cut one
        self.cut = {}
        units = []
        for c in range(ncomp):
            cut = (self._given(comp, c) or self._cut(comp, c)) if tree and nw >= 2 and size[c] * nw > s.nk else None
            if cut is None:
                units.append((size[c], c, None))
            else:
                self.cut[c] = cut
                units.append((len(cut["A"]) + len(cut["T"]), c, "A"))
                units.append((len(cut["B"]), c, "B"))
        units.sort(key=lambda u: (-u[0], u[1], u[2] or ""))
        nact = min(nw, len(units))
        load, uw = [0] * nact, {}
        for (wt, c, half) in units:
            other = uw.get((c, "A" if half == "B" else "B")) if half else None
            w = min((w_ for w_ in range(nact) if w_ != other), key=lambda w_: (load[w_], w_))
            uw[(c, half)] = w
            load[w] += wt
        self.active = nact
        self.load = load
        self.kw, self.fkw = [0] * s.nk, [0] * s.nk
        for k in range(s.nk):
            c = comp[k]
            if c in self.cut:
                self.kw[k] = uw[(c, "B")] if k in self.cut[c]["B"] else uw[(c, "A")]
                self.fkw[k] = self.kw[k] if FACTOR_SPLIT else uw[(c, "A")]
            else:
                self.kw[k] = self.fkw[k] = uw[(c, None)]
        self.varw = [self.kw[p.pinv[j]] for j in range(s.n)]
        self.roww = [self.kw[p.pinv[s.n + i]] for i in range(s.m)]
        for r in p.rows:
            if r["leaf"]:
                assert self.kw[r["r"]] == self.roww[r["i"]]
        # exchange words and the entries that cross a cut
        self.split = [dict(top=set(), cross=[], xrecv_f={}, xsend_f={}, xsend_b={}, xrecv_b={}, hand_out={}, hand_in={})
                      for _ in range(nact)] if self.cut else None
        self.fvarw = [self.fkw[p.pinv[j]] for j in range(s.n)]
        # 1/D that changes hands once per tick. One wave factorises the whole of a cut component (FACTOR_SPLIT off): 1/D of half B's
        # unknowns goes from its AGPRs to B's through LDS words that held the component's A entries. The halves factorise their own
        # subtrees (default): only 1/D of B's columns that reach into the separator goes to A, through the exchange words.
        A_p_ = list(s.tables["A_p"])
        for a_ in range(nact if self.cut else 0):
            words = [p.LW_X + q for j in range(s.n) if self.fvarw[j] == a_ for q in range(A_p_[j], A_p_[j + 1])]
            ks = [k for k in range(s.nk) if self.fkw[k] == a_ and self.kw[k] != a_]
            assert len(ks) <= len(words)
            for k, word in zip(ks, words):
                self.split[a_]["hand_out"][k] = word
                self.split[self.kw[k]]["hand_in"][k] = word
        quad = LW_XCH
        for c in sorted(self.cut):
            a_, b_ = uw[(c, "A")], uw[(c, "B")]
            T = self.cut[c]["T"]
            self.split[a_]["top"] |= T
            cross = [(r_, c_, j) for (r_, c_, j) in p.solve_entries if self.kw[c_] == b_ and self.kw[r_] == a_]
            assert all(r_ in T for (r_, _, _) in cross)
            self.split[a_]["cross"] += cross
            cols, tops = sorted({c_ for (_, c_, _) in cross}), sorted({r_ for (r_, _, _) in cross})
            assert len(cols) <= 4 and len(tops) <= 4 and quad + 8 <= LW_FLAGS, (cols, tops, quad)
            for q, c_ in enumerate(cols):          # (one float4 per direction and cut: one LDS read fetches it)
                self.split[a_]["xrecv_f"][c_] = self.split[b_]["xsend_f"][c_] = quad + q
            for q, r_ in enumerate(tops):
                self.split[a_]["xsend_b"][r_] = self.split[b_]["xrecv_b"][r_] = quad + 4 + q
            if FACTOR_SPLIT:
                for q, c_ in enumerate(cols):       # (the forward words of the solves, idle during the factorisation)
                    self.split[b_]["hand_out"][c_] = self.split[a_]["hand_in"][c_] = quad + q
                self.split[a_]["ftop"] = self.split[a_].get("ftop", set()) | T
            quad += 8
        for (r_, c_, j) in p.solve_entries:
            assert self.kw[r_] == self.kw[c_] or any((r_, c_, j) in sp_["cross"] for sp_ in self.split or [])

    def _given(self, comp, c):
        """the halves and the separator that the structure's ordering was made from (qpstruct.bisect_parts, tables["part"]), if it
        carries them and they satisfy what the solves need (see _cut); a leaf row goes where its parent is"""
        p = self.p
        s = p.s
        part = s.tables.get("part")
        if part is None:
            return None
        K = [k for k in range(s.nk) if comp[k] == c]
        of = {k: part[s.perm[k]] for k in K}
        for r in p.rows:
            if r["leaf"] and r["k"] in of:
                of[r["k"]] = 0 if of[r["r"]] == 2 else of[r["r"]]
        A, B, T = ({k for k in K if of[k] == h} for h in (0, 1, 2))
        if not T or not B:
            return None
        seen_b = set()
        for (r_, c_, j) in p.solve_entries:
            if c_ not in of:
                continue
            if r_ in T:
                if c_ in B:
                    seen_b.add(r_)
                elif c_ in A and r_ in seen_b:
                    return None
            if (c_ in A and r_ in B) or (c_ in B and r_ in A) or (c_ in T and r_ not in T):
                return None
        return dict(A=A, B=B, T=T)

    def _cut(self, comp, c):
        """component c as (half A, half B, top chain T) of its elimination tree, or None: T = the nodes from the root down while
        one child's subtree is more than 55 % of the component; the subtrees under T are dealt to two bins, largest first; a
        leaf row stays with its parent. Accepted when the smaller half is at least a quarter and, for every unknown of T, the
        solve entries of A's columns come before those of B's columns (the one-wave order of the additions)."""
        p = self.p
        s = p.s
        etree = list(s.etree)
        K = [k for k in range(s.nk) if comp[k] == c]
        kids = {k: [] for k in K}
        roots = []
        for k in K:
            (kids[etree[k]] if etree[k] != -1 else roots).append(k)
        if len(roots) != 1:
            return None
        sz = {}
        for k in K:                               # (children have smaller indices than their parent)
            sz[k] = 1 + sum(sz[x] for x in kids[k])
        leafk = {r["k"] for r in p.rows if r["leaf"]}

        def deal(T, frontier):
            T = set(T)
            bins = [set(), set()]
            for k in sorted(frontier, key=lambda k: (-sz[k], k)):
                sub, st = set(), [k]
                while st:
                    x = st.pop()
                    sub.add(x)
                    st += kids[x]
                b = 0 if (k in leafk and etree[k] in T) else (0 if len(bins[0]) + len(T) <= len(bins[1]) else 1)
                bins[b] |= sub
            A, B = bins
            if min(len(A) + len(T), len(B)) < 0.25 * len(K):
                return None
            seen_b = set()
            for (r_, c_, j) in p.solve_entries:
                if r_ in T:
                    if c_ in B:
                        seen_b.add(r_)
                    elif c_ in A and r_ in seen_b:
                        return None
                if (c_ in A and r_ in B) or (c_ in B and r_ in A) or (c_ in T and r_ not in T):
                    return None
            return dict(A=A, B=B, T=T)
        # every prefix of the chain root -> largest child -> ... (a few nodes) is a candidate top; the best balanced valid one wins
        T, frontier, cands = [roots[0]], list(kids[roots[0]]), []
        for _ in range(6):
            c_ = deal(T, frontier)
            if c_ is not None:
                cands.append((max(len(c_["A"]) + len(c_["T"]), len(c_["B"])), len(T), c_))
            if not frontier:
                break
            big = max(frontier, key=lambda k: (sz[k], k))
            T = T + [big]
            frontier = [k for k in frontier if k != big] + kids[big]
        return min(cands, key=lambda c: c[:2])[2] if cands else None

    def own(self, wave):
        """... and the W registers of the OTHER wavefronts' unknowns, which this wavefront's copy of the block never touches, hold
        its solve entries of L during the iterations: each is used twice per iteration (forward, backward solve), and an LDS
        instruction costs a lone wave ~6 ns whatever its width"""
        p = self.p
        pool = sorted(p.wreg[k] for k in p.nonleaf if self.kw[k] != wave)
        cross = set(self.split[wave]["cross"]) if self.split else set()
        mine = [j for (r_, c, j) in p.solve_entries if self.kw[c] == wave or (r_, c, j) in cross]
        lhome = dict(zip(mine, pool)) if L_HOMES else {}
        o = Own(self.varw, self.roww, self.kw, wave, lhome, self.fkw, self.split[wave] if self.split else None, self.fvarw)
        if G_HOMES and self.split:
            # The bodies of the GENERAL loop (finite bounds) and of the loose loop with non-zero multipliers loaded their read-only
            # items -- l, u, 1/rho, rho of the wave's inequality rows, q, l of its equality rows -- from the stream EVERY iteration
            # (150 VMEM instructions per wavefront-iteration). A wave of a cut workgroup uses a quarter of the 251 AGPR indices for
            # its 1/D: the items go into the others once per tick.
            mine = [it for it in p.stream if o.item(it) and it not in p.once]
            mine = list(dict.fromkeys(mine))
            taken = {k for k in range(p.nk) if self.kw[k] == wave} | set(self.split[wave]["hand_in"]) | \
                {h[1] for h in p.once.values() if h[0] == "A"} | set(self.split[wave].get("top", set()))
            apool_g = [k for k in range(256) if k not in taken]
            if len(mine) <= len(apool_g):
                o.ghome = dict(zip(mine, apool_g))
        if C_HOMES and L_HOMES:
            # ... and what is left of them the wave's other loop constants, which the bodies read from LDS every iteration (with all
            # four wavefronts at work the LDS pipe is the second bottleneck: 4 array cycles per float4 read, 13 per float4 write): the leaf rows'
            # entries of L, then (y0 bodies) q and l of the equality rows. A pair of words that the packed operations read as a
            # pair gets an aligned register pair.
            free = [r_ for r_ in pool if r_ not in set(lhome.values())]
            lw = [p.lpos[r["j"]] for r in p.rows if r["leaf"] and self.roww[r["i"]] == wave]
            qw = [h for (what, j), h in sorted(p.y0_home.items(), key=lambda kv: kv[1] if isinstance(kv[1], int) else -1)
                  if what == "q" and isinstance(h, int) and self.varw[j] == wave]
            ew = [h for (what, i), h in sorted(p.y0_home.items(), key=lambda kv: kv[1] if isinstance(kv[1], int) else -1)
                  if what == "l" and isinstance(h, int) and self.roww[i] == wave]
            for words, dst in ((lw, o.chome), (qw, o.yhome), (ew, o.yhome)):
                ws = set(words)
                for w_ in sorted(ws):
                    if w_ in dst:
                        continue
                    if w_ % 2 == 0 and w_ + 1 in ws:
                        base = next((r_ for r_ in free if r_ % 2 == 0 and r_ + 1 in free), None)
                        if base is None:
                            continue              # (no aligned pair left: these two stay in LDS)
                        free.remove(base)
                        free.remove(base + 1)
                        dst[w_], dst[w_ + 1] = base, base + 1
                    elif not (w_ % 2 == 1 and w_ - 1 in ws):
                        odd = [r_ for r_ in free if not (r_ % 2 == 0 and r_ + 1 in free) and not (r_ % 2 == 1 and r_ - 1 in free)]
                        if not (odd or free):
                            continue
                        r_ = (odd or free)[0]     # (singles first into registers that cannot form an aligned pair)
                        free.remove(r_)
                        dst[w_] = r_
            if S_HOMES and self.split:
                # The fused y0 bodies (the reference's problem: every iteration but the last) go one step further: the constants
                # move on into AGPRs that this wave's 1/D does not use (a v_accvgpr_read costs a VALU slot and no LDS cycle), and
                # the registers they held -- filled only on this path -- keep the wave's x, z and equality-row y words: no LDS
                # access at all for them between the first and the last iteration.
                topall = self.split[wave]["topall"] if "topall" in self.split[wave] else self.split[wave]["top"]
                apool = [k for k in p.nonleaf if self.kw[k] != wave and k not in topall and k not in self.split[wave]["hand_in"]]
                consts = sorted(set(lw) | set(qw) | set(ew))
                eqs = set(int(i) for i in p.eq_rows_set)
                sw = sorted([p.LW_X + j for j in range(p.n) if self.varw[j] == wave] +
                            [p.LW_Z + p.zpos[i] for i in p.zpos if self.roww[i] == wave] +
                            [p.LW_Y + i for i in range(p.m) if i in eqs and self.roww[i] == wave])
                free2 = [r_ for r_ in pool if r_ not in set(lhome.values())]
                sh, ok = {}, len(consts) <= len(apool)
                for w_ in sw:
                    if w_ in sh:
                        continue
                    if w_ % 2 == 0 and w_ + 1 in sw:
                        base = next((r_ for r_ in free2 if r_ % 2 == 0 and r_ + 1 in free2), None)
                        if base is None:
                            ok = False
                            break
                        free2.remove(base)
                        free2.remove(base + 1)
                        sh[w_], sh[w_ + 1] = base, base + 1
                    elif not (w_ % 2 == 1 and w_ - 1 in sw):
                        odd = [r_ for r_ in free2 if not (r_ % 2 == 0 and r_ + 1 in free2) and not (r_ % 2 == 1 and r_ - 1 in free2)]
                        if not (odd or free2):
                            ok = False
                            break
                        r_ = (odd or free2)[0]
                        free2.remove(r_)
                        sh[w_] = r_
                if ok:                # (all or nothing: a body either finds every word of the wave resident or none)
                    o.shome = sh
                    o.ahome = dict(zip(consts, apool))
                    if D_HOMES:
                        # what is idle in those bodies -- the registers left over, four of the six ring slots (the only LDS reads
                        # are the exchange words), the AGPR temporaries of the unpacked bodies -- keeps 1/D of the wave's own
                        # unknowns, pair for pair as the packed scaling reads them
                        spare = sorted(free2 + list(range(p.V_RING + 4 * Y_NRING, p.V_RING + 4 * NRING)) + list(range(p.V_AT, p.V_AT + N_AT)))
                        mine_k = sorted((p.wreg[k], k) for k in p.nonleaf if self.kw[k] == wave)
                        regk = dict(mine_k)
                        for reg, k in mine_k:
                            if k in o.dhome:
                                continue
                            if reg % 2 == 0 and reg + 1 in regk:
                                base = next((r_ for r_ in spare if r_ % 2 == 0 and r_ + 1 in spare), None)
                                if base is not None:
                                    spare.remove(base)
                                    spare.remove(base + 1)
                                    o.dhome[k], o.dhome[regk[reg + 1]] = base, base + 1
                            elif not (reg % 2 == 1 and reg - 1 in regk):
                                odd = [r_ for r_ in spare if not (r_ % 2 == 0 and r_ + 1 in spare) and not (r_ % 2 == 1 and r_ - 1 in spare)]
                                if odd or spare:
                                    r_ = (odd or spare)[0]
                                    spare.remove(r_)
                                    o.dhome[k] = r_
        return o


class Sched:
    """Emits a list of ops with their operand fetches: LDS quads through a ring (prefetched `ahead` ops before first use),
    AGPR words two ops ahead, stream items through the landing registers. op = dict(srcs=[...], emit=fn(regs)),
    src = ('L', lds word) | ('A', agpr) | ('S', stream item) | ('V', vgpr). emit may call lds_write()."""

    def __init__(self, e, plan, vm_outstanding, la=3, land_map=None):
        self.e, self.p, self.la = e, plan, la    # la: ops of look-ahead for AGPR reads (0: right before use)
        # land_map: this wave lands a SUBSET of the stream's landing items (positions in the stream, increasing)
        self.land_map = land_map if land_map is not None else getattr(plan, "land_map", None)
        self.n_land = len(self.land_map) if self.land_map is not None else getattr(plan, "n_land", 0)
        self.nlds = 0                       # LDS instructions issued so far (reads and writes complete in order)
        self.nvm = vm_outstanding           # VMEM loads issued so far; the first `vm_outstanding` are the preloads
        self.vmpos = {}                     # stream item -> its load's issue index
        self.s_issued = 0                   # landing items issued
        self.nland = getattr(plan, "NLAND", NLAND)
        self.sp_block = 0
        self.lds_at, self.vm_at = {}, {}    # issue position -> instruction index (for merging waits, see MERGE_*)
        self.vm_done = -1                   # VMEM loads up to this issue index are known to have arrived
        self.at_base, self.n_at = getattr(plan, "V_AT", None), N_AT      # temporaries of the AGPR reads (even base for 'A2')
        self.nring = NRING                  # ring slots this run may use (the rest may hold something else: Own.dhome)

    def vm_wait(self, pos):
        """the VMEM load with issue index `pos` must have arrived (loads arrive in order): emits a wait unless an earlier one
        already implies it; retires with it the later loads that are old enough to have arrived anyway (MERGE_VM)"""
        if pos <= self.vm_done:
            return
        now = len(self.e.ins)
        while MERGE_VM and pos + 1 < self.nvm and self.vm_at.get(pos + 1, -10 ** 9) <= now - MERGE_VM:
            pos += 1
        c = min(63, self.nvm - 1 - pos)
        self.e("s_waitcnt", "vmcnt(%d)" % c)
        self.vm_done = self.nvm - 1 - c

    def lds_write(self, word, reg):
        base, off = lds_addr(word)
        self.lds_at[self.nlds] = len(self.e.ins)
        self.e("ds_write_b32", base, "v%d" % reg, off)
        self.nlds += 1

    def lds_write2(self, word, reg):
        """words `word`, `word + 1` (even word: the same float4 of the lane) <- the aligned pair v[reg:reg+1]"""
        assert word % 2 == 0 and reg % 2 == 0
        base, off = lds_addr(word)
        self.lds_at[self.nlds] = len(self.e.ins)
        self.e("ds_write_b64", base, "v[%d:%d]" % (reg, reg + 1), off)
        self.nlds += 1

    def lds_write4(self, word, reg):
        """words word .. word + 3 (one float4 of the lane) <- v[reg:reg+3] (even reg)"""
        assert word % 4 == 0 and reg % 2 == 0
        base, off = lds_addr(word)
        self.lds_at[self.nlds] = len(self.e.ins)
        self.e("ds_write_b128", base, "v[%d:%d]" % (reg, reg + 3), off)
        self.nlds += 1

    def issue_stream(self, idx):
        e, p = self.e, self.p
        item = self.land_map[idx] if self.land_map is not None else idx     # (a wave that lands a subset of the items)
        if item // BLOCK > self.sp_block:
            e("s_add_u32", "s%d" % S_SP, "s%d" % S_SP, (item // BLOCK - self.sp_block) * BLOCK * 256)
            e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_SP + 1), 0)
            self.sp_block = item // BLOCK
        e("global_load_dword", "v%d" % (p.V_LAND + idx % self.nland), "v%d" % V_LANE, "s[%d:%d]" % (S_SP, S_SP + 1),
          (item % BLOCK) * 256, *(["nt"] if idx < NT_ITEMS else []))
        self.vmpos[idx] = self.nvm
        self.vm_at[self.nvm] = len(e.ins)
        self.nvm += 1

    def run(self, ops):
        e, p = self.e, self.p
        n = len(ops)
        # ---- static LDS ring analysis (a 'flush' op ends every residency: later reads see the words written since)
        NR = self.nring
        insts, slots, inst_of = [], [None] * NR, {}
        for i, op in enumerate(ops):
            if op.get("flush"):
                slots = [None] * NR
                continue
            for q, src in enumerate(op["srcs"]):
                if src[0] != "L":
                    continue
                qd = src[1] >> 2
                hit = [k for k in slots if k is not None and insts[k]["quad"] == qd]
                if hit:
                    insts[hit[0]]["last"] = i
                    inst_of[(i, q)] = hit[0]
                    continue
                free = [sl for sl in range(NR) if slots[sl] is None]
                if free:
                    sl, prev = free[0], None
                    # a slot emptied by a flush still holds an older instance's registers until its last use
                    olds = [k for k, it in enumerate(insts) if it["slot"] == sl]
                    prev = olds[-1] if olds else None
                else:
                    sl = min(range(NR), key=lambda z: insts[slots[z]]["last"])
                    prev = slots[sl]
                insts.append(dict(quad=qd, first=i, last=i, slot=sl, prev=prev, issued=None))
                slots[sl] = len(insts) - 1
                inst_of[(i, q)] = slots[sl]
        next_inst, next_acc, acc_rr = 0, 0, 0
        atemp = {}
        first_item = [min([src[1] for src in op.get("srcs", []) if src[0] == "S"], default=None) for op in ops]
        waited_vm, waited_lds = -1, -1

        def issue(it):
            base, off = lds_addr(4 * it["quad"])
            r = p.V_RING + 4 * it["slot"]
            self.lds_at[self.nlds] = len(e.ins)
            e("ds_read_b128", "v[%d:%d]" % (r, r + 3), base, off)
            it["issued"] = self.nlds
            self.nlds += 1
        for i in range(n):
            op = ops[i]
            if op.get("flush"):
                continue
            while next_acc < n and next_acc < i + max(1, self.la):
                for q, src in enumerate(ops[next_acc].get("srcs", [])):
                    if src[0] == "A":
                        t = self.at_base + acc_rr % self.n_at
                        acc_rr += 1
                        e("v_accvgpr_read_b32", "v%d" % t, "a%d" % src[1])
                        atemp[(next_acc, q)] = t
                    elif src[0] == "A2":                  # two AGPR words into an aligned temporary pair
                        acc_rr += acc_rr % 2
                        t = self.at_base + acc_rr % self.n_at
                        acc_rr += 2
                        assert t % 2 == 0 and self.n_at % 2 == 0
                        e("v_accvgpr_read_b32", "v%d" % t, "a%d" % src[1])
                        e("v_accvgpr_read_b32", "v%d" % (t + 1), "a%d" % src[2])
                        atemp[(next_acc, q)] = t
                next_acc += 1
            # stream: keep the landing registers full ahead of the consumer
            nxt = next((f for f in first_item[i:] if f is not None), None)
            if nxt is not None:
                while self.s_issued < self.n_land and self.s_issued < nxt + self.nland - 4:
                    self.issue_stream(self.s_issued)
                    self.s_issued += 1
            while next_inst < len(insts):
                it = insts[next_inst]
                prev = insts[it["prev"]] if it["prev"] is not None else None
                if it["first"] <= i + RING_AHEAD and (prev is None or prev["last"] < i):
                    # (never fetched across a flush: the words may be rewritten before it)
                    if any(ops[z].get("flush") for z in range(i, it["first"])):
                        break
                    issue(it)
                    next_inst += 1
                else:
                    break
            regs = []
            for q, src in enumerate(op["srcs"]):
                if src[0] == "V":
                    regs.append(src[1])
                elif src[0] in ("A", "A2"):
                    regs.append(atemp.pop((i, q)))
                elif src[0] == "S":
                    idx = src[1]
                    assert idx in self.vmpos, idx
                    if idx > waited_vm:
                        # one wait also covers the other items of this op (a wait costs an issue slot of the lone wave)
                        last = max(s_[1] for s_ in op["srcs"] if s_[0] == "S")
                        self.vm_wait(self.vmpos[last])
                        waited_vm = last
                    regs.append(p.V_LAND + idx % self.nland)
                else:
                    it = insts[inst_of[(i, q)]]
                    if it["issued"] is None:
                        assert inst_of[(i, q)] == next_inst
                        issue(it)
                        next_inst += 1
                    if it["issued"] > waited_lds:
                        J, now = it["issued"], len(e.ins)
                        while MERGE_LDS and J + 1 < self.nlds and self.lds_at[J + 1] <= now - MERGE_LDS:
                            J += 1
                        e("s_waitcnt", "lgkmcnt(%d)" % min(15, self.nlds - 1 - J))
                        waited_lds = J
                    regs.append(p.V_RING + 4 * it["slot"] + (src[1] & 3))
            op["emit"](regs)


def preloads(e, p, homes=(), own=ALL):
    """the tail of the stream: q -> W_x, l of the non-leaf equality rows -> their W registers (in place operands);
    homes: items that live in LDS in this variant (y0) and are not loaded"""
    idx = p.n_land - 1
    blk = None
    for (what, q) in p.stream[p.n_land:]:
        idx += 1
        if (what, q) in homes or not own.item((what, q)):
            continue
        if idx // BLOCK != blk:
            # pointer = this wave's block + (idx // BLOCK) * 4096
            e("s_add_u32", "s%d" % S_SP, "s%d" % S_S, (idx // BLOCK) * BLOCK * 256)
            e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_S + 1), 0)
            blk = idx // BLOCK
        k = p.pinv[q] if what == "q" else p.pinv[p.n + q]
        e("global_load_dword", "v%d" % p.wreg[k], "v%d" % V_LANE, "s[%d:%d]" % (S_SP, S_SP + 1), (idx % BLOCK) * 256)


def couples(words):
    """words: the (even) LDS words pair-wise results are written to, in emission order. Two consecutive pairs that fill one
    aligned float4 are written with ONE ds_write_b128 (an LDS instruction costs a lone wave ~6 ns whatever its width):
    returns ({word of the first pair of a couple}, {word of the second})"""
    first, second = set(), set()
    for a_, b_ in zip(words, words[1:]):
        if a_ % 4 == 0 and b_ == a_ + 2 and a_ not in second:
            first.add(a_)
            second.add(b_)
    return first, second


PAIR_QUADS = os.environ.get("UMPC_QP_PAIR_QUADS", "1") == "1"


def body(e, p, capture=False, loose=False, y0=False, rhs=True, fuse=False, own=ALL, group=False):
    """capture: the LAST iteration -- x_prev and delta_y go to rows R_XP / R_DY (auxil.c:362-512 consumes them)
    loose: every inequality row is a loose row (see S_RIMIN above)
    rhs / fuse (y0 bodies): fuse -- the row and x updates leave the NEXT iteration's right-hand side in the W registers (the new
    x, y, z words are in registers there: no second LDS read, and a leaf row's L entry is fetched once instead of twice);
    rhs=False -- the iteration before did that, this body starts at the solves"""
    n, m = p.n, p.m
    v = lambda r: "v%d" % r
    sA, sO, sS, sRe = ("s%d" % r for r in (S_ALPHA, S_OMA, S_SIGMA, S_RINVEQ))
    W = lambda k: v(p.wreg[k])
    T = lambda q: p.V_TT + q
    assert loose or not y0
    gh = {} if y0 else own.ghome             # (bodies that are not y0: the stream items of the wave in AGPRs, filled once per tick)
    homes = p.y0_home if y0 else {it: ("A", a_) for it, a_ in gh.items() if it in set(p.stream[p.n_land:])}
    # Where this body finds a loop constant (C: one word, C2: an aligned pair of words) and a word of x / y / z (S, S2):
    #   fused y0 bodies of a wave with resident iterates (Own.shome): constants in AGPRs (Own.ahome), x / y / z in VGPRs;
    #   the capturing body behind them: everything from LDS (the iterates are written back first, the registers are theirs);
    #   every other body: the constants' VGPR homes (Own.chome: leaf entries of L; Own.yhome, y0 bodies: q, l), iterates in LDS
    yreg = y0 and bool(own.shome)
    yfuse = yreg and fuse
    sh = own.shome if yfuse else {}
    if yfuse:
        C = lambda word: ("A", own.ahome[word]) if word in own.ahome else ("L", word)
        C2 = lambda word: ("A2", own.ahome[word], own.ahome[word + 1]) if (word in own.ahome and word + 1 in own.ahome) else ("L", word)
    elif yreg:
        C = C2 = lambda word: ("L", word)
    else:
        C = C2 = lambda word: ("V", own.chome[word]) if word in own.chome else ("V", own.yhome[word]) if (y0 and word in own.yhome) else ("L", word)
    S = S2 = lambda word: ("V", sh[word]) if word in sh else ("L", word)
    dh = own.dhome if yfuse else {}          # 1/D of own unknowns in VGPRs (registers idle in these bodies, ring slots among them)
    DI = lambda k: ("V", dh[k]) if k in dh else ("A", k)
    DI2 = lambda k, k1: ("V", dh[k]) if (k in dh and dh.get(k1) == dh[k] + 1 and dh[k] % 2 == 0) else ("A2", k, k1)
    pre_items = [it for it in p.stream[p.n_land:] if it not in homes and own.item(it)]
    npre = len(pre_items)
    own_land = None if own.all else [q for q, it in enumerate(p.stream[:p.n_land]) if own.item(it) and it not in gh]
    sc = Sched(e, p, npre, la=1 if yfuse else 3, land_map=own_land)     # (constants in AGPRs: up to six fetches per operation, eight temporaries)
    if dh:
        sc.nring = Y_NRING
    if gh:                                   # (up to five AGPR operands per row update; the landing registers are idle: nothing lands)
        sc.la, sc.at_base, sc.n_at = 1, p.V_LAND, 8
    e("s_mov_b64", "s[%d:%d]" % (S_SP, S_SP + 1), "s[%d:%d]" % (S_S, S_S + 1))
    ops = []

    def op(srcs, fn):
        ops.append(dict(srcs=srcs, emit=fn))
    pre_pos = {}                         # W register preloaded -> index of its load among the preloads
    for q, (what, idx) in enumerate(pre_items):
        pre_pos[p.wreg[p.pinv[idx] if what == "q" else p.pinv[n + idx]]] = q

    def wait_pre(reg):
        sc.vm_wait(pre_pos[reg])
    rhs_ops_start = len(ops)
    # ---- P1: W_x = sigma x - q (q preloaded)
    pack = loose and not capture and PACK_LOOSE          # the loose variant's middle iterations: two entries per instruction
    VP = lambda r_: ("v[%d:%d]" % (r_, r_ + 1), 0, 1)
    SB = lambda sreg, h: ("s[%d:%d]" % (sreg - sreg % 2, sreg - sreg % 2 + 1), h, h)
    xpair = lambda j: pack and j + 1 < n and (p.LW_X + j) % 2 == 0 and p.wreg[p.pinv[j]] % 2 == 0 and \
        p.wreg[p.pinv[j + 1]] == p.wreg[p.pinv[j]] + 1 and own.var(j) and own.var(j + 1)
    jskip = set()
    for j in range(n):
        k = p.pinv[j]
        if j in jskip or not own.var(j):
            continue
        qh = homes.get(("q", j))
        if isinstance(qh, tuple):
            op([S(p.LW_X + j), qh], lambda r, k=k: e("v_fma_f32", W(k), sS, v(r[0]), "-" + v(r[1])))
            continue
        if qh is not None and "1" in PACK_PARTS and xpair(j) and qh % 2 == 0 and homes.get(("q", j + 1)) == qh + 1:
            jskip.add(j + 1)
            op([S2(p.LW_X + j), C2(qh)], lambda r, j=j: _pk(e, "v_pk_fma_f32", p.wreg[p.pinv[j]],
                                                                  [SB(S_SIGMA, S_SIGMA % 2), VP(r[0]), VP(r[1])], [0, 0, 1]))
            continue
        if qh is not None:
            op([S(p.LW_X + j), C(qh)], lambda r, k=k: e("v_fma_f32", W(k), sS, v(r[0]), "-" + v(r[1])))
            continue
        if "1" in PACK_PARTS and xpair(j) and ("q", j + 1) not in homes:
            jskip.add(j + 1)

            def f2x(r, j=j):
                wr = p.wreg[p.pinv[j]]
                wait_pre(wr)
                wait_pre(wr + 1)
                _pk(e, "v_pk_fma_f32", wr, [SB(S_SIGMA, S_SIGMA % 2), VP(r[0]), VP(wr)], [0, 0, 1])
            op([("L", p.LW_X + j)], f2x)
            continue

        def f(r, k=k):
            wait_pre(p.wreg[k])
            e("v_fma_f32", W(k), sS, v(r[0]), "-" + W(k))
        op([("L", p.LW_X + j)], f)
    # ---- P2/P3: rhs of the rows; leaf rows push theirs into their variable's unknown
    land = [0]

    def src_of(item):
        """a constant word: its on-chip home (Plan.once) or the next landing item"""
        if item in p.once:
            return p.once[item]
        if item in gh:
            return ("A", gh[item])
        assert p.stream[land[0] if own_land is None else own_land[land[0]]] == item, (item, land[0])
        land[0] += 1
        return ("S", land[0] - 1)
    # ---- the loose variant's middle iterations take the leaf inequality rows (p5f: the 87 box rows) TWO per packed
    # instruction: rows i, i+1 whose y words, z words, L entries and variables' W registers are aligned pairs (Plan lays
    # them out that way). The capturing iteration keeps the scalar form (delta_y words are not pair-aligned).
    VP = lambda r_: ("v[%d:%d]" % (r_, r_ + 1), 0, 1)
    SB = lambda sreg, h: ("s[%d:%d]" % (sreg - sreg % 2, sreg - sreg % 2 + 1), h, h)
    eqfirst = {a_ for (a_, b_) in p.eqpairs if p.wreg[p.rows[a_]["k"]] % 2 == 0 and p.wreg[p.rows[b_]["k"]] == p.wreg[p.rows[a_]["k"]] + 1
               and own.row(a_) and own.row(b_)}
    eqskip = set()
    paired = {}
    if loose and not capture and PACK_LOOSE:
        for a_, b_ in p.pairs:
            if not (own.row(a_["i"]) and own.row(b_["i"])):
                continue                  # (a pair across two waves' components: each wave takes its row alone)
            assert (p.LW_Y + a_["i"]) % 2 == 0 and (p.LW_Z + p.zpos[a_["i"]]) % 2 == 0 and p.zpos[b_["i"]] == p.zpos[a_["i"]] + 1
            assert p.lpos[a_["j"]] % 2 == 0 and p.lpos[b_["j"]] == p.lpos[a_["j"]] + 1
            paired[a_["i"]] = b_
            paired[b_["i"]] = None            # handled with its partner
        sc.at_base, sc.n_at = p.V_LAND, 8         # (the landing registers are idle in this variant) pairs of 1/D words
    TPK = lambda set_, q: p.V_LAND + 8 + 8 * (set_ % 3) + q          # three sets of packed temporaries, also landing registers
    assert not paired or p.NLAND >= 32
    for r in p.rows:
        i, k = r["i"], r["k"]
        if not own.row(i):
            continue
        if i in paired:
            if paired[i] is None:
                continue
            rb = paired[i]

            def f2(g, r=r, cnt=len(ops)):
                t = T(2 * (cnt % 4))                 # (rotating temporaries: no back-to-back reuse of one pair)
                _pk(e, "v_pk_fma_f32", t, [SB(S_RIMIN, 0), VP(g[0]), VP(g[1])], [1, 0, 0])      # z - y / rho
                wr = p.wreg[r["r"]]
                _pk(e, "v_pk_fma_f32", wr, [VP(g[2]), VP(t), VP(wr)])                           # W(x_j) += (-L) rhs
            if y0:       # y == 0: the rhs is z
                op([S2(p.LW_Z + p.zpos[i]), C2(p.lpos[r["j"]])],
                   lambda g, r=r: _pk(e, "v_pk_fma_f32", p.wreg[r["r"]], [VP(g[1]), VP(g[0]), VP(p.wreg[r["r"]])]))
                continue
            op([("L", p.LW_Y + i), ("L", p.LW_Z + p.zpos[i]), C(p.lpos[r["j"]])], f2)
            continue
        if pack and "4" in PACK_PARTS and i in eqskip:
            continue
        lh = homes.get(("l", i))
        lh1 = homes.get(("l", i + 1))
        lsrc = lambda h: h if isinstance(h, tuple) else C(h)
        if pack and "4" in PACK_PARTS and i in eqfirst and isinstance(lh, tuple) and isinstance(lh1, tuple):
            eqskip.add(i + 1)           # both bounds in AGPR homes: read as a pair
            op([S2(p.LW_Y + i), ("A2", lh[1], lh1[1])], lambda g, k=k: _pk(e, "v_pk_fma_f32", p.wreg[k],
                                                                           [VP(g[0]), SB(S_RINVEQ, S_RINVEQ % 2), VP(g[1])], [1, 0, 0]))
            continue
        if pack and "4" in PACK_PARTS and i in eqfirst and not isinstance(lh, tuple) and not isinstance(lh1, tuple) and \
                (lh is None) == (lh1 is None) and (lh is None or (lh % 2 == 0 and lh1 == lh + 1)):
            eqskip.add(i + 1)

            def f2e(g, k=k):
                wr = p.wreg[k]
                wait_pre(wr)
                wait_pre(wr + 1)
                _pk(e, "v_pk_fma_f32", wr, [VP(g[0]), SB(S_RINVEQ, S_RINVEQ % 2), VP(wr)], [1, 0, 0])
            if lh is not None:       # l from its LDS home
                op([S2(p.LW_Y + i), C2(lh)], lambda g, k=k: _pk(e, "v_pk_fma_f32", p.wreg[k],
                                                                       [VP(g[0]), SB(S_RINVEQ, S_RINVEQ % 2), VP(g[1])], [1, 0, 0]))
            else:
                op([("L", p.LW_Y + i)], f2e)
            continue
        if r["eq"] and not r["leaf"]:
            def f(g, k=k):
                wait_pre(p.wreg[k])
                e("v_fma_f32", W(k), "-" + v(g[0]), sRe, W(k))
            if lh is not None:
                op([S(p.LW_Y + i), lsrc(lh)], lambda g, k=k: e("v_fma_f32", W(k), "-" + v(g[0]), sRe, v(g[1])))
            else:
                op([("L", p.LW_Y + i)], f)
        elif r["eq"]:
            op([S(p.LW_Y + i), C(p.lpos[r["j"]]), src_of(("l", i))],
               lambda g, r=r, i=i: (e("v_fma_f32", v(T(0)), "-" + v(g[0]), sRe, v(g[2])),
                                    e("v_fmac_f32", W(r["r"]), v(g[1]), v(T(0)))))
        elif y0 and not r["leaf"]:
            op([S(p.LW_Z + p.zpos[i])], lambda g, k=k: e("v_mov_b32", W(k), v(g[0])))
        elif y0:
            op([S(p.LW_Z + p.zpos[i]), C(p.lpos[r["j"]])], lambda g, r=r: e("v_fmac_f32", W(r["r"]), v(g[1]), v(g[0])))
        elif loose and not r["leaf"]:
            op([("L", p.LW_Y + i), ("L", p.LW_Z + p.zpos[i])],
               lambda g, k=k: e("v_fma_f32", W(k), "-s%d" % S_RIMIN, v(g[0]), v(g[1])))
        elif loose:
            op([("L", p.LW_Y + i), ("L", p.LW_Z + p.zpos[i]), C(p.lpos[r["j"]])],
               lambda g, r=r: (e("v_fma_f32", v(T(0)), "-s%d" % S_RIMIN, v(g[0]), v(g[1])),
                               e("v_fmac_f32", W(r["r"]), v(g[2]), v(T(0)))))
        elif not r["leaf"]:
            op([("L", p.LW_Y + i), ("L", p.LW_Z + p.zpos[i]), src_of(("rinv", i))],
               lambda g, k=k: e("v_fma_f32", W(k), "-" + v(g[2]), v(g[0]), v(g[1])))
        else:
            op([("L", p.LW_Y + i), ("L", p.LW_Z + p.zpos[i]), src_of(("rinv", i)), C(p.lpos[r["j"]])],
               lambda g, r=r: (e("v_fma_f32", v(T(0)), "-" + v(g[2]), v(g[0]), v(g[1])),
                               e("v_fmac_f32", W(r["r"]), v(g[3]), v(T(0)))))
    if not rhs:                  # (the definitions above are needed below; the operations are not)
        del ops[rhs_ops_start:]
    # ---- solves over the non-leaf unknowns (qdldl.c:250-293)
    lsrc_ = lambda j: ("V", own.lhome[j]) if j in own.lhome else ("L", p.lpos[j])
    kof = {reg: k for k, reg in p.wreg.items() if own.k(k)}

    def scale(which):
        """W <- W / D on this wave's unknowns selected by `which`; two per instruction where both registers of an aligned pair are"""
        kdone = set()
        for reg in sorted(kof) if pack else [p.wreg[k] for k in p.nonleaf if own.k(k)]:    # (by register when pairing: a pair is visited once)
            k = kof[reg]
            if k in kdone or not which(k):
                continue
            if "2" in PACK_PARTS and pack and reg % 2 == 0 and reg + 1 in kof and which(kof[reg + 1]):
                k1 = kof[reg + 1]
                kdone.add(k1)
                op([DI2(k, k1)], lambda g, reg=reg: _pk(e, "v_pk_mul_f32", reg, [VP(reg), VP(g[0])]))
            else:
                op([DI(k)], lambda g, k=k: e("v_mul_f32", W(k), W(k), v(g[0])))

    def meet():
        op([], lambda g: (e("s_waitcnt", "lgkmcnt(0)"), e("s_barrier")))
        ops.append(dict(flush=True))              # (nothing is fetched across it: the partner's words are read behind the barrier)
    if not own.xbar:
        for (r_, c, j) in p.solve_entries:
            if own.k(c):
                op([lsrc_(j)], lambda g, r_=r_, c=c: e("v_fmac_f32", W(r_), v(g[0]), W(c)))
        scale(lambda k: True)
        for (r_, c, j) in reversed(p.solve_entries):
            if own.k(c):
                op([lsrc_(j)], lambda g, r_=r_, c=c: e("v_fmac_f32", W(c), v(g[0]), W(r_)))
    else:
        # a component cut in two (LoopSplit): the subtrees of the elimination tree side by side, the separator between two barriers
        top = own.top
        for (r_, c, j) in p.solve_entries:
            if own.k(c) and c not in top and own.k(r_):
                op([lsrc_(j)], lambda g, r_=r_, c=c: e("v_fmac_f32", W(r_), v(g[0]), W(c)))
        for c, word in sorted(own.xsend_f.items()):
            op([], lambda g, c=c, word=word: sc.lds_write(word, p.wreg[c]))       # (the forward value: before the scaling below)
        # W / D of the subtree: a wave that only waits between the two barriers (half B, a whole small component) does it there;
        # the wave with the separator after the second barrier, while its partner's read of the separator's solution is in flight
        late = bool(top) and SCALE_LATE
        if not SCALE_LATE:
            scale(lambda k: k not in top)
        meet()
        if SCALE_LATE and not late:
            scale(lambda k: True)
        for (r_, c, j) in own.cross:
            op([lsrc_(j), ("L", own.xrecv_f[c])], lambda g, r_=r_: e("v_fmac_f32", W(r_), v(g[0]), v(g[1])))
        for (r_, c, j) in p.solve_entries:
            if own.k(c) and c in top:
                op([lsrc_(j)], lambda g, r_=r_, c=c: e("v_fmac_f32", W(r_), v(g[0]), W(c)))
        scale(lambda k: k in top)
        for (r_, c, j) in reversed(p.solve_entries):
            if own.k(c) and c in top:
                op([lsrc_(j)], lambda g, r_=r_, c=c: e("v_fmac_f32", W(c), v(g[0]), W(r_)))
        for r_, word in sorted(own.xsend_b.items()):
            op([], lambda g, r_=r_, word=word: sc.lds_write(word, p.wreg[r_]))
        meet()
        if late:
            scale(lambda k: k not in top)
        for (r_, c, j) in reversed(p.solve_entries):
            if own.k(c) and c not in top:
                if own.k(r_):
                    op([lsrc_(j)], lambda g, r_=r_, c=c: e("v_fmac_f32", W(c), v(g[0]), W(r_)))
                else:
                    op([lsrc_(j), ("L", own.xrecv_b[r_])], lambda g, c=c: e("v_fmac_f32", W(c), v(g[0]), v(g[1])))
    if capture and group:
        # x_prev and delta_y are written over the L words, ANY wave's: nobody writes them before everybody's last solve is done
        op([], lambda g: (e("s_waitcnt", "lgkmcnt(0)"), e("s_barrier")))
    ops.append(dict(flush=True))
    sptr = "s[%d:%d]" % (S_P, S_P + 1)

    def store_dy(i, compute, reg):
        """capturing iteration: delta_y of row i -> LDS word LW_DY + i (the solve block of L is dead by now)"""
        if not capture:
            return
        compute()
        sc.lds_write(p.LW_DY + i, reg)
    # ---- P6: row updates (auxil.c:203-228); leaf rows re-form their multiplier from the final unknown of their variable
    npk = [0]
    eqskip2 = set()
    zc1 = zc2 = yc1 = yc2 = xc1 = xc2 = frozenset()
    if pack and PAIR_QUADS:
        assert (p.V_LAND + 8) % 2 == 0
        if y0:
            zc1, zc2 = couples([p.LW_Z + p.zpos[r["i"]] for r in p.rows if paired.get(r["i"]) is not None and p.LW_Z + p.zpos[r["i"]] not in sh])
        if "4" in PACK_PARTS:
            yc1, yc2 = couples([p.LW_Y + i for i in sorted(eqfirst) if p.LW_Y + i not in sh])
        if "3" in PACK_PARTS:
            xc1, xc2 = couples([p.LW_X + j for j in range(0, n, 2) if xpair(j) and p.LW_X + j not in sh])
    assert not fuse or (y0 and pack and not capture and PAIR_QUADS and Y0_DLEAF)
    # fuse: the updates run eq rows, then the inequality rows that are not part of a fused pair, then the box-row pairs together with
    # the x update of their two variables (a variable's W register is rewritten there: every row that reads it as x~ went before),
    # then the other x updates, then the right-hand-side pushes of the leaf rows that could not be fused
    B_EQ, B_OTHER, B_PAIR, B_PUSH = [], [], [], []
    xbase = min(p.wreg[p.pinv[j]] for j in range(n))
    fused_x = set()                     # variables whose x update is part of a fused pair operation

    def rop(bucket, srcs, fn):
        (bucket if fuse else ops).append(dict(srcs=srcs, emit=fn))

    def fusable(r, rb):
        """box-row pair (r, rb) <-> x pair (j, j + 1) with q in an aligned LDS pair"""
        j = p.wreg[r["r"]] - xbase
        if not (0 <= j < n - 1 and p.pinv[j] == r["r"] and p.pinv[j + 1] == rb["r"] and xpair(j)):
            return None
        # (a variable that another leaf row pushes into as well keeps the classic order of its pushes: row order)
        if any(o["leaf"] and o["r"] in (r["r"], rb["r"]) and o is not r and o is not rb for o in p.rows):
            return None
        qh = homes.get(("q", j))
        if isinstance(qh, tuple) or qh is None or qh % 2 or homes.get(("q", j + 1)) != qh + 1:
            return None
        return j
    fpairs = [(r, paired[r["i"]], fusable(r, paired[r["i"]])) for r in p.rows if fuse and paired.get(r["i"]) is not None]
    fpairs = [(r, rb, j) for (r, rb, j) in fpairs if j is not None]
    fzc1, fzc2 = couples([p.LW_Z + p.zpos[r["i"]] for (r, _, _) in fpairs if p.LW_Z + p.zpos[r["i"]] not in sh]) if fuse else (set(), set())
    fxc1, fxc2 = couples([p.LW_X + j for (_, _, j) in fpairs if p.LW_X + j not in sh]) if fuse else (set(), set())
    fused_rows = {r["i"] for (r, _, _) in fpairs} | {rb["i"] for (_, rb, _) in fpairs}
    for (_, _, j) in fpairs:
        fused_x.update((j, j + 1))
    if fuse:      # the remaining pairs couple among themselves
        zc1, zc2 = couples([p.LW_Z + p.zpos[r["i"]] for r in p.rows if paired.get(r["i"]) is not None and r["i"] not in fused_rows
                            and p.LW_Z + p.zpos[r["i"]] not in sh])
        xc1, xc2 = couples([p.LW_X + j for j in range(0, n, 2) if xpair(j) and j not in fused_x and p.LW_X + j not in sh])
    for r in p.rows:
        i, k = r["i"], r["k"]
        if not own.row(i):
            continue
        yw = p.LW_Y + i
        if i in fused_rows:
            if paired[i] is None:
                continue
            rb = paired[i]
            zw = p.LW_Z + p.zpos[i]
            j = next(j_ for (r_, _, j_) in fpairs if r_ is r)

            def ffused(g, r=r, zw=zw, j=j):
                z, L_, xq, qq = VP(g[0]), VP(g[1]), VP(g[2]), VP(g[3])
                di = SB(S_DLEAF, S_DLEAF % 2)
                c = npk[0]
                a_ = TPK(c, 4) if zw in fzc1 else TPK(c, 6) if zw in fzc2 else TPK(c, 4)
                b_ = TPK(c, 0) if zw not in fzc2 else TPK(c, 2)
                xw = p.LW_X + j
                xt = T(0) if xw in fxc1 else T(2) if xw in fxc2 else T(0)
                rinv = SB(S_RIMIN, 0)
                wr = p.wreg[r["r"]]
                _pk(e, "v_pk_mul_f32", b_, [z, di])                              # nu = z / d ...
                _pk(e, "v_pk_fma_f32", b_, [L_, VP(wr), VP(b_)])                 # ... + (-L) x~_j
                _pk(e, "v_pk_fma_f32", a_, [rinv, VP(b_), z])                    # z~
                _pk(e, "v_pk_mul_f32", b_, [SB(S_OMA, S_OMA % 2), z])            # (1 - alpha) z
                zn = sh.get(zw, a_)                 # (a word with a register home: the last operation writes it there)
                _pk(e, "v_pk_fma_f32", zn, [SB(S_ALPHA, S_ALPHA % 2), VP(a_), VP(b_)])      # z_new
                if zw in sh:
                    pass
                elif zw in fzc2:
                    sc.lds_write4(zw - 2, TPK(c, 4))
                elif zw not in fzc1:
                    sc.lds_write2(zw, a_)
                xn = sh.get(xw, xt)
                _pk(e, "v_pk_mul_f32", xt, [SB(S_OMA, S_OMA % 2), xq])           # x update of the two variables
                _pk(e, "v_pk_fma_f32", xn, [SB(S_ALPHA, S_ALPHA % 2), VP(wr), VP(xt)])
                if xw in sh:
                    pass
                elif xw in fxc2:
                    sc.lds_write4(xw - 2, T(0))
                elif xw not in fxc1:
                    sc.lds_write2(xw, xt)
                _pk(e, "v_pk_fma_f32", wr, [SB(S_SIGMA, S_SIGMA % 2), VP(xn), qq], [0, 0, 1])   # next rhs: sigma x_new - q ...
                _pk(e, "v_pk_fma_f32", wr, [L_, VP(zn), VP(wr)])                 # ... + (-L) z_new  (y == 0)
                if zw not in fzc1:
                    npk[0] += 1
            rop(B_PAIR, [S2(zw), C2(p.lpos[r["j"]]), S2(p.LW_X + j), C2(homes[("q", j)])], ffused)
            continue
        if i in paired:
            if paired[i] is None:
                continue
            rb = paired[i]
            zw = p.LW_Z + p.zpos[i]

            def fp(g, r=r, yw=yw, zw=zw):
                # the scalar row update below, two rows per instruction (same operations on each half)
                y, z, L_, di = VP(g[0]), VP(g[1]), VP(g[2]), VP(g[3])
                a_, b_, c_ = TPK(npk[0], 0), TPK(npk[0], 2), TPK(npk[0], 4)
                npk[0] += 1
                rinv, rho = SB(S_RIMIN, 0), SB(S_RHOMIN, 1)
                wr = p.wreg[r["r"]]
                _pk(e, "v_pk_fma_f32", a_, [rinv, y, z], [1, 0, 0])              # t3 = z - y / rho
                _pk(e, "v_pk_mul_f32", b_, [VP(a_), di])                         # nu = t3 / d ...
                _pk(e, "v_pk_fma_f32", b_, [L_, VP(wr), VP(b_)])                 # ... + (-L) x~_j
                _pk(e, "v_pk_fma_f32", a_, [rinv, VP(b_), VP(a_)])               # z~
                _pk(e, "v_pk_mul_f32", b_, [SB(S_OMA, S_OMA % 2), z])            # (1 - alpha) z
                _pk(e, "v_pk_fma_f32", a_, [SB(S_ALPHA, S_ALPHA % 2), VP(a_), VP(b_)])      # t = alpha z~ + (1 - alpha) z
                _pk(e, "v_pk_fma_f32", b_, [rinv, y, VP(a_)])                    # z_new = t + y / rho (infinite bounds: no clip)
                _pk(e, "v_pk_add_f32", a_, [VP(a_), VP(b_)], [0, 1])             # t - z_new
                _pk(e, "v_pk_fma_f32", c_, [rho, VP(a_), y])                     # y_new = y + rho (t - z_new)
                sc.lds_write2(zw, b_)
                sc.lds_write2(yw, c_)
            def fp0(g, r=r, zw=zw):
                # y == 0: t3 = z, z_new = t, y_new = 0 -- five of the nine operations, no y word
                z, L_ = VP(g[0]), VP(g[1])
                di = SB(S_DLEAF, S_DLEAF % 2) if Y0_DLEAF else VP(g[2])
                if zw in zc1:                                        # first pair of a float4: the second one writes both
                    a_, b_ = TPK(npk[0], 4), TPK(npk[0], 0)
                elif zw in zc2:
                    a_, b_ = TPK(npk[0], 6), TPK(npk[0], 2)
                else:
                    a_, b_ = TPK(npk[0], 0), TPK(npk[0], 2)
                rinv = SB(S_RIMIN, 0)
                wr = p.wreg[r["r"]]
                _pk(e, "v_pk_mul_f32", b_, [z, di])                              # nu = z / d ...
                _pk(e, "v_pk_fma_f32", b_, [L_, VP(wr), VP(b_)])                 # ... + (-L) x~_j
                _pk(e, "v_pk_fma_f32", a_, [rinv, VP(b_), z])                    # z~
                _pk(e, "v_pk_mul_f32", b_, [SB(S_OMA, S_OMA % 2), z])            # (1 - alpha) z
                _pk(e, "v_pk_fma_f32", sh.get(zw, a_), [SB(S_ALPHA, S_ALPHA % 2), VP(a_), VP(b_)])      # z_new = alpha z~ + (1 - alpha) z
                if zw in sh:
                    pass
                elif zw in zc2:
                    sc.lds_write4(zw - 2, TPK(npk[0], 4))
                elif zw not in zc1:
                    sc.lds_write2(zw, a_)
                if zw not in zc1:
                    npk[0] += 1
            if y0:
                rop(B_OTHER, [S2(zw), C2(p.lpos[r["j"]])] + ([] if Y0_DLEAF else [("A2", k, rb["k"])]), fp0)
                if fuse:          # the pair's pushes into the next rhs: after the x updates
                    B_PUSH.append(dict(srcs=[S2(zw), C2(p.lpos[r["j"]])],
                                       emit=lambda g, r=r: _pk(e, "v_pk_fma_f32", p.wreg[r["r"]], [VP(g[1]), VP(g[0]), VP(p.wreg[r["r"]])])))
            else:
                op([("L", yw), ("L", zw), C(p.lpos[r["j"]]), ("A2", k, rb["k"])], fp)
            continue
        if r["eq"]:
            if r["leaf"]:
                def f(g, r=r, i=i, yw=yw):
                    e("v_fma_f32", v(T(0)), "-" + v(g[0]), sRe, v(g[3]))
                    e("v_mul_f32", v(T(0)), v(T(0)), v(g[2]))
                    e("v_fmac_f32", v(T(0)), v(g[1]), W(r["r"]))          # nu
                    e("v_sub_f32", v(T(1)), v(T(0)), v(g[0]))
                    store_dy(i, lambda: e("v_mul_f32", v(T(2)), sA, v(T(1))), T(2))
                    e("v_fma_f32", v(sh.get(yw, T(1))), sA, v(T(1)), v(g[0]))
                    if yw not in sh:
                        sc.lds_write(yw, T(1))
                rop(B_EQ, [S(yw), C(p.lpos[r["j"]]), DI(k), src_of(("l", i))], f)
                if fuse:          # its push into the next rhs (the classic operation), after the x updates
                    B_PUSH.append(dict(srcs=[S(yw), C(p.lpos[r["j"]]), src_of(("l", i))],
                                       emit=lambda g, r=r: (e("v_fma_f32", v(T(0)), "-" + v(g[0]), sRe, v(g[2])),
                                                            e("v_fmac_f32", W(r["r"]), v(g[1]), v(T(0))))))
            elif pack and "4" in PACK_PARTS and i in eqskip2:
                pass
            elif pack and "4" in PACK_PARTS and i in eqfirst:
                eqskip2.add(i + 1)

                def f2u(g, k=k, yw=yw):
                    t = TPK(npk[0], 4 if yw in yc1 else 6 if yw in yc2 else 0)
                    wr = p.wreg[k]
                    _pk(e, "v_pk_add_f32", t, [VP(wr), VP(g[0])], [0, 1])                  # nu - y
                    yn = sh.get(yw, t)
                    _pk(e, "v_pk_fma_f32", yn, [SB(S_ALPHA, S_ALPHA % 2), VP(t), VP(g[0])])  # y + alpha (nu - y)
                    if yw in sh:
                        pass
                    elif yw in yc2:
                        sc.lds_write4(yw - 2, TPK(npk[0], 4))
                    elif yw not in yc1:
                        sc.lds_write2(yw, t)
                    if fuse:          # next rhs: l - y_new / rho_eq (the solution words in W are consumed)
                        _pk(e, "v_pk_fma_f32", wr, [VP(yn), SB(S_RINVEQ, S_RINVEQ % 2), VP(g[1])], [1, 0, 0])
                    if yw not in yc1:
                        npk[0] += 1
                lh_, lh1_ = homes.get(("l", i)), homes.get(("l", i + 1))
                if fuse:
                    lsrc2 = ("A2", lh_[1], lh1_[1]) if isinstance(lh_, tuple) and isinstance(lh1_, tuple) else C(lh_)
                    assert isinstance(lh_, tuple) == isinstance(lh1_, tuple) and (isinstance(lh_, tuple) or (lh_ % 2 == 0 and lh1_ == lh_ + 1))
                    rop(B_EQ, [S2(yw), C2(lh_) if lsrc2[0] != "A2" else lsrc2], f2u)
                else:
                    op([("L", yw)], f2u)
            else:
                def f(g, k=k, yw=yw, i=i):
                    e("v_sub_f32", v(T(1)), W(k), v(g[0]))
                    store_dy(i, lambda: e("v_mul_f32", v(T(2)), sA, v(T(1))), T(2))
                    yn = sh.get(yw, T(1))
                    e("v_fma_f32", v(yn), sA, v(T(1)), v(g[0]))
                    if yw not in sh:
                        sc.lds_write(yw, T(1))
                    if fuse:
                        e("v_fma_f32", W(k), "-" + v(yn), sRe, v(g[1]))
                if fuse:
                    lh_ = homes[("l", i)]
                    rop(B_EQ, [S(yw), lh_ if isinstance(lh_, tuple) else C(lh_)], f)
                else:
                    op([("L", yw)], f)
            continue
        zw = p.LW_Z + p.zpos[i]
        if y0:
            def f0(g, r=r, k=k, zw=zw):
                z = v(g[0])
                t3, nu, t2, tt = (v(T(q)) for q in range(4))
                if r["leaf"]:
                    e("v_mul_f32", nu, z, "s%d" % S_DLEAF if Y0_DLEAF else v(g[2]))
                    e("v_fmac_f32", nu, v(g[1]), W(r["r"]))
                else:
                    nu = W(k)
                e("v_fma_f32", t3, "s%d" % S_RIMIN, nu, z)                    # z~
                e("v_mul_f32", t2, sO, z)
                if zw in sh:
                    tt = v(sh[zw])
                e("v_fma_f32", tt, sA, t3, t2)                                # z_new = alpha z~ + (1 - alpha) z
                if capture:
                    e("v_mov_b32", v(T(5)), 0)                                # delta_y = rho (t - z_new) = 0
                    store_dy(r["i"], lambda: None, T(5))
                if zw not in sh:
                    sc.lds_write(zw, T(3))
                if fuse and not r["leaf"]:
                    e("v_mov_b32", W(k), tt)                                  # next rhs of the row: z_new (y == 0)
            rop(B_OTHER, [S(zw)] + ([C(p.lpos[r["j"]])] + ([] if Y0_DLEAF else [("A", k)]) if r["leaf"] else []), f0)
            if fuse and r["leaf"]:
                B_PUSH.append(dict(srcs=[S(zw), C(p.lpos[r["j"]])],
                                   emit=lambda g, r=r: e("v_fmac_f32", W(r["r"]), v(g[1]), v(g[0]))))
            continue
        if loose:
            srcs, nfix = [("L", yw), ("L", zw)], 2
        else:
            srcs = [("L", yw), ("L", zw), src_of(("l", i)), src_of(("u", i)), src_of(("rinv", i))]
            nfix = 5
            if not KNOB["rho_select"]:
                srcs.append(src_of(("rho", i)))
                nfix = 6
        if r["leaf"]:
            srcs += [C(p.lpos[r["j"]]), ("A", k)]

        def f(g, r=r, k=k, yw=yw, zw=zw, nfix=nfix):
            if loose:
                y, z = v(g[0]), v(g[1])
                lo = up = None
                rinv = "s%d" % S_RIMIN
            else:
                y, z, lo, up, rinv = (v(x) for x in g[:5])
            t3, nu, t2, tt, t4, d = (v(T(q)) for q in range(6))
            if loose:
                rho = "s%d" % S_RHOMIN
            elif KNOB["rho_select"]:
                rho = v(T(6))
                # rho of the row: the class constant whose 1/rho this is (auxil.c:103-145 assigns one of three)
                e("v_cmp_eq_f32", "vcc", "s%d" % S_RINV0, rinv)
                e("v_cndmask_b32", rho, v(p.V_RHOMIN), v(p.V_RHO0), "vcc")
                e("v_cmp_eq_f32", "vcc", sRe, rinv)
                e("v_cndmask_b32", rho, rho, v(p.V_RHOEQ), "vcc")
            else:
                rho = v(g[5])
            e("v_fma_f32", t3, "-" + rinv, y, z)                          # z - y/rho (the rhs again)
            if r["leaf"]:
                e("v_mul_f32", nu, t3, v(g[nfix + 1]))
                e("v_fmac_f32", nu, v(g[nfix]), W(r["r"]))
            else:
                nu = W(k)
            e("v_fma_f32", t3, rinv, nu, t3)                              # z~
            e("v_mul_f32", t2, sO, z)
            e("v_fma_f32", tt, sA, t3, t2)                                # alpha z~ + (1 - alpha) z
            e("v_fma_f32", t4, rinv, y, tt)
            if not loose:                                                 # (infinite bounds never clip a finite value)
                e("v_max_f32", t4, t4, lo)
                e("v_min_f32", t4, t4, up)                                # z_new
            e("v_sub_f32", d, tt, t4)
            if capture:
                e("v_mul_f32", d, rho, d)                                 # delta_y
                store_dy(r["i"], lambda: None, T(5))
                e("v_add_f32", d, y, d)
            else:
                e("v_fma_f32", d, rho, d, y)                              # y_new = y + rho (t - z_new)
            sc.lds_write(zw, T(4))
            sc.lds_write(yw, T(5))
        op(srcs, f)
    assert land[0] == (0 if (loose or gh) else p.n_land if own_land is None else len(own_land))
    # ---- x <- alpha x~ + (1 - alpha) x
    if fuse:
        ops.extend(B_EQ + B_OTHER + B_PAIR)
    if capture and group:
        # x_prev goes over the leaf block of L, which ANY wave's row updates above still read: meet once more
        assert not fuse
        op([], lambda g: (e("s_waitcnt", "lgkmcnt(0)"), e("s_barrier")))
        ops.append(dict(flush=True))
    jskip = set(fused_x)
    for j in range(n):
        k = p.pinv[j]
        if j in jskip or not own.var(j):
            continue
        if "3" in PACK_PARTS and xpair(j) and j + 1 not in fused_x:
            jskip.add(j + 1)

            def fx2(g, j=j):
                xw = p.LW_X + j
                t = TPK(xw // 4, 4) if xw in xc1 else TPK(xw // 4, 6) if xw in xc2 else TPK(j // 2, 6)     # (a couple = one float4 of x words)
                wr = p.wreg[p.pinv[j]]
                xn = sh.get(xw, t)
                _pk(e, "v_pk_mul_f32", t, [SB(S_OMA, S_OMA % 2), VP(g[0])])
                _pk(e, "v_pk_fma_f32", xn, [SB(S_ALPHA, S_ALPHA % 2), VP(wr), VP(t)])
                if xw in sh:
                    pass
                elif xw in xc2:
                    sc.lds_write4(xw - 2, TPK(xw // 4, 4))
                elif xw not in xc1:
                    sc.lds_write2(xw, t)
                if fuse:          # next rhs: sigma x_new - q
                    _pk(e, "v_pk_fma_f32", wr, [SB(S_SIGMA, S_SIGMA % 2), VP(xn), VP(g[1])], [0, 0, 1])
            qh_ = homes.get(("q", j))
            if fuse and not (isinstance(qh_, int) and qh_ % 2 == 0 and homes.get(("q", j + 1)) == qh_ + 1):
                jskip.discard(j + 1)          # (no aligned q pair: the two variables go one at a time below)
            else:
                op([S2(p.LW_X + j)] + ([C2(qh_)] if fuse else []), fx2)
                continue

        def f(g, k=k, j=j):
            t = T(6 + j % 2)
            if capture:
                sc.lds_write(p.LW_XP + j, g[0])      # x_prev (every L word is dead by now)
            xn = sh.get(p.LW_X + j, t)
            e("v_mul_f32", v(t), sO, v(g[0]))
            e("v_fma_f32", v(xn), sA, W(k), v(t))
            if p.LW_X + j not in sh:
                sc.lds_write(p.LW_X + j, t)
            if fuse:
                e("v_fma_f32", W(k), sS, v(xn), "-" + v(g[1]))
        qh_ = homes.get(("q", j))
        op([S(p.LW_X + j)] + ([qh_ if isinstance(qh_, tuple) else C(qh_)] if fuse else []), f)
    if fuse:
        ops.append(dict(flush=True))          # the pushes read words this body has written
        ops.extend(B_PUSH)
    # (Own.dhome sits in ring slots and in the AGPR temporaries of the unpacked bodies: these bodies must need neither)
    assert not dh or (pack and all(len({s_[1] >> 2 for s_ in o_.get("srcs", []) if s_[0] == "L"}) <= 1 for o_ in ops))
    sc.run(ops)
    if not capture:
        preloads(e, p, homes, own)


def _row_ptr(e, sreg, row, base=S_W):
    e("s_mul_i32", "s%d" % sreg, "s%d" % S_STRIDE, row)
    e("s_mul_hi_u32", "s%d" % (sreg + 1), "s%d" % S_STRIDE, row)
    e("s_add_u32", "s%d" % sreg, "s%d" % sreg, "s%d" % base)
    e("s_addc_u32", "s%d" % (sreg + 1), "s%d" % (sreg + 1), "s%d" % (base + 1))


def _adv(e, sreg):
    e("s_add_u32", "s%d" % sreg, "s%d" % sreg, "s%d" % S_STRIDE)
    e("s_addc_u32", "s%d" % (sreg + 1), "s%d" % (sreg + 1), 0)


def prologue(e, p):
    """rows R_L.. (negated L in storage order), R_DI, R_X, R_Y, R_Z -> LDS / AGPRs; the once-only stream items"""
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")     # the C++ side's hand-off stores have left the wave
    e("v_add_u32", "v%d" % V_B1, 0x10000, "v1")
    e("v_add_u32", "v%d" % V_B2, 0x20000, "v1")
    # L, x, y, z -> LDS through the W registers as landing zone, a group of rows at a time
    lds_rows = [(p.R_L + q, p.LW_L + q) for q in range(p.LW_X)] + [(p.R_X + q, p.LW_X + q) for q in range(p.n)] + \
               [(p.R_Y + q, p.LW_Y + q) for q in range(p.m)] + [(p.R_Z + q, p.LW_Z + q) for q in range(p.LW_END - p.LW_Z)]
    nl = len(p.nonleaf)
    for g in range(0, len(lds_rows), nl):
        grp = lds_rows[g:g + nl]
        last_row = None
        for q, (row, word) in enumerate(grp):
            if last_row is None or row != last_row + 1:
                _row_ptr(e, S_P, row)
            else:
                _adv(e, S_P)
            last_row = row
            e("global_load_dword", "v%d" % (V_W + q), "v0", "s[%d:%d]" % (S_P, S_P + 1), 0)
        e("s_waitcnt", "vmcnt(0)")
        for q, (row, word) in enumerate(grp):
            base, off = lds_addr(word)
            e("ds_write_b32", base, "v%d" % (V_W + q), off)
        e("s_waitcnt", "lgkmcnt(0)")
    # 1/D -> AGPRs
    _row_ptr(e, S_P, p.R_DI)
    for k in range(p.nk):
        e("global_load_dword", "a%d" % k, "v0", "s[%d:%d]" % (S_P, S_P + 1), 0)
        _adv(e, S_P)
    prologue_tail(e, p)


def y0_fill(e, p, own=ALL):
    """y0 variant: the items that have an LDS home (Plan.y0_home) are fetched from the stream ONCE, through the W registers"""
    items = p.stream + p.extra
    todo = sorted((kv for kv in p.y0_home.items() if own.item(kv[0])), key=lambda kv: items.index(kv[0]))
    assert len(todo) <= len(p.nonleaf)
    blk = None
    for q, (item, word) in enumerate(todo):
        idx = items.index(item)
        if idx // BLOCK != blk:
            blk = idx // BLOCK
            e("s_add_u32", "s%d" % S_SP, "s%d" % S_S, blk * BLOCK * 256)
            e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_S + 1), 0)
        dst = "a%d" % word[1] if isinstance(word, tuple) else "v%d" % (V_W + q)
        e("global_load_dword", dst, "v%d" % V_LANE, "s[%d:%d]" % (S_SP, S_SP + 1), (idx % BLOCK) * 256)
    e("s_waitcnt", "vmcnt(0)")
    for q, (item, word) in enumerate(todo):
        if isinstance(word, tuple):
            continue
        base, off = lds_addr(word)
        e("ds_write_b32", base, "v%d" % (V_W + q), off)
    e("s_waitcnt", "lgkmcnt(0)")


def y0_restore(e, p, own=ALL):
    """y0 variant, after the last iteration: the y words that held q are the multipliers again (zero)"""
    words = sorted(w for (what, j_), w in p.y0_home.items() if what == "q" and not isinstance(w, tuple) and own.var(j_))
    z4 = p.V_TT
    for r in range(4):
        e("v_mov_b32", "v%d" % (z4 + r), 0)
    k = 0
    while k < len(words):
        w = words[k]
        base, off = lds_addr(w)
        if w % 4 == 0 and words[k:k + 4] == [w, w + 1, w + 2, w + 3]:
            e("ds_write_b128", base, "v[%d:%d]" % (z4, z4 + 3), off)
            k += 4
        else:
            e("ds_write_b32", base, "v%d" % z4, off)
            k += 1


def prologue_tail(e, p, loose=False, homes=(), own=ALL):
    """once-only stream items (l of the leaf equality rows) -> their registers; the first iteration's preloads"""
    idx = p.n_stream
    assert len(p.extra) <= len(p.nonleaf)
    blk = None
    skip = (lambda item: (loose and item[0] == "rinv") or not own.item(item))   # (the loose loop takes 1/rho of those rows from an SGPR)
    for q, item in enumerate(p.extra):
        if skip(item):
            continue
        if (idx + q) // BLOCK != blk:
            blk = (idx + q) // BLOCK
            e("s_add_u32", "s%d" % S_SP, "s%d" % S_S, blk * BLOCK * 256)
            e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_S + 1), 0)
        kind, where = p.once[item]
        dst = "a%d" % where if kind == "A" else "v%d" % where if kind == "V" else "v%d" % (V_W + q)   # (W: landing zone)
        e("global_load_dword", dst, "v%d" % V_LANE, "s[%d:%d]" % (S_SP, S_SP + 1), ((idx + q) % BLOCK) * 256)
    e("s_waitcnt", "vmcnt(0)")
    for q, item in enumerate(p.extra):
        kind, where = p.once[item]
        if kind == "L" and not skip(item):
            base, off = lds_addr(where)
            e("ds_write_b32", base, "v%d" % (V_W + q), off)
    e("s_waitcnt", "lgkmcnt(0)")
    if KNOB["rho_select"]:
        e("v_mov_b32", "v%d" % p.V_RHO0, "s%d" % S_RHO0)
        e("v_mov_b32", "v%d" % p.V_RHOEQ, "s%d" % S_RHOEQ)
        e("v_mov_b32", "v%d" % p.V_RHOMIN, f32bits(float(np.float32(RHO_MIN_F32))))
    if not homes and own.ghome:
        homes = {it: ("A", a_) for it, a_ in own.ghome.items()}
    preloads(e, p, homes, own)


def g_homes_fill(e, p, own):
    """bodies that are not y0 (the general loop; the loose loop with non-zero multipliers): the wave's read-only stream items ->
    the AGPRs they are read from in every iteration (Own.ghome), one load each per tick"""
    items = p.stream + p.extra
    blk = None
    for item, a_ in sorted(own.ghome.items(), key=lambda kv: items.index(kv[0])):
        idx = items.index(item)
        if idx // BLOCK != blk:
            blk = idx // BLOCK
            e("s_add_u32", "s%d" % S_SP, "s%d" % S_S, blk * BLOCK * 256)
            e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_S + 1), 0)
        e("global_load_dword", "a%d" % a_, "v%d" % V_LANE, "s[%d:%d]" % (S_SP, S_SP + 1), (idx % BLOCK) * 256)
    if own.ghome:
        e("s_waitcnt", "vmcnt(0)")


def epilogue(e, p):
    """x, y, z of the inequality rows (words LW_X.., LW_Y.., LW_Z..), x_prev (LW_XP..) and delta_y (LW_DY..) stay in LDS: the
    C++ side reads them there (hipcc fetches rows from global memory one exposed load at a time)"""
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")


S_FAST, S_XI, S_YI, S_ZI = 30, 24, 26, 28    # fast start: flag, the caller's x, y, z rows ([row][B] floats)


def _lstamp(e, own, k):
    """(diagnostics, UMPC_QP_LOOP_STAMPS=1: 100 MHz stamp k of wavefront 0's loose loop block in s[60+2k:61+2k])"""
    if LOOP_STAMPS and (own.all or own.wave == 0):
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_memrealtime", "s[%d:%d]" % (60 + 2 * k, 61 + 2 * k))
        e("s_waitcnt", "lgkmcnt(0)")
FAC_MIN = 638                                # LDS word: min |d_k| of the factorisation (0 = a zero pivot)


def prologue_fast(e, p, res, loose=False, y0check=False, own=ALL, group=False):
    """KKT fill + LDL' inside the block (factor_emit): the equilibrated A and P come from the wave's residual stream (written
    by the Ruiz block), 1/rho of the inequality rows from the loop's stream, the warm start straight from the caller's rows.
    -L lands in the loop's LDS words, 1/D in its AGPRs: no hand-off rows at all."""
    s = p.s
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    _lstamp(e, own, 0)
    e("v_add_u32", "v%d" % V_B1, 0x10000, "v1")
    e("v_add_u32", "v%d" % V_B2, 0x20000, "v1")
    cur = [None]

    def sload(reg, idx):
        if idx // BLOCK != cur[0]:
            cur[0] = idx // BLOCK
            e("s_add_u32", "s%d" % S_SP, "s%d" % S_S, cur[0] * BLOCK * 256)
            e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_S + 1), 0)
        e("global_load_dword", "v%d" % reg, "v%d" % V_LANE, "s[%d:%d]" % (S_SP, S_SP + 1), (idx % BLOCK) * 256)
    # A -> LDS words LW_X.. (the x, y, z words: the warm start arrives after the factorisation)
    assert p.LW_X + s.nnzA <= 640
    G = V_END - V_W - (1 if y0check else 0)      # landing registers V_W ..; y0check keeps the last one as its accumulator
    A_p_ = list(s.tables["A_p"])
    own_a = [k for j in range(s.n) if own.fvar(j) for k in range(A_p_[j], A_p_[j + 1])]      # (what this wave FACTORISES)
    for g in range(0, len(own_a), G):
        ks = own_a[g:g + G]
        for q, k in enumerate(ks):
            sload(V_W + q, res.it_A + k)
        e("s_waitcnt", "vmcnt(0)")
        for q, k in enumerate(ks):
            base, off = lds_addr(p.LW_X + k)
            e("ds_write_b32", base, "v%d" % (V_W + q), off)
        e("s_waitcnt", "lgkmcnt(0)")
    _lstamp(e, own, 1)
    # P, 1/rho of the inequality rows -> registers for the whole factorisation
    v_p, v_rinv = V_W, V_W + s.nnzP
    gen = sorted(p.zpos, key=lambda i: p.zpos[i])
    pool0 = v_rinv + len(gen)
    assert pool0 + 30 <= p.V_RING
    for j in sorted(res.it_p):
        if own.fvar(j):
            sload(v_p + res.pidx[j], res.it_p[j])
    items = p.stream + p.extra
    for q, i in enumerate(gen):
        if loose:
            e("v_mov_b32", "v%d" % (v_rinv + q), "s%d" % S_RIMIN)
        else:
            sload(v_rinv + q, items.index(("rinv", i)))
    v_fmin = p.V_TT + N_TT - 1
    e("v_mov_b32", "v%d" % v_fmin, 1.0)
    e("s_waitcnt", "vmcnt(0)")
    # (temporaries: the factorisation keeps <= 8 partial rows alive; the rest of the pool pins L entries that a later row
    # multiplies again)
    assert p.V_RING - pool0 >= 30
    factor_emit(e, s, p, p.LW_X, v_p, v_rinv, dict(p.zpos), S_SIGMA, S_RINVEQ, list(range(pool0, p.V_RING - 12)),
                list(range(p.V_LAND, p.V_LAND + p.NLAND)) + list(range(p.V_RING - 12, p.V_RING)), v_fmin, own)
    _lstamp(e, own, 2)
    for q, (k, word) in enumerate(sorted(own.hand_out.items()) if not FACTOR_SPLIT else []):   # a cut component: 1/D of the partner's half -> LDS
        t = p.V_TT + q % 4                 # (V_TT + N_TT - 1 is the pivot accumulator)
        e("v_accvgpr_read_b32", "v%d" % t, "a%d" % k)
        base, off = lds_addr(word)
        e("ds_write_b32", base, "v%d" % t, off)
    base, off = lds_addr(FAC_MIN)
    if group:
        # several waves: the A words above are the other waves' x, y words -- nobody loads its warm start before everybody
        # has factorised; the smallest pivot is folded into the flag word wave 0 set to 1 (float min, atomic)
        if own.wave == 0:
            e("v_mov_b32", "v%d" % (p.V_TT), 1.0)
            e("ds_write_b32", base, "v%d" % (p.V_TT), off)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_barrier")
        e("ds_min_f32", base, "v%d" % v_fmin, off)
        if own.xbar and not FACTOR_SPLIT:
            # ... and the other half takes its 1/D from there into its own AGPRs; nobody's warm start overwrites the words before
            hin = sorted(own.hand_in.items())
            assert len(hin) <= G
            for q, (k, word) in enumerate(hin):
                base_, off_ = lds_addr(word)
                e("ds_read_b32", "v%d" % (V_W + q), base_, off_)
            e("s_waitcnt", "lgkmcnt(0)")
            for q, (k, word) in enumerate(hin):
                e("v_accvgpr_write_b32", "a%d" % k, "v%d" % (V_W + q))
            e("s_barrier")
    else:
        e("ds_write_b32", base, "v%d" % v_fmin, off)
    e("s_waitcnt", "lgkmcnt(0)")
    _lstamp(e, own, 3)
    # the warm start: x, y, z of the inequality rows -> LDS
    rows = [(S_XI, j, p.LW_X + j) for j in range(p.n) if own.var(j)] + [(S_YI, i, p.LW_Y + i) for i in range(p.m) if own.row(i)] + \
           [(S_ZI, i, p.LW_Z + p.zpos[i]) for i in gen if own.row(i)]
    v_or = V_END - 1             # y0check: OR of the multipliers' bits over the inequality rows
    if y0check:
        e("v_mov_b32", "v%d" % v_or, 0)
    for g in range(0, len(rows), G):
        grp = rows[g:g + G]
        last = None
        for q, (sb, row, word) in enumerate(grp):
            if last is None or (sb, row) != (last[0], last[1] + 1):
                _row_ptr(e, S_P, row, sb)
            else:
                _adv(e, S_P)
            last = (sb, row)
            e("global_load_dword", "v%d" % (V_W + q), "v0", "s[%d:%d]" % (S_P, S_P + 1), 0)
        e("s_waitcnt", "vmcnt(0)")
        for q, (sb, row, word) in enumerate(grp):
            base, off = lds_addr(word)
            e("ds_write_b32", base, "v%d" % (V_W + q), off)
            if y0check and sb == S_YI and row in p.zpos:
                e("v_or_b32", "v%d" % v_or, "v%d" % v_or, "v%d" % (V_W + q))
        e("s_waitcnt", "lgkmcnt(0)")
    _lstamp(e, own, 4)
    if y0check:
        return v_or
    prologue_tail(e, p, loose, own=own)


def l_homes_fill(e, p, own, y0=False):
    """-L of the wave's solve entries and its other resident words: LDS -> their register homes, once per block. Every path:
    Own.lhome; a path without resident iterates: the constants' VGPR homes (Own.chome; y0: Own.yhome); the y0 path of a wave
    with resident iterates (Own.shome): the constants into AGPRs (Own.ahome), x / y / z into the VGPRs"""
    yreg = y0 and bool(own.shome)
    byword = {p.lpos[j]: ("v", reg) for j, reg in own.lhome.items()}
    if yreg:
        byword.update({w_: ("a", a_) for w_, a_ in own.ahome.items()})
        byword.update({w_: ("v", r_) for w_, r_ in own.shome.items()})
    else:
        byword.update({w_: ("v", r_) for w_, r_ in own.chome.items()})
        if y0:
            byword.update({w_: ("v", r_) for w_, r_ in own.yhome.items()})
    if not byword:
        return
    quads = sorted(set(w_ >> 2 for w_ in byword))
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")         # (the preloads into W registers have landed: the ring registers are free)
    for g in range(0, len(quads), NRING):
        grp = quads[g:g + NRING]
        for q, qd in enumerate(grp):
            base, off = lds_addr(4 * qd)
            e("ds_read_b128", "v[%d:%d]" % (p.V_RING + 4 * q, p.V_RING + 4 * q + 3), base, off)
        e("s_waitcnt", "lgkmcnt(0)")
        for q, qd in enumerate(grp):
            for h in range(4):
                if 4 * qd + h in byword:
                    kind, reg = byword[4 * qd + h]
                    if kind == "v":
                        e("v_mov_b32", "v%d" % reg, "v%d" % (p.V_RING + 4 * q + h))
                    else:
                        e("v_accvgpr_write_b32", "a%d" % reg, "v%d" % (p.V_RING + 4 * q + h))
    if yreg:                          # (last: some of these registers are ring slots, the staging area above)
        for k, reg in sorted(own.dhome.items()):
            e("v_accvgpr_read_b32", "v%d" % reg, "a%d" % k)


def state_writeback(e, p, own):
    """x / y / z of a wave with resident iterates (Own.shome) go back to their LDS words: the capturing iteration, the residual
    block and the C++ side read them there"""
    done = set()
    for w_ in sorted(own.shome):
        if w_ in done:
            continue
        base, off = lds_addr(w_)
        r_ = own.shome[w_]
        if w_ % 2 == 0 and own.shome.get(w_ + 1) == r_ + 1 and r_ % 2 == 0:
            e("ds_write_b64", base, "v[%d:%d]" % (r_, r_ + 1), off)
            done.update((w_, w_ + 1))
        else:
            e("ds_write_b32", base, "v%d" % r_, off)
            done.add(w_)
    if own.shome:
        e("s_waitcnt", "lgkmcnt(0)")


def program(s, eq_rows, res=None, loose=False, own=ALL, group=False):
    """s11 = number of non-capturing iterations (>= 0); one capturing iteration follows them.
    res: a ResPlan -> the block also holds the fast start (prologue_fast), taken when s30 != 0
    own, group (loose only): the copy of the block that one wavefront of a workgroup runs on its components (LoopSplit); it
    meets the other wavefronts at two barriers (after the factorisation, at the end)"""
    p = Plan(s, eq_rows)
    e = Emit()
    assert res is not None or (own.all and not group)

    def loop(**kw):
        e("s_mov_b32", "s%d" % S_CNT, "s%d" % S_ITERS)
        e("s_cmp_lt_i32", "s%d" % S_CNT, 1)
        e("s_cbranch_scc1", "8f")
        e("label", "7")
        body(e, p, loose=loose, own=own, **kw)
        e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_CNT, 1)
        e("s_cmp_gt_i32", "s%d" % S_CNT, 0)
        e("s_cbranch_scc1", "7b")
        e("label", "8")
        body(e, p, capture=True, loose=loose, own=own, group=group, **kw)
    if loose:
        # the loose variant exists for the all-assembly route only: always the fast start (the caller passes s30 != 0)
        assert res is not None
        e("s_mov_b32", "s%d" % S_RIMIN, f32bits(float(np.float32(1.0 / QP_RHO_MIN))))
        e("s_mov_b32", "s%d" % S_RHOMIN, f32bits(float(np.float32(QP_RHO_MIN))))
        if not Y0_VARIANT:
            prologue_fast(e, p, res, loose=True, own=own, group=group)
        else:
            # y0: a loose row's multiplier moves by rho (t - z_new) with z_new = t + y / rho unclipped, so a multiplier that
            # starts at exactly 0 stays exactly 0 (z_new = t, delta_y = rho * 0) -- in the reference as here. When the warm start
            # has y == 0 on every inequality row of the wave (a cold start, or any earlier result of this loop), the iterations
            # below drop those 87 words and their operations (same values: bit-identical), and q / l move into the freed LDS
            # words instead of being loaded from the stream every iteration. Any other warm start takes the loop after label 20.
            v_or = prologue_fast(e, p, res, loose=True, y0check=True, own=own, group=group)
            e("v_and_b32", "v%d" % v_or, 0x7FFFFFFF, "v%d" % v_or)
            e("v_cmp_ne_u32", "vcc", 0, "v%d" % v_or)
            e("s_cbranch_vccnz", "20f")
            dleaf = [k for k in p.y0_dleaf if own.k(k)]
            if Y0_DLEAF and dleaf:
                e("v_accvgpr_read_b32", "v%d" % v_or, "a%d" % dleaf[0])
                e("s_nop", 0)
                e("v_readfirstlane_b32", "s%d" % S_DLEAF, "v%d" % v_or)
            y0_fill(e, p, own)
            prologue_tail(e, p, True, p.y0_home, own)
            l_homes_fill(e, p, own, y0=True)
            _lstamp(e, own, 5)
            if not Y0_FUSE:
                loop(y0=True)
            else:
                # four bodies: the first iteration forms its right-hand side the classic way and, like every middle one, leaves
                # the NEXT one's in the W registers; the capturing iteration starts at the solves; a single iteration is plain
                e("s_mov_b32", "s%d" % S_CNT, "s%d" % S_ITERS)
                e("s_cmp_lt_i32", "s%d" % S_CNT, 1)
                e("s_cbranch_scc1", "30f")
                body(e, p, loose=True, y0=True, rhs=True, fuse=True, own=own)
                e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_CNT, 1)
                e("s_cmp_lt_i32", "s%d" % S_CNT, 1)
                e("s_cbranch_scc1", "8f")
                e("label", "7")
                body(e, p, loose=True, y0=True, rhs=False, fuse=True, own=own)
                e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_CNT, 1)
                e("s_cmp_gt_i32", "s%d" % S_CNT, 0)
                e("s_cbranch_scc1", "7b")
                e("label", "8")
                state_writeback(e, p, own)
                body(e, p, capture=True, loose=True, y0=True, rhs=False, own=own, group=group)
                e("s_branch", "31f")
                e("label", "30")
                body(e, p, capture=True, loose=True, y0=True, own=own, group=group)
                e("label", "31")
            y0_restore(e, p, own)
            e("s_branch", "29f")
            e("label", "20")
            g_homes_fill(e, p, own)
            prologue_tail(e, p, True, own=own)
            l_homes_fill(e, p, own)
            _lstamp(e, own, 5)
            loop()
            e("label", "29")
            epilogue(e, p)
            if LOOP_STAMPS and (own.all or own.wave == 0):
                _lstamp(e, own, 6)
                # intervals: A -> LDS, factorisation, barrier, warm start, y0 fill + preloads, the iterations -> stream items
                e("s_add_u32", "s%d" % S_SP, "s%d" % S_S, (STAMP_ITEM0 // BLOCK) * BLOCK * 256)
                e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_S + 1), 0)
                for k in range(6):
                    e("s_sub_u32", "s%d" % (60 + 2 * k), "s%d" % (62 + 2 * k), "s%d" % (60 + 2 * k))
                    e("v_cvt_f32_u32", "v%d" % (p.V_TT + k), "s%d" % (60 + 2 * k))
                    e("global_store_dword", "v%d" % V_LANE, "v%d" % (p.V_TT + k), "s[%d:%d]" % (S_SP, S_SP + 1), ((STAMP_ITEM0 + k) % BLOCK) * 256)
                e("s_waitcnt", "vmcnt(0)")
            if group:
                e("s_barrier")        # every wave's x, y, z, x_prev, delta_y words are in LDS for whoever reads them next
            return e.ins, p
    elif group:
        prologue_fast(e, p, res, own=own, group=True)          # (a shared block always factorises itself: the fast start)
        g_homes_fill(e, p, own)
        l_homes_fill(e, p, own)
    else:
        if res is not None:
            e("s_cmp_lg_u32", "s%d" % S_FAST, 0)
            e("s_cbranch_scc1", "5f")
        prologue(e, p)
        if res is not None:
            e("s_branch", "6f")
            e("label", "5")
            prologue_fast(e, p, res)
            e("label", "6")
    e("s_mov_b32", "s%d" % S_CNT, "s%d" % S_ITERS)
    e("s_cmp_lt_i32", "s%d" % S_CNT, 1)
    e("s_cbranch_scc1", "8f")
    e("label", "7")
    body(e, p, loose=loose, own=own)
    e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_CNT, 1)
    e("s_cmp_gt_i32", "s%d" % S_CNT, 0)
    e("s_cbranch_scc1", "7b")
    e("label", "8")
    body(e, p, capture=True, loose=loose, own=own, group=group)
    epilogue(e, p)
    if group:
        e("s_barrier")
    return e.ins, p


def loop_group_program(s, eq_rows, res, nw=4, loose=True):
    """The loop block (loose or general variant) for a workgroup of nw wavefronts that own the same 64 robots: each takes the
    components LoopSplit gives it (a wavefront without one only keeps the barriers company). s33 = the wave's index; the
    other inputs as program(...) with the fast start. Returns (instructions, plan, split)."""
    p = Plan(s, eq_rows)
    sp = LoopSplit(p, nw)
    assert not sp.cut or sp.active == nw      # (a cut component puts barriers inside the loop: every wavefront runs a copy of it)
    e = Emit()
    for w in range(nw):
        if w < nw - 1:
            e("s_cmp_lg_u32", "s%d" % S_LWAVE, w)
            e("s_cbranch_scc1", "48f")
        if w < sp.active:
            ins, _ = program(s, eq_rows, res, loose=loose, own=sp.own(w), group=True)
            e.ins.extend(ins)
        else:
            for _ in range(4):        # after the factorisation, twice in the capturing iteration (its stores over L), at the end
                e("s_barrier")
        if w < nw - 1:
            e("s_branch", "49f")
            e("label", "48")
    e("label", "49")
    return e.ins, p, sp


S_LWAVE = 33                       # loop_group_program: s33 = the wave's index in its workgroup


def fmt(t):
    m = t[0]
    if m == "label":
        return "%s:" % t[1]
    if isinstance(t[-1], dict):          # VOP3P (packed) instruction: operands + op_sel / op_sel_hi / neg_lo / neg_hi
        d = t[-1]
        return "%s %s %s" % (m, ", ".join(str(x) for x in t[1:-1]),
                             " ".join("%s:[%s]" % (k, ",".join(map(str, d[k]))) for k in ("op_sel", "op_sel_hi", "neg_lo", "neg_hi")))
    a = [("0x%x" % x if isinstance(x, int) and m in ("s_mov_b32", "v_add_u32", "v_and_b32", "v_mov_b32") else str(x)) for x in t[1:]]
    if m.startswith("ds_"):
        return "%s %s, %s offset:%s" % (m, a[0], a[1], a[2])
    if m.startswith("global_"):
        # cache policy (round 5, tools/ab_qp_nt.sh): "rows" = the caller's [row][B] arrays (lane offset v0: read or written once
        # per tick), "stream" = the kernel's own per-workgroup stream blocks (written once, read once, by the same workgroup)
        off = a[1] if m == "global_load_dword" else a[0]
        kind = "rows" if off == "v0" else "stream"
        nt = (len(a) > 4 and a[4] == "nt") or kind in NT_KINDS
        return "%s %s, %s, %s offset:%s%s" % (m, a[0], a[1], a[2], a[3], " nt" if nt else "")
    if m == "s_waitcnt":
        return "s_waitcnt " + " ".join(a)
    return "%s %s" % (m, ", ".join(a))


# ---------------------------------------------------------------------------
# CPU interpreter (one lane) and a numpy statement of the same iteration
# ---------------------------------------------------------------------------
class AddressFault(Exception):
    """A simulated global access outside every array handed to the interpreter (on the GPU: a memory access fault)."""


# Simulated device addresses: like real ones, the low word of a base has bit 31 set, so a 32-bit half that is
# sign-extended somewhere on its way into an SGPR pair (the bug fixed in 37f012f: `readfirstlane` returns int) lands
# 4 GiB below the array and is caught by the bounds check of every access.
SIM_BASE_W, SIM_BASE_S, SIM_BASE_REGION = 0x00007F4A90000000, 0x00007F4BA0000000, 0x00007F5080000000


def uni_scalar_operand(v, sign_extend_bug=False):
    """How the C++ side of the generated kernels makes a 64-bit scalar operand provably wave-uniform (codegen_qp.emit_structure,
    `uni`): two readfirstlane halves. The builtin returns int; before 37f012f the halves were widened as SIGNED ints, which
    fills the upper word with ones whenever bit 31 of the low half is set (sign_extend_bug=True restates that)."""
    lo, hi = v & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF
    if sign_extend_bug:
        sx = lambda w: w | (0xFFFFFFFF00000000 if w & 0x80000000 else 0)
        return ((sx(hi) << 32) | sx(lo)) & 0xFFFFFFFFFFFFFFFF
    return (hi << 32) | lo


def simulate(*args, **kw):
    """One wave: runs _simulate to the end (an s_barrier of a lone wave is a no-op). See _simulate for the arguments."""
    g = _simulate(*args, **kw)
    while True:
        try:
            next(g)
        except StopIteration as stop:
            return stop.value


def simulate_group(ins, nw, W, S, iters, consts, wave_sgpr, **kw):
    """nw waves of one workgroup run `ins` on the same LDS slots and the same global arrays (wave w gets SGPR `wave_sgpr` = w),
    each up to its next s_barrier in turn. Checks that every wave meets every barrier, and that between two barriers no LDS
    word is written by one wave and read or written by another (the data races a barrier-phased schedule can have). Returns
    the shared LDS words and the per-wave executed instruction counts."""
    lds = np.zeros(640, np.float32)
    lds0 = kw.pop("lds0", None)
    if lds0 is not None:
        lds[:len(lds0)] = lds0
    logs = [dict(r=set(), w=set()) for _ in range(nw)]
    counts = [[] for _ in range(nw)]
    gens = []
    for w in range(nw):
        sg = dict(kw.get("sgpr") or {})
        sg[wave_sgpr] = w
        gens.append(_simulate(ins, W, S, iters, consts, **dict(kw, sgpr=sg, lds_shared=lds, access_log=logs[w], count=counts[w])))
    live = [True] * nw
    nbar = 0
    hist = []                           # per barrier phase: the words each wave wrote
    while any(live):
        for w in range(nw):
            if live[w]:
                try:
                    next(gens[w])
                except StopIteration:
                    live[w] = False
        assert all(live) or not any(live), ("the waves disagree about barrier %d" % nbar, live)
        for a_ in range(nw):
            for b_ in range(nw):
                if a_ != b_:
                    clash = (logs[a_]["w"] & (logs[b_]["r"] | logs[b_]["w"] | logs[b_].get("a", set()))) | \
                            (logs[a_].get("a", set()) & logs[b_]["r"])
                    assert not clash, ("LDS race before barrier %d: words written by wave %d and touched by wave %d" % (nbar, a_, b_),
                                       sorted(clash)[:8], {k_: sorted(clash & v_)[:4] for k_, v_ in logs[b_].items()})
        hist.append([set(lg["w"]) | set(lg.get("a", set())) for lg in logs])
        for b_ in range(nw):
            for (ph, word) in logs[b_].get("late", set()):
                for a_ in range(nw):
                    assert a_ == b_ or ph >= len(hist) or word not in hist[ph][a_], \
                        ("LDS race: word %d fetched by wave %d in phase %d (looked at later) was written by wave %d in that phase" % (word, b_, ph, a_))
        for lg in logs:
            for st_ in lg.values():
                st_.clear()
        nbar += 1
    return lds, [c[0] for c in counts], nbar - 1


def _simulate(ins, W, S, iters, consts, regions=None, sgpr=None, lds0=None, ret_agpr=False, base_xform=None, count=None,
              lds_shared=None, access_log=None):
    """W: float32[rows] row workspace (one robot), S: float32[items] stream block (one lane); consts = (alpha, sigma, rinv_eq).
    Runs the program and returns the lane's LDS words (x, y, z, x_prev, delta_y are left there).
    Addresses are formed as the ISA does for global_* with an SGPR base: SGPR pair (64 bits) + zero-extended 32-bit VGPR
    offset + immediate; every access must fall on an element of an array handed in (AddressFault otherwise). base_xform
    models what the C++ glue does to a base pointer before it reaches the SGPR pair (uni_scalar_operand)."""
    f32 = np.float32
    V = np.zeros(256, np.uint32)
    A = np.zeros(256, np.uint32)
    lds = np.zeros(640, f32) if lds_shared is None else lds_shared
    SG = {}
    scc = 0
    labels = {}
    log_r = access_log["r"] if access_log is not None else set()
    log_w = access_log["w"] if access_log is not None else set()
    log_a = access_log.setdefault("a", set()) if access_log is not None else set()
    src_word = {}
    src_phase = {}                      # register -> barrier phase in which its LDS read executed
    phase = [0]
    log_late = access_log.setdefault("late", set()) if access_log is not None else set()
    for k, t in enumerate(ins):
        if t[0] == "label":
            labels.setdefault(t[1], []).append(k)
    STRIDE = 4096
    base_xform = base_xform or (lambda a: a)
    space = [(SIM_BASE_W, STRIDE, W), (SIM_BASE_S, 256, S)]          # (true base, bytes between elements, array)

    def setbase(sreg, base):
        val = base_xform(base)
        SG[sreg], SG[sreg + 1] = val & 0xFFFFFFFF, (val >> 32) & 0xFFFFFFFF
    setbase(S_W, SIM_BASE_W)
    setbase(S_S, SIM_BASE_S)
    SG[S_STRIDE], SG[S_ITERS] = STRIDE, iters
    # regions (the Ruiz block's inputs): [(SGPR pair, array)], row-major with the simulated stride, 8 GiB apart
    for k_, v_ in (sgpr or {}).items():
        if v_ == "S":
            setbase(k_, SIM_BASE_S)
        elif v_ == "W":
            setbase(k_, SIM_BASE_W)
        else:
            SG[k_] = v_
    if lds0 is not None:
        lds[:len(lds0)] = lds0
    for q, (sreg, arr) in enumerate(regions or []):
        base = SIM_BASE_REGION + (q << 33)
        space.append((base, STRIDE, arr))
        setbase(sreg, base)
    alpha, sigma, rinv_eq = consts[:3]
    rho0 = f32(consts[3] if len(consts) > 3 else 0.1)
    for reg, val in ((S_ALPHA, f32(alpha)), (S_OMA, f32(f32(1.0) - f32(alpha))), (S_SIGMA, f32(sigma)), (S_RINVEQ, f32(rinv_eq)),
                     (S_RHO0, rho0), (S_RINV0, f32(1.0 / float(rho0))), (S_RHOEQ, f32(1e3 * float(rho0)))):
        SG.setdefault(reg, f32bits(float(val)))

    def sval(x):
        if isinstance(x, int):
            return x
        if x.startswith("s["):
            lo = int(x[2:x.index(":")])
            return SG.get(lo, 0) | (SG.get(lo + 1, 0) << 32)
        return SG.get(int(x[1:]), 0)

    def bits2f(b):
        return np.frombuffer(struct.pack("<I", int(b) & 0xFFFFFFFF), f32)[0]

    def fval(x):
        if isinstance(x, float):
            return f32(x)                       # inline constant
        neg = x.startswith("-")
        if neg:
            x = x[1:]
        if x.startswith("|"):
            val = abs(fval(x[1:-1]))
        else:
            val = bits2f(V[int(x[1:])]) if x[0] == "v" else bits2f(SG[int(x[1:])])
        return -val if neg else val

    def setf(x, val):
        V[int(x[1:])] = f32bits(float(f32(val)))

    def gaddr(t):
        """(array, index) of a global access (mnemonic, data, v_off, s[base:base+1], imm): base + zext(v_off) + imm"""
        addr = (sval(t[3]) + int(V[int(t[2][1:])]) + t[4]) & 0xFFFFFFFFFFFFFFFF
        for base, stride, arr in space:
            if base <= addr < base + len(arr) * stride and (addr - base) % stride == 0:
                return arr, (addr - base) // stride
        raise AddressFault("%r touches 0x%016x, outside every array of the call" % (ins[pc], addr))

    def ldsword(basereg, off):
        byte = int(V[int(basereg[1:])]) + off
        return (byte // 1024) * 4 + (byte % 1024) // 4

    # completion model: LDS operations and VMEM loads complete in issue order; a register that an outstanding load will
    # write must not be read or written before an s_waitcnt has retired that load
    pend = {"lgkmcnt": [], "vmcnt": []}

    def regs_of(x):
        if not isinstance(x, str):
            return set()
        x = x.lstrip("-").strip("|")
        if x.startswith("v["):
            lo, hi = x[2:-1].split(":")
            return {("v", r) for r in range(int(lo), int(hi) + 1)}
        if x[0] in "va" and x[1:].isdigit():
            return {(x[0], int(x[1:]))}
        return set()

    def check(used):
        for q in pend.values():
            for dst in q:
                assert not (dst & used), ("register used before its load was waited for", ins[pc], sorted(dst & used))

    pc = nexec = 0
    while pc < len(ins):
        t = ins[pc]
        m = t[0]
        nexec += 1
        assert nexec < 3000000, "runaway program"
        if m == "s_waitcnt":
            for part in t[1].split():
                name, val = part[:-1].split("(")
                del pend[name][:max(0, len(pend[name]) - int(val))]
        elif m[0] == "v" or m.startswith("ds_") or m.startswith("global_"):
            used = set().union(*[regs_of(x) for x in t[1:]])
            check(used)
            if m not in ("ds_read_b128", "ds_read_b32"):   # (a word fetched from LDS counts as READ when its register is consumed: a
                wr_only = regs_of(t[1]) if m in ("global_load_dword", "v_mov_b32", "v_accvgpr_read_b32") else set()
                for r_ in wr_only:            # quad may carry a neighbour's words that this wave never looks at; a register
                    src_word.pop(r_, None)    # that is overwritten no longer stands for the word)
                for r_ in used - wr_only:
                    if r_ in src_word:
                        if src_phase.get(r_, phase[0]) == phase[0]:
                            log_r.add(src_word[r_])
                        else:                 # fetched before a barrier, looked at behind it: the access belongs to THAT phase
                            log_late.add((src_phase[r_], src_word[r_]))
            if m in ("ds_read_b128", "ds_read_b32"):
                pend["lgkmcnt"].append(regs_of(t[1]))
            elif m.startswith("ds_write") or m == "ds_min_f32":
                pend["lgkmcnt"].append(set())
            elif m == "global_load_dword":
                pend["vmcnt"].append(regs_of(t[1]))
            elif m == "global_store_dword":
                pend["vmcnt"].append(set())
        if m in ("label", "s_waitcnt", "s_nop"):
            pass
        elif m == "s_barrier":
            assert not pend["lgkmcnt"], ("s_barrier with LDS operations in flight: another wave may not see them", pc)
            yield pc
            phase[0] += 1
        elif m == "v_mov_b32":
            V[int(t[1][1:])] = (t[2] if isinstance(t[2], int) else f32bits(t[2]) if isinstance(t[2], float)
                                else SG[int(t[2][1:])] if t[2][0] == "s" else V[int(t[2][1:])])
        elif m == "v_cmp_nlt_f32":
            SG["vcc"] = int(not (fval(t[2]) < fval(t[3])))
        elif m == "v_cmp_lt_f32":
            SG["vcc"] = int(fval(t[2]) < fval(t[3]))
        elif m == "v_cmp_gt_f32":
            SG["vcc"] = int(fval(t[2]) > fval(t[3]))
        elif m == "v_cmp_eq_f32":
            SG["vcc"] = int(fval(t[2]) == fval(t[3]))
        elif m == "v_cmp_neq_f32":
            SG["vcc"] = int(not (fval(t[2]) == fval(t[3])))
        elif m == "v_cndmask_b32_e64":
            bits = lambda x: f32bits(x) if isinstance(x, float) else x if isinstance(x, int) else int(V[int(x[1:])])
            V[int(t[1][1:])] = bits(t[3]) if SG["vcc"] else bits(t[2])
        elif m == "v_cvt_f32_i32":
            setf(t[1], float(sval(t[2])))
        elif m == "v_cndmask_b32":
            src0 = f32bits(t[2]) if isinstance(t[2], float) else t[2] if isinstance(t[2], int) else int(V[int(t[2][1:])])
            V[int(t[1][1:])] = int(V[int(t[3][1:])]) if SG["vcc"] else src0
        elif m == "v_rsq_f32":
            setf(t[1], 1.0 / np.sqrt(np.float64(fval(t[2]))))
        elif m == "v_rcp_f32":
            setf(t[1], 1.0 / np.float64(fval(t[2])))
        elif m == "v_accvgpr_write_b32":
            A[int(t[1][1:])] = V[int(t[2][1:])]
        elif m == "s_mov_b32":
            SG[int(t[1][1:])] = t[2] if isinstance(t[2], int) else sval(t[2])
        elif m == "s_mov_b64":
            lo = int(t[1][2:t[1].index(":")])
            val = sval(t[2])
            SG[lo], SG[lo + 1] = val & 0xFFFFFFFF, val >> 32
        elif m == "s_mul_i32":
            SG[int(t[1][1:])] = (sval(t[2]) * sval(t[3])) & 0xFFFFFFFF
        elif m == "s_mul_hi_u32":
            SG[int(t[1][1:])] = ((sval(t[2]) * sval(t[3])) >> 32) & 0xFFFFFFFF
        elif m == "s_add_u32":
            r = sval(t[2]) + sval(t[3])
            SG[int(t[1][1:])] = r & 0xFFFFFFFF
            scc = r >> 32
        elif m == "s_addc_u32":
            r = sval(t[2]) + sval(t[3]) + scc
            SG[int(t[1][1:])] = r & 0xFFFFFFFF
            scc = r >> 32
        elif m == "s_sub_i32":
            SG[int(t[1][1:])] = (sval(t[2]) - sval(t[3])) & 0xFFFFFFFF
        elif m in ("s_cmp_gt_i32", "s_cmp_lt_i32"):
            a = sval(t[1])
            a = a - (1 << 32) if a & 0x80000000 else a
            scc = int(a > sval(t[2])) if m == "s_cmp_gt_i32" else int(a < sval(t[2]))
        elif m == "s_cmp_lg_u32":
            scc = int(sval(t[1]) != sval(t[2]))
        elif m == "s_branch":
            lab = t[1][:-1]
            pc = min(c for c in labels[lab] if c > pc)
        elif m == "s_cbranch_scc1":
            if scc:
                lab, d = t[1][:-1], t[1][-1]
                cands = labels[lab]
                pc = min(c for c in cands if c > pc) if d == "f" else max(c for c in cands if c < pc)
        elif m == "v_add_u32":
            V[int(t[1][1:])] = (t[2] + int(V[int(t[3][1:])])) & 0xFFFFFFFF
        elif m == "v_and_b32":
            V[int(t[1][1:])] = t[2] & int(V[int(t[3][1:])])
        elif m == "v_readfirstlane_b32":
            SG[int(t[1][1:])] = int(V[int(t[2][1:])])
        elif m == "v_or_b32":
            V[int(t[1][1:])] = int(V[int(t[2][1:])]) | int(V[int(t[3][1:])])
        elif m == "v_cmp_ne_u32":
            SG["vcc"] = int(t[2] != int(V[int(t[3][1:])]))
        elif m == "s_cbranch_vccz":
            if not SG["vcc"]:
                lab, d = t[1][:-1], t[1][-1]
                cands = labels[lab]
                pc = min(c for c in cands if c > pc) if d == "f" else max(c for c in cands if c < pc)
        elif m == "s_cbranch_vccnz":
            if SG["vcc"]:
                lab, d = t[1][:-1], t[1][-1]
                cands = labels[lab]
                pc = min(c for c in cands if c > pc) if d == "f" else max(c for c in cands if c < pc)
        elif m == "v_lshrrev_b32":
            V[int(t[1][1:])] = int(V[int(t[3][1:])]) >> t[2]
        elif m == "global_load_dword":
            arr, row = gaddr(t)
            b = f32bits(float(arr[row]))
            if t[1][0] == "a":
                A[int(t[1][1:])] = b
            else:
                V[int(t[1][1:])] = b
        elif m == "global_store_dword":
            arr, row = gaddr((t[0], None, t[1], t[3], t[4]))
            arr[row] = bits2f(V[int(t[2][1:])])
        elif m == "ds_read_b128":
            lo = int(t[1][2:t[1].index(":")])
            w = ldsword(t[2], t[3])
            for h in range(4):
                V[lo + h] = f32bits(float(lds[w + h]))
                src_word[("v", lo + h)] = w + h
                src_phase[("v", lo + h)] = phase[0]
        elif m == "ds_read_b32":
            w = ldsword(t[2], t[3])
            V[int(t[1][1:])] = f32bits(float(lds[w]))
            src_word[("v", int(t[1][1:]))] = w
            src_phase[("v", int(t[1][1:]))] = phase[0]
        elif m == "ds_write_b128":
            lo = int(t[2][2:t[2].index(":")])
            w = ldsword(t[1], t[3])
            for h in range(4):
                lds[w + h] = bits2f(V[lo + h])
            log_w.update(range(w, w + 4))
        elif m == "ds_write_b32":
            lds[ldsword(t[1], t[3])] = bits2f(V[int(t[2][1:])])
            log_w.add(ldsword(t[1], t[3]))
        elif m == "ds_min_f32":               # LDS float-min atomic (no return): other waves may do the same to the word
            w = ldsword(t[1], t[3])
            lds[w] = min(lds[w], bits2f(V[int(t[2][1:])]))
            log_a.add(w)
        elif m == "ds_write_b64":
            lo = int(t[2][2:t[2].index(":")])
            w = ldsword(t[1], t[3])
            lds[w], lds[w + 1] = bits2f(V[lo]), bits2f(V[lo + 1])
            log_w.update((w, w + 1))
        elif m in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"):
            d = t[-1]
            srcs = t[2:-1]
            dlo = int(t[1][2:t[1].index(":")])

            def half(x, sel):
                lo_ = int(x[2:x.index(":")])
                return bits2f(V[lo_ + sel]) if x[0] == "v" else bits2f(SG[lo_ + sel])
            out2 = []
            for hi in (0, 1):
                sel = d["op_sel_hi"] if hi else d["op_sel"]
                ng = d["neg_hi"] if hi else d["neg_lo"]
                vals = [np.float64(half(x, sel[q])) * (-1 if ng[q] else 1) for q, x in enumerate(srcs)]
                if m == "v_pk_fma_f32":
                    out2.append(f32(vals[0] * vals[1] + vals[2]))
                elif m == "v_pk_mul_f32":
                    out2.append(f32(f32(vals[0]) * f32(vals[1])))
                else:
                    out2.append(f32(f32(vals[0]) + f32(vals[1])))
            V[dlo], V[dlo + 1] = f32bits(float(out2[0])), f32bits(float(out2[1]))     # both halves from the OLD registers
        elif m == "v_accvgpr_read_b32":
            V[int(t[1][1:])] = A[int(t[2][1:])]
        elif m == "v_fma_f32":
            setf(t[1], np.float64(fval(t[2])) * np.float64(fval(t[3])) + np.float64(fval(t[4])))
        elif m == "v_fmac_f32":
            setf(t[1], np.float64(fval(t[2])) * np.float64(fval(t[3])) + np.float64(fval(t[1])))
        elif m == "v_mul_f32":
            setf(t[1], f32(fval(t[2])) * f32(fval(t[3])))
        elif m == "v_sub_f32":
            setf(t[1], f32(fval(t[2])) - f32(fval(t[3])))
        elif m == "v_add_f32":
            setf(t[1], f32(fval(t[2])) + f32(fval(t[3])))
        elif m == "v_max_f32":
            setf(t[1], max(fval(t[2]), fval(t[3])))
        elif m == "v_min_f32":
            setf(t[1], min(fval(t[2]), fval(t[3])))
        elif m == "v_max3_f32":
            setf(t[1], max(fval(t[2]), fval(t[3]), fval(t[4])))
        elif m == "v_min3_f32":
            setf(t[1], min(fval(t[2]), fval(t[3]), fval(t[4])))
        else:
            raise ValueError("unknown instruction %r" % (t,))
        pc += 1
    if count is not None:
        count.append(nexec)          # instructions executed (which variant of a loop ran)
    if ret_agpr:
        return lds, np.array([bits2f(b) for b in A], np.float32)
    return lds


def reference_iterations(p, d, iters, alpha, sigma):
    """float64 statement: iterations of the general OSQP loop on (x, y, z) with factor (L, DI) in the structure's order"""
    s = p.s
    n, m, nk = p.n, p.m, p.nk
    pinv, L_p, L_i = p.pinv, p.L_p, p.L_i
    x, y, z = d["x"].copy(), d["y"].copy(), d["z"].copy()
    for _ in range(iters):
        w = np.zeros(nk)
        for j in range(n):
            w[pinv[j]] = sigma * x[j] - d["q"][j]
        t3 = z - d["rinv"] * y
        for i in range(m):
            w[pinv[n + i]] = t3[i]
        for c in range(nk):
            for j in range(L_p[c], L_p[c + 1]):
                w[L_i[j]] -= d["L"][j] * w[c]
        w *= d["DI"]
        for c in range(nk - 1, -1, -1):
            for j in range(L_p[c], L_p[c + 1]):
                w[c] -= d["L"][j] * w[L_i[j]]
        xn = np.array([alpha * w[pinv[j]] + (1 - alpha) * x[j] for j in range(n)])
        nu = np.array([w[pinv[n + i]] for i in range(m)])
        zt = t3 + d["rinv"] * nu
        tt = alpha * zt + (1 - alpha) * z
        zn = np.minimum(np.maximum(tt + d["rinv"] * y, d["l"]), d["u"])
        y = y + d["rho"] * (tt - zn)
        x, z = xn, zn
    return x, y, z


# ---------------------------------------------------------------------------
# The Ruiz passes of a build-time-known structure (scaling.c:44-156) as one assembly block, fp32
# ---------------------------------------------------------------------------
# hipcc's version of this phase for p5f -- fetching Av / Pv / q one exposed global load at a time, then ten passes over
# arrays that live in scratch -- is 0.74 of the 2.9 ms tick (tools/p5f_timing.py). Here the block fetches its own inputs in
# batches and nothing leaves the chip for the ten passes:
#
#   v4..v(3+m)    Et: the row norms accumulate here while A streams by in column order, then become the row scalings
#   AGPRs         Dt (n), P (nnzP), q (n)
#   LDS words     A (nnzA, read twice and written once per pass through the ring), then the accumulated D (n) and E (m),
#                 c, and on exit P and q: RZ_* below; the C++ side picks everything up from there
#
# Arithmetic = the pass of codegen_qp.emit_structure statement by statement, except that 1/sqrt is v_rsq_f32 + one Newton
# step and the two divisions v_rcp_f32 + Newton (+ a correction step for csum / n): each within an ulp of the IEEE results.
RZ_MIN, RZ_MAX = 1e-4, 1e4
RUIZ_WQ = os.environ.get("UMPC_QP_RUIZ_WQ", "1") == "1"


class QuadWriter:
    """Collects a run of consecutive LDS words in the register quad v[base:base+3] (word w in register base + w % 4) and writes
    them back with the widest instructions: reg(w) names the register the producing instruction must write, done(w, last)
    is called after it."""

    def __init__(self, sc, base):
        self.sc, self.base, self.lo = sc, base, None

    def reg(self, w):
        return self.base + w % 4

    def done(self, w, last=False):
        if self.lo is None:
            self.lo = w
        if w % 4 != 3 and not last:
            return
        lo, q0 = self.lo, w - w % 4
        self.lo = None
        have = set(range(lo, w + 1))
        if have == set(range(q0, q0 + 4)):
            self.sc.lds_write4(q0, self.base)
            return
        for h in (0, 2):
            if {q0 + h, q0 + h + 1} <= have:
                self.sc.lds_write2(q0 + h, self.base + h)
            else:
                for z in (q0 + h, q0 + h + 1):
                    if z in have:
                        self.sc.lds_write(z, self.base + z % 4)


class RuizPlan:
    def __init__(self, s):
        t = s.tables
        self.s, self.n, self.m, self.nnzA, self.nnzP = s, s.n, s.m, s.nnzA, s.nnzP
        self.A_p, self.A_i, self.pidx = list(t["A_p"]), list(t["A_i"]), list(t["pidx"])
        self.V_ET = 4
        self.V_RING = self.V_ET + self.m
        self.V_LAND = self.V_RING + 4 * NRING          # (no stream in this block)
        self.V_AT = self.V_LAND
        self.V_TT = self.V_AT + N_AT
        self.NT = 14
        assert self.V_TT + self.NT <= V_END
        # three register quads that collect consecutive words of A, D and E for one ds_write_b128 each (round 3: the pass
        # wrote its 505 words back one ds_write_b32 at a time)
        self.V_WQ = (max(self.V_TT + self.NT, 211) + 1) // 2 * 2        # (v210 = V_RLANE is an input of the block)
        self.WQ = RUIZ_WQ and self.V_WQ + 12 <= V_END
        self.n_land = 0
        self.A_DT, self.A_P = 0, self.n
        self.A_Q = self.A_P + self.nnzP
        assert self.A_Q + self.n <= 256
        self.LW_A = 0
        self.LW_D = self.nnzA
        self.LW_EV = self.LW_D + self.n
        self.LW_C = self.LW_EV + self.m
        self.LW_P = self.LW_C + 1
        self.LW_Q = self.LW_P + self.nnzP
        self.LW_END = self.LW_Q + self.n
        assert self.LW_END <= 640 and self.nnzA <= 2 * self.m


class ResPlan:
    """Layout of the RESIDUAL stream of a wave (items behind the loop's stream in the same block, [item][lane] floats): the
    equilibrated data the residual block reads after the loop, written by the Ruiz block's epilogue (A, E, D, q, P, c) and
    by the C++ side (the scaled bounds of the equality rows, = their z)."""

    def __init__(self, s, eq_rows, rs0):
        t = s.tables
        n, m = s.n, s.m
        self.s, self.n, self.m, self.rs0 = s, n, m, rs0
        self.A_p, self.A_i, self.pidx = list(t["A_p"]), list(t["A_i"]), list(t["pidx"])
        self.eq = set(int(i) for i in eq_rows)
        idx = rs0
        self.it_A = idx
        idx += s.nnzA
        self.it_rows = idx                  # consumption order from here on
        self.it_ev, self.it_ls = {}, {}
        for i in range(m):
            self.it_ev[i] = idx
            idx += 1
            if i in self.eq:
                self.it_ls[i] = idx
                idx += 1
        self.it_d, self.it_q, self.it_p = {}, {}, {}
        for j in range(n):
            self.it_d[j], self.it_q[j] = idx, idx + 1
            idx += 2
            if self.pidx[j] >= 0:
                self.it_p[j] = idx
                idx += 1
        self.it_c = idx
        self.end = idx + 1


class RuizSplit:
    """The Ruiz passes of ONE group of 64 robots shared by the NW wavefronts of a workgroup (round 4): a batch of 16 384 robots
    is 256 wavefronts, one per CU, and three of a CU's four SIMDs idle while a lone wave issues the 4.1 k instructions of a
    pass. The columns of A are cut into NW contiguous stretches of the chain (a column's place = its first row), wave w owns
    stretch w: its entries of A, P, q, D, its rows' E. All waves address the SAME LDS slots (lane = robot); per pass they
    exchange, through LDS words that are free during the passes (the homes P and q get in the epilogue), (1) the partial
    norms of the few rows whose entries lie in two stretches (max: exact and order-free), (2) the new P entries, which every
    wave sums in the reference's order (the cost normalisation's mean column norm: the same sequence of additions as one
    wave), and max |q| per wave -- two s_barrier per pass. Every word comes out BIT-identical to ruiz_program(s, res)."""

    def __init__(self, p, nw):
        n, m = p.n, p.m
        self.nw = nw
        ent = {j: list(range(p.A_p[j], p.A_p[j + 1])) for j in range(n)}
        key = {j: (min(p.A_i[q] for q in ent[j]) if ent[j] else 0, j) for j in range(n)}
        order = sorted(range(n), key=lambda j: key[j])
        wt = {j: 8 + 5 * len(ent[j]) + (6 if p.pidx[j] >= 0 else 0) for j in range(n)}
        total, acc = float(sum(wt.values())), 0.0
        self.colw = {}
        for j in order:
            self.colw[j] = min(nw - 1, int(acc * nw / total))
            acc += wt[j]
        touch = {i: set() for i in range(m)}
        cnt = {i: {} for i in range(m)}
        for j in range(n):
            for q in ent[j]:
                i = p.A_i[q]
                touch[i].add(self.colw[j])
                cnt[i][self.colw[j]] = cnt[i].get(self.colw[j], 0) + 1
        assert all(touch[i] for i in range(m))
        self.touch = touch
        self.roww = {i: max(sorted(cnt[i]), key=lambda w: cnt[i][w]) for i in range(m)}
        self.shared = [i for i in range(m) if len(touch[i]) > 1]
        # exchange words, from the home of P upward (free until the epilogue)
        w0 = p.LW_P + (-p.LW_P) % 4
        self.X = {}
        for i in self.shared:
            for w in sorted(touch[i]):
                self.X[(i, w)] = w0
                w0 += 1
        w0 += (-w0) % 4
        self.PX = {k: w0 + k for k in range(p.nnzP)}
        w0 += p.nnzP + (-p.nnzP) % 4
        self.QN = {w: w0 + w for w in range(nw)}
        w0 += nw
        assert w0 <= p.LW_END, (w0, p.LW_END)


S_RWAVE = 26                       # ruiz_group_program: s26 = the wave's index in its workgroup
S_AV, S_PV, S_QV = 4, 6, 8          # s[4:5] Av rows, s[6:7] Pv rows, s[8:9] q rows (the block's inputs, [k][B] floats)
S_RSB, V_RLANE = 24, 210           # Ruiz block with a residual stream: s[24:25] = the wave's stream block, v210 = 4*lane
S_RMIN, S_RMAX = 20, 21            # 1e-4, 1e4 (float bits, set by the block)
RUIZ_STAMPS = os.environ.get("UMPC_QP_RUIZ_STAMPS") == "1"     # (diagnostics: see ruiz_program)
RUIZ_HOMES = os.environ.get("UMPC_QP_RUIZ_HOMES", "1") == "1"   # (A/B switch: ruiz_program, shared blocks)
RUIZ_FAST_LIMIT = os.environ.get("UMPC_QP_RUIZ_FAST_LIMIT", "1") == "1"   # (A/B switch: one wave-wide limit_scaling test per pass)
RUIZ_LEAN = os.environ.get("UMPC_QP_RUIZ_LEAN", "1") == "1"     # (A/B switch: in-place operations on the register homes, v_max3_f32)
LOOP_STAMPS = os.environ.get("UMPC_QP_LOOP_STAMPS") == "1"     # (diagnostics: see program(); the residual block copies the items)
STAMP_ITEM0 = 2040                 # spare items at the end of a wave's stream block (codegen_qp.ASM_STREAM_ITEMS = 2048)


def ruiz_program(s, res=None, split=None, wave=0):
    """s11 = number of passes (>= 1). Inputs as above, v0 = 4*robot, v1 = lane LDS address, s10 = 4*B.
    res: a ResPlan -> the epilogue also writes the equilibrated A, E, D, q, P and c to the wave's residual stream.
    split (a RuizSplit), wave: the program of wave `wave` of a workgroup that shares the passes (see RuizSplit)."""
    p = RuizPlan(s)
    n, m = p.n, p.m
    sp = split
    own_col = (lambda j: True) if sp is None else (lambda j: sp.colw[j] == wave)
    own_row = (lambda i: True) if sp is None else (lambda i: sp.roww[i] == wave)
    cols = [j for j in range(n) if own_col(j)]
    rows_own = [i for i in range(m) if own_row(i)]
    aq = [q for j in cols for q in range(p.A_p[j], p.A_p[j + 1])]                 # own entries of A, increasing
    aq_set = set(aq)
    pk = [p.pidx[j] for j in cols if p.pidx[j] >= 0]                              # own entries of P, increasing
    assert aq == sorted(aq) and pk == sorted(pk)
    # A wavefront of a shared block touches a quarter of the rows: the Et registers of the others hold ITS entries of A and of
    # D across the passes (RA, RD) -- no LDS round trip per pass for them (an LDS instruction costs a lone wave ~6 ns), and
    # none at all: nothing downstream of a shared block reads A or D from LDS (the factorisation and the residual block take
    # them from the residual stream). RUIZ_HOMES=0: the LDS form (A/B switch).
    RA, RD, HV = {}, {}, {}

    def asrc(a_):
        return ("V", HV[a_]) if a_ in HV else ("A", a_)

    def awrite(a_, t):
        if a_ in HV:
            if HV[a_] != t:
                e("v_mov_b32", v(HV[a_]), v(t))
        else:
            e("v_accvgpr_write_b32", "a%d" % a_, v(t))

    def adst(a_):
        return v(HV[a_]) if a_ in HV else "a%d" % a_
    if sp is not None and res is not None and RUIZ_HOMES:
        idle = [p.V_ET + i for i in range(m) if wave not in sp.touch[i]]
        assert len(idle) >= len(aq) + len(cols), (len(idle), len(aq), len(cols))
        RA = dict(zip(aq, idle))
        RD = dict(zip(cols, idle[len(aq):]))
        # ... and what is left of them, the two write-combining quads A and D no longer need and the registers behind the
        # quads are the homes of the wave's Dt, q and P words, which the one-wave block keeps in AGPRs (a v_accvgpr_read /
        # write costs a lone wave 3.7 ns, a VGPR operand nothing)
        pool = idle[len(aq) + len(cols):] + list(range(p.V_WQ, p.V_WQ + 8)) + list(range(p.V_WQ + 12, V_END))
        for a_ in [p.A_DT + j for j in cols] + [p.A_Q + j for j in cols] + [p.A_P + k for k in pk]:
            if pool:
                HV[a_] = pool.pop(0)
    e = Emit()
    v = lambda r: "v%d" % r
    T = lambda q: p.V_TT + q
    ET = lambda i: v(p.V_ET + i)
    sMIN, sMAX = "s%d" % S_RMIN, "s%d" % S_RMAX
    ab = lambda x: "|" + x + "|"

    def rowptr(base, row):
        e("s_mul_i32", "s%d" % S_P, "s%d" % S_STRIDE, row)
        e("s_mul_hi_u32", "s%d" % (S_P + 1), "s%d" % S_STRIDE, row)
        e("s_add_u32", "s%d" % S_P, "s%d" % S_P, "s%d" % base)
        e("s_addc_u32", "s%d" % (S_P + 1), "s%d" % (S_P + 1), "s%d" % (base + 1))

    def load_rows(base, rows, dst):
        """global rows `rows` (increasing) of the [row][B] array at s[base:base+1] -> the registers dst(position)"""
        last = None
        for q_, r_ in enumerate(rows):
            if last is None or r_ != last + 1:
                rowptr(base, r_)
            else:
                _adv(e, S_P)
            last = r_
            e("global_load_dword", dst(q_), "v0", "s[%d:%d]" % (S_P, S_P + 1), 0)

    def limit(t, t2):
        """t <- limit_scaling(t) (umpc_bqp_common.h): t < 1e-4 ? 1 : min(t, 1e4)"""
        e("v_cmp_nlt_f32", "vcc", v(t), sMIN)
        e("v_min_f32", v(t2), sMAX, v(t))
        e("v_cndmask_b32", v(t), 1.0, v(t2), "vcc")

    def rsqrt(y, t, a_):
        # v_rsq_f32 alone (1 ulp), as the uprightmpc2 step (asmstep.py) takes it: D and E only precondition, a last-bit
        # difference to the divide + sqrt of the C++ statement moves the unscaled solution at round-off level
        e("v_rsq_f32", v(y), v(t))
        e("s_nop", 0)

    def recip(y, t, a_):
        e("v_rcp_f32", v(y), v(t))
        e("s_nop", 0)
        e("v_fma_f32", v(a_), "-" + v(t), v(y), 1.0)
        e("v_fma_f32", v(y), v(y), v(a_), v(y))

    # (diagnostics, UMPC_QP_RUIZ_STAMPS=1 on a block with a residual stream: 100 MHz stamps -> five intervals in the spare
    # items STAMP_ITEM0.. of the stream -- prologue, sum of the norm phases, sum of the apply phases, P / q / c -> LDS,
    # stream stores; the glue block adds its own and the residual block copies all six over the info rows)
    RST = RUIZ_STAMPS and res is not None and wave == 0
    ST = 60

    def stamp(k):
        if RST:
            e("s_waitcnt", "lgkmcnt(0)")
            e("s_memrealtime", "s[%d:%d]" % (ST + 2 * k, ST + 2 * k + 1))
            e("s_waitcnt", "lgkmcnt(0)")

    def stamp_acc(k_from, k_to, acc):
        if RST:
            e("s_sub_u32", "s%d" % (ST + 2 * k_from), "s%d" % (ST + 2 * k_to), "s%d" % (ST + 2 * k_from))
            e("s_add_u32", "s%d" % acc, "s%d" % acc, "s%d" % (ST + 2 * k_from))
    # ---- prologue: A rows -> LDS through the Et registers (landing zone), P and q -> AGPRs, D = E = 1 in LDS, c = 1
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    stamp(0)
    if RST:
        e("s_mov_b32", "s80", 0)
        e("s_mov_b32", "s81", 0)
    e("v_add_u32", "v%d" % V_B1, 0x10000, "v1")
    e("v_add_u32", "v%d" % V_B2, 0x20000, "v1")
    e("s_mov_b32", sMIN, f32bits(RZ_MIN))
    e("s_mov_b32", sMAX, f32bits(RZ_MAX))
    load_rows(S_PV, pk, lambda q_: adst(p.A_P + pk[q_]))
    load_rows(S_QV, cols, lambda q_: adst(p.A_Q + cols[q_]))
    if RA:
        load_rows(S_AV, aq, lambda q_: v(RA[aq[q_]]))
    else:
        for g in range(0, len(aq), m):
            chunk = aq[g:g + m]
            load_rows(S_AV, chunk, lambda q_: v(p.V_ET + q_))
            e("s_waitcnt", "vmcnt(0)")
            for q_, k in enumerate(chunk):
                base, off = lds_addr(p.LW_A + k)
                e("ds_write_b32", base, v(p.V_ET + q_), off)
    e("v_mov_b32", v(T(13)), 1.0)                                  # c
    for j in RD:
        e("v_mov_b32", v(RD[j]), 1.0)
    for w in [p.LW_D + j for j in cols if j not in RD] + [p.LW_EV + i for i in rows_own]:
        base, off = lds_addr(w)
        e("ds_write_b32", base, v(T(13)), off)
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    stamp(1)
    e("s_mov_b32", "s%d" % S_CNT, "s%d" % S_ITERS)
    e("label", "7")
    sc = Sched(e, p, 0)
    ops = []

    def op(srcs, fn):
        ops.append(dict(srcs=srcs, emit=fn))

    def barrier():
        """every LDS write of this wave has landed, then all waves of the workgroup meet; nothing fetched before is reused"""
        op([], lambda g: (e("s_waitcnt", "lgkmcnt(0)"), e("s_barrier")))
        ops.append(dict(flush=True))
    op([], lambda g: stamp(2))
    # ---- norms: columns in order; the row norms accumulate in the Et registers
    touched = set()
    fast_cols = []                       # register homes that hold column NORMS until the wave-wide limit test below
    for j in cols:
        ents = list(range(p.A_p[j], p.A_p[j + 1]))
        if RUIZ_LEAN and all(q in RA for q in ents) and (p.pidx[j] < 0 or p.A_P + p.pidx[j] in HV):
            # every operand of the column's norm in a register (shared blocks): up to three per v_max3_f32 (max is exact and
            # order-free: the same bits as the chain of v_max_f32 below), the row norms as before
            def fcol(g, j=j, ents=ents, firsts=tuple(p.A_i[q] not in touched for q in ents)):
                opnds = ([HV[p.A_P + p.pidx[j]]] if p.pidx[j] >= 0 else []) + [RA[q] for q in ents]
                head, rest = opnds[:3], opnds[3:]
                nrm = HV[p.A_DT + j] if (RUIZ_FAST_LIMIT and p.A_DT + j in HV) else T(0)      # (limited wave-wide below)
                if not opnds:
                    e("v_mov_b32", v(nrm), 0)
                if len(head) == 3:
                    e("v_max3_f32", v(nrm), ab(v(head[0])), ab(v(head[1])), ab(v(head[2])))
                elif head:
                    e("v_max_f32", v(nrm), ab(v(head[0])), ab(v(head[-1])))
                while rest:
                    if len(rest) >= 2:
                        e("v_max3_f32", v(nrm), v(nrm), ab(v(rest[0])), ab(v(rest[1])))
                        rest = rest[2:]
                    else:
                        e("v_max_f32", v(nrm), v(nrm), ab(v(rest[0])))
                        rest = rest[1:]
                for q, first in zip(ents, firsts):
                    i = p.A_i[q]
                    e("v_max_f32", ET(i), ab(v(RA[q])) if first else ET(i), ab(v(RA[q])))
            op([], fcol)
            touched.update(p.A_i[q] for q in ents)
            if RUIZ_FAST_LIMIT and p.A_DT + j in HV:
                fast_cols.append(HV[p.A_DT + j])
                continue
        else:
            if p.pidx[j] >= 0:
                op([asrc(p.A_P + p.pidx[j])], lambda g: e("v_max_f32", v(T(0)), ab(v(g[0])), ab(v(g[0]))))
            else:
                op([], lambda g: e("v_mov_b32", v(T(0)), 0))
            for q in ents:
                i = p.A_i[q]

                def f(g, i=i, first=i not in touched):
                    e("v_max_f32", v(T(0)), v(T(0)), ab(v(g[0])))
                    e("v_max_f32", ET(i), ab(v(g[0])) if first else ET(i), ab(v(g[0])))
                op([("V", RA[q]) if q in RA else ("L", p.LW_A + q)], f)
                touched.add(i)

        def fin(g, j=j):
            limit(T(0), T(1))
            if RUIZ_LEAN and p.A_DT + j in HV:
                # straight into the column scaling's register home: no copy, and no wait state either (the next VALU
                # instruction starts the next column and does not read it)
                e("v_rsq_f32", v(HV[p.A_DT + j]), v(T(0)))
                return
            rsqrt(T(1), T(0), T(2))
            awrite(p.A_DT + j, T(1))
        op([], fin)
    if sp is None:
        assert len(touched) == m
    else:
        assert touched == set(i for i in range(m) if wave in sp.touch[i])
        # rows with entries in two stretches: the partial norms change hands (max: exact, order-free)
        mine = [i for i in sp.shared if wave in sp.touch[i]]
        for i in mine:
            op([], lambda g, i=i: sc.lds_write(sp.X[(i, wave)], p.V_ET + i))
        barrier()
        for i in mine:
            for w2 in sorted(sp.touch[i] - {wave}):
                op([("L", sp.X[(i, w2)])], lambda g, i=i: e("v_max_f32", ET(i), ET(i), v(g[0])))
    if fast_cols:
        # limit_scaling + 1/sqrt of all the wave's norms at once (as asmstep.Step.limit): running v_min3 / v_max3 over them
        # decide wave-wide whether ANY value of ANY robot needs limiting -- it never does once the data is equilibrated -- and
        # the exact compare / min / select sequence runs only then. Same values either way.
        def flim(g, regs=tuple(fast_cols + [p.V_ET + i for i in sorted(touched)])):
            regs = list(regs)
            mn, mx = T(0), T(2)
            take = (regs + [regs[0], regs[0]])[:3]
            e("v_min3_f32", v(mn), v(take[0]), v(take[1]), v(take[2]))
            e("v_max3_f32", v(mx), v(take[0]), v(take[1]), v(take[2]))
            for q in range(3, len(regs), 2):
                a_, b_ = regs[q], regs[min(q + 1, len(regs) - 1)]
                e("v_min3_f32", v(mn), v(mn), v(a_), v(b_))
                e("v_max3_f32", v(mx), v(mx), v(a_), v(b_))
            e("v_cmp_gt_f32", "vcc", sMIN, v(mn))
            e("s_cbranch_vccnz", "33f")
            e("v_cmp_lt_f32", "vcc", sMAX, v(mx))
            e("s_cbranch_vccz", "34f")
            e("label", "33")
            for r_ in regs:
                limit(r_, T(1))
            e("label", "34")
            for r_ in regs:
                e("v_rsq_f32", v(r_), v(r_))
            e("s_nop", 0)
        op([], flim)
    else:
        for i in sorted(touched):
            def fe(g, i=i):
                limit(p.V_ET + i, T(1))
                e("v_rsq_f32", ET(i), ET(i))        # in place; the next VALU instruction (the next row's compare, or the
            op([], fe)                               # csum initialisation) does not read it: no trans-use wait state needed
    # ---- apply; csum in T(4), qn in T(5), dt of the column in T(7)
    wqa, wqd, wqe = (QuadWriter(sc, p.V_WQ + 4 * z) for z in range(3))
    op([], lambda g: stamp(3))
    op([], lambda g: (e("v_mov_b32", v(T(4)), 0), e("v_mov_b32", v(T(5)), 0)))
    for j in cols:
        lean = RUIZ_LEAN and p.A_DT + j in HV
        DTJ = HV[p.A_DT + j] if lean else T(7)       # dt of the column: its register home itself (shared blocks), else a copy
        if not lean:
            op([asrc(p.A_DT + j)], lambda g: e("v_mov_b32", v(T(7)), v(g[0])))
        if p.pidx[j] >= 0:
            def fp(g, k=p.pidx[j], DTJ=DTJ, lean=lean):
                t = T(6) if sp is None else T(8 + k % 4)
                if lean and p.A_P + k in HV:
                    t = HV[p.A_P + k]             # in place (the same two products)
                e("v_mul_f32", v(t), v(g[0]), v(DTJ))
                e("v_mul_f32", v(t), v(t), v(DTJ))
                awrite(p.A_P + k, t)
                if sp is None:
                    e("v_add_f32", v(T(4)), v(T(4)), ab(v(t)))
                else:
                    sc.lds_write(sp.PX[k], t)       # (every wave sums all of them in the reference's order below)
            op([asrc(p.A_P + p.pidx[j])], fp)
        for q in range(p.A_p[j], p.A_p[j + 1]):
            def fa(g, q=q, i=p.A_i[q], DTJ=DTJ):
                if q in RA:                   # in place, in its register home
                    e("v_mul_f32", v(RA[q]), v(RA[q]), ET(i))
                    e("v_mul_f32", v(RA[q]), v(RA[q]), v(DTJ))
                    return
                t = wqa.reg(p.LW_A + q) if p.WQ else T(8 + q % 4)
                e("v_mul_f32", v(t), v(g[0]), ET(i))
                e("v_mul_f32", v(t), v(t), v(DTJ))
                if p.WQ:
                    wqa.done(p.LW_A + q, q + 1 not in aq_set)
                else:
                    sc.lds_write(p.LW_A + q, t)
            op([("V", RA[q]) if q in RA else ("L", p.LW_A + q)], fa)

        def fq(g, j=j, DTJ=DTJ, lean=lean):
            tq = HV[p.A_Q + j] if (lean and p.A_Q + j in HV) else T(6)         # (in place in its register home)
            e("v_mul_f32", v(tq), v(g[0]), v(DTJ))
            awrite(p.A_Q + j, tq)
            e("v_max_f32", v(T(5)), ab(v(tq)), v(T(5)))
            if j in RD:
                e("v_mul_f32", v(RD[j]), v(DTJ), v(RD[j]))
                return
            t = wqd.reg(p.LW_D + j) if p.WQ else T(12)
            e("v_mul_f32", v(t), v(DTJ), v(g[1]))
            if p.WQ:
                wqd.done(p.LW_D + j, not (j + 1 < n and own_col(j + 1)))
            else:
                sc.lds_write(p.LW_D + j, T(12))
        op([asrc(p.A_Q + j), ("V", RD[j]) if j in RD else ("L", p.LW_D + j)], fq)
    for i in rows_own:
        def fv(g, i=i):
            t = wqe.reg(p.LW_EV + i) if p.WQ else T(8 + i % 4)
            e("v_mul_f32", v(t), ET(i), v(g[0]))
            if p.WQ:
                wqe.done(p.LW_EV + i, not (i + 1 < m and own_row(i + 1)))
            else:
                sc.lds_write(p.LW_EV + i, t)
        op([("L", p.LW_EV + i)], fv)
    if sp is not None:
        # the cost normalisation needs the sum of |P_jj| over ALL columns in the reference's order and max |q_j|
        op([], lambda g: sc.lds_write(sp.QN[wave], T(5)))
        barrier()
        for k in range(p.nnzP):
            op([("L", sp.PX[k])], lambda g: e("v_add_f32", v(T(4)), v(T(4)), ab(v(g[0]))))
        for w2 in range(sp.nw):
            if w2 != wave:
                op([("L", sp.QN[w2])], lambda g: e("v_max_f32", v(T(5)), v(T(5)), v(g[0])))

    # ---- cost scaling: ct = 1 / limit(max(csum / n, limit(qn)))
    def cost(g):
        e("v_mov_b32", v(T(0)), f32bits(float(n)))
        recip(T(1), T(0), T(2))
        e("v_mul_f32", v(T(2)), v(T(4)), v(T(1)))
        e("v_fma_f32", v(T(3)), "-" + v(T(0)), v(T(2)), v(T(4)))
        e("v_fma_f32", v(T(4)), v(T(3)), v(T(1)), v(T(2)))           # csum / n
        limit(T(5), T(0))
        e("v_max_f32", v(T(4)), v(T(4)), v(T(5)))
        limit(T(4), T(0))
        recip(T(5), T(4), T(0))                                        # ct
        e("v_mul_f32", v(T(13)), v(T(13)), v(T(5)))
    op([], cost)
    for k in pk + [p.nnzP + j for j in cols]:
        def fs(g, k=k):
            t = T(8 + k % 4)
            if p.A_P + k in HV:
                e("v_mul_f32", v(HV[p.A_P + k]), v(g[0]), v(T(5)))     # in place, in its register home
                return
            e("v_mul_f32", v(t), v(g[0]), v(T(5)))
            e("v_accvgpr_write_b32", "a%d" % (p.A_P + k), v(t))        # (q follows P in the AGPRs)
        op([asrc(p.A_P + k)], fs)
    sc.run(ops)
    e("s_waitcnt", "lgkmcnt(0)")
    stamp(4)
    stamp_acc(2, 3, 80)
    stamp_acc(3, 4, 81)
    e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_CNT, 1)
    e("s_cmp_gt_i32", "s%d" % S_CNT, 0)
    e("s_cbranch_scc1", "7b")
    stamp(5)
    if sp is not None:
        e("s_barrier")            # (the exchange words are the homes of P and q: every wave has read them)
    # ---- epilogue: c, P, q -> LDS
    if wave == 0:
        base, off = lds_addr(p.LW_C)
        e("ds_write_b32", base, v(T(13)), off)
    for k in pk + [p.nnzP + j for j in cols]:
        t = T(k % 8)
        if p.A_P + k in HV:
            t = HV[p.A_P + k]
        else:
            e("v_accvgpr_read_b32", v(t), "a%d" % (p.A_P + k))
        base, off = lds_addr(p.LW_P + k)
        e("ds_write_b32", base, v(t), off)
    if res is not None:
        assert V_RLANE >= p.V_TT + p.NT
        e("s_waitcnt", "lgkmcnt(0)")
        stamp(6)

        rs_blk = [None]

        def put(item, reg):
            if item // BLOCK != rs_blk[0]:           # (consecutive items share the block pointer)
                rs_blk[0] = item // BLOCK
                e("s_add_u32", "s%d" % S_P, "s%d" % S_RSB, rs_blk[0] * BLOCK * 256)
                e("s_addc_u32", "s%d" % (S_P + 1), "s%d" % (S_RSB + 1), 0)
            e("global_store_dword", "v%d" % V_RLANE, v(reg), "s[%d:%d]" % (S_P, S_P + 1), (item % BLOCK) * 256)
        # LDS words (A, D, E) through the ring registers, a group of quads at a time
        for k in aq:
            if k in RA:
                put(res.it_A + k, RA[k])
        for j in cols:
            if j in RD:
                put(res.it_d[j], RD[j])
        words = [(p.LW_A + k, res.it_A + k) for k in aq if k not in RA] + [(p.LW_D + j, res.it_d[j]) for j in cols if j not in RD] + \
                [(p.LW_EV + i, res.it_ev[i]) for i in rows_own]
        item_of = dict(words)
        quads = sorted(set(w >> 2 for w, _ in words))
        for g in range(0, len(quads), NRING):
            grp = quads[g:g + NRING]
            for q, qd in enumerate(grp):
                base, off = lds_addr(4 * qd)
                e("ds_read_b128", "v[%d:%d]" % (p.V_RING + 4 * q, p.V_RING + 4 * q + 3), base, off)
            e("s_waitcnt", "lgkmcnt(0)")
            for q, qd in enumerate(grp):
                for h in range(4):
                    if 4 * qd + h in item_of:
                        put(item_of[4 * qd + h], p.V_RING + 4 * q + h)
        for j in cols:
            t_ = T(j % 8)
            if p.A_Q + j in HV:
                t_ = HV[p.A_Q + j]
            else:
                e("v_accvgpr_read_b32", v(t_), "a%d" % (p.A_Q + j))
            put(res.it_q[j], t_)
            if p.pidx[j] >= 0:
                t2 = T(8 + j % 4)
                if p.A_P + p.pidx[j] in HV:
                    t2 = HV[p.A_P + p.pidx[j]]
                else:
                    e("v_accvgpr_read_b32", v(t2), "a%d" % (p.A_P + p.pidx[j]))
                put(res.it_p[j], t2)
        if wave == 0:
            put(res.it_c, T(13))
    if RST:
        e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
        stamp(7)
        # intervals: 0 prologue (t1 - t0), 1 sum norms (s80), 2 sum apply (s81), 3 P / q / c -> LDS (t6 - t5), 4 stream (t7 - t6)
        e("s_sub_u32", "s%d" % (ST + 0), "s%d" % (ST + 2), "s%d" % (ST + 0))
        e("s_sub_u32", "s%d" % (ST + 14), "s%d" % (ST + 14), "s%d" % (ST + 12))
        e("s_sub_u32", "s%d" % (ST + 12), "s%d" % (ST + 12), "s%d" % (ST + 10))
        for k, sreg in enumerate((ST + 0, 80, 81, ST + 12, ST + 14)):
            e("v_cvt_f32_u32", v(T(k)), "s%d" % sreg)
            put(STAMP_ITEM0 + k, T(k))
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    if sp is not None:
        e("s_barrier")            # everything the block leaves (LDS, stream) is there for whichever wave reads it next
    return e.ins, p


def ruiz_group_program(s, res=None, nw=4):
    """The Ruiz block for a workgroup of nw wavefronts that share 64 robots (RuizSplit): s26 = the wave's index; the other
    inputs as ruiz_program. Every wave executes the same number of s_barrier."""
    p = RuizPlan(s)
    sp = RuizSplit(p, nw)
    e = Emit()
    for w in range(nw):
        if w < nw - 1:
            e("s_cmp_lg_u32", "s%d" % S_RWAVE, w)
            e("s_cbranch_scc1", "8f")
        ins, _ = ruiz_program(s, res, sp, w)
        e.ins.extend(ins)
        if w < nw - 1:
            e("s_branch", "9f")
            e("label", "8")
    e("label", "9")
    return e.ins, p, sp


# ---------------------------------------------------------------------------
# Batched row loader: [row][B] arrays -> LDS words, one round trip
# ---------------------------------------------------------------------------
# hipcc fetches the ~900 warm-start / bound words of a p5f step one exposed global load at a time (load, wait, spill):
# 0.45 ms of the tick for a lone wave. This block issues all loads of up to three arrays back to back into registers,
# waits once, and leaves the words in LDS (float4-interleaved) where the C++ side picks them up at LDS latency.
def loader_program(groups):
    """groups: [(n_rows, first LDS word)] for the arrays at s[4:5], s[6:7], s[8:9]; v0 = 4*robot, v1 = lane LDS address,
    s10 = 4*B"""
    e = Emit()
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    e("v_add_u32", "v%d" % V_B1, 0x10000, "v1")
    e("v_add_u32", "v%d" % V_B2, 0x20000, "v1")
    items = []
    for q, (nrows, w0) in enumerate(groups):
        items += [((S_W, S_S, 8)[q], r, w0 + r) for r in range(nrows)]
    cap = V_END - 6
    for g in range(0, len(items), cap):
        grp = items[g:g + cap]
        last = None
        for k, (base, row, word) in enumerate(grp):
            if last is None or last[0] != base or last[1] + 1 != row:
                e("s_mul_i32", "s%d" % S_P, "s%d" % S_STRIDE, row)
                e("s_mul_hi_u32", "s%d" % (S_P + 1), "s%d" % S_STRIDE, row)
                e("s_add_u32", "s%d" % S_P, "s%d" % S_P, "s%d" % base)
                e("s_addc_u32", "s%d" % (S_P + 1), "s%d" % (S_P + 1), "s%d" % (base + 1))
            else:
                _adv(e, S_P)
            last = (base, row)
            e("global_load_dword", "v%d" % (6 + k), "v0", "s[%d:%d]" % (S_P, S_P + 1), 0)
        e("s_waitcnt", "vmcnt(0)")
        k = 0
        while k < len(grp):
            word = grp[k][2]
            base, off = lds_addr(word)
            if word % 4 == 0 and k + 3 < len(grp) and all(grp[k + z][2] == word + z for z in range(4)) and (6 + k) % 2 == 0:
                e("ds_write_b128", base, "v[%d:%d]" % (6 + k, 6 + k + 3), off)
                k += 4
            else:
                e("ds_write_b32", base, "v%d" % (6 + k), off)
                k += 1
        e("s_waitcnt", "lgkmcnt(0)")
    return e.ins


# ---------------------------------------------------------------------------
# Residuals, termination test and the solution stores after the loop (auxil.c:243-307, osqp.c:524-573), fp32
# ---------------------------------------------------------------------------
# hipcc's version of this phase is ~0.45 ms of the p5f tick (arrays in scratch, one exposed access at a time). This block
# reads x, y, z where the loop left them (LDS), the equilibrated A / E / D / q / P / c from the wave's residual stream
# (ResPlan: A into AGPRs up front, the rest through the landing registers in consumption order), forms
#     pri_res = |E^-1 (A x - z)|_inf,  dua_res = |D^-1 (q + P x + A' y)|_inf / c   and the norms of the relative tolerances,
# stores x, y, z, D x, E y / c to the caller's rows, and decides ONLY the common case: every robot of the wave passes the
# termination test at the strict tolerances -> status 1 and the info rows are written and LDS word RES_FLAG is 1. Otherwise
# the flag is 0 and the C++ side redoes the phase for the wave (certificates, inaccurate statuses, cold start: unchanged).
RES_FLAG = 639
S_XO, S_YO, S_ZO, S_SX, S_SY, S_ST, S_IN = 24, 26, 28, 30, 32, 34, 36   # pointer pairs: x, y, z, sol_x, sol_y, status, info rows
S_EPSA, S_EPSR, S_MAXIT = 38, 39, 40                                    # eps_abs, eps_rel (float bits), max_iter (int)
S_EP = 54                                                                # pointer pair: the caller's Eprev rows (E of this solve)


class _ResRegs:
    def __init__(self, m):
        self.V_ACC = 6          # (even: the LDS ring behind it holds 4-register tuples)
        self.V_RING = self.V_ACC + m
        self.V_LAND = self.V_RING + 4 * NRING
        self.V_AT = self.V_LAND + NLAND
        self.V_TT = self.V_AT + N_AT
        self.NT = 12
        assert self.V_TT + self.NT <= V_END, self.V_TT + self.NT
        self.n_land = 0


def res_program(s, eq_rows, ap, res, own=ALL, nw=1):
    """ap: the loop's Plan (LDS words of x, y, z), res: the ResPlan. v0 = 4*robot, v1 = lane LDS address, v4 = 4*lane,
    s[6:7] = the wave's stream block, s10 = 4*B.
    own, nw: the copy of the block one of nw wavefronts runs on ITS rows (pass 1: A x of a row, accumulated over the columns in
    the one-wave block's order) and ITS columns (pass 2: A' y of a column, likewise) -- any rows, any columns: x and y are
    in LDS for everybody, so the split is an even one (own.row / own.var: ResSplit); the six partial norms (max: exact)
    meet in LDS between two barriers and wavefront 0 does the termination test."""
    n, m = s.n, s.m
    R = _ResRegs(m)
    A_p_ = res.A_p
    own_a = [k for j in range(n) for k in range(A_p_[j], A_p_[j + 1]) if own.var(j) or own.row(res.A_i[k])]
    land = [it for it in range(res.it_rows, res.it_c + 1)
            if any(it == res.it_ev[i] or it == res.it_ls.get(i) for i in range(m) if own.row(i)) or
            any(it in (res.it_d[j], res.it_q[j], res.it_p.get(j)) for j in range(n) if own.var(j))]
    lidx = {it: q for q, it in enumerate(land)}
    R.n_land = len(land)                           # the landing stream: the own items among it_rows .. it_c, in stream order
    e = Emit()
    v = lambda r: "v%d" % r
    T = lambda q: R.V_TT + q
    ACC = lambda i: v(R.V_ACC + i)
    ab = lambda x: "|" + x + "|"
    PRI, NZ, NAX, DUA, NQ, NATY, NPX, CINV, C = (T(q) for q in range(9))     # T(9..11): scratch

    def recip(y, t, a_):
        e("v_rcp_f32", v(y), v(t))
        e("s_nop", 0)
        e("v_fma_f32", v(a_), "-" + v(t), v(y), 1.0)
        e("v_fma_f32", v(y), v(y), v(a_), v(y))

    # (diagnostics, UMPC_QP_RES_STAMPS=1: 100 MHz stamps at the block's internal boundaries; the six intervals replace the
    # info rows -- A loads, pass 1, rows, y -> accumulators, columns, termination test)
    STAMPS = os.environ.get("UMPC_QP_RES_STAMPS") == "1" and own.wave == 0
    S_STAMP = 60

    def stamp(k):
        if STAMPS:
            e("s_waitcnt", "lgkmcnt(0)")
            e("s_memrealtime", "s[%d:%d]" % (S_STAMP + 2 * k, S_STAMP + 2 * k + 1))
            e("s_waitcnt", "lgkmcnt(0)")

    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    stamp(0)
    e("v_add_u32", "v%d" % V_B1, 0x10000, "v1")
    e("v_add_u32", "v%d" % V_B2, 0x20000, "v1")
    # A -> AGPRs (direct loads), c
    blk_a = None
    for k in own_a:
        it = res.it_A + k
        if it // BLOCK != blk_a:
            blk_a = it // BLOCK
            e("s_add_u32", "s%d" % S_SP, "s%d" % S_S, blk_a * BLOCK * 256)
            e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_S + 1), 0)
        e("global_load_dword", "a%d" % k, "v%d" % V_LANE, "s[%d:%d]" % (S_SP, S_SP + 1), (it % BLOCK) * 256)
    e("s_add_u32", "s%d" % S_SP, "s%d" % S_S, (res.it_c // BLOCK) * BLOCK * 256)
    e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_S + 1), 0)
    e("global_load_dword", v(C), "v%d" % V_LANE, "s[%d:%d]" % (S_SP, S_SP + 1), (res.it_c % BLOCK) * 256)
    e("s_waitcnt", "vmcnt(0)")
    recip(CINV, C, T(9))
    for r in (PRI, NZ, NAX, DUA, NQ, NATY, NPX):
        e("v_mov_b32", v(r), 0)
    stamp(1)

    class P_:               # what Sched needs
        pass
    pl = P_()
    pl.V_RING, pl.V_LAND, pl.V_AT, pl.n_land = R.V_RING, R.V_LAND, R.V_AT, R.n_land
    pl.land_map = None if own.all else [it - res.it_rows for it in land]
    # (diagnostics: fewer landing registers in use -> fewer stream loads in flight; if the block's time follows 1 / this,
    # it is bound by the latency of its landing stream)
    pl.NLAND = int(os.environ.get("UMPC_QP_RES_NLAND", str(NLAND)))
    assert 8 <= pl.NLAND <= NLAND
    sc = Sched(e, pl, 0)
    # the landing stream starts at item it_rows: pointer and block bookkeeping relative to it
    e("s_add_u32", "s%d" % S_SP, "s%d" % S_S, res.it_rows * 256)
    e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_S + 1), 0)
    SI = lambda item: ("S", lidx[item])
    ops = []

    def op(srcs, fn):
        ops.append(dict(srcs=srcs, emit=fn))
    # store pointers: running copies of the caller's row pointers (x, y, z, sol_x, sol_y, info), advanced after each store
    PTR = {"x": 42, "y": 44, "z": 46, "sx": 48, "sy": 50, "in": 52, "ep": 56}
    for name, src in (("x", S_XO), ("y", S_YO), ("z", S_ZO), ("sx", S_SX), ("sy", S_SY), ("in", S_IN), ("ep", S_EP)):
        e("s_mov_b64", "s[%d:%d]" % (PTR[name], PTR[name] + 1), "s[%d:%d]" % (src, src + 1))

    cur_row = {}

    def store(which, reg, row=None):
        b = PTR[which]
        if row is not None and row != cur_row.get(which, 0):        # (a wave that stores a subset of the rows: skip ahead)
            d_ = row - cur_row.get(which, 0)
            assert d_ > 0
            e("s_mul_i32", "s58", "s%d" % S_STRIDE, d_)
            e("s_mul_hi_u32", "s59", "s%d" % S_STRIDE, d_)
            e("s_add_u32", "s%d" % b, "s%d" % b, "s58")
            e("s_addc_u32", "s%d" % (b + 1), "s%d" % (b + 1), "s59")
        cur_row[which] = (row if row is not None else cur_row.get(which, 0)) + 1
        # Stores count in vmcnt like loads, in issue order: the scheduler must know about them, or its `vmcnt(N)` before a
        # landing item (N = the LOADS issued since) also drains every store issued since -- an exposed HBM write latency per
        # row (round 2: 82 us for this 7.7 k-instruction block). Registered as VMEM operations nobody waits for.
        if os.environ.get("UMPC_QP_RES_NOSTORE") == "1":          # (timing experiment: wrong results)
            return
        sc.vm_at[sc.nvm] = len(e.ins)
        sc.nvm += 1
        e("global_store_dword", "v0", v(reg), "s[%d:%d]" % (b, b + 1), 0)
        e("s_add_u32", "s%d" % b, "s%d" % b, "s%d" % S_STRIDE)
        e("s_addc_u32", "s%d" % (b + 1), "s%d" % (b + 1), 0)
    # ---- pass 1: A x by columns into the row accumulators
    touched = set()
    for j in range(n):
        for q in range(res.A_p[j], res.A_p[j + 1]):
            i = res.A_i[q]
            if not own.row(i):
                continue

            def f(g, i=i, first=i not in touched):
                if first:
                    e("v_mul_f32", ACC(i), v(g[0]), v(g[1]))
                else:
                    e("v_fmac_f32", ACC(i), v(g[0]), v(g[1]))
            op([("A", q), ("L", ap.LW_X + j)], f)
            touched.add(i)
    assert touched == set(i for i in range(m) if own.row(i))
    op([], lambda g: stamp(2))
    # rows: residual entries, E y / c, stores of y, z
    for i in sorted(touched):
        zsrc = SI(res.it_ls[i]) if i in res.eq else ("L", ap.LW_Z + ap.zpos[i])
        srcs = [SI(res.it_ev[i]), ("L", ap.LW_Y + i), zsrc]
        if i in res.eq:           # stream items must be named in consumption order
            srcs = [SI(res.it_ev[i]), SI(res.it_ls[i]), ("L", ap.LW_Y + i)]

        def f(g, i=i, eq=i in res.eq):
            ev, z, y = (g[0], g[1], g[2]) if eq else (g[0], g[2], g[1])
            e("v_rcp_f32", v(T(9)), v(ev))                            # 1 / E_i (1 ulp: it only unscales norms that are
            e("s_nop", 0)                                             # compared with tolerances)
            e("v_sub_f32", v(T(10)), ACC(i), v(z))
            e("v_mul_f32", v(T(10)), v(T(9)), v(T(10)))
            e("v_max_f32", v(PRI), v(PRI), ab(v(T(10))))
            e("v_mul_f32", v(T(10)), v(T(9)), v(z))
            e("v_max_f32", v(NZ), v(NZ), ab(v(T(10))))
            e("v_mul_f32", v(T(10)), v(T(9)), ACC(i))
            e("v_max_f32", v(NAX), v(NAX), ab(v(T(10))))
            e("v_mul_f32", v(T(11)), v(y), v(ev))
            e("v_mul_f32", v(T(11)), v(T(11)), v(CINV))              # sol_y = (y E) / c
            store("y", y, i)
            store("z", z, i)
            store("sy", T(11), i)
            store("ep", ev, i)
        op(srcs, f)
    op([], lambda g: e("v_max_f32", v(NZ), v(NZ), v(NAX)))            # prim_rel; NAX is a temporary from here on
    op([], lambda g: stamp(3))
    # ---- pass 2: y into the accumulator registers, then columns: A' y, P x, q
    yrows = sorted(set(res.A_i[q] for j in range(n) if own.var(j) for q in range(res.A_p[j], res.A_p[j + 1])))
    for i in yrows:
        op([("L", ap.LW_Y + i)], lambda g, i=i: e("v_mov_b32", ACC(i), v(g[0])))
    op([], lambda g: stamp(4))
    for j in range(n):
        if not own.var(j):
            continue
        cols = list(range(res.A_p[j], res.A_p[j + 1]))
        for qn, q in enumerate(cols):
            op([("A", q)], lambda g, q=q, qn=qn: e("v_mul_f32" if qn == 0 else "v_fmac_f32", v(T(9)), v(g[0]), ACC(res.A_i[q])))
        if not cols:
            op([], lambda g: e("v_mov_b32", v(T(9)), 0))
        has_p = j in res.it_p
        srcs = [SI(res.it_d[j]), SI(res.it_q[j])] + ([SI(res.it_p[j])] if has_p else []) + [("L", ap.LW_X + j)]

        def f(g, has_p=has_p, j=j):
            d, qv, x = g[0], g[1], g[-1]
            if has_p:
                e("v_mul_f32", v(T(10)), v(g[2]), v(x))              # px
                e("v_add_f32", v(T(11)), v(qv), v(T(10)))
                scratch = g[2]                                        # (P_j's landing register is dead now)
            else:
                e("v_mov_b32", v(T(11)), v(qv))
                scratch = T(10)
            e("v_add_f32", v(T(11)), v(T(11)), v(T(9)))               # (q + P x) + A' y     [T(9) = A' y]
            e("v_mul_f32", v(NAX), v(x), v(d))                        # sol_x = x D
            store("x", x, j)
            store("sx", NAX, j)
            e("v_rcp_f32", v(NAX), v(d))                              # 1 / D_j (likewise)
            e("s_nop", 0)
            e("v_mul_f32", v(T(11)), v(NAX), v(T(11)))
            e("v_max_f32", v(DUA), v(DUA), ab(v(T(11))))
            e("v_mul_f32", v(T(11)), v(NAX), v(qv))
            e("v_max_f32", v(NQ), v(NQ), ab(v(T(11))))
            e("v_mul_f32", v(T(11)), v(NAX), v(T(9)))
            e("v_max_f32", v(NATY), v(NATY), ab(v(T(11))))
            if has_p:
                e("v_mul_f32", v(T(11)), v(NAX), v(T(10)))
                e("v_max_f32", v(NPX), v(NPX), ab(v(T(11))))
        op(srcs, f)
    sc.run(ops)
    stamp(5)
    if nw > 1:
        # the partial norms of every wave -> LDS (words that only the loop used), barrier, wavefront 0 folds them (max)
        # (exchange words: what nothing uses after the loop -- the tail of the L block behind delta_y and the words behind z,
        # which held constants of the loop)
        e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
        parts = (PRI, NZ, DUA, NQ, NATY, NPX)
        free_w = list(range(ap.LW_DY + m, ap.LW_X)) + list(range(ap.LW_END, LW_FLAGS))
        assert len(free_w) >= len(parts) * nw, (len(free_w), ap.LW_DY + m, ap.LW_X, ap.LW_END)
        xw = lambda w: free_w[len(parts) * w:len(parts) * (w + 1)]
        for q, reg in enumerate(parts):
            base, off = lds_addr(xw(own.wave)[q])
            e("ds_write_b32", base, v(reg), off)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_barrier")
        if own.wave != 0:
            e("s_barrier")
            return e.ins, R
        for w in range(1, nw):
            for q, reg in enumerate(parts):
                base, off = lds_addr(xw(w)[q])
                e("ds_read_b32", v(R.V_RING + q), base, off)
            e("s_waitcnt", "lgkmcnt(0)")
            for q, reg in enumerate(parts):
                e("v_max_f32", v(reg), v(reg), v(R.V_RING + q))
    # ---- termination test at the strict tolerances (osqp.c:524-573), flag, status and info rows
    e("v_mul_f32", v(DUA), v(CINV), v(DUA))
    e("v_max_f32", v(NQ), v(NQ), v(NATY))
    e("v_max_f32", v(NQ), v(NQ), v(NPX))
    e("v_mul_f32", v(NQ), v(NQ), v(CINV))                              # dual_rel (prim_rel is in NZ)
    e("v_mov_b32", v(T(9)), "s%d" % S_EPSA)
    e("v_fma_f32", v(T(10)), "s%d" % S_EPSR, v(NZ), v(T(9)))
    e("v_fma_f32", v(T(11)), "s%d" % S_EPSR, v(NQ), v(T(9)))
    e("v_cmp_lt_f32", "vcc", v(PRI), v(T(10)))
    e("v_cndmask_b32_e64", v(T(9)), 0, 1.0, "vcc")
    e("v_cmp_lt_f32", "vcc", v(DUA), v(T(11)))
    e("v_cndmask_b32", v(T(10)), 0, v(T(9)), "vcc")                    # 1.0 iff both tests pass
    base, off = lds_addr(RES_FLAG)
    e("ds_write_b32", base, v(T(10)), off)
    e("v_mov_b32", v(T(9)), 1)
    e("global_store_dword", "v0", v(T(9)), "s[%d:%d]" % (S_ST, S_ST + 1), 0)
    e("v_cvt_f32_i32", v(NAX), "s%d" % S_MAXIT)
    e("v_mov_b32", v(NATY), 0)
    if (RUIZ_STAMPS or LOOP_STAMPS) and not STAMPS:
        e("s_waitcnt", "vmcnt(0)")
        e("s_add_u32", "s%d" % S_SP, "s%d" % S_S, (STAMP_ITEM0 // BLOCK) * BLOCK * 256)
        e("s_addc_u32", "s%d" % (S_SP + 1), "s%d" % (S_S + 1), 0)
        for k in range(6):
            e("global_load_dword", v(T(k)), "v%d" % V_LANE, "s[%d:%d]" % (S_SP, S_SP + 1), ((STAMP_ITEM0 + k) % BLOCK) * 256)
        e("s_waitcnt", "vmcnt(0)")
        for k in range(6):
            store("in", T(k))
    if STAMPS:
        e("s_waitcnt", "vmcnt(0)")
        stamp(6)
        for k in range(6):
            e("s_sub_u32", "s%d" % (S_STAMP + 2 * k), "s%d" % (S_STAMP + 2 * k + 2), "s%d" % (S_STAMP + 2 * k))
            e("v_cvt_f32_u32", v(T(9)), "s%d" % (S_STAMP + 2 * k))
            store("in", T(9))
            e("s_nop", 1)
    for reg in (() if STAMPS or RUIZ_STAMPS or LOOP_STAMPS else (PRI, DUA, C, NATY, NAX, NATY)):     # info rows: pri, dua, c, 0 (no zero pivot), max_iter, 0
        store("in", reg)
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    if nw > 1:
        e("s_barrier")              # RES_FLAG is there for every wavefront
    return e.ins, R


def res_group_program(s, eq_rows, ap, res, nw=4):
    """The residual block for a workgroup of nw wavefronts that own the same 64 robots, each a quarter of the rows and a
    quarter of the columns; s41 = the wavefront's index, the other inputs as res_program. Two barriers."""
    e = Emit()
    for w in range(nw):
        if w < nw - 1:
            e("s_cmp_lg_u32", "s%d" % S_XWAVE, w)
            e("s_cbranch_scc1", "48f")
        # an even split: rows i * nw // m == w for pass 1 and the row phase, columns j * nw // n == w for pass 2
        own = Own([j * nw // s.n for j in range(s.n)], [i * nw // s.m for i in range(s.m)], None, w)
        ins, R = res_program(s, eq_rows, ap, res, own, nw)
        e.ins.extend(ins)
        if w < nw - 1:
            e("s_branch", "49f")
            e("label", "48")
    e("label", "49")
    return e.ins, R


S_XWAVE = 41                       # res_group_program: s41 = the wave's index in its workgroup


# ---------------------------------------------------------------------------
# KKT fill + LDL' (kkt.c:184-222, qdldl.c:86-247) of a build-time-known structure, fp32, into the loop's homes
# ---------------------------------------------------------------------------
def factor_emit(e, s, p, lw_a, v_p, v_rinv, gen_pos, s_sigma, s_rinveq, v_pool, v_pin, v_fmax, own=ALL):
    """Emits the up-looking factorisation recorded in s.factor_ops: -L goes to the loop's LDS words (p.lpos), 1/D to the
    loop's AGPRs (a[k], permuted index k); v_fmax accumulates min |d_k| (0 = a zero pivot, qdldl.c:221-224).
    lw_a: LDS word of A entry 0 (the equilibrated A, CSC order); v_p: first VGPR of P (nnzP); v_rinv: first VGPR of 1/rho of
    the inequality rows in gen_pos order; equality rows use s_rinveq; v_pool: list of temporaries; v_pin: temporaries that
    keep the L entries a later row's elimination multiplies again."""
    n, nk = s.n, s.nk
    t = s.tables
    pidx = list(t["pidx"])
    v = lambda r: "v%d" % r
    fops = [op_ for op_ in s.factor_ops if own.fk(op_["k"])]     # (components factorise independently: LoopSplit)
    reused = sorted({j for op_ in fops for (_, upd, _) in op_["elim"] for (j, _) in upd})
    assert len(reused) <= len(v_pin), (len(reused), len(v_pin))
    pin = {j: v_pin[q] for q, j in enumerate(reused)}

    class P_:
        pass
    pl = P_()
    pl.V_RING, pl.V_LAND, pl.V_AT, pl.n_land = p.V_RING, p.V_LAND, p.V_AT, 0
    sc = Sched(e, pl, 0, la=0)
    ops = []
    free = list(v_pool)
    T_DK, T_LV, T_A = free.pop(), free.pop(), free.pop()

    def op(srcs, fn):
        ops.append(dict(srcs=srcs, emit=fn))
    split = own.xbar and FACTOR_SPLIT
    met = [not split]

    def meet():
        """a cut component whose halves factorise their own subtrees: before the separator's rows the waves meet once -- half B
        has left 1/D of its columns that reach into the separator in LDS, and its entries of L that the separator's rows
        multiply again are where every entry of L goes"""
        for k_, word in sorted(own.hand_out.items()):
            def fo(g, k_=k_, word=word):
                sc.lds_write(word, g[0])
            op([("A", k_)], fo)
        op([], lambda g: (e("s_waitcnt", "lgkmcnt(0)"), e("s_barrier")))
        ops.append(dict(flush=True))
        for k_, word in sorted(own.hand_in.items()):
            op([("L", word)], lambda g, k_=k_: e("v_accvgpr_write_b32", "a%d" % k_, v(g[0])))
        mine = {op_["k"] for op_ in fops}
        theirs = sorted(j for j in pin if not any(new == j for op_ in fops for (_, _, new) in op_["elim"]))
        for j in theirs:
            op([("L", p.LW_L + p.lpos[j])], lambda g, j=j: e("v_mov_b32", v(pin[j]), v(g[0])))
        met[0] = True
    fops = [op_ for op_ in fops if op_["k"] not in own.ftop] + [op_ for op_ in fops if op_["k"] in own.ftop]
    for op_ in fops:
        k = op_["k"]
        if not met[0] and k in own.ftop:
            meet()
        yv = {}
        for (bb, pk) in op_["init"]:
            reg = free.pop()
            yv[bb] = reg
            op([("L", lw_a + s.K_src[pk][1])], lambda g, reg=reg: e("v_mov_b32", v(reg), v(g[0])))
        orig = s.perm[k]
        if orig < n:
            if pidx[orig] >= 0:
                op([], lambda g, r=v_p + pidx[orig]: e("v_add_f32", v(T_DK), "s%d" % s_sigma, v(r)))
            else:
                op([], lambda g: e("v_mov_b32", v(T_DK), "s%d" % s_sigma))
        elif (orig - n) in gen_pos:
            op([], lambda g, r=v_rinv + gen_pos[orig - n]: e("v_mul_f32", v(T_DK), -1.0, v(r)))
        else:
            op([], lambda g: e("v_mul_f32", v(T_DK), -1.0, "s%d" % s_rinveq))
        for (cidx, upd, new) in op_["elim"]:
            for (j, row) in upd:
                if row in yv:
                    op([], lambda g, row=row, j=j, c=cidx, yv=dict(yv): e("v_fmac_f32", v(yv[row]), v(pin[j]), v(yv[c])))
                else:
                    reg = free.pop()
                    yv[row] = reg
                    op([], lambda g, reg=reg, j=j, c=yv[cidx]: e("v_mul_f32", v(reg), v(pin[j]), v(c)))

            def fe(g, yc=yv[cidx], new=new):
                e("v_mul_f32", v(T_LV), "-" + v(yc), v(g[0]))              # -L = -(y_c / D_c)
                sc.lds_write(p.LW_L + p.lpos[new], T_LV)
                e("v_fmac_f32", v(T_DK), v(yc), v(T_LV))                   # d_k -= y_c L
                if new in pin:
                    e("v_mov_b32", v(pin[new]), v(T_LV))
            op([("A", cidx)], fe)
        for reg in yv.values():
            free.append(reg)

        def fin(g, k=k):
            e("v_rcp_f32", v(T_LV), v(T_DK))
            e("s_nop", 0)
            e("v_fma_f32", v(T_A), "-" + v(T_DK), v(T_LV), 1.0)
            e("v_fma_f32", v(T_LV), v(T_LV), v(T_A), v(T_LV))
            e("v_accvgpr_write_b32", "a%d" % k, v(T_LV))
            e("v_min_f32", v(v_fmax), v(v_fmax), "|" + v(T_DK) + "|")
        op([], fin)
    if not met[0]:
        meet()
    sc.run(ops)
    e("s_waitcnt", "lgkmcnt(0)")


# ---------------------------------------------------------------------------
# Glue between the Ruiz block and the loop: rho classification, scaled bounds, the loop's stream
# ---------------------------------------------------------------------------
# After the Ruiz block (E in LDS words RuizPlan.LW_EV.., q in LW_Q..) this block does what the C++ side of
# codegen_qp.emit_structure does between scaling and factorisation -- update_rho_vec on the bounds scaled by the previous E
# (auxil.c:103-145), l E and u E (scaling.c:scale_data), the loop's read-only stream, z of the equality rows for the
# residual block -- from batched row loads instead of one exposed load per word, and decides whether the wave may take the
# all-assembly route: LDS word GLUE_FLAG = 1 iff every row of eq_rows is an equality (rho == rho_eq, l E == u E) whose
# warm-start z equals its bound.
GLUE_FLAG = 637
LOOSE_FLAG = 636      # 1 iff, in addition, every other row is a loose row (rho = RHO_MIN): the wave takes the loose loop variant
S_LR, S_UR, S_ER, S_ZR = 4, 8, 24, 26         # pointer pairs: l, u, Eprev, z rows (s[6:7] = the wave's stream block)
GV_RHO0, GV_RINV0, GV_RHOEQ, GV_RINVEQ = 5, 6, 7, 8      # inputs (VGPRs, wave-uniform floats)
GLUE_PTRS = (S_SP, 36, 38, 40)                           # SGPR pairs of the block pointers for the stream stores
QP_RHO_MIN, QP_RHO_TOL, QP_INF_SCALED = 1e-6, 1e-4, 1e20 * 1e-4


def _f32_strict_bounds():
    """float thresholds that make the fp32 comparisons equal to umpc_bqp_common.h's comparisons in double:
    (double)x > 1e16 <=> x > F with F the largest float <= 1e16; (double)d < 1e-4 <=> d < TOL with TOL the smallest
    float >= 1e-4"""
    f32 = np.float32
    F = f32(QP_INF_SCALED)
    if float(F) > QP_INF_SCALED:
        F = np.nextafter(F, f32(0))
    TOL = f32(QP_RHO_TOL)
    if float(TOL) < QP_RHO_TOL:
        TOL = np.nextafter(TOL, f32(1))
    return float(F), float(TOL)


S_GWAVE = 28                       # glue_group_program: s28 = the wave's index in its workgroup
GLUE_TWO_PASS = os.environ.get("UMPC_QP_GLUE_TWO_PASS", "1") == "1"   # (A/B switch: glue_program, shared blocks)
GLUE_BALANCE = os.environ.get("UMPC_QP_GLUE_BALANCE", "1") == "1"     # (A/B switch: equality and other rows dealt out separately)


def glue_program(s, eq_rows, p, res, rp, split=None):
    """p: Plan (stream positions), res: ResPlan (z of the equality rows), rp: RuizPlan (where E and q are in LDS).
    split = (wave, nw): the program of one of nw wavefronts that share the block (each a quarter of the rows and of q; the two
    flags are combined with LDS float-min atomics between two barriers). Returns the instructions (and, with split, the set of
    stream items this wave writes)."""
    n, m = s.n, s.m
    wave, nw = split if split is not None else (0, 1)
    my_rows = [i for i in range(m) if i * nw // m == wave]
    if split is not None and GLUE_TWO_PASS and GLUE_BALANCE:
        # a quarter of the equality rows AND a quarter of the other rows each: when the robots take the loose loop nothing of the
        # other rows is stored, and the equality rows -- all of the work that is left -- must not sit on two of the four waves
        eqr, ineqr = [i for i in range(m) if i in set(int(i_) for i_ in eq_rows)], [i for i in range(m) if i not in set(int(i_) for i_ in eq_rows)]
        my_rows = sorted([i for q_, i in enumerate(eqr) if q_ * nw // len(eqr) == wave] +
                         [i for q_, i in enumerate(ineqr) if q_ * nw // len(ineqr) == wave]) if eqr and ineqr else my_rows
    my_cols = [j for j in range(n) if j * nw // n == wave]
    eq = set(int(i) for i in eq_rows)
    pos = {}
    for q, it in enumerate(p.stream + p.extra):
        pos.setdefault(it, []).append(q)
    e = Emit()
    v = lambda r: "v%d" % r
    V_RMIN, V_RIMIN, V_F, V_NF, V_TOL, V_FLAG, V_LFLAG = 9, 10, 11, 12, 13, 14, 15
    R = 44
    V_L, V_U, V_E, V_Z = 16, 16 + R, 16 + 2 * R, 16 + 3 * R

    class P_:
        pass
    pl = P_()
    pl.V_RING = 16 + 4 * R
    pl.V_LAND = pl.V_AT = pl.V_RING + 4 * NRING
    pl.n_land = 0
    V_T = pl.V_AT + N_AT
    NSET = (V_END - V_T) // 8
    assert NSET >= 2
    F, TOL = _f32_strict_bounds()
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    if RUIZ_STAMPS and wave == 0:
        e("s_memrealtime", "s[60:61]")
        e("s_waitcnt", "lgkmcnt(0)")
    e("v_add_u32", "v%d" % V_B1, 0x10000, "v1")
    e("v_add_u32", "v%d" % V_B2, 0x20000, "v1")
    for reg, val in ((V_RMIN, np.float32(QP_RHO_MIN)), (V_RIMIN, np.float32(1.0 / QP_RHO_MIN)), (V_F, F), (V_NF, -F), (V_TOL, TOL)):
        e("v_mov_b32", v(reg), f32bits(float(val)))
    e("v_mov_b32", v(V_FLAG), 1.0)
    e("v_mov_b32", v(V_LFLAG), 1.0)
    if split is not None:
        if wave == 0:             # the flag words start at 1; every wave then folds its own verdict in (float min, atomic)
            for word in (GLUE_FLAG, LOOSE_FLAG):
                base, off = lds_addr(word)
                e("ds_write_b32", base, v(V_FLAG), off)
            e("s_waitcnt", "lgkmcnt(0)")
        e("s_barrier")
    # a row's items go to three regions of the stream (1/rho list, per-row items, the residual stream): four block
    # pointers are kept, least recently used replaced
    ptrs = [[sreg, None, 0] for sreg in GLUE_PTRS]          # [SGPR pair, block, last use]
    tick = [0]

    def put(item, reg):
        blk = item // BLOCK
        tick[0] += 1
        hit = [q for q in ptrs if q[1] == blk]
        if hit:
            q = hit[0]
        else:
            q = min(ptrs, key=lambda z: z[2])
            q[1] = blk
            e("s_add_u32", "s%d" % q[0], "s%d" % S_S, blk * BLOCK * 256)
            e("s_addc_u32", "s%d" % (q[0] + 1), "s%d" % (S_S + 1), 0)
        q[2] = tick[0]
        put_items.append(item)
        e("global_store_dword", "v%d" % V_LANE, v(reg), "s[%d:%d]" % (q[0], q[0] + 1), (item % BLOCK) * 256)

    def check(a_, b_):
        e("v_cmp_eq_f32", "vcc", v(a_), v(b_))
        e("v_cndmask_b32", v(V_FLAG), 0, v(V_FLAG), "vcc")
    nrow = 0
    put_items = []
    # A shared block decides FIRST whether the 64 robots take the loose loop (every inequality row of every robot a loose row:
    # the reference's problem) and then leaves the per-row items of those rows -- 1/rho, l, u, rho: what only the general loop
    # and the C++ routes read -- unwritten: 5 of the 7 stores of such a row, 1.7 kB per robot-tick (GLUE_TWO_PASS; the C++
    # residual phase forms l E, u E of those rows itself: codegen_qp.emit_fast_route_reload). One chunk of rows per wave.
    two_pass = split is not None and GLUE_TWO_PASS and len(my_rows) <= R
    for c00 in range(0, len(my_rows), R):
        rows = my_rows[c00:c00 + R]
        c0 = rows[0]
        slot = {i: q_ for q_, i in enumerate(rows)}          # row -> its place in the chunk's register arrays
        for base_s, v0_, sel in ((S_LR, V_L, rows), (S_UR, V_U, rows), (S_ER, V_E, rows), (S_ZR, V_Z, [i for i in rows if i in eq])):
            last = None
            for i in sel:
                if last is None or i != last + 1:
                    _row_ptr(e, S_P, i, base_s)
                else:
                    _adv(e, S_P)
                last = i
                e("global_load_dword", v(v0_ + slot[i]), "v0", "s[%d:%d]" % (S_P, S_P + 1), 0)
        e("s_waitcnt", "vmcnt(0)")

        def classify(i, T):
            """rho, 1/rho of row i from its bounds scaled by the previous E (auxil.c:103-145) -> T(4), T(5); T(0), T(1) = l E', u E'"""
            l_, u_, e_ = V_L + slot[i], V_U + slot[i], V_E + slot[i]
            le, ue, d, t, rho, rinv = (T(q) for q in range(6))
            e("v_mul_f32", v(le), v(l_), v(e_))
            e("v_mul_f32", v(ue), v(u_), v(e_))
            e("v_sub_f32", v(d), v(ue), v(le))
            e("v_mov_b32", v(rho), v(GV_RHO0))
            e("v_mov_b32", v(rinv), v(GV_RINV0))
            e("v_cmp_gt_f32", "vcc", v(V_TOL), v(d))                        # u - l < RHO_TOL: an equality row
            e("v_cndmask_b32", v(rho), v(rho), v(GV_RHOEQ), "vcc")
            e("v_cndmask_b32", v(rinv), v(rinv), v(GV_RINVEQ), "vcc")
            e("v_cmp_lt_f32", "vcc", v(V_F), v(ue))                         # both bounds infinite: a loose row
            e("v_cndmask_b32", v(t), 0, v(le), "vcc")
            e("v_cmp_gt_f32", "vcc", v(V_NF), v(t))
            e("v_cndmask_b32", v(rho), v(rho), v(V_RMIN), "vcc")
            e("v_cndmask_b32", v(rinv), v(rinv), v(V_RIMIN), "vcc")
        if two_pass:
            for i in rows:
                if i not in eq:
                    classify(i, lambda q: V_T + q)
                    e("v_cmp_eq_f32", "vcc", v(V_T + 4), v(V_RMIN))
                    e("v_cndmask_b32", v(V_LFLAG), 0, v(V_LFLAG), "vcc")
            base, off = lds_addr(LOOSE_FLAG)
            e("ds_min_f32", base, v(V_LFLAG), off)
            e("s_waitcnt", "lgkmcnt(0)")
            e("s_barrier")
            e("ds_read_b32", v(V_T), base, off)
            e("s_waitcnt", "lgkmcnt(0)")
            e("v_cmp_neq_f32", "vcc", 1.0, v(V_T))                          # lanes whose robot is NOT all loose
            e("s_cbranch_vccz", "31f")

        def rows_pass(lean, nrow0):
            sc = Sched(e, pl, 0)
            ops = []
            for k_, i in enumerate(rows):
                def f(g, i=i, k=nrow0 + k_):
                    T = lambda q: V_T + 8 * (k % NSET) + q
                    l_, u_, z_ = V_L + slot[i], V_U + slot[i], V_Z + slot[i]
                    rho, rinv, ls, us = (T(q) for q in range(4, 8))
                    classify(i, T)
                    e("v_mul_f32", v(ls), v(l_), v(g[0]))
                    e("v_mul_f32", v(us), v(u_), v(g[0]))
                    if i in eq:
                        check(rho, GV_RHOEQ)
                        check(ls, us)
                        check(z_, ls)
                        for q in pos[("l", i)]:
                            put(q, ls)
                        put(res.it_ls[i], ls)
                    else:
                        e("v_cmp_eq_f32", "vcc", v(rho), v(V_RMIN))                 # a loose row? (the loose loop variant)
                        e("v_cndmask_b32", v(V_LFLAG), 0, v(V_LFLAG), "vcc")
                        if not lean:
                            for what, reg in (("rinv", rinv), ("l", ls), ("u", us), ("rho", rho)):
                                for q in pos.get((what, i), []):
                                    put(q, reg)
                if lean and i not in eq:
                    continue              # (nothing of such a row is stored, and its verdict is in already)
                ops.append(dict(srcs=[("L", rp.LW_EV + i)], emit=f))
            sc.run(ops)
        if two_pass:
            mark = len(put_items)
            rows_pass(False, nrow)
            e("s_branch", "32f")
            e("label", "31")
            n_full = len(put_items)
            rows_pass(True, nrow)
            del put_items[n_full:]        # (the items of a wave are what its full pass writes)
            for q_ in ptrs:               # (the two passes leave different block pointers behind)
                q_[1] = None
            e("label", "32")
        else:
            rows_pass(False, nrow)
        nrow += len(rows)
        e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")        # the landing registers are loaded again by the next chunk
    sc = Sched(e, pl, 0)
    ops = []
    for j in my_cols:
        def fq(g, j=j):
            for q in pos[("q", j)]:
                put(q, g[0])
        ops.append(dict(srcs=[("L", rp.LW_Q + j)], emit=fq))
    sc.run(ops)
    for word, reg in ((GLUE_FLAG, V_FLAG), (LOOSE_FLAG, V_LFLAG)):
        if word == LOOSE_FLAG and two_pass:
            continue                  # (folded in before the rows were stored; the waves are reading the word by now)
        base, off = lds_addr(word)
        e("ds_min_f32" if split is not None else "ds_write_b32", base, v(reg), off)
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
    if split is not None:
        e("s_barrier")
    if RUIZ_STAMPS and wave == 0:
        e("s_memrealtime", "s[62:63]")
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_sub_u32", "s60", "s62", "s60")
        e("v_cvt_f32_u32", v(V_T), "s60")
        put(STAMP_ITEM0 + 5, V_T)
        e("s_waitcnt", "vmcnt(0)")
    written = sorted(q for lst in pos.values() for q in lst)
    assert written == list(range(p.n_stream + len(p.extra)))
    if split is not None:
        return e.ins, put_items
    assert sorted(q for q in put_items if q < res.rs0) == written
    return e.ins


def glue_group_program(s, eq_rows, p, res, rp, nw=4):
    """The glue block for a workgroup of nw wavefronts that share 64 robots: s28 = the wave's index, the other inputs as
    glue_program. Every wave meets the same two barriers."""
    e = Emit()
    items = []
    for w in range(nw):
        if w < nw - 1:
            e("s_cmp_lg_u32", "s%d" % S_GWAVE, w)
            e("s_cbranch_scc1", "8f")
        ins, put_w = glue_program(s, eq_rows, p, res, rp, (w, nw))
        items += put_w
        e.ins.extend(ins)
        if w < nw - 1:
            e("s_branch", "9f")
            e("label", "8")
    e("label", "9")
    assert sorted(q for q in items if q < res.rs0) == list(range(p.n_stream + len(p.extra))), "every stream item exactly once"
    return e.ins

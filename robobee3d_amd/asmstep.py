"""The ALL-ASSEMBLY fp32 step kernel (robobee3d_amd/csrc/umpc_step_asm.h): K closed-loop steps of one wavefront =
64 robots as ONE generated gfx950 instruction stream -- phase A (load, assemble, classify, 10 Ruiz passes, D/E
recovery, KKT fill + LDL'), the ADMM loop of asmgen.py (unchanged placement: W, x, y, z in VGPRs, L in LDS / z
registers / AGPRs, 1/D, q, bounds in AGPRs), phase C (residuals, termination tests, certificates, extraction,
record/output stores) and the RK4 plant substeps.

Why (round-1 VERDICT item 4 and the round-2 microbenchmarks, tools/microbench2.hip): around the assembly loop the
compiler-generated phases ran at 8.4 cycles per instruction (spills into AGPRs cost 8.6 cycles per access for a lone
wave, 1.4 kB/lane of scratch, ~400 hand-off rows per robot through L2). Here nothing crosses a compiler boundary:
the factor is written straight into its loop homes (LDS / AGPR), x, y, z and delta_x / delta_y of the last iteration
are consumed by phase C where the loop left them, and the only extra HBM rows are D, E, c (85 words per robot-step)
parked across the loop.

Reference mapping (template/uprightmpc2/...): uprightmpc2.c:209-269 (assembly, extraction), osqp.c:784-833 +
auxil.c:103-145 (classification), scaling.c:44-156 (Ruiz), kkt.c:184-222 + qdldl.c:86-247 (factor),
osqp.c:354-370 + auxil.c:164-228 (iterations, asmgen.py), auxil.c:243-362, 684-789 + osqp.c:524-573 (residuals,
status), auxil.c:517-565 (store_solution), template/genqp.py:24-30 (vector field; RK4 is build-defined).

Arithmetic vs the C++ statement of the same step (csrc/umpc_step.h, kept for fp64 and as the cross-check of this
stream): phase C works in UNSCALED variables (xu = D x, yu = E y / c; the residual norms are
algebraically the reference's Einv / Dinv / cinv-weighted norms), sums are associated for packed arithmetic, 1/x is
v_rcp_f32 + one Newton step. Everything is checked on CPU by interpreting the emitted instructions
(asmgen.simulate) against the oracle before it reaches a GPU (tests/test_asm_step.py).

Scope of this path: fp32, both plant steps; optional per-robot Ib / thrust gain / actualT0 / stats / status / info and --
round 3 -- the SURVEY 8(f) workloads as options of the same stream: a task table (flight_tasks.py generators: 8 floats per
step, scalar loads), per-robot objective weights (gain sweeps; parked in AGPRs across Ruiz and the loop), the fused WL step
(funapprox.c:118-165) and a completion word for hosts that poll (the B = 1 drop-in). umpc_mi355x.hip dispatches.
"""
import os
import struct

from . import asmgen, symbolic
from .asmgen import (A_D, A_L, A_LO, A_M, A_Q, NLDS, NVZ, S_ALPHA, S_CNT, S_ITERS, S_OMA, S_RHO, S_RINV, S_SIGMA,
                     V_W, V_WZ, V_X, V_Y, V_Z, Emit, f32bits, pk, _sb, _vp)

HERE = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------------------------------------------------
# Parameter block (kernarg-resident struct umpcasm::StepParams, 4-byte words; the kernel s_loads it once)
# ----------------------------------------------------------------------------------------------------------
PTRS = ["state", "ctrl", "ref", "ws", "out", "stats", "status", "info", "Ib", "gain", "aT0",
        # SURVEY 8(f) options: per-step task table (8 floats per step, see phase_a), per-robot weights [8][B], the
        # wrench-linearisation parameters (struct umpc::WLDev, 150 floats) with its state rows u4 [4][B] and w0 [6][B]
        "taskf", "weights", "wl", "wlu", "wlw",
        # B = 1 drop-in: a host-visible word the stream writes `seq` to, system scope, after its last store (null: nothing)
        "done"]
NPTR_RES = 11             # the first 11 pointers live in s40..s61 for the whole kernel
INTS = ["stride", "K", "maxIter", "nsub", "plant", "seq"]
FLOATS = ["dt", "dtg", "Tmax", "wpr", "wpf", "ws_", "wvr", "wvf", "wds", "wthrust", "wmom",
          "iwpr", "iwpf", "iws", "iwvr", "iwvf", "iwds", "iwthrust", "iwmom",
          "Ib0", "Ib1", "Ib2", "Ibi0", "Ibi1", "Ibi2", "h", "hh", "h6", "taulim", "gpl", "idt", "mbg"]
WNAMES = ["wpr", "wpf", "ws_", "wvr", "wvf", "wds", "wthrust", "wmom"]
WROW = {"ws_": 0, "wds": 1, "wpr": 2, "wpf": 3, "wvr": 4, "wvf": 5, "wthrust": 6, "wmom": 7}     # rows of the weights table
A_W, A_IW = 230, 238      # AGPR parking of the step's weights and their reciprocals (the loop owns a0..a229)
# SGPR homes. s[4:5] = parameter block (input; kept in s[2:3]). asmgen's loop owns s14 (S_CNT) and s20..s29 (alpha,
# 1-alpha, sigma, 1/rho_eq, rho_eq as even pairs); s11 = S_ITERS.
S_PARAM = 4
S_PBLK = 2                                                           # s[2:3]: the parameter block, for the late s_loads
S_PTR = {n: 40 + 2 * k for k, n in enumerate(PTRS[:NPTR_RES])}       # s40..s61
S_PTR["taskf"], S_PTR["weights"] = 0, 96                             # s[0:1], s[96:97]
S_PTR["wl"], S_PTR["wlu"], S_PTR["wlw"] = 30, 32, 34                 # loaded where the WL step starts (masks are dead there)
S_INT = {"stride": 10, "K": 13, "maxIter": S_ITERS, "nsub": 5, "plant": 39}       # (seq is read at the very end)
S_F = {n: 64 + k for k, n in enumerate(FLOATS)}                      # s64..s95
assert max(S_F.values()) <= 95
S_STEP, S_SUB, S_RUIZ = 12, 15, 15          # loop counters: closed-loop step; plant substep / Ruiz pass (never nested)
S_MBAD, S_MP0 = 62, 14                      # phase C masks: s[62:63]; s[14:15] (ADMM / Ruiz counters are dead there)
S_M0, S_M1, S_M2, S_M3 = 30, 32, 34, 36     # lane masks (pairs)
S_C = {"minscal": 16, "maxscal": 17, "c45": 18, "eps10": 19, "eps": 38, "rho": 6, "rinv": 7, "rmin": 8,
       "rmininv": 9, "infty_ms": 100, "rhotol": 101}   # scalar constants set by the prologue
S_TMP = 4     # s4 is free once the prologue has read the parameter block (s5 = nsub)
OFF = {}
_o = 0
for _n in PTRS:
    OFF[_n] = _o
    _o += 8
for _n in INTS + FLOATS:
    OFF[_n] = _o
    _o += 4
PARAM_BYTES = _o

VFIRST, VEND = 2, 256
XV_N, XV_B = 10, 246      # L entries kept in VGPRs beyond the loop's fixed layout (asmgen.XV_COUNT / XV_BASE)
# generator switches for A/B timing of variants on one box (tools/build_variant.py); defaults = the shipped kernel
# v246..v255 hold either ten L words (UMPC_ASM_RING=4) or two more slots of the LDS read ring plus two L words (6, the
# default: with six slots the reads run 20 operations ahead and waits merge in pairs, 1.2 % faster, profiles/README.md)
OPT_RING = int(os.environ.get("UMPC_ASM_RING", "6"))
assert OPT_RING in (4, 6)
OPT_XV = os.environ.get("UMPC_ASM_XV", "1") == "1"
if OPT_RING == 6:
    XV_N, XV_B = 2, 254
# LIMIT_FAST: wave-wide min / max test that skips the exact limit_scaling sequence (-1 500 instructions per step).
# Measured SLOWER on the MI355X (same box, K = 500: 0.1261 vs 0.1239 ms per step): five VALU -> SGPR -> s_cbranch_vccz
# round trips per Ruiz pass cost more than the 150 instructions they skip. Off.
OPT_LIMIT_FAST = os.environ.get("UMPC_ASM_LIMIT_FAST", "0") == "1"
# ZSKIP: the right-hand side of entries whose q / l is structurally zero is formed without the AGPR read (asmgen.body)
OPT_ZSKIP = os.environ.get("UMPC_ASM_ZSKIP", "1") == "1"
# NT: the once-per-step STREAMING rows (state, ctrl, ref, out, stats: read in phase A, rewritten in phase C) carry the
# non-temporal hint, so that the 85 workspace rows D, E, c a wave parks across its 50 iterations (21.7 KB per wave, 2.8 MB per
# XCD against a 4-MB L2) are not evicted by them and need not round-trip through HBM. "1" (default, lane form): loads and
# stores, "ld" / "st": one side, "0": off. Measured (tools/ab_nt.sh, profiles/r05_nt_ab.txt, one box, K = 500): read bytes per
# robot-step 969 -> 616 (algorithmic 604), written 977 -> 980, 1.61x -> 1.32x algorithmic; 0.1208 -> 0.1216 ms per step. Either
# side alone does nothing ("ld": 1071 read; "st": 964).
OPT_NT = os.environ.get("UMPC_ASM_NT", "1")
NT_PTRS = ("state", "ctrl", "ref", "out", "stats")
# The ten AGPRs nobody owns (the loop: a0..a229, the weights: a230..a245) take c and the first nine D words across the loop
# instead of workspace rows (lane form only: the quad entry composes words in AGPRs): 10 stores + 10 loads fewer per step
A_SPARE, N_SPARE_D = 246, 9
# Cache-policy bits on the workspace stores / loads of the parked D, E, c rows (lane form). Measured (tools/ab_ws.sh,
# profiles/r05_nt_ab.txt, one box, on top of the streaming hint): plain 565 B read + 936 B written per robot-step = 1.24x
# algorithmic; "nt" on both sides 502 + 779 = 1.06x, same ms per step (0.1206 / 0.1208); sc0 no change; sc1 / sc0 sc1
# (write-through scopes) 805 read = 1.44x. Shipped: nt / nt.
OPT_WS_ST = os.environ.get("UMPC_ASM_WS_ST", "nt")
OPT_WS_LD = os.environ.get("UMPC_ASM_WS_LD", "nt")


class Pool:
    """VGPR bookkeeping for the straight-line phases: explicit get / free, aligned pairs and quads, peak tracking.
    free() also emits a `kill` pseudo-instruction: asmgen.simulate poisons the register, so a use-after-free shows up
    as a NaN in the CPU tests."""

    def __init__(self, e):
        self.e = e
        self.free_ = set(range(VFIRST, VEND))
        self.peak = 0
        self.fixed = set()

    def _take(self, regs):
        for r in regs:
            assert r in self.free_, "v%d is not free" % r
            self.free_.discard(r)
        self.peak = max(self.peak, VEND - VFIRST - len(self.free_))

    def reserve(self, lo, n):
        self._take(range(lo, lo + n))
        return lo

    def get(self):
        cand = sorted(self.free_)
        assert cand, "out of VGPRs"
        for r in cand:          # prefer a register whose pair partner is taken (keeps aligned pairs available)
            if (r ^ 1) not in self.free_:
                self._take([r])
                return r
        self._take([cand[0]])
        return cand[0]

    def get2(self):
        for r in sorted(self.free_):
            if r % 2 == 0 and r + 1 in self.free_:
                self._take([r, r + 1])
                return r
        raise AssertionError("out of aligned VGPR pairs")

    def get4(self):
        for r in sorted(self.free_):
            if r % 4 == 0 and all(r + k in self.free_ for k in range(4)):
                self._take([r, r + 1, r + 2, r + 3])
                return r
        raise AssertionError("out of aligned VGPR quads")

    def getn(self, n):
        """n consecutive registers, even base"""
        for r in sorted(self.free_):
            if r % 2 == 0 and all(r + k in self.free_ for k in range(n)):
                self._take(range(r, r + n))
                return r
        raise AssertionError("out of consecutive VGPRs")

    def getn_high(self, n):
        """n consecutive registers ending at the top of the file (even base)"""
        r = VEND - n - (VEND - n) % 2
        assert all(r + k in self.free_ for k in range(n)), "top of the register file is not free"
        self._take(range(r, r + n))
        return r

    def free(self, *regs, kill=True):
        for r in regs:
            assert r not in self.free_ and VFIRST <= r < VEND, "double free v%d" % r
            self.free_.add(r)
            if kill:
                self.e("kill", "v%d" % r)

    def free_range(self, lo, n, kill=True):
        self.free(*range(lo, lo + n), kill=kill)


def v(n):
    return "v%d" % n


def SF_(n):
    return "s%d" % S_F[n]


def sg(n):
    return "s%d" % n


def sp(n):
    return "s[%d:%d]" % (n, n + 1)


def vp2(n):
    return "v[%d:%d]" % (n, n + 1)


# packed-operand descriptors (asmgen.pk): (register-pair string, select for the low result, select for the high result)
def P2(n):
    return _vp(n)


def PB(n):          # one VGPR broadcast
    lo = n - n % 2
    return ("v[%d:%d]" % (lo, lo + 1), n % 2, n % 2)


def PSEL(lo_reg, hi_reg):
    """operand whose low result comes from lo_reg and high result from hi_reg (both inside one aligned pair)"""
    assert lo_reg // 2 == hi_reg // 2
    base = lo_reg - lo_reg % 2
    return ("v[%d:%d]" % (base, base + 1), lo_reg % 2, hi_reg % 2)


def PS(sreg):       # SGPR broadcast: the pair s[sreg:sreg+1] read with both selects 0 needs an even sreg
    if sreg % 2 == 0:
        return ("s[%d:%d]" % (sreg, sreg + 1), 0, 0)
    return ("s[%d:%d]" % (sreg - 1, sreg), 1, 1)


# ----------------------------------------------------------------------------------------------------------
# Structure helpers
# ----------------------------------------------------------------------------------------------------------
class Struct:
    def __init__(self, N=3, perm=None):
        s = symbolic.analyse(N, perm)
        self.s = s
        self.plan = asmgen.solve_plan(s)
        self.lpos = self.plan[2]
        self.xs, self.zs, self.xinv, self.zinv = asmgen.slot_maps(s)
        nx, nc = s.nx, s.nc
        self.neq = 2 * s.N * symbolic.NY
        self.col_of = [0] * len(s.A_i)
        self.rows = [[] for _ in range(nc)]
        for j in range(nx):
            for p in range(s.A_p[j], s.A_p[j + 1]):
                self.col_of[p] = j
                self.rows[s.A_i[p]].append(p)
        # first +-1 entry of every row in CSC order (codegen.py UMPC_GEN_E_FROM_A)
        self.unit = {}
        for p, tag in enumerate(s.A_tag):
            if tag[0] == 'c' and abs(tag[1]) == 1.0 and s.A_i[p] not in self.unit:
                self.unit[s.A_i[p]] = (p, self.col_of[p])
        assert len(self.unit) == nc
        # packable entry pairs: rows are slot partners, columns are slot partners or the same column
        used, pairs = set(), []
        for p in range(len(s.A_i)):
            if p in used:
                continue
            i0, j0 = s.A_i[p], self.col_of[p]
            for q_ in range(p + 1, len(s.A_i)):
                if q_ in used:
                    continue
                i1, j1 = s.A_i[q_], self.col_of[q_]
                if (self.zs[i0] ^ 1) == self.zs[i1] and (j0 == j1 or (self.xs[j0] ^ 1) == self.xs[j1]):
                    lo, hi = (p, q_) if self.zs[i0] % 2 == 0 else (q_, p)
                    pairs.append((lo, hi))
                    used.update((p, q_))
                    break
        self.apairs = pairs
        self.asingles = [p for p in range(len(s.A_i)) if p not in used]
        self.apos = [0] * len(s.A_i)
        k = 0
        for (lo, hi) in pairs:
            self.apos[lo], self.apos[hi] = k, k + 1
            k += 2
        for p in self.asingles:
            self.apos[p] = k
            k += 1
        self.na = k
        self.pair_of = {}
        for (lo, hi) in pairs:
            self.pair_of[lo] = self.pair_of[hi] = (lo, hi)
        # q: entries that can be non-zero (ydes on the y part, dpdes on the dp triples)
        NYv = symbolic.NY
        self.qnz = list(range(s.N * NYv)) + [s.N * NYv + k_ * NYv + i for k_ in range(s.N) for i in range(3)]
        self.qslot = {}
        for j in range(s.N * NYv):
            self.qslot[j] = self.xs[j]
        base = s.N * NYv
        for k_ in range(s.N):
            for i in range(3):
                self.qslot[s.N * NYv + k_ * NYv + i] = base + 3 * k_ + i
        self.nq = base + 3 * s.N
        # structural zeros of the right-hand side data (uprightmpc2.c:126-207): q outside the y / dp entries; the raw
        # bounds of the dynamics rows outside rows 0..5 (-y1), 18..23, 24..26 and 32 (the constant dt g)
        self.qzero = frozenset(j for j in range(nx) if j not in self.qslot)
        lnz = set(range(6)) | set(range(3 * 6, 3 * 6 + 6)) | set(range(4 * 6, 4 * 6 + 3)) | {5 * 6 + 2}
        self.lzero = frozenset(i for i in range(self.neq) if i not in lnz) if s.N == 3 else frozenset()

    def weight_of(self, j):
        """name of the objective weight of column j (umpcInit layout, uprightmpc2.c:27-36)"""
        N, NYv = self.s.N, symbolic.NY
        if j < N * NYv:
            k, i = divmod(j, NYv)
            return ("wpf" if k == N - 1 else "wpr") if i < 3 else "ws_"
        if j < 2 * N * NYv:
            k, i = divmod(j - N * NYv, NYv)
            return ("wvf" if k == N - 1 else "wvr") if i < 3 else "wds"
        return "wthrust" if (j - 2 * N * NYv) % 3 == 0 else "wmom"


# ----------------------------------------------------------------------------------------------------------
# The generator
# ----------------------------------------------------------------------------------------------------------
class StepGen:
    def __init__(self, N=3, perm=None, quad=False):
        """quad: one robot per lane QUAD (asmquad.py) -- every phase runs redundantly in the four lanes of a quad and the
        ADMM iterations from the second on split the unknowns over lanes 0..2; the stream of the latency-bound shapes
        (B = 1 drop-in, batches that cannot give every SIMD a wave)."""
        self.st = Struct(N, perm)
        self.s = self.st.s
        self.e = Emit()
        self.pool = Pool(self.e)
        self.lab = 20
        self.quad = quad
        # the quad form serves batches whose whole working set lives in the eight L2s (0.14-0.44x algorithmic HBM traffic):
        # there the state / ctrl rows are exactly what should stay cached, so the hint is a lane-form matter
        self.nt = "0" if quad else OPT_NT

    # ---- small emit helpers -----------------------------------------------------------------------------
    def label(self):
        self.lab += 1
        return str(self.lab)

    def rcp_nr(self, dst, src, t):
        """dst = 1 / src: v_rcp_f32 (1 ulp) + one Newton step (the C++ statement divides)."""
        e = self.e
        e("v_rcp_f32", v(dst), v(src))
        e("s_nop", 0)                                   # trans result -> next VALU (gfx940 forwarding hazard)
        e("v_fma_f32", v(t), "-" + v(src), v(dst), 1.0)
        e("v_fma_f32", v(dst), v(t), v(dst), v(dst))

    def rows_ptr(self, voff, first_row):
        """voff = 4*b + first_row * stride"""
        e = self.e
        if first_row == 0:
            e("v_mov_b32", v(voff), "v0")
        else:
            e("s_mul_i32", sg(S_TMP), sg(S_INT["stride"]), first_row)
            e("v_add_u32", v(voff), sg(S_TMP), "v0")

    def load_taskf(self):
        """s[S_M0 .. S_M0+7] = the 8 floats of this step's task-table entry (32 B per step; an x8 load needs a 4-aligned
        destination, the masks start at s30: three loads)"""
        e = self.e
        e("s_lshl_b32", sg(S_TMP), sg(S_STEP), 5)
        e("s_load_dwordx2", sp(S_M0), sp(S_PTR["taskf"]), sg(S_TMP), "offset:0")
        e("s_load_dwordx4", "s[%d:%d]" % (S_M0 + 2, S_M0 + 5), sp(S_PTR["taskf"]), sg(S_TMP), "offset:8")
        e("s_load_dwordx2", sp(S_M0 + 6), sp(S_PTR["taskf"]), sg(S_TMP), "offset:24")
        e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")

    def adv(self, voff):
        self.e("v_add_u32", v(voff), sg(S_INT["stride"]), v(voff))

    def load_rows(self, ptr, first_row, regs, voff=None):
        own = voff is None
        if own:
            voff = self.pool.get()
        self.rows_ptr(voff, first_row)
        for k, r in enumerate(regs):
            self.e("global_load_dword", r if isinstance(r, str) else v(r), v(voff), sp(S_PTR[ptr]),
                   *(("nt",) if self.nt in ("1", "ld") and ptr in NT_PTRS else (OPT_WS_LD,) if ptr == "ws" and OPT_WS_LD and not self.quad else ()))
            if k + 1 < len(regs):
                self.adv(voff)
        if own:
            self.pool.free(voff, kill=False)

    def store_rows(self, ptr, first_row, regs, voff=None):
        own = voff is None
        if own:
            voff = self.pool.get()
        self.rows_ptr(voff, first_row)
        for k, r in enumerate(regs):
            self.e("global_store_dword", v(voff), v(r), sp(S_PTR[ptr]),
                   *(("nt",) if self.nt in ("1", "st") and ptr in NT_PTRS else ()))
            if k + 1 < len(regs):
                self.adv(voff)
        if own:
            self.pool.free(voff, kill=False)

    # ---- prologue: parameters and constants into SGPRs ---------------------------------------------------
    def prologue(self):
        e = self.e
        P = sp(S_PARAM)
        e("s_mov_b64", sp(S_PBLK), P)
        P = sp(S_PBLK)
        e("s_load_dwordx16", "s[40:55]", P, OFF["state"])
        e("s_load_dwordx4", "s[56:59]", P, OFF["state"] + 64)
        e("s_load_dwordx2", "s[60:61]", P, OFF["state"] + 80)
        e("s_load_dwordx2", sp(S_PTR["taskf"]), P, OFF["taskf"])
        e("s_load_dwordx2", sp(S_PTR["weights"]), P, OFF["weights"])
        e("s_load_dwordx16", "s[64:79]", P, OFF[FLOATS[0]])
        e("s_load_dwordx16", "s[80:95]", P, OFF[FLOATS[0]] + 64)
        for n in S_INT:
            e("s_load_dword", sg(S_INT[n]), P, OFF[n])
        assert len(FLOATS) == 32
        import numpy as np
        f32 = np.float32
        for reg, val in ((S_ALPHA, 1.6), (S_OMA, float(f32(1.0) - f32(1.6))), (S_SIGMA, 1e-6), (S_RINV, 0.01), (S_RHO, 100.0)):
            e("s_mov_b32", sg(reg), f32bits(val))
            e("s_mov_b32", sg(reg + 1), f32bits(val))
        for name, val in (("minscal", 1e-4), ("maxscal", 1e4), ("c45", float(f32(1.0) / f32(self.s.nx))),
                          ("eps", 1e-4), ("eps10", float(f32(10) * f32(1e-4))), ("rho", 0.1), ("rinv", float(f32(1.0) / f32(0.1))),
                          ("rmin", 1e-6), ("rmininv", float(f32(1.0) / f32(1e-6))),
                          ("infty_ms", float(f32(1e30) * f32(1e-4))), ("rhotol", 1e-4)):
            e("s_mov_b32", sg(S_C[name]), f32bits(val))
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_mov_b32", sg(S_STEP), 0)

    # ---- phase A ---------------------------------------------------------------------------------------
    def phase_a(self):
        e, pool, st, s = self.e, self.pool, self.st, self.s
        N, NYv = s.N, symbolic.NY
        nx, nc, neq = s.nx, s.nc, st.neq
        SF = lambda n: sg(S_F[n])
        # A1: loads
        ST = [pool.get() for _ in range(18)]       # p0[3] R0[9] (column-major) dq0[6]
        RF = [pool.get() for _ in range(9)]        # pdes dpdes sdes
        T0, EP = pool.get(), [pool.get() for _ in range(N)]
        voff = pool.get()
        self.load_rows("state", 0, ST, voff)
        self.load_rows("ref", 0, RF, voff)
        self.load_rows("ctrl", nx + 2 * nc, [T0] + EP, voff)
        # per-robot objective weights (gain sweeps, uprightmpc2.py:272-303) or the batch constants; their reciprocals
        # (D is recovered as sqrt(P_scaled / (P_raw c))) -- both parked in AGPRs across Ruiz and the loop
        WV = {n: pool.get() for n in WNAMES}
        IWV = {n: pool.get() for n in WNAMES}
        self.WV = WV
        lab_w, lab_w2 = self.label(), self.label()
        e("s_cmp_eq_u64", sp(S_PTR["weights"]), 0)
        e("s_cbranch_scc1", lab_w + "f")
        self.rows_ptr(voff, 0)
        by_row = sorted(WNAMES, key=lambda n: WROW[n])
        for k, n in enumerate(by_row):
            e("global_load_dword", v(WV[n]), v(voff), sp(S_PTR["weights"]))
            if k + 1 < len(by_row):
                self.adv(voff)
        e("s_waitcnt", "vmcnt(0)")
        t = pool.get()
        for n in WNAMES:
            self.rcp_nr(IWV[n], WV[n], t)
        pool.free(t)
        e("s_branch", lab_w2 + "f")
        e("label", lab_w)
        for n in WNAMES:
            e("v_mov_b32", v(WV[n]), SF(n))
            e("v_mov_b32", v(IWV[n]), SF("i" + n.rstrip("_")))
        e("label", lab_w2)
        for k, n in enumerate(WNAMES):
            e("v_accvgpr_write_b32", "a%d" % (A_W + k), v(WV[n]))
            e("v_accvgpr_write_b32", "a%d" % (A_IW + k), v(IWV[n]))
        pool.free(*IWV.values(), *WV.values())
        # task generator (template/flight_tasks.py:6-49): the time-dependent part of (pdes, dpdes, sdes) is the same for
        # every robot, so the host side evaluates it once per step into a table of 8 floats per step -- (dp[3], dpdes[3],
        # sdes_x, sdes_z); sdes_y is 0 in every task -- read here with ONE scalar load; rows 0..2 of ref are initialPos
        lab_tk = self.label()
        e("s_cmp_eq_u64", sp(S_PTR["taskf"]), 0)
        e("s_cbranch_scc1", lab_tk + "f")
        self.load_taskf()
        for i in range(3):
            e("v_add_f32", v(RF[i]), sg(S_M0 + i), v(RF[i]))
            e("v_mov_b32", v(RF[3 + i]), sg(S_M0 + 3 + i))
        e("v_mov_b32", v(RF[6]), sg(S_M0 + 6))
        e("v_mov_b32", v(RF[7]), 0)
        e("v_mov_b32", v(RF[8]), sg(S_M0 + 7))
        e("label", lab_tk)
        ibi0, ibi1 = pool.get(), pool.get()
        lab_ib, lab_ib2 = self.label(), self.label()
        e("s_cmp_eq_u64", sp(S_PTR["Ib"]), 0)
        e("s_cbranch_scc1", lab_ib + "f")
        t = pool.get()
        self.load_rows("Ib", 0, [ibi0, ibi1], voff)          # raw Ib, inverted below
        e("s_waitcnt", "vmcnt(0)")
        tt = pool.get()
        for r in (ibi0, ibi1):
            e("v_mov_b32", v(tt), v(r))
            self.rcp_nr(r, tt, t)
        pool.free(t, tt)
        e("s_branch", lab_ib2 + "f")
        e("label", lab_ib)
        e("v_mov_b32", v(ibi0), SF("Ibi0"))
        e("v_mov_b32", v(ibi1), SF("Ibi1"))
        e("label", lab_ib2)
        e("s_waitcnt", "vmcnt(0)")
        # actualT0 (uprightmpc2.c:215-216): first step of the launch only
        lab_t = self.label()
        e("s_cmp_eq_u64", sp(S_PTR["aT0"]), 0)
        e("s_cbranch_scc1", lab_t + "f")
        e("s_cmp_lg_u32", sg(S_STEP), 0)
        e("s_cbranch_scc1", lab_t + "f")
        t = pool.get()
        e("global_load_dword", v(t), "v0", sp(S_PTR["aT0"]))
        e("s_waitcnt", "vmcnt(0)")
        e("v_cmp_le_f32", "vcc", 0, v(t))
        e("v_cndmask_b32", v(T0), v(T0), v(t), "vcc")
        pool.free(t)
        e("label", lab_t)
        self.store_rows("ctrl", nx + 2 * nc, [T0], voff)      # the T0 this step assembles with (phase C reloads it)
        pool.free(voff, kill=False)
        p0, R0, dq0 = ST[0:3], ST[3:12], ST[12:18]
        # A2: assembly (uprightmpc2.c:209-245, 126-207), operation order of csrc/umpc_step.h assemble()
        g_ = pool.get
        ds0 = [g_() for _ in range(3)]
        Btau = [g_() for _ in range(6)]
        t = g_()
        for r in range(3):
            # ds0 = -(R0[r] * (-wy) + R0[r+3] * wx)
            e("v_mul_f32", v(t), "-" + v(R0[r]), v(dq0[4]))
            e("v_mul_f32", v(ds0[r]), v(R0[r + 3]), v(dq0[3]))
            e("v_add_f32", v(ds0[r]), v(t), v(ds0[r]))
            e("v_mul_f32", v(ds0[r]), -1.0, v(ds0[r]))
            e("v_mul_f32", v(Btau[r]), "-" + v(R0[r + 3]), v(ibi0))       # -(R0[r+3] * Ibi0)
            e("v_mul_f32", v(Btau[3 + r]), v(R0[r]), v(ibi1))             # -(R0[r] * (-Ibi1))
        pool.free(ibi0, ibi1)
        s0 = R0[6:9]
        y0 = p0 + s0
        dy0 = dq0[0:3] + ds0
        # 17 distinct non-zero raw bounds, kept across the Ruiz loop in LDS quads 0..4:
        #   [-y1 (6) | l18..l23 (6) | l24..l26 (3) | -T0 | Tmax - T0 | pad pad pad]
        LS = pool.getn(20)
        y1 = [g_() for _ in range(6)]
        for i in range(6):
            e("v_mul_f32", v(t), SF("dt"), v(dy0[i]))
            e("v_add_f32", v(y1[i]), v(y0[i]), v(t))
            e("v_mul_f32", v(LS + i), -1.0, v(y1[i]))
        for i in range(6):
            if i < 3:
                e("v_mul_f32", v(t), v(T0), v(y0[i + 3]))
                e("v_mul_f32", v(t), SF("dt"), v(t))
                e("v_sub_f32", v(LS + 6 + i), "-" + v(dy0[i]), v(t))
                if i == 2:
                    e("v_add_f32", v(LS + 6 + i), SF("dtg"), v(LS + 6 + i))     # - dt * c0[2],  c0[2] = -g
            else:
                e("v_mul_f32", v(LS + 6 + i), -1.0, v(dy0[i]))
        for i in range(3):
            e("v_mul_f32", v(t), v(T0), v(y1[i + 3]))
            e("v_mul_f32", v(t), SF("dt"), v(t))
            if i == 2:
                e("v_sub_f32", v(LS + 12 + i), SF("dtg"), v(t))
            else:
                e("v_mul_f32", v(LS + 12 + i), -1.0, v(t))
        e("v_mul_f32", v(LS + 15), -1.0, v(T0))
        e("v_sub_f32", v(LS + 16), SF("Tmax"), v(T0))
        for k in range(17, 20):
            e("v_mov_b32", v(LS + k), 0)
        pool.free(*y1)
        # thrust-row classification with the PREVIOUS call's E (osqp.c:812-820 -> auxil.c:103-145) -> AGPR homes
        ls, us = g_(), g_()
        rr, ri = g_(), g_()
        c_rhoeq, c_rinveq, c_rmin, c_rmininv = g_(), g_(), g_(), g_()
        e("v_mov_b32", v(c_rhoeq), sg(S_RHO))
        e("v_mov_b32", v(c_rinveq), sg(S_RINV))
        e("v_mov_b32", v(c_rmin), sg(S_C["rmin"]))
        e("v_mov_b32", v(c_rmininv), sg(S_C["rmininv"]))
        for k in range(N):
            e("v_mul_f32", v(ls), v(LS + 15), v(EP[k]))
            e("v_mul_f32", v(us), v(LS + 16), v(EP[k]))
            # default: inequality; equality if us - ls < RHO_TOL; loose if ls < -INFTY*MIN_SCALING and us > INFTY*MIN_SCALING
            e("v_sub_f32", v(t), v(us), v(ls))
            e("v_mov_b32", v(rr), sg(S_C["rho"]))
            e("v_mov_b32", v(ri), sg(S_C["rinv"]))
            e("v_cmp_gt_f32", "vcc", sg(S_C["rhotol"]), v(t))
            e("v_cndmask_b32", v(rr), v(rr), v(c_rhoeq), "vcc")
            e("v_cndmask_b32", v(ri), v(ri), v(c_rinveq), "vcc")
            e("v_cmp_lt_f32_e64", sp(S_M0), v(ls), "-" + sg(S_C["infty_ms"]))
            e("v_cmp_gt_f32_e64", sp(S_M1), v(us), sg(S_C["infty_ms"]))
            e("s_and_b64", "vcc", sp(S_M0), sp(S_M1))
            e("v_cndmask_b32", v(rr), v(rr), v(c_rmin), "vcc")
            e("v_cndmask_b32", v(ri), v(ri), v(c_rmininv), "vcc")
            e("v_accvgpr_write_b32", "a%d" % (A_M + 6 + k), v(rr))
            e("v_accvgpr_write_b32", "a%d" % (A_M + 9 + k), v(ri))
        pool.free(ls, us, rr, ri, c_rhoeq, c_rinveq, c_rmin, c_rmininv, *EP)
        # LDS stash of the raw bounds
        for qd in range(5):
            e("ds_write_b128", "v1", "v[%d:%d]" % (LS + 4 * qd, LS + 4 * qd + 3), qd * 1024)
        # raw A scalars
        dtT0 = g_()
        e("v_mul_f32", v(dtT0), SF("dt"), v(T0))
        s0dt = [g_() for _ in range(3)]
        Bdt = [g_() for _ in range(6)]
        for i in range(3):
            e("v_mul_f32", v(s0dt[i]), SF("dt"), v(s0[i]))
        for i in range(6):
            e("v_mul_f32", v(Bdt[i]), SF("dt"), v(Btau[i]))
        pool.free(*Btau, *ds0, T0)
        # A3: working arrays of the equilibration
        VP = pool.getn(nx + 1)
        VQ = pool.getn_high(st.nq + st.nq % 2)       # covers v246..v255, which factor() needs free (q is retired before)
        VA = pool.getn(st.na + st.na % 2)
        self.VP, self.VQ, self.VA = VP, VQ, VA
        RP = lambda j: VP + st.xs[j]
        RA = lambda p: VA + st.apos[p]
        RQ = lambda j: VQ + st.qslot[j]
        self.RP, self.RA, self.RQ = RP, RA, RQ
        for j in range(nx):
            e("v_accvgpr_read_b32", v(RP(j)), "a%d" % (A_W + WNAMES.index(st.weight_of(j))))
        e("v_mov_b32", v(VP + nx), 0)
        ydes = RF[0:3] + RF[6:9]
        dpdes = RF[3:6]
        WV = {n: pool.get() for n in ("wpr", "wpf", "ws_", "wvr", "wvf")}      # the weights q is built from
        for n, r in WV.items():
            e("v_accvgpr_read_b32", v(r), "a%d" % (A_W + WNAMES.index(n)))
        for j in st.qnz:
            if j < N * NYv:
                k, i = divmod(j, NYv)
                wname = ("wpf" if k == N - 1 else "wpr") if i < 3 else "ws_"
                e("v_mul_f32", v(RQ(j)), "-" + v(WV[wname]), v(ydes[i]))                 # (-w) * ydes
            else:
                k, i = divmod(j - N * NYv, NYv)
                e("v_mul_f32", v(RQ(j)), "-" + v(WV["wvf" if k == N - 1 else "wvr"]), v(dpdes[i]))
        if st.nq % 2:
            e("v_mov_b32", v(VQ + st.nq), 0)
        for p, tag in enumerate(s.A_tag):
            if tag[0] == 'c':
                e("v_mov_b32", v(RA(p)), float(tag[1]))
            elif tag[0] == 'dt':
                e("v_mov_b32", v(RA(p)), SF("dt"))
            elif tag[0] == 'T0dt':
                e("v_mov_b32", v(RA(p)), v(dtT0))
            elif tag[0] == 's0':
                e("v_mov_b32", v(RA(p)), v(s0dt[tag[1]]))
            else:
                e("v_mov_b32", v(RA(p)), v(Bdt[tag[1]]))
        if st.na % 2:
            e("v_mov_b32", v(VA + st.na), 0)
        pool.free(dtT0, *s0dt, *Bdt, t)
        pool.free(*ST)
        pool.free(*RF)
        pool.free(*WV.values())
        pool.free_range(LS, 20)
        self.ruiz_quad() if self.quad else self.ruiz()
        self.recover_and_bounds()
        self.factor()

    # ---- helpers for norms --------------------------------------------------------------------------------
    def maxabs(self, dst, regs, acc_in=False):
        """dst = max(|regs...|) (acc_in: dst already holds a non-negative running maximum)"""
        e = self.e
        regs = list(regs)
        if not acc_in:
            if len(regs) == 1:
                e("v_and_b32", v(dst), 0x7fffffff, v(regs[0]))
                return
            if len(regs) == 2:
                e("v_max_f32", v(dst), "|%s|" % v(regs[0]), "|%s|" % v(regs[1]))
                return
            e("v_max3_f32", v(dst), "|%s|" % v(regs[0]), "|%s|" % v(regs[1]), "|%s|" % v(regs[2]))
            regs = regs[3:]
        while len(regs) >= 2:
            e("v_max3_f32", v(dst), v(dst), "|%s|" % v(regs[0]), "|%s|" % v(regs[1]))
            regs = regs[2:]
        if regs:
            e("v_max_f32", v(dst), v(dst), "|%s|" % v(regs[0]))

    def limit(self, regs, rsq):
        """limit_scaling (scaling.c:7-14: v < 1e-4 -> 1, v > 1e4 -> 1e4) on every register, then 1/sqrt if rsq.
        For a list, one running min / max (v_min3 / v_max3, two registers per instruction) decides wave-wide whether
        ANY value of ANY robot needs limiting -- it never does once the data is equilibrated -- and the exact
        compare / select / min sequence (software-pipelined: the vcc compare -> select pair never feeds the next
        instruction) runs only then."""
        e = self.e
        regs = list(regs)

        def exact():
            prev = None
            for r in regs + [None]:
                if r is not None:
                    e("v_cmp_lt_f32", "vcc", sg(S_C["minscal"]), v(r))
                if prev is not None:
                    e("v_min_f32", v(prev), sg(S_C["maxscal"]), v(prev))
                if r is not None:
                    e("v_cndmask_b32", v(r), 1.0, v(r), "vcc")
                prev = r
        if len(regs) < 6 or not OPT_LIMIT_FAST:
            exact()
        else:
            # two independent min chains and two max chains, interleaved: no instruction reads its predecessor's result
            mn, mx, mn2, mx2 = [self.pool.get() for _ in range(4)]
            halves = [regs[0::2], regs[1::2]]
            chains = [(mn, mx), (mn2, mx2)]
            pos = [0, 0]
            while any(pos[h] < len(halves[h]) for h in (0, 1)):
                for h in (0, 1):
                    vals, p_ = halves[h], pos[h]
                    if p_ >= len(vals):
                        continue
                    rmin, rmax = chains[h]
                    if p_ == 0:
                        take = (vals + [vals[0], vals[0]])[:3]
                        e("v_min3_f32", v(rmin), v(take[0]), v(take[1]), v(take[2]))
                        e("v_max3_f32", v(rmax), v(take[0]), v(take[1]), v(take[2]))
                        pos[h] = 3
                    else:
                        take = (vals[p_:p_ + 2] + [vals[p_]])[:2]
                        e("v_min3_f32", v(rmin), v(rmin), v(take[0]), v(take[1]))
                        e("v_max3_f32", v(rmax), v(rmax), v(take[0]), v(take[1]))
                        pos[h] = p_ + 2
            e("v_min_f32", v(mn), v(mn), v(mn2))
            e("v_max_f32", v(mx), v(mx), v(mx2))
            self.pool.free(mn2, mx2)
            lab = self.label()
            e("v_cmp_gt_f32_e64", sp(S_M0), sg(S_C["minscal"]), v(mn))
            e("v_cmp_lt_f32", "vcc", sg(S_C["maxscal"]), v(mx))
            e("s_or_b64", "vcc", "vcc", sp(S_M0))
            e("s_cbranch_vccz", lab + "f")
            exact()
            e("label", lab)
            self.pool.free(mn, mx)
        if rsq:
            for r in regs:
                e("v_rsq_f32", v(r), v(r))
            e("s_nop", 0)

    # ---- Ruiz equilibration, scaling.c:44-156 ---------------------------------------------------------------
    def ruiz(self):
        e, pool, st, s = self.e, self.pool, self.st, self.s
        nx, nc = s.nx, s.nc
        RP, RA, RQ = self.RP, self.RA, self.RQ
        VE = pool.getn(nc + 1)
        self.VE = VE
        RE = lambda i: VE + st.zs[i]
        self.RE = RE
        cs = pool.get()
        self.cs = cs
        e("v_mov_b32", v(cs), 1.0)
        e("v_mov_b32", v(VE + nc), 0)
        e("s_mov_b32", sg(S_RUIZ), 10)
        top = self.label()
        e("label", top)
        # (a) row norms -> Et
        for i in range(nc):
            self.maxabs(RE(i), [RA(p) for p in st.rows[i]])
        self.limit([RE(i) for i in range(nc)], rsq=True)
        # (b) columns, three slot pairs at a time: norms -> Dt (transient) -> apply
        colpairs = [(st.xinv[2 * k], st.xinv[2 * k + 1]) for k in range(nx // 2)]
        if nx % 2:
            colpairs.append((st.xinv[nx - 1], None))
        CH = 6 if OPT_LIMIT_FAST else 3
        for c0 in range(0, len(colpairs), CH):
            chunk = colpairs[c0:c0 + CH]
            T = [pool.get2() for _ in chunk]
            for (j0, j1), t2 in zip(chunk, T):
                for h, j in enumerate((j0, j1)):
                    if j is None:
                        e("v_mov_b32", v(t2 + 1), 1.0)
                        continue
                    self.maxabs(t2 + h, [RP(j)] + [RA(p) for p in range(s.A_p[j], s.A_p[j + 1])])
            self.limit([t2 + h for (j0, j1), t2 in zip(chunk, T) for h, j in enumerate((j0, j1)) if j is not None], rsq=True)
            for (j0, j1), t2 in zip(chunk, T):
                cols = [j for j in (j0, j1) if j is not None]
                tcol = {j0: t2}
                if j1 is not None:
                    tcol[j1] = t2 + 1
                ents = [p for j in cols for p in range(s.A_p[j], s.A_p[j + 1])]
                pairs = sorted({st.pair_of[p] for p in ents if p in st.pair_of})
                singles = [p for p in ents if p not in st.pair_of]
                for (lo, hi) in pairs:
                    assert st.col_of[lo] in tcol and st.col_of[hi] in tcol
                # A <- diag(Et) A
                for (lo, hi) in pairs:
                    pk(e, "v_pk_mul_f32", RA(lo), [P2(RA(lo)), PSEL(RE(s.A_i[lo]), RE(s.A_i[hi]))])
                for p in singles:
                    e("v_mul_f32", v(RA(p)), v(RA(p)), v(RE(s.A_i[p])))
                # P <- Dt P Dt (first factor), q <- Dt q
                if j1 is not None:
                    pk(e, "v_pk_mul_f32", RP(j0), [P2(RP(j0)), P2(t2)])
                else:
                    e("v_mul_f32", v(RP(j0)), v(RP(j0)), v(t2))
                qj = [j for j in cols if j in st.qslot]
                if len(qj) == 2 and RQ(qj[0]) % 2 == 0 and RQ(qj[1]) == RQ(qj[0]) + 1:
                    pk(e, "v_pk_mul_f32", RQ(qj[0]), [P2(RQ(qj[0])), P2(t2)])
                else:
                    for j in qj:
                        e("v_mul_f32", v(RQ(j)), v(RQ(j)), v(tcol[j]))
                # A <- A diag(Dt)
                for (lo, hi) in pairs:
                    pk(e, "v_pk_mul_f32", RA(lo), [P2(RA(lo)), PSEL(tcol[st.col_of[lo]], tcol[st.col_of[hi]])])
                for p in singles:
                    e("v_mul_f32", v(RA(p)), v(RA(p)), v(tcol[st.col_of[p]]))
                if j1 is not None:
                    pk(e, "v_pk_mul_f32", RP(j0), [P2(RP(j0)), P2(t2)])
                else:
                    e("v_mul_f32", v(RP(j0)), v(RP(j0)), v(t2))
            for t2 in T:
                pool.free(t2, t2 + 1)
        # (c) cost normalisation: c_t = 1 / limit(max(mean_j P_jj, limit(|q|_inf)))
        acc = pool.get2()
        VP, VQ = self.VP, self.VQ
        pk(e, "v_pk_add_f32", acc, [P2(VP), P2(VP + 2)])
        for k in range(4, nx + 1, 2):
            pk(e, "v_pk_add_f32", acc, [P2(acc), P2(VP + k)])
        qn, ct, t = pool.get(), pool.get(), pool.get()
        self.maxabs(qn, [VQ + k for k in range(st.nq)])
        e("v_add_f32", v(acc), v(acc), v(acc + 1))
        e("v_mul_f32", v(acc), sg(S_C["c45"]), v(acc))
        self.limit([qn], rsq=False)
        e("v_max_f32", v(acc), v(acc), v(qn))
        self.limit([acc], rsq=False)
        self.rcp_nr(ct, acc, t)
        for k in range(0, nx + 1, 2):
            pk(e, "v_pk_mul_f32", VP + k, [P2(VP + k), PB(ct)])
        for k in range(0, st.nq + st.nq % 2, 2):
            pk(e, "v_pk_mul_f32", VQ + k, [P2(VQ + k), PB(ct)])
        e("v_mul_f32", v(cs), v(cs), v(ct))
        pool.free(acc, acc + 1, qn, ct, t)
        e("s_sub_i32", sg(S_RUIZ), sg(S_RUIZ), 1)
        e("s_cmp_gt_i32", sg(S_RUIZ), 0)
        e("s_cbranch_scc1", top + "b")


    # ---- Ruiz equilibration on the lane QUAD (the quad form of the stream, asmquad.py) ---------------------------------
    def ruiz_quad(self):
        """The ten passes with a third of the matrix per lane. Every lane of the quad holds the whole raw problem in the
        one-lane registers (the phases before ran redundantly), so nothing is allocated: the register of the lane-0 member
        of every entry slot / column triple HOSTS the slot -- lanes 1 and 2 move their member in under a lane mask -- and the
        other members' registers serve as temporaries. A register then holds the x / y / z members of a row / column triple
        (or the three horizon steps of an input): 13 row norms + 15 column norms per pass instead of 39 + 45, and nothing
        crosses lanes but the nine input columns (their entries sit in the lanes of their ROWS: norms reduced by DPP,
        scalings fetched from the lane of the step) and the cost normalisation's sum / maximum (a four-lane butterfly; lane
        3 duplicates lane 0 up to there and is zeroed out of it). At the end every word goes back to its one-lane home in
        all four lanes. Same operations per value as ruiz(); the sum of P_jj is associated per lane (rounding only)."""
        from . import asmquad, asmquad64
        e, pool, st, s = self.e, self.pool, self.st, self.s
        nx, nc = s.nx, s.nc
        RP, RA, RQ = self.RP, self.RA, self.RQ
        VE = pool.getn(nc + 1)
        self.VE = VE
        self.RE = lambda i: VE + st.zs[i]
        cs = pool.get()
        self.cs = cs
        e("v_mov_b32", v(cs), 1.0)
        e("v_mov_b32", v(VE + nc), 0)
        qp = asmquad.QuadPlan(st)
        rp = asmquad64.RuizQuadPlan(s, qp)
        # hosts and the registers they free
        hostA = {key: RA(d[min(d)]) for key, d in rp.slots.items()}
        colmem = {}                       # column register -> {lane: x index}
        for j in range(nx):
            ln, C = qp.xhome[j]
            colmem.setdefault(C, {})[ln] = j
        hostP = {C: RP(m[min(m)]) for C, m in colmem.items()}
        hostQ = {C: RQ(min(jj for jj in m.values() if jj in st.qslot)) for C, m in colmem.items() if any(jj in st.qslot for jj in m.values())}
        # (q is structurally zero on the other columns)
        for C, m in colmem.items():
            if C in hostQ:
                assert all(jj in st.qslot for jj in m.values())
        dead = [RA(p) for key, d in rp.slots.items() for ln, p in d.items() if RA(p) != hostA[key]] + \
               [RP(j) for C, m in colmem.items() for j in m.values() if RP(j) != hostP[C]]
        dead = sorted(set(dead))
        ET = [dead.pop() for _ in range(13)]
        DT = [dead.pop() for _ in range(15)]
        t, m7, acc, qn, ct, tt = (dead.pop() for _ in range(6))
        masks = (asmquad.S_L0, asmquad.S_L1, asmquad.S_L2)
        S_L3 = 98
        keep = sorted(set(range(self.VP, self.VP + nx + 1)) | set(range(self.VQ, self.VQ + st.nq + st.nq % 2)) |
                      set(range(self.VA, self.VA + st.na + st.na % 2)) | {cs, VE + nc})
        e("quad_begin", "ruiz")
        e("s_mov_b64", sp(asmquad.S_EXEC), "exec")
        for ln, m in enumerate(masks + (S_L3,)):
            e("s_mov_b32", sg(m), 0x11111111 << ln)
            e("s_mov_b32", sg(m + 1), 0x11111111 << ln)
            e("s_and_b64", sp(m), sp(m), sp(asmquad.S_EXEC))
        # ---- entry: lanes 1 and 2 bring their members into the hosts
        for ln in (1, 2):
            e("s_mov_b64", "exec", sp(masks[ln]))
            for key, d in rp.slots.items():
                if ln in d and RA(d[ln]) != hostA[key]:
                    e("v_mov_b32", v(hostA[key]), v(RA(d[ln])))
                elif ln not in d:
                    e("v_mov_b32", v(hostA[key]), 0)
            for C, m in colmem.items():
                if ln in m:
                    if RP(m[ln]) != hostP[C]:
                        e("v_mov_b32", v(hostP[C]), v(RP(m[ln])))
                    if C in hostQ and RQ(m[ln]) != hostQ[C]:
                        e("v_mov_b32", v(hostQ[C]), v(RQ(m[ln])))
        # (a slot without a lane-0 member: lane 0 and its duplicate, lane 3, hold another lane's entry there -- zero)
        for key, d in rp.slots.items():
            if 0 not in d:
                e("s_mov_b64", "exec", sp(masks[0]))
                e("v_mov_b32", v(hostA[key]), 0)
                e("s_mov_b64", "exec", sp(S_L3))
                e("v_mov_b32", v(hostA[key]), 0)
        e("s_mov_b64", "exec", sp(asmquad.S_EXEC))
        e("s_nop", 4)
        e("s_mov_b32", sg(S_RUIZ), 10)
        top = self.label()
        e("label", top)
        # (a) row norms -> Et
        byrow, bycol = {}, {}
        for key in sorted(rp.slots):
            byrow.setdefault(key[0], []).append(key)
            bycol.setdefault(key[1], []).append(key)
        for R in range(13):
            self.maxabs(ET[R], [hostA[k] for k in byrow[R]])
        self.limit(ET, rsq=True)
        # (b) column norms from the untouched columns -> Dt
        for C in range(15):
            own = [k for k in bycol.get(C, []) if rp.kind[k] == "own"]
            self.maxabs(DT[C], [hostP[C]] + [hostA[k] for k in own])
            for k in [k for k in bycol.get(C, []) if rp.kind[k] != "own"]:
                r = hostA[k]
                e("v_max_f32_dpp", v(m7), "|%s|" % v(r), "|%s|" % v(r), asmquad.qperm(asmquad64.ROT1))
                e("v_max_f32_dpp", v(m7), "|%s|" % v(r), v(m7), asmquad.qperm(asmquad64.ROT2))       # max over lanes 0..2
                e("s_mov_b64", "exec", sp(masks[rp.kind[k]]))
                e("v_max_f32", v(DT[C]), v(DT[C]), v(m7))
                e("s_mov_b64", "exec", sp(asmquad.S_EXEC))
                e("s_nop", 4)
        self.limit(DT, rsq=True)
        # apply: A <- diag(Et) A diag(Dt), P <- Dt P Dt, q <- Dt q
        for C in range(15):
            for k in bycol.get(C, []):
                r = hostA[k]
                e("v_mul_f32", v(r), v(r), v(ET[k[0]]))
                if rp.kind[k] == "own":
                    e("v_mul_f32", v(r), v(r), v(DT[C]))
                else:
                    kk = rp.kind[k]
                    e("v_mul_f32_dpp", v(r), v(DT[C]), v(r), asmquad.qperm([kk] * 4))
            e("v_mul_f32", v(hostP[C]), v(hostP[C]), v(DT[C]))
            e("v_mul_f32", v(hostP[C]), v(hostP[C]), v(DT[C]))
            if C in hostQ:
                e("v_mul_f32", v(hostQ[C]), v(hostQ[C]), v(DT[C]))
        # (c) cost normalisation: c_t = 1 / limit(max(mean_j P_jj, limit(|q|_inf)))
        e("v_add_f32", v(acc), v(hostP[0]), v(hostP[1]))
        for C in range(2, 15):
            e("v_add_f32", v(acc), v(acc), v(hostP[C]))
        self.maxabs(qn, [hostQ[C] for C in sorted(hostQ)])
        e("s_mov_b64", "exec", sp(S_L3))
        e("v_mov_b32", v(acc), 0)
        e("v_mov_b32", v(qn), 0)
        e("s_mov_b64", "exec", sp(asmquad.S_EXEC))
        e("s_nop", 4)
        for perm_ in ([1, 0, 3, 2], [2, 3, 0, 1]):
            e("v_add_f32_dpp", v(acc), v(acc), v(acc), asmquad.qperm(perm_))
            e("v_max_f32_dpp", v(qn), v(qn), v(qn), asmquad.qperm(perm_))
            e("s_nop", 1)
        e("v_mul_f32", v(acc), sg(S_C["c45"]), v(acc))
        self.limit([qn], rsq=False)
        e("v_max_f32", v(acc), v(acc), v(qn))
        self.limit([acc], rsq=False)
        self.rcp_nr(ct, acc, tt)
        for C in range(15):
            e("v_mul_f32", v(hostP[C]), v(hostP[C]), v(ct))
            if C in hostQ:
                e("v_mul_f32", v(hostQ[C]), v(hostQ[C]), v(ct))
        e("v_mul_f32", v(cs), v(cs), v(ct))
        e("s_sub_i32", sg(S_RUIZ), sg(S_RUIZ), 1)
        e("s_cmp_gt_i32", sg(S_RUIZ), 0)
        e("s_cbranch_scc1", top + "b")
        # ---- exit: every word back to its one-lane home in all four lanes (the hosts last: they are sources until then)
        e("s_nop", 1)

        def bcast(dst, src, ln):
            e("v_mov_b32_dpp", v(dst), v(src), asmquad.qperm([ln] * 4))
        for key, d in rp.slots.items():
            for ln in sorted(d, reverse=True):
                bcast(RA(d[ln]), hostA[key], ln)
        for C, m in colmem.items():
            for ln in sorted(m, reverse=True):
                bcast(RP(m[ln]), hostP[C], ln)
            if C in hostQ:
                for ln in sorted(m, reverse=True):
                    bcast(RQ(m[ln]), hostQ[C], ln)
        e("v_mov_b32", v(self.VP + nx), 0)
        if st.nq % 2:
            e("v_mov_b32", v(self.VQ + st.nq), 0)
        if st.na % 2:
            e("v_mov_b32", v(self.VA + st.na), 0)
        e("s_nop", 1)
        e("quad_end", tuple(keep))

    # ---- D, E recovery, scaled bounds, q / bounds -> loop homes ---------------------------------------------------
    def recover_and_bounds(self):
        e, pool, st, s = self.e, self.pool, self.st, self.s
        N, nx, nc, neq = s.N, s.nx, s.nc, st.neq
        RP, RA, RQ, RE, cs = self.RP, self.RA, self.RQ, self.RE, self.cs
        SF = lambda n: sg(S_F[n])
        cinv, t, d, r = pool.get(), pool.get(), pool.get(), pool.get()
        self.rcp_nr(cinv, cs, t)
        voff = pool.get()
        npark = 0 if self.quad else N_SPARE_D
        WS_ST = (OPT_WS_ST,) if OPT_WS_ST and not self.quad else ()
        self.rows_ptr(voff, npark)     # the kernel's `ws` pointer is the workspace row WS_DS (host side): rows D 0.., E 45.., c 84
        rows_of_col = {}
        for i in range(nc):
            rows_of_col.setdefault(st.unit[i][1], []).append(i)
        IWV = {n: pool.get() for n in WNAMES}
        for k, n in enumerate(WNAMES):
            e("v_accvgpr_read_b32", v(IWV[n]), "a%d" % (A_IW + k))
        for j in range(nx):
            # D_j = sqrt((P_jj / c) / P_raw,jj)   
            e("v_mul_f32", v(d), v(RP(j)), v(cinv))
            e("v_mul_f32", v(d), v(IWV[st.weight_of(j)]), v(d))
            e("v_sqrt_f32", v(d), v(d))
            e("s_nop", 0)
            if j < npark:
                e("v_accvgpr_write_b32", "a%d" % (A_SPARE + j), v(d))
            else:
                e("global_store_dword", v(voff), v(d), sp(S_PTR["ws"]), *WS_ST)
                self.adv(voff)
            if j in rows_of_col:
                e("v_rcp_f32", v(r), v(d))
                e("s_nop", 0)
                for i in rows_of_col[j]:
                    e("v_mul_f32", v(RE(i)), "|%s|" % v(RA(st.unit[i][0])), v(r))       # E_i = |A_ip| / D_p
        assert asmgen.WS_ES == asmgen.WS_DS + nx and asmgen.WS_C == asmgen.WS_ES + nc
        for i in range(nc):
            e("global_store_dword", v(voff), v(RE(i)), sp(S_PTR["ws"]), *WS_ST)
            self.adv(voff)
        if npark:
            e("v_accvgpr_write_b32", "a%d" % (A_SPARE + N_SPARE_D), v(cs))
        else:
            e("global_store_dword", v(voff), v(cs), sp(S_PTR["ws"]))
        self.store_rows("ctrl", nx + 2 * nc + 1, [RE(neq + k) for k in range(N)], voff)     # Eprev of the next step
        pool.free(cinv, d, r, voff, cs, *IWV.values())
        # scaled bounds from the LDS stash
        LS = pool.getn(20)
        for qd in range(5):
            e("ds_read_b128", "v[%d:%d]" % (LS + 4 * qd, LS + 4 * qd + 3), "v1", qd * 1024)
        zero = pool.get()
        e("v_mov_b32", v(zero), 0)
        # q -> AGPR home while the LDS reads are in flight
        for j in range(nx):
            e("v_accvgpr_write_b32", "a%d" % (A_Q + j), v(RQ(j)) if j in st.qslot else v(zero))
        e("s_waitcnt", "lgkmcnt(0)")
        src = {}
        for i in range(6):
            src[i] = LS + i
            src[3 * 6 + i] = LS + 6 + i
        for i in range(3):
            src[4 * 6 + i] = LS + 12 + i
        for i in range(neq):
            if i in src:
                e("v_mul_f32", v(t), v(src[i]), v(RE(i)))
                e("v_accvgpr_write_b32", "a%d" % (A_LO + i), v(t))
            elif i == 5 * 6 + 2:
                e("v_mul_f32", v(t), SF("dtg"), v(RE(i)))
                e("v_accvgpr_write_b32", "a%d" % (A_LO + i), v(t))
            else:
                e("v_accvgpr_write_b32", "a%d" % (A_LO + i), v(zero))
        for k in range(N):
            e("v_mul_f32", v(t), v(LS + 15), v(RE(neq + k)))
            e("v_accvgpr_write_b32", "a%d" % (A_M + k), v(t))
            e("v_mul_f32", v(t), v(LS + 16), v(RE(neq + k)))
            e("v_accvgpr_write_b32", "a%d" % (A_M + N + k), v(t))
        pool.free(zero, t)
        pool.free_range(LS, 20)
        pool.free_range(self.VE, nc + 1)
        pool.free_range(self.VQ, st.nq + st.nq % 2)

    # ---- KKT fill (kkt.c:184-222) + up-looking LDL' (qdldl.c:86-247); L, 1/D go straight to their loop homes ----------
    def factor(self):
        e, pool, st, s = self.e, self.pool, self.st, self.s
        nx, nc, nk, neq = s.nx, s.nc, s.nk, st.neq
        RP, RA = self.RP, self.RA
        lpos = st.lpos
        last_row = {}       # column -> row of its last L entry
        for c in range(nk):
            if s.L_p[c + 1] > s.L_p[c]:
                last_row[c] = s.L_i[s.L_p[c + 1] - 1]
        done_at = {}
        for c, r in last_row.items():
            done_at.setdefault(r, []).append(c)
        nL, Dinv = {}, {}
        # XV_N more L entries live in v246..v255 through the loop (2 AGPR reads per iteration each otherwise)
        asmgen.XV_COUNT, asmgen.XV_BASE = (XV_N if OPT_XV and not self.quad else 0), XV_B   # (quad: v254 / v255 hold y)
        pool.reserve(246, 10)         # L words and/or ring slots (OPT_RING)
        t = pool.get()

        def retire_dinv(k):
            e("v_accvgpr_write_b32", "a%d" % (A_D + k), v(Dinv[k]))
            pool.free(Dinv.pop(k))

        def retire_col(c):
            for j in range(s.L_p[c], s.L_p[c + 1]):
                pos = lpos[j]
                if pos < NLDS:
                    e("ds_write_b32", "v1", v(nL[j]), (pos // 4) * 1024 + (pos % 4) * 4)
                elif NLDS + NVZ <= pos < NLDS + NVZ + asmgen.XV_COUNT:
                    e("v_mov_b32", v(XV_B + pos - NLDS - NVZ), v(nL[j]))
                else:
                    e("v_accvgpr_write_b32", "a%d" % (A_L + pos - NLDS), v(nL[j]))
                pool.free(nL.pop(j))
            retire_dinv(c)

        for op in s.factor_ops:
            k = op["k"]
            src = s.K_src[op["diag"]]
            dk = pool.get()
            if src[0] == 'P':
                e("v_add_f32", v(dk), sg(S_SIGMA), v(RP(src[1])))
                pool.free(RP(src[1]))
            else:
                assert src[0] == 'R'
                i = src[1]
                if i < neq:
                    e("v_mul_f32", v(dk), -1.0, self._vconst_rinv())
                else:
                    e("v_accvgpr_read_b32", v(dk), "a%d" % (A_M + 9 + i - neq))
                    e("s_nop", 0)
                    e("v_mul_f32", v(dk), -1.0, v(dk))
            y = {}          # row -> (reg, owned)
            for (b, p) in op["init"]:
                assert s.K_src[p][0] == 'A'
                y[b] = (RA(s.K_src[p][1]), False)
            for (c, upd, new) in op["elim"]:
                yc, owned = y.pop(c)
                for (lj, row) in upd:
                    if row in y:
                        reg, own = y[row]
                        if own:
                            e("v_fmac_f32", v(reg), v(nL[lj]), v(yc))
                        else:
                            nr = pool.get()
                            e("v_fma_f32", v(nr), v(nL[lj]), v(yc), v(reg))
                            pool.free(reg)          # the K entry is consumed
                            y[row] = (nr, True)
                    else:
                        nr = pool.get()
                        e("v_mul_f32", v(nr), v(nL[lj]), v(yc))
                        y[row] = (nr, True)
                nl = pool.get()
                e("v_mul_f32", v(nl), "-" + v(yc), v(Dinv[c]))       # the factor is kept negated (asmgen: dst += (-L) src)
                e("v_fmac_f32", v(dk), v(yc), v(nl))                  # d_k -= y_c L_kc
                nL[new] = nl
                pool.free(yc)
            assert not y
            di = pool.get()
            self.rcp_nr(di, dk, t)
            pool.free(dk)
            Dinv[k] = di
            for c in done_at.get(k, []):
                retire_col(c)
            if k not in last_row:
                retire_dinv(k)
        assert not nL and not Dinv
        pool.free(t)
        if self._rinv_reg is not None:
            pool.free(self._rinv_reg)
            self._rinv_reg = None
        # pads of the working arrays
        if st.na % 2:
            pool.free(self.VA + st.na)
        pool.free(self.VP + nx)

    _rinv_reg = None

    def _vconst_rinv(self):
        if self._rinv_reg is None:
            self._rinv_reg = self.pool.get()
            self.e("v_mov_b32", v(self._rinv_reg), sg(S_RINV))
        return v(self._rinv_reg)

    # ---- ADMM iterations: asmgen.body with the factor / bounds already in their homes --------------------------
    def admm(self):
        e, pool, st, s = self.e, self.pool, self.st, self.s
        nx, nc = s.nx, s.nc
        left = set(range(VFIRST, VEND)) - pool.free_ - set(range(246, 256))
        assert not left, "phase A leaked registers: %s" % sorted(left)
        pool.reserve(V_W, V_X - V_W)
        pool.reserve(V_X, V_Y - V_X)
        pool.reserve(V_Y, V_Z - V_Y)
        pool.reserve(V_Z, 40)
        voff = 245          # free here: the loop's temporaries start at v214, its extra L registers at v246
        self.load_rows("ctrl", 0, [V_X + st.xs[r] for r in range(nx)] + [V_Y + st.zs[r] for r in range(nc)] +
                       [V_Z + st.zs[r] for r in range(nc)], voff)
        for pad in (V_X + nx, V_Y + nc, V_Z + nc, V_W + nx, V_WZ + nc):
            e("v_mov_b32", v(pad), 0)
        e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
        plan = st.plan
        zk = dict(qzero=st.qzero, lzero=st.lzero) if OPT_ZSKIP else {}
        asmgen.body(e, s, first=True, capture=True, plan=plan, delta_in_w=True, **zk)
        if self.quad:
            # iterations 2..maxIter on the lane quad: entry transposition, the quad bodies, the broadcast back into the
            # one-lane homes phase C reads (asmquad.section); nothing else of the stream changes
            from . import asmquad
            lab6 = self.label()
            e("s_cmp_lt_i32", sg(S_ITERS), 2)
            e("s_cbranch_scc1", lab6 + "f")
            asmquad.section(e, asmquad.QuadPlan(st), self.label)
            e("label", lab6)
        else:
            self._admm_rest(plan, zk)
        # x, y, thrust-row z, delta_x (x part of W), delta_y (z part of W) stay where they are; z == l on the dynamics rows
        for i in range(st.neq):
            pool.free(V_Z + st.zs[i], kill=True)
        pool.free_range(246, 10)


    def _admm_rest(self, plan, zk):
        """Iterations 2 .. maxIter: ONE body (round 5; rounds 2-4 carried a third, capturing copy of the 6.7-KB body for the
        last iteration). What phase C needs of the last iteration -- delta_x, delta_y for the infeasibility certificates
        (auxil.c:362-512) -- follows from what the plain body leaves behind: W still holds the KKT solution (x~ | nu), and
        x_new = alpha x~ + (1 - alpha) x_prev gives delta_x = x_new - x_prev = k (x_new - x~) with k = alpha / (alpha - 1);
        on the dynamics rows y_new = (1 - alpha) y_prev + alpha nu gives delta_y = k (y_new - nu);
        the three thrust rows leave their delta_y in W themselves (asmgen.body dy3_in_w). 41 packed instructions once per
        step instead of a body copy: the stream is 7 KB shorter and the default path's footprint fits the 64-KB
        instruction cache. x, y, z are the plain body's, bit for bit; delta_x / delta_y differ from the captured
        differences by rounding and feed only the certificate tests."""
        e, s, st = self.e, self.s, self.st
        import numpy as np
        for p_ in range(NVZ):      # the z registers of the dynamics rows take the L entries parked in a0..a35
            e("v_accvgpr_read_b32", v(V_Z + p_), "a%d" % (A_L + p_))
        lab7, lab6 = self.label(), self.label()
        e("s_sub_i32", sg(S_CNT), sg(S_ITERS), 1)
        e("s_cmp_lt_i32", sg(S_CNT), 1)
        e("s_cbranch_scc1", lab6 + "f")
        e("label", lab7)
        asmgen.body(e, s, first=False, capture=False, plan=plan, lv=True, dy3_in_w=True, **zk)
        e("s_sub_i32", sg(S_CNT), sg(S_CNT), 1)
        e("s_cmp_gt_i32", sg(S_CNT), 0)
        e("s_cbranch_scc1", lab7 + "b")
        # delta_x = x_new - x_prev with x_prev = (x_new - alpha x~) / (1 - alpha):  delta_x = k (x_new - x~),  k = alpha / (alpha - 1)
        # delta_y (dynamics rows) = y_new - y_prev with y_new = (1 - alpha) y_prev + alpha nu:  delta_y = k (y_new - nu)
        k = float(np.float32(1.6) / (np.float32(1.6) - np.float32(1.0)))
        e("s_mov_b32", sg(S_TMP), f32bits(k))
        nx, neq = s.nx, st.neq
        for p_ in range(0, nx - 1, 2):
            pk(e, "v_pk_add_f32", V_W + p_, [P2(V_X + p_), P2(V_W + p_)], [0, 1])
        for p_ in range(0, nx - 1, 2):
            pk(e, "v_pk_mul_f32", V_W + p_, [PS(S_TMP), P2(V_W + p_)])
        if nx % 2:
            jl = st.xinv[nx - 1]
            e("v_sub_f32", v(V_W + st.xs[jl]), v(V_X + st.xs[jl]), v(V_W + st.xs[jl]))
            e("v_mul_f32", v(V_W + st.xs[jl]), sg(S_TMP), v(V_W + st.xs[jl]))
        for p_ in range(0, neq, 2):
            pk(e, "v_pk_add_f32", V_WZ + p_, [P2(V_Y + p_), P2(V_WZ + p_)], [0, 1])
        for p_ in range(0, neq, 2):
            pk(e, "v_pk_mul_f32", V_WZ + p_, [PS(S_TMP), P2(V_WZ + p_)])
        e("label", lab6)

    # ---- phase C ------------------------------------------------------------------------------------------
    def phase_c(self):
        e, pool, st, s = self.e, self.pool, self.st, self.s
        N, NYv = s.N, symbolic.NY
        nx, nc, neq = s.nx, s.nc, st.neq
        SF = lambda n: sg(S_F[n])
        g_ = pool.get
        XR = lambda j: V_X + st.xs[j]
        YR = lambda i: V_Y + st.zs[i]
        ZR = lambda i: V_Z + st.zs[i]
        DXR = lambda j: V_W + st.xs[j]
        DYR = lambda i: V_WZ + st.zs[i]
        voff = g_()
        # 1. D, c, T0
        def slot_blocks(n, chunk=16):
            """registers for n (+ pad) slots in aligned blocks of <= chunk slots: slot -> register, blocks"""
            n2 = n + n % 2
            blocks = [(lo, pool.getn(min(chunk, n2 - lo))) for lo in range(0, n2, chunk)]
            def reg(slot):
                for lo, base in reversed(blocks):
                    if slot >= lo:
                        return base + slot - lo
            return reg, [(base, min(chunk, n2 - lo)) for lo, base in blocks]
        DS, dblocks = slot_blocks(nx)
        DR = lambda j: DS(st.xs[j])
        cr, T0 = g_(), g_()
        npark = 0 if self.quad else N_SPARE_D
        self.load_rows("ws", npark, [DR(j) for j in range(npark, nx)], voff)
        if npark:
            for j in range(npark):
                e("v_accvgpr_read_b32", v(DR(j)), "a%d" % (A_SPARE + j))
            e("v_accvgpr_read_b32", v(cr), "a%d" % (A_SPARE + N_SPARE_D))
        else:
            self.load_rows("ws", asmgen.WS_C - asmgen.WS_DS, [cr], voff)
        self.load_rows("ctrl", nx + 2 * nc, [T0], voff)
        e("v_mov_b32", v(DS(nx)), 0)
        # 2. the controller record goes back now (a cold start, if any, rewrites it below): x, y, z
        zt = [g_() for _ in range(4)]
        NT_ST = ("nt",) if self.nt in ("1", "st") else ()
        self.rows_ptr(voff, 0)
        for r in range(nx):
            e("global_store_dword", v(voff), v(XR(r)), sp(S_PTR["ctrl"]), *NT_ST)
            self.adv(voff)
        for r in range(nc):
            e("global_store_dword", v(voff), v(YR(r)), sp(S_PTR["ctrl"]), *NT_ST)
            self.adv(voff)
        for r in range(nc):
            if r < neq:
                t_ = zt[r % 4]
                e("v_accvgpr_read_b32", v(t_), "a%d" % (A_LO + r))
                e("global_store_dword", v(voff), v(t_), sp(S_PTR["ctrl"]), *NT_ST)
            else:
                e("global_store_dword", v(voff), v(ZR(r)), sp(S_PTR["ctrl"]), *NT_ST)
            self.adv(voff)
        pool.free(*zt)
        e("s_waitcnt", "vmcnt(0)")
        # 3. unscaled step and solution: dxu = D dx (in W), xu = D x (in X)
        for k in range(0, nx + 1, 2):
            pk(e, "v_pk_mul_f32", V_W + k, [P2(DS(k)), P2(V_W + k)])
            pk(e, "v_pk_mul_f32", V_X + k, [P2(DS(k)), P2(V_X + k)])
        for base, n_ in dblocks:
            pool.free_range(base, n_)
        # 4. E, state, ref
        ES, eblocks = slot_blocks(nc)
        ER = lambda i: ES(st.zs[i])
        ST = [g_() for _ in range(18)]
        RF = [g_() for _ in range(9)]
        self.load_rows("ws", asmgen.WS_ES - asmgen.WS_DS, [ER(i) for i in range(nc)], voff)
        self.load_rows("state", 0, ST, voff)
        self.load_rows("ref", 0, RF, voff)
        e("v_mov_b32", v(ES(nc)), 1.0)
        # the step's reference again (phase A's task-table entry)
        lab_tk = self.label()
        e("s_cmp_eq_u64", sp(S_PTR["taskf"]), 0)
        e("s_cbranch_scc1", lab_tk + "f")
        self.load_taskf()
        for i in range(3):
            e("v_add_f32", v(RF[i]), sg(S_M0 + i), v(RF[i]))
            e("v_mov_b32", v(RF[3 + i]), sg(S_M0 + 3 + i))
        e("v_mov_b32", v(RF[6]), sg(S_M0 + 6))
        e("v_mov_b32", v(RF[7]), 0)
        e("v_mov_b32", v(RF[8]), sg(S_M0 + 7))
        e("label", lab_tk)
        cinv, t = g_(), g_()
        self.rcp_nr(cinv, cr, t)
        pool.free(cr)
        ibi0, ibi1 = g_(), g_()
        lab_ib, lab_ib2 = self.label(), self.label()
        e("s_cmp_eq_u64", sp(S_PTR["Ib"]), 0)
        e("s_cbranch_scc1", lab_ib + "f")
        self.load_rows("Ib", 0, [ibi0, ibi1], voff)
        e("s_waitcnt", "vmcnt(0)")
        tt = g_()
        for r in (ibi0, ibi1):
            e("v_mov_b32", v(tt), v(r))
            self.rcp_nr(r, tt, t)
        pool.free(tt)
        e("s_branch", lab_ib2 + "f")
        e("label", lab_ib)
        e("v_mov_b32", v(ibi0), SF("Ibi0"))
        e("v_mov_b32", v(ibi1), SF("Ibi1"))
        e("label", lab_ib2)
        e("s_waitcnt", "vmcnt(0)")
        # 5. dyE = E dy (in W), yun = E y / c (in Y), unscaled thrust-row z
        for k in range(0, nc + 1, 2):
            pk(e, "v_pk_mul_f32", V_WZ + k, [P2(ES(k)), P2(V_WZ + k)])
            pk(e, "v_pk_mul_f32", V_Y + k, [P2(ES(k)), P2(V_Y + k)])
        for k in range(0, nc + 1, 2):
            pk(e, "v_pk_mul_f32", V_Y + k, [P2(V_Y + k), PB(cinv)])
        pool.free(cinv)
        zu3 = [g_() for _ in range(N)]
        for k in range(N):
            e("v_rcp_f32", v(zu3[k]), v(ER(neq + k)))
        e("s_nop", 0)
        for k in range(N):
            e("v_mul_f32", v(zu3[k]), v(ZR(neq + k)), v(zu3[k]))
        for base, n_ in eblocks:
            pool.free_range(base, n_)
        for k in range(N):
            pool.free(ZR(neq + k))
        pool.free(V_Z + nc)
        p0, R0, dq0 = ST[0:3], ST[3:12], ST[12:18]
        # 6. raw A scalars and raw bounds of this step's QP (phase A's assembly again)
        dtT0 = g_()
        e("v_mul_f32", v(dtT0), SF("dt"), v(T0))
        s0dt = [g_() for _ in range(3)]
        Bdt = [g_() for _ in range(6)]
        for i in range(3):
            e("v_mul_f32", v(s0dt[i]), SF("dt"), v(R0[6 + i]))
            e("v_mul_f32", v(Bdt[i]), "-" + v(R0[i + 3]), v(ibi0))
            e("v_mul_f32", v(Bdt[3 + i]), v(R0[i]), v(ibi1))
        for i in range(6):
            e("v_mul_f32", v(Bdt[i]), SF("dt"), v(Bdt[i]))
        pool.free(ibi0, ibi1)
        ds0 = [g_() for _ in range(3)]
        for r in range(3):
            e("v_mul_f32", v(t), "-" + v(R0[r]), v(dq0[4]))
            e("v_mul_f32", v(ds0[r]), v(R0[r + 3]), v(dq0[3]))
            e("v_add_f32", v(ds0[r]), v(t), v(ds0[r]))
            e("v_mul_f32", v(ds0[r]), -1.0, v(ds0[r]))
        y0 = p0 + R0[6:9]
        dy0 = dq0[0:3] + ds0
        LR = {}                 # row -> register of its raw bound (rows not listed are 0; row 32 is the constant dt g)
        y1 = [g_() for _ in range(6)]
        for i in range(6):
            LR[i] = g_()
            e("v_mul_f32", v(t), SF("dt"), v(dy0[i]))
            e("v_add_f32", v(y1[i]), v(y0[i]), v(t))
            e("v_mul_f32", v(LR[i]), -1.0, v(y1[i]))
        for i in range(6):
            LR[18 + i] = g_()
            if i < 3:
                e("v_mul_f32", v(t), v(T0), v(y0[i + 3]))
                e("v_mul_f32", v(t), SF("dt"), v(t))
                e("v_sub_f32", v(LR[18 + i]), "-" + v(dy0[i]), v(t))
                if i == 2:
                    e("v_add_f32", v(LR[18 + i]), SF("dtg"), v(LR[18 + i]))
            else:
                e("v_mul_f32", v(LR[18 + i]), -1.0, v(dy0[i]))
        for i in range(3):
            LR[24 + i] = g_()
            e("v_mul_f32", v(t), v(T0), v(y1[i + 3]))
            e("v_mul_f32", v(t), SF("dt"), v(t))
            if i == 2:
                e("v_sub_f32", v(LR[24 + i]), SF("dtg"), v(t))
            else:
                e("v_mul_f32", v(LR[24 + i]), -1.0, v(t))
        lT, uT = g_(), g_()
        e("v_mul_f32", v(lT), -1.0, v(T0))
        e("v_sub_f32", v(uT), SF("Tmax"), v(T0))
        pool.free(*y1, *ds0)
        CONST_ROW = 5 * 6 + 2

        def coef(p):
            tag = s.A_tag[p]
            if tag[0] == 'c':
                return tag[1]
            return {"dt": SF("dt"), "T0dt": v(dtT0)}.get(tag[0]) or (v(s0dt[tag[1]]) if tag[0] == 's0' else v(Bdt[tag[1]]))

        def dot(dst, ents, vec, idx):
            """dst = sum over entries p of raw_p * vec(idx(p))"""
            first = True
            for p in ents:
                c, x = coef(p), v(vec(idx(p)))
                if first:
                    if c == 1.0:
                        e("v_mov_b32", v(dst), x)
                    elif c == -1.0:
                        e("v_mul_f32", v(dst), -1.0, x)
                    else:
                        e("v_mul_f32", v(dst), c, x)
                    first = False
                elif c == 1.0:
                    e("v_add_f32", v(dst), v(dst), x)
                elif c == -1.0:
                    e("v_sub_f32", v(dst), v(dst), x)
                else:
                    e("v_fmac_f32", v(dst), c, x)

        # 7. update_info in unscaled variables (auxil.c:243-307): pri_res = |A xu - zu|_inf, dua_res = |q + P xu + A' yu|_inf
        # the step's objective weights, parked in AGPRs by phase A: the registers are nearly all taken here, so the
        # weights come through a 2-register LRU cache (columns visit them in runs: ~30 reads per step)
        wc_regs, wc_names = [g_() for _ in range(2)], [None, None]

        class _W:
            def __getitem__(self_, name):
                if name in wc_names:
                    k = wc_names.index(name)
                else:
                    k = 0
                    e("v_accvgpr_read_b32", v(wc_regs[k]), "a%d" % (A_W + WNAMES.index(name)))
                    e("s_nop", 0)
                    wc_names[k] = name
                wc_regs.append(wc_regs.pop(k))          # most recently used last
                wc_names.append(wc_names.pop(k))
                return wc_regs[-1]
        WV = _W()
        pri, nAx, nz, dua, nq, nAty, nPx, nan = [g_() for _ in range(8)]
        for r in (pri, nAx, nz, dua, nq, nAty, nPx, nan):
            e("v_mov_b32", v(r), 0)
        a1, a2 = g_(), g_()
        for i in range(nc):
            dot(a1, st.rows[i], XR, lambda p: st.col_of[p])
            res = a2
            if i < neq:
                if i in LR:
                    e("v_sub_f32", v(a2), v(a1), v(LR[i]))
                    e("v_max_f32", v(nz), v(nz), "|%s|" % v(LR[i]))
                elif i == CONST_ROW:
                    e("v_subrev_f32", v(a2), SF("dtg"), v(a1))
                    e("v_max_f32", v(nz), v(nz), "|%s|" % SF("dtg"))
                else:
                    res = a1
            else:
                e("v_sub_f32", v(a2), v(a1), v(zu3[i - neq]))
                e("v_max_f32", v(nz), v(nz), "|%s|" % v(zu3[i - neq]))
            e("v_max_f32", v(pri), v(pri), "|%s|" % v(res))
            e("v_max_f32", v(nAx), v(nAx), "|%s|" % v(a1))
            e("v_fma_f32", v(nan), 0, v(res), v(nan))
        ydes = RF[0:3] + RF[6:9]
        dpdes = RF[3:6]

        def qraw(dst, j):
            if j < N * NYv:
                k, i = divmod(j, NYv)
                wname = ("wpf" if k == N - 1 else "wpr") if i < 3 else "ws_"
                e("v_mul_f32", v(dst), "-" + v(WV[wname]), v(ydes[i]))
            else:
                k, i = divmod(j - N * NYv, NYv)
                e("v_mul_f32", v(dst), "-" + v(WV["wvf" if k == N - 1 else "wvr"]), v(dpdes[i]))
        a3, a4 = g_(), g_()
        for j in range(nx):
            dot(a1, range(s.A_p[j], s.A_p[j + 1]), YR, lambda p: s.A_i[p])
            e("v_mul_f32", v(a2), v(WV[st.weight_of(j)]), v(XR(j)))
            if j in st.qslot:
                qraw(a3, j)
                e("v_add_f32", v(a4), v(a3), v(a2))
                e("v_max_f32", v(nq), v(nq), "|%s|" % v(a3))
                e("v_add_f32", v(a4), v(a4), v(a1))
            else:
                e("v_add_f32", v(a4), v(a2), v(a1))
            e("v_max_f32", v(dua), v(dua), "|%s|" % v(a4))
            e("v_max_f32", v(nAty), v(nAty), "|%s|" % v(a1))
            e("v_max_f32", v(nPx), v(nPx), "|%s|" % v(a2))
            e("v_fma_f32", v(nan), 0, v(a4), v(nan))
        # 8. certificate scalars (auxil.c:362-512), unscaled: dyE = E delta_y, dxu = D delta_x
        ndy, lhs, nrm, ndx, qdx, nP, nAdx = [g_() for _ in range(7)]
        self.maxabs(ndy, [V_WZ + k for k in range(nc)])
        e("v_mov_b32", v(lhs), 0)
        for i in sorted(LR):
            e("v_fmac_f32", v(lhs), v(LR[i]), v(DYR(i)))
        e("v_fma_f32", v(lhs), SF("dtg"), v(DYR(CONST_ROW)), v(lhs))
        for k in range(N):
            e("v_max_f32", v(a1), 0, v(DYR(neq + k)))
            e("v_min_f32", v(a2), 0, v(DYR(neq + k)))
            e("v_fmac_f32", v(lhs), v(uT), v(a1))
            e("v_fmac_f32", v(lhs), v(lT), v(a2))
        self.maxabs(ndx, [V_W + k for k in range(nx)])
        e("v_mov_b32", v(qdx), 0)
        e("v_mov_b32", v(nP), 0)
        for j in range(nx):
            if j in st.qslot:
                qraw(a3, j)
                e("v_fmac_f32", v(qdx), v(a3), v(DXR(j)))
            e("v_mul_f32", v(a2), v(WV[st.weight_of(j)]), v(DXR(j)))
            e("v_max_f32", v(nP), v(nP), "|%s|" % v(a2))
        # |A' dyE|_inf and |A dxu|_inf only when some robot passes the cheap parts of a certificate (never, in practice)
        infv = g_()
        e("v_mov_b32", v(infv), f32bits(1e30))
        e("v_mov_b32", v(nrm), v(infv))
        e("v_mov_b32", v(nAdx), v(infv))
        lab_p, lab_d = self.label(), self.label()
        e("v_mul_f32", v(a1), sg(S_C["eps"]), v(ndy))
        e("v_cmp_lt_f32_e64", sp(S_M0), sg(S_C["eps"]), v(ndy))
        e("v_cmp_lt_f32_e64", sp(S_M1), v(lhs), "-" + v(a1))
        e("s_and_b64", "vcc", sp(S_M0), sp(S_M1))
        e("s_cbranch_vccz", lab_p + "f")
        e("v_mov_b32", v(nrm), 0)
        for j in range(nx):
            dot(a1, range(s.A_p[j], s.A_p[j + 1]), DYR, lambda p: s.A_i[p])
            e("v_max_f32", v(nrm), v(nrm), "|%s|" % v(a1))
        e("label", lab_p)
        e("v_mul_f32", v(a1), sg(S_C["eps"]), v(ndx))
        e("v_cmp_lt_f32_e64", sp(S_M0), sg(S_C["eps"]), v(ndx))
        e("v_cmp_lt_f32_e64", sp(S_M1), v(qdx), "-" + v(a1))
        e("s_and_b64", "vcc", sp(S_M0), sp(S_M1))
        e("s_cbranch_vccz", lab_d + "f")
        e("v_mov_b32", v(nAdx), 0)
        for i in range(nc):
            dot(a1, st.rows[i], DXR, lambda p: st.col_of[p])
            e("v_max_f32", v(nAdx), v(nAdx), "|%s|" % v(a1))
        e("label", lab_d)
        pool.free(dtT0, *s0dt, *Bdt, lT, uT, *[LR[i] for i in LR], *zu3, *wc_regs)
        # 9. check_termination, exact then approximate (osqp.c:524-573, auxil.c:684-789)
        relp, reld, stv = g_(), g_(), g_()
        e("v_max_f32", v(relp), v(nz), v(nAx))
        e("v_max3_f32", v(reld), v(nq), v(nAty), v(nPx))
        M = lambda k: sp(S_M0 + 2 * k)
        MP0, MD0, MP1, MD1, MT = sp(S_MP0), M(1), M(2), M(3), M(0)
        MBAD = sp(S_MBAD)

        def lt(mask, a_, b_):
            e("v_cmp_lt_f32_e64", mask, a_, b_)
        # tolerances: eps + eps * rel
        e("v_fma_f32", v(a1), sg(S_C["eps"]), v(relp), sg(S_C["eps"]))
        lt(MP0, v(pri), v(a1))
        e("v_fma_f32", v(a1), sg(S_C["eps"]), v(reld), sg(S_C["eps"]))
        lt(MD0, v(dua), v(a1))
        e("v_fma_f32", v(a1), sg(S_C["eps10"]), v(relp), sg(S_C["eps10"]))
        lt(MP1, v(pri), v(a1))
        e("v_fma_f32", v(a1), sg(S_C["eps10"]), v(reld), sg(S_C["eps10"]))
        lt(MD1, v(dua), v(a1))
        e("v_mov_b32", v(stv), -2)                                  # OSQP_MAX_ITER_REACHED

        def cert(epsname, okmask, code, kind):
            """stv <- code where !ok and the certificate holds at eps (MT, vcc are scratch masks)"""
            nrm_, n_, s_, third = (ndy, lhs, nrm, None) if kind == "p" else (ndx, qdx, nP, nAdx)
            e("v_mul_f32", v(a1), sg(S_C[epsname]), v(nrm_))        # eps * norm
            lt(MT, sg(S_C[epsname]), v(nrm_))                       # norm > eps
            e("s_andn2_b64", MT, MT, okmask)
            lt("vcc", v(n_), "-" + v(a1))
            e("s_and_b64", MT, MT, "vcc")
            lt("vcc", v(s_), v(a1))
            e("s_and_b64", MT, MT, "vcc")
            if third is not None:
                lt("vcc", v(a1), v(third))                           # nAdx > eps ndx kills the certificate
                e("s_andn2_b64", MT, MT, "vcc")
            e("v_cndmask_b32_e64", v(stv), v(stv), code, MT)
        cert("eps10", MD1, 4, "d")
        cert("eps10", MP1, 3, "p")
        e("s_and_b64", MT, MP1, MD1)
        e("v_cndmask_b32_e64", v(stv), v(stv), 2, MT)
        cert("eps", MD0, -4, "d")
        cert("eps", MP0, -3, "p")
        e("s_and_b64", MT, MP0, MD0)
        e("v_cndmask_b32_e64", v(stv), v(stv), 1, MT)
        # residual beyond OSQP_INFTY, or a NaN residual entry: OSQP_NON_CVX
        lt(MT, v(infv), v(pri))
        lt("vcc", v(infv), v(dua))
        e("s_or_b64", MT, MT, "vcc")
        e("v_cmp_u_f32", "vcc", v(nan), v(nan))
        e("s_or_b64", MT, MT, "vcc")
        e("v_cndmask_b32_e64", v(stv), v(stv), -7, MT)
        # !has_solution (auxil.c:527-565)
        e("v_cmp_eq_i32_e64", MBAD, -7, v(stv))
        for code in (-3, 3, -4, 4):
            e("v_cmp_eq_i32_e64", MT, code, v(stv))
            e("s_or_b64", MBAD, MBAD, MT)
        pool.free(ndy, lhs, nrm, ndx, qdx, nP, nAdx, relp, reld, nAx, nz, nq, nAty, nPx, nan, infv, a3, a4)
        # 10. store_solution + extraction (uprightmpc2.c:253-269)
        nanv = g_()
        e("v_mov_b32", v(nanv), f32bits(2143289344.0))
        sol = [XR(2 * N * NYv + i) for i in range(3)] + [XR(N * NYv + i) for i in range(6)]       # u0 u1 u2 | dy1[6]
        e("s_mov_b64", "vcc", MBAD)
        for r in sol:
            e("v_cndmask_b32", v(r), v(r), v(nanv), "vcc")
        pool.free(nanv)
        u0, u1, u2 = sol[0:3]
        dy1 = sol[3:9]
        OUT = [g_() for _ in range(9)]
        e("v_add_f32", v(OUT[0]), v(T0), v(u0))                     # T0 += u0; uquad[0] = T0
        e("v_mov_b32", v(OUT[1]), v(u1))
        e("v_mov_b32", v(OUT[2]), v(u2))
        # dq1des = (dy1[0:3], e3h R0' dy1[3:6]) = (dy1[0:3], -(R0' v)_y, (R0' v)_x, 0)
        e("v_mul_f32", v(a1), v(R0[0]), v(dy1[3]))
        e("v_mul_f32", v(t), v(R0[1]), v(dy1[4]))
        e("v_add_f32", v(a1), v(a1), v(t))
        e("v_mul_f32", v(t), v(R0[2]), v(dy1[5]))
        e("v_add_f32", v(a1), v(a1), v(t))                          # rx
        e("v_mul_f32", v(a2), v(R0[3]), v(dy1[3]))
        e("v_mul_f32", v(t), v(R0[4]), v(dy1[4]))
        e("v_add_f32", v(a2), v(a2), v(t))
        e("v_mul_f32", v(t), v(R0[5]), v(dy1[5]))
        e("v_add_f32", v(a2), v(a2), v(t))                          # ry
        for i in range(3):
            e("v_sub_f32", v(OUT[3 + i]), v(dy1[i]), v(dq0[i]))
        e("v_sub_f32", v(OUT[6]), "-" + v(a2), v(dq0[3]))
        e("v_sub_f32", v(OUT[7]), v(a1), v(dq0[4]))
        e("v_sub_f32", v(OUT[8]), 0, v(dq0[5]))
        for i in range(6):
            e("v_mul_f32", v(OUT[3 + i]), SF("idt"), v(OUT[3 + i]))
        self.store_rows("out", 0, OUT, voff)
        lab_s, lab_i, lab_c = self.label(), self.label(), self.label()
        e("s_cmp_eq_u64", sp(S_PTR["status"]), 0)
        e("s_cbranch_scc1", lab_s + "f")
        e("global_store_dword", "v0", v(stv), sp(S_PTR["status"]))
        e("label", lab_s)
        e("s_cmp_eq_u64", sp(S_PTR["info"]), 0)
        e("s_cbranch_scc1", lab_i + "f")
        self.store_rows("info", 0, [pri, dua], voff)
        e("label", lab_i)
        # cold start of the lanes without a solution (auxil.c:563): rewrite their x, y, z rows with zeros
        e("s_and_saveexec_b64", sp(S_M1), MBAD)
        e("s_cbranch_execz", lab_c + "f")
        e("v_mov_b32", v(a1), 0)
        self.rows_ptr(voff, 0)
        for r in range(nx + 2 * nc):
            e("global_store_dword", v(voff), v(a1), sp(S_PTR["ctrl"]))
            self.adv(voff)
        e("label", lab_c)
        e("s_mov_b64", "exec", sp(S_M1))
        pool.free(pri, dua, stv, a1, a2, t, T0, *RF)
        # free the loop's arrays
        pool.free_range(V_W, V_X - V_W, kill=False)
        pool.free_range(V_X, V_Y - V_X, kill=False)
        pool.free_range(V_Y, V_Z - V_Y, kill=False)
        # thrust accumulator of the next step: T0 + u0, or the WL step's actualT0 (uprightmpc2.c:215-216, 256-257)
        T0n = g_()
        e("v_mov_b32", v(T0n), v(OUT[0]))
        self.wl_step(OUT, R0, T0n, voff)
        self.store_rows("ctrl", nx + 2 * nc, [T0n], voff)
        pool.free(T0n)
        self.plant(ST, OUT, voff)
        pool.free(*OUT, *ST, voff)
        assert len(pool.free_) == VEND - VFIRST, "phase C leaked registers: %s" % sorted(set(range(VFIRST, VEND)) - pool.free_)

    # ---- MPC -> WL -> actualT0 (robobee_test_controllers.py:162-171, funapprox.c:118-165) ---------------
    def wl_step(self, OUT, R0, T0n, voff):
        """h0 = (Rb' (0, 0, mb g), 0), pdotdes = M0 accdes, (u4, w0) = wlConUpdate(h0, pdotdes); the NEXT controller
        step assembles with actualT0 = w0[2] / M0[2,2] when that is >= 0 (uprightmpc2.c:215-216). Operation order of
        umpc::wl_step (csrc/umpc_step.h; no fused multiply-adds there). The parameters are batch constants (struct
        umpc::WLDev in global memory): every lane loads the same word."""
        e, pool = self.e, self.pool
        g_ = pool.get
        lab_end = self.label()
        P = sp(S_PBLK)
        e("s_load_dwordx2", sp(S_PTR["wl"]), P, OFF["wl"])
        e("s_load_dwordx2", sp(S_PTR["wlu"]), P, OFF["wlu"])
        e("s_load_dwordx2", sp(S_PTR["wlw"]), P, OFF["wlw"])
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_cmp_eq_u64", sp(S_PTR["wl"]), 0)
        e("s_cbranch_scc1", lab_end + "f")
        # WLDev word offsets
        O_UMIN, O_UMAX, O_DUMAX, O_QW, O_A0, O_A1, O_A2, O_MD = 0, 4, 8, 12, 18, 24, 48, 144
        a1, a2, t = g_(), g_(), g_()
        zero = g_()
        e("v_mov_b32", v(zero), 0)

        def prm(dst, word):
            e("global_load_dword", v(dst), v(zero), sp(S_PTR["wl"]), "offset:%d" % (4 * word))
        u0 = [g_() for _ in range(4)]
        self.load_rows("wlu", 0, u0, voff)
        Md = [g_() for _ in range(6)]
        for i in range(6):
            prm(Md[i], O_MD + i)
        e("s_waitcnt", "vmcnt(0)")
        h0 = [g_() for _ in range(3)]
        pd = [g_() for _ in range(6)]
        for c in range(3):
            e("v_mul_f32", v(h0[c]), SF_("mbg"), v(R0[2 + 3 * c]))
        for i in range(6):
            e("v_mul_f32", v(pd[i]), v(Md[i]), v(OUT[3 + i]))
        w0 = [g_() for _ in range(6)]
        a0v = [g_() for _ in range(6)]
        A1 = [[g_() for _ in range(4)] for _ in range(6)]
        PA = [g_() for _ in range(21)]          # a0, a1[4], A2[16] of one output
        vout = [g_() for _ in range(4)]
        dot, quad = g_(), g_()
        for i in range(6):
            prm(PA[0], O_A0 + i)
            for l in range(4):
                prm(PA[1 + l], O_A1 + 4 * i + l)
            for k in range(16):
                prm(PA[5 + k], O_A2 + 16 * i + k)
            e("s_waitcnt", "vmcnt(0)")
            # dot = sum_l u0[l] a1[l]  (accumulated from 0 like the reference's matMult)
            e("v_mul_f32", v(dot), v(u0[0]), v(PA[1]))
            for l in range(1, 4):
                e("v_mul_f32", v(t), v(u0[l]), v(PA[1 + l]))
                e("v_add_f32", v(dot), v(dot), v(t))
            for r in range(4):
                e("v_mul_f32", v(vout[r]), v(PA[5 + r]), v(u0[0]))
                for l in range(1, 4):
                    e("v_mul_f32", v(t), v(PA[5 + r + 4 * l]), v(u0[l]))
                    e("v_add_f32", v(vout[r]), v(vout[r]), v(t))
            e("v_mul_f32", v(quad), v(u0[0]), v(vout[0]))
            for l in range(1, 4):
                e("v_mul_f32", v(t), v(u0[l]), v(vout[l]))
                e("v_add_f32", v(quad), v(quad), v(t))
            e("v_add_f32", v(a1), v(PA[0]), v(dot))
            e("v_mul_f32", v(a2), 0.5, v(quad))
            e("v_add_f32", v(w0[i]), v(a1), v(a2))
            if i < 3:
                e("v_sub_f32", v(a0v[i]), v(w0[i]), v(h0[i]))
            else:
                e("v_mov_b32", v(a0v[i]), v(w0[i]))                 # h0[3..5] = 0
            e("v_sub_f32", v(a0v[i]), v(a0v[i]), v(pd[i]))
            for j in range(4):
                e("v_add_f32", v(A1[i][j]), v(PA[1 + j]), v(vout[j]))
        # one clipped gradient step (funapprox.c:139-164)
        lim = [g_() for _ in range(12)]                             # umin[4] umax[4] dumax[4]
        qw = [g_() for _ in range(6)]
        for k in range(12):
            prm(lim[k], O_UMIN + k)
        for i in range(6):
            prm(qw[i], O_QW + i)
        e("s_waitcnt", "vmcnt(0)")
        for i in range(6):
            e("v_mul_f32", v(a0v[i]), v(qw[i]), v(a0v[i]))          # Qw[i] * a0v[i]
        Lb, Ub, acc = g_(), g_(), g_()
        for j in range(4):
            e("v_mul_f32", v(acc), v(A1[0][j]), v(a0v[0]))
            for i in range(1, 6):
                e("v_mul_f32", v(t), v(A1[i][j]), v(a0v[i]))
                e("v_add_f32", v(acc), v(acc), v(t))
            e("v_mul_f32", v(acc), -1000.0, v(acc))
            e("v_mul_f32", v(Lb), -1.0, v(lim[8 + j]))
            e("v_mov_b32", v(Ub), v(lim[8 + j]))
            # if (u0 < umin) Lb = 0; else if (u0 > umax) Ub = 0
            e("v_cmp_lt_f32_e64", sp(S_M3), v(u0[j]), v(lim[j]))
            e("v_cmp_gt_f32", "vcc", v(u0[j]), v(lim[4 + j]))
            e("s_andn2_b64", "vcc", "vcc", sp(S_M3))
            e("v_cndmask_b32_e64", v(Lb), v(Lb), 0, sp(S_M3))
            e("v_cndmask_b32", v(Ub), v(Ub), v(zero), "vcc")
            # if (d < Lb) d = Lb; else if (d > Ub) d = Ub
            e("v_cmp_lt_f32_e64", sp(S_M3), v(acc), v(Lb))
            e("v_cmp_gt_f32", "vcc", v(acc), v(Ub))
            e("s_andn2_b64", "vcc", "vcc", sp(S_M3))
            e("v_cndmask_b32", v(acc), v(acc), v(Ub), "vcc")
            e("v_cndmask_b32_e64", v(acc), v(acc), v(Lb), sp(S_M3))
            e("v_add_f32", v(u0[j]), v(u0[j]), v(acc))
        self.store_rows("wlu", 0, u0, voff)
        lab_w = self.label()
        e("s_cmp_eq_u64", sp(S_PTR["wlw"]), 0)
        e("s_cbranch_scc1", lab_w + "f")
        self.store_rows("wlw", 0, w0, voff)
        e("label", lab_w)
        # actualT0 of the next step: w0[2] / M0[2,2], taken when >= 0
        self.rcp_nr(a1, Md[2], t)
        e("v_mul_f32", v(a1), v(w0[2]), v(a1))
        e("v_cmp_le_f32", "vcc", 0, v(a1))
        e("v_cndmask_b32", v(T0n), v(T0n), v(a1), "vcc")
        pool.free(zero, *u0, *Md, *h0, *pd, *w0, *a0v, *[r for row in A1 for r in row], *PA, *vout, dot, quad, *lim, *qw,
                  Lb, Ub, acc, a1, a2, t)
        e("label", lab_end)

    # ---- plant: nsub RK4 substeps of template/genqp.py:24-30 (build-defined integrator), statistics -----------------
    def plant(self, ST, OUT, voff):
        e, pool = self.e, self.pool
        SF = lambda n: sg(S_F[n])
        g_ = pool.get
        lab_end = self.label()
        e("s_cmp_lt_i32", sg(S_INT["nsub"]), 1)
        e("s_cbranch_scc1", lab_end + "f")
        Y0, YS, KK, AC = pool.getn(18), pool.getn(18), pool.getn(18), pool.getn(18)
        for i in range(18):
            e("v_mov_b32", v(Y0 + i), v(ST[i]))
        Th, u1, u2, einc = g_(), g_(), g_(), g_()
        Ib = [g_() for _ in range(3)]
        Ibi = [g_() for _ in range(3)]
        serr, seff, t = g_(), g_(), g_()
        lab_g, lab_g2, lab_i, lab_i2, lab_s, lab_s2 = [self.label() for _ in range(6)]
        # thrust gain (Monte-Carlo mass sweep), inertia, statistics
        e("s_cmp_eq_u64", sp(S_PTR["gain"]), 0)
        e("s_cbranch_scc1", lab_g + "f")
        e("global_load_dword", v(t), "v0", sp(S_PTR["gain"]))
        e("s_waitcnt", "vmcnt(0)")
        e("v_mul_f32", v(Th), v(t), v(OUT[0]))
        e("s_branch", lab_g2 + "f")
        e("label", lab_g)
        e("v_mov_b32", v(Th), v(OUT[0]))
        e("label", lab_g2)
        e("s_cmp_eq_u64", sp(S_PTR["Ib"]), 0)
        e("s_cbranch_scc1", lab_i + "f")
        self.load_rows("Ib", 0, Ib, voff)
        e("s_waitcnt", "vmcnt(0)")
        for k in range(3):
            self.rcp_nr(Ibi[k], Ib[k], t)
        e("s_branch", lab_i2 + "f")
        e("label", lab_i)
        for k in range(3):
            e("v_mov_b32", v(Ib[k]), SF("Ib%d" % k))
            e("v_mov_b32", v(Ibi[k]), SF("Ibi%d" % k))
        e("label", lab_i2)
        e("v_mov_b32", v(serr), 0)
        e("v_mov_b32", v(seff), 0)
        e("s_cmp_eq_u64", sp(S_PTR["stats"]), 0)
        e("s_cbranch_scc1", lab_s + "f")
        self.load_rows("stats", 0, [serr, seff], voff)
        e("s_waitcnt", "vmcnt(0)")
        e("label", lab_s)
        # the harness clips what the PLANT sees (template/uprightmpc2.py:148-149)
        e("v_med3_f32", v(u1), v(OUT[1]), "-" + SF("taulim"), SF("taulim"))
        e("v_med3_f32", v(u2), v(OUT[2]), "-" + SF("taulim"), SF("taulim"))
        e("v_mul_f32", v(einc), v(u1), v(u1))
        e("v_fmac_f32", v(einc), v(u2), v(u2))
        two = g_()
        e("v_mov_b32", v(two), 2.0)
        a, b = g_(), g_()

        def vf(src, dst):
            """dst[0:18] = f(src): dp = v, dR = R skew(w), dv = Th R e3 - g e3, dw = Ib^-1 (tau - w x Ib w)"""
            R = lambda k: src + 3 + k
            vx, vy, vz = src + 12, src + 13, src + 14
            wx, wy, wz = src + 15, src + 16, src + 17
            for i in range(3):
                e("v_mov_b32", v(dst + i), v(src + 12 + i))
            for r in range(3):
                e("v_mul_f32", v(a), v(R(r + 6)), v(wy))
                e("v_fma_f32", v(dst + 3 + r), v(R(r + 3)), v(wz), "-" + v(a))
                e("v_mul_f32", v(b), v(R(r)), v(wz))
                e("v_fma_f32", v(dst + 6 + r), v(R(r + 6)), v(wx), "-" + v(b))
                e("v_mul_f32", v(a), v(R(r + 3)), v(wx))
                e("v_fma_f32", v(dst + 9 + r), v(R(r)), v(wy), "-" + v(a))
            e("v_mul_f32", v(dst + 12), v(Th), v(R(6)))
            e("v_mul_f32", v(dst + 13), v(Th), v(R(7)))
            e("v_mul_f32", v(dst + 14), v(Th), v(R(8)))
            e("v_subrev_f32", v(dst + 14), SF("gpl"), v(dst + 14))
            hx, hy, hz = dst + 15, dst + 16, dst + 17           # staged in place
            e("v_mul_f32", v(hx), v(Ib[0]), v(wx))
            e("v_mul_f32", v(hy), v(Ib[1]), v(wy))
            e("v_mul_f32", v(hz), v(Ib[2]), v(wz))
            e("v_mul_f32", v(a), v(wz), v(hy))
            e("v_fma_f32", v(a), v(wy), v(hz), "-" + v(a))       # cx
            e("v_mul_f32", v(b), v(wx), v(hz))
            e("v_fma_f32", v(b), v(wz), v(hx), "-" + v(b))       # cy
            e("v_mul_f32", v(hz), v(wy), v(hx))
            e("v_fma_f32", v(hz), v(wx), v(hy), "-" + v(hz))     # cz (hy still intact)
            e("v_sub_f32", v(a), v(u1), v(a))
            e("v_mul_f32", v(dst + 15), v(a), v(Ibi[0]))
            e("v_sub_f32", v(b), v(u2), v(b))
            e("v_mul_f32", v(dst + 16), v(b), v(Ibi[1]))
            e("v_mul_f32", v(dst + 17), "-" + v(hz), v(Ibi[2]))
        lab_rk4, lab_done = self.label(), self.label()
        e("s_cmp_lg_u32", sg(S_INT["plant"]), 0)
        e("s_cbranch_scc1", lab_rk4 + "f")
        # ---- mode 0: the reference's own step (template/genqp.py:32-41): p += h v, R <- R expm(skew(w) h) (Rodrigues
        # form of the scipy expm), dq += h ddq, all from the OLD state. sin(th)/th and (1 - cos th)/th^2 as 8-term
        # series in t = th^2 (the C++ / oracle statement switches to sin / cos at t >= 1e-2; the series agree with it to
        # fp32 round-off for |w| h < 1.5 rad per substep, far beyond any physical spin rate).
        e("s_mov_b32", sg(S_SUB), sg(S_INT["nsub"]))
        top0 = self.label()
        e("label", top0)
        R = lambda k: Y0 + 3 + k
        wx, wy, wz = Y0 + 15, Y0 + 16, Y0 + 17
        dd = [YS + k for k in range(6)]                      # ddq
        e("v_mul_f32", v(dd[0]), v(Th), v(R(6)))
        e("v_mul_f32", v(dd[1]), v(Th), v(R(7)))
        e("v_mul_f32", v(dd[2]), v(Th), v(R(8)))
        e("v_subrev_f32", v(dd[2]), SF("gpl"), v(dd[2]))
        hx, hy, hz = YS + 6, YS + 7, YS + 8
        e("v_mul_f32", v(hx), v(Ib[0]), v(wx))
        e("v_mul_f32", v(hy), v(Ib[1]), v(wy))
        e("v_mul_f32", v(hz), v(Ib[2]), v(wz))
        e("v_mul_f32", v(a), v(wz), v(hy))
        e("v_fma_f32", v(a), v(wy), v(hz), "-" + v(a))
        e("v_mul_f32", v(b), v(wx), v(hz))
        e("v_fma_f32", v(b), v(wz), v(hx), "-" + v(b))
        e("v_mul_f32", v(hz), v(wy), v(hx))
        e("v_fma_f32", v(hz), v(wx), v(hy), "-" + v(hz))
        e("v_sub_f32", v(a), v(u1), v(a))
        e("v_mul_f32", v(dd[3]), v(a), v(Ibi[0]))
        e("v_sub_f32", v(b), v(u2), v(b))
        e("v_mul_f32", v(dd[4]), v(b), v(Ibi[1]))
        e("v_mul_f32", v(dd[5]), "-" + v(hz), v(Ibi[2]))
        for i in range(3):
            e("v_fmac_f32", v(Y0 + i), SF("h"), v(Y0 + 12 + i))            # p += h v (old v)
        ax, ay, az, tt, ca, cb = [KK + k for k in range(6)]
        e("v_mul_f32", v(ax), SF("h"), v(wx))
        e("v_mul_f32", v(ay), SF("h"), v(wy))
        e("v_mul_f32", v(az), SF("h"), v(wz))
        e("v_mul_f32", v(tt), v(ax), v(ax))
        e("v_fmac_f32", v(tt), v(ay), v(ay))
        e("v_fmac_f32", v(tt), v(az), v(az))
        import math
        ca_c = [(-1.0) ** k / math.factorial(2 * k + 1) for k in range(8)]
        cb_c = [(-1.0) ** k / math.factorial(2 * k + 2) for k in range(8)]
        e("v_mov_b32", v(ca), f32bits(ca_c[7]))
        e("v_mov_b32", v(cb), f32bits(cb_c[7]))
        for k in range(6, -1, -1):
            e("v_fmaak_f32", v(ca), v(ca), v(tt), f32bits(ca_c[k]))
            e("v_fmaak_f32", v(cb), v(cb), v(tt), f32bits(cb_c[k]))
        # E = I + a K + b K^2
        E = [[KK + 6 + 3 * r + c for c in range(3)] for r in range(3)]
        xx, yy, zz, xy, xz, yz = [AC + k for k in range(6)]
        e("v_mul_f32", v(xx), v(ax), v(ax))
        e("v_mul_f32", v(yy), v(ay), v(ay))
        e("v_mul_f32", v(zz), v(az), v(az))
        e("v_mul_f32", v(xy), v(ax), v(ay))
        e("v_mul_f32", v(xz), v(ax), v(az))
        e("v_mul_f32", v(yz), v(ay), v(az))
        e("v_mul_f32", v(xy), v(cb), v(xy))
        e("v_mul_f32", v(xz), v(cb), v(xz))
        e("v_mul_f32", v(yz), v(cb), v(yz))
        t1 = AC + 6
        e("v_add_f32", v(t1), v(yy), v(zz))
        e("v_fma_f32", v(E[0][0]), "-" + v(cb), v(t1), 1.0)
        e("v_add_f32", v(t1), v(xx), v(zz))
        e("v_fma_f32", v(E[1][1]), "-" + v(cb), v(t1), 1.0)
        e("v_add_f32", v(t1), v(xx), v(yy))
        e("v_fma_f32", v(E[2][2]), "-" + v(cb), v(t1), 1.0)
        e("v_fma_f32", v(E[0][1]), "-" + v(ca), v(az), v(xy))
        e("v_fma_f32", v(E[1][0]), v(ca), v(az), v(xy))
        e("v_fma_f32", v(E[0][2]), v(ca), v(ay), v(xz))
        e("v_fma_f32", v(E[2][0]), "-" + v(ca), v(ay), v(xz))
        e("v_fma_f32", v(E[1][2]), "-" + v(ca), v(ax), v(yz))
        e("v_fma_f32", v(E[2][1]), v(ca), v(ax), v(yz))
        Rn = [AC + 7 + k for k in range(9)]
        for c in range(3):
            for r in range(3):
                e("v_mul_f32", v(Rn[r + 3 * c]), v(R(r)), v(E[0][c]))
                e("v_fmac_f32", v(Rn[r + 3 * c]), v(R(r + 3)), v(E[1][c]))
                e("v_fmac_f32", v(Rn[r + 3 * c]), v(R(r + 6)), v(E[2][c]))
        for k in range(9):
            e("v_mov_b32", v(R(k)), v(Rn[k]))
        for i in range(6):
            e("v_fmac_f32", v(Y0 + 12 + i), SF("h"), v(dd[i]))              # dq += h ddq
        e("v_fmac_f32", v(serr), v(Y0), v(Y0))
        e("v_fmac_f32", v(serr), v(Y0 + 1), v(Y0 + 1))
        e("v_fmac_f32", v(serr), v(Y0 + 2), v(Y0 + 2))
        e("v_add_f32", v(seff), v(seff), v(einc))
        e("s_sub_i32", sg(S_SUB), sg(S_SUB), 1)
        e("s_cmp_gt_i32", sg(S_SUB), 0)
        e("s_cbranch_scc1", top0 + "b")
        e("s_branch", lab_done + "f")
        # ---- mode 1: classical RK4 on the same vector field (build-defined)
        e("label", lab_rk4)
        if self.quad:
            self._rk4_quad(Y0, YS, KK, AC, Th, u1, u2, einc, Ib, Ibi, serr, seff, a, b, two)
            e("s_branch", lab_done + "f")
        e("s_mov_b32", sg(S_SUB), sg(S_INT["nsub"]))
        top = self.label()
        e("label", top)
        vf(Y0, AC)                                                             # k1
        for k in range(0, 18, 2):
            pk(e, "v_pk_fma_f32", YS + k, [PS(S_F["hh"]), P2(AC + k), P2(Y0 + k)])
        vf(YS, KK)                                                             # k2
        for k in range(0, 18, 2):
            pk(e, "v_pk_fma_f32", AC + k, [PB(two), P2(KK + k), P2(AC + k)])
            pk(e, "v_pk_fma_f32", YS + k, [PS(S_F["hh"]), P2(KK + k), P2(Y0 + k)])
        vf(YS, KK)                                                             # k3
        for k in range(0, 18, 2):
            pk(e, "v_pk_fma_f32", AC + k, [PB(two), P2(KK + k), P2(AC + k)])
            pk(e, "v_pk_fma_f32", YS + k, [PS(S_F["h"]), P2(KK + k), P2(Y0 + k)])
        vf(YS, KK)                                                             # k4
        for k in range(0, 18, 2):
            pk(e, "v_pk_add_f32", AC + k, [P2(AC + k), P2(KK + k)])
        for k in range(0, 18, 2):
            pk(e, "v_pk_fma_f32", Y0 + k, [PS(S_F["h6"]), P2(AC + k), P2(Y0 + k)])
        e("v_fmac_f32", v(serr), v(Y0), v(Y0))
        e("v_fmac_f32", v(serr), v(Y0 + 1), v(Y0 + 1))
        e("v_fmac_f32", v(serr), v(Y0 + 2), v(Y0 + 2))
        e("v_add_f32", v(seff), v(seff), v(einc))
        e("s_sub_i32", sg(S_SUB), sg(S_SUB), 1)
        e("s_cmp_gt_i32", sg(S_SUB), 0)
        e("s_cbranch_scc1", top + "b")
        e("label", lab_done)
        self.store_rows("state", 0, [Y0 + i for i in range(18)], voff)
        e("s_cmp_eq_u64", sp(S_PTR["stats"]), 0)
        e("s_cbranch_scc1", lab_s2 + "f")
        self.store_rows("stats", 0, [serr, seff], voff)
        e("label", lab_s2)
        pool.free(Th, u1, u2, einc, *Ib, *Ibi, serr, seff, t, two, a, b)
        for base in (Y0, YS, KK, AC):
            pool.free_range(base, 18)
        e("label", lab_end)


    def _rk4_quad(self, Y0, YS, KK, AC, Th, u1, u2, einc, Ib, Ibi, serr, seff, a, b, two):
        """The RK4 substeps on the lane quad (the quad form of the stream): lane r integrates ROW r of the state -- p_r, v_r and
        row r of R, whose derivative R skew(w) needs nothing but that row and w -- and every lane carries w (its 13-instruction
        vector field runs replicated). 8 words per lane instead of 18: 123 instructions per substep instead of 230. Every
        element goes through the one-lane sequence of operations, so the result equals the lane form's bit for bit; the
        position-error sum takes the three p from the lanes in the one-lane order."""
        from . import asmquad
        e = self.e
        SF = lambda n: sg(S_F[n])
        masks = (asmquad.S_L0, asmquad.S_L1, asmquad.S_L2)
        S_L3, S_L03 = 98, 36                       # lane 3 alone (scratch), lanes 0 and 3 (lane 3 mirrors lane 0: finite, unused)
        S_EX = 62                                  # s[62:63]: the entry EXEC (S_MBAD is dead here)
        Q0, QS, QK, QA = YS, KK, AC, KK + 8         # 8-word vectors: [p_r, v_r, R_r0, R_r1, R_r2, wx, wy, wz]
        G, PA = AC + 8, AC + 10                     # g in lane 2 only; the three p of the robot (AC + 10 .. + 12)
        keep = sorted(set(range(Y0, Y0 + 18)) | {serr, seff})
        e("quad_begin", "plant")
        e("s_mov_b64", sp(S_EX), "exec")
        for ln, m in enumerate(masks + (S_L3,)):
            e("s_mov_b32", sg(m), 0x11111111 << ln)
            e("s_mov_b32", sg(m + 1), 0x11111111 << ln)
            e("s_and_b64", sp(m), sp(m), sp(S_EX))
        e("s_or_b64", sp(S_L03), sp(masks[0]), sp(S_L3))
        e("v_mov_b32", v(G), 0)
        for r, m in ((0, S_L03), (1, masks[1]), (2, masks[2])):
            e("s_mov_b64", "exec", sp(m))
            e("v_mov_b32", v(Q0), v(Y0 + r))
            e("v_mov_b32", v(Q0 + 1), v(Y0 + 12 + r))
            for c in range(3):
                e("v_mov_b32", v(Q0 + 2 + c), v(Y0 + 3 + r + 3 * c))
            if r == 2:
                e("v_mov_b32", v(G), SF("gpl"))
        e("s_mov_b64", "exec", sp(S_EX))
        for i in range(3):
            e("v_mov_b32", v(Q0 + 5 + i), v(Y0 + 15 + i))
        e("s_nop", 4)

        def vf(src, dst):
            p_, v_, r0, r1, r2, wx, wy, wz = (src + k for k in range(8))
            e("v_mov_b32", v(dst), v(v_))
            e("v_mul_f32", v(dst + 1), v(Th), v(r2))
            e("v_sub_f32", v(dst + 1), v(dst + 1), v(G))
            e("v_mul_f32", v(a), v(r2), v(wy))
            e("v_fma_f32", v(dst + 2), v(r1), v(wz), "-" + v(a))
            e("v_mul_f32", v(b), v(r0), v(wz))
            e("v_fma_f32", v(dst + 3), v(r2), v(wx), "-" + v(b))
            e("v_mul_f32", v(a), v(r1), v(wx))
            e("v_fma_f32", v(dst + 4), v(r0), v(wy), "-" + v(a))
            hx, hy, hz = dst + 5, dst + 6, dst + 7
            e("v_mul_f32", v(hx), v(Ib[0]), v(wx))
            e("v_mul_f32", v(hy), v(Ib[1]), v(wy))
            e("v_mul_f32", v(hz), v(Ib[2]), v(wz))
            e("v_mul_f32", v(a), v(wz), v(hy))
            e("v_fma_f32", v(a), v(wy), v(hz), "-" + v(a))
            e("v_mul_f32", v(b), v(wx), v(hz))
            e("v_fma_f32", v(b), v(wz), v(hx), "-" + v(b))
            e("v_mul_f32", v(hz), v(wy), v(hx))
            e("v_fma_f32", v(hz), v(wx), v(hy), "-" + v(hz))
            e("v_sub_f32", v(a), v(u1), v(a))
            e("v_mul_f32", v(dst + 5), v(a), v(Ibi[0]))
            e("v_sub_f32", v(b), v(u2), v(b))
            e("v_mul_f32", v(dst + 6), v(b), v(Ibi[1]))
            e("v_mul_f32", v(dst + 7), "-" + v(hz), v(Ibi[2]))
        e("s_mov_b32", sg(S_SUB), sg(S_INT["nsub"]))
        top = self.label()
        e("label", top)
        vf(Q0, QA)
        for k in range(0, 8, 2):
            pk(e, "v_pk_fma_f32", QS + k, [PS(S_F["hh"]), P2(QA + k), P2(Q0 + k)])
        vf(QS, QK)
        for k in range(0, 8, 2):
            pk(e, "v_pk_fma_f32", QA + k, [PB(two), P2(QK + k), P2(QA + k)])
            pk(e, "v_pk_fma_f32", QS + k, [PS(S_F["hh"]), P2(QK + k), P2(Q0 + k)])
        vf(QS, QK)
        for k in range(0, 8, 2):
            pk(e, "v_pk_fma_f32", QA + k, [PB(two), P2(QK + k), P2(QA + k)])
            pk(e, "v_pk_fma_f32", QS + k, [PS(S_F["h"]), P2(QK + k), P2(Q0 + k)])
        vf(QS, QK)
        for k in range(0, 8, 2):
            pk(e, "v_pk_add_f32", QA + k, [P2(QA + k), P2(QK + k)])
        for k in range(0, 8, 2):
            pk(e, "v_pk_fma_f32", Q0 + k, [PS(S_F["h6"]), P2(QA + k), P2(Q0 + k)])
        e("s_nop", 1)
        for r in range(3):
            e("v_mov_b32_dpp", v(PA + r), v(Q0), asmquad.qperm([r] * 4))
        e("s_nop", 1)
        for r in range(3):
            e("v_fmac_f32", v(serr), v(PA + r), v(PA + r))
        e("v_add_f32", v(seff), v(seff), v(einc))
        e("s_sub_i32", sg(S_SUB), sg(S_SUB), 1)
        e("s_cmp_gt_i32", sg(S_SUB), 0)
        e("s_cbranch_scc1", top + "b")
        # back to the one-lane state vector, in all four lanes
        e("s_nop", 1)
        for r in range(3):
            e("v_mov_b32_dpp", v(Y0 + r), v(Q0), asmquad.qperm([r] * 4))
            e("v_mov_b32_dpp", v(Y0 + 12 + r), v(Q0 + 1), asmquad.qperm([r] * 4))
            for c in range(3):
                e("v_mov_b32_dpp", v(Y0 + 3 + r + 3 * c), v(Q0 + 2 + c), asmquad.qperm([r] * 4))
        for i in range(3):
            e("v_mov_b32", v(Y0 + 15 + i), v(Q0 + 5 + i))
        e("s_nop", 1)
        e("quad_end", tuple(keep))

    # ---- the whole kernel body --------------------------------------------------------------------------------
    def program(self):
        e = self.e
        self.prologue()
        top = self.label()
        e("label", top)
        try:
            asmgen.NRING = OPT_RING
            self.phase_a()      # factor() switches asmgen's extra-VGPR L homes on (module state) ...
            self.admm()
        finally:
            asmgen.XV_COUNT = 0  # ... for this stream only: asmgen.program() of the C++ kernel must not see them
            asmgen.NRING = 4
        self.phase_c()
        e("s_add_i32", sg(S_STEP), sg(S_STEP), 1)
        e("s_cmp_lt_i32", sg(S_STEP), sg(S_INT["K"]))
        e("s_cbranch_scc1", top + "b")
        e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")
        # completion word for a host that polls instead of synchronising the stream (the B = 1 drop-in): every store of
        # the wave has been acknowledged (vmcnt 0); write back L2 at system scope, then the flag itself at system scope
        lab = self.label()
        e("s_load_dwordx2", sp(S_M0), sp(S_PBLK), OFF["done"])
        e("s_load_dword", sg(S_M1), sp(S_PBLK), OFF["seq"])
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_cmp_eq_u64", sp(S_M0), 0)
        e("s_cbranch_scc1", lab + "f")
        e("buffer_wbl2", "sc0 sc1")
        e("s_waitcnt", "vmcnt(0)")
        a, b = self.pool.get(), self.pool.get()
        e("v_mov_b32", v(a), 0)
        e("v_mov_b32", v(b), sg(S_M1))
        e("global_store_dword", v(a), v(b), sp(S_M0), "sc0 sc1")
        e("s_waitcnt", "vmcnt(0)")
        self.pool.free(a, b)
        e("label", lab)
        return e.ins


# ----------------------------------------------------------------------------------------------------------
# Text output
# ----------------------------------------------------------------------------------------------------------
def fmt(t):
    m = t[0]
    if m == "label":
        return "%s:" % t[1]
    mods = ""
    if isinstance(t[-1], dict):
        d = t[-1]
        t = t[:-1]
        keys = ("op_sel",) if m == "v_pk_mov_b32" else ("op_sel", "op_sel_hi", "neg_lo", "neg_hi")
        mods = " " + " ".join("%s:[%s]" % (k, ",".join(map(str, d[k]))) for k in keys)

    def a_(x):
        if isinstance(x, float):
            return repr(x)
        if isinstance(x, int):
            return ("0x%x" % x) if x > 64 else str(x)
        return str(x)
    a = [a_(x) for x in t[1:]]
    if m in ("ds_read_b128", "ds_write_b128", "ds_write_b32", "ds_read_b32"):
        return "%s %s, %s offset:%s" % (m, a[0], a[1], t[3])
    if m.startswith("s_load_"):
        return "%s %s, %s, %s%s" % (m, a[0], a[1], ("0x%x" % t[3]) if isinstance(t[3], int) else t[3],
                                    (" " + t[4]) if len(t) > 4 else "")
    if m.startswith("global_") and isinstance(t[-1], str) and (t[-1].startswith("offset:") or t[-1].startswith("sc") or t[-1].startswith("nt")):
        return "%s %s %s" % (m, ", ".join(a[:-1]), t[-1])
    if m == "buffer_wbl2":
        return "buffer_wbl2 %s" % t[1]
    if m == "s_waitcnt":
        return "s_waitcnt " + " ".join(a)
    if m.endswith("_dpp"):          # the DPP control follows the operands without a comma
        return "%s %s %s" % (m, ", ".join(a[:-1]), t[-1])
    return "%s %s%s" % (m, ", ".join(a), mods)


PSEUDO = ("kill", "quad_begin", "quad_end")      # markers for the CPU interpreters, not instructions


def write(path=None, N=3, perm=None, quad=False):
    """quad: the one-robot-per-lane-quad stream (asmquad.py) -> csrc/umpc_step_asm_quad.h, macro UMPC_STEP_ASM_QUAD; it
    shares struct StepParams with the one-lane header, which must be included first."""
    path = path or os.path.join(HERE, "csrc", "umpc_step_asm_quad.h" if quad else "umpc_step_asm.h")
    g = StepGen(N, perm, quad=quad)
    ins = [t for t in g.program() if t[0] not in PSEUDO]
    used_s = sorted(set(range(4, 102)))
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in range(2, 256)] + ['"a%d"' % i for i in range(256)] + \
           ['"s%d"' % i for i in used_s]
    if quad:
        out = ["// GENERATED by robobee3d_amd/asmstep.py (quad=True) + asmquad.py -- do not edit.", asmgen.switch_banner(),
               "// The all-assembly fp32 step kernel body, ONE ROBOT PER LANE QUAD (16 robots per wavefront): %d instructions."
               % len(ins),
               "// Every lane of a quad gets the same v0 (= 4 * robot) and its own LDS slice; include umpc_step_asm.h first (StepParams).",
               "#pragma once",
               "#define UMPC_STEP_ASM_QUAD(voff, ldsaddr, params) asm volatile( \\"]
    else:
        out = ["// GENERATED by robobee3d_amd/asmstep.py -- do not edit.", asmgen.switch_banner(),
               "// The all-assembly fp32 step kernel body: %d instructions (K closed-loop steps of one wavefront)." % len(ins),
               "#pragma once", "#include <stdint.h>", "namespace umpcasm {",
               "// parameter block read by the kernel with s_load (byte offsets are part of the generated code)",
               "struct StepParams {"]
        for n in PTRS:
            out.append("  const void *%s;" % n)
        for n in INTS:
            out.append("  int32_t %s;" % n)
        for n in FLOATS:
            out.append("  float %s;" % n)
        out += ["};", "static_assert(sizeof(StepParams) == %d, \"StepParams layout\");" % ((PARAM_BYTES + 7) // 8 * 8),
                "constexpr int STEP_LDS_BYTES_PER_LANE = %d;" % (NLDS * 4), "}  // namespace umpcasm",
                "// inputs: v0 = 4 * robot, v1 = lane LDS address, s[4:5] = &StepParams (kernarg)",
                "#define UMPC_STEP_ASM(voff, ldsaddr, params) asm volatile( \\"]
    for t in ins:
        out.append('  "%s\\n" \\' % fmt(t))
    out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{s[4:5]}"(params) \\')
    out.append("  : " + ", ".join(clob) + ")")
    txt = "\n".join(out) + "\n"
    old = open(path).read() if os.path.exists(path) else None
    if old != txt:
        with open(path, "w") as fh:
            fh.write(txt)
    return path, len(ins)


# ----------------------------------------------------------------------------------------------------------
# CPU interpreter of the emitted stream (one lane), for tests/test_asm_step.py
# ----------------------------------------------------------------------------------------------------------
def simulate(ins, arrays, ints, floats, max_exec=3000000, ptr_xform=None):
    """arrays: name -> float32 / int32 numpy vector indexed by ROW (one robot), or None for a null pointer; ints /
    floats: the StepParams scalars (ints without `stride`). Runs the whole kernel; arrays are updated in place.
    Returns the executed instruction count (pseudo-instructions excluded)."""
    import numpy as np
    f32, u32 = np.float32, np.uint32
    STRIDE = 4096
    V = np.zeros(256, u32)
    A = np.zeros(256, u32)
    S = {}
    lds = np.zeros(NLDS, u32)
    POISON = u32(0x7fc0dead)
    V[:] = POISON
    A[:] = POISON
    names = list(PTRS)
    # simulated device addresses with bit 31 of the low word set (like real ones): a half that gets sign-extended on its
    # way into an address lands outside every array and is caught below
    base_of = {n: 0x00007F0080000000 + (k << 36) for k, n in enumerate(names)}
    ptr_xform = ptr_xform or (lambda a: a)
    blob = bytearray(PARAM_BYTES)
    for n in PTRS:
        struct.pack_into("<Q", blob, OFF[n], ptr_xform(base_of[n]) if arrays.get(n) is not None else 0)
    allints = dict(ints)
    allints["stride"] = STRIDE
    allints.setdefault("seq", 0)
    for n in INTS:
        struct.pack_into("<i", blob, OFF[n], int(allints[n]))
    for n in FLOATS:
        struct.pack_into("<f", blob, OFF[n], float(floats[n]))
    PBASE = 0x00007E0080000000
    S[S_PARAM], S[S_PARAM + 1] = PBASE & 0xFFFFFFFF, PBASE >> 32
    V[0], V[1] = 0, 0
    exec_ = 1
    scc = 0
    labels = {}
    for k, t in enumerate(ins):
        if t[0] == "label":
            labels.setdefault(t[1], []).append(k)

    def asf(bits):
        return np.array([bits], u32).view(f32)[0]

    def bits(val):
        return np.array([val], f32).view(u32)[0]

    def sreg(x):
        if x == "vcc":
            return 106
        if x == "exec":
            return 126
        return int(x[2:x.index(":")]) if x.startswith("s[") else int(x[1:])

    def s64(x):
        if isinstance(x, int):
            return x & 0xFFFFFFFFFFFFFFFF
        if x == "exec":
            return exec_
        lo = sreg(x)
        return (S.get(lo, 0) & 0xFFFFFFFF) | ((S.get(lo + 1, 0) & 0xFFFFFFFF) << 32)

    def set64(x, val):
        nonlocal exec_
        if x == "exec":
            exec_ = val & 1
            return
        lo = sreg(x)
        S[lo], S[lo + 1] = val & 0xFFFFFFFF, (val >> 32) & 0xFFFFFFFF

    def s32(x):
        if isinstance(x, int):
            return x & 0xFFFFFFFF
        return S.get(sreg(x), 0) & 0xFFFFFFFF

    def src_bits(x):
        """raw 32 bits of an operand (no modifiers)"""
        if isinstance(x, float):
            return int(bits(f32(x)))
        if isinstance(x, int):
            return x & 0xFFFFFFFF
        if x[0] == "v":
            return int(V[int(x[1:])])
        if x[0] == "s":
            return s32(x)
        raise ValueError(x)

    def fsrc(x):
        if isinstance(x, (int, float)) and not isinstance(x, bool):
            if isinstance(x, int):
                # integer inline constants used as float operands are only 0 here
                assert x == 0, x
                return f32(0)
            return f32(x)
        neg = x.startswith("-")
        if neg:
            x = x[1:]
        ab = x.startswith("|")
        if ab:
            x = x[1:-1]
        val = asf(u32(src_bits(x)))
        if ab:
            val = f32(abs(val))
        return f32(-val) if neg else val

    def setv(x, val):
        V[int(x[1:])] = bits(f32(val))

    def setvb(x, b):
        V[int(x[1:])] = u32(b & 0xFFFFFFFF)

    def half(x, sel):
        lo = int(x[2:x.index(":")])
        if x[0] == "v":
            return asf(V[lo + sel])
        return asf(u32(S.get(lo + sel, 0)))

    FLAT = ("taskf", "wl")      # plain word arrays (not [row][B]): the byte offset / 4 indexes them

    def mem(addr):
        """array and element of a byte address: SGPR base (64 bits) + zero-extended VGPR offset + immediate, as the ISA forms
        it; anything that is not an element of an array handed in is a fault (asmqp.AddressFault)"""
        addr &= 0xFFFFFFFFFFFFFFFF
        if PBASE <= addr < PBASE + PARAM_BYTES:
            return None, addr - PBASE
        for name in names:
            arr = arrays.get(name)
            if arr is None:
                continue
            stride = 4 if name in FLAT else STRIDE
            off = addr - base_of[name]
            if 0 <= off < len(arr) * stride and off % stride == 0:
                return arr, off // stride
        from .asmqp import AddressFault
        raise AddressFault("%r touches 0x%016x, outside every array of the call" % (ins[pc], addr))

    def setmask(dst, cond):
        if dst == "vcc":
            S[106], S[107] = int(bool(cond)), 0
        else:
            set64(dst, int(bool(cond)))

    cmpf = {"lt": lambda a, b: a < b, "le": lambda a, b: a <= b, "gt": lambda a, b: a > b, "ge": lambda a, b: a >= b,
            "eq": lambda a, b: a == b, "u": lambda a, b: (a != a) or (b != b)}
    pc = nexec = 0
    nquad = [0]
    sections = {}
    self_neq = 2 * 3 * symbolic.NY
    simulate.last_quad_instructions = 0
    simulate.last_quad_sections = sections
    with np.errstate(all="ignore"):
        while pc < len(ins):
            t = ins[pc]
            m = t[0]
            if m == "kill":
                V[int(t[1][1:])] = POISON
                pc += 1
                continue
            if m == "label":
                pc += 1
                continue
            if m == "quad_begin":
                # the one-robot-per-quad section (asmquad.py): the four lanes of a quad have run everything so far
                # redundantly, so each starts from THIS lane's registers, AGPRs and LDS slice; afterwards the four lanes must
                # agree on every one-lane home of the loop's outputs, and nothing else may be read again (poisoned)
                from . import asmquad
                assert exec_ == 1
                V4, A4, L4 = np.tile(V, (4, 1)), np.tile(A, (4, 1)), np.tile(lds, (4, 1))
                pc, nq = asmquad.simulate(ins, pc, V4, A4, L4, S)
                nexec += nq
                nquad[0] += nq
                sections[t[1] if len(t) > 1 else "admm"] = sections.get(t[1] if len(t) > 1 else "admm", 0) + nq
                endm = ins[pc - 1]
                if len(endm) > 1:           # a section that names what it hands back (the Ruiz passes): everything else is as before
                    for r in endm[1]:
                        assert (V4[1:4, r] == V4[0, r]).all(), "lanes of the quad disagree on v%d after the %s section" % (r, t[1])
                        V[r] = V4[0, r]
                    assert (A4[1:4] == A4[0]).all()
                    continue
                keep = set([0, 1]) | set(range(V_W, V_Z)) | set(range(V_Z + self_neq, V_Z + 40))
                for r in range(256):
                    if r in keep:
                        assert (V4[1:4, r] == V4[0, r]).all(), "lanes of the quad disagree on v%d after the quad section" % r
                        V[r] = V4[0, r]
                    else:
                        V[r] = POISON
                assert (A4[1:4, A_LO:] == A4[0, A_LO:]).all()
                A[:] = A4[0]
                A[:A_LO] = POISON          # L, 1/D, q homes were consumed; bounds, thrust-row words and weights stay
                continue
            nexec += 1
            assert nexec < max_exec, "runaway program"
            if m in ("s_waitcnt", "s_nop", "buffer_wbl2"):
                pass
            elif m.startswith("s_load_dword"):
                n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[m]
                imm = int(t[4].split(":")[1]) if len(t) > 4 else 0
                arr, off = mem(s64(t[2]) + (t[3] if isinstance(t[3], int) else s32(t[3])) + imm)
                lo = sreg(t[1])
                for k in range(n):
                    if arr is None:
                        S[lo + k] = struct.unpack_from("<I", blob, off + 4 * k)[0]
                    else:
                        S[lo + k] = int(bits(arr[off + k]))
            elif m == "s_mov_b32":
                S[sreg(t[1])] = s32(t[2])
            elif m == "s_mov_b64":
                set64(t[1], 0xFFFFFFFFFFFFFFFF if t[2] == -1 else s64(t[2]))
                if t[1] != "exec" and t[2] == -1:
                    set64(t[1], 1)
            elif m == "s_mul_i32":
                S[sreg(t[1])] = (s32(t[2]) * s32(t[3])) & 0xFFFFFFFF
            elif m == "s_lshl_b32":
                S[sreg(t[1])] = (s32(t[2]) << (s32(t[3]) & 31)) & 0xFFFFFFFF
            elif m in ("s_add_i32", "s_sub_i32"):
                a, b = s32(t[2]), s32(t[3])
                S[sreg(t[1])] = (a + b if m == "s_add_i32" else a - b) & 0xFFFFFFFF
            elif m in ("s_cmp_lt_i32", "s_cmp_gt_i32"):
                a, b = s32(t[1]), s32(t[2])
                a = a - (1 << 32) if a & 0x80000000 else a
                b = b - (1 << 32) if b & 0x80000000 else b
                scc = int(a < b) if m == "s_cmp_lt_i32" else int(a > b)
            elif m in ("s_cmp_lg_u32", "s_cmp_eq_u32"):
                scc = int((s32(t[1]) != s32(t[2])) == (m == "s_cmp_lg_u32"))
            elif m == "s_cmp_eq_u64":
                scc = int(s64(t[1]) == s64(t[2]))
            elif m in ("s_and_b64", "s_or_b64", "s_andn2_b64"):
                a, b = s64(t[2]) & 1, s64(t[3]) & 1
                r = (a & b) if m == "s_and_b64" else (a | b) if m == "s_or_b64" else (a & (1 - b))
                setmask(t[1], r) if t[1] == "vcc" else set64(t[1], r)
                scc = int(r != 0)
            elif m == "s_and_saveexec_b64":
                set64(t[1], exec_)
                exec_ = exec_ & (s64(t[2]) & 1)
                scc = int(exec_ != 0)
            elif m in ("s_branch", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_execz"):
                take = m == "s_branch" or (m == "s_cbranch_scc1" and scc) or \
                    (m == "s_cbranch_vccz" and (S.get(106, 0) & exec_) == 0) or (m == "s_cbranch_execz" and exec_ == 0)
                if take:
                    lab, d = t[1][:-1], t[1][-1]
                    cands = labels[lab]
                    pc = min(c for c in cands if c > pc) if d == "f" else max(c for c in cands if c < pc)
            elif not exec_ and m[0] in "vgd":
                pass                                      # the lane is masked off
            elif m == "global_load_dword":
                imm = int(t[4].split(":")[1]) if len(t) > 4 and isinstance(t[4], str) and t[4].startswith("offset:") else 0
                arr, row = mem(s64(t[3]) + int(V[int(t[2][1:])]) + imm)
                val = np.array([arr[row]]).view(u32)[0] if arr.dtype != np.float32 else bits(arr[row])
                if t[1][0] == "a":
                    A[int(t[1][1:])] = val
                else:
                    V[int(t[1][1:])] = val
            elif m == "global_store_dword":
                arr, row = mem(s64(t[3]) + int(V[int(t[1][1:])]))
                raw = V[int(t[2][1:])]
                arr[row] = asf(raw) if arr.dtype == np.float32 else np.array([raw], u32).view(np.int32)[0]
            elif m in ("ds_write_b128", "ds_read_b128", "ds_write_b32", "ds_read_b32"):
                w0 = t[3] // 1024 * 4 + (t[3] % 1024) // 4
                if m == "ds_write_b128":
                    lo = int(t[2][2:t[2].index(":")])
                    lds[w0:w0 + 4] = V[lo:lo + 4]
                elif m == "ds_read_b128":
                    lo = int(t[1][2:t[1].index(":")])
                    V[lo:lo + 4] = lds[w0:w0 + 4]
                elif m == "ds_write_b32":
                    lds[w0] = V[int(t[2][1:])]
                else:
                    V[int(t[1][1:])] = lds[w0]
            elif m == "v_accvgpr_read_b32":
                V[int(t[1][1:])] = A[int(t[2][1:])]
            elif m == "v_accvgpr_write_b32":
                A[int(t[1][1:])] = u32(src_bits(t[2]))
            elif m == "v_mov_b32":
                x = t[2]
                if isinstance(x, str) and (x.startswith("-") or x.startswith("|")):
                    setv(t[1], fsrc(x))
                else:
                    setvb(t[1], src_bits(x))
            elif m == "v_pk_mov_b32":
                d = t[-1]
                lo = int(t[1][2:t[1].index(":")])
                r0, r1 = half(t[2], d["op_sel"][0]), half(t[3], d["op_sel"][1])
                V[lo], V[lo + 1] = bits(r0), bits(r1)
            elif m == "v_add_u32":
                setvb(t[1], (src_bits(t[2]) + src_bits(t[3])) & 0xFFFFFFFF)
            elif m == "v_and_b32":
                setvb(t[1], src_bits(t[2]) & src_bits(t[3]))
            elif m == "v_fma_f32":
                setv(t[1], f32(np.float64(fsrc(t[2])) * np.float64(fsrc(t[3])) + np.float64(fsrc(t[4]))))
            elif m == "v_fmac_f32":
                setv(t[1], f32(np.float64(fsrc(t[2])) * np.float64(fsrc(t[3])) + np.float64(fsrc(t[1]))))
            elif m == "v_fmaak_f32":
                setv(t[1], f32(np.float64(fsrc(t[2])) * np.float64(fsrc(t[3])) + np.float64(asf(u32(t[4])))))
            elif m == "v_mul_f32":
                setv(t[1], f32(fsrc(t[2]) * fsrc(t[3])))
            elif m == "v_add_f32":
                setv(t[1], f32(fsrc(t[2]) + fsrc(t[3])))
            elif m == "v_sub_f32":
                setv(t[1], f32(fsrc(t[2]) - fsrc(t[3])))
            elif m == "v_subrev_f32":
                setv(t[1], f32(fsrc(t[3]) - fsrc(t[2])))
            elif m in ("v_max_f32", "v_min_f32"):
                a, b = fsrc(t[2]), fsrc(t[3])
                r = (a if b != b else b if a != a else (max(a, b) if m == "v_max_f32" else min(a, b)))
                setv(t[1], r)
            elif m == "v_max3_f32":
                vals = [x for x in (fsrc(t[2]), fsrc(t[3]), fsrc(t[4])) if x == x]
                setv(t[1], max(vals) if vals else f32(np.nan))
            elif m == "v_min3_f32":
                vals = [x for x in (fsrc(t[2]), fsrc(t[3]), fsrc(t[4])) if x == x]
                setv(t[1], min(vals) if vals else f32(np.nan))
            elif m == "v_med3_f32":
                vals = sorted((fsrc(t[2]), fsrc(t[3]), fsrc(t[4])))
                setv(t[1], vals[1])
            elif m == "v_rcp_f32":
                setv(t[1], f32(1.0) / fsrc(t[2]))
            elif m == "v_rsq_f32":
                setv(t[1], f32(1.0 / np.sqrt(np.float64(fsrc(t[2])))))
            elif m == "v_sqrt_f32":
                setv(t[1], f32(np.sqrt(np.float64(fsrc(t[2])))))
            elif m.startswith("v_cmp_") and m.endswith("_f32") or m.startswith("v_cmp_") and m.endswith("_f32_e64"):
                op = m[len("v_cmp_"):].split("_")[0]
                setmask(t[1], cmpf[op](fsrc(t[2]), fsrc(t[3])))
            elif m == "v_cmp_eq_i32_e64":
                a, b = src_bits(t[2]), src_bits(t[3])
                setmask(t[1], a == b)
            elif m in ("v_cndmask_b32", "v_cndmask_b32_e64"):
                sel = s64(t[4]) & 1
                setvb(t[1], src_bits(t[3]) if sel else src_bits(t[2]))
            elif m in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"):
                d = t[-1]
                srcs = t[2:-1]
                dlo = int(t[1][2:t[1].index(":")])
                res = []
                for hi in (0, 1):
                    sel = d["op_sel_hi"] if hi else d["op_sel"]
                    ng = d["neg_hi"] if hi else d["neg_lo"]
                    vals = [np.float64(half(x, sel[q])) * (-1 if ng[q] else 1) for q, x in enumerate(srcs)]
                    if m == "v_pk_fma_f32":
                        res.append(f32(vals[0] * vals[1] + vals[2]))
                    elif m == "v_pk_mul_f32":
                        res.append(f32(f32(vals[0]) * f32(vals[1])))
                    else:
                        res.append(f32(f32(vals[0]) + f32(vals[1])))
                V[dlo], V[dlo + 1] = bits(res[0]), bits(res[1])
            else:
                raise ValueError("unknown instruction %r" % (t,))
            pc += 1
    simulate.last_quad_instructions = nquad[0]
    return nexec


if __name__ == "__main__":
    p, n = write()
    print("wrote", p, n, "instructions")


def host_floats(dt=5.0, g=9.81e-3, TtoWmax=2.0, ws=1e1, wds=1e3, wpr=1.0, wpf=5.0, wvr=1e3, wvf=2e3, wthrust=1e-1, wmom=1e-2,
                Ib=(3333.0, 3333.0, 1000.0), dtsim=0.2, taulim=100.0, mb=100.0):
    """The float members of StepParams exactly as umpc_mi355x.hip fills them (fp32 arithmetic on the host)."""
    import numpy as np
    f = np.float32
    one = f(1.0)
    w = dict(wpr=f(wpr), wpf=f(wpf), ws_=f(ws), wvr=f(wvr), wvf=f(wvf), wds=f(wds), wthrust=f(wthrust), wmom=f(wmom))
    d = dict(dt=f(dt), dtg=f(f(dt) * f(g)), Tmax=f(f(TtoWmax) * f(g)), **w)
    for k, val in w.items():
        d["i" + k.rstrip("_")] = f(one / val)
    for i in range(3):
        d["Ib%d" % i] = f(Ib[i])
        d["Ibi%d" % i] = f(one / f(Ib[i]))
    d.update(h=f(dtsim), hh=f(f(0.5) * f(dtsim)), h6=f(f(dtsim) / f(6.0)), taulim=f(taulim), gpl=f(9.81e-3), idt=f(one / f(dt)),
             mbg=f(f(mb) * f(g)))
    assert set(d) == set(FLOATS), set(FLOATS) ^ set(d)
    return {k: float(val) for k, val in d.items()}

"""Emits the hand-scheduled gfx950 assembly for the ADMM phase of the step
kernel (robobee3d_amd/csrc/umpc_admm_asm.h, one `asm volatile` block).

Why assembly: the loop's working set (L 213 + 1/D 84 + q 45 + W 84 + x,y,z 123
+ 12 bound/rho words = 561 words per lane) only fits a CU when VGPRs, AGPRs
and LDS are all used and every word has a fixed home; hipcc's allocator sees
256 VGPRs and spills the factor to scratch (measured: 144 scratch loads per
iteration, 2-3 GB of spill traffic per launch, rocprof FETCH_SIZE). Here the
placement is static (one lane = one robot, one wave per SIMD):

    v0            robot byte offset (4*b)              input
    v1            lane LDS address (base + 16*lane)    input
    v2..v86       W      KKT rhs / solution, stored by ORIGINAL index (x part v2..v46, z part v48..v86)
    v88..v132     x      v134..v172  y      v174..v212  z        (even bases: v_pk_* register pairs)
    v214..v229    LDS read ring (4 x float4)
    v230..v237    AGPR read temporaries (4 pairs)
    v238..v245    arithmetic temporaries
    v246..v255    not touched (left to the compiler for values that live across the block)
    a0..a52       L[160..212]   a53..a136  1/D   a137..a181  q   a182..a217  l(=u) dynamics rows
    a218..a229    lo3 up3 rho3 rinv3 (thrust rows)
    LDS           L[0..159] as 40 float4 per lane (ds_read_b128, conflict-free); x, y, z on exit

A lone wave issues one VALU instruction per ~5 cycles whether it is v_fma_f32 or v_pk_fma_f32
(tools/microbench.hip), so everything elementwise runs packed, two rows per instruction: the rhs,
the 1/D scaling, the x update and the y update of the dynamics rows. The triangular solves stay
scalar (426 FMAs). Per middle iteration: ~1000 instructions, no scratch, no HBM.

Arithmetic vs UMPC_GEN_ADMM_ITER (the C++ / fp64 statement of the same iteration): same operation
order and the same FMA placement, except that after the first iteration the dynamics rows (z == l == u)
use delta_y = alpha (nu - y) instead of the seven-operation chain it is algebraically equal to.

Reference mapping: auxil.c:164-228 (compute_rhs, update_x, update_z, update_y),
qdldl_interface.c:322-369, qdldl.c:250-293, proj.c:4-14.

`simulate()` interprets the emitted instruction list on numpy float32 so that
the CPU test-suite can check the schedule (register reuse, fetch distances,
packed-operand selects) against the oracle without a GPU.
"""
import os
import struct

from . import symbolic

HERE = os.path.dirname(os.path.abspath(__file__))

# Generation-time switches are read from the environment when the generator modules are imported (A/B timing of kernel
# variants, tools/build_variant.py). Everything that matches these prefixes changes the emitted instruction streams;
# the names below it are RUN-time diagnostics of the host side and do not.
SWITCH_PREFIXES = ("UMPC_ASM_", "UMPC_ASM64_", "UMPC_QP_", "UMPC_X_")
# RUN-time variables that happen to match a prefix (read with getenv() by the host code in csrc/, or by batchqp.py, when a
# kernel is launched; they select among kernels that exist, they do not change an emitted stream).
# tests/test_generated_headers.py greps csrc/ and keeps this list honest.
NOT_SWITCHES = ("UMPC_QP_KERNEL", "UMPC_QP_NO_ASM", "UMPC_ASM_SKEW_US", "UMPC_ASM_SKEW_GROUPS")


def generator_switches():
    """{name: value} of the generator switches present in the environment (empty = the shipped kernels)"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith(SWITCH_PREFIXES) and k not in NOT_SWITCHES}


def switch_banner():
    """one comment line for a generated header: which switches produced it"""
    sw = generator_switches()
    return "// generator switches: " + (" ".join("%s=%s" % kv for kv in sw.items()) if sw else "none (defaults = the shipped kernels)")

XV_COUNT, XV_BASE = 0, 246   # asmstep.py: storage positions NLDS+NVZ .. +XV_COUNT-1 of L live in VGPRs XV_BASE.. (free there)
NLDS = 160  # L storage positions kept in LDS
NVZ = 36    # positions NLDS .. NLDS+NVZ-1 move into the z registers of the dynamics rows after the first iteration
# workspace rows shared with the C++ phases (see umpc_step.h)
FAC_L, FAC_DI, FAC_Q, FAC_LOEQ, FAC_M = 0, 213, 297, 342, 378
FAC_ROWS = 390
WS_DS, WS_ES, WS_C, WS_XPREV, WS_DY = 390, 435, 474, 475, 520
WS_ROWS = 559

V_W, V_WZ, V_X, V_Y, V_Z = 2, 48, 88, 134, 174
V_RING, V_AT, V_TT, V_END = 214, 230, 238, 246
N_ATP = 4  # AGPR-read temporaries, in pairs
# LDS ring slots of the Fetcher. EXPERIMENT (asmstep.py, UMPC_ASM_RING=6 together with UMPC_ASM_XV=0): two more slots in
# v246..v253 instead of ten L words there; set by asmstep for its own stream only (module state, like XV_COUNT)
NRING = 4


def ring_base(slot):
    return V_RING + 4 * slot if slot < 4 else 246 + 4 * (slot - 4)
A_L, A_D, A_Q, A_LO, A_M = 0, 53, 137, 182, 218
S_WS, S_CTRL, S_STRIDE, S_ITERS = 4, 6, 10, 11
S_P, S_CNT, S_P2 = 12, 14, 16
S_ALPHA, S_OMA, S_SIGMA, S_RINV, S_RHO = 20, 22, 24, 26, 28  # even: low half of an SGPR pair (packed broadcast)


def f32bits(v):
    return struct.unpack("<I", struct.pack("<f", v))[0]


def _vgprs(x):
    """the VGPR numbers an operand string names ('v7', '-v7', '|v7|', 'v[4:7]'); empty for anything else"""
    if not isinstance(x, str):
        return set()
    x = x.strip("-|")
    if x.startswith("v[") and x.endswith("]"):
        lo, hi = x[2:-1].split(":")
        return set(range(int(lo), int(hi) + 1))
    if x.startswith("v") and x[1:].isdigit():
        return {int(x[1:])}
    return set()


WIDE_STORES = {"ds_write_b128": 2, "ds_write_b96": 2, "global_store_dwordx4": 2, "global_store_dwordx3": 2}   # mnemonic -> data operand
WIDE_STORE_WAIT = 2     # wait states (gfx940 and later) before a VALU may overwrite the data registers of such a store


def valu_dst(t):
    """VGPRs a VALU instruction tuple writes"""
    m = t[0]
    if not m.startswith("v_") or m.startswith("v_cmp") or m in ("v_accvgpr_write_b32", "v_readfirstlane_b32", "v_readlane_b32"):
        return set()
    return _vgprs(t[1])


def wide_store_hazards(ins):
    """[(index of the store, index of the VALU)]: a store of more than 64 bits reads its data registers over several cycles; a
    VALU instruction that overwrites one of them within WIDE_STORE_WAIT wait states races with it (the hardware does not
    interlock this; hipcc's hazard recogniser inserts the nops for compiled code, nobody does for inline assembly)."""
    out = []
    for k, t in enumerate(ins):
        if t[0] in WIDE_STORES:
            data = _vgprs(t[WIDE_STORES[t[0]]])
            ws, j = 0, k + 1
            while j < len(ins) and ws < WIDE_STORE_WAIT:
                u = ins[j]
                if u[0] == "label":
                    j += 1
                    continue
                if u[0] == "s_nop":
                    ws += int(u[1]) + 1
                elif valu_dst(u) & data:
                    out.append((k, j))
                    ws += 1
                else:
                    ws += 1
                j += 1
    return out


class Emit:
    """Collects instruction tuples. Keeps the one hazard of straight-line LDS / VMEM stores out of every generated stream: a VALU
    write to the data registers of a store of more than 64 bits waits WIDE_STORE_WAIT states (wide_store_hazards)."""

    def __init__(self):
        self.ins = []  # tuples (mnemonic, operands..., [dict of VOP3P modifiers])

    def __call__(self, *t):
        dst = valu_dst(t)
        if dst:
            ws = 0
            for u in reversed(self.ins[-(WIDE_STORE_WAIT + 1):]):
                if ws >= WIDE_STORE_WAIT or u[0] == "label":
                    break            # (a label: whatever precedes it is not necessarily what ran before)
                if u[0] in WIDE_STORES and _vgprs(u[WIDE_STORES[u[0]]]) & dst:
                    self.ins.append(("s_nop", WIDE_STORE_WAIT - ws - 1))
                    break
                ws += int(u[1]) + 1 if u[0] == "s_nop" else 1
        self.ins.append(t)


def _row_ptr(e, sreg, base, row):
    """s[sreg:sreg+1] = base + row * stride"""
    e("s_mul_i32", "s%d" % sreg, "s%d" % S_STRIDE, row)
    e("s_mul_hi_u32", "s%d" % (sreg + 1), "s%d" % S_STRIDE, row)
    e("s_add_u32", "s%d" % sreg, "s%d" % sreg, "s%d" % base)
    e("s_addc_u32", "s%d" % (sreg + 1), "s%d" % (sreg + 1), "s%d" % (base + 1))


def _adv(e, sreg):
    e("s_add_u32", "s%d" % sreg, "s%d" % sreg, "s%d" % S_STRIDE)
    e("s_addc_u32", "s%d" % (sreg + 1), "s%d" % (sreg + 1), 0)


# ---------------------------------------------------------------------------
# Register slots and L storage order
# ---------------------------------------------------------------------------
def slot_maps(s):
    """x / constraint-row index -> register slot. The KKT pattern is invariant under the cyclic relabelling of the
    (x, y, z) components of every position / orientation triple, so the solve applies the same operation to the x and
    the y member of a triple with partner operands: the slots put those two members on an even-aligned register
    pair (v_pk_fma_f32 operands); the z member and the inputs keep single slots."""
    def one(n, ntrip):
        sl = list(range(n))
        for t in range(ntrip):
            st = 3 * t
            if st % 2:  # (x, y) must start on an even slot: rotate the triple to (z, x, y)
                sl[st], sl[st + 1], sl[st + 2] = st + 1, st + 2, st
        return sl
    ntrip = 4 * s.N  # 2N blocks of (p, s) per part
    xs, zs = one(s.nx, ntrip), one(s.nc, ntrip)
    xinv, zinv = [0] * s.nx, [0] * s.nc
    for j, p in enumerate(xs):
        xinv[p] = j
    for i, p in enumerate(zs):
        zinv[p] = i
    return xs, zs, xinv, zinv


def w_reg(s, k, xs, zs):
    """VGPR of the permuted KKT unknown k (W is stored by slot of its ORIGINAL index)."""
    o = s.perm[k]
    return V_W + xs[o] if o < s.nx else V_WZ + zs[o - s.nx]


# EXPERIMENT, off by default (UMPC_ASM_BCAST=1): broadcast-source pairs in the solves -- two entries of one column whose rows
# are slot partners (forwards), two entries of one row whose columns are (backwards) as ONE packed instruction with the
# source broadcast: 150 -> 140 / 141 instructions per solve, 778 -> 759 per iteration. Measured on the MI355X (same box,
# K = 500): 0.12145 / 0.12175 ms per step against 0.1220 / 0.1224 -- 0.4 % for 2.4 % fewer instructions (the packed ops
# lengthen dependent chains), and it moves the rounding of every iterate. Not shipped.
BCAST = os.environ.get("UMPC_ASM_BCAST", "0") == "1"


def solve_schedule(s, reg, direction, allowed=None, same_src=False):
    """List-schedules one triangular solve. Ops are the L entries j = (row r, column c): forward W[r] -= L_j W[c]
    (ready once W[c] is final), backward W[c] -= L_j W[r] (ready once W[r] is final). Two ready ops whose destination
    registers AND source registers each form an aligned pair are issued as one packed instruction; an op whose
    structural partner is not ready yet waits for it. Returns [(op,) | (op_lo, op_hi)] with op = (dst_k, src_k, j),
    op_lo the member whose destination register is even. Operation order inside one unknown differs from QDLDL's
    column sweep (rounding only)."""
    nk = s.nk
    ops = [(s.L_i[j], c, j) for c in range(nk) for j in range(s.L_p[c], s.L_p[c + 1])]
    if direction == "bwd":
        ops = [(c, r, j) for (r, c, j) in ops]

    def ok(o, o2):
        # partner destinations; partner sources, or (same_src) ONE source broadcast to both halves: two entries of one
        # column whose rows are slot partners
        return (o2[2] != o[2] and reg[o2[0]] == (reg[o[0]] ^ 1)
                and (reg[o2[1]] == (reg[o[1]] ^ 1) or (same_src and o2[1] == o[1]))
                and (allowed is None or frozenset((o[2], o2[2])) in allowed))
    indeg = {k: 0 for k in range(nk)}
    for (d, _, _) in ops:
        indeg[d] += 1
    final = {k for k in range(nk) if indeg[k] == 0}
    remaining, out = list(ops), []
    while remaining:
        ready = [o for o in remaining if o[1] in final]
        used, emitted = set(), []
        for o in ready:
            if o[2] in used:
                continue
            for o2 in ready:
                if o2[2] not in used and ok(o, o2):
                    used.update((o[2], o2[2]))
                    emitted.append((o, o2) if reg[o[0]] % 2 == 0 else (o2, o))
                    break
        deferred = []
        for o in ready:
            if o[2] in used:
                continue
            if any(ok(o, o2) for o2 in remaining if o2[1] not in final and o2[2] not in used):
                deferred.append(o)
            else:
                used.add(o[2])
                emitted.append((o,))
        if not emitted:
            used.add(deferred[0][2])
            emitted.append((deferred[0],))
        out += emitted
        for o in remaining:
            if o[2] in used:
                indeg[o[0]] -= 1
                if indeg[o[0]] == 0:
                    final.add(o[0])
        remaining = [o for o in remaining if o[2] not in used]
    return out


def solve_plan(s):
    """Forward / backward schedules with the SAME entry pairs (so that one storage order serves both) and the
    storage position of every L entry: in order of first use by the forward solve, paired entries adjacent on an
    even position (an LDS float4 holds two pairs; an AGPR pair is fetched with two v_accvgpr_read)."""
    xs, zs, _, _ = slot_maps(s)
    reg = [w_reg(s, k, xs, zs) for k in range(s.nk)]
    fwd = solve_schedule(s, reg, "fwd", same_src=BCAST)
    # Broadcast couples for the BACKWARD solve: two forward singles (r <- c), (r <- c') into the same unknown r whose
    # columns c, c' are slot partners become, backwards, (c <- r), (c' <- r): partner destinations, one source -- one
    # packed instruction with the source broadcast, if their L entries are an aligned pair in storage. The earlier of the
    # two forward ops is moved next to the later one (legal: W[r] is read by nobody before all its updates are done, and
    # the source of the delayed op stays final), so the storage order below makes them adjacent.
    couples, late = {}, {}
    if BCAST:
        singles = [(k, g[0]) for k, g in enumerate(fwd) if len(g) == 1]
        used = set()
        for a_, (ka, oa) in enumerate(singles):
            if oa[2] in used:
                continue
            for kb, ob in singles[a_ + 1:]:
                if ob[2] not in used and ob[0] == oa[0] and reg[ob[1]] == (reg[oa[1]] ^ 1):
                    couples[oa[2]] = ob[2]
                    used.update((oa[2], ob[2]))
                    break
        moved = {j for j in couples}                    # the earlier op of every couple leaves its place ...
        late = {jb: ja for ja, jb in couples.items()}   # ... and is re-inserted right before its partner
        byj = {g[0][2]: g for g in fwd if len(g) == 1}
        out = []
        for g in fwd:
            if len(g) == 1 and g[0][2] in moved:
                continue
            if len(g) == 1 and g[0][2] in late:
                out.append(byj[late[g[0][2]]])
            out.append(g)
        fwd = out
    # The reversed forward order with the roles of row and column swapped is a legal backward order (an entry of
    # column r follows every entry of row r in the forward solve) with the same pairs, and it walks the storage
    # backwards: each LDS float4 is fetched once per solve. A forward pair with ONE source (two rows of a column) has ONE
    # destination backwards: two singles there; a couple is two singles forwards and one pair backwards.
    bwd = []
    k = len(fwd) - 1
    while k >= 0:
        g = fwd[k]
        if len(g) == 1 and g[0][2] in late:                         # (its partner is fwd[k - 1])
            ob, oa = g[0], fwd[k - 1][0]
            h = [(oa[1], oa[0], oa[2]), (ob[1], ob[0], ob[2])]      # (dst c, src r, j), (dst c', src r, j')
            if reg[h[0][0]] % 2:
                h.reverse()
            bwd.append(tuple(h))
            k -= 2
            continue
        h = [(sr, d, j) for (d, sr, j) in g]
        if len(h) == 2 and h[0][0] == h[1][0]:                      # one destination: not a pair in this direction
            bwd.append((h[1],))
            bwd.append((h[0],))
        else:
            if len(h) == 2 and reg[h[0][0]] % 2:
                h.reverse()
            bwd.append(tuple(h))
        k -= 1
    # storage: in order of first use by the forward solve; pairs and couples adjacent on an even position
    units = []
    k = 0
    while k < len(fwd):
        g = fwd[k]
        if len(g) == 1 and g[0][2] in couples and k + 1 < len(fwd) and len(fwd[k + 1]) == 1 and fwd[k + 1][0][2] == couples[g[0][2]]:
            units.append([g[0][2], fwd[k + 1][0][2]])
            k += 2
        else:
            units.append([o[2] for o in g])
            k += 1
    order, pending = [], []
    for u in units:
        if len(u) == 2 and len(order) % 2:
            pending.append(u)      # wait for a single to restore the parity
            continue
        order += u
        while pending and len(order) % 2 == 0:
            order += pending.pop(0)
    assert not pending, "odd number of single entries before a trailing pair"
    pos = [0] * len(s.L_i)
    for p_, j in enumerate(order):
        pos[j] = p_
    for g in fwd + bwd:
        if len(g) == 2:
            a, b = pos[g[0][2]], pos[g[1][2]]
            assert a // 2 == b // 2, "paired entries must share an aligned pair"
    return fwd, bwd, pos


def l_positions(N=3, perm=None):
    """Storage position (LDS word < NLDS, else workspace row FAC_L + pos -> AGPR) of every L entry, by CSC index."""
    return solve_plan(symbolic.analyse(N, perm))[2]


def prologue(e, s):
    import numpy as np
    for reg, val in ((S_ALPHA, 1.6), (S_OMA, float(np.float32(1.0) - np.float32(1.6))), (S_SIGMA, 1e-6),
                     (S_RINV, 0.01), (S_RHO, 100.0)):
        e("s_mov_b32", "s%d" % reg, f32bits(val))
        e("s_mov_b32", "s%d" % (reg + 1), f32bits(val))
    # L[0..NLDS) was written into LDS by phase A (same lane, same layout); everything else arrives through
    # the workspace rows, issued back to back with ONE wait (every wave of the grid is in this phase at the
    # same time, so each exposed round trip costs microseconds).
    _row_ptr(e, S_P, S_WS, FAC_L + NLDS)
    # the rest of L, 1/D, q, the dynamics-row bounds and the thrust-row words -> AGPRs (consecutive rows)
    nrest = len(s.L_i) - NLDS + s.nk + s.nx + 2 * s.N * symbolic.NY + 12
    assert A_LO == nrest - 12 - 2 * s.N * symbolic.NY and A_M == nrest - 12 and nrest <= 256
    for r in range(nrest):
        e("global_load_dword", "a%d" % r, "v0", "s[%d:%d]" % (S_P, S_P + 1))
        _adv(e, S_P)
    # x, y, z
    e("s_mov_b64", "s[%d:%d]" % (S_P, S_P + 1), "s[%d:%d]" % (S_CTRL, S_CTRL + 1))
    xs, zs, _, _ = slot_maps(s)
    for base, n, sl in ((V_X, s.nx, xs), (V_Y, s.nc, zs), (V_Z, s.nc, zs)):
        for r in range(n):
            e("global_load_dword", "v%d" % (base + sl[r]), "v0", "s[%d:%d]" % (S_P, S_P + 1))
            _adv(e, S_P)
    e("s_waitcnt", "vmcnt(0)")


def epilogue(e, s, z_is_l):
    # x, y, z stay on chip: the factor in LDS is dead now, phase C reads the iterates from LDS words 0..122.
    # Registers are [x 45 | pad | y 39 | pad | z 39]; LDS words are contiguous, so y and z are written one
    # word at a time where they straddle a pad. After >= 1 iteration z of the dynamics rows is l (AGPR A_LO) and
    # their registers hold L entries.
    w = 0
    xs, zs, _, _ = slot_maps(s)
    neq = 2 * s.N * symbolic.NY
    for base, n, sl in ((V_X, s.nx, xs), (V_Y, s.nc, zs), (V_Z, s.nc, zs)):
        for r in range(n):
            if base == V_Z and z_is_l and r < neq:
                t = V_RING + r % 16
                e("v_accvgpr_read_b32", "v%d" % t, "a%d" % (A_LO + r))
                e("ds_write_b32", "v1", "v%d" % t, (w // 4) * 1024 + (w % 4) * 4)
            else:
                e("ds_write_b32", "v1", "v%d" % (base + sl[r]), (w // 4) * 1024 + (w % 4) * 4)
            w += 1
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")


class Fetcher:
    """Issues the L / 1-over-D / q operand fetches a fixed distance ahead of their consumers.
    src: None | ('A', a) -> one VGPR | ('A2', a_lo, a_hi) -> an aligned VGPR pair | ('L', lds_word)."""

    def __init__(self, e, la=3):
        self.e, self.la = e, la
        self.nds = 0          # ds_reads issued so far in this body
        self.waited = -1      # issue index of the last ds_read known to have returned
        self.lds_ahead = int(os.environ.get("UMPC_ASM_LDS_AHEAD", "12" if NRING == 4 else "20"))  # ops between a ds_read and its first consumer
        self.merge = int(os.environ.get("UMPC_ASM_LDS_MERGE", "1" if NRING == 4 else "2"))   # also wait for this many later reads if they are already issued

    def run(self, ops):
        e = self.e
        n = len(ops)
        # LDS instances: a quad stays resident in one of the 4 ring slots until it is the least recently used
        # one when another quad needs a slot (the backward solve revisits quads that straddle two rows)
        inst_of = [None] * n
        insts = []  # dict(quad, first, last, slot, prev, issued)
        slots = [None] * NRING  # instance index resident in each slot
        for i, op in enumerate(ops):
            if op["src"] and op["src"][0] == "L":
                qd = op["src"][1] // 4
                hit = [k for k in slots if k is not None and insts[k]["quad"] == qd]
                if hit:
                    insts[hit[0]]["last"] = i
                    inst_of[i] = hit[0]
                    continue
                free = [sl for sl in range(NRING) if slots[sl] is None]
                sl = free[0] if free else min(range(NRING), key=lambda q: insts[slots[q]]["last"])
                insts.append(dict(quad=qd, first=i, last=i, slot=sl, prev=slots[sl], issued=None))
                slots[sl] = len(insts) - 1
                inst_of[i] = slots[sl]

        def issue(it):
            e("ds_read_b128", "v[%d:%d]" % (ring_base(it["slot"]), ring_base(it["slot"]) + 3), "v1",
              it["quad"] * 1024)
            it["issued"] = self.nds
            self.nds += 1

        next_inst = 0
        atemp = {}
        next_acc = 0  # next op index whose AGPR fetch has not been issued
        acc_rr = 0
        for i in range(n):
            # AGPR fetches for ops i .. i+la-1 (one temporary PAIR per fetch, la < N_ATP pairs in flight)
            while next_acc < n and next_acc < i + self.la:
                op = ops[next_acc]
                if op["src"] and op["src"][0] in ("A", "A2"):
                    t = V_AT + 2 * (acc_rr % N_ATP)
                    acc_rr += 1
                    e("v_accvgpr_read_b32", "v%d" % t, "a%d" % op["src"][1])
                    if op["src"][0] == "A2":
                        e("v_accvgpr_read_b32", "v%d" % (t + 1), "a%d" % op["src"][2])
                    atemp[next_acc] = t
                next_acc += 1
            # LDS instance reads whose first consumer is within the window and whose slot is free
            while next_inst < len(insts):
                it = insts[next_inst]
                prev = insts[it["prev"]] if it["prev"] is not None else None
                if it["first"] <= i + self.lds_ahead and (prev is None or prev["last"] < i):
                    issue(it)
                    next_inst += 1
                else:
                    break
            op = ops[i]
            if op["src"] is None:
                op["emit"](None)
            elif op["src"][0] in ("A", "A2"):
                op["emit"](atemp.pop(i))
            else:
                it = insts[inst_of[i]]
                if it["issued"] is None:  # slot was busy until now: fetch on demand (instances issue in order)
                    assert inst_of[i] == next_inst and (it["prev"] is None or insts[it["prev"]]["last"] < i)
                    issue(it)
                    next_inst += 1
                if it["first"] == i and it["issued"] > self.waited:
                    # one wait covers this read and, when it is already in flight, the next one too (LDS reads
                    # return in order): half as many s_waitcnt in the instruction stream
                    upto = min(self.nds - 1, it["issued"] + self.merge)
                    e("s_waitcnt", "lgkmcnt(%d)" % min(15, self.nds - 1 - upto))
                    self.waited = upto if self.nds - 1 - upto <= 15 else it["issued"]
                op["emit"](ring_base(it["slot"]) + op["src"][1] % 4)


# ---- packed (VOP3P) operand helpers: every operand is a 64-bit register pair plus a half select ----
def _vp(n):   # both halves of the aligned VGPR pair starting at even n
    assert n % 2 == 0
    return ("v[%d:%d]" % (n, n + 1), 0, 1)


def _vb(n):   # one VGPR broadcast to both halves
    lo = n - (n % 2)
    return ("v[%d:%d]" % (lo, lo + 1), n % 2, n % 2)


def _sb(n):   # SGPR constant (low half of an even pair) broadcast
    assert n % 2 == 0
    return ("s[%d:%d]" % (n, n + 1), 0, 0)


def pk(e, mnem, dst, srcs, neg=None):
    """srcs: list of (reg, sel_lo, sel_hi); neg: list of 0/1 per source (applied to both halves)."""
    assert dst % 2 == 0
    neg = neg or [0] * len(srcs)
    mods = dict(op_sel=[s_[1] for s_ in srcs], op_sel_hi=[s_[2] for s_ in srcs], neg_lo=list(neg), neg_hi=list(neg))
    e(mnem, "v[%d:%d]" % (dst, dst + 1), *[s_[0] for s_ in srcs], mods)


def despace(ops, window=int(os.environ.get("UMPC_ASM_WINDOW", "12"))):
    """EXPERIMENT, off by default (UMPC_ASM_DESPACE=1): reorders the op list so that an instruction does not read what
    the previous one (or the one before) wrote (back-to-back dependencies 124 -> 32 per iteration). Measured on the
    MI355X it is SLOWER (0.182 vs 0.176 ms per step): in the real loop the dependent-issue penalty seen in
    tools/microbench.hip is hidden behind the AGPR / LDS operand fetches, and the reordering costs fetch locality.
    Greedy, register-exact: an op may move ahead of earlier ops it has no RAW / WAR / WAW
    relation with, inside a window; among the movable ones the first that is independent of the last two issued wins."""
    pending, out = list(ops), []
    last = [frozenset(), frozenset()]   # registers written by the previous two issued ops
    while pending:
        best, bscore = 0, None
        blocked_w, blocked_r = set(), set()   # written / read by the earlier, still pending ops
        for k, o in enumerate(pending[:window]):
            movable = not (o["r"] & blocked_w) and not (o["w"] & blocked_w) and not (o["w"] & blocked_r)
            if movable:
                score = (2 if (o["r"] | o["w"]) & last[1] else 0) + (1 if (o["r"] | o["w"]) & last[0] else 0)
                if bscore is None or score < bscore:
                    best, bscore = k, score
                    if score == 0:
                        break
            blocked_w |= o["w"]
            blocked_r |= o["r"]
        o = pending.pop(best)
        out.append(o)
        last = [last[1], o["w"]]
    return out


def body(e, s, first, capture, plan, lv=False, delta_in_w=False, qzero=frozenset(), lzero=frozenset(), dy3_in_w=False):
    """delta_in_w (asmstep.py, the all-assembly step kernel): a capturing iteration leaves delta_x = x - x_prev in
    the x part of W and delta_y in the z part of W (registers) instead of writing x_prev / delta_y to the workspace.
    qzero / lzero: x indices whose q and dynamics rows whose l (= u) are STRUCTURALLY zero for this QP (asmstep.Struct:
    q is non-zero only on the y and dp entries, l on rows 0..5, 18..26 and 32): their right-hand side is sigma x /
    -y / rho without the AGPR read (8.6 cycles each for a lone wave) -- 35 of an iteration's 214 reads."""
    nx, nc, nk = s.nx, s.nc, s.nk
    neq = 2 * s.N * symbolic.NY
    xs, zs, xinv, zinv = slot_maps(s)
    fwd, bwd, lpos = plan

    def wreg(k):
        return w_reg(s, k, xs, zs)
    W = lambda k: "v%d" % wreg(k)
    WX = lambda j: V_W + xs[j]        # register of the KKT unknown paired with x_j
    WZ = lambda i: V_WZ + zs[i]       # ... with constraint row i
    XR = lambda j: V_X + xs[j]
    YR = lambda i: V_Y + zs[i]
    ZR = lambda i: V_Z + zs[i]
    X = lambda j: "v%d" % XR(j)
    Y = lambda i: "v%d" % YR(i)
    Z = lambda i: "v%d" % ZR(i)
    v = lambda n: "v%d" % n
    sA, sO, sS, sRi, sRh = ("s%d" % r for r in (S_ALPHA, S_OMA, S_SIGMA, S_RINV, S_RHO))
    ptr = "s[%d:%d]" % (S_P2, S_P2 + 1)
    f = Fetcher(e)
    ops = []

    def op(src, fn, w=(), r=()):
        """w / r: the VGPRs this op writes / reads besides its fetched operand (for despace())."""
        ops.append(dict(src=src, emit=fn, w=frozenset(w), r=frozenset(r)))

    assert NVZ == neq and not (lv and first)

    def l_src(j):
        if lpos[j] < NLDS:
            return ("L", lpos[j])
        if lv and lpos[j] < NLDS + NVZ:   # resident in the z registers of the dynamics rows (z == l there, kept in AGPRs)
            return ("V", V_Z + lpos[j] - NLDS)
        if NLDS + NVZ <= lpos[j] < NLDS + NVZ + XV_COUNT:
            return ("V", XV_BASE + lpos[j] - NLDS - NVZ)
        return ("A", A_L + lpos[j] - NLDS)

    if capture and not delta_in_w:  # x_prev of this iteration -> workspace
        _row_ptr(e, S_P2, S_WS, WS_XPREV)
        for j in range(nx):
            e("global_store_dword", "v0", X(j), ptr)
            _adv(e, S_P2)
    # ---- rhs: W = [sigma x - q ; z - y / rho]  (auxil.c:164-178), two slots per instruction
    for p_ in range(0, nx - 1, 2):
        j0, j1 = xinv[p_], xinv[p_ + 1]
        rw = dict(w=(V_W + p_, V_W + p_ + 1), r=(V_X + p_, V_X + p_ + 1))
        if j0 in qzero and j1 in qzero:
            op(None, lambda t, p_=p_: pk(e, "v_pk_mul_f32", V_W + p_, [_sb(S_SIGMA), _vp(V_X + p_)]), **rw)
        elif j0 in qzero or j1 in qzero:
            jn, h = (j1, 1) if j0 in qzero else (j0, 0)

            def mixed(t, p_=p_, h=h):
                pk(e, "v_pk_mul_f32", V_W + p_, [_sb(S_SIGMA), _vp(V_X + p_)])
                e("v_sub_f32", v(V_W + p_ + h), v(V_W + p_ + h), v(t))
            op(("A", A_Q + jn), mixed, **rw)
        else:
            op(("A2", A_Q + j0, A_Q + j1),
               lambda t, p_=p_: pk(e, "v_pk_fma_f32", V_W + p_, [_sb(S_SIGMA), _vp(V_X + p_), _vp(t)], [0, 0, 1]), **rw)
    if nx % 2:
        jl = xinv[nx - 1]
        if jl in qzero:
            op(None, lambda t, jl=jl: e("v_mul_f32", v(WX(jl)), sS, X(jl)), w=(WX(jl),), r=(XR(jl),))
        else:
            op(("A", A_Q + jl), lambda t, jl=jl: e("v_fma_f32", v(WX(jl)), sS, X(jl), "-" + v(t)), w=(WX(jl),), r=(XR(jl),))
    assert all(zinv[p_] < neq for p_ in range(neq)) and neq % 2 == 0
    for p_ in range(0, neq, 2):
        if lv:   # z of the dynamics rows is l (== u): read it from its AGPR home
            i0, i1 = zinv[p_], zinv[p_ + 1]
            rw = dict(w=(V_WZ + p_, V_WZ + p_ + 1), r=(V_Y + p_, V_Y + p_ + 1))
            if i0 in lzero and i1 in lzero:
                op(None, lambda t, p_=p_: pk(e, "v_pk_mul_f32", V_WZ + p_, [_sb(S_RINV), _vp(V_Y + p_)], [1, 0]), **rw)
            elif i0 in lzero or i1 in lzero:
                inz, h = (i1, 1) if i0 in lzero else (i0, 0)

                def mixed_z(t, p_=p_, h=h):
                    pk(e, "v_pk_mul_f32", V_WZ + p_, [_sb(S_RINV), _vp(V_Y + p_)], [1, 0])
                    e("v_add_f32", v(V_WZ + p_ + h), v(V_WZ + p_ + h), v(t))
                op(("A", A_LO + inz), mixed_z, **rw)
            else:
                op(("A2", A_LO + i0, A_LO + i1),
                   lambda t, p_=p_: pk(e, "v_pk_fma_f32", V_WZ + p_, [_sb(S_RINV), _vp(V_Y + p_), _vp(t)], [1, 0, 0]), **rw)
        else:
            op(None, lambda t, p_=p_: pk(e, "v_pk_fma_f32", V_WZ + p_, [_sb(S_RINV), _vp(V_Y + p_), _vp(V_Z + p_)], [1, 0, 0]),
               w=(V_WZ + p_, V_WZ + p_ + 1), r=(V_Y + p_, V_Y + p_ + 1, V_Z + p_, V_Z + p_ + 1))
    for i in range(neq, nc):
        op(("A", A_M + 9 + i - neq), lambda t, i=i: e("v_fma_f32", v(WZ(i)), "-" + v(t), Y(i), Z(i)),
           w=(WZ(i),), r=(YR(i), ZR(i)))

    # ---- triangular solves (qdldl.c:250-277) from the list schedule: dst -= L * src, packed where both the
    # destination and the source registers of two ready entries form aligned pairs
    def solve_ops(sched):
        for g in sched:
            if len(g) == 1:
                d, sr, j = g[0]
                src = l_src(j)
                # the factor is stored NEGATED: dst += (-L) * src is the 4-byte VOP2 v_fmac_f32, which a lone wave
                # issues every ~4.4 cycles against ~5.0 for the 8-byte VOP3 v_fma_f32 (tools/microbench.hip)
                if src[0] == "V":
                    op(None, lambda t, d=d, sr=sr, r=src[1]: e("v_fmac_f32", W(d), v(r), W(sr)),
                       w=(wreg(d),), r=(wreg(d), wreg(sr)))
                else:
                    op(src, lambda t, d=d, sr=sr: e("v_fmac_f32", W(d), v(t), W(sr)),
                       w=(wreg(d),), r=(wreg(d), wreg(sr)))
                continue
            (d0, s0_, j0), (d1, s1_, j1) = g          # d0 has the even destination register
            rd, r0, r1 = wreg(d0), wreg(s0_), wreg(s1_)
            assert rd % 2 == 0 and wreg(d1) == rd + 1 and r0 // 2 == r1 // 2 and lpos[j0] // 2 == lpos[j1] // 2
            srcp = ("v[%d:%d]" % (r0 - r0 % 2, r0 - r0 % 2 + 1), r0 % 2, r1 % 2)
            rw = dict(w=(rd, rd + 1), r=(rd, rd + 1, r0, r1))
            pe = lpos[j0] - lpos[j0] % 2
            lsel = (lpos[j0] % 2, lpos[j1] % 2)
            if pe < NLDS:
                op(("L", pe), lambda t, rd=rd, srcp=srcp, lsel=lsel:
                   pk(e, "v_pk_fma_f32", rd, [("v[%d:%d]" % (t, t + 1), lsel[0], lsel[1]), srcp, _vp(rd)], [0, 0, 0]), **rw)
            elif (lv and pe < NLDS + NVZ) or NLDS + NVZ <= pe < NLDS + NVZ + XV_COUNT:
                t = V_Z + pe - NLDS if pe < NLDS + NVZ else XV_BASE + pe - NLDS - NVZ
                op(None, lambda _t, t=t, rd=rd, srcp=srcp, lsel=lsel:
                   pk(e, "v_pk_fma_f32", rd, [("v[%d:%d]" % (t, t + 1), lsel[0], lsel[1]), srcp, _vp(rd)], [0, 0, 0]), **rw)
            else:
                src = ("A2", A_L + lpos[j0] - NLDS, A_L + lpos[j1] - NLDS)
                op(src, lambda t, rd=rd, srcp=srcp: pk(e, "v_pk_fma_f32", rd, [_vp(t), srcp, _vp(rd)], [0, 0, 0]), **rw)

    solve_ops(fwd)
    # ---- diagonal (qdldl.c:289): two unknowns per instruction, paired by register
    kof = {wreg(k): k for k in range(nk)}
    for r0 in range(V_W, V_Z, 2):
        k0, k1 = kof.get(r0), kof.get(r0 + 1)
        if k0 is not None and k1 is not None:
            op(("A2", A_D + k0, A_D + k1), lambda t, r0=r0: pk(e, "v_pk_mul_f32", r0, [_vp(r0), _vp(t)]),
               w=(r0, r0 + 1), r=(r0, r0 + 1))
        elif k0 is not None or k1 is not None:
            k, r = (k0, r0) if k0 is not None else (k1, r0 + 1)
            op(("A", A_D + k), lambda t, r=r: e("v_mul_f32", v(r), v(t), v(r)), w=(r,), r=(r,))
    solve_ops(bwd)
    f.run(despace(ops) if os.environ.get("UMPC_ASM_DESPACE", "0") == "1" else ops)
    # ---- x <- alpha x~ + (1 - alpha) x   (auxil.c:188-201)
    for p_ in range(0, nx - 1, 2):
        t = V_TT + 2 * ((p_ // 2) % 4)
        pk(e, "v_pk_mul_f32", t, [_sb(S_OMA), _vp(V_X + p_)])
        if capture and delta_in_w:   # x_new in t, delta_x = x_new - x_prev into W, then x <- x_new
            pk(e, "v_pk_fma_f32", t, [_sb(S_ALPHA), _vp(V_W + p_), _vp(t)])
            pk(e, "v_pk_add_f32", V_W + p_, [_vp(t), _vp(V_X + p_)], [0, 1])
            e("v_pk_mov_b32", "v[%d:%d]" % (V_X + p_, V_X + p_ + 1), "v[%d:%d]" % (t, t + 1), "v[%d:%d]" % (t, t + 1),
              dict(op_sel=[0, 1], op_sel_hi=[0, 0], neg_lo=[0, 0], neg_hi=[0, 0]))
        else:
            pk(e, "v_pk_fma_f32", V_X + p_, [_sb(S_ALPHA), _vp(V_W + p_), _vp(t)])
    if nx % 2:
        jl = xinv[nx - 1]
        e("v_mul_f32", v(V_TT), sO, X(jl))
        if capture and delta_in_w:
            e("v_fma_f32", v(V_TT), sA, v(WX(jl)), v(V_TT))
            e("v_sub_f32", v(WX(jl)), v(V_TT), X(jl))
            e("v_mov_b32", X(jl), v(V_TT))
        else:
            e("v_fma_f32", X(jl), sA, v(WX(jl)), v(V_TT))
    # ---- z, y  (auxil.c:203-228, qdldl_interface.c:364-366, proj.c:4-14)
    if capture and first and not delta_in_w:
        _row_ptr(e, S_P2, S_WS, WS_DY)
    if not first:
        # Dynamics rows after the first iteration: z == l == u, so z stays and
        #   delta_y = rho (alpha z~ + (1-alpha) z - z) = rho alpha (z~ - z) = rho alpha rinv (nu - y) = alpha (nu - y)
        # (rho rinv = 1). Two packed instructions per two rows instead of seven per row; same value up to the
        # rounding of the longer chain.
        for p_ in range(0, neq, 2):
            t = V_TT + 2 * ((p_ // 2) % 2)
            pk(e, "v_pk_add_f32", t, [_vp(V_WZ + p_), _vp(V_Y + p_)], [0, 1])
            if capture and delta_in_w:   # delta_y = alpha (nu - y) stays in the z part of W
                pk(e, "v_pk_mul_f32", V_WZ + p_, [_sb(S_ALPHA), _vp(t)])
            elif capture:
                pk(e, "v_pk_mul_f32", t + 4, [_sb(S_ALPHA), _vp(t)])
                for h in range(2):
                    _row_ptr(e, S_P2, S_WS, WS_DY + zinv[p_ + h])
                    e("global_store_dword", "v0", v(t + 4 + h), ptr)
            pk(e, "v_pk_fma_f32", V_Y + p_, [_sb(S_ALPHA), _vp(t), _vp(V_Y + p_)])
    for i in range(nc):
        eq = i < neq
        if eq and not first:
            continue
        if first and eq and i % 16 == 0:
            # l == u of the dynamics rows sit in spare AGPRs; 16 at a time into the (idle) ring registers
            for w in range(min(16, neq - i)):
                e("v_accvgpr_read_b32", v(V_RING + w), "a%d" % (A_LO + i + w))
        b = V_TT + 4 * (i % 2)
        t1, t2, t3, t4 = v(b), v(b + 1), v(b + 2), v(b + 3)
        if eq:
            rinv, rho = sRi, sRh
        else:  # thrust rows: their own rho / bounds, kept in AGPRs (a: lo3 up3 rho3 rinv3)
            k = i - neq
            rinv, rho = v(V_AT), v(V_AT + 1)
            e("v_accvgpr_read_b32", rinv, "a%d" % (A_M + 9 + k))
            e("v_accvgpr_read_b32", rho, "a%d" % (A_M + 6 + k))
            e("v_accvgpr_read_b32", v(V_AT + 2), "a%d" % (A_M + k))
            e("v_accvgpr_read_b32", v(V_AT + 3), "a%d" % (A_M + 3 + k))
        nu = v(WZ(i))
        e("v_fma_f32", t1, "-" + rinv, Y(i), Z(i))        # z - y/rho (the rhs again)
        e("v_fma_f32", t1, rinv, nu, t1)                  # z~
        e("v_mul_f32", t2, sO, Z(i))
        e("v_fma_f32", t1, sA, t1, t2)                    # t = alpha z~ + (1-alpha) z
        if eq:
            zn = v(V_RING + i % 16)
            e("v_sub_f32", t2, t1, zn)   # z <- l: from here on z of this row is read from its AGPR (A_LO)
        else:
            e("v_fma_f32", t3, rinv, Y(i), t1)
            e("v_max_f32", t3, t3, v(V_AT + 2))
            e("v_min_f32", Z(i), t3, v(V_AT + 3))
            e("v_sub_f32", t2, t1, Z(i))
        if dy3_in_w and not eq and not capture:
            # (asmstep.py, round 5) the thrust rows' delta_y lands in their W register -- nu has been consumed above -- in EVERY
            # iteration, at no cost, so that the last middle iteration can stand in for a capturing one
            e("v_mul_f32", nu, rho, t2)
            e("v_add_f32", Y(i), Y(i), nu)
            continue
        e("v_mul_f32", t2, rho, t2)                       # delta_y
        e("v_add_f32", Y(i), Y(i), t2)
        if capture and delta_in_w:
            e("v_mov_b32", nu, t2)
        elif capture:
            if not first:
                _row_ptr(e, S_P2, S_WS, WS_DY + i)
            e("global_store_dword", "v0", t2, ptr)
            if first:
                _adv(e, S_P2)


def program(N=3, perm=None):
    s = symbolic.analyse(N, perm)
    plan = solve_plan(s)
    e = Emit()
    prologue(e, s)
    e("s_cmp_lt_i32", "s%d" % S_ITERS, 1)
    e("s_cbranch_scc1", "9f")
    body(e, s, first=True, capture=True, plan=plan)
    # the z registers of the dynamics rows are free now (z == l): they take the L entries of storage positions
    # NLDS .. NLDS+NVZ-1, read twice per iteration, out of the AGPRs
    for p_ in range(NVZ):
        e("v_accvgpr_read_b32", "v%d" % (V_Z + p_), "a%d" % (A_L + p_))
    e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_ITERS, 2)
    e("s_cmp_lt_i32", "s%d" % S_CNT, 1)
    e("s_cbranch_scc1", "8f")
    e("label", "7")
    body(e, s, first=False, capture=False, plan=plan, lv=True)
    e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_CNT, 1)
    e("s_cmp_gt_i32", "s%d" % S_CNT, 0)
    e("s_cbranch_scc1", "7b")
    e("label", "8")
    e("s_cmp_lt_i32", "s%d" % S_ITERS, 2)
    e("s_cbranch_scc1", "6f")
    body(e, s, first=False, capture=True, plan=plan, lv=True)
    e("label", "6")
    epilogue(e, s, z_is_l=True)
    e("s_branch", "5f")
    e("label", "9")
    epilogue(e, s, z_is_l=False)
    e("label", "5")
    return e.ins, s


def fmt(t):
    m = t[0]
    if m == "label":
        return "%s:" % t[1]
    mods = ""
    if isinstance(t[-1], dict):
        d = t[-1]
        t = t[:-1]
        mods = " " + " ".join("%s:[%s]" % (k, ",".join(map(str, d[k]))) for k in ("op_sel", "op_sel_hi", "neg_lo", "neg_hi"))
    a = [("0x%x" % x if (m == "s_mov_b32" and isinstance(x, int)) else str(x)) for x in t[1:]]
    if m in ("ds_read_b128", "ds_write_b128", "ds_write_b32"):
        return "%s %s, %s offset:%s" % (m, a[0], a[1], a[2])
    if m in ("global_load_dword", "global_store_dword"):
        return "%s %s, %s, %s" % (m, a[0], a[1], a[2])
    if m == "s_waitcnt":
        return "s_waitcnt " + " ".join(a)
    return "%s %s%s" % (m, ", ".join(a), mods)


def write(path=None, N=3, perm=None):
    path = path or os.path.join(HERE, "csrc", "umpc_admm_asm.h")
    ins, s = program(N, perm)
    lpos = solve_plan(s)[2]
    store_l = " ".join("LDSW_(%d) = -LX_(%d);" % (lpos[j], j) if lpos[j] < NLDS else "ROW_(%d) = -LX_(%d);" % (FAC_L + lpos[j], j)
                       for j in range(len(lpos)))
    used_s = [S_P, S_P + 1, S_CNT, S_P2, S_P2 + 1] + list(range(S_ALPHA, S_RHO + 2))
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in range(2, V_END)] + \
           ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in used_s]
    lab7 = [k for k, t in enumerate(ins) if t == ("label", "7")][0]
    lab8 = [k for k, t in enumerate(ins) if t == ("label", "8")][0]
    out = ["// GENERATED by robobee3d_amd/asmgen.py -- do not edit.", switch_banner(),
           "// ADMM phase of the fp32 step kernel: %d instructions, middle-iteration body %d." % (len(ins), lab8 - lab7),
           "#pragma once",
           "namespace umpcasm {",
           "constexpr int FAC_L = %d, FAC_DI = %d, FAC_Q = %d, FAC_LOEQ = %d, FAC_M = %d, FAC_ROWS = %d;" %
           (FAC_L, FAC_DI, FAC_Q, FAC_LOEQ, FAC_M, FAC_ROWS),
           "constexpr int WS_DS = %d, WS_ES = %d, WS_C = %d, WS_XPREV = %d, WS_DY = %d, WS_ROWS = %d;" %
           (WS_DS, WS_ES, WS_C, WS_XPREV, WS_DY, WS_ROWS),
           "constexpr int LDS_BYTES_PER_LANE = %d;" % (NLDS * 4),
           "}  // namespace umpcasm",
           "// Phase A -> loop hand-off of the factor: entry j of L (CSC order) goes NEGATED to its storage position (asmgen.solve_plan):",
           "// LDS word LDSW_(p) for p < %d, workspace row ROW_(FAC_L + p) (-> AGPR) otherwise." % NLDS,
           "#define UMPC_ASM_STORE_L(LDSW_, ROW_, LX_) do { %s } while (0)" % store_l,
           "// inputs: v0 = 4*robot, v1 = lane LDS address, s[4:5] = workspace, s[6:7] = ctrl, s10 = 4*B, s11 = maxIter",
           "#define UMPC_ADMM_ASM(voff, ldsaddr, ws, ctrl, stride, iters) asm volatile( \\"]
    for t in ins:
        out.append('  "%s\\n" \\' % fmt(t))
    out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{s[4:5]}"(ws), "{s[6:7]}"(ctrl), "{s10}"(stride), "{s11}"(iters) \\')
    out.append("  : " + ", ".join(clob) + ")")
    txt = "\n".join(out) + "\n"
    old = open(path).read() if os.path.exists(path) else None
    if old != txt:
        with open(path, "w") as fh:
            fh.write(txt)
    return path, len(ins)


# ---------------------------------------------------------------------------
# CPU interpreter of the emitted instruction list (one lane), for the tests
# ---------------------------------------------------------------------------
def simulate(ins, mem_ws, mem_ctrl, iters, lds=None):
    """mem_ws: float32[WS_ROWS], mem_ctrl: float32[127], lds: float32[160] (one robot / one lane).
    Runs the program; memories are updated in place (lds holds L[0..160) on entry, x,y,z on exit).
    Branch targets are the numeric local labels used above. Returns the executed instruction count."""
    import numpy as np
    f32 = np.float32
    V = np.zeros(256, f32)
    A = np.zeros(256, f32)
    S = {}
    if lds is None:
        lds = np.zeros(40 * 4, f32)
    scc = 0
    labels = {}
    for k, t in enumerate(ins):
        if t[0] == "label":
            labels.setdefault(t[1], []).append(k)

    def sval(x):
        if isinstance(x, int):
            return x
        if x.startswith("s["):
            lo = int(x[2:x.index(":")])
            return S.get(lo, 0) | (S.get(lo + 1, 0) << 32)
        return S.get(int(x[1:]), 0)

    def sf(n):
        return np.frombuffer(struct.pack("<I", S[n] & 0xFFFFFFFF), f32)[0]

    def fval(x):
        neg = x.startswith("-")
        if neg:
            x = x[1:]
        if x[0] == "v":
            val = V[int(x[1:])]
        elif x[0] == "s":
            val = sf(int(x[1:]))
        else:
            raise ValueError(x)
        return -val if neg else val

    def half(x, sel):  # one half of a 64-bit packed operand
        lo = int(x[2:x.index(":")])
        return V[lo + sel] if x[0] == "v" else sf(lo + sel)

    STRIDE = 4096
    S[S_WS], S[S_WS + 1] = 1 << 20, 0          # workspace "address"
    S[S_CTRL], S[S_CTRL + 1] = 1 << 30, 0
    S[S_STRIDE], S[S_ITERS] = STRIDE, iters

    def mem(addr):
        if addr >= (1 << 30):
            return mem_ctrl, (addr - (1 << 30)) // STRIDE
        return mem_ws, (addr - (1 << 20)) // STRIDE

    pc = 0
    nexec = 0
    while pc < len(ins):
        t = ins[pc]
        m = t[0]
        nexec += 1
        assert nexec < 400000, "runaway program"
        if m == "label" or m == "s_waitcnt":
            pass
        elif m == "s_mov_b32":
            S[int(t[1][1:])] = t[2] if isinstance(t[2], int) else sval(t[2])
        elif m == "s_mov_b64":
            lo = int(t[1][2:t[1].index(":")])
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
        elif m == "s_branch" or m == "s_cbranch_scc1":
            if m == "s_branch" or scc:
                lab, d = t[1][:-1], t[1][-1]
                cands = labels[lab]
                pc = min(c for c in cands if c > pc) if d == "f" else max(c for c in cands if c < pc)
        elif m == "global_load_dword":
            arr, row = mem(sval(t[3]))
            if t[1][0] == "a":
                A[int(t[1][1:])] = arr[row]
            else:
                V[int(t[1][1:])] = arr[row]
        elif m == "global_store_dword":
            arr, row = mem(sval(t[3]))
            arr[row] = V[int(t[2][1:])]
        elif m == "ds_write_b128":
            lo = int(t[2][2:t[2].index(":")])
            lds[t[3] // 1024 * 4:t[3] // 1024 * 4 + 4] = V[lo:lo + 4]
        elif m == "ds_write_b32":
            lds[t[3] // 1024 * 4 + (t[3] % 1024) // 4] = V[int(t[2][1:])]
        elif m == "ds_read_b128":
            lo = int(t[1][2:t[1].index(":")])
            V[lo:lo + 4] = lds[t[3] // 1024 * 4:t[3] // 1024 * 4 + 4]
        elif m == "v_accvgpr_read_b32":
            V[int(t[1][1:])] = A[int(t[2][1:])]
        elif m == "v_mov_b32":
            V[int(t[1][1:])] = fval(t[2])
        elif m == "v_fma_f32":
            V[int(t[1][1:])] = f32(np.float64(fval(t[2])) * np.float64(fval(t[3])) + np.float64(fval(t[4])))
        elif m == "v_fmac_f32":
            V[int(t[1][1:])] = f32(np.float64(fval(t[2])) * np.float64(fval(t[3])) + np.float64(V[int(t[1][1:])]))
        elif m == "v_mul_f32":
            V[int(t[1][1:])] = f32(fval(t[2]) * fval(t[3]))
        elif m == "v_add_f32":
            V[int(t[1][1:])] = f32(fval(t[2]) + fval(t[3]))
        elif m == "v_sub_f32":
            V[int(t[1][1:])] = f32(fval(t[2]) - fval(t[3]))
        elif m == "v_max_f32":
            V[int(t[1][1:])] = max(fval(t[2]), fval(t[3]))
        elif m == "v_min_f32":
            V[int(t[1][1:])] = min(fval(t[2]), fval(t[3]))
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
            V[dlo], V[dlo + 1] = res  # both halves are computed from the OLD register contents
        else:
            raise ValueError("unknown instruction %r" % (t,))
        pc += 1
    return nexec


if __name__ == "__main__":
    p, n = write()
    print("wrote", p, n, "instructions")

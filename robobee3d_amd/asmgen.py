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
    v2..v85       W      KKT rhs / solution (permuted order)
    v86..v130     x      v131..v169  y      v170..v208  z
    v209..v220    lo3 up3 rho3 rinv3 (thrust rows)
    v222..v237    LDS read ring (4 x float4)
    v238..v243    AGPR read temporaries
    v244..v251    arithmetic temporaries
    v221, v252..v255   not touched (left to the compiler for values that live across the block)
    a0..a52       L[160..212]    a53..a136  1/D    a137..a181  q    a182..a217  l(=u) of the dynamics rows
    LDS           L[0..159] as 40 float4 per lane (ds_read_b128, conflict-free)

Per middle iteration: 84 rhs FMAs, 426 solve FMAs + 84 multiplies, 90 x-update
and ~260 z/y-update ops, 235 v_accvgpr_read and 80 ds_read_b128 -- no scratch,
no HBM traffic. The arithmetic (operation order, where an FMA replaces a
multiply-add) is identical to UMPC_GEN_ADMM_ITER in umpc_gen.h, which stays the
fp64 / reference implementation of the same iteration.

Reference mapping: auxil.c:164-228 (compute_rhs, update_x, update_z, update_y),
qdldl_interface.c:322-369, qdldl.c:250-293, proj.c:4-14.

`simulate()` interprets the emitted instruction list on numpy float32 so that
the CPU test-suite can check the schedule (register reuse, fetch distances)
against the oracle without a GPU.
"""
import os
import struct

from . import symbolic

HERE = os.path.dirname(os.path.abspath(__file__))

NLDS = 160  # L entries kept in LDS
# workspace rows shared with the C++ phases (see umpc_step.h)
FAC_L, FAC_DI, FAC_Q, FAC_LOEQ, FAC_M = 0, 213, 297, 342, 378
FAC_ROWS = 390
WS_DS, WS_ES, WS_C, WS_XPREV, WS_DY = 390, 435, 474, 475, 520
WS_ROWS = 559

V_W, V_X, V_Y, V_Z, V_M = 2, 86, 131, 170, 209
V_RING, V_AT, V_TT = 222, 238, 244   # v221 and v252..v255 are left to the compiler (SGPR spill lanes)
N_AT = 6
A_L, A_D, A_Q, A_LO = 0, 53, 137, 182
S_WS, S_CTRL, S_STRIDE, S_ITERS = 4, 6, 10, 11
S_P, S_CNT, S_P2 = 12, 14, 16
S_ALPHA, S_OMA, S_SIGMA, S_RINV, S_RHO = 20, 21, 22, 23, 24


def f32bits(v):
    return struct.unpack("<I", struct.pack("<f", v))[0]


class Emit:
    def __init__(self):
        self.ins = []  # tuples (mnemonic, operands...)

    def __call__(self, *t):
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


def prologue(e, s):
    e("s_mov_b32", "s%d" % S_ALPHA, f32bits(1.6))
    import numpy as np
    e("s_mov_b32", "s%d" % S_OMA, f32bits(float(np.float32(1.0) - np.float32(1.6))))
    e("s_mov_b32", "s%d" % S_SIGMA, f32bits(1e-6))
    e("s_mov_b32", "s%d" % S_RINV, f32bits(0.01))
    e("s_mov_b32", "s%d" % S_RHO, f32bits(100.0))
    # Two memory round trips in all (every wave of the grid is in this phase at the same time, so each
    # exposed round trip costs microseconds): (1) L[0..NLDS) staged through v2..v161 (W/x/y are not live
    # yet) into LDS; (2) everything else, back to back, one wait.
    # L[0..NLDS) was written into LDS by phase A (same lane, same layout); everything else arrives through
    # the workspace rows, issued back to back with ONE wait.
    _row_ptr(e, S_P, S_WS, FAC_L + NLDS)
    # the rest of L, 1/D, q and the dynamics-row bounds -> AGPRs (rows are consecutive in the workspace)
    nrest = len(s.L_i) - NLDS + s.nk + s.nx + 2 * s.N * symbolic.NY
    assert A_LO == nrest - 2 * s.N * symbolic.NY and nrest <= 256
    for r in range(nrest):
        e("global_load_dword", "a%d" % r, "v0", "s[%d:%d]" % (S_P, S_P + 1))
        _adv(e, S_P)
    # thrust-row words (rows follow FAC_LOEQ)
    for k in range(12):
        e("global_load_dword", "v%d" % (V_M + k), "v0", "s[%d:%d]" % (S_P, S_P + 1))
        _adv(e, S_P)
    # x, y, z
    e("s_mov_b64", "s[%d:%d]" % (S_P, S_P + 1), "s[%d:%d]" % (S_CTRL, S_CTRL + 1))
    for r in range(s.nx + 2 * s.nc):
        e("global_load_dword", "v%d" % (V_X + r), "v0", "s[%d:%d]" % (S_P, S_P + 1))
        _adv(e, S_P)
    e("s_waitcnt", "vmcnt(0)")


def epilogue(e, s):
    # x, y, z stay on chip: the factor in LDS is dead now, phase C reads the iterates from LDS words 0..122
    # (v86..v208 are consecutive; the last quad also carries v209, which nobody reads)
    n = s.nx + 2 * s.nc
    for g in range((n + 3) // 4):
        e("ds_write_b128", "v1", "v[%d:%d]" % (V_X + 4 * g, V_X + 4 * g + 3), g * 1024)
    e("s_waitcnt", "vmcnt(0) lgkmcnt(0)")


class Fetcher:
    """Issues the L / 1-over-D / q operand fetches a fixed distance ahead of their consumers."""

    def __init__(self, e, la=4):
        self.e, self.la = e, la
        self.nds = 0          # ds_reads issued so far in this body

    def run(self, ops):
        """ops: list of dicts {emit: fn(src_reg_name), src: ('A', areg) | ('L', ldsword) | None}"""
        e = self.e
        n = len(ops)
        # LDS instances: a quad stays resident in one of the 4 ring slots until it is the least recently used
        # one when another quad needs a slot (the backward solve revisits quads that straddle two rows)
        inst_of = [None] * n
        insts = []  # dict(quad, first, last, slot, prev, issued)
        slots = [None] * 4  # instance index resident in each slot
        for i, op in enumerate(ops):
            if op["src"] and op["src"][0] == "L":
                qd = op["src"][1] // 4
                hit = [k for k in slots if k is not None and insts[k]["quad"] == qd]
                if hit:
                    insts[hit[0]]["last"] = i
                    inst_of[i] = hit[0]
                    continue
                free = [sl for sl in range(4) if slots[sl] is None]
                sl = free[0] if free else min(range(4), key=lambda q: insts[slots[q]]["last"])
                insts.append(dict(quad=qd, first=i, last=i, slot=sl, prev=slots[sl], issued=None))
                slots[sl] = len(insts) - 1
                inst_of[i] = slots[sl]

        def issue(it):
            e("ds_read_b128", "v[%d:%d]" % (V_RING + 4 * it["slot"], V_RING + 4 * it["slot"] + 3), "v1",
              it["quad"] * 1024)
            it["issued"] = self.nds
            self.nds += 1

        next_inst = 0
        atemp = {}
        next_acc = 0  # next op index whose AGPR fetch has not been issued
        acc_rr = 0
        for i in range(n):
            # AGPR fetches for ops i .. i+la-1
            while next_acc < n and next_acc < i + self.la:
                op = ops[next_acc]
                if op["src"] and op["src"][0] == "A":
                    t = V_AT + (acc_rr % N_AT)
                    acc_rr += 1
                    e("v_accvgpr_read_b32", "v%d" % t, "a%d" % op["src"][1])
                    atemp[next_acc] = t
                next_acc += 1
            # LDS instance reads whose first consumer is within the window and whose slot is free
            while next_inst < len(insts):
                it = insts[next_inst]
                prev = insts[it["prev"]] if it["prev"] is not None else None
                if it["first"] <= i + 3 * self.la and (prev is None or prev["last"] < i):
                    issue(it)
                    next_inst += 1
                else:
                    break
            op = ops[i]
            if op["src"] is None:
                op["emit"](None)
            elif op["src"][0] == "A":
                op["emit"]("v%d" % atemp.pop(i))
            else:
                it = insts[inst_of[i]]
                if it["issued"] is None:  # slot was busy until now: fetch on demand (instances issue in order)
                    assert inst_of[i] == next_inst and (it["prev"] is None or insts[it["prev"]]["last"] < i)
                    issue(it)
                    next_inst += 1
                if it["first"] == i:
                    e("s_waitcnt", "lgkmcnt(%d)" % min(15, self.nds - 1 - it["issued"]))
                op["emit"]("v%d" % (V_RING + 4 * it["slot"] + op["src"][1] % 4))


def l_src(eidx):
    return ("L", eidx) if eidx < NLDS else ("A", A_L + eidx - NLDS)


def body(e, s, first, capture):
    nx, nc, nk = s.nx, s.nc, s.nk
    neq = 2 * s.N * symbolic.NY
    W = lambda k: "v%d" % (V_W + k)
    X = lambda j: "v%d" % (V_X + j)
    Y = lambda i: "v%d" % (V_Y + i)
    Z = lambda i: "v%d" % (V_Z + i)
    M = lambda k: "v%d" % (V_M + k)  # lo3[0:3] up3[3:6] rho3[6:9] rinv3[9:12]
    sA, sO, sS, sRi, sRh = ("s%d" % r for r in (S_ALPHA, S_OMA, S_SIGMA, S_RINV, S_RHO))
    ptr = "s[%d:%d]" % (S_P2, S_P2 + 1)
    f = Fetcher(e)
    ops = []

    def op(src, fn):
        ops.append(dict(src=src, emit=fn))

    if capture:  # x_prev of this iteration -> workspace
        _row_ptr(e, S_P2, S_WS, WS_XPREV)
        for j in range(nx):
            e("global_store_dword", "v0", X(j), ptr)
            _adv(e, S_P2)
    # ---- rhs: W = [sigma x - q ; z - y / rho]  (auxil.c:164-178), written in permuted order
    for j in range(nx):
        op(("A", A_Q + j), lambda r, j=j: e("v_fma_f32", W(s.pinv[j]), sS, X(j), "-" + r))
    for i in range(nc):
        rinv = sRi if i < neq else M(9 + i - neq)
        op(None, lambda r, i=i, rinv=rinv: e("v_fma_f32", W(s.pinv[nx + i]), "-" + rinv, Y(i), Z(i)))
    # ---- forward substitution (qdldl.c:250-262)
    for c in range(nk):
        for j in range(s.L_p[c], s.L_p[c + 1]):
            r_ = s.L_i[j]
            op(l_src(j), lambda r, r_=r_, c=c: e("v_fma_f32", W(r_), "-" + r, W(c), W(r_)))
    # ---- diagonal (qdldl.c:289)
    for k in range(nk):
        op(("A", A_D + k), lambda r, k=k: e("v_mul_f32", W(k), r, W(k)))
    # ---- backward substitution (qdldl.c:265-277)
    for c in range(nk - 1, -1, -1):
        for j in range(s.L_p[c], s.L_p[c + 1]):
            r_ = s.L_i[j]
            op(l_src(j), lambda r, r_=r_, c=c: e("v_fma_f32", W(c), "-" + r, W(r_), W(c)))
    f.run(ops)
    # ---- x <- alpha x~ + (1 - alpha) x   (auxil.c:188-201)
    for j in range(nx):
        t = "v%d" % (V_TT + j % 8)
        e("v_mul_f32", t, sO, X(j))
        e("v_fma_f32", X(j), sA, W(s.pinv[j]), t)
    # ---- z, y  (auxil.c:203-228, qdldl_interface.c:364-366, proj.c:4-14)
    if capture:
        _row_ptr(e, S_P2, S_WS, WS_DY)
    for i in range(nc):
        eq = i < neq
        if first and eq and i % 16 == 0:
            # l == u of the dynamics rows sit in spare AGPRs; 16 at a time into the (idle) ring registers
            for w in range(min(16, neq - i)):
                e("v_accvgpr_read_b32", "v%d" % (V_RING + w), "a%d" % (A_LO + i + w))
        b = V_TT + 4 * (i % 2)
        t1, t2, t3 = "v%d" % b, "v%d" % (b + 1), "v%d" % (b + 2)
        rinv = sRi if eq else M(9 + i - neq)
        rho = sRh if eq else M(6 + i - neq)
        nu = W(s.pinv[nx + i])
        if eq and not first:
            # Dynamics rows after the first iteration: z == l == u, so z stays and
            #   delta_y = rho (alpha z~ + (1-alpha) z - z) = rho alpha (z~ - z) = rho alpha rinv (nu - y) = alpha (nu - y)
            # (rho rinv = 1). Two instructions instead of seven; same value up to the rounding of the longer chain.
            e("v_sub_f32", t1, nu, Y(i))
            if capture:
                e("v_mul_f32", t2, sA, t1)
                e("global_store_dword", "v0", t2, ptr)
                _adv(e, S_P2)
            e("v_fma_f32", Y(i), sA, t1, Y(i))
            continue
        e("v_fma_f32", t1, "-" + rinv, Y(i), Z(i))        # z - y/rho (the rhs again)
        e("v_fma_f32", t1, rinv, nu, t1)                  # z~
        e("v_mul_f32", t2, sO, Z(i))
        e("v_fma_f32", t1, sA, t1, t2)                    # t = alpha z~ + (1-alpha) z
        if eq:
            zn = ("v%d" % (V_RING + i % 16)) if first else Z(i)
            e("v_sub_f32", t2, t1, zn)
            if first:
                e("v_mov_b32", Z(i), zn)
        else:
            k = i - neq
            e("v_fma_f32", t3, rinv, Y(i), t1)
            e("v_max_f32", t3, t3, M(k))
            e("v_min_f32", Z(i), t3, M(3 + k))
            e("v_sub_f32", t2, t1, Z(i))
        e("v_mul_f32", t2, rho, t2)                       # delta_y
        e("v_add_f32", Y(i), Y(i), t2)
        if capture:
            e("global_store_dword", "v0", t2, ptr)
            _adv(e, S_P2)


def program(N=3, perm=None):
    s = symbolic.analyse(N, perm)
    e = Emit()
    prologue(e, s)
    e("s_cmp_lt_i32", "s%d" % S_ITERS, 1)
    e("s_cbranch_scc1", "9f")
    body(e, s, first=True, capture=True)
    e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_ITERS, 2)
    e("s_cmp_lt_i32", "s%d" % S_CNT, 1)
    e("s_cbranch_scc1", "8f")
    e("label", "7")
    body(e, s, first=False, capture=False)
    e("s_sub_i32", "s%d" % S_CNT, "s%d" % S_CNT, 1)
    e("s_cmp_gt_i32", "s%d" % S_CNT, 0)
    e("s_cbranch_scc1", "7b")
    e("label", "8")
    e("s_cmp_lt_i32", "s%d" % S_ITERS, 2)
    e("s_cbranch_scc1", "9f")
    body(e, s, first=False, capture=True)
    e("label", "9")
    epilogue(e, s)
    return e.ins, s


def fmt(t):
    m = t[0]
    if m == "label":
        return "%s:" % t[1]
    a = [("0x%x" % x if (m == "s_mov_b32" and isinstance(x, int)) else str(x)) for x in t[1:]]
    if m in ("ds_read_b128", "ds_write_b128"):
        return "%s %s, %s offset:%s" % (m, a[0], a[1], a[2])
    if m == "global_load_dword":
        return "%s %s, %s, %s" % (m, a[0], a[1], a[2])
    if m == "global_store_dword":
        return "%s %s, %s, %s" % (m, a[0], a[1], a[2])
    if m == "s_waitcnt":
        return "s_waitcnt " + " ".join(a)
    return "%s %s" % (m, ", ".join(a))


def write(path=None, N=3, perm=None):
    path = path or os.path.join(HERE, "csrc", "umpc_admm_asm.h")
    ins, s = program(N, perm)
    used_s = [S_P, S_P + 1, S_CNT, S_P2, S_P2 + 1, S_ALPHA, S_OMA, S_SIGMA, S_RINV, S_RHO]
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in range(2, 252) if i != 221] + ['"a%d"' % i for i in range(256)] + \
           ['"s%d"' % i for i in used_s]
    out = ["// GENERATED by robobee3d_amd/asmgen.py -- do not edit.",
           "// ADMM phase of the fp32 step kernel: %d instructions, middle-iteration body %d." %
           (len(ins), sum(1 for _ in ins) // 3),
           "#pragma once",
           "namespace umpcasm {",
           "constexpr int FAC_L = %d, FAC_DI = %d, FAC_Q = %d, FAC_LOEQ = %d, FAC_M = %d, FAC_ROWS = %d;" %
           (FAC_L, FAC_DI, FAC_Q, FAC_LOEQ, FAC_M, FAC_ROWS),
           "constexpr int WS_DS = %d, WS_ES = %d, WS_C = %d, WS_XPREV = %d, WS_DY = %d, WS_ROWS = %d;" %
           (WS_DS, WS_ES, WS_C, WS_XPREV, WS_DY, WS_ROWS),
           "constexpr int LDS_BYTES_PER_LANE = %d;" % (NLDS * 4),
           "}  // namespace umpcasm",
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

    def fval(x):
        neg = x.startswith("-")
        if neg:
            x = x[1:]
        if x[0] == "v":
            v = V[int(x[1:])]
        elif x[0] == "s":
            v = np.frombuffer(struct.pack("<I", S[int(x[1:])] & 0xFFFFFFFF), f32)[0]
        else:
            raise ValueError(x)
        return -v if neg else v

    def ptr_row(x):  # 's[a:b]' -> (which memory, row): pointers are simulated as base_id * 2^40 + row * stride
        return sval(x)

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
            v = sval(t[2])
            S[lo], S[lo + 1] = v & 0xFFFFFFFF, v >> 32
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
        elif m == "s_cbranch_scc1":
            if scc:
                lab, d = t[1][:-1], t[1][-1]
                cands = labels[lab]
                pc = min(c for c in cands if c > pc) if d == "f" else max(c for c in cands if c < pc)
        elif m == "global_load_dword":
            arr, row = mem(ptr_row(t[3]))
            if t[1][0] == "a":
                A[int(t[1][1:])] = arr[row]
            else:
                V[int(t[1][1:])] = arr[row]
        elif m == "global_store_dword":
            arr, row = mem(ptr_row(t[3]))
            arr[row] = V[int(t[2][1:])]
        elif m == "ds_write_b128":
            lo = int(t[2][2:t[2].index(":")])
            lds[t[3] // 1024 * 4:t[3] // 1024 * 4 + 4] = V[lo:lo + 4]
        elif m == "ds_read_b128":
            lo = int(t[1][2:t[1].index(":")])
            V[lo:lo + 4] = lds[t[3] // 1024 * 4:t[3] // 1024 * 4 + 4]
        elif m == "v_accvgpr_read_b32":
            V[int(t[1][1:])] = A[int(t[2][1:])]
        elif m == "v_mov_b32":
            V[int(t[1][1:])] = fval(t[2])
        elif m == "v_fma_f32":
            V[int(t[1][1:])] = f32(np.float64(fval(t[2])) * np.float64(fval(t[3])) + np.float64(fval(t[4])))
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
        else:
            raise ValueError("unknown instruction %r" % (t,))
        pc += 1
    return nexec


if __name__ == "__main__":
    p, n = write()
    print("wrote", p, n, "instructions")

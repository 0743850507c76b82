"""Straight-line specialisations of the general batch QP kernel (csrc/umpc_bqp.hip) for structures known at build
time: every index of the table-driven kernel becomes a literal, every per-robot word a scalar the compiler can
keep in a register or a fixed scratch slot and schedule freely. Same arithmetic, same order as
`bqp_solve_kernel` (so fp64 results are bit-identical to the table-driven path; tests/test_bqp.py checks it).

Emits one translation unit per (structure, dtype) under csrc/gen/ (compiled in parallel by _lib.build) and the
registry header csrc/umpc_bqp_registry.h keyed by the FNV-1a hash of the structure's table blob; umpcQPCreate picks
the specialisation when the blob it is handed matches (umpcQPUseTables switches back). Built-in structures: planar p5f N = 10, the v1
template QP N = 3, UprightMPC2 N = 5.
"""
import os

from . import batchqp, qpstruct

HERE = os.path.dirname(os.path.abspath(__file__))


def fnv1a(words):
    h = 0xcbf29ce484222325
    for w in words:
        v = int(w) & 0xffffffff
        for k in range(4):
            h ^= (v >> (8 * k)) & 0xff
            h = (h * 0x100000001b3) & 0xffffffffffffffff
    return h


def builtin_structures():
    out = []
    out.append(("p5f10", batchqp.p5f_analysis(10)[1]))   # (the labelling and elimination order PlanarP5fMPC uses)
    st = batchqp.v1_structure(3)
    out.append(("v1n3", qpstruct.analyse_qp(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"])))
    st = batchqp.uprightmpc2_structure(5)
    out.append(("umpc2n5", qpstruct.analyse_qp(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"])))
    return out


ASM_STRUCTURES = {"p5f10": list(range(77))}   # structure -> rows assumed to be equalities by the assembly loop (asmqp.py)
ASM_STREAM_ROW = 1024                          # the hand-off rows of an assembly specialisation stay below this row
ASM_STREAM_ITEMS = 2048                        # items of a wave's stream block: the loop's stream, then the residual stream
ASM_RES_ITEM0 = 600                            # first item of the residual stream (asmqp.ResPlan)
ASM_GROUP_WAVES = 4                            # wavefronts per workgroup of an assembly specialisation (asmqp.RuizSplit)


def emit_structure(name, s, asm=None):
    """asm: an asmqp.Plan -> emits bqp_fixed_<name>_asm (fp32 only): the middle ADMM iterations run as the generated
    assembly (csrc/gen/bqp_<name>_asm.h), the first and the last one in C++"""
    n, m, nk = s.n, s.m, s.nk
    t = s.tables
    pinv, pidx, A_p, A_i = t["pinv"], t["pidx"], t["A_p"], t["A_i"]
    Ar_p, Ar_j, Ar_k = t["Ar_p"], t["Ar_j"], t["Ar_k"]
    L_p, L_i, Lr_p, Lr_j, Lr_k = t["L_p"], t["L_i"], t["Lr_p"], t["Lr_j"], t["Lr_k"]
    o = []
    E = o.append
    E("template <typename T>")
    if asm:
        E("// wave = this group of 64 robots (its stream block), lane = the robot's lane, wv = which of the workgroup's %d wavefronts runs" % ASM_GROUP_WAVES)
        E("// (all of them own the SAME 64 robots and LDS slots: asmqp.RuizSplit; only wavefront 0 goes past the Ruiz block)")
        E("__device__ __forceinline__ void bqp_fixed_%s_asm(const QPArgs<T> &a, const int b, const int wave, const unsigned lane, const unsigned wv, const unsigned ldsaddr, const float *ldsf) {" % name)
    else:
        E("__device__ __forceinline__ void bqp_fixed_%s(const QPArgs<T> &a, const int b) {" % name)
    E("  const size_t B = (size_t)a.B;")
    E("#define IN(arr, i) (arr)[(size_t)(i) * B + b]")
    E("  const T sigma = a.sigma, alpha = a.alpha, oma = T(1.0) - a.alpha;")
    if s.nnzP:
        E("  T Ps[%d];" % s.nnzP)
    E("  T As[%d], qs[%d], D[%d], Ev[%d], Dt[%d], Et[%d], rho[%d], rinv[%d], ls[%d], us[%d];" %
      (s.nnzA, n, n, m, n, m, m, m, m, m))
    E("  T Lx[%d], DI[%d], yv[%d], w[%d], x[%d], y[%d], z[%d], xp[%d], dy[%d], t3[%d], t1[%d];" %
      (s.nnzL, nk, nk, nk, n, m, m, n, m, m, n))
    load_lines = ["  Ps[%d] = IN(a.Pv, %d);" % (k, k) for k in range(s.nnzP)] + \
                 ["  As[%d] = IN(a.Av, %d);" % (k, k) for k in range(s.nnzA)] + \
                 ["  qs[%d] = IN(a.q, %d);" % (j, j) for j in range(n)]
    if not asm:
        o.extend(load_lines)      # (the assembly variant's Ruiz block fetches Pv, Av, q itself)
    E("  const T rho_eq = T(QP_RHO_EQ_OVER_RHO_INEQ * (double)a.rho);")
    TIMING = asm is not None and os.environ.get("UMPC_QP_TIMING") == "1"   # diagnostic builds: phase intervals -> info rows
    mark = (lambda k: E("  tmark[%d] = __builtin_amdgcn_s_memrealtime();" % k)) if TIMING else (lambda k: None)
    if TIMING:
        E("  long long tmark[8];")
    if asm:
        # the warm start and the bounds arrive through LDS: asmqp.loader_program fetches three [row][B] arrays in one round
        # trip (hipcc would fetch the ~900 words one exposed load at a time: 0.45 ms of the tick)
        E("#define LDSQ(w) ldsf[((w) >> 2) * 256 + ((w) & 3)]")
        E("  auto uni = [](unsigned long long v_) { return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v_ >> 32)) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)v_); };  // (the builtin returns int: no sign extension)")
        E("  const unsigned voff = (unsigned)b * 4u, s_stride = __builtin_amdgcn_readfirstlane((unsigned)a.B * 4u);")
        E("  T *const sblk = a.S + (size_t)wave * %d;   // this wave's stream block ([item][lane]); wave-uniform (SGPR) base" % (ASM_STREAM_ITEMS * 64))
        E("  const unsigned long long ssp = a.asm_ok ? uni((unsigned long long)sblk) : 0ull;")
        E("  const unsigned lane4 = lane * 4u;")
        emit_fast_route(E, name, s, asm, TIMING, mark)
        if not TIMING:
            # (leave here: hipcc computes and spills ~700 row addresses of the routes below in their common dominator, which
            # would otherwise be this point -- 1 000 scratch stores per tick that a settled wave never reads)
            E("  if (mode == 2) return;")
        E("  if (mode == 0) {   // the general route: C++ glue around the assembly blocks, or all C++")
        E("  BQP_%s_LOAD_XYZ(voff, ldsaddr, uni((unsigned long long)a.x), uni((unsigned long long)a.y), uni((unsigned long long)a.z), s_stride);" % name.upper())
        for j in range(n):
            E("  D[%d] = T(1.0); x[%d] = LDSQ(%d);" % (j, j, j))
        for i in range(m):
            E("  Ev[%d] = T(1.0); y[%d] = LDSQ(%d); z[%d] = LDSQ(%d);" % (i, i, n + i, i, n + m + i))
        E("  BQP_%s_LOAD_LUE(voff, ldsaddr, uni((unsigned long long)a.l), uni((unsigned long long)a.u), uni((unsigned long long)a.Eprev), s_stride);" % name.upper())
        for i in range(m):
            E("  { const T e = LDSQ(%d); qp_classify(LDSQ(%d) * e, LDSQ(%d) * e, a.rho, rho_eq, rho[%d], rinv[%d]); }"
              % (2 * m + i, i, m + i, i, i))
    else:
        for j in range(n):
            E("  D[%d] = T(1.0); x[%d] = IN(a.x, %d);" % (j, j, j))
        for i in range(m):
            E("  { const T e = IN(a.Eprev, %d); qp_classify(IN(a.l, %d) * e, IN(a.u, %d) * e, a.rho, rho_eq, rho[%d], rinv[%d]); "
              "Ev[%d] = T(1.0); y[%d] = IN(a.y, %d); z[%d] = IN(a.z, %d); }" % (i, i, i, i, i, i, i, i, i, i))
    mark(0)
    # ---- Ruiz
    E("  c = T(1.0);" if asm else "  T c = T(1.0);")
    if asm:
        RP = asm.ruiz
        E("  // scaling.c:44-156 as generated assembly (asmqp.ruiz_program): the block fetches Pv, Av, q in batches, keeps the row")
        E("  // scalings in VGPRs, Dt / P / q in AGPRs, A and the accumulated D, E in LDS, and leaves everything in LDS")
        E("  bool rs_valid = false;   // the residual stream is written (by the Ruiz block; the residual block needs it)")
        E("  if (a.scaling >= 1) {")
        E("    const unsigned s_pass = __builtin_amdgcn_readfirstlane((unsigned)a.scaling);")
        E("    const unsigned long long avp = uni((unsigned long long)a.Av), pvp = uni((unsigned long long)a.Pv), qvp = uni((unsigned long long)a.q);")
        E("    if (a.asm_ok) { BQP_%s_RUIZ_RS_ASM(voff, ldsaddr, lane4, avp, pvp, qvp, ssp, s_stride, s_pass); rs_valid = true; }" % name.upper())
        E("    else BQP_%s_RUIZ_ASM(voff, ldsaddr, avp, pvp, qvp, s_stride, s_pass);" % name.upper())
        for k in range(s.nnzA):
            E("    As[%d] = LDSQ(%d);" % (k, RP.LW_A + k))
        for k in range(s.nnzP):
            E("    Ps[%d] = LDSQ(%d);" % (k, RP.LW_P + k))
        for j in range(n):
            E("    qs[%d] = LDSQ(%d); D[%d] = LDSQ(%d);" % (j, RP.LW_Q + j, j, RP.LW_D + j))
        for i in range(m):
            E("    Ev[%d] = LDSQ(%d);" % (i, RP.LW_EV + i))
        E("    c = LDSQ(%d);" % RP.LW_C)
        E("  } else {")
        o.extend("  " + ln for ln in load_lines)
        E("  }")
    E("  for (int pass = 0; pass < (%s); ++pass) {" % ("0" if asm else "a.scaling"))
    for j in range(n):
        terms = "T(0.0)"
        dP = "qmax(qabs(Ps[%d]), T(0.0))" % pidx[j] if pidx[j] >= 0 else "T(0.0)"
        for p in range(A_p[j], A_p[j + 1]):
            terms = "qmax(qabs(As[%d]), %s)" % (p, terms)
        E("    Dt[%d] = T(1.0) / qsqrt(limit_scaling(qmax(%s, %s)));" % (j, dP, terms))
    for i in range(m):
        terms = "T(0.0)"
        for p in range(Ar_p[i], Ar_p[i + 1]):
            terms = "qmax(qabs(As[%d]), %s)" % (Ar_k[p], terms)
        E("    Et[%d] = T(1.0) / qsqrt(limit_scaling(%s));" % (i, terms))
    E("    T qn = T(0.0), csum = T(0.0);")
    for j in range(n):
        if pidx[j] >= 0:
            E("    { T pv = Ps[%d]; pv *= Dt[%d]; pv *= Dt[%d]; Ps[%d] = pv; csum += qabs(pv); }" % (pidx[j], j, j, pidx[j]))
        else:
            E("    csum += T(0.0);")
        for p in range(A_p[j], A_p[j + 1]):
            E("    { T v = As[%d]; v *= Et[%d]; v *= Dt[%d]; As[%d] = v; }" % (p, A_i[p], j, p))
        E("    { const T qv = qs[%d] * Dt[%d]; qs[%d] = qv; qn = qmax(qabs(qv), qn); D[%d] = Dt[%d] * D[%d]; }" % (j, j, j, j, j, j))
    for i in range(m):
        E("    Ev[%d] = Et[%d] * Ev[%d];" % (i, i, i))
    E("    T ct = csum / T(%d);" % n)
    E("    qn = limit_scaling(qn); ct = qmax(ct, qn); ct = limit_scaling(ct); ct = T(1.0) / ct;")
    for k in range(s.nnzP):
        E("    Ps[%d] *= ct;" % k)
    for j in range(n):
        E("    qs[%d] *= ct;" % j)
    E("    c *= ct;")
    E("  }")
    E("  cinv = T(1.0) / c;" if asm else "  const T cinv = T(1.0) / c;")
    if asm:
        E("  BQP_%s_LOAD_LUE(voff, ldsaddr, uni((unsigned long long)a.l), uni((unsigned long long)a.u), uni((unsigned long long)a.Eprev), s_stride);" % name.upper())
        for i in range(m):
            E("  ls[%d] = LDSQ(%d) * Ev[%d]; us[%d] = LDSQ(%d) * Ev[%d]; IN(a.Eprev, %d) = Ev[%d];" % (i, i, i, i, m + i, i, i, i))
        E("#undef LDSQ")
        E("  if (a.asm_ok) {   // z of the equality rows (= their scaled bound) for the residual block")
        for i in sorted(asm.res.it_ls):
            E("    sblk[%d + lane] = ls[%d];" % (asm.res.it_ls[i] * 64, i))
        E("  }")
    else:
        for i in range(m):
            E("  ls[%d] = IN(a.l, %d) * Ev[%d]; us[%d] = IN(a.u, %d) * Ev[%d]; IN(a.Eprev, %d) = Ev[%d];" % (i, i, i, i, i, i, i, i))
    mark(1)
    FI = "  "
    if asm:
        P = asm
        # the assembly loop takes the rows in ASM_STRUCTURES[name] for equalities: checked here, per wave
        E("  static_assert(BQP_%s_ASM_STREAM_ITEMS <= 1024 && BQP_%s_ASM_ROWS <= %d, \"stream buffer / hand-off rows\");"
          % (name.upper(), name.upper(), ASM_STREAM_ROW))
        E("  bool eqok = a.asm_ok != 0;")
        for i in sorted(r["i"] for r in P.rows if r["eq"]):
            E("  eqok = eqok && (rho[%d] == rho_eq) && (ls[%d] == us[%d]);" % (i, i, i))
        E("  const int mid = a.max_iter - 2;")
        E("  const bool use_asm = mid >= 1 && __all(eqok);")
        E("  // the first iteration needs C++ only where the warm-start z of an equality row differs from its bound (the loop")
        E("  // takes z == l there); with constant bounds -- p5f -- it never does after the first call")
        E("  bool z0ok = true;")
        for i in sorted(r["i"] for r in P.rows if r["eq"]):
            E("  z0ok = z0ok && (z[%d] == ls[%d]);" % (i, i))
        E("  const bool asm_first = use_asm && __all(z0ok);")
    # ---- factor (the general route hands L and 1/D to the loop through the workspace rows; the loop block's own
    # factorisation, asmqp.prologue_fast, belongs to the all-assembly route)
    E("  fail = 0;" if asm else "  int fail = 0;")
    for k in range(nk):
        E(FI + "yv[%d] = T(0.0);" % k)
    for op in s.factor_ops:
        k = op["k"]
        for (bb, p) in op["init"]:
            E(FI + "yv[%d] = As[%d];" % (bb, s.K_src[p][1]))
        orig = s.perm[k]
        if orig < n:
            dk = "Ps[%d] + sigma" % pidx[orig] if pidx[orig] >= 0 else "sigma"
        else:
            dk = "-rinv[%d]" % (orig - n)
        E(FI + "{ T dk = %s;" % dk)
        for (cidx, upd, new) in op["elim"]:
            E(FI + "  { const T yc = yv[%d];" % cidx)
            for (j, row) in upd:
                E(FI + "    yv[%d] -= Lx[%d] * yc;" % (row, j))
            E(FI + "    const T lv = yc * DI[%d]; Lx[%d] = lv; dk -= yc * lv; yv[%d] = T(0.0); }" % (cidx, new, cidx))
        E(FI + "  if (dk == T(0.0)) fail = 1;")
        E(FI + "  DI[%d] = T(1.0) / dk; }" % k)
    mark(2)
    # ---- ADMM
    for j in range(n):
        E("  xp[%d] = x[%d];" % (j, j))
    for i in range(m):
        E("  dy[%d] = T(0.0);" % i)
    it_lines = []
    I = it_lines.append
    for j in range(n):
        I("    xp[%d] = x[%d]; w[%d] = sigma * xp[%d] - qs[%d];" % (j, j, pinv[j], j, j))
    for i in range(m):
        I("    t3[%d] = z[%d] - rinv[%d] * y[%d]; w[%d] = t3[%d];" % (i, i, i, i, pinv[n + i], i))
    for r in range(nk):
        for p in range(Lr_p[r], Lr_p[r + 1]):
            I("    w[%d] -= Lx[%d] * w[%d];" % (r, Lr_k[p], Lr_j[p]))
    for r in range(nk):
        I("    w[%d] *= DI[%d];" % (r, r))
    for r in range(nk - 1, -1, -1):
        for j in range(L_p[r], L_p[r + 1]):
            I("    w[%d] -= Lx[%d] * w[%d];" % (r, j, L_i[j]))
    for j in range(n):
        I("    x[%d] = alpha * w[%d] + oma * xp[%d];" % (j, pinv[j], j))
    for i in range(m):
        I("    { const T zt = t3[%d] + rinv[%d] * w[%d]; T zn = alpha * zt + oma * z[%d] + rinv[%d] * y[%d]; "
          "zn = qmin(qmax(zn, ls[%d]), us[%d]); const T d = rho[%d] * (alpha * zt + oma * z[%d] - zn); z[%d] = zn; "
          "dy[%d] = d; y[%d] = y[%d] + d; }" % (i, i, pinv[n + i], i, i, i, i, i, i, i, i, i, i, i))
    if not asm:
        E("#pragma nounroll")
        E("  for (int it = 0; it < a.max_iter; ++it) {")
        o.extend(it_lines)
        E("  }")
    else:
        P = asm
        E("  auto iterate = [&]() __attribute__((always_inline)) {")
        o.extend(it_lines)
        E("  };")
        E("  if (a.max_iter >= 1 && !asm_first) iterate();")
        mark(3)
        E("  if (use_asm) {")
        E("    // hand-off: negated L in the loop's storage order, 1/D, x, y, z of the inequality rows -> workspace rows;")
        E("    // one iteration's read-only words -> this wave's stream block, in consumption order (asmqp.Plan.stream)")
        for j, pos in sorted(P.lpos.items()):
            E("    IN(a.W, %d) = -Lx[%d];" % (P.R_L + pos, j))
        for k in range(nk):
            E("    IN(a.W, %d) = DI[%d];" % (P.R_DI + k, k))
        for j in range(n):
            E("    IN(a.W, %d) = x[%d];" % (P.R_X + j, j))
        for i in range(m):
            E("    IN(a.W, %d) = y[%d];" % (P.R_Y + i, i))
        for i, q in sorted(P.zpos.items()):
            E("    IN(a.W, %d) = z[%d];" % (P.R_Z + q, i))
        nst = P.n_stream + len(P.extra)
        src = {"rinv": "rinv[%d]", "l": "ls[%d]", "u": "us[%d]", "rho": "rho[%d]", "q": "qs[%d]"}
        for q, (what, i) in enumerate(P.stream + P.extra):
            E("    sblk[%d + lane] = %s;" % (q * 64, src[what] % i))
        mark(4)
        E("    {")

        E("      // float constants come straight from the kernel arguments (SGPRs): a value computed with float arithmetic lives")
        E("      // in a VGPR and hipcc fails to copy it back (\"illegal VGPR to SGPR copy\"), so 1 - alpha and 1/rho_eq are the host's")
        E("      const unsigned s_alpha = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.alpha));")
        E("      const unsigned s_oma = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.oma));")
        E("      const unsigned s_sigma = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.sigma));")
        E("      const unsigned s_rinveq = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.rinv_eq));")
        E("      // every scalar operand is made provably wave-uniform (the values are; the compiler cannot always see it)")
        E("      const unsigned long long wsp = uni((unsigned long long)a.W);")
        E("      const unsigned s_mid = __builtin_amdgcn_readfirstlane((unsigned)(asm_first ? mid + 1 : mid));")
        E("      BQP_%s_ASM(voff, ldsaddr, lane4, wsp, ssp, s_stride, s_mid, s_alpha, s_oma, s_sigma, s_rinveq, "
          "uni((unsigned long long)a.x), uni((unsigned long long)a.y), uni((unsigned long long)a.z), 0u%s);" % (name.upper(), RHO_ARGS))
        E("    }")
        mark(5)
        E("#define LDSQ(w) ldsf[((w) >> 2) * 256 + ((w) & 3)]")
        E("    // residuals, the termination test at the strict tolerances and the solution stores as generated assembly")
        E("    // (asmqp.res_program); it settles the wave only if every robot is SOLVED -- otherwise the C++ phase below runs")
        E("    if (rs_valid && a.sol_x && a.sol_y && a.status && a.info && __all(fail == 0)) {")
        E("      const unsigned s_epsa = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.eps_abs));")
        E("      const unsigned s_epsr = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.eps_rel));")
        E("      const unsigned s_maxit = __builtin_amdgcn_readfirstlane((unsigned)a.max_iter);")
        E("      BQP_%s_RES_ASM(voff, ldsaddr, lane4, ssp, s_stride, uni((unsigned long long)a.x), uni((unsigned long long)a.y), "
          "uni((unsigned long long)a.z), uni((unsigned long long)a.sol_x), uni((unsigned long long)a.sol_y), "
          "uni((unsigned long long)a.status), uni((unsigned long long)a.info), s_epsa, s_epsr, s_maxit, uni((unsigned long long)a.Eprev));" % name.upper())
        E("      resdone = __all(LDSQ(%d) == T(1.0));" % 639)
        E("    }")
        E("    if (!resdone) {")
        E("    // the loop left x, y, z of the inequality rows, x_prev and delta_y in LDS (float4-interleaved words)")
        for j in range(n):
            E("    x[%d] = LDSQ(%d);" % (j, P.LW_X + j))
        for i in range(m):
            E("    y[%d] = LDSQ(%d);" % (i, P.LW_Y + i))
        for r in P.rows:
            if r["eq"]:
                E("    z[%d] = ls[%d];" % (r["i"], r["i"]))
            else:
                E("    z[%d] = LDSQ(%d);" % (r["i"], P.LW_Z + P.zpos[r["i"]]))
        for j in range(n):
            E("    xp[%d] = LDSQ(%d);" % (j, P.LW_XP + j))
        for i in range(m):
            E("    dy[%d] = LDSQ(%d);" % (i, P.LW_DY + i))
        E("    }")
        E("#undef LDSQ")
        E("  } else {")
        E("#pragma nounroll")
        E("    for (int it = 1; it < a.max_iter - 1; ++it) iterate();")
        E("    if (a.max_iter >= 2) iterate();")
        E("  }")
    mark(6)
    # ---- residuals
    if asm:
        E("  }   // general route")
        emit_fast_route_reload(E, name, s, asm)
        E("  if (!resdone) {")
    E("  T pri_res = T(0.0), nz = T(0.0), nAx = T(0.0);")
    for i in range(m):
        E("  { T acc = T(0.0);")
        for p in range(Ar_p[i], Ar_p[i + 1]):
            E("    acc += As[%d] * x[%d];" % (Ar_k[p], Ar_j[p]))
        E("    t3[%d] = acc; const T einv = T(1.0) / Ev[%d]; pri_res = qmax(pri_res, qabs(einv * (acc - z[%d]))); "
          "nz = qmax(nz, qabs(einv * z[%d])); nAx = qmax(nAx, qabs(einv * acc)); }" % (i, i, i, i))
    E("  T dua_res = T(0.0), nq = T(0.0), nAty = T(0.0), nPx = T(0.0);")
    for j in range(n):
        E("  { T px = T(0.0);%s T aty = T(0.0);" % (" px += Ps[%d] * x[%d];" % (pidx[j], j) if pidx[j] >= 0 else ""))
        for p in range(A_p[j], A_p[j + 1]):
            E("    aty += As[%d] * y[%d];" % (p, A_i[p]))
        E("    const T dinv = T(1.0) / D[%d]; dua_res = qmax(dua_res, qabs(dinv * ((qs[%d] + px) + aty))); "
          "nq = qmax(nq, qabs(dinv * qs[%d])); nAty = qmax(nAty, qabs(dinv * aty)); nPx = qmax(nPx, qabs(dinv * px)); }" % (j, j, j))
    E("  dua_res = cinv * dua_res;")
    E("  const T dual_rel = qmax(qmax(nq, nAty), nPx) * cinv, prim_rel = qmax(nz, nAx);")
    # the infeasibility certificates (auxil.c:362-512) are consulted only when a residual test fails at the strict
    # tolerances: a wave whose robots all pass skips the ~1 500 operations (same results; with the defaults below every
    # certificate test of the status logic is false)
    E("  T norm_dy = T(0.0), ineq_lhs = T(0.0), nAtdy = T(0.0), norm_dx = T(0.0), qdx = T(0.0), nPdx = T(0.0);")
    E("  const bool cert = !((pri_res < a.eps_abs + a.eps_rel * prim_rel) && (dua_res < a.eps_abs + a.eps_rel * dual_rel));")
    E("  if (cert) {")
    for i in range(m):
        E("  { T d = dy[%d]; const bool up = (double)us[%d] > QP_INFTY * QP_MIN_SCALING, lo = (double)ls[%d] < -QP_INFTY * QP_MIN_SCALING; "
          "if (up) d = lo ? T(0.0) : qmin(d, T(0.0)); else if (lo) d = qmax(d, T(0.0)); dy[%d] = d; "
          "norm_dy = qmax(norm_dy, qabs(d * Ev[%d])); ineq_lhs += us[%d] * qmax(d, T(0.0)) + ls[%d] * qmin(d, T(0.0)); }" %
          (i, i, i, i, i, i, i))
    for j in range(n):
        E("  { T acc = T(0.0);")
        for p in range(A_p[j], A_p[j + 1]):
            E("    acc += As[%d] * dy[%d];" % (p, A_i[p]))
        E("    nAtdy = qmax(nAtdy, qabs(acc * (T(1.0) / D[%d]))); }" % j)
    for j in range(n):
        E("  { const T dx = x[%d] - xp[%d]; t1[%d] = dx; norm_dx = qmax(norm_dx, qabs(D[%d] * dx)); qdx += qs[%d] * dx; "
          "T pdx = T(0.0);%s nPdx = qmax(nPdx, qabs(pdx * (T(1.0) / D[%d]))); }" %
          (j, j, j, j, j, " pdx += Ps[%d] * dx;" % pidx[j] if pidx[j] >= 0 else "", j))
    E("  }")
    E("  int status = -10;")
    E("  if (((double)pri_res > QP_INFTY) || ((double)dua_res > QP_INFTY)) status = -7;")
    E("#pragma nounroll")
    E("  for (int approx = 0; approx < 2 && status == -10; ++approx) {")
    E("    const T k = approx ? T(10) : T(1);")
    E("    const T eps_abs = a.eps_abs * k, eps_rel = a.eps_rel * k, eps_pinf = a.eps_pinf * k, eps_dinf = a.eps_dinf * k;")
    E("    const bool prim_ok = pri_res < eps_abs + eps_rel * prim_rel;")
    E("    const bool dual_ok = dua_res < eps_abs + eps_rel * dual_rel;")
    E("    bool pinf = false, dinf = false;")
    E("    if (!prim_ok && norm_dy > eps_pinf && ineq_lhs < -eps_pinf * norm_dy) pinf = nAtdy < eps_pinf * norm_dy;")
    E("    if (!dual_ok && norm_dx > eps_dinf && qdx < -c * eps_dinf * norm_dx && nPdx < c * eps_dinf * norm_dx) {")
    E("      dinf = true;")
    E("      const T thr = eps_dinf * norm_dx;")
    for i in range(m):
        E("      { T acc = T(0.0);")
        for p in range(Ar_p[i], Ar_p[i + 1]):
            E("        acc += As[%d] * t1[%d];" % (Ar_k[p], Ar_j[p]))
        E("        acc = acc * (T(1.0) / Ev[%d]);" % i)
        E("        if ((((double)us[%d] < QP_INFTY * QP_MIN_SCALING) && (acc > thr)) || "
          "(((double)ls[%d] > -QP_INFTY * QP_MIN_SCALING) && (acc < -thr))) dinf = false; }" % (i, i))
    E("    }")
    E("    if (prim_ok && dual_ok) status = approx ? 2 : 1;")
    E("    else if (pinf) status = approx ? 3 : -3;")
    E("    else if (dinf) status = approx ? 4 : -4;")
    E("  }")
    E("  if (status == -10) status = -2;")
    E("  const bool bad = status == -3 || status == 3 || status == -4 || status == 4 || status == -7;")
    E("  const T qnan = std::numeric_limits<T>::quiet_NaN();")
    for j in range(n):
        E("  if (a.sol_x) IN(a.sol_x, %d) = bad ? qnan : x[%d] * D[%d]; IN(a.x, %d) = bad ? T(0.0) : x[%d];" % (j, j, j, j, j))
    for i in range(m):
        E("  if (a.sol_y) IN(a.sol_y, %d) = bad ? qnan : (y[%d] * Ev[%d]) * cinv; IN(a.y, %d) = bad ? T(0.0) : y[%d]; "
          "IN(a.z, %d) = bad ? T(0.0) : z[%d];" % (i, i, i, i, i, i, i))
    E("  if (a.status) a.status[b] = status;")
    E("  if (a.info) { IN(a.info, 0) = pri_res; IN(a.info, 1) = dua_res; IN(a.info, 2) = c; IN(a.info, 3) = fail ? T(1) : T(0); IN(a.info, 4) = T(a.max_iter); IN(a.info, 5) = T(0); }")
    if asm:
        E("  }")
    if TIMING:
        E("  tmark[7] = __builtin_amdgcn_s_memrealtime();")
        E("  if (a.info) { IN(a.info, 0) = T(tmark[1] - tmark[0]); IN(a.info, 1) = T(tmark[2] - tmark[1]); IN(a.info, 2) = T(tmark[3] - tmark[2]); "
          "IN(a.info, 3) = T(tmark[4] - tmark[3]); IN(a.info, 4) = T(tmark[5] - tmark[4]); IN(a.info, 5) = T(tmark[7] - tmark[5]); }")
    E("#undef IN")
    E("}")
    if asm:
        E("__global__ void __launch_bounds__(%d) bqp_fixed_%s_asm_kernel(const QPArgs<float> a) {" % (64 * ASM_GROUP_WAVES, name))
        E("  __shared__ float4 lds[160 * 64];   // the whole CU: 640 words per lane (asmqp.py)")
        E("  // %d wavefronts (one per SIMD of the CU) share the 64 robots of the workgroup: lane l of EVERY wavefront is robot" % ASM_GROUP_WAVES)
        E("  // 64 * blockIdx.x + l and addresses the same LDS slot")
        E("  const unsigned lane = threadIdx.x & 63u, wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));")
        E("  const int b = blockIdx.x * 64 + (int)lane;")
        E("  if (b >= a.B) return;              // (the same lanes in every wavefront: none of them is left without a lane)")
        if name == "p5f10":
            # the planar-p5f tick as the kernel's prologue (umpcP5fTick): getLin at the previous state (every wavefront
            # evaluates it for its lanes: the same function of the same numbers), then -- behind a barrier, because the
            # plant tick overwrites the state the others read -- wavefront 0 writes lin and advances the plant while all
            # four rewrite the state-dependent entries of A (a quarter each); a second barrier publishes A to the Ruiz
            # block's loads. Two launches fewer per tick.
            E("  if (a.tick_y) {")
            E("    const size_t Bz = (size_t)a.B;")
            E("    float o[5], yy[7];")
            E("    for (int i = 0; i < 7; ++i) yy[i] = a.tick_y[(size_t)i * Bz + b];")
            E("    p5f_getlin<float>(a.tick_u, yy[0], yy[3], o);")
            E("    __syncthreads();")
            E("    if (wv == 0u) {")
            E("      if (a.tick_lin) for (int i = 0; i < 5; ++i) a.tick_lin[(size_t)i * Bz + b] = o[i];")
            E("      p5f_plant_tick<float>(o, a.tick_u, a.tick_dt, yy);")
            E("      for (int i = 0; i < 7; ++i) a.tick_y[(size_t)i * Bz + b] = yy[i];")
            E("    }")
            E("    float *Aw = const_cast<float *>(a.Av);")
            E("    for (int k = (int)wv; k < a.tick_nnz; k += %d) {" % ASM_GROUP_WAVES)
            E("      const int sidx = a.tick_src[k];      // (wave-uniform: scalar loads)")
            E("      if (sidx >= 0) Aw[(size_t)k * Bz + b] = o[sidx] * a.tick_cst[k];")
            E("    }")
            E("    __syncthreads();")
            E("  }")
        E("  bqp_fixed_%s_asm<float>(a, b, (int)blockIdx.x, lane, wv, (unsigned)(size_t)(&lds[lane]), reinterpret_cast<const float *>(lds) + 4 * lane);" % name)
        E("}")
        return "\n".join(o) + "\n"
    E("template <typename T>")
    E("__global__ void __launch_bounds__(64) bqp_fixed_%s_kernel(const QPArgs<T> a) {" % name)
    E("  const int b = blockIdx.x * 64 + threadIdx.x;")
    E("  if (b >= a.B) return;")
    E("  bqp_fixed_%s<T>(a, b);" % name)
    E("}")
    return "\n".join(o) + "\n"


DTYPES = (("f32", "float"), ("f64", "double"))


# the three rho classes of update_rho_vec as scalar operands of the loop block (kernel arguments: host-evaluated floats)
RHO_ARGS = (", __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.rho)), "
            "__builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.rinv0)), "
            "__builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.rho_eq))")


def emit_fast_route(E, name, s, P, TIMING, mark):
    """The all-assembly route of the fp32 assembly variant: Ruiz block, glue block (asmqp.glue_program), the loop with its fast
    start (the block factorises), the residual block. Taken when the glue block finds every row of ASM_STRUCTURES[name] an
    equality whose warm-start z equals its bound; otherwise nothing the caller can see has been touched and the general
    route below runs. Sets mode: 0 not taken, 1 taken but the residual block did not settle the wave, 2 done."""
    from . import asmqp
    U = name.upper()
    E("  int mode = 0, fail = 0;")
    E("  T c = T(1.0), cinv = T(1.0);")
    E("  bool resdone = false;")
    E("  if (!(a.asm_ok && a.scaling >= 1) && wv != 0u) return;   // (the other routes are one wavefront's)")
    E("  if (a.asm_ok && a.scaling >= 1) {")
    mark(0)
    E("    const unsigned s_pass = __builtin_amdgcn_readfirstlane((unsigned)a.scaling);")
    E("    // the passes shared by the workgroup's wavefronts (asmqp.ruiz_group_program); ends behind a barrier")
    E("    BQP_%s_RUIZ_RS4_ASM(voff, ldsaddr, lane4, uni((unsigned long long)a.Av), uni((unsigned long long)a.Pv), "
      "uni((unsigned long long)a.q), ssp, s_stride, s_pass, wv);" % U)
    E("    BQP_%s_GLUE4_ASM(voff, ldsaddr, lane4, uni((unsigned long long)a.l), ssp, uni((unsigned long long)a.u), s_stride, "
      "uni((unsigned long long)a.Eprev), uni((unsigned long long)a.z), a.rho, T(1. / (double)a.rho), rho_eq, T(1. / (double)rho_eq), wv);" % U)
    mark(1)
    if TIMING:
        E("    tmark[2] = tmark[3] = tmark[4] = tmark[1];")
    E("    if (__all(LDSQ(%d) == T(1.0))) {" % asmqp.GLUE_FLAG)
    E("      const unsigned s_alpha = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.alpha));")
    E("      const unsigned s_oma = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.oma));")
    E("      const unsigned s_sigma = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.sigma));")
    E("      const unsigned s_rinveq = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.rinv_eq));")
    E("      const unsigned s_iters = __builtin_amdgcn_readfirstlane((unsigned)(a.max_iter - 1));")
    E("      // every inequality row of the wave a loose row (planar p5f: all 87 are): the loop variant that streams nothing per row")
    E("      // (the flags are LDS words of the 64 robots every wavefront of the workgroup owns: all of them decide alike)")
    E("      if (__all(LDSQ(%d) == T(1.0)))" % asmqp.LOOSE_FLAG)
    E("        BQP_%s_ASM_LOOSE4(voff, ldsaddr, lane4, uni((unsigned long long)a.W), ssp, s_stride, s_iters, s_alpha, s_oma, s_sigma, s_rinveq, "
      "uni((unsigned long long)a.x), uni((unsigned long long)a.y), uni((unsigned long long)a.z), 1u%s, wv);" % (U, RHO_ARGS))
    E("      else")
    E("        BQP_%s_ASM4(voff, ldsaddr, lane4, uni((unsigned long long)a.W), ssp, s_stride, s_iters, s_alpha, s_oma, s_sigma, s_rinveq, "
      "uni((unsigned long long)a.x), uni((unsigned long long)a.y), uni((unsigned long long)a.z), 1u%s, wv);" % (U, RHO_ARGS))
    mark(5)
    E("      fail = (LDSQ(%d) == T(0.0)) ? 1 : 0;   // a zero pivot of the block's factorisation (qdldl.c:221-224)" % asmqp.FAC_MIN)
    E("      if (a.sol_x && a.sol_y && a.status && a.info && __all(fail == 0)) {")
    E("        const unsigned s_epsa = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.eps_abs));")
    E("        const unsigned s_epsr = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a.eps_rel));")
    E("        const unsigned s_maxit = __builtin_amdgcn_readfirstlane((unsigned)a.max_iter);")
    E("        BQP_%s_RES4_ASM(voff, ldsaddr, lane4, ssp, s_stride, uni((unsigned long long)a.x), uni((unsigned long long)a.y), "
      "uni((unsigned long long)a.z), uni((unsigned long long)a.sol_x), uni((unsigned long long)a.sol_y), "
      "uni((unsigned long long)a.status), uni((unsigned long long)a.info), s_epsa, s_epsr, s_maxit, uni((unsigned long long)a.Eprev), wv);" % U)
    E("        resdone = __all(LDSQ(%d) == T(1.0));" % asmqp.RES_FLAG)
    E("      }")
    E("      mode = resdone ? 2 : 1;")
    E("    }")
    E("    if (wv != 0u) return;   // (a workgroup that does not take the all-assembly route: wavefront 0 alone goes on)")
    E("  }")


def emit_fast_route_reload(E, name, s, P):
    """mode 1 (rare: a robot of the wave is not SOLVED at the strict tolerances, or the caller wants no solution rows): the C++
    residual phase needs the equilibrated data and the iterates in registers; they are in the wave's streams and in LDS"""
    n, m = s.n, s.m
    res = P.res
    pos = {}
    for q, it in enumerate(P.stream + P.extra):
        pos.setdefault(it, q)
    E("  if (mode == 1) {")
    E("#define SB(item) sblk[(item) * 64 + lane]")
    for k in range(s.nnzA):
        E("    As[%d] = SB(%d);" % (k, res.it_A + k))
    for j in range(n):
        E("    D[%d] = SB(%d); qs[%d] = SB(%d);" % (j, res.it_d[j], j, res.it_q[j]))
        if j in res.it_p:
            E("    Ps[%d] = SB(%d);" % (res.pidx[j], res.it_p[j]))
    for i in range(m):
        E("    Ev[%d] = SB(%d); IN(a.Eprev, %d) = Ev[%d];" % (i, res.it_ev[i], i, i))
        if i in res.eq:
            E("    ls[%d] = us[%d] = SB(%d);" % (i, i, res.it_ls[i]))
        else:
            # (l E, u E as the glue block forms them: it leaves these items of a wave that takes the loose loop unwritten)
            E("    ls[%d] = IN(a.l, %d) * Ev[%d]; us[%d] = IN(a.u, %d) * Ev[%d];" % (i, i, i, i, i, i))
    E("    c = SB(%d); cinv = T(1.0) / c;" % res.it_c)
    E("#undef SB")
    E("#define LDSQ(w) ldsf[((w) >> 2) * 256 + ((w) & 3)]")
    for j in range(n):
        E("    x[%d] = LDSQ(%d); xp[%d] = LDSQ(%d);" % (j, P.LW_X + j, j, P.LW_XP + j))
    for i in range(m):
        E("    y[%d] = LDSQ(%d); dy[%d] = LDSQ(%d);" % (i, P.LW_Y + i, i, P.LW_DY + i))
    for r in P.rows:
        if r["eq"]:
            E("    z[%d] = ls[%d];" % (r["i"], r["i"]))
        else:
            E("    z[%d] = LDSQ(%d);" % (r["i"], P.LW_Z + P.zpos[r["i"]]))
    E("#undef LDSQ")
    E("  }")


def _stamp_clobbers():
    """(diagnostic builds, UMPC_QP_RES_STAMPS / UMPC_QP_RUIZ_STAMPS: the blocks keep 100 MHz stamps in s60..s81)"""
    on = any(os.environ.get("UMPC_QP_%s_STAMPS" % k_) == "1" for k_ in ("RES", "RUIZ", "LOOP"))
    return ['"s%d"' % i for i in range(60, 82)] if on else []


def glue_macro(name, ins, group=False):
    from . import asmqp
    clob = ['"memory"', '"scc"', '"vcc"'] + _stamp_clobbers() + ['"v%d"' % i for i in [2, 3] + list(range(9, asmqp.V_END))] + \
           ['"s%d"' % i for i in (asmqp.S_P, asmqp.S_P + 1)] + ['"s%d"' % (q + h) for q in asmqp.GLUE_PTRS for h in (0, 1)]
    out = ["// Glue between the Ruiz block and the loop (asmqp.glue_program): rho classification, scaled bounds, the loop's stream,",
           "// LDS word %d = 1 iff the wave may take the all-assembly route. %d instructions." % (asmqp.GLUE_FLAG, len(ins)),
           "// inputs: v0 = 4*robot, v1 = lane LDS address, v4 = 4*lane, s[4:5] / s[8:9] / s[24:25] / s[26:27] = l, u, Eprev, z rows,",
           "// s[6:7] = the wave's stream block, s10 = 4*B, v5..v8 = rho, 1/rho, rho_eq, 1/rho_eq (floats)",
           "#define BQP_%s_GLUE_ASM(voff, ldsaddr, lane4, lp, sblk, up, stride, ep, zp, rho0, rinv0, rhoeq, rinveq) asm volatile( \\" % name.upper()]
    if group:
        out = ["// The glue block SHARED by the %d wavefronts of a workgroup (asmqp.glue_group_program): each a quarter of the rows and of q,"
               % ASM_GROUP_WAVES,
               "// the two flag words combined with LDS float-min atomics between two barriers; s%d = the wavefront's index. %d instructions."
               % (asmqp.S_GWAVE, len(ins)),
               "#define BQP_%s_GLUE4_ASM(voff, ldsaddr, lane4, lp, sblk, up, stride, ep, zp, rho0, rinv0, rhoeq, rinveq, wave) asm volatile( \\" % name.upper()]
    for t_ in ins:
        out.append('  "%s\\n" \\' % asmqp.fmt(t_))
    out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{v4}"(lane4), "{s[4:5]}"(lp), "{s[6:7]}"(sblk), "{s[8:9]}"(up), "{s10}"(stride), '
               '"{s[24:25]}"(ep), "{s[26:27]}"(zp), "{v5}"(rho0), "{v6}"(rinv0), "{v7}"(rhoeq), "{v8}"(rinveq)%s \\'
               % (', "{s%d}"(wave)' % asmqp.S_GWAVE if group else ""))
    out.append("  : " + ", ".join(clob) + ")")
    return "\n".join(out) + "\n"


def asm_macro(name, ins, plan, loose=False, group=None):
    """csrc/gen/bqp_<name>_asm.h: the instruction stream of asmqp.program as one asm volatile statement
    (loose: the variant for waves whose inequality rows are all loose rows, asmqp.S_RIMIN; same interface, fast start only;
    group: an asmqp.LoopSplit -- the loose variant shared by the workgroup's wavefronts, asmqp.loop_group_program)"""
    from . import asmqp
    if group is not None:
        used_s = [asmqp.S_P, asmqp.S_P + 1, asmqp.S_CNT, asmqp.S_SP, asmqp.S_SP + 1, asmqp.S_RIMIN, asmqp.S_RHOMIN, asmqp.S_DLEAF]
        clob = ['"memory"', '"scc"', '"vcc"'] + _stamp_clobbers() + ['"v%d"' % i for i in [2, 3] + list(range(5, asmqp.V_END))] + \
               ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in used_s]
        nv = [sum(1 for w_ in group.varw if w_ == w) for w in range(group.active)]
        nk_ = [sum(1 for w_ in group.kw if w_ == w) for w in range(group.active)]
        out = ["// The %s variant SHARED by the workgroup's wavefronts (asmqp.loop_group_program / LoopSplit): the QP's connected" % ("LOOSE" if loose else "GENERAL"),
               "// components are independent QPs, dealt out to the wavefronts (%s of %d variables, %s of %d KKT unknowns); they meet at"
               % (" + ".join(map(str, nv)), len(group.varw), " + ".join(map(str, nk_)), len(group.kw)),
               "// %d barriers, none inside the loop; same LDS layout, disjoint words, bit-identical"
               % (sum(t_[0] == "s_barrier" for t_ in ins) // group.nw),
               "// words. s%d = the wavefront's index; the other inputs as BQP_%s_ASM_LOOSE. %d instructions."
               % (asmqp.S_LWAVE, name.upper(), len(ins)),
               "#define BQP_%s_ASM%s4(voff, ldsaddr, lane4, ws, sblk, stride, iters, alpha, oma, sigma, rinveq, xi, yi, zi, fast, rho0, rinv0, rhoeq, wave) asm volatile( \\" % (name.upper(), "_LOOSE" if loose else "")]
        for t_ in ins:
            out.append('  "%s\\n" \\' % asmqp.fmt(t_))
        out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{v4}"(lane4), "{s[4:5]}"(ws), "{s[6:7]}"(sblk), "{s10}"(stride), '
                   '"{s11}"(iters), "{s20}"(alpha), "{s21}"(oma), "{s22}"(sigma), "{s23}"(rinveq), "{s[24:25]}"(xi), "{s[26:27]}"(yi), '
                   '"{s[28:29]}"(zi), "{s30}"(fast), "{s31}"(rho0), "{s34}"(rinv0), "{s35}"(rhoeq), "{s%d}"(wave) \\' % asmqp.S_LWAVE)
        out.append("  : " + ", ".join(clob) + ")")
        return "\n".join(out) + "\n"
    used_s = [asmqp.S_P, asmqp.S_P + 1, asmqp.S_CNT, asmqp.S_SP, asmqp.S_SP + 1] + ([asmqp.S_RIMIN, asmqp.S_RHOMIN, asmqp.S_DLEAF] if loose else [])
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in [2, 3] + list(range(5, asmqp.V_END))] + \
           ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in used_s]
    lab7 = [k for k, t_ in enumerate(ins) if t_ == ("label", "7")][0]
    from . import asmgen
    out = ["// GENERATED by robobee3d_amd/asmqp.py via codegen_qp.py -- do not edit.", asmgen.switch_banner(),
           "// %s ADMM iterations of the %s structure, fp32, one lane per robot, one wave per CU: %d instructions, %d"
           % ("LOOSE variant (every inequality row a loose row: nothing streamed per row, no clipping) of the" if loose else "Middle",
              name, len(ins), sum(1 for t_ in ins[lab7:] if t_[0] != "label")),
           "// from the loop label on (loop body + epilogue). Stream block: %d items per iteration + %d loaded once."
           % (plan.n_stream, len(plan.extra))] + ([] if loose else [
           "#pragma once",
           "constexpr int BQP_%s_ASM_STREAM_ITEMS = %d, BQP_%s_ASM_ROWS = %d;" % (name.upper(), plan.n_stream + len(plan.extra),
                                                                          name.upper(), plan.R_END)]) + [
           "// inputs: v0 = 4*robot, v1 = lane LDS address, v4 = 4*lane, s[4:5] = row workspace, s[6:7] = the wave's stream",
           "// block, s10 = 4*B, s11 = iterations (>= 1), s20..s23 = alpha, 1 - alpha, sigma, 1/rho_eq (float bits);",
           "// s30 != 0: fast start (asmqp.prologue_fast: the block factorises; no hand-off rows) with s[24:25], s[26:27], s[28:29] =",
           "// the caller's x, y, z rows; min |d_k| of the factorisation -> LDS word %d;" % asmqp.FAC_MIN,
           "// s31 / s34 / s35 = rho, 1/rho, rho_eq (float bits): rho of a row is selected from its streamed 1/rho",
           "#define BQP_%s_ASM%s(voff, ldsaddr, lane4, ws, sblk, stride, iters, alpha, oma, sigma, rinveq, xi, yi, zi, fast, rho0, rinv0, rhoeq) asm volatile( \\" % (name.upper(), "_LOOSE" if loose else "")]
    for t_ in ins:
        out.append('  "%s\\n" \\' % asmqp.fmt(t_))
    out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{v4}"(lane4), "{s[4:5]}"(ws), "{s[6:7]}"(sblk), "{s10}"(stride), '
               '"{s11}"(iters), "{s20}"(alpha), "{s21}"(oma), "{s22}"(sigma), "{s23}"(rinveq), "{s[24:25]}"(xi), "{s[26:27]}"(yi), '
               '"{s[28:29]}"(zi), "{s30}"(fast), "{s31}"(rho0), "{s34}"(rinv0), "{s35}"(rhoeq) \\')
    out.append("  : " + ", ".join(clob) + ")")
    return "\n".join(out) + "\n"


def res_macro(name, ins, group=False):
    from . import asmqp
    clob = ['"memory"', '"scc"', '"vcc"'] + _stamp_clobbers() + ['"v%d"' % i for i in [2, 3] + list(range(5, asmqp.V_END))] + \
           ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in [asmqp.S_SP, asmqp.S_SP + 1] + list(range(42, 54)) + [56, 57, 58, 59]]
    out = ["// Residuals, strict termination test and solution stores after the loop (asmqp.res_program): %d instructions." % len(ins),
           "// inputs: v0 = 4*robot, v1 = lane LDS address, v4 = 4*lane, s[6:7] = the wave's stream block, s10 = 4*B,",
           "// s[24:25] .. s[36:37] = x, y, z, sol_x, sol_y, status, info rows, s38 / s39 = eps_abs / eps_rel (float bits), s40 = max_iter,",
           "// s[54:55] = Eprev rows (E of this solve is stored there)",
           "#define BQP_%s_RES_ASM(voff, ldsaddr, lane4, sblk, stride, xo, yo, zo, sxo, syo, sto, ino, epsa, epsr, maxit, epo) asm volatile( \\" % name.upper()]
    if group:
        out = ["// The residual block SHARED by the workgroup's wavefronts (asmqp.res_group_program): each a quarter of the rows (A x, the row",
               "// norms) and a quarter of the columns (A' y, P x, q) -- every accumulation in the one-wavefront block's order --, the partial",
               "// norms folded by wavefront 0 between two barriers; s%d = the wavefront's index. %d instructions." % (asmqp.S_XWAVE, len(ins)),
               "#define BQP_%s_RES4_ASM(voff, ldsaddr, lane4, sblk, stride, xo, yo, zo, sxo, syo, sto, ino, epsa, epsr, maxit, epo, wave) asm volatile( \\" % name.upper()]
    for t_ in ins:
        out.append('  "%s\\n" \\' % asmqp.fmt(t_))
    out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{v4}"(lane4), "{s[6:7]}"(sblk), "{s10}"(stride), "{s[24:25]}"(xo), '
               '"{s[26:27]}"(yo), "{s[28:29]}"(zo), "{s[30:31]}"(sxo), "{s[32:33]}"(syo), "{s[34:35]}"(sto), "{s[36:37]}"(ino), '
               '"{s38}"(epsa), "{s39}"(epsr), "{s40}"(maxit), "{s[54:55]}"(epo)%s \\' % (', "{s%d}"(wave)' % asmqp.S_XWAVE if group else ""))
    out.append("  : " + ", ".join(clob) + ")")
    return "\n".join(out) + "\n"


def ruiz_macro(name, ins, rp, rs=False, group=False):
    from . import asmqp
    if group:
        clob = ['"memory"', '"scc"', '"vcc"'] + _stamp_clobbers() + ['"v%d"' % i for i in range(2, asmqp.V_END) if i != asmqp.V_RLANE] + \
               ['"a%d"' % i for i in range(256)] + \
               ['"s%d"' % i for i in (asmqp.S_P, asmqp.S_P + 1, asmqp.S_CNT, asmqp.S_RMIN, asmqp.S_RMAX)]
        nbar = sum(t_[0] == "s_barrier" for t_ in ins)
        out = ["// The passes and the residual stream as above, SHARED by the %d wavefronts of a workgroup that own the same 64 robots"
               % ASM_GROUP_WAVES,
               "// (asmqp.ruiz_group_program / RuizSplit: each wavefront takes a stretch of the columns; two s_barrier per pass; s26 = the",
               "// wavefront's index). %d instructions in %d sections; every word left in LDS and in the stream is bit-identical to the"
               % (len(ins), ASM_GROUP_WAVES),
               "// one-wavefront block's. Ends behind a barrier: whichever wavefront continues sees all of it.",
               "#define BQP_%s_RUIZ_RS4_ASM(voff, ldsaddr, lane4, av, pv, qv, sblk, stride, passes, wave) asm volatile( \\" % name.upper()]
        assert nbar % ASM_GROUP_WAVES == 0
        for t_ in ins:
            out.append('  "%s\\n" \\' % asmqp.fmt(t_))
        out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{v%d}"(lane4), "{s[4:5]}"(av), "{s[6:7]}"(pv), "{s[8:9]}"(qv), '
                   '"{s[24:25]}"(sblk), "{s10}"(stride), "{s11}"(passes), "{s%d}"(wave) \\' % (asmqp.V_RLANE, asmqp.S_RWAVE))
        out.append("  : " + ", ".join(clob) + ")")
        return "\n".join(out) + "\n"
    clob = ['"memory"', '"scc"', '"vcc"'] + _stamp_clobbers() + ['"v%d"' % i for i in range(2, asmqp.V_END) if not (rs and i == asmqp.V_RLANE)] + \
           ['"a%d"' % i for i in range(256)] + \
           ['"s%d"' % i for i in (asmqp.S_P, asmqp.S_P + 1, asmqp.S_CNT, asmqp.S_RMIN, asmqp.S_RMAX)]
    if rs:
        lab7 = [k for k, t_ in enumerate(ins) if t_ == ("label", "7")][0]
        out = ["// The same passes, and the equilibrated A, E, D, q, P, c written to the wave's residual stream (s[24:25], v%d = 4*lane)." % asmqp.V_RLANE,
               "#define BQP_%s_RUIZ_RS_ASM(voff, ldsaddr, lane4, av, pv, qv, sblk, stride, passes) asm volatile( \\" % name.upper()]
        for t_ in ins:
            out.append('  "%s\\n" \\' % asmqp.fmt(t_))
        out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{v%d}"(lane4), "{s[4:5]}"(av), "{s[6:7]}"(pv), "{s[8:9]}"(qv), '
                   '"{s[24:25]}"(sblk), "{s10}"(stride), "{s11}"(passes) \\' % asmqp.V_RLANE)
        out.append("  : " + ", ".join(clob) + ")")
        return "\n".join(out) + "\n"
    lab7 = [k for k, t_ in enumerate(ins) if t_ == ("label", "7")][0]
    br = [k for k, t_ in enumerate(ins) if t_[0] == "s_cbranch_scc1"][0]
    out = ["// The Ruiz passes of the %s structure (asmqp.ruiz_program), fp32: %d instructions, %d per pass." % (name, len(ins), br - lab7),
           "// inputs: v0 = 4*robot, v1 = lane LDS address, s[4:5] = Av rows, s[6:7] = Pv rows, s[8:9] = q rows, s10 = 4*B, s11 = passes >= 1",
           "#define BQP_%s_RUIZ_ASM(voff, ldsaddr, av, pv, qv, stride, passes) asm volatile( \\" % name.upper()]
    for t_ in ins:
        out.append('  "%s\\n" \\' % asmqp.fmt(t_))
    out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{s[4:5]}"(av), "{s[6:7]}"(pv), "{s[8:9]}"(qv), "{s10}"(stride), "{s11}"(passes) \\')
    out.append("  : " + ", ".join(clob) + ")")
    return "\n".join(out) + "\n"


def loader_macro(name, tag, groups):
    from . import asmqp
    ins = asmqp.loader_program(groups)
    clob = ['"memory"', '"scc"', '"vcc"'] + ['"v%d"' % i for i in range(2, asmqp.V_END)] + \
           ['"s%d"' % i for i in (asmqp.S_P, asmqp.S_P + 1)]
    out = ["// Batched loader (asmqp.loader_program): rows %s of the arrays at s[4:5], s[6:7], s[8:9] -> LDS words, one round trip"
           % ", ".join("%d -> %d.." % g for g in groups),
           "#define BQP_%s_LOAD_%s(voff, ldsaddr, p0, p1, p2, stride) asm volatile( \\" % (name.upper(), tag)]
    for t_ in ins:
        out.append('  "%s\\n" \\' % asmqp.fmt(t_))
    out.append('  : : "{v0}"(voff), "{v1}"(ldsaddr), "{s[4:5]}"(p0), "{s[6:7]}"(p1), "{s[8:9]}"(p2), "{s10}"(stride) \\')
    out.append("  : " + ", ".join(clob) + ")")
    return "\n".join(out) + "\n"


def generate():
    """Returns {relative path under csrc/: source}: one translation unit per (structure, dtype) so that the build can
    compile them in parallel, plus the registry header umpc_bqp.hip includes."""
    files = {}
    reg, decl = [], []
    for name, s in builtin_structures():
        body = emit_structure(name, s)
        asm_body, asm_hdr = None, None
        if name in ASM_STRUCTURES:
            from . import asmqp
            res = asmqp.ResPlan(s, ASM_STRUCTURES[name], ASM_RES_ITEM0)
            ins, plan = asmqp.program(s, ASM_STRUCTURES[name], res)
            ins_loose, _ = asmqp.program(s, ASM_STRUCTURES[name], res, loose=True)
            ins_loose4, _, lsplit = asmqp.loop_group_program(s, ASM_STRUCTURES[name], res, ASM_GROUP_WAVES)
            ins_gen4, _, _ = asmqp.loop_group_program(s, ASM_STRUCTURES[name], res, ASM_GROUP_WAVES, loose=False)
            assert plan.n_stream + len(plan.extra) <= ASM_RES_ITEM0
            plan.res = res
            assert plan.res.end <= ASM_STREAM_ITEMS
            rins, plan.ruiz = asmqp.ruiz_program(s)
            rsins, _ = asmqp.ruiz_program(s, plan.res)
            rs4ins, _, _ = asmqp.ruiz_group_program(s, plan.res, ASM_GROUP_WAVES)
            resins, _ = asmqp.res_program(s, ASM_STRUCTURES[name], plan, plan.res)
            res4ins, _ = asmqp.res_group_program(s, ASM_STRUCTURES[name], plan, plan.res, ASM_GROUP_WAVES)
            glins = asmqp.glue_program(s, ASM_STRUCTURES[name], plan, plan.res, plan.ruiz)
            gl4ins = asmqp.glue_group_program(s, ASM_STRUCTURES[name], plan, plan.res, plan.ruiz, ASM_GROUP_WAVES)
            asm_body = emit_structure(name, s, asm=plan)
            asm_hdr = "bqp_%s_asm.h" % name
            assert plan.ruiz.LW_END <= asmqp.LW_FLAGS
            files["gen/" + asm_hdr] = asm_macro(name, ins, plan) + asm_macro(name, ins_loose, plan, loose=True) + \
                asm_macro(name, ins_loose4, plan, loose=True, group=lsplit) + asm_macro(name, ins_gen4, plan, loose=False, group=lsplit) + \
                ruiz_macro(name, rins, plan.ruiz) + \
                ruiz_macro(name, rsins, plan.ruiz, rs=True) + ruiz_macro(name, rs4ins, plan.ruiz, rs=True, group=True) + \
                res_macro(name, resins) + res_macro(name, res4ins, group=True) + glue_macro(name, glins) + glue_macro(name, gl4ins, group=True) + \
                loader_macro(name, "XYZ", [(s.n, 0), (s.m, s.n), (s.m, s.n + s.m)]) + \
                loader_macro(name, "LUE", [(s.m, 0), (s.m, s.m), (s.m, 2 * s.m)])
        for tag, ctype in DTYPES:
            with_asm = asm_body is not None and tag == "f32"
            src = ["// GENERATED by robobee3d_amd/codegen_qp.py -- do not edit.",
                   "// Straight-line specialisation of bqp_solve_kernel: %s, n = %d, m = %d, nnz(A) = %d, nnz(L) = %d, %s."
                   % (name, s.n, s.m, s.nnzA, s.nnzL, ctype),
                   '#include "../umpc_bqp_common.h"'] + (['#include "%s"' % asm_hdr] if with_asm else []) + [
                   "", "namespace {", "using namespace umpcqp;", "", body] + ([asm_body] if with_asm else []) + [
                   "}  // namespace", "",
                   '__attribute__((visibility("hidden"))) void bqp_launch_%s_%s(const umpcqp::QPArgs<%s> &a, hipStream_t s) {'
                   % (name, tag, ctype)] + ([
                   "  // middle iterations as generated assembly (asmqp.py) when the host found room for the stream buffer",
                   "  if (a.asm_ok && a.max_iter >= 3) {",
                   "    hipLaunchKernelGGL(bqp_fixed_%s_asm_kernel, dim3((a.B + 63) / 64), dim3(%d), 0, s, a);" % (name, 64 * ASM_GROUP_WAVES),
                   "    return;",
                   "  }"] if with_asm else []) + [
                   "  hipLaunchKernelGGL(bqp_fixed_%s_kernel<%s>, dim3((a.B + 63) / 64), dim3(64), 0, s, a);" % (name, ctype),
                   "}", ""]
            files["gen/bqp_%s_%s.hip" % (name, tag)] = "\n".join(src)
            decl.append('__attribute__((visibility("hidden"))) void bqp_launch_%s_%s(const umpcqp::QPArgs<%s> &, hipStream_t);'
                        % (name, tag, ctype))
        reg.append('  {0x%016xull, "%s", bqp_launch_%s_f32, bqp_launch_%s_f64, %d},'
                   % (fnv1a(s.blob), name, name, name, 1 if name in ASM_STRUCTURES else 0))
    hdr = ["// GENERATED by robobee3d_amd/codegen_qp.py -- do not edit.",
           "// Registry of the build-time specialisations (csrc/gen/bqp_*.hip), keyed by the FNV-1a hash of the table blob.",
           "#pragma once", '#include "umpc_bqp_common.h"', "",
           "// items of one wave's stream block ([item][lane]; loop stream, then the residual stream): the allocation in",
           "// umpcQPCreate and the sblk indexing of the generated kernels both use THIS constant",
           "constexpr int BQP_ASM_STREAM_ITEMS_PER_WAVE = %d;" % ASM_STREAM_ITEMS, ""] + decl + [
           "struct FixedKernel { uint64_t hash; const char *name; void (*f32)(const umpcqp::QPArgs<float> &, hipStream_t); "
           "void (*f64)(const umpcqp::QPArgs<double> &, hipStream_t); int asm_f32; };",
           "static const FixedKernel kFixedKernels[] = {"] + reg + ["};",
           "constexpr int kNumFixedKernels = %d;" % len(reg), ""]
    files["umpc_bqp_registry.h"] = "\n".join(hdr)
    return files


def write():
    """Writes the generated files under csrc/ (only when their content changed); returns (registry path, [.hip paths])."""
    csrc = os.path.join(HERE, "csrc")
    os.makedirs(os.path.join(csrc, "gen"), exist_ok=True)
    files = generate()
    for rel, src in files.items():
        path = os.path.join(csrc, rel)
        old = open(path).read() if os.path.exists(path) else None
        if old != src:
            with open(path, "w") as f:
                f.write(src)
    keep = {os.path.basename(r) for r in files if r.startswith("gen/")}
    for fn in os.listdir(os.path.join(csrc, "gen")):
        if fn not in keep:
            os.remove(os.path.join(csrc, "gen", fn))
    return os.path.join(csrc, "umpc_bqp_registry.h"), sorted(os.path.join(csrc, r) for r in files if r.endswith(".hip"))


if __name__ == "__main__":
    reg, hips = write()
    print(reg, *hips, sep="\n")

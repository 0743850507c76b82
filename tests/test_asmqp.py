"""CPU checks of the generated assembly for the middle ADMM iterations of the planar p5f structure
(robobee3d_amd/asmqp.py): the instruction list is interpreted on numpy float32 -- with a completion model that rejects a
register used before the s_waitcnt covering its load -- and compared with a float64 numpy statement of the OSQP
iteration (osqp 0.6.0 auxil.c:164-228, qdldl.c:250-293) on random data of the structure."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def prog():
    from robobee3d_amd import asmqp, batchqp, codegen_qp, qpstruct
    st, s = batchqp.p5f_analysis(10)
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    # (the block codegen_qp emits: both starts -- hand-off rows, s30 == 0, and the in-block factorisation, s30 != 0)
    ins, p = asmqp.program(s, eq, asmqp.ResPlan(s, eq, codegen_qp.ASM_RES_ITEM0))
    return asmqp, ins, p


def _data(p, seed, eq):
    rng = np.random.default_rng(seed)
    n, m, nk = p.n, p.m, p.nk
    f = lambda a: a.astype(np.float32).astype(np.float64)
    d = dict(x=f(rng.normal(size=n)), y=f(rng.normal(size=m)), z=f(rng.normal(size=m)), q=f(rng.normal(size=n)),
             L=f(rng.normal(size=len(p.L_i)) * 0.3), DI=f(rng.normal(size=nk) * 0.5))
    l = f(rng.normal(size=m) - 1.0)
    u = l + f(np.abs(rng.normal(size=m)) + 0.1)
    rho = np.full(m, 0.1)
    rho[rng.random(m) < 0.2] = 1e-6          # "loose" rows
    rho[eq] = 100.0
    rho = f(rho)
    u[eq] = l[eq]
    d["z"][eq] = l[eq]                        # what the first (C++) iteration leaves
    d.update(l=l, u=u, rho=rho, rinv=f(1.0 / rho))
    return d


def _run(asmqp, ins, p, d, iters, eq, sigma=1e-6):
    n, m, nk = p.n, p.m, p.nk
    gen = sorted(p.zpos, key=lambda i: p.zpos[i])          # the inequality rows in the order of their z words
    W = np.zeros(p.R_END, np.float32)
    for j, pos in p.lpos.items():
        W[p.R_L + pos] = -d["L"][j]
    W[p.R_DI:p.R_DI + nk] = d["DI"]
    W[p.R_X:p.R_X + n] = d["x"]
    W[p.R_Y:p.R_Y + m] = d["y"]
    W[p.R_Z:p.R_Z + len(gen)] = d["z"][gen]
    S = np.zeros(p.n_stream + len(p.extra), np.float32)
    for q, (what, i) in enumerate(p.stream + p.extra):
        S[q] = {"rinv": d["rinv"], "l": d["l"], "u": d["u"], "rho": d["rho"], "q": d["q"]}[what][i]
    lds = asmqp.simulate(ins, W, S, iters, (1.6, sigma, float(np.float32(1.0 / 100.0))))
    return (lds[p.LW_X:p.LW_X + n], lds[p.LW_Y:p.LW_Y + m], lds[p.LW_Z:p.LW_Z + len(gen)], gen,
            lds[p.LW_XP:p.LW_XP + n], lds[p.LW_DY:p.LW_DY + m])


@pytest.mark.parametrize("iters", [0, 1, 2, 3])   # (the random system is not a contraction: more iterations overflow fp32)
def test_generated_p5f_iterations_match_numpy(prog, iters):
    """`iters` plain iterations, then the capturing one (x_prev, delta_y of the last iteration to their rows)"""
    from robobee3d_amd import codegen_qp
    asmqp, ins, p = prog
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    for seed in (0, 1):
        d = _data(p, seed, eq)
        gx, gy, gz, gen, gxp, gdy = _run(asmqp, ins, p, d, iters, eq)
        xr, yr, zr = asmqp.reference_iterations(p, d, iters + 1, 1.6, 1e-6)
        xq, yq, _ = asmqp.reference_iterations(p, d, iters, 1.6, 1e-6)
        for got, ref in ((gx, xr), (gy, yr), (gz, zr[gen]), (gxp, xq)):
            assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), (iters, seed)
        assert np.abs(gdy - (yr - yq)).max() <= 2e-5 * np.abs(yr).max(), (iters, seed)


def _ldl_numpy(s, A, Pv, rinv, sigma):
    """qdldl.c:86-247 on the KKT matrix of kkt.c:184-222, in the elimination order the generator recorded; float64"""
    n, nk = s.n, s.nk
    pidx = list(s.tables["pidx"])
    Lx, DI, yv = np.zeros(s.nnzL), np.zeros(nk), np.zeros(nk)
    for op in s.factor_ops:
        k = op["k"]
        for (bb, pk) in op["init"]:
            yv[bb] = A[s.K_src[pk][1]]
        orig = s.perm[k]
        dk = ((Pv[pidx[orig]] if pidx[orig] >= 0 else 0.0) + sigma) if orig < n else -rinv[orig - n]
        for (cidx, upd, new) in op["elim"]:
            yc = yv[cidx]
            for (j, row) in upd:
                yv[row] -= Lx[j] * yc
            lv = yc * DI[cidx]
            Lx[new] = lv
            dk -= yc * lv
            yv[cidx] = 0.0
        DI[k] = 1.0 / dk
    return Lx, DI


@pytest.mark.parametrize("iters", [0, 2])
def test_fast_start_factorises_in_the_block(prog, iters):
    """s30 != 0: A, P from the residual stream, 1/rho from the loop's stream, x, y, z from the caller's rows; the factor the
    block computes must drive the same iterations as a float64 LDL' handed over through the rows"""
    from robobee3d_amd import codegen_qp
    asmqp, ins, p = prog
    s = p.s
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    res = asmqp.ResPlan(s, eq, codegen_qp.ASM_RES_ITEM0)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    for seed in (0, 1):
        rng = np.random.default_rng(10 + seed)
        d = _data(p, seed, eq)
        A = f(rng.normal(size=s.nnzA))
        Pv = f(np.abs(rng.normal(size=s.nnzP)) + 0.5)
        # (a well-conditioned KKT matrix, so that fp32 against float64 factors is a sharp comparison: no loose rows, large sigma)
        sigma = float(np.float32(0.5))
        d["rho"][[i for i in range(p.m) if i not in set(eq)]] = f(np.array([0.1]))[0]
        d["rinv"] = f(1.0 / d["rho"])
        d["L"], d["DI"] = _ldl_numpy(s, A, Pv, d["rinv"], sigma)
        ref = _run(asmqp, ins, p, d, iters, eq, sigma)
        gen = ref[3]
        S = np.zeros(res.end, np.float32)
        for q, (what, i) in enumerate(p.stream + p.extra):
            S[q] = {"rinv": d["rinv"], "l": d["l"], "u": d["u"], "rho": d["rho"], "q": d["q"]}[what][i]
        S[res.it_A:res.it_A + s.nnzA] = A
        for j, it in res.it_p.items():
            S[it] = Pv[res.pidx[j]]
        arrs = [d[k].astype(np.float32) for k in ("x", "y", "z")]
        lds = asmqp.simulate(ins, np.full(p.R_END, np.nan, np.float32), S, iters, (1.6, sigma, float(np.float32(0.01))),
                             regions=[(asmqp.S_XI, arrs[0]), (asmqp.S_YI, arrs[1]), (asmqp.S_ZI, arrs[2])],
                             sgpr={asmqp.S_FAST: 1})
        got = (lds[p.LW_X:p.LW_X + p.n], lds[p.LW_Y:p.LW_Y + p.m], lds[p.LW_Z:p.LW_Z + len(gen)], gen,
               lds[p.LW_XP:p.LW_XP + p.n], lds[p.LW_DY:p.LW_DY + p.m])
        for k in (0, 1, 2, 4, 5):
            assert np.isfinite(ref[k]).all()
            assert np.abs(got[k] - ref[k]).max() <= 2e-4 * max(1.0, np.abs(ref[k]).max()), (seed, k)
        dmin = np.abs(1.0 / d["DI"]).min()
        assert abs(lds[asmqp.FAC_MIN] - min(1.0, dmin)) <= 1e-5 * min(1.0, dmin)


def test_plan_fits_the_lane(prog):
    from robobee3d_amd import codegen_qp
    asmqp, ins, p = prog
    assert len(p.nonleaf) == 162 and sum(r["leaf"] for r in p.rows) == 89
    assert p.V_TT + asmqp.N_TT <= asmqp.V_END <= 246 and p.nk <= 256 and p.LW_END <= 640
    assert p.R_END <= codegen_qp.ASM_STREAM_ROW and p.n_stream + len(p.extra) <= 1024
    import re
    for t in ins:
        for x in t[1:]:
            if isinstance(x, str):
                for a in re.findall(r"\bv(\d+)\b", x):
                    assert int(a) < asmqp.V_END
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", x):
                    assert int(b) < asmqp.V_END
        if t[0].startswith("ds_"):
            assert 0 <= t[3] < 65536
        if t[0].startswith("global_"):
            assert 0 <= t[4] < 4096


def test_completion_model_catches_a_missing_wait(prog):
    """the interpreter's in-order completion model is what guards the s_waitcnt placement: without the waits it must object"""
    from robobee3d_amd import codegen_qp
    asmqp, ins, p = prog
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    d = _data(p, 3, eq)
    stripped = [t for t in ins if t[0] != "s_waitcnt"]
    with pytest.raises(AssertionError):
        _run(asmqp, stripped, p, d, 1, eq)


def test_p5f_stream_assembles(prog):
    import os, subprocess, tempfile
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    from robobee3d_amd import codegen_qp
    asmqp, ins, p = prog
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    ins = ins + asmqp.glue_program(p.s, eq, p, asmqp.ResPlan(p.s, eq, codegen_qp.ASM_RES_ITEM0), asmqp.RuizPlan(p.s))
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write("\n".join(asmqp.fmt(t) for t in ins) + "\n")
    try:
        r = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, f.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[:3000]
    finally:
        os.unlink(f.name)


def _ruiz_numpy(p, P, A, q, passes):
    """the pass of codegen_qp.emit_structure (scaling.c:44-156), float64"""
    n, m = p.n, p.m
    P, A, q = P.copy(), A.copy(), q.copy()
    D, E, c = np.ones(n), np.ones(m), 1.0
    lim = lambda v: min(1.0 if v < 1e-4 else v, 1e4)
    for _ in range(passes):
        Dt, Et = np.zeros(n), np.zeros(m)
        for j in range(n):
            Dt[j] = abs(P[p.pidx[j]]) if p.pidx[j] >= 0 else 0.0
            for k in range(p.A_p[j], p.A_p[j + 1]):
                Dt[j] = max(Dt[j], abs(A[k]))
                Et[p.A_i[k]] = max(Et[p.A_i[k]], abs(A[k]))
        Dt = np.array([1.0 / np.sqrt(lim(x)) for x in Dt])
        Et = np.array([1.0 / np.sqrt(lim(x)) for x in Et])
        csum, qn = 0.0, 0.0
        for j in range(n):
            if p.pidx[j] >= 0:
                P[p.pidx[j]] *= Dt[j] * Dt[j]
                csum += abs(P[p.pidx[j]])
            for k in range(p.A_p[j], p.A_p[j + 1]):
                A[k] *= Et[p.A_i[k]] * Dt[j]
            q[j] *= Dt[j]
            qn = max(qn, abs(q[j]))
        D, E = D * Dt, E * Et
        ct = 1.0 / lim(max(csum / n, lim(qn)))
        P, q, c = P * ct, q * ct, c * ct
    return P, A, q, D, E, c


@pytest.mark.parametrize("passes", [1, 10])
def test_generated_p5f_ruiz_block_matches_numpy(passes):
    from robobee3d_amd import asmqp, batchqp, qpstruct
    st, s = batchqp.p5f_analysis(10)
    ins, p = asmqp.ruiz_program(s)
    rng = np.random.default_rng(passes)
    f = lambda a: a.astype(np.float32)
    for scale in (1.0, 1e-6, 3e5):           # the last two drive limit_scaling's branches
        P = f(np.abs(rng.normal(size=p.nnzP)) * 10 * scale + 1e-3 * scale)
        A = f(rng.normal(size=p.nnzA) * scale)
        A[rng.random(p.nnzA) < 0.3] = 1.0
        q = f(rng.normal(size=p.n) * scale)
        lds = asmqp.simulate(ins, np.zeros(1, np.float32), np.zeros(1, np.float32), passes, (1.6, 1e-6, 0.01),
                             regions=[(asmqp.S_AV, A), (asmqp.S_PV, P), (asmqp.S_QV, q)])
        Pr, Ar, qr, Dr, Er, cr = _ruiz_numpy(p, P.astype(np.float64), A.astype(np.float64), q.astype(np.float64), passes)
        for name, got, ref in (("A", lds[p.LW_A:p.LW_A + p.nnzA], Ar), ("P", lds[p.LW_P:p.LW_P + p.nnzP], Pr),
                               ("q", lds[p.LW_Q:p.LW_Q + p.n], qr), ("D", lds[p.LW_D:p.LW_D + p.n], Dr),
                               ("E", lds[p.LW_EV:p.LW_EV + p.m], Er)):
            assert np.abs(got - ref).max() <= 5e-6 * np.abs(ref).max(), (name, passes, scale)
        assert abs(lds[p.LW_C] - cr) <= 5e-6 * cr


def _p5f():
    from robobee3d_amd import asmqp, batchqp, codegen_qp, qpstruct
    st, s = batchqp.p5f_analysis(10)
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    return asmqp, s, eq, asmqp.Plan(s, eq), asmqp.ResPlan(s, eq, codegen_qp.ASM_RES_ITEM0)


def test_ruiz_block_writes_the_residual_stream():
    """with a ResPlan the Ruiz block's epilogue also leaves the equilibrated A, E, D, q, P, c in the wave's residual stream"""
    asmqp, s, eq, ap, res = _p5f()
    ins, p = asmqp.ruiz_program(s, res)
    rng = np.random.default_rng(5)
    P = np.abs(rng.normal(size=p.nnzP)).astype(np.float32) + 0.1
    A = rng.normal(size=p.nnzA).astype(np.float32)
    q = rng.normal(size=p.n).astype(np.float32)
    S = np.full(res.end, np.nan, np.float32)
    lds = asmqp.simulate(ins, np.zeros(1, np.float32), S, 3, (1.6, 1e-6, 0.01),
                         regions=[(asmqp.S_AV, A), (asmqp.S_PV, P), (asmqp.S_QV, q)], sgpr={asmqp.S_RSB: "S"})
    assert np.array_equal(S[res.it_A:res.it_A + p.nnzA], lds[p.LW_A:p.LW_A + p.nnzA])
    for i in range(p.m):
        assert S[res.it_ev[i]] == lds[p.LW_EV + i]
    for j in range(p.n):
        assert S[res.it_d[j]] == lds[p.LW_D + j] and S[res.it_q[j]] == lds[p.LW_Q + j]
        if j in res.it_p:
            assert S[res.it_p[j]] == lds[p.LW_P + p.pidx[j]]
    assert S[res.it_c] == lds[p.LW_C]
    assert all(np.isnan(S[res.it_ls[i]]) for i in res.it_ls)       # (the C++ side writes these)


@pytest.mark.parametrize("eps, expect", [(1e-3, 0.0), (1e3, 1.0)])
def test_residual_block_matches_numpy(eps, expect):
    """pri_res / dua_res, the strict termination flag, the info rows and the solution stores of asmqp.res_program against a
    float64 statement of auxil.c:243-307 + osqp.c:524-573 on random data (the flag must follow the test, both ways)"""
    asmqp, s, eq, ap, res = _p5f()
    ins, R = asmqp.res_program(s, eq, ap, res)
    rng = np.random.default_rng(0)
    n, m = s.n, s.m
    f32 = lambda a: a.astype(np.float32)
    A, P, q = f32(rng.normal(size=s.nnzA)), f32(np.abs(rng.normal(size=s.nnzP))), f32(rng.normal(size=n))
    D, Ev, c = f32(np.abs(rng.normal(size=n)) + 0.5), f32(np.abs(rng.normal(size=m)) + 0.5), np.float32(0.7)
    x, y, z = f32(rng.normal(size=n)), f32(rng.normal(size=m)), f32(rng.normal(size=m))
    Ax, Aty = np.zeros(m), np.zeros(n)
    for j in range(n):
        for k in range(res.A_p[j], res.A_p[j + 1]):
            Ax[res.A_i[k]] += float(A[k]) * float(x[j])
            Aty[j] += float(A[k]) * float(y[res.A_i[k]])
    einv, dinv = 1 / Ev.astype(np.float64), 1 / D.astype(np.float64)
    px = np.array([float(P[res.pidx[j]]) * float(x[j]) if res.pidx[j] >= 0 else 0.0 for j in range(n)])
    pri, dua = np.abs(einv * (Ax - z)).max(), np.abs(dinv * (q + px + Aty)).max() / c
    prel = max(np.abs(einv * z).max(), np.abs(einv * Ax).max())
    drel = max(np.abs(dinv * q).max(), np.abs(dinv * Aty).max(), np.abs(dinv * px).max()) / c
    assert float(pri < eps + eps * prel and dua < eps + eps * drel) == expect
    S = np.zeros(res.end, np.float32)
    S[res.it_A:res.it_A + s.nnzA] = A
    for i in range(m):
        S[res.it_ev[i]] = Ev[i]
        if i in res.eq:
            S[res.it_ls[i]] = z[i]
    for j in range(n):
        S[res.it_d[j]], S[res.it_q[j]] = D[j], q[j]
        if j in res.it_p:
            S[res.it_p[j]] = P[res.pidx[j]]
    S[res.it_c] = c
    lds0 = np.zeros(640, np.float32)
    lds0[ap.LW_X:ap.LW_X + n], lds0[ap.LW_Y:ap.LW_Y + m] = x, y
    for i, qq in ap.zpos.items():
        lds0[ap.LW_Z + qq] = z[i]
    xo, yo, zo, sx, sy = [np.full(k, np.nan, np.float32) for k in (n, m, m, n, m)]
    stt, info, epo = np.zeros(1, np.float32), np.zeros(6, np.float32), np.full(m, np.nan, np.float32)
    sg = {asmqp.S_EPSA: asmqp.f32bits(eps), asmqp.S_EPSR: asmqp.f32bits(eps), asmqp.S_MAXIT: 50}
    lds = asmqp.simulate(ins, np.zeros(1, np.float32), S, 1, (1.6, 1e-6, 0.01),
                         regions=[(asmqp.S_XO, xo), (asmqp.S_YO, yo), (asmqp.S_ZO, zo), (asmqp.S_SX, sx), (asmqp.S_SY, sy),
                                  (asmqp.S_ST, stt), (asmqp.S_IN, info), (asmqp.S_EP, epo)], sgpr=sg, lds0=lds0)
    assert lds[asmqp.RES_FLAG] == expect and np.array_equal(epo, Ev)       # (E of this solve -> the caller's Eprev rows)
    assert abs(info[0] - pri) <= 2e-6 * pri and abs(info[1] - dua) <= 2e-6 * dua and info[2] == c and info[3] == 0 and info[4] == 50
    assert np.array_equal(xo, x) and np.array_equal(yo, y) and np.array_equal(zo, z)
    assert np.abs(sx - x * D).max() <= 1e-6 * np.abs(x * D).max() and np.abs(sy - y * Ev / c).max() <= 1e-6 * np.abs(y * Ev / c).max()
    # the block shared by the workgroup's four wavefronts (res_group_program: each on its components): every store, the info
    # rows, the status word and the flag bit for bit the one-wave block's; two barriers, no LDS race
    grp, _ = asmqp.res_group_program(s, eq, ap, res, 4)
    out4 = [np.full(k, np.nan, np.float32) for k in (n, m, m, n, m)]
    stt4, info4, epo4 = np.zeros(1, np.float32), np.zeros(6, np.float32), np.full(m, np.nan, np.float32)
    n1 = []
    asmqp.simulate(ins, np.zeros(1, np.float32), S, 1, (1.6, 1e-6, 0.01), count=n1,
                   regions=[(asmqp.S_XO, xo.copy()), (asmqp.S_YO, yo.copy()), (asmqp.S_ZO, zo.copy()), (asmqp.S_SX, sx.copy()),
                            (asmqp.S_SY, sy.copy()), (asmqp.S_ST, stt.copy()), (asmqp.S_IN, info.copy()), (asmqp.S_EP, epo.copy())],
                   sgpr=sg, lds0=lds0)
    lds4, counts, nbar = asmqp.simulate_group(grp, 4, np.zeros(1, np.float32), S, 1, (1.6, 1e-6, 0.01), asmqp.S_XWAVE,
                                              regions=[(asmqp.S_XO, out4[0]), (asmqp.S_YO, out4[1]), (asmqp.S_ZO, out4[2]),
                                                       (asmqp.S_SX, out4[3]), (asmqp.S_SY, out4[4]), (asmqp.S_ST, stt4),
                                                       (asmqp.S_IN, info4), (asmqp.S_EP, epo4)], sgpr=sg, lds0=lds0)
    assert nbar == 2 and lds4[asmqp.RES_FLAG] == lds[asmqp.RES_FLAG]
    for got, ref in zip(out4 + [stt4, info4, epo4], [xo, yo, zo, sx, sy, stt, info, epo]):
        assert np.array_equal(got, ref)
    assert max(counts) < 0.6 * n1[0], (counts, n1)


def _glue_expected(p, res, eq, l, u, ep, ev, q, rho0):
    """the C++ glue of codegen_qp.emit_structure in numpy float32 (update_rho_vec auxil.c:103-145 with the comparisons in
    double; l E, u E)"""
    f32 = np.float32
    rho_eq = f32(1e3 * float(rho0))
    le, ue = l * ep, u * ep
    rho = np.where((le.astype(np.float64) < -1e16) & (ue.astype(np.float64) > 1e16), f32(1e-6),
                   np.where((ue - le).astype(np.float64) < 1e-4, rho_eq, f32(rho0))).astype(f32)
    rinv = np.where(rho == f32(1e-6), f32(1.0 / 1e-6), np.where(rho == rho_eq, f32(1.0 / float(rho_eq)), f32(1.0 / float(rho0)))).astype(f32)
    ls, us = l * ev, u * ev
    S = np.zeros(res.end, f32)
    src = {"rinv": rinv, "l": ls, "u": us, "rho": rho, "q": q}
    for k, (what, i) in enumerate(p.stream + p.extra):
        S[k] = src[what][i]
    for i, it in res.it_ls.items():
        S[it] = ls[i]
    return S, rho, ls, us


def test_glue_block_classifies_and_writes_the_stream(prog):
    from robobee3d_amd import codegen_qp
    asmqp, _, p = prog
    s = p.s
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    res = asmqp.ResPlan(s, eq, codegen_qp.ASM_RES_ITEM0)
    rp = asmqp.RuizPlan(s)
    ins = asmqp.glue_program(s, eq, p, res, rp)
    f32 = np.float32
    rng = np.random.default_rng(5)
    rho0 = f32(0.1)
    consts = {asmqp.GV_RHO0: rho0, asmqp.GV_RINV0: f32(1.0 / float(rho0)), asmqp.GV_RHOEQ: f32(1e3 * float(rho0)),
              asmqp.GV_RINVEQ: f32(1.0 / float(f32(1e3 * float(rho0))))}
    pre = [("v_mov_b32", "v%d" % r, asmqp.f32bits(float(val))) for r, val in consts.items()]
    for case in ("ok", "cold", "noteq", "loose"):
        m, n = p.m, p.n
        l = rng.normal(size=m).astype(f32)
        u = (l + np.abs(rng.normal(size=m)).astype(f32) + f32(0.1)).astype(f32)
        loose = [i for i in range(m) if i not in set(eq)][::1 if case == "loose" else 7]
        l[loose], u[loose] = f32(-1e20), f32(1e20)
        half = [] if case == "loose" else [i for i in range(m) if i not in set(eq)][3::7]
        u[half] = f32(1e20)                              # one-sided rows are ordinary inequality rows
        u[eq] = l[eq]
        ep = (np.abs(rng.normal(size=m)) + 0.5).astype(f32)
        ev = (np.abs(rng.normal(size=m)) + 0.5).astype(f32)
        q = rng.normal(size=n).astype(f32)
        if case == "noteq":
            u[eq[5]] = l[eq[5]] + f32(1.0)
        z = rng.normal(size=m).astype(f32)
        z[eq] = (l * ev)[eq]
        if case == "cold":
            z[eq[11]] += f32(0.5)
        Sx, rho, ls, us = _glue_expected(p, res, eq, l, u, ep, ev, q, rho0)
        lds0 = np.zeros(640, f32)
        lds0[rp.LW_EV:rp.LW_EV + m] = ev
        lds0[rp.LW_Q:rp.LW_Q + n] = q
        S = np.full(res.end, np.nan, f32)
        lds = asmqp.simulate(pre + ins, np.zeros(1, f32), S, 0, (1.6, 1e-6, 0.01), lds0=lds0,
                             regions=[(asmqp.S_LR, l), (asmqp.S_UR, u), (asmqp.S_ER, ep), (asmqp.S_ZR, z)])
        nst = p.n_stream + len(p.extra)
        assert np.array_equal(S[:nst], Sx[:nst]), case
        its = sorted(res.it_ls.values())
        assert np.array_equal(S[its], Sx[its]), case
        assert lds[asmqp.GLUE_FLAG] == (1.0 if case in ("ok", "loose") else 0.0), case
        assert lds[asmqp.LOOSE_FLAG] == (1.0 if case == "loose" else 0.0), case
        # the block shared by four wavefronts (glue_group_program): the same stream, the same flags, no LDS race, three barriers.
        # It decides about the loose loop BEFORE it stores (round 5): when every inequality row of the robot is a loose row the
        # per-row items of those rows -- what only the general loop and the C++ routes read -- stay unwritten
        S4 = np.full(res.end, np.nan, f32)
        lds4, _, nbar = asmqp.simulate_group(pre + asmqp.glue_group_program(s, eq, p, res, rp, 4), 4, np.zeros(1, f32), S4, 0,
                                             (1.6, 1e-6, 0.01), asmqp.S_GWAVE, lds0=lds0,
                                             regions=[(asmqp.S_LR, l), (asmqp.S_UR, u), (asmqp.S_ER, ep), (asmqp.S_ZR, z)])
        lean = [q_ for q_, (what, i) in enumerate(p.stream + p.extra) if case == "loose" and i not in set(eq) and what in ("rinv", "l", "u", "rho")]
        assert (len(lean) > 4 * 87) == (case == "loose") and np.isnan(S4[lean]).all()
        S4[lean] = S[lean]
        assert nbar == 3 and np.array_equal(S, S4, equal_nan=True), case
        assert lds4[asmqp.GLUE_FLAG] == lds[asmqp.GLUE_FLAG] and lds4[asmqp.LOOSE_FLAG] == lds[asmqp.LOOSE_FLAG], case
        assert (rho == f32(1e-6)).sum() == len(loose) and (rho == f32(100.0)).sum() == len(eq) - (case == "noteq")


def test_interpreter_faults_on_a_sign_extended_base_pointer():
    """The GPU memory fault of round 2 (a uniform scalar operand widened as a SIGNED int on its way into a 64-bit base
    pointer, fixed in 37f012f) is now caught on the CPU: the interpreters form addresses as the ISA does -- SGPR pair +
    zero-extended VGPR offset + immediate, on simulated device addresses whose low word has bit 31 set, like real ones --
    and every access must land on an element of an array handed in. The pre-37f012f operand faults; today's passes."""
    asmqp, s, eq, ap, res = _p5f()
    ins, p = asmqp.ruiz_program(s, res)
    rng = np.random.default_rng(5)
    P = np.abs(rng.normal(size=p.nnzP)).astype(np.float32) + 0.1
    A = rng.normal(size=p.nnzA).astype(np.float32)
    q = rng.normal(size=p.n).astype(np.float32)

    def run(xform):
        S = np.full(res.end, np.nan, np.float32)
        return asmqp.simulate(ins, np.zeros(1, np.float32), S, 3, (1.6, 1e-6, 0.01), base_xform=xform,
                              regions=[(asmqp.S_AV, A), (asmqp.S_PV, P), (asmqp.S_QV, q)], sgpr={asmqp.S_RSB: "S"})
    run(asmqp.uni_scalar_operand)                                              # the shipped glue: fine
    assert asmqp.uni_scalar_operand(asmqp.SIM_BASE_S) == asmqp.SIM_BASE_S
    assert asmqp.uni_scalar_operand(asmqp.SIM_BASE_S, sign_extend_bug=True) != asmqp.SIM_BASE_S
    with pytest.raises(asmqp.AddressFault):
        run(lambda a: asmqp.uni_scalar_operand(a, sign_extend_bug=True))       # the pre-37f012f operand
    # an access one row past an array is a fault too (not a silent read of a neighbour)
    with pytest.raises(asmqp.AddressFault):
        asmqp.simulate(ins, np.zeros(1, np.float32), np.full(res.end, np.nan, np.float32), 3, (1.6, 1e-6, 0.01),
                       regions=[(asmqp.S_AV, A[:-1]), (asmqp.S_PV, P), (asmqp.S_QV, q)], sgpr={asmqp.S_RSB: "S"})


def test_step_interpreter_faults_on_a_bad_pointer():
    """asmstep.simulate applies the same address model to the all-assembly step kernel: a parameter-block pointer whose
    low half was sign-extended, or an array one row short, faults instead of reading something else."""
    from robobee3d_amd import asmgen, asmstep
    from robobee3d_amd.batch import hover_initial_conditions
    ins = asmstep.StepGen().program()
    st, ref = hover_initial_conditions(1, 1, np.float32)

    def arrays(nctrl=127):
        a = dict(state=st[:, 0].copy(), ctrl=np.zeros(nctrl, np.float32), ref=ref[:, 0].copy(),
                 ws=np.zeros(asmgen.WS_ROWS, np.float32), out=np.zeros(9, np.float32), stats=np.zeros(2, np.float32),
                 status=np.zeros(1, np.int32), info=np.zeros(2, np.float32))
        a["ctrl"][124:] = 1
        return a
    prm = dict(K=1, maxIter=2, nsub=1, plant=1)
    asmstep.simulate(ins, arrays(), prm, asmstep.host_floats())
    with pytest.raises(asmqp_fault()):
        asmstep.simulate(ins, arrays(), prm, asmstep.host_floats(), ptr_xform=lambda a: _sx(a))
    with pytest.raises(asmqp_fault()):
        asmstep.simulate(ins, arrays(nctrl=126), prm, asmstep.host_floats())


def asmqp_fault():
    from robobee3d_amd import asmqp
    return asmqp.AddressFault


def _sx(a):
    from robobee3d_amd import asmqp
    return asmqp.uni_scalar_operand(a, sign_extend_bug=True)


_Y0_COUNTS = {}


@pytest.mark.parametrize("zero_y", [False, True])
@pytest.mark.parametrize("iters", [0, 2])
def test_loose_loop_variant_is_bit_identical_to_the_general_loop(prog, iters, zero_y):
    """asmqp.program(..., loose=True): when every inequality row is a loose row (rho = RHO_MIN, bounds beyond +-1e26 -- the
    reference's planar p5f problem, planar/mpc_osqp_p5f.py:94-97) the variant that takes rho and 1 / rho from SGPRs, never
    clips and streams nothing per inequality row must leave EXACTLY the words the general loop leaves (fast start, the
    block's own factorisation included). And it really is shorter.
    zero_y: the warm start has y == 0 on every inequality row (a cold start, or an earlier result of the loop): the loose
    program then takes its y0 loop (no multiplier words for those rows, q and l of the packed equality rows in LDS instead of
    the per-iteration loads) -- still bit for bit the general loop's words, in fewer instructions again."""
    from robobee3d_amd import codegen_qp
    asmqp, ins, p = prog
    s = p.s
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    res = asmqp.ResPlan(s, eq, codegen_qp.ASM_RES_ITEM0)
    ins_l, pl = asmqp.program(s, eq, res, loose=True)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    gen = [i for i in range(p.m) if i not in set(eq)]
    rng = np.random.default_rng(3)
    d = _data(p, 4, eq)
    d["rho"][gen] = f(np.array([1e-6]))[0]
    d["rinv"] = f(np.float32(1.0) / d["rho"].astype(np.float32))
    d["l"][gen], d["u"][gen] = -1e26, 1e26
    A = f(rng.normal(size=s.nnzA))
    Pv = f(np.abs(rng.normal(size=s.nnzP)) + 0.5)
    S = np.zeros(res.end, np.float32)
    for q, (what, i) in enumerate(p.stream + p.extra):
        S[q] = {"rinv": d["rinv"], "l": d["l"], "u": d["u"], "rho": d["rho"], "q": d["q"]}[what][i]
    S[res.it_A:res.it_A + s.nnzA] = A
    for j, it in res.it_p.items():
        S[it] = Pv[res.pidx[j]]
    if zero_y:
        d["y"][gen] = 0.0
    arrs = [d[k].astype(np.float32) for k in ("x", "y", "z")]
    out, nexec = [], []
    for prog_ins in (ins, ins_l):
        lds = asmqp.simulate(prog_ins, np.full(p.R_END, np.nan, np.float32), S.copy(), iters, (1.6, 0.5, float(np.float32(0.01))),
                             regions=[(asmqp.S_XI, arrs[0].copy()), (asmqp.S_YI, arrs[1].copy()), (asmqp.S_ZI, arrs[2].copy())],
                             sgpr={asmqp.S_FAST: 1}, count=nexec)
        out.append(np.concatenate([lds[p.LW_X:p.LW_X + p.n], lds[p.LW_Y:p.LW_Y + p.m], lds[p.LW_Z:p.LW_Z + len(gen)],
                                   lds[p.LW_XP:p.LW_XP + p.n], lds[p.LW_DY:p.LW_DY + p.m], lds[asmqp.FAC_MIN:asmqp.FAC_MIN + 1]]))
    assert np.isfinite(out[0]).all() and np.array_equal(out[0], out[1])
    if zero_y:
        assert not out[1][p.n:p.n + p.m][gen].any()            # the multipliers of the loose rows: exactly zero, restored
    _Y0_COUNTS[(iters, zero_y)] = nexec[1]
    if (iters, True) in _Y0_COUNTS and (iters, False) in _Y0_COUNTS and iters:
        assert _Y0_COUNTS[(iters, True)] < 0.97 * _Y0_COUNTS[(iters, False)]     # the y0 loop ran, and is shorter
    loop = lambda L: [k for k, t in enumerate(L) if t == ("label", "8")][0] - [k for k, t in enumerate(L) if t == ("label", "7")][0]
    assert loop(ins_l) <= 0.83 * loop(ins)
    assert sum(1 for t in ins_l if t[0] == "global_load_dword") < 0.5 * sum(1 for t in ins if t[0] == "global_load_dword")


def test_lds_write_combiners():
    """asmqp.couples / asmqp.QuadWriter (round 3: an LDS instruction costs a lone wave the same whatever its width): which
    pair writes share a ds_write_b128, and which instructions a run of consecutive words is written back with."""
    from robobee3d_amd import asmqp
    # pairs at words 8, 10 | 12, 14 fill two float4; 18 is alone (16 is missing); 20, 22 couple; a repeated word never couples twice
    first, second = asmqp.couples([8, 10, 12, 14, 18, 20, 22])
    assert first == {8, 12, 20} and second == {10, 14, 22}
    assert asmqp.couples([2, 4, 6]) == ({4}, {6}) and asmqp.couples([]) == (set(), set())

    class Rec:
        def __init__(self):
            self.log = []

        def lds_write(self, w, r):
            self.log.append(("b32", w, r))

        def lds_write2(self, w, r):
            assert w % 2 == 0 and r % 2 == 0
            self.log.append(("b64", w, r))

        def lds_write4(self, w, r):
            assert w % 4 == 0 and r % 2 == 0
            self.log.append(("b128", w, r))
    sc = Rec()
    qw = asmqp.QuadWriter(sc, 100)
    words = list(range(341, 352))                      # a run that starts at position 1 of its float4 and ends at position 3
    for w in words:
        assert qw.reg(w) == 100 + w % 4
        qw.done(w, w == words[-1])
    assert sc.log == [("b32", 341, 101), ("b64", 342, 102), ("b128", 344, 100), ("b128", 348, 100)]
    sc.log.clear()
    for w in (0, 1, 2, 3, 4, 5):                       # ... and a run that ends in the middle of one
        qw.done(w, w == 5)
    assert sc.log == [("b128", 0, 100), ("b64", 4, 100)]


@pytest.mark.parametrize("passes", [1, 3, 10])
def test_ruiz_block_shared_by_four_waves_is_bit_identical_to_one_wave(passes):
    """asmqp.ruiz_group_program: the four wavefronts of a workgroup split the columns of the SAME 64 robots (RuizSplit) and meet
    at two barriers per pass. The interpreter runs the four programs barrier phase by barrier phase on one LDS image, rejects
    a word written by one wave and touched by another between two barriers, and a wave that misses a barrier; every LDS word
    the block leaves and every item of the residual stream must equal the one-wave block's BIT for bit."""
    asmqp, s, eq, ap, res = _p5f()
    one, p = asmqp.ruiz_program(s, res)
    grp, _, sp = asmqp.ruiz_group_program(s, res, 4)
    assert sorted(set(sp.colw.values())) == [0, 1, 2, 3] and 0 < len(sp.shared) <= 40
    rng = np.random.default_rng(40 + passes)
    f = lambda a: a.astype(np.float32)
    for scale in (1.0, 1e-6, 3e5):
        P = f(np.abs(rng.normal(size=p.nnzP)) * 10 * scale + 1e-3 * scale)
        A = f(rng.normal(size=p.nnzA) * scale)
        A[rng.random(p.nnzA) < 0.3] = 1.0
        q = f(rng.normal(size=p.n) * scale)
        S1, S4 = np.full(res.end, np.nan, np.float32), np.full(res.end, np.nan, np.float32)
        kw = dict(regions=[(asmqp.S_AV, A), (asmqp.S_PV, P), (asmqp.S_QV, q)], sgpr={asmqp.S_RSB: "S"})
        lds1 = asmqp.simulate(one, np.zeros(1, np.float32), S1, passes, (1.6, 1e-6, 0.01), **kw)
        lds4, counts, nbar = asmqp.simulate_group(grp, 4, np.zeros(1, np.float32), S4, passes, (1.6, 1e-6, 0.01), asmqp.S_RWAVE, **kw)
        assert nbar == 2 * passes + 2
        # E, c, P, q in LDS (what the glue block and the C++ side read there); A and D of a shared block live in registers across
        # the passes and leave through the residual stream only, which the factorisation and the residual block read
        keep = list(range(p.LW_EV, p.LW_END))
        assert np.array_equal(lds1[keep], lds4[keep]), scale
        assert np.array_equal(S1, S4, equal_nan=True), scale
    # the point of it: the longest of the four programs executes about a third of the one-wave block's instructions
    n1 = []
    asmqp.simulate(one, np.zeros(1, np.float32), S1, passes, (1.6, 1e-6, 0.01), count=n1, **kw)
    assert max(counts) < 0.42 * n1[0], (counts, n1)


@pytest.mark.parametrize("iters,zero_y", [(0, True), (1, True), (3, True), (3, False), (2, "mixed")])
def test_loose_loop_shared_by_the_workgroup_is_bit_identical_to_one_wave(prog, iters, zero_y):
    """asmqp.loop_group_program: the QP's connected components (two chains of the horizon and five small pieces for planar
    p5f) are independent QPs; the wavefronts of the workgroup run the loose loop block each on its own components, side by
    side on disjoint words of the same LDS layout, meeting after the factorisation, twice in the capturing iteration (its
    stores reuse the L words) and at the end. The interpreter runs the four wavefronts barrier phase by barrier phase with
    its LDS race check; x, y, z, x_prev, delta_y and the pivot flag must equal the one-wave loose program's bit for bit, and
    the longest wavefront must execute little more than half the one-wave block's instructions."""
    from robobee3d_amd import codegen_qp
    asmqp, _, p = prog
    s = p.s
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    res = asmqp.ResPlan(s, eq, codegen_qp.ASM_RES_ITEM0)
    one, _ = asmqp.program(s, eq, res, loose=True)
    grp, pg, sp = asmqp.loop_group_program(s, eq, res, 4)
    # KKT unknowns per wavefront: the two horizon chains (118 and 112 unknowns) are each cut in two halves under a two-vertex
    # separator (qpstruct.bisect_ordering / LoopSplit), the small pieces fill up the lighter wavefronts
    assert sp.active == 4 and sp.load == [64, 63, 62, 62] and sorted(len(c["T"]) for c in sp.cut.values()) == [1, 2]
    f = lambda a: a.astype(np.float32).astype(np.float64)
    gen = [i for i in range(p.m) if i not in set(eq)]
    rng = np.random.default_rng(3)
    d = _data(p, 4, eq)
    d["rho"][gen] = f(np.array([1e-6]))[0]
    d["rinv"] = f(np.float32(1.0) / d["rho"].astype(np.float32))
    d["l"][gen], d["u"][gen] = -1e26, 1e26
    A = f(rng.normal(size=s.nnzA))
    Pv = f(np.abs(rng.normal(size=s.nnzP)) + 0.5)
    S = np.zeros(res.end, np.float32)
    for q, (what, i) in enumerate(p.stream + p.extra):
        S[q] = {"rinv": d["rinv"], "l": d["l"], "u": d["u"], "rho": d["rho"], "q": d["q"]}[what][i]
    S[res.it_A:res.it_A + s.nnzA] = A
    for j, it in res.it_p.items():
        S[it] = Pv[res.pidx[j]]
    if zero_y:
        d["y"][gen] = 0.0
    if zero_y == "mixed":
        # multipliers of wavefront 1's loose rows only: wavefront 1 takes the general-loose bodies, the others their y0 bodies --
        # every path through the block meets the same four barriers
        for i in gen:
            if sp.roww[i] == 1:
                d["y"][i] = 0.25
    arrs = [d[k].astype(np.float32) for k in ("x", "y", "z")]
    consts = (1.6, 0.5, float(np.float32(0.01)))
    words = lambda lds: np.concatenate([lds[p.LW_X:p.LW_X + p.n], lds[p.LW_Y:p.LW_Y + p.m], lds[p.LW_Z:p.LW_Z + len(gen)],
                                        lds[p.LW_XP:p.LW_XP + p.n], lds[p.LW_DY:p.LW_DY + p.m], lds[asmqp.FAC_MIN:asmqp.FAC_MIN + 1]])
    n1 = []
    regs = lambda: [(asmqp.S_XI, arrs[0].copy()), (asmqp.S_YI, arrs[1].copy()), (asmqp.S_ZI, arrs[2].copy())]
    # ("mixed": the one-wave block decides for all rows at once -- with any multiplier set it runs the general-loose bodies, whose
    # words are the y0 bodies' bit for bit: test_loose_loop_variant_is_bit_identical_to_the_general_loop)
    lds1 = asmqp.simulate(one, np.full(p.R_END, np.nan, np.float32), S.copy(), iters, consts, regions=regs(),
                          sgpr={asmqp.S_FAST: 1}, count=n1)
    lds4, counts, nbar = asmqp.simulate_group(grp, 4, np.full(p.R_END, np.nan, np.float32), S.copy(), iters, consts, asmqp.S_LWAVE,
                                              regions=regs(), sgpr={asmqp.S_FAST: 1})
    # barriers: after the factorisation + the 1/D hand-over of the cut components, two per solve (iters + 1 bodies), two in the
    # capturing iteration, one at the end
    assert nbar == 5 + 2 * (iters + 1), nbar
    assert np.isfinite(words(lds1)).all() and np.array_equal(words(lds1), words(lds4))
    # (the halves of a cut component factorise their own subtrees too: every wavefront executes about 28 % of the one-wave block)
    assert max(counts) < 0.33 * n1[0], (counts, n1)


@pytest.mark.parametrize("iters", [0, 2])
def test_general_loop_shared_by_the_workgroup_is_bit_identical_to_one_wave(prog, iters):
    """loop_group_program(loose=False): the GENERAL variant of the loop block (finite bounds: l, u, 1/rho, rho of every inequality
    row streamed, clipping) split by the QP's components over the wavefronts like the loose one -- every wavefront lands its own
    items of the stream (a landing map in the scheduler). Bit-identical LDS words, no race, the same four barriers."""
    from robobee3d_amd import codegen_qp
    asmqp, one, p = prog
    s = p.s
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    res = asmqp.ResPlan(s, eq, codegen_qp.ASM_RES_ITEM0)
    grp, _, sp = asmqp.loop_group_program(s, eq, res, 4, loose=False)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(21)
    d = _data(p, 2, eq)
    gen = [i for i in range(p.m) if i not in set(eq)]
    A = f(rng.normal(size=s.nnzA))
    Pv = f(np.abs(rng.normal(size=s.nnzP)) + 0.5)
    S = np.zeros(res.end, np.float32)
    for q, (what, i) in enumerate(p.stream + p.extra):
        S[q] = {"rinv": d["rinv"], "l": d["l"], "u": d["u"], "rho": d["rho"], "q": d["q"]}[what][i]
    S[res.it_A:res.it_A + s.nnzA] = A
    for j, it in res.it_p.items():
        S[it] = Pv[res.pidx[j]]
    arrs = [d[k].astype(np.float32) for k in ("x", "y", "z")]
    consts = (1.6, 0.5, float(np.float32(0.01)))
    regs = lambda: [(asmqp.S_XI, arrs[0].copy()), (asmqp.S_YI, arrs[1].copy()), (asmqp.S_ZI, arrs[2].copy())]
    words = lambda lds: np.concatenate([lds[p.LW_X:p.LW_X + p.n], lds[p.LW_Y:p.LW_Y + p.m], lds[p.LW_Z:p.LW_Z + len(gen)],
                                        lds[p.LW_XP:p.LW_XP + p.n], lds[p.LW_DY:p.LW_DY + p.m], lds[asmqp.FAC_MIN:asmqp.FAC_MIN + 1]])
    n1 = []
    lds1 = asmqp.simulate(one, np.full(p.R_END, np.nan, np.float32), S.copy(), iters, consts, regions=regs(),
                          sgpr={asmqp.S_FAST: 1}, count=n1)
    lds4, counts, nbar = asmqp.simulate_group(grp, 4, np.full(p.R_END, np.nan, np.float32), S.copy(), iters, consts, asmqp.S_LWAVE,
                                              regions=regs(), sgpr={asmqp.S_FAST: 1})
    assert nbar == 5 + 2 * (iters + 1) and np.isfinite(words(lds1)).all() and np.array_equal(words(lds1), words(lds4))
    assert max(counts) < 0.45 * n1[0], (counts, n1)


def test_no_wide_store_data_hazard_in_any_p5f_block(prog):
    """A store of more than 64 bits (ds_write_b128) reads its data registers over several cycles: a VALU instruction that
    overwrites one of them within two wait states races with it on gfx940 and later, and nothing inserts the nops for inline
    assembly (asmgen.Emit does, at emission). Round 5 found one such pair in the cut loop block -- the interpreter cannot see
    it, the GPU can."""
    from robobee3d_amd import asmgen, codegen_qp
    asmqp, _, p = prog
    s = p.s
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    res = asmqp.ResPlan(s, eq, codegen_qp.ASM_RES_ITEM0)
    rp = asmqp.RuizPlan(s)
    blocks = [asmqp.loop_group_program(s, eq, res, 4)[0], asmqp.loop_group_program(s, eq, res, 4, loose=False)[0],
              asmqp.program(s, eq, res, loose=True)[0], asmqp.ruiz_group_program(s, res, 4)[0],
              asmqp.res_group_program(s, eq, p, res, 4)[0], asmqp.glue_group_program(s, eq, p, res, rp, 4)]
    for ins in blocks:
        assert len(ins) > 100 and asmgen.wide_store_hazards(ins) == []
    # the checker sees a planted one, and the emitter repairs it
    bad = [("ds_write_b128", "v1", "v[8:11]", 0), ("v_mov_b32", "v9", 0)]
    assert asmgen.wide_store_hazards(bad) == [(0, 1)]
    e = asmgen.Emit()
    for t in bad:
        e(*t)
    assert e.ins == [bad[0], ("s_nop", 1), bad[1]] and asmgen.wide_store_hazards(e.ins) == []


def test_register_homes_of_the_cut_loop_are_disjoint(prog):
    """LoopSplit.own(): what a wavefront of the cut loop keeps resident through the fused y0 iterations -- its solve entries of L
    (lhome), its x / z / equality-row y words (shome) and 1/D of its own unknowns (dhome) in VGPRs, its other constants in AGPRs
    (ahome) -- must not overlap each other, the W registers the wave computes in, or the AGPRs that hold its 1/D, the once-only
    items and the y0 homes; the registers of the other regime (chome / yhome: bodies without resident iterates) are filled on
    the other path only and may coincide with shome."""
    from robobee3d_amd import codegen_qp
    asmqp, _, p = prog
    sp = asmqp.LoopSplit(p, 4)
    assert sp.cut
    for w in range(4):
        o = sp.own(w)
        assert len(o.shome) >= 60 and len(o.ahome) >= 40 and len(o.dhome) >= 25 and len(o.lhome) >= 40
        wregs = {p.wreg[k] for k in p.nonleaf if o.k(k)} | {p.wreg[k] for k in o.top}
        groups = [set(o.lhome.values()), set(o.shome.values()), set(o.dhome.values()), wregs]
        for a_ in range(len(groups)):
            assert len(groups[a_]) == [len(o.lhome), len(o.shome), len(o.dhome), len(wregs)][a_]
            for b_ in range(a_ + 1, len(groups)):
                assert not groups[a_] & groups[b_], (w, a_, b_)
        # the other regime shares registers with shome only
        other = set(o.chome.values()) | set(o.yhome.values())
        assert not other & (set(o.lhome.values()) | wregs)
        # pairs the packed operations read as pairs sit on aligned register pairs
        for home in (o.shome, o.chome, o.yhome):
            for w_, r_ in home.items():
                if w_ % 2 == 0 and w_ + 1 in home:
                    assert r_ % 2 == 0 and home[w_ + 1] == r_ + 1
        # 1/D homes: left-over registers, ring slots 2..5, the unpacked bodies' AGPR temporaries -- nothing the fused bodies use
        allowed = set(range(asmqp.V_W, asmqp.V_W + len(p.nonleaf))) | set(range(p.V_RING + 4 * asmqp.Y_NRING, p.V_RING + 4 * asmqp.NRING)) | \
            set(range(p.V_AT, p.V_AT + asmqp.N_AT))
        assert set(o.dhome.values()) <= allowed
        # AGPRs: constants vs 1/D of the unknowns this wave uses, handed-over 1/D, once-only items, y0 homes
        agpr_used = {k for k in range(p.nk) if o.k(k)} | set(o.hand_in) | {h[1] for h in p.once.values() if h[0] == "A"} | \
            {h[1] for h in p.y0_home.values() if isinstance(h, tuple)}
        assert len(set(o.ahome.values())) == len(o.ahome) and not set(o.ahome.values()) & agpr_used


def test_no_memory_instruction_inside_the_iterations_of_the_cut_loop(prog):
    """What round 5 ends with: between the first and the last iteration of a tick no wavefront of the cut loop issues a VMEM
    instruction -- on the loose route (the reference's problem) nor on the general one (finite bounds: its read-only items sit in
    AGPRs, Own.ghome) -- and on the loose route the only LDS instructions left are the halves' exchange words."""
    from robobee3d_amd import codegen_qp
    asmqp, _, p = prog
    s = p.s
    eq = codegen_qp.ASM_STRUCTURES["p5f10"]
    res = asmqp.ResPlan(s, eq, codegen_qp.ASM_RES_ITEM0)
    for loose in (True, False):
        ins = asmqp.loop_group_program(s, eq, res, 4, loose=loose)[0]
        bodies, k = [], 0
        while k < len(ins):
            if ins[k] == ("label", "7"):
                j = next(q for q in range(k, len(ins)) if ins[q][0].startswith("s_cbranch") and ins[q][1] == "7b")
                bodies.append(ins[k:j])
                k = j
            k += 1
        assert len(bodies) == (8 if loose else 4)         # (loose: the fused y0 body and the non-zero-multiplier body of each wavefront)
        for body in bodies:
            assert not any(t[0].startswith("global_") for t in body)
            assert sum(t[0] == "s_barrier" for t in body) == 2
        if loose:
            fused = bodies[0::2]
            for body in fused:
                lds = [t for t in body if t[0].startswith("ds_")]
                assert len(lds) <= 4 and len(body) <= 380, (len(lds), len(body))

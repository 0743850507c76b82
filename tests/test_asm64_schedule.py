"""CPU check of the generated fp64 ADMM assembly (robobee3d_amd/asmgen64.py, BASELINE config 2): the emitted instruction
list is interpreted in exact-rounded float64 (one lane) and compared with the fp64 oracle's iterates after the same
number of iterations from the same factorisation (reference: osqp.c:354-370 loop body, auxil.c:164-228,
qdldl.c:250-293)."""
import numpy as np
import pytest

from conftest import golden


@pytest.fixture(scope="module")
def prog():
    from robobee3d_amd import asmgen64
    ins, s = asmgen64.program()
    return asmgen64, ins, s


def _case(oracle_built, g, ins, s, seq, k, iters):
    from robobee3d_amd import asmgen
    perm = np.array(s.perm, np.int32)
    args = (seq["p0"][k], seq["R0"][k], seq["dq0"][k], seq["pdes"][k], seq["dpdes"][k], seq["sdes"][k],
            float(seq["actualT0"][k]))
    pre = [np.asarray(seq[n][k], np.float64) for n in ("pre_x", "pre_y", "pre_z")]

    def fresh(mi):
        o = oracle_built.Oracle(np.float64, perm=perm, maxIter=mi)
        o.set_canonical(True, np.asarray(seq["pre_E3"][k], np.float64))
        o.set_iterates(*pre)
        o.set_T0(float(seq["pre_T0"][k]))
        o.update(*args)
        return o

    o = fresh(0)  # factorisation only
    ws = np.full(asmgen.WS_ROWS, np.nan)
    ctrl = np.zeros(123)
    lds = np.zeros(320)
    lds[:213] = o.get("L_x")
    lds[213:297] = o.get("Ddinv")
    ws[asmgen.FAC_Q:asmgen.FAC_Q + 45] = o.get("q")
    l, u = o.get("l"), o.get("u")
    ws[asmgen.FAC_LOEQ:asmgen.FAC_LOEQ + 36] = l[:36]
    ws[asmgen.FAC_M:asmgen.FAC_M + 3] = l[36:]
    ws[asmgen.FAC_M + 3:asmgen.FAC_M + 6] = u[36:]
    ws[asmgen.FAC_M + 6:asmgen.FAC_M + 9] = o.get("rho_vec")[36:]
    ws[asmgen.FAC_M + 9:asmgen.FAC_M + 12] = o.get("rho_inv_vec")[36:]
    ctrl[:45], ctrl[45:84], ctrl[84:123] = pre
    ws[asmgen.WS_DS:asmgen.WS_DS + 45] = 100.0 + np.arange(45)      # phase A's D, E: the epilogue stages them in LDS
    ws[asmgen.WS_ES:asmgen.WS_ES + 39] = 200.0 + np.arange(39)
    x_before_last = fresh(iters - 1).get("x") if iters > 1 else pre[0]
    g.simulate(ins, ws, ctrl, iters, lds)
    o2 = fresh(iters)
    for name, sl in (("x", slice(0, 45)), ("y", slice(45, 84)), ("z", slice(84, 123))):
        ref = o2.get(name)
        assert np.abs(ctrl[sl] - ref).max() <= 1e-12 * max(1e-6, np.abs(ref).max()), (name, k, iters)
    # what phase C reads from LDS: x, y where the loop keeps them, z, D, E, the thrust-row bounds, the captures
    assert np.array_equal(lds[g.LW_X:g.LW_X + 45], ctrl[:45]) and np.array_equal(lds[g.LW_Y:g.LW_Y + 39], ctrl[45:84])
    assert np.array_equal(lds[g.PC_Z:g.PC_Z + 39], ctrl[84:123])
    assert np.array_equal(lds[g.PC_DS:g.PC_DS + 45], 100.0 + np.arange(45))
    assert np.array_equal(lds[g.PC_ES:g.PC_ES + 39], 200.0 + np.arange(39))
    assert np.array_equal(lds[g.PC_LO3:g.PC_LO3 + 3], l[36:]) and np.array_equal(lds[g.PC_UP3:g.PC_UP3 + 3], u[36:])
    xp = lds[g.PC_XP:g.PC_XP + 45]
    assert np.abs(xp - x_before_last).max() <= 1e-12 * np.abs(x_before_last).max()
    # delta_y of the last iteration = y_new - y_prev
    yprev = fresh(iters - 1).get("y") if iters > 1 else pre[1]
    dy = lds[g.PC_DY:g.PC_DY + 39]
    ref = o2.get("y") - yprev
    assert np.abs(dy - ref).max() <= 1e-9 * max(1e-6, np.abs(o2.get("y")).max())


@pytest.fixture(scope="module")
def prog_quad():
    """the same ADMM phase with one robot per lane quad (asmquad64.py): iterations 2.. on the four lanes of a quad"""
    from robobee3d_amd import asmgen64

    class G:        # asmgen64 with simulate() bound to the program's permutation (the quad plan needs it)
        pass
    ins, s = asmgen64.program(quad=True)
    g = G()
    for k in dir(asmgen64):
        if not k.startswith("__"):
            setattr(g, k, getattr(asmgen64, k))
    g.simulate = lambda ins_, ws, ctrl, iters, lds: asmgen64.simulate(ins_, ws, ctrl, iters, lds, perm=s.perm)
    return g, ins, s


@pytest.mark.parametrize("iters", [1, 2, 3, 6])
def test_generated_fp64_quad_program_matches_oracle(oracle_built, prog_quad, iters):
    """the quad form against the same harness: iterates 1e-12 from the fp64 oracle, everything phase C reads back in the
    lane's LDS words (x, y, z, x_prev, delta_y, the thrust-row bounds), l back in its AGPR homes for the epilogue; the
    four lanes of the quad agree on every LDS word and thrust-row register (asserted by the interpreter's hook)."""
    from robobee3d_amd import asmgen64
    g, ins, s = prog_quad
    seq = golden("seq_iter50.npz")
    for k in (0, 11):
        _case(oracle_built, g, ins, s, seq, k, iters)
        assert (asmgen64.simulate.last_quad_instructions > 0) == (iters >= 2)


def test_fp64_quad_iteration_size_and_table(prog_quad):
    from robobee3d_amd import asmquad64
    g, ins, s = prog_quad
    qb, qe = ins.index(("quad_begin",)), ins.index(("quad_end",))
    sec = ins[qb:qe]
    l17, l18 = sec.index(("label", "17")), sec.index(("label", "18"))
    body = sec[l17:l18]
    f64 = sum(1 for t in body if t[0] in ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64"))
    # the one-lane middle iteration: 1 483 instructions, 786 of them fp64
    assert len(body) <= 1200 and f64 <= 330, (len(body), f64)
    assert not any(t[0] == "s_mov_b64" and t[1] == "exec" for t in body)
    plan = asmquad64.plan_for(s)
    tab = asmquad64.table(plan)
    assert tab.shape == (4, asmquad64.NTAB) and plan.ncoef <= asmquad64.NTAB
    # round 5: the per-lane coefficients are gathered ONCE per step: AGPR pairs for the first NAC instructions, a compact LDS
    # array (uniform addresses, two coefficients per ds_read_b128) for the rest -- no per-lane address in the iterations
    assert asmquad64.coef_word(plan.ncoef - 1) < asmquad64.ZERO_WORD and asmquad64.coef_word(asmquad64.NAC) % 2 == 0
    lds_rd = [t for t in body if t[0].startswith("ds_read")]
    assert all(t[0] == "ds_read_b128" for t in lds_rd) and len(lds_rd) == (plan.ncoef - asmquad64.NAC) // 2
    assert not any(t[0].startswith("ds_write") or t[0].startswith("global_") for t in body)
    assert len(body) <= 1010, len(body)
    assert (tab[3] == asmquad64.rel_addr(asmquad64.ZERO_WORD)).all()            # lane 3 of a quad idles on the zero word
    assert (tab % 8 == 0).all() and tab.max() < 160 * 1024 // 64 * 64
    # every entry of L is addressed exactly twice (once per solve direction), by the lane that owns its destination
    words, counts = np.unique(tab[:3, :plan.ncoef][tab[:3, :plan.ncoef] != asmquad64.rel_addr(asmquad64.ZERO_WORD)], return_counts=True)
    assert len(words) == len(s.L_i) and (counts == 2).all()


def test_fp64_quad_stream_assembles(prog_quad):
    import os, subprocess, tempfile
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    g, ins, s = prog_quad
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write("\n".join(g.fmt(t) for t in ins if t[0] not in g.PSEUDO) + "\n")
    try:
        r = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, f.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[:3000]
    finally:
        os.unlink(f.name)


@pytest.mark.parametrize("iters", [1, 2, 3, 6])
def test_generated_fp64_admm_program_matches_oracle(oracle_built, prog, iters):
    g, ins, s = prog
    seq = golden("seq_iter50.npz")
    for k in (0, 11):
        _case(oracle_built, g, ins, s, seq, k, iters)


def test_fp64_register_and_lds_plan(prog):
    import re
    g, ins, s = prog
    assert g.V_W + 2 * s.nk == g.V_C and g.V_C + 2 * 5 * s.N == g.V_RING
    assert g.V_RING + 4 * g.NSLOT == g.V_AT and g.V_AT + 2 * g.N_AT == g.V_TT and g.V_TT + 2 * g.N_TT == g.V_END <= 256
    assert g.A_D + 2 * s.nk == g.A_Z and g.A_Z + 2 * 2 * s.N * 6 <= 256
    assert g.LW_X == len(s.L_i) and g.LW_Y == g.LW_X + s.nx and g.LW_END == g.LW_Y + s.nc
    assert g.LW_END * 8 <= g.LDS_BYTES_PER_LANE and g.LDS_BYTES_PER_LANE * 64 <= 160 * 1024
    for t in ins:
        for x in t[1:]:
            if isinstance(x, str):
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", x):
                    assert int(b) < g.V_END
                for a in re.findall(r"\bv(\d+)\b", x):
                    assert int(a) < g.V_END
        if t[0].startswith("ds_"):
            assert 0 <= t[3] < 65536 and t[3] % 8 == 0


def test_fp64_stream_assembles(prog):
    import os, subprocess, tempfile
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    g, ins, s = prog
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write("\n".join(g.fmt(t) for t in ins) + "\n")
    try:
        r = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, f.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[:3000]
    finally:
        os.unlink(f.name)


def _ruiz_numpy(s, P, A, q, passes):
    """scaling.c:44-156 as csrc/umpc_step.h states it (fp64, exact sqrt and divide)"""
    nx, nc = s.nx, s.nc
    P, A, q = P.copy(), A.copy(), q.copy()
    lim = lambda v: 1.0 if v < 1e-4 else min(v, 1e4)
    col = [range(s.A_p[j], s.A_p[j + 1]) for j in range(nx)]
    c = 1.0
    for _ in range(passes):
        Dt = np.array([1.0 / np.sqrt(lim(max([abs(P[j])] + [abs(A[p]) for p in col[j]]))) for j in range(nx)])
        rows = [[] for _ in range(nc)]
        for j in range(nx):
            for p in col[j]:
                rows[s.A_i[p]].append(p)
        Et = np.array([1.0 / np.sqrt(lim(max(abs(A[p]) for p in rows[i]))) for i in range(nc)])
        P = (P * Dt) * Dt
        for j in range(nx):
            for p in col[j]:
                A[p] = (A[p] * Et[s.A_i[p]]) * Dt[j]
        q = q * Dt
        ct = 1.0 / lim(max(np.abs(P).sum() / nx, lim(np.abs(q).max())))
        P, q, c = P * ct, q * ct, c * ct
    return P, A, q, c


@pytest.mark.parametrize("passes", [1, 10])
def test_generated_fp64_ruiz_block_matches_numpy(passes):
    from robobee3d_amd import asmgen64 as g
    ins, s = g.ruiz_program()
    rng = np.random.default_rng(passes)
    nx, nnz = s.nx, len(s.A_i)
    for scale in (1.0, 1e-6, 3e5):       # the last two drive limit_scaling's branches
        P = np.abs(rng.normal(size=nx)) * 10 * scale + 1e-3 * scale
        A = rng.normal(size=nnz) * scale
        A[rng.random(nnz) < 0.3] = 1.0
        q = rng.normal(size=nx) * scale
        lds = np.zeros(320)
        lds[g.RZ_P:g.RZ_P + nx], lds[g.RZ_Q:g.RZ_Q + nx], lds[g.RZ_A:g.RZ_A + nnz] = P, q, A
        ws, ctrl = np.zeros(4), np.zeros(4)
        g.simulate(ins, ws, ctrl, passes, lds)
        Pr, Ar, qr, cr = _ruiz_numpy(s, P, A, q, passes)
        for got, ref in ((lds[g.RZ_P:g.RZ_P + nx], Pr), (lds[g.RZ_A:g.RZ_A + nnz], Ar), (lds[g.RZ_Q:g.RZ_Q + nx], qr)):
            assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), (passes, scale)
        assert abs(lds[g.RZ_C] - cr) <= 1e-13 * cr


def test_fp64_ruiz_stream_assembles_and_fits():
    import os, re, subprocess, tempfile
    from robobee3d_amd import asmgen64 as g
    ins, s = g.ruiz_program()
    for t in ins:
        for x in t[1:]:
            if isinstance(x, str):
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", x):
                    assert int(b) < g.V_END
                for a in re.findall(r"\bv(\d+)\b", x):
                    assert int(a) < g.V_END
                for a in re.findall(r"\ba(\d+)\b", x):
                    assert int(a) < 256
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write("\n".join(g.fmt(t) for t in ins) + "\n")
    try:
        r = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, f.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[:3000]
    finally:
        os.unlink(f.name)


def test_fp64_completion_model_catches_a_missing_wait(oracle_built, prog):
    """the interpreter's in-order completion model guards the s_waitcnt placement of the fp64 streams too"""
    g, ins, s = prog
    seq = golden("seq_iter50.npz")
    stripped = [t for t in ins if t[0] != "s_waitcnt"]
    with pytest.raises(AssertionError):
        _case(oracle_built, g, stripped, s, seq, 0, 2)


def test_rhs_structure_is_the_fp32_generators():
    """asmgen64.rhs_structure (which q entries / dynamics rows get an AGPR home in the fp64 loop, the others being
    structural zeros of uprightmpc2.c:126-207) states the same facts as asmstep.Struct.qzero / lzero of the fp32 stream."""
    from robobee3d_amd import asmgen64, asmstep, symbolic
    s = symbolic.analyse(3)
    st = asmstep.Struct(3)
    qnz, lnz = asmgen64.rhs_structure(s)
    assert set(qnz) == set(range(s.nx)) - set(st.qzero) and set(lnz) == set(range(st.neq)) - set(st.lzero)
    h = asmgen64.homes(s)
    assert len(h) == 43 and max(h.values()) + 1 <= 255 and min(h.values()) == asmgen64.A_H


def _resid_numpy(s, g, x, y, z, D, E, par, c, w8, dt, q):
    """update_info's norms as csrc/umpc_step.h states them (auxil.c:243-307), plain float64 with exact divisions"""
    def raw(tag):
        return {"c": lambda: tag[1], "dt": lambda: dt, "T0dt": lambda: par[0], "s0": lambda: par[1 + tag[1]],
                "Btau": lambda: par[4 + tag[1]]}[tag[0]]()
    Ax, Aty = np.zeros(s.nc), np.zeros(s.nx)
    for j in range(s.nx):
        for p in range(s.A_p[j], s.A_p[j + 1]):
            i = s.A_i[p]
            a = (raw(s.A_tag[p]) * E[i]) * D[j]
            Ax[i] += a * x[j]
            Aty[j] += a * y[i]
    Px = np.array([(((w8[g.weight_class(s, j)] * D[j]) * D[j]) * c) * x[j] for j in range(s.nx)])
    return dict(pri=np.abs((Ax - z) / E).max(), dua=np.abs(((q + Px) + Aty) / D).max(), nz=np.abs(z / E).max(),
                nAx=np.abs(Ax / E).max(), nq=np.abs(q / D).max(), nAty=np.abs(Aty / D).max(), nPx=np.abs(Px / D).max())


def test_generated_fp64_residual_block_matches_numpy():
    """asmgen64.resid_program (round 3: the residual norms of the fp64 step as assembly instead of hipcc's phase C):
    interpreted on random iterates, scalings and raw matrix parameters against a numpy statement of the same norms;
    a NaN in x reaches the NaN accumulator."""
    from robobee3d_amd import asmgen64 as g, asmgen
    ins, s = g.resid_program()
    rng = np.random.default_rng(11)
    qnz, _ = g.rhs_structure(s)
    for trial in range(4):
        x, y, z = rng.normal(size=s.nx), rng.normal(size=s.nc) * 10, rng.normal(size=s.nc)
        D, E = np.exp(rng.normal(size=s.nx)), np.exp(rng.normal(size=s.nc))
        par, c, w8, dt = rng.normal(size=10), float(np.exp(rng.normal())), np.exp(rng.normal(size=8)), 5.0
        q = np.zeros(s.nx)
        q[qnz] = rng.normal(size=len(qnz)) * 3
        if trial == 3:
            x[7] = np.nan
        lds = np.zeros(320)
        lds[g.LW_X:g.LW_X + s.nx], lds[g.LW_Y:g.LW_Y + s.nc], lds[g.PC_Z:g.PC_Z + s.nc] = x, y, z
        lds[g.PC_DS:g.PC_DS + s.nx], lds[g.PC_ES:g.PC_ES + s.nc] = D, E
        lds[g.RS_PAR:g.RS_PAR + 10], lds[g.RS_C], lds[g.RS_DT] = par, c, dt
        lds[g.RS_W:g.RS_W + 8] = w8
        ws, ctrl = np.zeros(asmgen.WS_ROWS), np.zeros(4)
        ws[asmgen.FAC_Q:asmgen.FAC_Q + s.nx] = q
        g.simulate(ins, ws, ctrl, 1, lds)
        out = lds[g.RS_OUT:g.RS_OUT + 8]
        if trial == 3:
            assert np.isnan(out[g.N_NAN])
            continue
        ref = _resid_numpy(s, g, x, y, z, D, E, par, c, w8, dt, q)
        for k, name in ((g.N_PRI, "pri"), (g.N_DUA, "dua"), (g.N_Z, "nz"), (g.N_AX, "nAx"), (g.N_Q, "nq"), (g.N_ATY, "nAty"),
                        (g.N_PX, "nPx")):
            assert abs(out[k] - ref[name]) <= 1e-12 * max(1.0, abs(ref[name])), (trial, name, out[k], ref[name])
        assert out[g.N_NAN] == 0.0


def test_fp64_residual_stream_assembles_and_fits():
    import os, re, subprocess, tempfile
    from robobee3d_amd import asmgen64 as g
    ins, s = g.resid_program()
    for t in ins:
        for x in t[1:]:
            if isinstance(x, str):
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", x):
                    assert int(b) < g.V_END
                for a, b in re.findall(r"a\[(\d+):(\d+)\]", x):
                    assert int(b) < 256
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write("\n".join(g.fmt(t) for t in ins) + "\n")
    try:
        r = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, f.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[:3000]
    finally:
        os.unlink(f.name)


@pytest.mark.parametrize("passes", [1, 10])
def test_generated_fp64_quad_ruiz_block_matches_numpy(passes):
    """the Ruiz passes on the lane quad (asmquad64.ruiz_program): every lane reads its own third of P, q, A from its LDS
    slice, the nine input columns are reduced / scaled across lanes, and all four slices end with the whole equilibrated
    problem -- equal to the numpy statement of scaling.c:44-156 to 1e-13 (the sum of |P_j| is associated per lane)."""
    from robobee3d_amd import asmgen64 as g, asmquad64 as q4
    ins, s = q4.ruiz_program()
    rng = np.random.default_rng(passes)
    nx, nnz = s.nx, len(s.A_i)
    for scale in (1.0, 1e-6, 3e5):
        P = np.abs(rng.normal(size=nx)) * 10 * scale + 1e-3 * scale
        A = rng.normal(size=nnz) * scale
        A[rng.random(nnz) < 0.3] = 1.0
        q = rng.normal(size=nx) * scale
        lds = np.zeros((4, 320))
        lds[:, g.RZ_P:g.RZ_P + nx], lds[:, g.RZ_Q:g.RZ_Q + nx], lds[:, g.RZ_A:g.RZ_A + nnz] = P, q, A
        V, A4 = np.zeros((4, 256), np.uint32), np.zeros((4, 256), np.uint32)
        from robobee3d_amd.asmgen import S_ITERS
        _, n = q4.simulate(ins, 0, V, A4, lds, {S_ITERS: passes}, None)
        Pr, Ar, qr, cr = _ruiz_numpy(s, P, A, q, passes)
        for ln in range(4):
            for got, ref in ((lds[ln, g.RZ_P:g.RZ_P + nx], Pr), (lds[ln, g.RZ_A:g.RZ_A + nnz], Ar), (lds[ln, g.RZ_Q:g.RZ_Q + nx], qr)):
                assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), (passes, scale, ln)
            assert abs(lds[ln, g.RZ_C] - cr) <= 1e-13 * cr
        assert n < 400 + 640 + passes * 800


def test_fp64_quad_ruiz_stream_assembles():
    import os, subprocess, tempfile
    from robobee3d_amd import asmgen64 as g, asmquad64 as q4
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not available")
    ins, s = q4.ruiz_program()
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write("\n".join(g.fmt(t) for t in ins if t[0] not in g.PSEUDO) + "\n")
    try:
        r = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, f.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[:3000]
    finally:
        os.unlink(f.name)

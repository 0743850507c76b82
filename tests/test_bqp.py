"""General-structure batch QP solver: csrc/umpc_bqp.hip walks symbolic tables
built by robobee3d_amd/qpstruct.py. The oracle is oracle/osqp_table.py (numpy, vectorised over robots only);
it is PINNED by reproducing, bit for bit in fp32 with the reference's KKT permutation, the C restatement
(oracle/umpc_oracle.c, itself bit-identical to the compiled reference on tests/golden/seq_iter*.npz) on the
uprightmpc2 N = 3 problem."""
import os
import sys

import numpy as np
import pytest

from conftest import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def raw_uprightmpc2_qp(g, idx, dtype):
    """Raw (Pv, Av, q, l, u) [rows][B] of the uprightmpc2 N = 3 QP from a seq_iter fixture: constants of A as
    assembled (+-1, template_controllers.py:28-63), the 48 state-dependent entries from the fixture's Ax."""
    from robobee3d_amd import symbolic
    A_p, A_i, A_tag = symbolic.build_A(3)
    B = len(idx)
    Av = np.zeros((len(A_i), B), dtype)
    for p, tag in enumerate(A_tag):
        if tag[0] == 'c':
            Av[p] = tag[1]
    axidx = symbolic.ax_idx(3)
    Av[axidx] = g["Ax"][idx].T
    return A_p, A_i, g["Px"][idx].T.astype(dtype), Av, g["q"][idx].T.astype(dtype), g["l"][idx].T.astype(dtype), \
        g["u"][idx].T.astype(dtype)


def test_qpstruct_matches_specialised_symbolic(structure):
    """qpstruct.analyse_qp on the uprightmpc2 pattern = symbolic.analyse = the reference's generated tables."""
    from robobee3d_amd import qpstruct, symbolic
    A_p, A_i, _ = symbolic.build_A(3)
    s = qpstruct.analyse_qp(45, 39, A_p, A_i, list(range(45)), perm=structure["perm"])
    assert s.K_p == [int(v) for v in structure["K_p"]] and s.K_i == [int(v) for v in structure["K_i"]]
    assert s.L_p == [int(v) for v in structure["L_p"]] and s.L_i == [int(v) for v in structure["L_i"]]
    assert s.etree == [int(v) for v in structure["etree"]]
    own = qpstruct.analyse_qp(45, 39, A_p, A_i, list(range(45)))
    assert own.nnzL == 213 and own.perm == symbolic.analyse(3).perm
    # CSR views are permutations of the CSC ones
    assert sorted(own.tables["Ar_k"]) == list(range(own.nnzA)) and sorted(own.tables["Lr_k"]) == list(range(own.nnzL))
    assert int(own.blob[0]) == 45 and int(own.blob[6]) == own.nrows


@pytest.mark.parametrize("fixture", ["seq_iter2", "seq_iter50"])
def test_table_oracle_is_bitwise_the_c_oracle(oracle_built, structure, fixture):
    import osqp_table
    g = golden(fixture + ".npz")
    n = min(len(g["p0"]), 48)
    idx = np.arange(n)
    iters = int(g["maxIter"])
    perm = structure["perm"]
    A_p, A_i, Pv, Av, q, l, u = raw_uprightmpc2_qp(g, idx, np.float32)
    Eprev = np.ones((39, n), np.float32)
    Eprev[36:] = g["pre_E3"][idx].T
    r = osqp_table.solve(45, 39, A_p, A_i, list(range(45)), perm, Pv, Av, q, l, u, g["pre_x"][idx].T,
                         g["pre_y"][idx].T, g["pre_z"][idx].T, Eprev, osqp_table.Settings(max_iter=iters),
                         dtype=np.float32)
    for k in idx:
        o = oracle_built.Oracle(np.float32, perm=perm, maxIter=iters)
        o.set_canonical(True, g["pre_E3"][k])
        o.set_iterates(g["pre_x"][k], g["pre_y"][k], g["pre_z"][k])
        o.set_T0(g["pre_T0"][k])
        o.update(g["p0"][k], g["R0"][k], g["dq0"][k], g["pdes"][k], g["dpdes"][k], g["sdes"][k], g["actualT0"][k])
        for name, mine in (("D", r["D"]), ("E", r["E"]), ("L_x", r["L"]), ("Ddinv", r["Dinv"]), ("x", r["x"]),
                           ("y", r["y"]), ("z", r["z"]), ("sol_x", r["sol_x"])):
            assert np.array_equal(o.get(name), mine[:, k], equal_nan=True), (fixture, k, name)
        assert o.get("c")[0] == r["c"][k] and o.get("pri_res")[0] == r["pri_res"][k]
        assert o.get("dua_res")[0] == r["dua_res"][k]
        assert int(o.get("status_val")[0]) == int(r["status"][k])


# ------------------------------------------------------------------------------------------------------
# structures of the reference's other MPC formulations (CPU)
# ------------------------------------------------------------------------------------------------------
def test_v1_structure_is_the_reference_pattern():
    """genqp.UprightMPC(3): CSC pattern of A, the diagonal pattern of P and Axidx (tests/golden/v1_qp.npz)."""
    from robobee3d_amd.batchqp import v1_structure
    g = golden("v1_qp.npz")
    st = v1_structure(int(g["N"]))
    assert st["A_p"] == g["A_indptr"].tolist() and st["A_i"] == g["A_indices"].tolist()
    assert st["P_cols"] == g["P_indices"].tolist() and st["Axidx"] == g["Axidx"].tolist()
    assert (st["n"], st["m"]) == (27, 27)


def p5f_dense_A(N, Ad, Bd):
    """planar/mpc_osqp_p5f.py:168-171 restated with numpy's kron (test-side)."""
    nx = 7
    Ax = np.kron(np.eye(N + 1), -np.eye(nx)) + np.kron(np.eye(N + 1, k=-1), Ad)
    Bu = np.kron(np.vstack([np.zeros((1, N)), np.eye(N)]), Bd.reshape(7, 1))
    Aeq = np.hstack([Ax, Bu])
    return np.vstack([Aeq, np.eye((N + 1) * nx + N)])


def test_p5f_structure_reproduces_the_kron_construction():
    from robobee3d_amd.batchqp import p5f_structure
    g = golden("planar_p5f.npz")
    st = p5f_structure(10)
    assert (st["n"], st["m"]) == (87, 164)
    for k in (5, 17, 40):
        Ad, Bd = g["lin_Ad"][k], g["lin_Bd"][k]
        lin = np.array([Ad[4, 3], Ad[5, 3], Bd[4], Bd[5], Bd[6]])
        vals = np.where(st["src"] < 0, st["cst"], st["cst"] * lin[np.maximum(st["src"], 0)])
        A = np.zeros((st["m"], st["n"]))
        for j in range(st["n"]):
            for p in range(st["A_p"][j], st["A_p"][j + 1]):
                A[st["A_i"][p], j] = vals[p]
        assert np.array_equal(A, p5f_dense_A(10, Ad, Bd))
    assert st["P_cols"] == [j for j in range(87) if (j < 77 and j % 7 in (1, 2, 3)) or j >= 77]
    # grouped=True: the same problem relabelled so that its connected components are contiguous (what PlanarP5fMPC solves)
    from robobee3d_amd.batchqp import qp_components
    sg = p5f_structure(10, grouped=True)
    vo, ro = sg["var_order"], sg["row_order"]
    assert sorted(vo) == list(range(87)) and sorted(ro) == list(range(164)) and list(ro[77:]) == [77 + j for j in vo]
    vc, rc = qp_components(sg["n"], sg["m"], sg["A_p"], sg["A_i"])
    assert vc == sorted(vc) and rc[:77] == sorted(rc[:77]) and [vc.count(c) for c in range(max(vc) + 1)] == [41, 39, 2, 2, 1, 1, 1]
    Ad, Bd = g["lin_Ad"][5], g["lin_Bd"][5]
    lin = np.array([Ad[4, 3], Ad[5, 3], Bd[4], Bd[5], Bd[6]])
    vals = np.where(sg["src"] < 0, sg["cst"], sg["cst"] * lin[np.maximum(sg["src"], 0)])
    A = np.zeros((164, 87))
    for j in range(87):
        for p in range(sg["A_p"][j], sg["A_p"][j + 1]):
            A[ro[sg["A_i"][p]], vo[j]] = vals[p]
    assert np.array_equal(A, p5f_dense_A(10, Ad, Bd))
    for key, order in (("q", vo), ("l", ro), ("u", ro)):
        assert np.array_equal(sg[key], st[key][order])
    Pc, Pg = np.zeros(87), np.zeros(87)
    Pc[st["P_cols"]], Pg[sg["P_cols"]] = st["Pv"], sg["Pv"]
    assert np.array_equal(Pg, Pc[vo])


def test_p5f_qp_data_is_the_reference_scripts(structure):
    """q, l, u, the dense A at getLin(0, 0, 0) and the weights of planar/mpc_osqp_p5f.py:87-147,
    evaluated from the script where it lies (tests/golden/make_golden.py planar_p5f), against p5f_structure()."""
    from robobee3d_amd.batchqp import p5f_structure, OSQP_INFTY
    g = golden("planar_p5f.npz")
    N, nx, nu = int(g["N"]), int(g["nx"]), int(g["nu"])
    st = p5f_structure(N)
    assert (st["n"], st["m"]) == g["qp_A"].shape[::-1] == (87, 164)
    np.testing.assert_array_equal(st["q"], g["qp_q"])
    # the script's +-inf box rows reach OSQP as +-OSQP_INFTY (osqp's Python interface clips them at setup)
    np.testing.assert_array_equal(st["l"], np.maximum(g["qp_l"], -OSQP_INFTY))
    np.testing.assert_array_equal(st["u"], np.minimum(g["qp_u"], OSQP_INFTY))
    # P = block_diag(kron(eye(N), Q), QN, kron(eye(N), R)) (:116-117; the statement itself cannot run under this
    # scipy, its operands are the evaluated Q, QN, R)
    Pdiag = np.hstack([np.tile(np.diag(g["qp_Q"]), N), np.diag(g["qp_QN"]), np.full(N * nu, float(g["qp_R"]))])
    Pfull = np.zeros(st["n"])
    Pfull[st["P_cols"]] = st["Pv"]
    np.testing.assert_array_equal(Pfull, Pdiag)
    assert np.count_nonzero(g["qp_Q"] - np.diag(np.diag(g["qp_Q"]))) == 0
    # A: the script's dense kron construction at the LTI point == our pattern filled with getLin(0, 0, 0)
    k0 = int(np.nonzero((g["lin_u"] == 0) & (np.signbit(g["lin_u"]) == False))[0][0])   # noqa: E712
    Ad, Bd = None, None
    A = np.zeros((st["m"], st["n"]))
    lin0 = None
    # getLin(0, 0, 0): entries (4,3), (5,3) of Ad and Bd[4:7] -- recover them from the script's own A
    Ascr = g["qp_A"]
    lin0 = np.array([Ascr[nx + 4, 3], Ascr[nx + 5, 3], Ascr[nx + 4, (N + 1) * nx], Ascr[nx + 5, (N + 1) * nx],
                     Ascr[nx + 6, (N + 1) * nx]])
    for j in range(st["n"]):
        for p in range(st["A_p"][j], st["A_p"][j + 1]):
            A[st["A_i"][p], j] = st["cst"][p] if st["src"][p] < 0 else st["cst"][p] * lin0[st["src"][p]]
    np.testing.assert_array_equal(A, Ascr)
    assert list(g["skipped"]) == ["line 116: ValueError", "line 136: NameError", "line 147: NameError"]


def test_table_oracle_converges_on_v1_and_p5f():
    """Self-consistency of the oracle on the two unpinned problems: run long, KKT residuals vanish."""
    import osqp_table
    from robobee3d_amd.batchqp import p5f_structure, v1_structure
    from robobee3d_amd import qpstruct
    g = golden("v1_qp.npz")
    st = v1_structure(3)
    s = qpstruct.analyse_qp(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"])
    k = [0, 1, 2, 3]
    z = lambda r: np.zeros((r, len(k)))
    r = osqp_table.solve(27, 27, st["A_p"], st["A_i"], st["P_cols"], s.perm, g["Pdata"][k].T, g["Adata"][k].T,
                         g["q"][k].T, g["l"][k].T, g["u"][k].T, z(27), z(27), z(27), np.ones((27, len(k))),
                         osqp_table.Settings(max_iter=1500))
    assert np.all(r["status"] == 1) and r["pri_res"].max() < 1e-6 and r["dua_res"].max() < 1e-6
    x = r["sol_x"]
    for c, kk in enumerate(k):
        A = np.zeros((27, 27))
        for j in range(27):
            for p in range(st["A_p"][j], st["A_p"][j + 1]):
                A[st["A_i"][p], j] = g["Adata"][kk][p]
        Ax = A @ x[:, c]
        assert np.all(Ax >= g["l"][kk] - 1e-5) and np.all(Ax <= g["u"][kk] + 1e-5)


def test_p5f_elimination_order_keeps_fp32_accurate_and_cuts_the_chains():
    """The order PlanarP5fMPC and the build-time specialisation eliminate in (batchqp.p5f_analysis: each horizon chain cut in two
    halves under a small separator, min-fill inside, variables without a cost term held back behind a unit-coefficient
    dynamics row): (1) no more fill than plain min-fill, (2) an elimination tree of two balanced subtrees per chain, (3) fp32
    stays fp32 -- the float32 table oracle within 5e-6 of the float64 one on the reference's own linearisations
    (planar_p5f.npz: getLin at 64 random states), after 3 and after 50 iterations. Without the hold-back the same check reads
    7.6e-2 (five digits lost to pivots of 2e-6: the GPU caught it, the interpreter's random data cannot)."""
    import osqp_table
    from robobee3d_amd import batchqp, qpstruct, symbolic
    g = golden("planar_p5f.npz")
    st, s = batchqp.p5f_analysis(10)
    n, m = st["n"], st["m"]
    plain = qpstruct.analyse_qp(n, m, st["A_p"], st["A_i"], st["P_cols"])
    assert s.nnzL == 264 <= plain.nnzL
    from robobee3d_amd import asmqp, codegen_qp
    sp = asmqp.LoopSplit(asmqp.Plan(s, codegen_qp.ASM_STRUCTURES["p5f10"]), 4)
    assert sorted(sp.cut) == [0, 1] and max(sp.load) <= 64 and min(sp.load) >= 61          # (118 + 112 + 12 + 9 unknowns before)
    for cut in sp.cut.values():
        assert len(cut["T"]) <= 2 and min(len(cut["A"]), len(cut["B"])) >= 54
    assert sum(len(sp.own(w).cross) for w in range(4)) <= 6          # entries of L that reach from a half B into a separator
    col = lambda v: np.asarray(v, np.float64)[:, None]
    z = lambda r: np.zeros((r, 1))
    cst, src = np.asarray(st["cst"], float), np.asarray(st["src"])

    def worst(perm, samples):
        w = 0.0
        for k in samples:
            lin = np.array([g["lin_Ad"][k][4, 3], g["lin_Ad"][k][5, 3], g["lin_Bd"][k][4], g["lin_Bd"][k][5], g["lin_Bd"][k][6]])
            Av = np.where(src >= 0, cst * lin[np.maximum(src, 0)], cst)[:, None]
            for iters in (3, 50):
                out = [osqp_table.solve(n, m, st["A_p"], st["A_i"], st["P_cols"], perm, col(st["Pv"]), Av, col(st["q"]), col(st["l"]),
                                        col(st["u"]), z(n), z(m), z(m), np.ones((m, 1)), osqp_table.Settings(max_iter=iters), dtype=dt)
                       for dt in (np.float64, np.float32)]
                d = np.abs(out[1]["x"].astype(np.float64) - out[0]["x"]) / np.maximum(1.0, np.abs(out[0]["x"]))
                w = max(w, float(d.max()))
        return w
    assert worst(s.perm, range(0, 64, 5)) <= 5e-6
    unheld = qpstruct.bisect_ordering(n, m, st["A_p"], st["A_i"], parts=st["parts"])
    assert worst(unheld, [17]) >= 1e-2


def _tiny_infeasible_qps():
    """(name, n, m, A_p, A_i, P_cols, Pv, Av, q, l, u, expected status): primal infeasible -- x0 >= 1 and x0 <= 0;
    dual infeasible -- min -x1 with x1 >= 0 only and no curvature in x1 (auxil.c:362-512)."""
    inf = 1e30
    pinf = ("primal infeasible", 2, 3, [0, 2, 3], [0, 1, 2], [0, 1], [1.0, 1.0], [1.0, 1.0, 1.0], [0.0, 0.0],
            [1.0, -inf, -1.0], [inf, 0.0, 1.0], -3)
    dinf = ("dual infeasible", 2, 2, [0, 1, 2], [0, 1], [0], [1.0], [1.0, 1.0], [0.0, -1.0],
            [-1.0, 0.0], [1.0, inf], -4)
    return [pinf, dinf]


def test_table_oracle_termination_mode_and_certificates():
    """check_termination = k: robots stop at the first multiple of k where a criterion holds (osqp.c:411-450), the
    frozen iterate is the fixed-count iterate at that count; the certificates fire on the two tiny infeasible QPs
    (status -3 / -4, NaN solution, cold-started iterates: auxil.c:539-564)."""
    import osqp_table
    g = golden("seq_iter50.npz")
    idx = np.arange(16)
    A_p, A_i, Pv, Av, q, l, u = raw_uprightmpc2_qp(g, idx, np.float64)
    st = golden("structure.npz")
    z = lambda r: np.zeros((r, len(idx)))
    args = (45, 39, A_p, A_i, list(range(45)), st["perm"], Pv, Av, q, l, u, z(45), z(39), z(39), np.ones((39, len(idx))))
    r = osqp_table.solve(*args, osqp_table.Settings(max_iter=4000, check_termination=25))
    # (fixed rho: pip osqp's adaptive rho is not reproduced, so a few robots need thousands of iterations and the
    # slowest end "solved inaccurate" at max_iter)
    assert np.all(np.isin(r["status"], (1, 2))) and np.all(r["iters"] % 25 == 0) and len(set(r["iters"])) > 2
    assert np.all((r["status"] == 1) == (r["iters"] < 4000)) and np.count_nonzero(r["status"] == 1) >= 12
    ra = osqp_table.solve(*args, osqp_table.Settings(max_iter=4000, check_termination=25, adaptive_rho_interval=25))
    assert np.all(ra["status"] == 1) and ra["iters"].max() <= 300 and ra["rho_updates"].max() >= 1     # adapt_rho converges all
    np.testing.assert_allclose(ra["sol_x"][:, r["status"] == 1], r["sol_x"][:, r["status"] == 1], rtol=5e-3, atol=2e-2)   # both eps-optimal
    for k in (0, 7):
        one = [a[:, k:k + 1] if isinstance(a, np.ndarray) and a.ndim == 2 else a for a in args]
        rf = osqp_table.solve(*one, osqp_table.Settings(max_iter=int(r["iters"][k])))
        assert np.array_equal(rf["x"][:, 0], r["x"][:, k]) and rf["status"][0] == 1
    for name, n, m, Ap, Ai, Pc, Pv, Av, q, l, u, want in _tiny_infeasible_qps():
        c = lambda a: np.array(a, np.float64)[:, None]
        r = osqp_table.solve(n, m, Ap, Ai, Pc, list(range(n + m)), c(Pv), c(Av), c(q), c(l), c(u), np.zeros((n, 1)),
                             np.zeros((m, 1)), np.zeros((m, 1)), np.ones((m, 1)),
                             osqp_table.Settings(max_iter=4000, check_termination=25))
        assert r["status"][0] == want, (name, r["status"], r["iters"])
        assert np.isnan(r["sol_x"]).all() and not r["x"].any() and not r["y"].any() and r["iters"][0] < 4000


# ------------------------------------------------------------------------------------------------------
# GPU: the table-driven kernel against the oracle
# ------------------------------------------------------------------------------------------------------
def _run_gpu_qp(n, m, A_p, A_i, P_cols, perm, Pv, Av, q, l, u, x, y, z, Eprev, dtype, max_iter):
    import torch
    from robobee3d_amd.batchqp import BatchQP
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    B = q.shape[1]
    qp = BatchQP(n, m, A_p, A_i, P_cols, B, tdt, perm=perm, max_iter=max_iter)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype)).cuda()
    qp.x.copy_(dev(x)); qp.y.copy_(dev(y)); qp.z.copy_(dev(z)); qp.Eprev.copy_(dev(Eprev))
    qp.solve(dev(Pv), dev(Av), dev(q), dev(l), dev(u))
    torch.cuda.synchronize()
    f = lambda t: t.cpu().numpy()
    return dict(x=f(qp.x), y=f(qp.y), z=f(qp.z), E=f(qp.Eprev), sol_x=f(qp.sol_x), sol_y=f(qp.sol_y),
                status=f(qp.status), info=f(qp.info)), qp


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gpu_qp_on_uprightmpc2_fixture(structure, dtype):
    """The generic kernel on the N = 3 uprightmpc2 QP with the reference's KKT permutation: the fp64 run agrees
    with the fp64 oracle to 1e-9, the fp32 run with the fp32 oracle (= the reference C, bit-pinned) to round-off."""
    import osqp_table
    g = golden("seq_iter50.npz")
    idx = np.arange(100)
    perm = structure["perm"]
    A_p, A_i, Pv, Av, q, l, u = raw_uprightmpc2_qp(g, idx, np.float64)
    Eprev = np.ones((39, len(idx)))
    Eprev[36:] = g["pre_E3"][idx].T
    args = (Pv, Av, q, l, u, g["pre_x"][idx].T, g["pre_y"][idx].T, g["pre_z"][idx].T, Eprev)
    ref = osqp_table.solve(45, 39, A_p, A_i, list(range(45)), perm, *args, osqp_table.Settings(max_iter=50), dtype=dtype)
    got, _ = _run_gpu_qp(45, 39, A_p, A_i, list(range(45)), perm, *args, dtype, 50)
    if dtype == np.float64:
        for k in ("x", "y", "z", "E", "sol_x", "sol_y"):
            assert np.allclose(got[k], ref[k], rtol=1e-9, atol=1e-11, equal_nan=True), k
        assert np.array_equal(got["status"], ref["status"])
    else:
        nbit = sum(int(np.array_equal(got[k], ref[k], equal_nan=True)) for k in ("x", "y", "z", "E", "sol_x"))
        print("fp32 arrays bit-identical to the oracle: %d of 5;" % nbit, "max |dE| rel", np.max(np.abs(got["E"] / ref["E"] - 1)),
              "max |dx|", np.max(np.abs(got["x"] - ref["x"])))
        sx = np.maximum(1.0, np.abs(ref["sol_x"]))
        assert np.nanmax(np.abs(got["sol_x"] - ref["sol_x"]) / sx) < 2e-3
        assert np.allclose(got["E"], ref["E"], rtol=1e-5)
        assert np.count_nonzero(got["status"] != ref["status"]) <= len(idx) // 8
    if dtype == np.float64:
        assert np.allclose(got["info"][0], ref["pri_res"], rtol=1e-6, atol=1e-12)
    else:   # fp32 residuals of a converged iterate are round-off
        assert np.allclose(got["info"][0], ref["pri_res"], rtol=2e-2, atol=1e-4)


@pytest.mark.gpu
def test_gpu_p5f_getlin_and_plant_match_reference_fixture():
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC
    g = golden("planar_p5f.npz")
    B = len(g["lin_u"])
    for tdt, tol in ((torch.float64, 1e-13), (torch.float32, 2e-6)):
        mpc = PlanarP5fMPC(B, tdt)
        mpc.y[0] = torch.as_tensor(g["lin_sigma"]).to(mpc.y)
        mpc.y[3] = torch.as_tensor(g["lin_phi"]).to(mpc.y)
        y0 = np.random.default_rng(3).normal(size=(7, B))
        y0[0], y0[3] = g["lin_sigma"], g["lin_phi"]
        mpc.y.copy_(torch.as_tensor(y0).to(mpc.y))
        u = torch.as_tensor(g["lin_u"]).to(mpc.y)
        lin = mpc.linearise(u).cpu().numpy().astype(np.float64)
        want = np.stack([g["lin_Ad"][:, 4, 3], g["lin_Ad"][:, 5, 3], g["lin_Bd"][:, 4], g["lin_Bd"][:, 5], g["lin_Bd"][:, 6]])
        sc = np.maximum(np.abs(want), 1e-3)
        assert np.max(np.abs(lin - want) / sc) < tol * 50, tdt
        # assembled A values = the kron construction
        st = mpc.st
        Av = mpc.Av.cpu().numpy().astype(np.float64)
        for k in (4, 9, 33):
            A = np.zeros((st["m"], st["n"]))
            for j in range(st["n"]):
                for p in range(st["A_p"][j], st["A_p"][j + 1]):
                    A[st["row_order"][st["A_i"][p]], st["var_order"][j]] = Av[p, k]      # (back to the script's labels)
            assert np.allclose(A, p5f_dense_A(10, g["lin_Ad"][k], g["lin_Bd"][k]), rtol=tol * 50, atol=tol)
        # plant tick y += (Ad y + Bd u) dt, planar/mpc_osqp_p5f.py:176
        mpc.L.umpcP5fStep(B, 0 if tdt == torch.float32 else 1, 1, 0.002, u.data_ptr(), mpc.y.data_ptr(), None, None)
        torch.cuda.synchronize()
        y1 = mpc.y.cpu().numpy().astype(np.float64)
        for k in range(B):
            yk = y0[:, k] + (g["lin_Ad"][k] @ y0[:, k] + g["lin_Bd"][k] * g["lin_u"][k]) * 0.002
            assert np.allclose(y1[:, k], yk, rtol=tol * 100, atol=tol * 10), (tdt, k)


@pytest.mark.gpu
def test_gpu_p5f_scalar_input_entry_and_wide_gather_agree_with_the_array_entry():
    """umpcP5fStepU (ONE nominal input for the batch, as the reference's loop has it: planar/mpc_osqp_p5f.py:157) against
    umpcP5fStep with a filled [B] array: bit-identical lin, A values and plant tick over three ticks of a ragged batch (the
    gather kernel's grid covers B x nnz: 300 robots leave a partial block in both directions)."""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC
    B = 300
    for tdt in (torch.float32, torch.float64):
        a, b, c = PlanarP5fMPC(B, tdt), PlanarP5fMPC(B, tdt), PlanarP5fMPC(B, tdt)
        y0 = torch.as_tensor(np.random.default_rng(5).normal(size=(7, B)) * 0.1).to(a.y)
        a.y.copy_(y0)
        b.y.copy_(y0)
        c.y.copy_(y0)
        for ti in range(2, 5):
            unom = 15.0 * np.sin(2 * np.pi * 170 * 0.002 * ti)
            a.linearise(unom)                                               # scalar entry
            b.linearise(torch.full((B,), unom, dtype=torch.float64).to(b.y))    # array entry
            assert torch.equal(a.lin, b.lin) and torch.equal(a.Av, b.Av), (tdt, ti)
            # (linearise is ONE launch, umpcP5fLinearise: the two launches it stands for give the same bits)
            c._p5f_step(0, unom, c.lin)
            c.qp.gather(c.cst, c.src, c.lin, c.Av, update=ti > 2)
            assert torch.equal(a.lin, c.lin) and torch.equal(a.Av, c.Av), (tdt, ti)
            c._p5f_step(1, unom, None)
            a._p5f_step(1, unom, None)
            b._p5f_step(1, torch.full((B,), unom, dtype=torch.float64).to(b.y), None)
            assert torch.equal(a.y, b.y), (tdt, ti)
        # every entry of the value array: constants where src < 0, cst * lin[src] elsewhere
        Av, lin = a.Av.cpu().numpy(), a.lin.cpu().numpy()
        cst, src = a.cst.cpu().numpy(), a.src.cpu().numpy()
        want = np.where(src[:, None] < 0, cst[:, None], cst[:, None] * lin[np.maximum(src, 0)])
        assert np.array_equal(Av, want.astype(Av.dtype)), tdt


@pytest.mark.gpu
def test_gpu_p5f_loop_matches_oracle():
    """Config 4 path: per tick getLin -> A -> one 50-iteration QP step -> plant; fp64 GPU vs the table oracle fed with
    the GPU's own A values (so the comparison isolates the solver), warm-started tick to tick."""
    import torch
    import osqp_table
    from robobee3d_amd.batchqp import PlanarP5fMPC
    B = 24
    mpc = PlanarP5fMPC(B, torch.float64)
    rng = np.random.default_rng(20201119)
    y0 = np.zeros((7, B))
    y0[0] = rng.uniform(-0.1, 0.1, B)
    y0[3] = rng.uniform(-0.1, 0.1, B)
    mpc.y.copy_(torch.as_tensor(y0).cuda())
    st = mpc.st
    perm = mpc.qp.s.perm
    x, y, z, E = np.zeros((87, B)), np.zeros((164, B)), np.zeros((164, B)), np.ones((164, B))
    f = lambda t: t.cpu().numpy()
    for ti in range(2, 6):
        t = 0.002 * ti
        mpc.linearise(15.0 * np.sin(2 * np.pi * 170 * t))
        Av = f(mpc.Av)
        r = osqp_table.solve(87, 164, st["A_p"], st["A_i"], st["P_cols"], perm, f(mpc.Pv), Av, f(mpc.q), f(mpc.l),
                             f(mpc.u), x, y, z, E, osqp_table.Settings(max_iter=50))
        x, y, z, E = r["x"], r["y"], r["z"], r["E"]
        mpc.tick(t)
        torch.cuda.synchronize()
        for name, mine, want in (("x", f(mpc.qp.x), x), ("y", f(mpc.qp.y), y), ("z", f(mpc.qp.z), z),
                                 ("sol_x", f(mpc.qp.sol_x), r["sol_x"])):
            assert np.allclose(mine, want, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(want).max())), (ti, name)
        assert np.array_equal(f(mpc.qp.status), r["status"])
    assert np.isfinite(f(mpc.y)).all()


@pytest.mark.gpu
def test_gpu_v1_assembly_matches_reference_and_solves():
    import torch
    import osqp_table
    from robobee3d_amd.batchqp import UprightMPCv1
    g = golden("v1_qp.npz")
    B = len(g["dt"])
    # dt is a scalar argument of the reference: group fixture rows by their own dt -> run one robot batch per row
    mpc = UprightMPCv1(B, 3, torch.float64, max_iter=400)
    st = mpc.st
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a.T)).cuda()
    for k in range(0, B, 7):
        dt = float(g["dt"][k])
        rows = np.full(B, k)
        mpc.assemble(T(g["q0"][rows]), T(g["qdes"][rows]), T(g["Qf"][rows]), T(g["Rd"][rows]), T(g["smin"][rows]),
                     T(g["smax"][rows]), dt, T(g["snom"][rows]), torch.as_tensor(g["vT0"][rows]).cuda())
        torch.cuda.synchronize()
        for name, mine in (("Pdata", mpc.Pv), ("Adata", mpc.Av), ("q", mpc.q), ("l", mpc.l), ("u", mpc.u)):
            assert np.allclose(mine.cpu().numpy()[:, 0], g[name][k], rtol=1e-14, atol=1e-15), (k, name)
    # solve: all robots take fixture row b with dt of row 0 (the solver does not care that dt is shared)
    rows = np.arange(B)
    dt = float(g["dt"][0])
    x, uu = mpc.update(T(g["q0"]), T(g["qdes"]), T(g["Qf"]), T(g["Rd"]), T(g["smin"]), T(g["smax"]), dt,
                       T(g["snom"]), torch.as_tensor(g["vT0"]).cuda())
    torch.cuda.synchronize()
    f = lambda t: t.cpu().numpy()
    z = lambda r: np.zeros((r, B))
    ref = osqp_table.solve(27, 27, st["A_p"], st["A_i"], st["P_cols"], mpc.qp.s.perm, f(mpc.Pv), f(mpc.Av), f(mpc.q),
                           f(mpc.l), f(mpc.u), z(27), z(27), z(27), np.ones((27, B)), osqp_table.Settings(max_iter=400))
    assert np.allclose(f(x), ref["sol_x"], rtol=1e-8, atol=1e-9, equal_nan=True)
    assert np.array_equal(f(mpc.qp.status), ref["status"])
    assert np.allclose(f(uu), ref["sol_x"][18:21], rtol=1e-8, atol=1e-9, equal_nan=True)
    assert np.allclose(f(mpc.vT0), g["vT0"] + ref["sol_x"][18], rtol=1e-8, atol=1e-9, equal_nan=True)


# ------------------------------------------------------------------------------------------------------
# UprightMPC2 at a general horizon N (template/template_controllers.py:170-258)
# ------------------------------------------------------------------------------------------------------
def test_general_horizon_structure_is_initConstraint():
    from robobee3d_amd import symbolic
    from robobee3d_amd.batchqp import uprightmpc2_structure
    g = golden("assembly_fp64_N5.npz")
    st = uprightmpc2_structure(5)
    assert st["A_p"] == g["A_indptr"].tolist() and st["A_i"] == g["A_indices"].tolist()
    assert symbolic.ax_idx(5) == g["Axidx"][0].tolist()
    assert (st["n"], st["m"]) == (75, 65)


def _state_ref_from_seq(seq, idx, dtype):
    st = np.zeros((18, len(idx)), dtype)
    st[0:3] = seq["p0"][idx].T
    st[3:12] = seq["R0"][idx].transpose(2, 1, 0).reshape(9, -1)     # column-major
    st[12:18] = seq["dq0"][idx].T
    # (vstack of transposed views comes back Fortran-ordered: force the [rows][B] layout)
    ref = np.ascontiguousarray(np.vstack((seq["pdes"][idx].T, seq["dpdes"][idx].T, seq["sdes"][idx].T)), dtype)
    T0 = np.where(seq["actualT0"][idx] >= 0, seq["actualT0"][idx], seq["pre_T0"][idx]).astype(dtype)
    return st, ref, T0


@pytest.mark.gpu
def test_gpu_general_horizon_assembly_and_step():
    import torch
    import osqp_table
    from robobee3d_amd.batchqp import UprightMPC2N
    g = golden("assembly_fp64_N5.npz")
    seq = golden("seq_iter50.npz")
    idx = np.arange(int(g["n"]))
    st, ref, T0 = _state_ref_from_seq(seq, idx, np.float64)
    mpc = UprightMPC2N(len(idx), 5, dtype=torch.float64)
    mpc.T0.copy_(torch.as_tensor(T0).cuda())
    S, R = torch.as_tensor(st).cuda(), torch.as_tensor(ref).cuda()
    mpc.assemble(S, R)
    torch.cuda.synchronize()
    f = lambda t: t.cpu().numpy()
    for name, mine in (("l", mpc.l), ("u", mpc.u), ("q", mpc.q), ("Px", mpc.Pv), ("Adata", mpc.Av)):
        assert np.allclose(f(mine).T, g[name], rtol=1e-13, atol=1e-15), name
    out = f(mpc.update(S, R)).copy()
    z = lambda r: np.zeros((r, len(idx)))
    s = mpc.st
    r = osqp_table.solve(75, 65, s["A_p"], s["A_i"], s["P_cols"], mpc.qp.s.perm, g["Px"].T, g["Adata"].T, g["q"].T,
                         g["l"].T, g["u"].T, z(75), z(65), z(65), np.ones((65, len(idx))), osqp_table.Settings(max_iter=50))
    assert np.allclose(f(mpc.qp.sol_x), r["sol_x"], rtol=1e-8, atol=1e-10, equal_nan=True)
    ok = r["status"] > 0
    assert np.allclose(out[0][ok], (T0 + r["sol_x"][60])[ok], rtol=1e-8, atol=1e-12)
    assert np.allclose(out[1:3][:, ok], r["sol_x"][61:63][:, ok], rtol=1e-8, atol=1e-10)
    # getAccDes (template_controllers.py:244-250) restated
    e3h = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 0.0]])
    for k in np.nonzero(ok)[0][:16]:
        dy1 = r["sol_x"][30:36, k]
        R0 = seq["R0"][k].astype(np.float64)
        dq1 = np.hstack((dy1[:3], e3h @ R0.T @ dy1[3:6]))
        assert np.allclose(out[3:, k], (dq1 - seq["dq0"][k]) / 5.0, rtol=1e-8, atol=1e-11)


@pytest.mark.gpu
def test_gpu_general_path_at_N3_agrees_with_the_specialised_kernel():
    """The same controller step through both product paths (table-driven kernel vs the generated straight-line +
    assembly kernel), fp32, own orderings: outputs agree within the fp32 parity band of test_gpu_parity.py."""
    import torch
    from robobee3d_amd.batch import BatchUprightMPC
    from robobee3d_amd.batchqp import UprightMPC2N
    seq = golden("seq_iter50.npz")
    idx = np.arange(128)
    st, ref, T0 = _state_ref_from_seq(seq, idx, np.float32)
    S, R = torch.as_tensor(st).cuda(), torch.as_tensor(ref).cuda()
    gen = UprightMPC2N(len(idx), 3, dtype=torch.float32)
    gen.T0.copy_(torch.as_tensor(T0).cuda())
    o1 = gen.update(S, R).cpu().numpy()
    spec = BatchUprightMPC(len(idx), torch.float32)
    spec.set_state(st, ref)
    spec.ctrl[123] = torch.as_tensor(T0).cuda()
    spec.update()
    o2 = spec.out.cpu().numpy()
    assert np.all(np.abs(o1[0] - o2[0]) <= 3e-5)
    assert np.all(np.abs(o1[1:3] - o2[1:3]) <= np.maximum(2e-2, 1e-3 * np.abs(o2[1:3])))
    assert np.all(np.abs(o1[3:] - o2[3:]) <= 3e-5)


@pytest.mark.gpu
def test_gpu_specialised_kernels_equal_the_table_driven_kernel():
    """codegen_qp.py's straight-line kernels carry out the table-driven kernel's operations in the same order:
    fp64 results are bit-identical on all three built-in structures, fp32 ones too."""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC, UprightMPC2N, UprightMPCv1
    seq = golden("seq_iter50.npz")
    for tdt, ndt in ((torch.float64, np.float64), (torch.float32, np.float32)):
        B = 96
        # p5f
        mpc = PlanarP5fMPC(B, tdt)
        # PlanarP5fMPC's choice: fp32 the lane specialisation with its assembly loop, fp64 the wave kernel; compared here
        # are the two lane-per-robot C++ kernels
        assert mpc.qp.kernel_name == ("p5f10+asm" if tdt == torch.float32 else "wave")
        mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
        mpc.y[3] = torch.linspace(0.1, -0.1, B).to(mpc.y)
        mpc.linearise(7.0)
        res = []
        for tables in (False, True):
            mpc.qp.reset()
            mpc.qp.set_kernel("tables" if tables else "lane_cpp")
            assert mpc.qp.kernel_name == ("tables" if tables else "p5f10")
            mpc.qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u)
            torch.cuda.synchronize()
            res.append([t.cpu().numpy().copy() for t in (mpc.qp.x, mpc.qp.y, mpc.qp.z, mpc.qp.sol_x, mpc.qp.status, mpc.qp.info)])
        for a, b in zip(*res):
            assert np.array_equal(a, b, equal_nan=True)
        # UprightMPC2 N = 5
        idx = np.arange(B)
        st, ref, T0 = _state_ref_from_seq(seq, idx, ndt)
        S, R = torch.as_tensor(st).cuda(), torch.as_tensor(ref).cuda()
        res = []
        for tables in (False, True):
            m5 = UprightMPC2N(B, 5, dtype=tdt)
            m5.qp.use_tables(tables)
            assert m5.qp.kernel_name == ("tables" if tables else "umpc2n5")
            m5.T0.copy_(torch.as_tensor(T0).cuda())
            res.append(m5.update(S, R).cpu().numpy().copy())
        assert np.array_equal(res[0], res[1], equal_nan=True)
        # v1
        g = golden("v1_qp.npz")
        T = lambda a: torch.as_tensor(np.ascontiguousarray(a.T, ndt)).cuda()
        res = []
        for tables in (False, True):
            v1 = UprightMPCv1(len(g["dt"]), 3, tdt, max_iter=100)
            v1.qp.use_tables(tables)
            assert v1.qp.kernel_name == ("tables" if tables else "v1n3")
            x, uu = v1.update(T(g["q0"]), T(g["qdes"]), T(g["Qf"]), T(g["Rd"]), T(g["smin"]), T(g["smax"]), 2.5,
                              T(g["snom"]), torch.as_tensor(g["vT0"].astype(ndt)).cuda())
            res.append(x.cpu().numpy().copy())
        assert np.array_equal(res[0], res[1], equal_nan=True)


@pytest.mark.gpu
def test_gpu_p5f_full_size_properties():
    """Config 4 at its full batch (B = 16 384, fp32): every robot solved and finite after three ticks; robots that
    start from the same state produce the same bits wherever they sit in the batch (lane / block independence)."""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC
    B = 16384
    mpc = PlanarP5fMPC(B, torch.float32)
    assert mpc.qp.kernel_name == "p5f10+asm"          # (the build-time specialisation is found by the hash of the structure's tables)
    rng = np.random.default_rng(20201119)
    pert = rng.uniform(-0.1, 0.1, (2, B // 2)).astype(np.float32)
    pert = np.concatenate((pert, pert[:, ::-1]), 1)                  # robot b and robot B-1-b are twins
    mpc.y[0] = torch.as_tensor(pert[0]).cuda()
    mpc.y[3] = torch.as_tensor(pert[1]).cuda()
    for ti in range(2, 5):
        mpc.tick(0.002 * ti)
    torch.cuda.synchronize()
    y = mpc.y.cpu().numpy()
    x = mpc.qp.sol_x.cpu().numpy()
    assert np.isfinite(y).all() and np.isfinite(x).all()
    assert (mpc.qp.status.cpu().numpy() > 0).all()
    assert np.array_equal(y, y[:, ::-1]) and np.array_equal(x, x[:, ::-1])
    # the dynamics equalities hold at the solution to the ADMM tolerance: A_eq x = 0 (l = u = 0 on those rows)
    st = mpc.st
    Av = mpc.Av.cpu().numpy().astype(np.float64)[:, :64]
    Ax = np.zeros((st["m"], 64))
    for j in range(st["n"]):
        for p in range(st["A_p"][j], st["A_p"][j + 1]):
            Ax[st["A_i"][p]] += Av[p] * x[j, :64]
    assert np.abs(Ax[:77]).max() < 5e-3 * max(1.0, np.abs(x[:, :64]).max())


@pytest.mark.gpu
def test_gpu_table_kernel_on_a_large_structure_matches_oracle():
    """UprightMPC2 at N = 10 (n = 150, m = 130, nnz(L) ~ 1.5e3) has no build-time specialisation: the table-driven
    kernel, fp64, against the table oracle."""
    import torch
    import osqp_table
    from robobee3d_amd.batchqp import UprightMPC2N
    seq = golden("seq_iter50.npz")
    idx = np.arange(12)
    st, ref, T0 = _state_ref_from_seq(seq, idx, np.float64)
    mpc = UprightMPC2N(len(idx), 10, dtype=torch.float64)
    assert mpc.qp.kernel_name == "tables"
    mpc.T0.copy_(torch.as_tensor(T0).cuda())
    mpc.update(torch.as_tensor(st).cuda(), torch.as_tensor(ref).cuda())
    torch.cuda.synchronize()
    f = lambda t: t.cpu().numpy()
    s = mpc.st
    z = lambda r: np.zeros((r, len(idx)))
    r = osqp_table.solve(s["n"], s["m"], s["A_p"], s["A_i"], s["P_cols"], mpc.qp.s.perm, f(mpc.Pv), f(mpc.Av), f(mpc.q),
                         f(mpc.l), f(mpc.u), z(s["n"]), z(s["m"]), z(s["m"]), np.ones((s["m"], len(idx))),
                         osqp_table.Settings(max_iter=50))
    assert np.allclose(f(mpc.qp.x), r["x"], rtol=1e-8, atol=1e-10)
    assert np.allclose(f(mpc.qp.sol_x), r["sol_x"], rtol=1e-8, atol=1e-10, equal_nan=True)
    assert np.array_equal(f(mpc.qp.status), r["status"])


@pytest.mark.gpu
def test_gpu_createMPC_pair_cross_check():
    """createMPC() returns (pyver, cver) like template_controllers.py:260-280, and the two agree on the reference's
    own cross-check inputs (template_controllers.py:303-320: identity attitude, random-ish state, 6-argument update)."""
    from robobee3d_amd.uprightmpc2py import createMPC, UprightMPC2
    up, upc = createMPC()
    p = np.array([0.0, 0, 0])
    R0 = np.eye(3)
    dq = np.array([0.1, 0, 0, 0, 0, 0.0])
    pdes, dpdes, sdes = np.array([0.0, 0, 10]), np.array([0.0, 0, 0.05]), np.array([0.0, 0, 1])
    # the twin with the C path's semantics (exactly 50 iterations) tracks the compiled-C twin call by call
    up50 = UprightMPC2(3, 5, 9.81e-3, 2, 1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2, np.array([3333.0, 3333.0, 1000.0]), maxIter=50)
    for _ in range(3):
        u1, a1 = up50.update(p, R0, dq, pdes, dpdes, sdes, -1.0)
        u2, a2 = upc.update(p, R0, dq, pdes, dpdes, sdes)
        assert abs(u1[0] - u2[0]) <= 3e-5 and np.all(np.abs(u1[1:] - u2[1:]) <= np.maximum(2e-2, 1e-3 * np.abs(u2[1:])))
        assert np.all(np.abs(a1 - a2) <= 3e-5)
    # the default twin has the reference's semantics (template_controllers.py:190-191,216-219): solve() runs to the
    # termination criteria at eps 1e-4, checked every 25 iterations
    for _ in range(3):
        u1, a1 = up.update(p, R0, dq, pdes, dpdes, sdes, -1.0)
        assert up.status_val == 1 and up.iterations % 25 == 0 and 25 <= up.iterations < 4000
        assert np.isfinite(u1).all() and np.isfinite(a1).all()
    up5, _ = createMPC(N=5)
    u5, a5 = up5.update(p, R0, dq, pdes, dpdes, sdes)
    assert np.isfinite(u5).all() and np.isfinite(a5).all() and up5.prevsol.shape == (75,) and up5.status_val == 1 and up5.iterations < 4000


@pytest.mark.gpu
def test_gpu_p5f_assembly_loop_agrees_with_the_table_kernel(margin):
    """fp32 planar p5f: the specialisation whose middle ADMM iterations run as generated assembly (asmqp.py: leaf rows
    folded, fused multiply-adds, equality-row shortcut) against the table-driven kernel, cold and warm-started calls,
    a full-wave batch, ragged ones, one of five robots; other Ruiz pass counts (the block's loop; no block)."""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC
    for B, scaling in ((200, 10), (64, 10), (5, 10), (130, 3), (130, 0)):     # (other pass counts: the block's loop / no block)
        mpc = PlanarP5fMPC(B, torch.float32, scaling=scaling)
        mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
        mpc.y[3] = torch.linspace(0.1, -0.1, B).to(mpc.y)
        res = []
        for mode in ("lane", "tables"):
            mpc.qp.reset()
            mpc.qp.set_kernel(mode)
            assert mpc.qp.kernel_name == {"lane": "p5f10+asm", "tables": "tables"}[mode]
            for ti in (2, 3, 4):      # later calls are warm-started and classify with the previous call's E
                mpc.linearise(15.0 * np.sin(2 * np.pi * 170 * 0.002 * ti))
                mpc.qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u)
            torch.cuda.synchronize()
            res.append([t.cpu().numpy().astype(np.float64).copy() for t in (mpc.qp.x, mpc.qp.y, mpc.qp.z, mpc.qp.sol_x, mpc.qp.Eprev)]
                       + [mpc.qp.status.cpu().numpy().copy(), mpc.qp.info.cpu().numpy().astype(np.float64).copy(),
                          mpc.qp.sol_y.cpu().numpy().astype(np.float64).copy()])
        worst = max(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) for a, b in zip(res[0][:5], res[1][:5]))
        margin("B=%d, %d Ruiz passes: iterates / solution, |d| / max(1, |ref|)" % (B, scaling), worst, 3e-6)
        # what the residual block writes: the scaled multipliers, the residuals and the bookkeeping rows of info
        margin("B=%d, %d Ruiz passes: sol_y, |d| / max(1, |ref|)" % (B, scaling),
               np.max(np.abs(res[0][7] - res[1][7]) / np.maximum(1.0, np.abs(res[1][7]))), 1e-5)
        ia, ib = res[0][6], res[1][6]
        # (at convergence the residuals are differences of nearly equal numbers: two fp32 evaluations agree in magnitude only)
        ratio = np.maximum(ia[:2], 1e-12) / np.maximum(ib[:2], 1e-12)
        margin("B=%d, %d Ruiz passes: pri_res, dua_res, median ratio (max of r, 1/r)" % (B, scaling),
               np.median(np.maximum(ratio, 1.0 / ratio)), 3.0)
        margin("B=%d, %d Ruiz passes: pri_res, dua_res, |d|" % (B, scaling), np.max(np.abs(ia[:2] - ib[:2])), 1e-5)
        assert np.array_equal(ia[3:], ib[3:]) and np.max(np.abs(ia[2] - ib[2]) / ib[2]) <= 1e-5
        assert np.count_nonzero(res[0][5] != res[1][5]) <= B // 8
        assert np.all(np.isfinite(res[0][0]))


@pytest.mark.gpu
def test_gpu_p5f_all_assembly_route_hands_unsolved_waves_to_the_cpp_residual_phase(margin):
    """A warm-started tick takes the all-assembly route (Ruiz, glue, LDL' + loop, residual block). When a robot of the wave
    is not SOLVED at the strict tolerances the residual block does not settle the wave: the C++ residual phase (approximate
    tolerances, certificates, status, solution rows) reloads the equilibrated data from the wave's streams and the iterates
    from LDS. Forced here with few iterations after a large change of the linearisation; robots 0..63 keep the old
    linearisation and stay solved where they can. Against the table kernel, incl. status and the info rows."""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC
    B = 200
    mpc = PlanarP5fMPC(B, torch.float32)
    mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
    mpc.y[3] = torch.linspace(0.1, -0.1, B).to(mpc.y)
    for iters in (3, 7):
        res = []
        for mode in ("lane", "tables"):
            mpc.qp.reset()
            mpc.qp.set_kernel(mode)
            for ti in (2, 3):
                mpc.linearise(15.0 * np.sin(2 * np.pi * 170 * 0.002 * ti))
                mpc.qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u, max_iter=50)
            mpc.linearise(-15.0)
            q = mpc.q.clone()
            q[:, 64:] *= 3.0                      # (same bounds: z of the equality rows still equals them)
            mpc.qp.solve(mpc.Pv, mpc.Av, q, mpc.l, mpc.u, max_iter=iters)
            torch.cuda.synchronize()
            res.append([t.cpu().numpy().astype(np.float64).copy() for t in (mpc.qp.x, mpc.qp.y, mpc.qp.z, mpc.qp.Eprev)]
                       + [mpc.qp.status.cpu().numpy().copy(), mpc.qp.info.cpu().numpy().astype(np.float64).copy(),
                          mpc.qp.sol_x.cpu().numpy().astype(np.float64).copy(), mpc.qp.sol_y.cpu().numpy().astype(np.float64).copy()])
        st_a, st_b = res[0][4], res[1][4]
        assert np.count_nonzero(st_b != 1) >= B // 2, "the case must leave robots unsolved"
        worst = max(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) for a, b in zip(res[0][:4], res[1][:4]))
        margin("%d iterations: iterates, |d| / max(1, |ref|)" % iters, worst, 1e-5)
        margin("%d iterations: robots whose status differs" % iters, float(np.count_nonzero(st_a != st_b)), 2.0)
        ia, ib = res[0][5], res[1][5]
        margin("%d iterations: pri_res, dua_res, relative" % iters,
               np.max(np.abs(ia[:2] - ib[:2]) / np.maximum(1e-6, np.abs(ib[:2]))), 1e-4)
        assert np.array_equal(ia[3:], ib[3:])
        same = st_a == st_b
        for a, b in zip(res[0][6:], res[1][6:]):      # solution rows (NaN for robots flagged infeasible)
            assert np.array_equal(np.isnan(a[:, same]), np.isnan(b[:, same]))
            ok = same & ~np.isnan(b).any(0)
            assert np.max(np.abs(a[:, ok] - b[:, ok]) / np.maximum(1.0, np.abs(b[:, ok]))) <= 1e-5
    mpc.qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u, max_iter=50)       # (leave the default iteration count behind)


@pytest.mark.gpu
def test_gpu_p5f_all_assembly_route_with_null_output_rows():
    """umpcQPSolve allows NULL for sol_x, sol_y, status and info (include/umpc_mi355x.h). The residual block of the
    all-assembly route writes all of them, so a call without them must leave it out and finish in the C++ residual phase:
    same iterates and E as the call with outputs, bit for bit up to the route's own rounding (none: same blocks)."""
    import ctypes as C
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC, _ptr
    B = 130
    mpc = PlanarP5fMPC(B, torch.float32)
    mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
    qp = mpc.qp
    assert qp.kernel_name == "p5f10+asm"
    res = []
    for with_outputs in (True, False):
        qp.reset()
        for ti in (2, 3, 4):
            mpc.linearise(15.0 * np.sin(2 * np.pi * 170 * 0.002 * ti))
            if with_outputs or ti < 4:
                qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u)
            else:
                null = C.c_void_p(None)
                stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                rc = qp.L.umpcQPSolve(qp.h, _ptr(mpc.Pv), _ptr(mpc.Av), _ptr(mpc.q), _ptr(mpc.l), _ptr(mpc.u), _ptr(qp.x),
                                      _ptr(qp.y), _ptr(qp.z), _ptr(qp.Eprev), null, null, null, null, stream)
                assert rc == 0
        torch.cuda.synchronize()
        res.append([t.cpu().numpy().copy() for t in (qp.x, qp.y, qp.z, qp.Eprev)])
    for a, b in zip(*res):
        assert np.all(np.isfinite(a)) and np.array_equal(a, b)


@pytest.mark.gpu
def test_gpu_p5f_assembly_kernel_falls_back_when_a_dynamics_row_is_not_an_equality(margin):
    """The assembly loop takes the 77 dynamics rows for equalities; the kernel checks it per wave and runs the C++ loop where
    it does not hold. Robots 64..127 (one whole wave) and robot 150 (one lane of the third wave) get l < u on a dynamics
    row: their waves must take the fallback, the other waves the assembly loop -- every robot agrees with the table
    kernel either way (the Ruiz block and the batched loader run on both paths)."""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC
    B = 256
    mpc = PlanarP5fMPC(B, torch.float32)
    assert mpc.qp.kernel_name == "p5f10+asm"
    mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
    mpc.y[3] = torch.linspace(0.1, -0.1, B).to(mpc.y)
    mpc.u[5, 64:128] += 0.25          # row 5 of the dynamics block: now an inequality for these robots
    mpc.u[40, 150] += 0.5
    res = []
    for mode in ("lane", "tables"):
        mpc.qp.reset()
        mpc.qp.set_kernel(mode)
        for ti in (2, 3):
            mpc.linearise(15.0 * np.sin(2 * np.pi * 170 * 0.002 * ti))
            mpc.qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u)
        torch.cuda.synchronize()
        res.append([t.cpu().numpy().astype(np.float64).copy() for t in (mpc.qp.x, mpc.qp.y, mpc.qp.z, mpc.qp.sol_x)])
    d = np.max(np.stack([(np.abs(a - b) / np.maximum(1.0, np.abs(b))).max(0) for a, b in zip(*res)]), axis=0)   # per robot
    fallback = np.zeros(B, bool)
    fallback[64:192] = True           # waves 1 and 2
    margin("robots on the assembly path, |d| / max(1, |ref|)", d[~fallback].max(), 3e-6)
    margin("robots on the fallback path, |d| / max(1, |ref|)", d[fallback].max(), 3e-6)
    assert np.all(np.isfinite(res[0][0]))


@pytest.mark.gpu
def test_gpu_wave_kernel_agrees_with_lane_kernels():
    """One wavefront per robot (LDS-resident, level-scheduled; the default) against the lane-per-robot table kernel on
    three structures: the same iterates to rounding (the factorisation and the Ruiz cost sums associate differently)."""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC, UprightMPC2N
    seq = golden("seq_iter50.npz")
    for tdt, ndt, tol in ((torch.float64, np.float64, 1e-10), (torch.float32, np.float32, 2e-3)):
        B = 70
        mpc = PlanarP5fMPC(B, tdt)
        mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
        mpc.y[3] = torch.linspace(0.1, -0.1, B).to(mpc.y)
        res = []
        for mode in ("wave", "tables"):
            mpc.qp.reset()
            mpc.qp.set_kernel(mode)
            assert mpc.qp.kernel_name == mode
            for ti in (2, 3):      # second call is warm-started and classifies with the first call's E
                mpc.linearise(15.0 * np.sin(2 * np.pi * 170 * 0.002 * ti))
                mpc.qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u)
            torch.cuda.synchronize()
            res.append([t.cpu().numpy().astype(np.float64).copy() for t in (mpc.qp.x, mpc.qp.y, mpc.qp.z, mpc.qp.sol_x, mpc.qp.Eprev)]
                       + [mpc.qp.status.cpu().numpy().copy()])
        for a, b in zip(res[0][:5], res[1][:5]):
            assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < tol, tdt
        assert np.count_nonzero(res[0][5] != res[1][5]) <= (0 if tdt == torch.float64 else B // 8)
        for N in (3, 10):
            idx = np.arange(40)
            st, ref, T0 = _state_ref_from_seq(seq, idx, ndt)
            S, R = torch.as_tensor(st).cuda(), torch.as_tensor(ref).cuda()
            outs = []
            for mode in ("wave", "tables"):
                m = UprightMPC2N(len(idx), N, dtype=tdt)
                m.qp.set_kernel(mode)
                m.T0.copy_(torch.as_tensor(T0).cuda())
                outs.append(m.update(S, R).cpu().numpy().astype(np.float64).copy())
            if tdt == torch.float64:
                assert np.allclose(outs[0], outs[1], rtol=1e-8, atol=1e-10, equal_nan=True), N
            else:   # two fp32 evaluations, each inside the fp32 parity band of the fp64 value: twice the band
                assert np.all(np.abs(outs[0][0] - outs[1][0]) <= 6e-5)
                assert np.all(np.abs(outs[0][1:3] - outs[1][1:3]) <= 2 * np.maximum(2e-2, 1e-3 * np.abs(outs[1][1:3])))
                assert np.all(np.abs(outs[0][3:] - outs[1][3:]) <= 6e-5)


@pytest.mark.gpu
def test_gpu_eps_terminated_mode_matches_oracle(structure):
    """check_termination = 25, max_iter = 4000 (pip-osqp defaults, the reference twin's solve()): per-robot early
    exit. fp64: same statuses, same iteration counts, same iterates as the oracle run to the same criterion; fp32:
    counts may differ by one check interval at a tolerance boundary."""
    import osqp_table
    from conftest import record_margin
    g = golden("seq_iter50.npz")
    idx = np.arange(96)
    perm = structure["perm"]
    A_p, A_i, Pv, Av, q, l, u = raw_uprightmpc2_qp(g, idx, np.float64)
    Eprev = np.ones((39, len(idx)))
    z = lambda r: np.zeros((r, len(idx)))
    args = (Pv, Av, q, l, u, z(45), z(39), z(39), Eprev)
    for dtype in (np.float64, np.float32):
        import torch
        from robobee3d_amd.batchqp import BatchQP
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        ref = osqp_table.solve(45, 39, A_p, A_i, list(range(45)), perm, *args,
                               osqp_table.Settings(max_iter=4000, check_termination=25), dtype=dtype)
        qp = BatchQP(45, 39, A_p, A_i, list(range(45)), len(idx), tdt, perm=perm, max_iter=4000, check_termination=25)
        assert qp.kernel_name == "tables"
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype)).cuda()
        qp.solve(dev(Pv), dev(Av), dev(q), dev(l), dev(u))
        it = qp.info[4].cpu().numpy().astype(int)
        status = qp.status.cpu().numpy()
        assert np.all(np.isin(status, (1, 2, -2))) and np.all(it % 25 == 0) and len(set(it)) > 2   # fixed rho: slow tails
        assert np.all((status == 1) == (it < 4000))
        if dtype == np.float64:
            assert np.array_equal(it, ref["iters"]) and np.array_equal(status, ref["status"])
            np.testing.assert_allclose(qp.x.cpu().numpy(), ref["x"], rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(qp.sol_x.cpu().numpy(), ref["sol_x"], rtol=1e-9, atol=1e-11)
        else:
            # fp32: WHEN a robot crosses the eps boundary is round-off (fixed rho: the residual creeps down over
            # hundreds of iterations), so counts are compared statistically and solutions where both solved
            ndiff = int(np.count_nonzero(it != ref["iters"]))
            record_margin("eps-terminated mode fp32 (fixed rho)", "robots whose iteration count differs (of 96)", ndiff, 96)
            both = (status == 1) & (ref["status"] == 1)
            assert both.sum() >= 60
            sx = np.maximum(1.0, np.abs(ref["sol_x"][:, both]))
            assert (np.abs(qp.sol_x.cpu().numpy()[:, both] - ref["sol_x"][:, both]) / sx).max() < 5e-2   # both eps-optimal
        # adaptive rho (interval 25): every robot reaches "solved" within a few hundred iterations, as in the oracle
        refa = osqp_table.solve(45, 39, A_p, A_i, list(range(45)), perm, *args,
                                osqp_table.Settings(max_iter=4000, check_termination=25, adaptive_rho_interval=25), dtype=dtype)
        qp.set_termination(25, 4000, adaptive_rho_interval=25)
        qp.reset()
        qp.solve(dev(Pv), dev(Av), dev(q), dev(l), dev(u))
        ita, upa = qp.info[4].cpu().numpy().astype(int), qp.info[5].cpu().numpy().astype(int)
        # (fp32: the dual residual of a converged iterate is round-off near eps, so the last robots take longer)
        assert np.all(qp.status.cpu().numpy() == 1) and ita.max() <= (400 if dtype == np.float64 else 3999) and upa.max() >= 1
        if dtype == np.float64:
            assert np.array_equal(ita, refa["iters"]) and np.array_equal(upa, refa["rho_updates"])
            np.testing.assert_allclose(qp.sol_x.cpu().numpy(), refa["sol_x"], rtol=1e-8, atol=1e-10)
        else:
            record_margin("eps-terminated mode fp32 (adaptive rho)", "robots whose iteration count differs (of 96)",
                          int(np.count_nonzero(ita != refa["iters"])), 96)
        # set_termination(0) restores the fixed count
        qp.set_termination(0, 50)
        qp.reset()
        qp.solve(dev(Pv), dev(Av), dev(q), dev(l), dev(u))
        assert np.all(qp.info[4].cpu().numpy() == 50)


@pytest.mark.gpu
def test_gpu_infeasibility_certificates_nan_and_cold_start():
    """The `!has_solution` path of the general solver on synthetic QPs (auxil.c:362-512, 539-564): primal / dual
    infeasibility detected, NaN solution, iterates cold-started -- same status and same iteration count as the oracle."""
    import torch
    import osqp_table
    from robobee3d_amd.batchqp import BatchQP
    for name, n, m, Ap, Ai, Pc, Pv, Av, q, l, u, want in _tiny_infeasible_qps():
        for dtype, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
            B = 70                      # one full wave + a ragged one; robot 3 gets a FEASIBLE variant of the data
            c = lambda a: np.repeat(np.array(a, dtype)[:, None], B, 1)
            lv, uv, qv = c(l), c(u), c(q)
            if want == -3:
                lv[0, 3] = -1.0         # x0 >= -1 and x0 <= 0: feasible
            else:
                uv[1, 3] = 5.0          # x1 <= 5: bounded
            ref = osqp_table.solve(n, m, Ap, Ai, Pc, list(range(n + m)), c(Pv), c(Av), qv, lv, uv, np.zeros((n, B)),
                                   np.zeros((m, B)), np.zeros((m, B)), np.ones((m, B)),
                                   osqp_table.Settings(max_iter=4000, check_termination=25), dtype=dtype)
            qp = BatchQP(n, m, Ap, Ai, Pc, B, tdt, perm=list(range(n + m)), max_iter=4000, check_termination=25)
            dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype)).cuda()
            qp.solve(dev(c(Pv)), dev(c(Av)), dev(qv), dev(lv), dev(uv))
            st = qp.status.cpu().numpy()
            assert np.array_equal(st, ref["status"]), (name, dtype, st[:5], ref["status"][:5])
            assert st[0] == want and st[3] == 1
            assert np.array_equal(qp.info[4].cpu().numpy().astype(int), ref["iters"])
            sol = qp.sol_x.cpu().numpy()
            assert np.isnan(sol[:, 0]).all() and np.isfinite(sol[:, 3]).all()
            assert not qp.x[:, 0].any().item() and not qp.y[:, 0].any().item() and not qp.z[:, 0].any().item()
            np.testing.assert_allclose(sol[:, 3], ref["sol_x"][:, 3], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_gpu_p5f_loose_loop_y0_variant_and_its_fallback_agree(margin):
    """The loose loop of the p5f assembly route has two bodies (asmqp.program): the y0 one when the warm start has y == 0
    on every inequality row of the wave (cold start, or any earlier result of the loop: a loose row's multiplier stays
    exactly 0), the general loose one otherwise. Same ticks from (a) the cold start and (b) the cold start with
    multipliers of 1e-30 planted on the inequality rows of every robot: (b) runs the other body on every wave, and a
    multiplier of 1e-30 moves nothing a float can see (y / rho = 1e-24 against z ~ 1): the iterates must agree to the
    last bits. A ragged batch; waves of (a) keep y == 0 on those rows through warm-started ticks."""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC
    B = 200
    mpc = PlanarP5fMPC(B, torch.float32)
    assert mpc.qp.kernel_name == "p5f10+asm"
    mpc.y[0] = torch.linspace(-0.1, 0.1, B).to(mpc.y)
    mpc.y[3] = torch.linspace(0.1, -0.1, B).to(mpc.y)
    n, m = mpc.qp.n, mpc.qp.m
    ineq = slice(m - n, m)                      # planar/mpc_osqp_p5f.py:120-128: the identity block below the dynamics rows
    res = []
    for plant in (False, True):
        mpc.qp.reset()
        if plant:
            mpc.qp.y[ineq] = 1e-30
        ys = []
        for ti in (2, 3, 4):
            mpc.linearise(15.0 * np.sin(2 * np.pi * 170 * 0.002 * ti))
            mpc.qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u)
            ys.append(mpc.qp.y[ineq].abs().max().item())
        torch.cuda.synchronize()
        res.append([t.cpu().numpy().astype(np.float64).copy() for t in (mpc.qp.x, mpc.qp.y, mpc.qp.z, mpc.qp.sol_x, mpc.qp.sol_y)]
                   + [mpc.qp.status.cpu().numpy().copy()])
        if not plant:
            assert ys == [0.0, 0.0, 0.0]          # exactly zero after every tick: the y0 body ran, and keeps running
        else:
            assert 0.0 < ys[0] < 1e-25            # the planted multipliers decay like round-off, never to an exact zero
    worst = max(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) for a, b in zip(res[0][:5], res[1][:5]))
    margin("y0 body vs general loose body, 3 ticks, B = 200: iterates / solution, |d| / max(1, |ref|)", worst, 1e-12)
    assert np.array_equal(res[0][5], res[1][5]) and np.all(res[0][5] == 1)


@pytest.mark.gpu
def test_gpu_p5f_relabelled_problem_is_the_scripts_problem():
    """PlanarP5fMPC solves p5f_structure(grouped=True) -- the script's QP with variables and rows relabelled so that its connected
    components are contiguous. The same ticks on the script's own labelling (a BatchQP on p5f_structure(), data un-permuted)
    must give the same solution: fp64, both through the general kernels, to 1e-9; solution() undoes the relabelling."""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC, BatchQP, p5f_structure
    B = 70
    mpc = PlanarP5fMPC(B, torch.float64)
    sc = p5f_structure(10)
    st = mpc.st
    vo, ro = torch.as_tensor(st["var_order"]).cuda(), torch.as_tensor(st["row_order"]).cuda()
    col = lambda v: torch.as_tensor(np.repeat(np.asarray(v, np.float64)[:, None], B, 1)).cuda().contiguous()
    Pv, q, l, u = col(sc["Pv"]), col(sc["q"]), col(sc["l"]), col(sc["u"])
    cst, src = torch.as_tensor(sc["cst"]).cuda(), torch.as_tensor(sc["src"]).cuda()
    Av = torch.zeros((len(sc["A_i"]), B), dtype=torch.float64, device="cuda")
    mpc.y.copy_(torch.as_tensor(np.random.default_rng(8).normal(size=(7, B)) * 0.05).to(mpc.y))
    for ti in range(2, 6):
        mpc.tick(0.002 * ti)
    # the same four warm-started solves, step by step, on both labellings
    qp2 = BatchQP(sc["n"], sc["m"], sc["A_p"], sc["A_i"], sc["P_cols"], B, torch.float64)
    qp2.set_kernel("wave")
    mpc2 = PlanarP5fMPC(B, torch.float64)
    mpc2.y.copy_(torch.as_tensor(np.random.default_rng(8).normal(size=(7, B)) * 0.05).to(mpc2.y))
    for ti in range(2, 6):
        unom = 15.0 * np.sin(2 * np.pi * 170 * 0.002 * ti)
        mpc2.linearise(unom)
        qp2.gather(cst, src, mpc2.lin, Av)
        sx, sy, status = qp2.solve(Pv, Av, q, l, u)
        mpc2.qp.solve(mpc2.Pv, mpc2.Av, mpc2.q, mpc2.l, mpc2.u)
        mpc2._p5f_step(1, unom, None)
    got = mpc2.solution()
    assert torch.equal(got[vo], mpc2.qp.sol_x)
    scale = max(1.0, float(sx.abs().max()))
    assert float((got - sx).abs().max()) <= 1e-9 * scale
    assert float((mpc2.qp.sol_y - sy[ro]).abs().max()) <= 1e-9 * max(1.0, float(sy.abs().max()))
    assert float((mpc2.qp.sol_x - mpc.qp.sol_x).abs().max()) <= 1e-12 * scale     # (and the tick() path is that sequence)


@pytest.mark.gpu
def test_gpu_p5f_entries_reject_bad_arguments():
    """the C entries of the p5f tick report misuse instead of launching: -1 and a message naming the entry"""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC, _ptr
    mpc = PlanarP5fMPC(8, torch.float32)
    L = mpc.L
    nnz = int(mpc.cst.numel())
    assert L.umpcP5fLinearise(8, 0, None, 1.0, _ptr(mpc.y), _ptr(mpc.lin), 0, _ptr(mpc.cst), _ptr(mpc.src), _ptr(mpc.Av), 0, None) == -1
    assert b"umpcP5fLinearise" in L.umpcLastError()
    assert L.umpcP5fLinearise(8, 0, None, 1.0, None, _ptr(mpc.lin), nnz, _ptr(mpc.cst), _ptr(mpc.src), _ptr(mpc.Av), 0, None) == -1
    assert L.umpcQPGatherUpdate(8, 0, nnz, _ptr(mpc.cst), _ptr(mpc.src), None, _ptr(mpc.Av), None) == -1
    assert b"umpcQPGatherUpdate" in L.umpcLastError()
    assert L.umpcP5fStepU(8, 0, 2, 0.002, 1.0, _ptr(mpc.y), None, None) == -1
    assert b"umpcP5fStepU" in L.umpcLastError()
    assert L.umpcP5fStepU(8, 0, 0, 0.002, 1.0, _ptr(mpc.y), None, None) == -1          # mode 0 needs lin
    qp = mpc.qp
    tick = lambda ystate, n_: L.umpcP5fTick(qp.h, _ptr(mpc.Pv), _ptr(mpc.Av), _ptr(mpc.q), _ptr(mpc.l), _ptr(mpc.u), _ptr(qp.x), _ptr(qp.y),
                                            _ptr(qp.z), _ptr(qp.Eprev), _ptr(qp.sol_x), _ptr(qp.sol_y), _ptr(qp.status), _ptr(qp.info), 1.0,
                                            0.002, ystate, _ptr(mpc.lin), n_, _ptr(mpc.cst), _ptr(mpc.src), None)
    assert tick(None, nnz) == -1 and b"umpcP5fTick" in L.umpcLastError()                # no state array
    assert tick(_ptr(mpc.y), nnz - 1) == -2                                              # not this structure's A
    qp.set_kernel("tables")
    assert tick(_ptr(mpc.y), nnz) == -2 and b"does not dispatch" in L.umpcLastError()    # not the assembly kernel: nothing launched
    qp.set_kernel("lane")
    # and the good calls still go through afterwards
    assert L.umpcP5fLinearise(8, 0, None, 1.0, _ptr(mpc.y), _ptr(mpc.lin), nnz, _ptr(mpc.cst), _ptr(mpc.src), _ptr(mpc.Av), 0, None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(mpc.Av).all() and torch.isfinite(mpc.lin).all()


# ------------------------------------------------------------------------------------------------------
# planar/mpc_osqp_p5f_stroke.py (SURVEY 8(f-4)'s second planar structure)
# ------------------------------------------------------------------------------------------------------
def test_stroke_structure_reproduces_the_script():
    """tests/golden/planar_p5f_stroke.npz = what planar/mpc_osqp_p5f_stroke.py computes without osqp, executed where it lies
    (make_golden.py::planar_p5f_stroke): getLin samples, the QP data, and (A, l, u) of every tick of its loop. The registered
    structure + the host-side getLin restatement reproduce them: getLin to round-off, the assembled dense A of all 29 ticks
    EXACTLY (same float64 expressions), P's non-zeros, q, and the bounds with the wrapper's +-1e30 for +-inf."""
    from robobee3d_amd import batchqp as bq
    g = golden("planar_p5f_stroke.npz")
    N = int(g["N"])
    st = bq.p5f_stroke_structure(N)
    assert (st["n"], st["m"]) == g["tick_A"].shape[:0:-1] == (16, 30) and int(g["nx"]) == 7 and int(g["nu"]) == 2
    par = bq.stroke_getlin(g["lin_u0"], g["lin_tf0"], g["lin_y"].T)
    for b in range(len(g["lin_u0"])):
        A = bq.stroke_dense_A(st, par[:, b])
        assert np.abs(A[7:14, 0:7] - g["lin_Ad"][b]).max() < 1e-15 and np.abs(A[7:14, 14:16] - g["lin_Bd"][b]).max() < 1e-15
    full = bq.stroke_dense_A(st, np.ones(st["npar"])) != 0
    assert np.all((g["A_setup"] != 0) <= full)                       # the LTI matrix of prob.setup lies inside the pattern
    for t in range(len(g["tick_u0"])):
        par = bq.stroke_getlin(g["tick_u0"][t:t + 1], g["tick_tf0"][t:t + 1], g["tick_ympc_prev"][t][:, None])
        A = bq.stroke_dense_A(st, par[:, 0])
        assert np.array_equal(A, g["tick_A"][t]) and np.all((g["tick_A"][t] != 0) <= full)
        l, u = st["l"].copy(), st["u"].copy()
        l[:7] = u[:7] = -g["tick_ympc_prev"][t]
        assert np.array_equal(l, np.clip(g["tick_l"][t], -bq.OSQP_INFTY, bq.OSQP_INFTY))
        assert np.array_equal(u, np.clip(g["tick_u"][t], -bq.OSQP_INFTY, bq.OSQP_INFTY))
        # the open-loop state of the script's own plant line (:218)
        Ad0, Bd0 = g["tick_Ad0"][t], g["tick_Bd0"][t]
        assert np.allclose(Ad0 @ g["tick_y_prev"][t] + Bd0 @ np.array([g["tick_u0"][t], g["tick_tf0"][t]]), g["tick_y_ol"][t], rtol=0, atol=1e-12)
    # P: scipy's block_diag of dense blocks keeps explicit zeros (102 stored entries); its NON-ZEROS are the structure's
    Pd = np.zeros(16)
    for j in range(16):
        for k in range(g["P_indptr"][j], g["P_indptr"][j + 1]):
            if g["P_indices"][k] == j:
                Pd[j] = g["P_data"][k]
            else:
                assert g["P_data"][k] == 0
    assert [j for j in range(16) if Pd[j] != 0] == st["P_cols"] and np.array_equal(Pd[st["P_cols"]], st["Pv"])
    assert np.array_equal(np.abs(g["q"]), np.abs(st["q"]))           # (-0. in the script, 0. here)


@pytest.mark.gpu
def test_gpu_stroke_structure_on_the_table_kernel(margin):
    """The stroke structure has no build-time specialisation: `umpcQPSolve` takes it on the table-driven kernel (and the
    wave kernel), fp64 and fp32, for the script's 29 ticks as ONE batch (robot t = tick t: its A through umpcQPGather from
    the stroke_getlin parameters, l[:7] = u[:7] = -ympc), against oracle/osqp_table.py on the fixture's own A, l, u:
    (i) 50 fixed iterations, iterates to 1e-9 (fp64); (ii) the script's solve semantics -- eps 1e-2 (:196), termination test
    every 25 iterations -- same iteration counts and status words (every tick SOLVED after 25 iterations). The script's own
    solve needs pip osqp and is unpinnable (and hands `prob.update(Ax=...)` a dense matrix, :222, hence its `FIXME: says primal
    infeasible`, :228); what is pinned is the data it assembles and the solver that takes them."""
    import torch
    import osqp_table
    from robobee3d_amd import batchqp as bq
    g = golden("planar_p5f_stroke.npz")
    T = len(g["tick_u0"])
    st = bq.p5f_stroke_structure(1)
    par = bq.stroke_getlin(g["tick_u0"], g["tick_tf0"], g["tick_ympc_prev"].T)
    Av_ref = np.zeros((len(st["A_i"]), T))
    for t in range(T):
        k = 0
        for j in range(st["n"]):
            for p_ in range(st["A_p"][j], st["A_p"][j + 1]):
                Av_ref[p_, t] = g["tick_A"][t][st["A_i"][p_], j]
    l = np.clip(g["tick_l"].T, -bq.OSQP_INFTY, bq.OSQP_INFTY)
    u = np.clip(g["tick_u"].T, -bq.OSQP_INFTY, bq.OSQP_INFTY)
    rep = lambda v: np.repeat(np.asarray(v, np.float64)[:, None], T, 1)
    z = lambda r: np.zeros((r, T))
    for dtype, ndt, tol in ((torch.float64, np.float64, 1e-9), (torch.float32, np.float32, 2e-4)):
        for kern in ("tables", "wave"):
            mpc = bq.PlanarP5fStrokeMPC(T, dtype)
            mpc.qp.set_kernel(kern)
            mpc.update(par, g["tick_ympc_prev"].T)
            torch.cuda.synchronize()
            assert np.array_equal(mpc.Av.cpu().numpy().astype(np.float64), Av_ref.astype(ndt).astype(np.float64))
            assert np.array_equal(mpc.l.cpu().numpy().astype(np.float64), l.astype(ndt).astype(np.float64))
            # (i) fixed 50 iterations
            mpc.qp.reset()
            mpc.qp.set_termination(check_every=0, max_iter=50)
            mpc.solve()
            torch.cuda.synchronize()
            r = osqp_table.solve(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"], mpc.qp.s.perm, rep(st["Pv"]), Av_ref, rep(st["q"]),
                                 l, u, z(16), z(30), z(30), np.ones((30, T)), osqp_table.Settings(max_iter=50, eps_abs=1e-2, eps_rel=1e-2),
                                 dtype=ndt)
            lab = "stroke %s %s: " % (kern, "fp64" if ndt is np.float64 else "fp32")
            sc = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))
            f = lambda t_: t_.cpu().numpy().astype(np.float64)
            margin(lab + "iterates after 50 iterations |d| / max(1, |ref|)", max(sc(f(mpc.qp.x), r["x"]), sc(f(mpc.qp.y), r["y"]), sc(f(mpc.qp.z), r["z"])), tol)
            if ndt is np.float64:
                assert np.array_equal(f(mpc.qp.status), r["status"])
            # (ii) the script's solve semantics
            mpc.qp.reset()
            mpc.qp.set_termination(check_every=25, max_iter=4000)
            mpc.solve()
            torch.cuda.synchronize()
            r2 = osqp_table.solve(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"], mpc.qp.s.perm, rep(st["Pv"]), Av_ref, rep(st["q"]),
                                  l, u, z(16), z(30), z(30), np.ones((30, T)),
                                  osqp_table.Settings(max_iter=4000, eps_abs=1e-2, eps_rel=1e-2, check_termination=25), dtype=ndt)
            if ndt is np.float64:
                assert np.array_equal(f(mpc.qp.status), r2["status"]) and np.array_equal(f(mpc.qp.info[4]).astype(np.int32), r2["iters"])
                ok = r2["status"] > 0
                margin(lab + "solution at eps 1e-2 |d| / max(1, |ref|)", sc(f(mpc.qp.sol_x)[:, ok], r2["sol_x"][:, ok]), 1e-8)
            else:
                margin(lab + "status words differing from the float32 oracle (of %d)" % T, int(np.count_nonzero(f(mpc.qp.status) != r2["status"])), 2)


@pytest.mark.gpu
def test_gpu_p5f_fused_tick_equals_the_three_launches():
    """umpcP5fTick (round 5): getLin, the A update and the plant tick as the prologue of the p5f10 assembly kernel -- one
    launch per tick of planar/mpc_osqp_p5f.py:157-176 instead of three. Same function of the same numbers: after 6 ticks a
    fused and an unfused controller hold the same bits in every array (state, lin, A, iterates, solution, status), for a
    ragged batch (B = 200: a last workgroup with 8 lanes) and for the benchmarked B = 16 384; with per-robot finite input
    limits on half of the workgroups (the general variant of the loop block) as well."""
    import torch
    from robobee3d_amd.batchqp import PlanarP5fMPC
    for B in (200, 16384):
        rng = np.random.default_rng(20201119)
        pert = rng.uniform(-0.1, 0.1, (2, B)).astype(np.float32)
        pair = []
        for fused in (True, False):
            mpc = PlanarP5fMPC(B, torch.float32)
            mpc.fused = fused
            mpc.y[0] = torch.as_tensor(pert[0]).cuda()
            mpc.y[3] = torch.as_tensor(pert[1]).cuda()
            rows = [77 + t for t, j in enumerate(mpc.st["var_order"]) if j >= 77]
            odd = torch.as_tensor(((np.arange(B) // 64) % 2) == 1).cuda()
            for r in rows:
                mpc.l[r][odd] = -4.0
                mpc.u[r][odd] = 4.0
            for ti in range(2, 8):
                mpc.tick(0.002 * ti)
            torch.cuda.synchronize()
            assert mpc.fused == fused and mpc.qp.kernel_name == "p5f10+asm"
            pair.append(mpc)
        a, b = pair
        for name in ("y", "lin", "Av"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (B, name)
        for name in ("x", "y", "z", "sol_x", "sol_y", "status", "Eprev"):
            assert torch.equal(getattr(a.qp, name), getattr(b.qp, name)), (B, name)
        assert bool((a.qp.status > 0).all()) and bool(torch.isfinite(a.y).all())
    # a handle that does not dispatch the assembly kernel refuses the fused entry and the front end falls back
    m64 = PlanarP5fMPC(64, torch.float32)
    m64.qp.set_kernel("tables")
    m64.tick(0.004); m64.tick(0.006)
    assert m64.fused is False and bool(torch.isfinite(m64.y).all())

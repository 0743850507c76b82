"""GPU parity evidence added in round 3 (run on the MI355X box: pytest -m gpu). Everything goes through the C ABI of
libumpc_mi355x.so; the oracle is only the checker.

1. The benchmarked planar-p5f fp32 ASSEMBLY route (bqp_fixed_p5f10_asm_kernel) and the UprightMPC2 N = 5 fp32
   specialisation straight against oracle/osqp_table.py in float32 (bit-pinned to the reference C on the N = 3
   fixtures, tests/test_bqp.py::test_table_oracle_is_bitwise_the_c_oracle) -- not through another GPU kernel.
2. The reference's own closed-loop log (template/uprightmpc2.py:120-159, tests/golden/closed_loop_hover.npz: 2 500
   substeps, 96 MPC fires at the float-jittered substeps 26, 52, 77 ...) replayed through the product.
3. osqp_update_bounds' reject path (template/uprightmpc2/osqp.c:801-808), reachable only with TtoWmax < 0.
"""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


@pytest.mark.parametrize("ulim", [None, 4.0], ids=["reference_bounds", "finite_input_limits"])
def test_p5f_fp32_assembly_route_against_the_table_oracle(torch_cuda, margin, ulim):
    """(ulim: the reference's problem has infinite state / input bounds, planar/mpc_osqp_p5f.py:94-97 -- every inequality row
    is a loose row and the wave takes the LOOSE variant of the loop; with finite input limits +-ulim the same waves take
    the general loop: rho = 0.1 rows, streamed bounds, active clipping.)
    BASELINE configs[3]'s kernel (fp32, n = 87, m = 164, 50 iterations per tick) for 5 consecutive ticks of a ragged
    batch (B = 200: three full waves and 8 lanes), every tick compared with osqp_table.solve(dtype=float32) on the SAME
    per-robot A values (read back from the GPU's own getLin), warm start carried on both sides independently. Tick 1 is
    a cold start (the wave is refused by the all-assembly route and takes C++ glue around the assembly blocks), ticks
    2.. take the all-assembly route."""
    torch = torch_cuda
    import osqp_table
    from robobee3d_amd.batchqp import PlanarP5fMPC
    B = 200
    mpc = PlanarP5fMPC(B, torch.float32)
    assert mpc.qp.kernel_name == "p5f10+asm"
    rng = np.random.default_rng(20201119)
    y0 = np.zeros((7, B), np.float32)
    y0[0] = rng.uniform(-0.1, 0.1, B)
    y0[3] = rng.uniform(-0.1, 0.1, B)
    mpc.y.copy_(torch.as_tensor(y0).cuda())
    st, perm = mpc.st, mpc.qp.s.perm
    if ulim is not None:
        # the box rows of the N inputs: umin <= u_k <= umax (the solver's labelling groups the QP's components: the box row of
        # the script's variable j is row 77 + its place in st["var_order"])
        rows = [77 + t for t, j in enumerate(st["var_order"]) if j >= 77]
        assert len(rows) == 10
        mpc.l[rows] = -ulim
        mpc.u[rows] = ulim
    f = lambda t: t.cpu().numpy()
    z32 = lambda r: np.zeros((r, B), np.float32)
    x, y, z, E = z32(87), z32(164), z32(164), np.ones((164, B), np.float32)
    flips = 0
    clipped = 0
    for ti in range(2, 7):
        t = 0.002 * ti
        mpc.linearise(15.0 * np.sin(2 * np.pi * 170 * t))
        r = osqp_table.solve(87, 164, st["A_p"], st["A_i"], st["P_cols"], perm, f(mpc.Pv), f(mpc.Av), f(mpc.q),
                             f(mpc.l), f(mpc.u), x, y, z, E, osqp_table.Settings(max_iter=50), dtype=np.float32)
        x, y, z, E = r["x"], r["y"], r["z"], r["E"]
        mpc.qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u)
        torch.cuda.synchronize()
        got = {k: f(getattr(mpc.qp, k)).astype(np.float64) for k in ("x", "y", "z", "sol_x", "sol_y", "Eprev")}
        lab = "tick %d: " % (ti - 1)
        margin(lab + "iterates x, y, z  |d| / max(1, |ref|)", max(_rel(got[k], r[k]) for k in ("x", "y", "z")), 1.2e-5)
        margin(lab + "sol_x, sol_y  |d| / max(1, |ref|)", max(_rel(got[k], r[k]) for k in ("sol_x", "sol_y")), 6e-5)
        margin(lab + "E (Ruiz row scaling) relative", float(np.max(np.abs(got["Eprev"] / r["E"] - 1))), 1e-6)
        info = f(mpc.qp.info).astype(np.float64)
        # residuals of a converged fp32 iterate are differences of nearly equal numbers: magnitudes agree, digits do not
        margin(lab + "|pri_res - ref|", float(np.max(np.abs(info[0] - r["pri_res"]))), 1e-5)
        margin(lab + "|dua_res - ref|", float(np.max(np.abs(info[1] - r["dua_res"]))), 3.5e-5)
        flips += int(np.count_nonzero(f(mpc.qp.status) != r["status"]))
        assert set(np.unique(f(mpc.qp.status))).issubset({1, 2, -2})
        if ulim is not None:
            clipped += int(np.count_nonzero(np.abs(np.abs(r["sol_x"][-10:]) - ulim) < 1e-3))
    margin("status flips over 5 ticks x 200 robots", flips, 10)
    assert ulim is None or clipped > 0          # the limits are active: the projection does clip


ITER_TOL_N5 = 3e-3    # measured 6e-4 .. 1e-3 (the N = 3 bound against the reference is 1e-3)


def test_umpc2_n5_fp32_specialisation_against_the_table_oracle(torch_cuda, margin):
    """UprightMPC2N(N = 5) fp32 (the `umpc2n5` straight-line specialisation, template_controllers.py:170-258 with a
    longer horizon): three consecutive controller steps from the reference's recorded states against
    osqp_table.solve(dtype=float32) on the GPU-assembled data (itself pinned by assembly_fp64_N5.npz in tests/test_bqp.py)."""
    torch = torch_cuda
    import osqp_table
    from robobee3d_amd.batchqp import UprightMPC2N
    from test_bqp import _state_ref_from_seq
    seq = golden("seq_iter50.npz")
    B = 130
    idx = np.arange(B)
    st, ref, T0 = _state_ref_from_seq(seq, idx, np.float32)
    mpc = UprightMPC2N(B, 5, dtype=torch.float32)
    assert mpc.qp.kernel_name == "umpc2n5"
    mpc.T0.copy_(torch.as_tensor(T0).cuda())
    S, R = torch.as_tensor(st).cuda(), torch.as_tensor(ref).cuda()
    s = mpc.st
    f = lambda t: t.cpu().numpy()
    z32 = lambda r: np.zeros((r, B), np.float32)
    x, y, z, E = z32(75), z32(65), z32(65), np.ones((65, B), np.float32)
    flips = 0
    for step in range(3):
        mpc.assemble(S, R)           # with the accumulator T0 the next update() will use
        torch.cuda.synchronize()
        data = (f(mpc.Pv), f(mpc.Av), f(mpc.q), f(mpc.l), f(mpc.u))
        start = (x, y, z, E)
        r = osqp_table.solve(75, 65, s["A_p"], s["A_i"], s["P_cols"], mpc.qp.s.perm, *data, *start,
                             osqp_table.Settings(max_iter=50), dtype=np.float32)
        x, y, z, E = r["x"], r["y"], r["z"], r["E"]
        T0_before = f(mpc.T0).astype(np.float64)
        out = f(mpc.update(S, R)).astype(np.float64)
        # the next step starts both sides from the ORACLE's iterates (single-step comparisons from an identical state,
        # like the N = 3 fixtures): the stated band is per step, errors carried through the warm start add up
        nxt = {"x": x, "y": y, "z": z, "Eprev": E}
        got = {k: f(getattr(mpc.qp, k)).astype(np.float64) for k in ("x", "y", "z", "sol_x", "sol_y")}
        lab = "step %d: " % (step + 1)
        # the metric of test_single_step_matches_reference_golden (N = 3 against the reference): per robot, relative to the
        # largest entry of the vector -- after 50 fp32 iterations of this ill-conditioned KKT system (sigma = 1e-6, rho_eq =
        # 100) single entries carry the accumulated round-off of the whole vector
        def vrel(a, b):
            return float(np.max(np.abs(a - b) / (1e-3 + np.abs(b).max(axis=0, keepdims=True))))
        margin(lab + "iterates x, y, z  |d| / (1e-3 + max|ref|)", max(vrel(got[k], r[k]) for k in ("x", "y", "z")), ITER_TOL_N5)
        margin(lab + "sol_x, sol_y  |d| / (1e-3 + max|ref|)", max(vrel(got[k], r[k]) for k in ("sol_x", "sol_y")), ITER_TOL_N5)
        ok = r["status"] > 0
        # the stated fp32 band of the path on the controller outputs (thrust, moments): 3e-5, max(2e-2, 1e-3 |u|)
        margin(lab + "thrust = T0 + sol_x[60]", float(np.max(np.abs(out[0][ok] - (T0_before + r["sol_x"][60])[ok]))), 3e-5)
        mref = r["sol_x"][61:63][:, ok]
        band = np.maximum(2e-2, 1e-3 * np.abs(mref))
        # N = 5 is worse conditioned than the N = 3 path the band was stated for: the yardstick is the oracle's OWN
        # fp32-vs-fp64 distance on the same data and start (what the band is 3x of at N = 3)
        r64 = osqp_table.solve(75, 65, s["A_p"], s["A_i"], s["P_cols"], mpc.qp.s.perm, *[a.astype(np.float64) for a in data],
                               *[a.astype(np.float64) for a in start], osqp_table.Settings(max_iter=50), dtype=np.float64)
        own = float(np.max(np.abs(r64["sol_x"][61:63][:, ok] - mref) / band))
        margin(lab + "oracle fp32 vs oracle fp64 moments, band units (yardstick)", own, 10.0)
        margin(lab + "|d moment| / max(2e-2, 1e-3|u|)", float(np.max(np.abs(out[1:3][:, ok] - mref) / band)), 3.0 * max(1.0, own))
        flips += int(np.count_nonzero(f(mpc.qp.status) != r["status"]))
        for k, val in nxt.items():
            getattr(mpc.qp, k).copy_(torch.as_tensor(val).cuda())
    margin("status flips over 3 steps x 130 robots", flips, 40)


def _hover_ic():
    from scipy.spatial.transform import Rotation
    R = Rotation.from_euler("xyz", [0.5, -0.5, 0]).as_matrix()          # template/uprightmpc2.py:101-103
    st = np.zeros((18, 1))
    st[3:12, 0] = R.T.ravel()                                             # column-major
    st[12, 0] = 0.1
    ref = np.zeros((9, 1))
    ref[8] = 1.0
    return st, ref


def test_reference_closed_loop_log_replayed_on_the_gpu(torch_cuda, margin):
    """controlTest(mdl, 500, hlInterval=5) (template/uprightmpc2.py:87-159) with the reference's own C controller,
    recorded in tests/golden/closed_loop_hover.npz: the product is driven at the RECORDED fire substeps (first fire at
    substep 26, gaps 25 / 26: `tt[ti] - thlPrev > hlInterval` in floating point) and must reproduce the y / u / accdes
    log and the logMetric pair.
      (a) the reference's arithmetic split: fp32 controller step (umpcBatchUpdate) + fp64 Euler/expm plant substeps
          (umpcBatchPlant) -- the tight comparison;
      (b) BatchUprightMPC.control_test_log(fire=...) all in fp32 and (c) all in fp64."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC
    g = golden("closed_loop_hover.npz")
    fire = set(g["fire"].tolist())
    Nt = len(g["t"])
    assert Nt == 2500 and len(fire) == 96 and min(fire) == 26
    st, ref = _hover_ic()
    # (a) mixed precision exactly like the reference harness
    c32 = BatchUprightMPC(1, torch.float32, plant_mode=0)
    p64 = BatchUprightMPC(1, torch.float64, plant_mode=0)
    p64.set_state(st, ref)
    c32.set_state(st.astype(np.float32), ref.astype(np.float32))
    u = torch.zeros((3, 1), dtype=torch.float64, device="cuda")
    Y, U, ACC = np.zeros((Nt, 12)), np.zeros((Nt, 3)), np.zeros((Nt, 6))
    for ti in range(Nt):
        if ti in fire:
            c32.state.copy_(p64.state.to(torch.float32))
            c32.update()
            u = c32.out[0:3].to(torch.float64).clone()
            ACC[ti] = c32.out[3:9, 0].cpu().numpy()
        u[1:3].clamp_(-100.0, 100.0)
        p64.plant(u, 1)
        s = p64.state[:, 0].cpu().numpy()
        Y[ti] = np.hstack((s[0:3], s[9:12], s[12:18]))
        U[ti] = u[:, 0].cpu().numpy()
    lab = "fp32 controller + fp64 plant: "
    margin(lab + "|p - ref log| mm", float(np.abs(Y[:, 0:3] - g["y"][:, 0:3]).max()), 6e-4)
    margin(lab + "|s - ref log|, |dq - ref log|", float(np.abs(Y[:, 3:] - g["y"][:, 3:]).max()), 6e-5)
    margin(lab + "|thrust - ref log|", float(np.abs(U[:, 0] - g["u"][:, 0]).max()), 3e-5)
    margin(lab + "|moment - ref log| / max(2e-2, 1e-3 |u|)",
           float((np.abs(U[:, 1:] - g["u"][:, 1:]) / np.maximum(2e-2, 1e-3 * np.abs(g["u"][:, 1:]))).max()), 1.0)
    rows = sorted(fire)
    margin(lab + "|accdes - ref log| at the 96 fires", float(np.abs(ACC[rows] - g["accdes"][rows]).max()), 3e-5)
    met = np.array([np.mean(np.sum(Y[:, :3] ** 2, axis=1)), np.mean(np.sum(U[:, 1:3] ** 2, axis=1))])
    margin(lab + "logMetric pair, relative", float(np.max(np.abs(met / g["metric"] - 1))), 2e-4)
    assert np.linalg.norm(Y[-1, :3]) < 0.02 and abs(Y[-1, 5] - 1) < 1e-4      # hover converges to the origin, upright
    # (b), (c): the product's own harness at the recorded schedule, one precision throughout. fp32 runs BOTH forms of the
    # stream (B = 2 takes the quad form by default since round 4). The lane form keeps its round-3 bound on the logMetric pair
    # (1.5e-4; achieved 2.7e-5). The quad form ends this ONE trajectory at 1.9e-4: not a property of the form -- over 1 024
    # trajectories of the same length the two forms have the same error distribution against the fp64 oracle
    # (tests/test_r5_evidence.py::test_lane_and_quad_forms_share_their_closed_loop_error_statistics, whose 95th percentile
    # is what the quad bound here is taken from) -- a closed loop amplifies a different accumulation order differently
    # along each trajectory.
    for tdt, form, name, bp, bs, bm in ((torch.float32, "lane", "fp32 harness (lane form): ", 1e-3, 1e-4, 1.5e-4),
                                        (torch.float32, "quad", "fp32 harness (quad form): ", 1e-3, 1e-4, 5e-4),
                                        (torch.float64, "auto", "fp64 harness: ", 1e-3, 1e-4, 1.5e-4)):
        m = BatchUprightMPC(2, tdt, plant_mode=0)
        m.set_step_kernel(form)
        m.set_state(np.repeat(st, 2, 1), np.repeat(ref, 2, 1))
        log = m.control_test_log(500.0, robots=(0, 1), fire=g["fire"])
        if tdt == torch.float32:
            assert m.kernel_name == ("umpc_rollout_asm_kernel" if form == "lane" else "umpc_rollout_asm_quad_kernel")
        for r in (0, 1):
            lg = log[r]
            assert lg["y"].shape == (Nt, 12)
            margin(name + "robot %d |p - ref log| mm" % r, float(np.abs(lg["y"][:, 0:3] - g["y"][:, 0:3]).max()), bp)
            margin(name + "robot %d |s, dq - ref log|" % r, float(np.abs(lg["y"][:, 3:] - g["y"][:, 3:]).max()), bs)
            margin(name + "robot %d logMetric relative" % r, float(np.max(np.abs(np.array(lg["metric"]) / g["metric"] - 1))), bm)
            assert np.array_equal(np.nonzero(np.abs(lg["accdes"]).sum(axis=1))[0], np.array(rows))
        assert np.array_equal(log[0]["y"], log[1]["y"])


def test_bounds_reject_path_documented_difference(torch_cuda, oracle_built, structure, margin):
    """TtoWmax < 0 crosses the thrust rows' bounds (l = -T0 > u = Tmax - T0). The reference's osqp_update_bounds then
    returns 1 WITHOUT applying any bound (template/uprightmpc2/osqp.c:801-808); umpcUpdate drops that return value
    (uprightmpc2.c:246) and keeps solving the code-generated placeholder problem (l = 0, u = 1e30 on all 39 rows,
    every row an inequality at rho = 0.1, workspace.c:476-557) -- finite but meaningless commands, return value 0:
    tests/golden/bounds_reject.npz (12 calls of the compiled reference; the oracle's FAITHFUL mode reproduces them bit
    for bit, tests/test_oracle_golden.py). The product applies the crossed pair as assembled: the rows
    classify as equalities (u - l < 1e-4, auxil.c:118) and project onto u (proj.c:4-14: min(max(v, l), u) = u), i.e. the
    dynamics constraints STAY in force and the thrust is driven to Tmax -- the oracle's CANONICAL mode. This test pins
    that documented difference from both sides: product == canonical oracle inside the fp32 band over the 12-call
    sequence (same return value 0 as the reference), and it records how far both are from the reference's placeholder
    solve."""
    from robobee3d_amd.uprightmpc2py import UprightMPC2C
    seq = golden("bounds_reject.npz")
    o = oracle_built.Oracle(np.float32, perm=structure["perm"], TtoWmax=-2.0)
    o.set_canonical(True)
    upc = UprightMPC2C(5, 9.81e-3, -2.0, 1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2, np.array([3333., 3333., 1000.]), 50)
    wt = wm = wa = dref = 0.0
    for k in range(len(seq["p0"])):
        args = (seq["p0"][k], seq["R0"][k], seq["dq0"][k], seq["pdes"][k], seq["dpdes"][k], seq["sdes"][k],
                float(seq["actualT0"][k]))
        uq_o, ac_o = o.update(*args)
        uq, ac = upc.update(*args)          # raises on a non-zero return value: the reference returns 0 here too
        l, u, _ = upc.vectors()
        assert np.all(l[36:] > u[36:])      # the crossed pair, exactly as the reference assembles it
        if k == 0:                          # (later calls assemble with a different accumulated thrust T0)
            np.testing.assert_allclose(l, seq["l"][k], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(u, seq["u"][k], rtol=1e-6, atol=1e-7)
        wt = max(wt, abs(float(uq[0]) - float(uq_o[0])))
        wm = max(wm, float(np.max(np.abs(uq[1:] - uq_o[1:]) / np.maximum(2e-2, 1e-3 * np.abs(uq_o[1:])))))
        wa = max(wa, float(np.max(np.abs(ac - ac_o))))
        dref = max(dref, float(np.max(np.abs(uq[1:] - seq["uquad"][k][1:]))))
        assert upc.status() in (1, 2, -2)
    # errors of earlier calls feed later ones through the warm start: the sequence band of
    # test_reference_boundary_dropin_sequence (1e-4 / 3x the moment band / 1e-4)
    margin("Tmax < 0: product vs canonical oracle |d thrust| (12-call sequence)", wt, 5e-6)
    margin("Tmax < 0: product vs canonical oracle |d moment| / max(2e-2, 1e-3|u|)", wm, 0.15)
    margin("Tmax < 0: product vs canonical oracle |d accdes|", wa, 6e-6)
    assert dref > 1.0, dref     # ... and it is NOT the reference's placeholder solve (moments differ by O(1..10))


def test_bounds_reject_compat_switch_reproduces_the_reference(torch_cuda, margin):
    """Round 4: with umpcSetCompat(up, UMPC_COMPAT_BOUNDS_REJECT) the drop-in reproduces the REFERENCE on
    crossed bounds (TtoWmax < 0): osqp_update_bounds returns before applying anything (osqp.c:801-808), umpcUpdate drops
    the value (uprightmpc2.c:246), the step solves with the bounds the workspace still holds -- the generated placeholder
    l = 0, u = 1e30, every row an inequality at rho = 0.1 (workspace.c:476-557) -- and the new q, P, A. Checked against the
    12 calls of the COMPILED REFERENCE in tests/golden/bounds_reject.npz: commands inside the fp32 band of a warm-started
    sequence (test_reference_boundary_dropin_sequence's), every status word equal, up->l / up->u the assembled (crossed)
    pair; then the default (switch off) is still the documented canonical behaviour (thrust driven to Tmax)."""
    from robobee3d_amd.uprightmpc2py import COMPAT_BOUNDS_REJECT, UprightMPC2C
    seq = golden("bounds_reject.npz")
    Ib = np.array([3333., 3333., 1000.])
    upc = UprightMPC2C(5, 9.81e-3, -2.0, 1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2, Ib, 50)
    assert upc.set_compat(COMPAT_BOUNDS_REJECT) == 0
    wt = wm = wa = 0.0
    nstat = 0
    for k in range(len(seq["p0"])):
        uq, ac = upc.update(seq["p0"][k], seq["R0"][k], seq["dq0"][k], seq["pdes"][k], seq["dpdes"][k], seq["sdes"][k],
                            float(seq["actualT0"][k]))
        l, u, _ = upc.vectors()
        assert np.all(l[36:] > u[36:])
        ru, ra = seq["uquad"][k].astype(np.float64), seq["accdes"][k].astype(np.float64)
        wt = max(wt, abs(float(uq[0]) - ru[0]))
        wm = max(wm, float(np.max(np.abs(uq[1:] - ru[1:]) / np.maximum(2e-2, 1e-3 * np.abs(ru[1:])))))
        wa = max(wa, float(np.max(np.abs(ac - ra))))
        nstat += int(upc.status() != int(seq["status"][k]))
    # measured on the MI355X (round 4): 1.4e-5 / 2.5e-3 band units / 5.5e-8, no status word differs
    margin("compat bounds-reject vs the compiled reference |d thrust| (12-call sequence)", wt, 4e-5)
    margin("compat bounds-reject vs the compiled reference |d moment| / max(2e-2, 1e-3|u|)", wm, 0.01)
    margin("compat bounds-reject vs the compiled reference |d accdes|", wa, 2e-7)
    margin("compat bounds-reject: status words differing from the reference's (of 12)", nstat, 0)
    # default (no switch): the documented canonical behaviour, NOT the reference's placeholder solve
    upd = UprightMPC2C(5, 9.81e-3, -2.0, 1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2, Ib, 50)
    uq_d, _ = upd.update(seq["p0"][0], seq["R0"][0], seq["dq0"][0], seq["pdes"][0], seq["dpdes"][0], seq["sdes"][0],
                         float(seq["actualT0"][0]))
    assert np.max(np.abs(uq_d[1:] - seq["uquad"][0][1:])) > 1.0


def test_init_on_a_copy_of_a_live_pod_does_not_destroy_the_original(torch_cuda):
    """`b = a; umpcInit(&b, ...)` (the reference's plain-value struct allows it) gives b its own controller and
    leaves a's alive; re-initialising a in place still releases a's previous controller."""
    import ctypes as C
    from robobee3d_amd import _lib
    L = _lib.lib()
    seq = golden("seq_iter50.npz")
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    Ib = f32([3333.0, 3333.0, 1000.0])
    prm = [C.c_float(v) for v in (5.0, 9.81e-3, 2.0, 1e1, 1e3, 1.0, 5.0, 1e3, 2e3, 1e-1, 1e-2)]
    n0 = L.umpcLiveControllers()
    a = _lib.UprightMPC_t()
    L.umpcInit(C.byref(a), *prm, fp(Ib), C.c_int(50))
    b = _lib.UprightMPC_t()
    C.memmove(C.byref(b), C.byref(a), C.sizeof(a))          # b = a
    L.umpcInit(C.byref(b), *prm, fp(Ib), C.c_int(50))
    assert L.umpcLiveControllers() == n0 + 2
    args = [f32(seq["p0"][0]), f32(seq["R0"][0].T.ravel()), f32(seq["dq0"][0]), f32(seq["pdes"][0]),
            f32(seq["dpdes"][0]), f32(seq["sdes"][0])]
    ua, ub, ac = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(6, np.float32)
    assert L.umpcUpdate(C.byref(a), fp(ua), fp(ac), *[fp(x) for x in args], C.c_float(-1.0)) == 0     # a is still alive
    assert L.umpcUpdate(C.byref(b), fp(ub), fp(ac), *[fp(x) for x in args], C.c_float(-1.0)) == 0
    assert np.array_equal(ua, ub)
    L.umpcInit(C.byref(a), *prm, fp(Ib), C.c_int(50))       # in place: releases a's previous controller
    assert L.umpcLiveControllers() == n0 + 2
    L.umpcRelease(C.byref(a)); L.umpcRelease(C.byref(b))
    assert L.umpcLiveControllers() == n0


def test_reinitialising_a_pod_releases_the_previous_controller(torch_cuda):
    """The reference allows umpcInit on the same UprightMPC_t again (a gain sweep re-creates controllers,
    template/uprightmpc2.py:272-303; Simulink / MCU start and stop): here that must release the previous controller's
    device state instead of orphaning it (umpcRelease is documented as optional)."""
    import ctypes as C
    from robobee3d_amd import _lib
    L = _lib.lib()
    seq = golden("seq_iter50.npz")
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    Ib = f32([3333.0, 3333.0, 1000.0])
    up = _lib.UprightMPC_t()
    n0 = L.umpcLiveControllers()
    outs = []
    for rep in range(5):
        prm = (5.0, 9.81e-3, 2.0, 1e1, 1e3, 1.0 + (rep % 2), 5.0, 1e3, 2e3, 1e-1, 1e-2)
        L.umpcInit(C.byref(up), *[C.c_float(v) for v in prm], fp(Ib), C.c_int(50))
        assert L.umpcLiveControllers() == n0 + 1, rep
        uq, ac = np.zeros(3, np.float32), np.zeros(6, np.float32)
        args = [f32(seq["p0"][0]), f32(seq["R0"][0].T.ravel()), f32(seq["dq0"][0]), f32(seq["pdes"][0]),
                f32(seq["dpdes"][0]), f32(seq["sdes"][0])]
        assert L.umpcUpdate(C.byref(up), fp(uq), fp(ac), *[fp(a) for a in args], C.c_float(-1.0)) == 0
        outs.append(uq.copy())
    # a re-initialised controller starts pristine: equal weights give equal first calls, other weights differ
    assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[3]) and not np.array_equal(outs[0], outs[1])
    L.umpcRelease(C.byref(up))
    assert L.umpcLiveControllers() == n0


def test_device_draws_on_the_gpu_equal_the_host_draws(torch_cuda):
    """initial tilts and the Monte-Carlo inertia / thrust-gain draws are generated ON the device from the
    counter hash keyed by the global robot index. Bit-identical to batch.monte_carlo_draws; the tilt ANGLES are
    bit-identical, the rotation entries agree to the last fp32 bit or two (device sin / cos)."""
    torch = torch_cuda
    from robobee3d_amd import batch
    lo = 7 * 131072                     # the last rank's block of config 5
    Ib, g = batch.monte_carlo_draws(131072, 20201120, np.float32, index_offset=lo)
    Ibd, gd = batch.monte_carlo_draws_device(131072, 20201120, torch.float32, index_offset=lo)
    assert np.array_equal(Ib, Ibd.cpu().numpy()) and np.array_equal(g, gd.cpu().numpy())
    st, ref = batch.hover_initial_conditions(65536, 20201118, np.float32, index_offset=lo)
    std, refd, _ = batch.hover_initial_conditions_device(65536, 20201118, torch.float32, index_offset=lo)
    assert np.array_equal(ref, refd.cpu().numpy())
    d = np.abs(st - std.cpu().numpy())
    assert d.max() <= 1.2e-7 and np.count_nonzero(d) <= st.size // 100, (d.max(), np.count_nonzero(d))


def test_assembly_kernel_options_agree_with_the_cpp_kernel(torch_cuda, margin):
    """SURVEY 8(f) workloads on the all-assembly fp32 kernel (task table, per-robot weights, fused WL step, alone and
    together, ragged batch) against the C++ kernel around the assembly loop (`set_step_kernel("cpp")`, csrc/umpc_step.h,
    itself checked against the oracle in tests/test_tasks_weights.py and tests/test_wl_step.py): two fp32 evaluation
    orders of the same closed loop."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, BatchWLCon, hover_initial_conditions
    from test_wl_step import _args
    gl = golden("mpc_wl_loop.npz")
    B, K = 200, 6
    st, ref = hover_initial_conditions(B, 11, np.float32, tilt=0.2)
    rng = np.random.default_rng(5)
    W = np.tile(np.array([1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2], np.float32)[:, None], (1, B))
    W[2] = rng.uniform(0.05, 8.0, B)
    W[4] = rng.uniform(50.0, 5e3, B)
    for opts in (("task",), ("weights",), ("wl",), ("task", "weights", "wl")):
        res = {}
        for mode in ("auto", "cpp"):
            m = BatchUprightMPC(B, torch.float32, plant_mode=1)
            m.set_step_kernel(mode)
            r = ref.copy()
            s0 = st.copy()
            if "task" in opts:
                r[:] = 0
                r[0:3] = np.random.default_rng(6).normal(size=(3, B))
                s0[0:3] = r[0:3]
                m.set_task("helix", t_ms=35.0, trajAmp=40, trajFreq=2, dz=0.1, useY=True)
            m.set_state(s0, r)
            if "weights" in opts:
                m.set_weights(W)
            wl = None
            if "wl" in opts:
                wl = BatchWLCon(B, *_args(gl), dtype=torch.float32)
                m.set_wl(wl)
            m.rollout(K // 2)
            m.rollout(K - K // 2)               # the task clock and the WL state carry over between launches
            assert m.kernel_name == ("umpc_rollout_asm_quad_kernel" if mode == "auto" else "umpc_rollout_kernel<float>")   # B <= 16 384
            res[mode] = [t.cpu().numpy().astype(np.float64) for t in (m.state, m.out, m.stats, m.ctrl[123:124])] + \
                        ([wl.u.cpu().numpy().astype(np.float64), wl.w0.cpu().numpy().astype(np.float64)] if wl else [])
        a, c = res["auto"], res["cpp"]
        lab = "asm vs C++ kernel, options %s: " % "+".join(opts)
        margin(lab + "|dp| mm", float(np.abs(a[0][0:3] - c[0][0:3]).max()), 5e-4)
        margin(lab + "|dR|, |ddq|", float(np.abs(a[0][3:] - c[0][3:]).max()), 6e-5)
        margin(lab + "|d thrust|", float(np.abs(a[1][0] - c[1][0]).max()), 1e-5)
        margin(lab + "|d moment| / max(2e-2, 1e-3|u|)", float((np.abs(a[1][1:3] - c[1][1:3]) / np.maximum(2e-2, 1e-3 * np.abs(c[1][1:3]))).max()), 0.6)
        margin(lab + "stats relative", float(np.max(np.abs(a[2] - c[2]) / (1e-6 + np.abs(c[2])))), 3e-4)
        margin(lab + "|d T0 accumulator|", float(np.abs(a[3] - c[3]).max()), 4e-6)
        if "wl" in opts:
            margin(lab + "|d u4| / rate limit", float((np.abs(a[4] - c[4]) / np.array([5.0, 0.01, 0.01, 0.01])[:, None]).max()), 1.2e-3)
            margin(lab + "|d w0|", float(np.abs(a[5] - c[5]).max()), 2.5e-4)


def test_batch_beyond_the_old_31_bit_workspace_limit(torch_cuda):
    """Round 2 sent batches with WS_ROWS x B x 4 >= 2^31 (B >~ 960 k on one GPU) to the C++ kernel silently: the assembly
    stream formed workspace offsets in 32 bits. Its workspace pointer is now the first row it uses (85 rows, not 559), so
    a million robots per GPU (the 288 GB of HBM hold far more) stay on the all-assembly kernel -- and robot i still
    computes what it computes in a small batch (lanes are independent; same start, same steps)."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions_device
    B, K = 1048576, 2
    assert 559 * B * 4 >= (1 << 31)
    m = BatchUprightMPC(B, torch.float32, plant_mode=1)
    st, ref, _ = hover_initial_conditions_device(B, 20201118, torch.float32)
    m.set_state(st, ref)
    m.rollout(K)
    assert m.kernel_name == "umpc_rollout_asm_kernel"
    for lo in (0, B - 4096):
        s = BatchUprightMPC(4096, torch.float32, plant_mode=1, global_batch=B)   # a block of the million-robot job
        st4, ref4, _ = hover_initial_conditions_device(4096, 20201118, torch.float32, index_offset=lo)
        s.set_state(st4, ref4)
        s.rollout(K)
        assert torch.equal(s.state, m.state[:, lo:lo + 4096]) and torch.equal(s.out, m.out[:, lo:lo + 4096])
        assert torch.equal(s.ctrl, m.ctrl[:, lo:lo + 4096])
    assert bool(torch.isfinite(m.state).all()) and float((m.status > 0).float().mean()) > 0.99

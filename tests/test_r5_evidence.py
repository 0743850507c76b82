"""GPU parity evidence added in round 5 (run on the MI355X box: pytest -m gpu). Everything goes through the C ABI of
libumpc_mi355x.so; the oracle is only the checker.

1. BASELINE configs[3] (planar p5f, N = 10) AT ITS BENCHMARKED SIZE, B = 16 384, on the shipped four-wavefront route, with a
   sample of robots replayed through oracle/osqp_table.py in float32 (round 4 checked B = 200 only).
2. Partition invariance without pinning the kernel form by hand: the form is chosen from the size of the whole job
   (umpcBatchSetGlobalBatch), so 65 536 robots as 8 blocks of 8 192 equal the undivided batch bit for bit.
3. The lane and the quad form of the fp32 step stream have the same CLOSED-LOOP error statistics against the fp64 oracle
   (on the single trajectory of the reference's log the quad form was 7x further from the reference than the
   lane form; here 1 024 trajectories say whether that is a property of the form or one draw).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def test_p5f_full_size_batch_against_the_table_oracle(torch_cuda, margin):
    """planar/mpc_osqp_p5f.py:157-176 for B = 16 384 robots (seed 20201119, sigma / phi perturbed U(-0.1, 0.1): SURVEY 8d
    config 4), 5 ticks on the kernel bench.py times (`bqp_fixed_p5f10_asm_kernel`: 256 workgroups of four wavefronts).
    64 robots -- one per fourth workgroup at a varying lane, plus the first and the last robot of the batch -- are replayed
    through osqp_table.solve(dtype=float32) on the SAME per-robot A values (the GPU's own getLin), warm start carried on
    both sides independently. Workgroups with an odd index get finite input limits |u_k| <= 4 on all their robots, so they
    run the GENERAL variant of the loop block while the even ones run the LOOSE variant (the reference's infinite bounds):
    both variants, and the first / last workgroup, are inside the sample. Margins: those of
    test_p5f_fp32_assembly_route_against_the_table_oracle (B = 200)."""
    torch = torch_cuda
    import osqp_table
    from robobee3d_amd.batchqp import PlanarP5fMPC
    B = 16384
    mpc = PlanarP5fMPC(B, torch.float32)
    assert mpc.qp.kernel_name == "p5f10+asm"
    rng = np.random.default_rng(20201119)
    pert = rng.uniform(-0.1, 0.1, (2, B)).astype(np.float32)
    mpc.y[0] = torch.as_tensor(pert[0]).cuda()
    mpc.y[3] = torch.as_tensor(pert[1]).cuda()
    st, perm = mpc.st, mpc.qp.s.perm
    # finite input limits on the robots of the odd workgroups (64 robots per workgroup)
    rows = [77 + t for t, j in enumerate(st["var_order"]) if j >= 77]
    assert len(rows) == 10
    odd = torch.as_tensor(((np.arange(B) // 64) % 2) == 1).cuda()
    ulim = 4.0
    for r in rows:
        mpc.l[r][odd] = -ulim
        mpc.u[r][odd] = ulim
    wgs = np.arange(1, 256, 4 + 0)                      # 64 workgroups: 1, 5, 9, ... (odd) -- shifted below to mix parities
    wgs = (wgs + (np.arange(64) % 2)) % 256             # 1, 6, 9, 14, ...: odd and even workgroups alternate
    idx = wgs * 64 + (np.arange(64) * 37) % 64
    idx[0], idx[-1] = 0, B - 1                          # first robot of workgroup 0 (loose), last robot of workgroup 255 (general)
    idx = np.unique(idx)
    n = len(idx)
    assert n >= 60 and {0, 255} <= set((idx // 64).tolist())
    general = ((idx // 64) % 2) == 1
    assert 20 <= int(general.sum()) <= n - 20
    sel = torch.as_tensor(idx).cuda()
    f = lambda t: t[:, sel].cpu().numpy() if t.dim() == 2 else t[sel].cpu().numpy()
    z32 = lambda r: np.zeros((r, n), np.float32)
    x, y, z, E = z32(87), z32(164), z32(164), np.ones((164, n), np.float32)
    flips = clipped = 0
    for ti in range(2, 7):
        unom = 15.0 * np.sin(2 * np.pi * 170 * 0.002 * ti)
        mpc.linearise(unom)
        r = osqp_table.solve(87, 164, st["A_p"], st["A_i"], st["P_cols"], perm, f(mpc.Pv), f(mpc.Av), f(mpc.q),
                             f(mpc.l), f(mpc.u), x, y, z, E, osqp_table.Settings(max_iter=50), dtype=np.float32)
        x, y, z, E = r["x"], r["y"], r["z"], r["E"]
        mpc.qp.solve(mpc.Pv, mpc.Av, mpc.q, mpc.l, mpc.u)
        mpc._p5f_step(1, unom, None)                    # the reference's plant tick: the next getLin sees the moved state
        torch.cuda.synchronize()
        got = {k: f(getattr(mpc.qp, k)).astype(np.float64) for k in ("x", "y", "z", "sol_x", "sol_y", "Eprev")}
        lab = "B=16384 tick %d: " % (ti - 1)
        margin(lab + "iterates x, y, z  |d| / max(1, |ref|)", max(_rel(got[k], r[k]) for k in ("x", "y", "z")), 1.2e-5)
        margin(lab + "sol_x, sol_y  |d| / max(1, |ref|)", max(_rel(got[k], r[k]) for k in ("sol_x", "sol_y")), 6e-5)
        margin(lab + "E (Ruiz row scaling) relative", float(np.max(np.abs(got["Eprev"] / r["E"] - 1))), 1e-6)
        stat = f(mpc.qp.status)
        flips += int(np.count_nonzero(stat != r["status"]))
        assert set(np.unique(mpc.qp.status.cpu().numpy())).issubset({1, 2, -2})
        clipped += int(np.count_nonzero(np.abs(np.abs(r["sol_x"][-10:, general]) - ulim) < 1e-3))
    margin("B=16384: status flips over 5 ticks x %d sampled robots" % n, flips, 3)
    assert clipped > 0                                  # the limits of the general-variant workgroups are active
    # after the first tick every workgroup is on the all-assembly route; the whole batch is finite and solved
    assert bool(torch.isfinite(mpc.y).all()) and bool((mpc.qp.status > 0).all())


def test_blocks_of_a_job_take_the_kernel_form_of_the_whole_job(torch_cuda):
    """results are partition-invariant. 65 536 robots run whole (lane form: one wavefront per SIMD) and as 8
    blocks of 8 192 -- a strong-scaling split over 8 ranks, shard.split_range -- WITHOUT set_step_kernel: each block is
    created with global_batch = 65 536 and so runs the lane form too (on its own 8 192 robots alone the automatic choice
    would be the quad form, equal only up to rounding). torch.equal on state, controller state, outputs and statistics.
    The same for fp64 (4 096 whole = quad form; a 16 384-robot job's block of 4 096 = lane form)."""
    torch = torch_cuda
    from robobee3d_amd import shard
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions_device
    B, K, W = 65536, 6, 8
    whole = BatchUprightMPC(B, torch.float32, plant_mode=1)
    st, ref, _ = hover_initial_conditions_device(B, 20201118, torch.float32)
    whole.set_state(st, ref)
    whole.rollout(K)
    assert whole.kernel_name == "umpc_rollout_asm_kernel"
    for rank in range(W):
        lo, hi = shard.split_range(B, rank, W)
        assert hi - lo == 8192
        blk = BatchUprightMPC(hi - lo, torch.float32, plant_mode=1, global_batch=B)
        s4, r4, _ = hover_initial_conditions_device(hi - lo, 20201118, torch.float32, index_offset=lo)
        blk.set_state(s4, r4)
        blk.rollout(K)
        assert blk.kernel_name == "umpc_rollout_asm_kernel"
        for a, b in ((blk.state, whole.state), (blk.ctrl, whole.ctrl), (blk.out, whole.out), (blk.stats, whole.stats)):
            assert torch.equal(a, b[:, lo:hi])
        assert torch.equal(blk.status, whole.status[lo:hi])
    alone = BatchUprightMPC(8192, torch.float32, plant_mode=1)       # no hint: the block on its own is a small job
    s4, r4, _ = hover_initial_conditions_device(8192, 20201118, torch.float32)
    alone.set_state(s4, r4)
    alone.rollout(1)
    assert alone.kernel_name == "umpc_rollout_asm_quad_kernel"
    with pytest.raises(RuntimeError):
        BatchUprightMPC(8192, torch.float32, global_batch=4096)      # the whole cannot be smaller than its block
    # fp64: the whole job of 16 384 runs the lane form of the assembly ADMM phase; its block of 4 096 must as well
    w64 = BatchUprightMPC(16384, torch.float64, plant_mode=0)
    st, ref, _ = hover_initial_conditions_device(16384, 20201117, torch.float64)
    w64.set_state(st, ref)
    w64.rollout(2)
    b64 = BatchUprightMPC(4096, torch.float64, plant_mode=0, global_batch=16384)
    s4, r4, _ = hover_initial_conditions_device(4096, 20201117, torch.float64, index_offset=8192)
    b64.set_state(s4, r4)
    b64.rollout(2)
    assert b64.kernel_name == w64.kernel_name == "umpc_rollout_kernel<double, LDSF, ASM64>"
    assert torch.equal(b64.state, w64.state[:, 8192:12288]) and torch.equal(b64.out, w64.out[:, 8192:12288])


def test_lane_and_quad_forms_share_their_closed_loop_error_statistics(torch_cuda, oracle_built, margin):
    """replaying the reference's ONE closed-loop log in an all-fp32 harness, the quad form of the step stream ended
    1.9e-4 (relative logMetric) from the reference where the lane form had ended 2.7e-5, and the bound of that test was widened
    on the strength of "rounding only". The two forms differ in the ORDER in which an unknown's triangular-solve updates are
    added up, nothing else: per step both are equally far from the fp64 oracle (tests/test_asm_quad.py). A
    closed loop amplifies per-step rounding differently along every trajectory, so one trajectory says nothing about a form;
    1 024 do. Here: 1 024 random-tilt hover starts, 96 closed-loop steps (the length of the reference's log: 2 400 plant
    substeps), the reference's Euler + expm plant, both forms on the GPU against the fp64 oracle on the host. Asserted: the
    median and the 95th percentile of the relative logMetric error and of the final-state error of the quad form are within a
    factor 2 of the lane form's -- and the two single-trajectory numbers of the log replay (2.7e-5, 1.9e-4) both lie inside
    the spread of EITHER form."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    B, K = 1024, 96
    st, ref = hover_initial_conditions(B, 20201118, np.float32)
    perm = np.array(oracle_perm())
    s_o = st.astype(np.float64)
    ctrl = np.zeros((127, B)); ctrl[124:] = 1
    _, stats_o, _ = oracle_built.batch_rollout(s_o, ctrl, ref.astype(np.float64), K, dtype=np.float64, perm=perm, plant_mode=0,
                                               nthreads=8)
    res = {}
    for form in ("lane", "quad"):
        m = BatchUprightMPC(B, torch.float32, plant_mode=0)
        m.set_step_kernel(form)
        m.set_state(st, ref)
        m.rollout(K)
        torch.cuda.synchronize()
        assert m.kernel_name == ("umpc_rollout_asm_kernel" if form == "lane" else "umpc_rollout_asm_quad_kernel")
        s = m.state.cpu().numpy().astype(np.float64)
        stats = m.stats.cpu().numpy().astype(np.float64)
        res[form] = dict(metric=np.max(np.abs(stats / stats_o - 1), axis=0),          # per robot: the logMetric pair, relative
                         state=np.max(np.abs(s - s_o), axis=0))
    pct = lambda v: (float(np.median(v)), float(np.percentile(v, 95)), float(v.max()))
    for key, unit in (("metric", "relative logMetric error"), ("state", "final-state error")):
        (ml, pl, xl), (mq, pq, xq) = pct(res["lane"][key]), pct(res["quad"][key])
        margin("closed loop K=96, %s: median quad / lane" % unit, mq / ml, 2.0)
        margin("closed loop K=96, %s: median lane / quad" % unit, ml / mq, 2.0)
        margin("closed loop K=96, %s: 95th percentile quad / lane" % unit, pq / pl, 2.0)
        margin("closed loop K=96, %s: 95th percentile lane / quad" % unit, pl / pq, 2.0)
        margin("closed loop K=96, %s: lane median (recorded)" % unit, ml, 1.0)
        margin("closed loop K=96, %s: quad median (recorded)" % unit, mq, 1.0)
        margin("closed loop K=96, %s: lane max (recorded)" % unit, xl, 1.0)
        margin("closed loop K=96, %s: quad max (recorded)" % unit, xq, 1.0)
    # where the log replay's two single-trajectory numbers (2.7e-5 lane, 1.9e-4 quad) sit in each form's distribution
    for form in ("lane", "quad"):
        v = np.sort(res[form]["metric"])
        for x in (2.7e-5, 1.9e-4):
            margin("closed loop K=96: fraction of %s-form trajectories with metric error below %.1e (recorded)" % (form, x),
                   float(np.searchsorted(v, x)) / B, 1.0)


def oracle_perm():
    from robobee3d_amd import _lib
    return _lib.lib().umpcKKTPerm().contents

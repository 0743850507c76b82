"""GPU parity tests (run on the MI355X box: pytest -m gpu). Everything goes
through the C ABI of libumpc_mi355x.so; the oracle is only the checker.

Stated fp32 tolerance of the path (same start state, same ADMM iteration count):
    |d thrust| <= 1e-5, |d moment| <= max(2e-2, 1e-3 |moment|), |d accdes| <= 1e-5
asserted AS STATED against the reference's own converged fixtures (seq_iter50, seq_iter10: GOLDEN_TOL; achieved 8.9e-6 /
9.2e-6 and 2.7e-6 / 3.0e-6 on the MI355X). 3e-5 (TOL_T, TOL_A) is kept only where something other than one converged
step is compared, and each such place says why: iterates cut off after 1 or 2 iterations (not yet contracted: the fp32
reference is itself 1.2e-4 / 1.4e-5 from fp64 there), multi-call sequences (errors of earlier calls feed later ones through
the warm start), and comparisons against the FP64 oracle (the fp32 reference itself is 0.9e-5 / 3.4e-3 / 0.9e-5 from
the fp64 evaluation of the same algorithm on these vectors: tests/test_oracle_golden.py).
fp64 kernel vs fp64 oracle: 1e-9 relative.
"""
import numpy as np
import pytest

from conftest import golden, record_margin

pytestmark = pytest.mark.gpu

TOL_T, TOL_A = 3e-5, 3e-5
# SURVEY 8(c)'s stated single-step bound, asserted on the reference fixtures that hold converged steps
GOLDEN_TOL = {"seq_iter50.npz": 1e-5, "seq_iter10.npz": 1e-5}


def tol_tau(ref):
    return np.maximum(2e-2, 1e-3 * np.abs(ref))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _load_seq_into(mpc, seq, torch, n=None):
    n = n or len(seq["p0"])
    B = mpc.B
    assert B == n
    dt = np.float32 if mpc.dtype == torch.float32 else np.float64
    state = np.zeros((18, B), dt)
    state[0:3] = seq["p0"][:n].T
    state[3:12] = seq["R0"][:n].transpose(2, 1, 0).reshape(9, n)  # column-major
    state[12:18] = seq["dq0"][:n].T
    ref = np.vstack((seq["pdes"][:n].T, seq["dpdes"][:n].T, seq["sdes"][:n].T)).astype(dt)
    ctrl = np.vstack((seq["pre_x"][:n].T, seq["pre_y"][:n].T, seq["pre_z"][:n].T, seq["pre_T0"][:n][None, :],
                      seq["pre_E3"][:n].T)).astype(dt)
    mpc.set_state(state, ref)
    mpc.ctrl.copy_(torch.as_tensor(ctrl))
    mpc.actualT0 = torch.as_tensor(seq["actualT0"][:n].astype(dt)).to(mpc.device)
    return state, ref, ctrl


def test_native_library_is_loaded(torch_cuda):
    from robobee3d_amd import _lib
    L = _lib.lib()
    assert b"umpc_rollout" in L.umpcKernelName(0, 0)
    with open("/proc/self/maps") as f:
        assert "libumpc_mi355x.so" in f.read()


def test_assembly_matches_reference(torch_cuda):
    """l,u,q,Px,Ax of the kernel == the reference C's UprightMPC_t fields (a4,a5)."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC
    seq = golden("seq_iter50.npz")
    mpc = BatchUprightMPC(256, torch.float32)
    _load_seq_into(mpc, seq, torch)
    # the T0 the reference assembled with: actualT0 if >= 0 else the accumulator
    T0 = np.where(seq["actualT0"] >= 0, seq["actualT0"], seq["pre_T0"]).astype(np.float32)
    mpc.ctrl[123].copy_(torch.as_tensor(T0))
    l, u, q, Px, Ax = [t.cpu().numpy().T for t in mpc.assemble()]
    np.testing.assert_allclose(l, seq["l"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(u, seq["u"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(q, seq["q"], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(Px, seq["Px"])
    np.testing.assert_allclose(Ax, seq["Ax"], rtol=1e-6, atol=1e-9)


# After ONE iteration from a cold/perturbed start the iterate is huge (|moment| up to 7.7e3 in the
# fixture) and the fp32 reference is itself 1.2e-4 (thrust) / 1.6e-4 (accdes) from fp64: scale that case.
FIXTURE_SCALE = {"seq_iter1.npz": 15.0}


def _check_outputs(out, seq, n, label, scale=1.0, sel=None, tau_scale=None, tol=None):
    TOL_T, TOL_A = (tol, tol) if tol is not None else (globals()["TOL_T"], globals()["TOL_A"])
    sel = np.arange(n) if sel is None else sel
    uq, ac = out[:3].T[sel], out[3:].T[sel]
    ru, ra = seq["uquad"][:n].astype(np.float64)[sel], seq["accdes"][:n].astype(np.float64)[sel]
    d0 = np.abs(uq[:, 0] - ru[:, 0]).max()
    dt_ = np.abs(uq[:, 1:] - ru[:, 1:])
    da = np.abs(ac - ra).max()
    record_margin(label, "|d thrust|", d0, TOL_T * scale)
    tau_scale = scale if tau_scale is None else tau_scale
    record_margin(label, "|d moment| / max(2e-2, 1e-3|u|)", (dt_ / tol_tau(ru[:, 1:])).max(), tau_scale)
    record_margin(label, "|d accdes|", da, TOL_A * scale)
    assert d0 <= TOL_T * scale, (label, "thrust", d0)
    assert np.all(dt_ <= tol_tau(ru[:, 1:]) * tau_scale), (label, "moment", dt_.max())
    assert da <= TOL_A * scale, (label, "accdes", da)
    return d0, dt_.max(), da


# Measured on the MI355X (round 2, profiles/README.md "parity margins"); every bound below is <= 3x its measured value.
ITERATE_TOL = {"seq_iter50.npz": 1e-3, "seq_iter10.npz": 4e-4, "seq_iter2.npz": 2.5e-5, "seq_iter1.npz": 8e-5}
STATUS_FLIPS = {"seq_iter50.npz": 32, "seq_iter10.npz": 1, "seq_iter2.npz": 1, "seq_iter1.npz": 1}
DUA_RATIO = {"seq_iter50.npz": 5.0, "seq_iter10.npz": 1.5, "seq_iter2.npz": 1.5, "seq_iter1.npz": 1.5}
PRI_REL = {"seq_iter50.npz": 4e-4, "seq_iter10.npz": 4e-6, "seq_iter2.npz": 1e-6, "seq_iter1.npz": 1e-6}
# A status word may differ from the reference's only where the REFERENCE's own deciding residual sits at its own eps
# threshold (tests/status_boundary.py restates compute_pri_tol / compute_dua_tol, auxil.c:262-349, from the fixture's
# sol_x, sol_y, z: it reproduces all 256 reference status words). Factor = max(res / eps, eps / res) of the comparison that
# came out the other way. After 50 fp32 iterations the dual residual is round-off: CPU variants of the same
# algorithm flip at factors up to 3.1 (fp32 canonical oracle) and 7.0 (its fp64 build) -- tests/test_oracle_golden.py.
FLIP_FACTOR = {"seq_iter50.npz": 8.0, "seq_iter10.npz": 1.5, "seq_iter2.npz": 1.5, "seq_iter1.npz": 1.5}


@pytest.mark.parametrize("fname", ["seq_iter50.npz", "seq_iter10.npz", "seq_iter2.npz", "seq_iter1.npz"])
def test_single_step_matches_reference_golden(torch_cuda, structure, fname):
    """Every call of the reference sequences, replayed as one batch: robot k
    starts from the reference's state before call k (warm start x,y,z, T0, E)."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC
    seq = golden(fname)
    n = len(seq["p0"])
    mpc = BatchUprightMPC(n, torch.float32, maxIter=int(seq["maxIter"]))
    _load_seq_into(mpc, seq, torch)
    mpc.update()
    torch.cuda.synchronize()
    out = mpc.out.cpu().numpy().astype(np.float64)
    sc = FIXTURE_SCALE.get(fname, 1.0)
    _check_outputs(out, seq, n, fname, sc, tau_scale=1.0, tol=GOLDEN_TOL.get(fname))
    ctrl = mpc.ctrl.cpu().numpy()
    # iterates: scaled x, y, z after the same number of iterations
    for name, sl in (("x", slice(0, 45)), ("y", slice(45, 84)), ("z", slice(84, 123))):
        ref = seq[name].T
        err = np.abs(ctrl[sl] - ref) / (1e-3 + np.abs(ref).max(axis=0, keepdims=True))
        record_margin(fname, "iterate %s: |d| / (1e-3 + max|ref|)" % name, err.max(), ITERATE_TOL[fname])
        assert err.max() < ITERATE_TOL[fname], (name, err.max())
    np.testing.assert_allclose(ctrl[123], seq["T0"], rtol=0, atol=TOL_T * sc)
    np.testing.assert_allclose(ctrl[124:127].T, seq["E"][:, 36:39], rtol=1e-5)
    status = mpc.status.cpu().numpy()
    # Status: after 50 fp32 iterations the DUAL residual is round-off noise (the fp32
    # reference's dua_res is ~100x the fp64 value of the same iterate, and two fp32
    # evaluation orders differ by a median 40 %: measured with the oracle),
    # so solved / solved-inaccurate / max-iter flips at the check_termination boundary
    # are inherent: 8..23 of 256 between CPU variants of the same algorithm. Require the
    # same family (never an infeasibility / non-convex code) and a bounded flip count.
    mismatch = int(np.sum(status != seq["status"]))
    assert set(np.unique(status)).issubset({1, 2, -2}), np.unique(status)
    record_margin(fname, "status flips of %d" % n, mismatch, STATUS_FLIPS[fname])
    assert mismatch <= STATUS_FLIPS[fname], (mismatch, n)
    # ... and every robot that flipped must sit AT the reference's tolerance boundary: a flip away from it fails
    import status_boundary
    nflip, worst, arg = status_boundary.worst_flip(structure, seq, status)
    assert nflip == mismatch
    record_margin(fname, "worst flip: reference residual vs its eps (factor)", worst, FLIP_FACTOR[fname],
                  "robot %d" % arg if arg >= 0 else "no flip")
    assert worst <= FLIP_FACTOR[fname], (fname, "status flip away from the tolerance boundary", arg, worst)
    info = mpc.info.cpu().numpy()
    rel_pri = np.abs(info[0] - seq["pri_res"]) / seq["pri_res"]
    record_margin(fname, "median rel. error of pri_res", np.median(rel_pri), PRI_REL[fname])
    assert np.median(rel_pri) < PRI_REL[fname], np.median(rel_pri)
    ratio = info[1] / seq["dua_res"]
    med = float(np.median(ratio))
    record_margin(fname, "median dua_res ratio (max of r, 1/r)", max(med, 1 / med), DUA_RATIO[fname])
    assert 1 / DUA_RATIO[fname] < med < DUA_RATIO[fname], med


def test_single_step_fp32_vs_canonical_oracle_and_fp64(torch_cuda, oracle_built, structure):
    """HIP fp32 vs (a) the fp32 oracle in canonical mode with the kernel's own
    elimination order, (b) the fp64 oracle: the kernel is not further from fp64
    truth than 3x what the reference itself is."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC
    from robobee3d_amd import _lib
    seq = golden("seq_iter50.npz")
    n = 256
    mpc = BatchUprightMPC(n, torch.float32)
    _load_seq_into(mpc, seq, torch)
    mpc.update()
    out = mpc.out.cpu().numpy().astype(np.float64)
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    worst = np.zeros(3)
    for dtype, scale in ((np.float32, 1.0), (np.float64, 1.0)):
        o = oracle_built.Oracle(dtype, perm=perm)
        for k in range(n):
            o.set_canonical(True, seq["pre_E3"][k])
            o.set_iterates(seq["pre_x"][k], seq["pre_y"][k], seq["pre_z"][k])
            o.set_T0(float(seq["pre_T0"][k]))
            uq, ac = o.update(seq["p0"][k], seq["R0"][k], seq["dq0"][k], seq["pdes"][k], seq["dpdes"][k],
                              seq["sdes"][k], float(seq["actualT0"][k]))
            assert abs(out[0, k] - uq[0]) <= TOL_T
            assert np.all(np.abs(out[1:3, k] - uq[1:]) <= tol_tau(uq[1:]))
            assert np.all(np.abs(out[3:, k] - ac) <= TOL_A)


def test_fp64_kernel_matches_fp64_oracle(torch_cuda, oracle_built):
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC
    from robobee3d_amd import _lib
    seq = golden("seq_iter50.npz")
    n = 64
    mpc = BatchUprightMPC(n, torch.float64)
    _load_seq_into(mpc, seq, torch, n)
    mpc.update()
    out = mpc.out.cpu().numpy()
    ctrl = mpc.ctrl.cpu().numpy()
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    o = oracle_built.Oracle(np.float64, perm=perm)
    for k in range(n):
        o.set_canonical(True, seq["pre_E3"][k].astype(np.float64))
        o.set_iterates(seq["pre_x"][k], seq["pre_y"][k], seq["pre_z"][k])
        o.set_T0(float(seq["pre_T0"][k]))
        uq, ac = o.update(seq["p0"][k], seq["R0"][k], seq["dq0"][k], seq["pdes"][k], seq["dpdes"][k],
                          seq["sdes"][k], float(seq["actualT0"][k]))
        np.testing.assert_allclose(out[:3, k], uq, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(out[3:, k], ac, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(ctrl[:45, k], o.get("x"), rtol=1e-8, atol=1e-10)
        assert int(mpc.status[k]) == int(o.get("status_val")[0])


def test_reference_boundary_dropin_sequence(torch_cuda):
    """uprightmpc2py.UprightMPC2C through umpcInit/umpcUpdate: one controller
    carried across 24 calls of the reference's sequence from a pristine start
    (state persists inside the library like the reference's global workspace);
    vectors()/matrices() expose the same debug fields."""
    from robobee3d_amd.uprightmpc2py import UprightMPC2C
    seq = golden("seq_iter50.npz")
    upc = UprightMPC2C(5, 9.81e-3, 2, 1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2, np.array([3333., 3333., 1000.]), 50)
    st = golden("structure.npz")
    for k in range(24):
        uq, ac = upc.update(seq["p0"][k], seq["R0"][k], seq["dq0"][k], seq["pdes"][k], seq["dpdes"][k],
                            seq["sdes"][k], float(seq["actualT0"][k]))
        assert uq.shape == (3,) and ac.shape == (6,) and uq.dtype == np.float32
        # errors of earlier calls feed later ones through the warm start: looser than single-step
        assert abs(uq[0] - seq["uquad"][k][0]) <= 1e-4
        assert np.all(np.abs(uq[1:] - seq["uquad"][k][1:]) <= 3 * tol_tau(seq["uquad"][k][1:]))
        assert np.all(np.abs(ac - seq["accdes"][k]) <= 1e-4)
        l, u, q = upc.vectors()
        Px, Ax, Aidx = upc.matrices()
        np.testing.assert_allclose(q, seq["q"][k], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(Px, seq["Px"][k])
        np.testing.assert_array_equal(Aidx, st["Ax_idx"])
        np.testing.assert_allclose(Ax, seq["Ax"][k], rtol=1e-5, atol=5e-4)  # Ax[0:3] = dt*T0, T0 accumulates (1e-4 band)
    # six-argument call of the reference harness (template/uprightmpc2.py:139)
    uq, ac = upc.update(np.zeros(3), np.eye(3), np.zeros(6), np.zeros(3), np.zeros(3), [0, 0, 1])
    assert np.all(np.isfinite(uq))


def test_plant_kernel_matches_reference_python(torch_cuda, oracle_built):
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC
    g = golden("plant.npz")
    n = len(g["p"])
    for dt_sub in (0.2, 0.1, 1.0, 5.0):
        sel = np.nonzero(g["dt"] == dt_sub)[0]
        for dtype, tol in ((torch.float64, 1e-11), (torch.float32, 2e-5)):
            mpc = BatchUprightMPC(len(sel), dtype, dtsim=dt_sub)
            state = np.zeros((18, len(sel)))
            state[0:3] = g["p"][sel].T
            state[3:12] = g["R"][sel].transpose(2, 1, 0).reshape(9, -1)
            state[12:18] = g["dq"][sel].T
            mpc.set_state(state)
            mpc.plant(g["u"][sel].T.copy(), nsub=1)
            s = mpc.state.cpu().numpy().astype(np.float64)
            sc = 1.0 if dtype == torch.float64 else 10.0
            np.testing.assert_allclose(s[0:3].T, g["p2"][sel], rtol=tol, atol=tol * sc)
            np.testing.assert_allclose(s[3:12].reshape(3, 3, -1).transpose(2, 1, 0), g["R2"][sel], rtol=0, atol=tol)
            np.testing.assert_allclose(s[12:18].T, g["dq2"][sel], rtol=tol * 10, atol=tol)


@pytest.mark.parametrize("plant_mode", [0, 1])
def test_closed_loop_rollout_matches_oracle(torch_cuda, oracle_built, plant_mode):
    """K closed-loop steps (QP + 25 substeps) for 512 random-tilt robots: HIP
    fp64 vs the fp64 oracle (tight), HIP fp32 vs fp64 oracle (fp32 band)."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    from robobee3d_amd import _lib
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    B, K = 512, 12
    st64, ref64 = hover_initial_conditions(B, 20201118, np.float64)
    ctrl0 = np.zeros((127, B)); ctrl0[124:] = 1
    s_o, c_o = st64.copy(), ctrl0.copy()
    out_o, stats_o, status_o = oracle_built.batch_rollout(s_o, c_o, ref64, K, dtype=np.float64, perm=perm,
                                                          plant_mode=plant_mode)
    m64 = BatchUprightMPC(B, torch.float64, plant_mode=plant_mode)
    m64.set_state(st64, ref64)
    m64.rollout(K)
    np.testing.assert_allclose(m64.state.cpu().numpy(), s_o, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(m64.out.cpu().numpy(), out_o, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(m64.stats.cpu().numpy(), stats_o, rtol=1e-7)
    m32 = BatchUprightMPC(B, torch.float32, plant_mode=plant_mode)
    m32.set_state(st64.astype(np.float32), ref64.astype(np.float32))
    for _ in range(K):  # also exercises launch-per-step == K-in-one-launch
        m32.rollout(1)
    s32 = m32.state.cpu().numpy().astype(np.float64)
    # fp32 band after K = 12 closed-loop steps: the fp32 CPU oracle itself is 4.7e-4..6.1e-4 mm /
    # 0.9e-4..1.1e-4 (attitude, velocity) away from the fp64 oracle on this workload (measured with
    # both elimination orders); tolerance = 3x that band.
    lab = "closed_loop_rollout[plant_mode=%d] K=12" % plant_mode
    record_margin(lab, "fp32 vs fp64 oracle |dp| mm", np.abs(s32[0:3] - s_o[0:3]).max(), 1.5e-3)
    record_margin(lab, "fp32 vs fp64 oracle |dR|,|ddq|", np.abs(s32[3:] - s_o[3:]).max(), 3e-4)
    record_margin(lab, "fp64 vs fp64 oracle state", np.abs(m64.state.cpu().numpy() - s_o).max(), 1e-7)
    np.testing.assert_allclose(s32[0:3], s_o[0:3], rtol=0, atol=1.5e-3)
    np.testing.assert_allclose(s32[3:], s_o[3:], rtol=0, atol=3e-4)
    np.testing.assert_allclose(m32.stats.cpu().numpy(), stats_o, rtol=1e-3)


def test_full_size_hover_properties(torch_cuda):
    """BASELINE config 3 size (B = 65536, fp32, closed loop): size-independent
    properties -- every robot converges to hover at the origin, R stays
    orthonormal, thrust stays inside [0, Tmax], and the result is invariant to
    how the batch is partitioned (a sharded run == the single-GPU run)."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    B, K = 65536, 100
    st, ref = hover_initial_conditions(B, 20201118)
    mpc = BatchUprightMPC(B, torch.float32)
    mpc.set_state(st, ref)
    mpc.rollout(K)
    s = mpc.state.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(s))
    assert np.linalg.norm(s[0:3], axis=0).max() < 0.05          # mm
    assert np.abs(s[11] - 1).max() < 1e-3 and np.abs(s[9:11]).max() < 2e-2   # s = R e3 -> e3
    R = s[3:12].reshape(3, 3, B)  # [col, row, b]
    G = np.einsum("crb,drb->cdb", R, R)
    assert np.abs(G - np.eye(3)[:, :, None]).max() < 1e-3
    out = mpc.out.cpu().numpy()
    assert out[0].min() >= -1e-6 and out[0].max() <= 2 * 9.81e-3 + 1e-6
    assert np.abs(out[0] - 9.81e-3).max() < 2e-4                # hover thrust = g
    # partition invariance: robots [1000, 1000+4096) alone
    off, n = 1000, 4096
    st2, ref2 = hover_initial_conditions(n, 20201118, index_offset=off)
    np.testing.assert_array_equal(st2, st[:, off:off + n])
    sub = BatchUprightMPC(n, torch.float32, global_batch=B)   # a block of the B-robot job: it runs the form the whole runs
    sub.set_state(st2, ref2)
    sub.rollout(K)
    np.testing.assert_array_equal(sub.state.cpu().numpy(), mpc.state[:, off:off + n].cpu().numpy())
    np.testing.assert_array_equal(sub.stats.cpu().numpy(), mpc.stats[:, off:off + n].cpu().numpy())
    assert sub.kernel_name == mpc.kernel_name == "umpc_rollout_asm_kernel"


def test_ragged_and_tiny_batches(torch_cuda):
    """B = 1 and B not a multiple of the 64-lane wavefront."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    st, ref = hover_initial_conditions(131, 7)
    big = BatchUprightMPC(131, torch.float32)
    big.set_state(st, ref)
    big.rollout(3)
    for n in (1, 63, 65):
        m = BatchUprightMPC(n, torch.float32)
        m.set_state(st[:, :n].copy(), ref[:, :n].copy())
        m.rollout(3)
        np.testing.assert_array_equal(m.state.cpu().numpy(), big.state[:, :n].cpu().numpy())
        np.testing.assert_array_equal(m.out.cpu().numpy(), big.out[:, :n].cpu().numpy())


def test_monte_carlo_inertia_and_mass(torch_cuda, oracle_built):
    """Config 5 inputs: per-robot Ib (controller + plant) and thrust gain."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    from robobee3d_amd import _lib
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    B, K = 256, 6
    rng = np.random.default_rng(20201120)
    st, ref = hover_initial_conditions(B, 20201120, np.float64)
    Ib = np.array([3333., 3333., 1000.])[:, None] * (1 + rng.uniform(-0.2, 0.2, (3, B)))
    gain = 1 + rng.uniform(-0.2, 0.2, B)
    ctrl = np.zeros((127, B)); ctrl[124:] = 1
    s_o = st.copy()
    out_o, stats_o, _ = oracle_built.batch_rollout(s_o, ctrl, ref, K, dtype=np.float64, perm=perm, Ib=Ib, gain=gain)
    m = BatchUprightMPC(B, torch.float64)
    m.set_state(st, ref)
    m.Ib = torch.as_tensor(Ib).cuda()
    m.gain = torch.as_tensor(gain).cuda()
    m.rollout(K)
    np.testing.assert_allclose(m.state.cpu().numpy(), s_o, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(m.out.cpu().numpy(), out_o, rtol=1e-6, atol=1e-8)


def test_simulink_entry_points_equal_the_object_api(torch_cuda):
    """umpcS (uprightmpc2.c:275-284) and wlconS (funapprox.c:171-176), the lazily initialised singletons behind the
    Simulink S-functions (legacy_code_gen.m:6), give the same sequences as umpcInit/umpcUpdate and
    wlConInit/wlConUpdate."""
    import ctypes as C
    from robobee3d_amd import _lib
    from robobee3d_amd.uprightmpc2py import UprightMPC2C, WLCon
    L = _lib.lib()
    seq = golden("seq_iter50.npz")
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    prm = (5.0, 9.81e-3, 2.0, 1e1, 1e3, 1.0, 5.0, 1e3, 2e3, 1e-1, 1e-2)
    Ib = f32([3333.0, 3333.0, 1000.0])
    obj = UprightMPC2C(*prm, Ib, 50)
    for k in range(6):
        args = [f32(seq[n][k]) for n in ("p0",)] + [f32(seq["R0"][k].T.ravel())] + \
               [f32(seq[n][k]) for n in ("dq0", "pdes", "dpdes", "sdes")]
        u1, a1 = obj.update(seq["p0"][k], seq["R0"][k], seq["dq0"][k], seq["pdes"][k], seq["dpdes"][k], seq["sdes"][k],
                            float(seq["actualT0"][k]))
        u2, a2 = np.zeros(3, np.float32), np.zeros(6, np.float32)
        L.umpcS(fp(u2), fp(a2), *[fp(a) for a in args], *[C.c_float(v) for v in prm], fp(Ib), C.c_int(50),
                C.c_float(float(seq["actualT0"][k])))
        assert np.array_equal(u1, u2) and np.array_equal(a1, a2), k
    g = golden("wl_step.npz")
    wl = WLCon(g["u0"], g["umin"], g["umax"], g["dumax"], g["Qw"], float(g["controlRate"]), g["popts"])
    for k in range(8):
        u1, w1 = wl.update(g["h0"][k], g["pdotdes"][k])
        u2, w2 = np.zeros(4, np.float32), np.zeros(6, np.float32)
        L.wlconS(fp(u2), fp(w2), fp(f32(g["u0"])), fp(f32(g["umin"])), fp(f32(g["umax"])), fp(f32(g["dumax"])),
                 fp(f32(g["Qw"])), C.c_float(float(g["controlRate"])), fp(f32(g["popts"])), fp(f32(g["h0"][k])),
                 fp(f32(g["pdotdes"][k])))
        assert np.array_equal(u1, u2) and np.array_equal(w1, w2), k


def test_config2_fp64_full_run_against_oracle_sample(torch_cuda, oracle_built):
    """BASELINE configs[1] at full size: B = 4096, fp64, seed 20201117, K = 200 closed-loop
    steps in one launch. Every 128th robot is replayed on the fp64 oracle (same elimination order): states agree to
    1e-6 after 5 000 plant substeps; the whole batch hovers."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    from robobee3d_amd import _lib
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    B, K = 4096, 200
    st, ref = hover_initial_conditions(B, 20201117, np.float64)
    mpc = BatchUprightMPC(B, torch.float64)
    mpc.set_state(st, ref)
    mpc.rollout(K)
    s = mpc.state.cpu().numpy()
    assert np.isfinite(s).all() and np.linalg.norm(s[0:3], axis=0).max() < 0.05
    idx = np.arange(0, B, 128)
    s_o = np.ascontiguousarray(st[:, idx])
    ctrl = np.zeros((127, len(idx))); ctrl[124:] = 1
    out_o, stats_o, _ = oracle_built.batch_rollout(s_o, ctrl, np.ascontiguousarray(ref[:, idx]), K, dtype=np.float64, perm=perm)
    np.testing.assert_allclose(s[:, idx], s_o, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(mpc.out.cpu().numpy()[:, idx], out_o, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(mpc.stats.cpu().numpy()[:, idx], stats_o, rtol=1e-7)


def test_headline_configuration_against_oracle(torch_cuda, oracle_built):
    """EXACTLY what bench.py times (BASELINE configs[2]): B = 65 536, fp32, RK4 plant (plant_mode = 1), K closed-loop
    steps in ONE launch -- a 20-step launch (the driver's --steps 20 shape) followed by a 180-step launch (wave skew
    on, as in the default 500-step bench launch). Every 512th robot is replayed on the fp64 oracle and on the fp32
    oracle (same elimination order, same RK4) from the same start.

    SURVEY 8(c) states the closed-loop band as 1e-3 mm / 1e-4 after K = 200 steps for hover: asserted below at
    K = 200. At K = 20 the robots are still in the 0.5 rad recovery transient, where round-off differences between
    two fp32 evaluation orders of the same 50-iteration step are amplified by the closed loop: there the bound is
    3x the band the fp32 CPU oracle itself is away from the fp64 oracle, measured in this test."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    from robobee3d_amd import _lib
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    B = 65536
    st, ref = hover_initial_conditions(B, 20201118, np.float32)
    mpc = BatchUprightMPC(B, torch.float32, plant_mode=1)
    mpc.set_state(st, ref)
    idx = np.arange(0, B, 512)
    n = len(idx)

    def oracle_run(dtype, K, s, c):
        out, stats, status = oracle_built.batch_rollout(s, c, np.ascontiguousarray(ref[:, idx]).astype(dtype), K,
                                                        dtype=dtype, perm=perm, plant_mode=1)
        return out

    s64 = np.ascontiguousarray(st[:, idx]).astype(np.float64); c64 = np.zeros((127, n)); c64[124:] = 1
    s32 = np.ascontiguousarray(st[:, idx]).astype(np.float32); c32 = np.zeros((127, n), np.float32); c32[124:] = 1
    for K, label in ((20, "K=20 (one 20-step launch)"), (180, "K=200 (+ one 180-step launch, skew on)")):
        mpc.rollout(K)
        o64 = oracle_run(np.float64, K, s64, c64)
        o32 = oracle_run(np.float32, K, s32, c32)
        g = mpc.state.cpu().numpy().astype(np.float64)[:, idx]
        out = mpc.out.cpu().numpy().astype(np.float64)[:, idx]
        assert np.isfinite(g).all()
        dp, ds = np.abs(g[0:3] - s64[0:3]).max(), np.abs(g[3:] - s64[3:]).max()
        bp, bs = np.abs(s32[0:3] - s64[0:3]).max(), np.abs(s32[3:] - s64[3:]).max()   # fp32 oracle vs fp64 oracle
        lab = "headline B=65536 fp32 RK4 " + label
        if K == 20:
            record_margin(lab, "fp32 ORACLE vs fp64 oracle |dp| mm (the band)", bp, bp)
            record_margin(lab, "fp32 ORACLE vs fp64 oracle |dR|,|ddq| (the band)", bs, bs)
            tol_p, tol_s = max(3 * bp, 1e-3), max(3 * bs, 1e-4)
        else:
            tol_p, tol_s = 1e-3, 1e-4
        record_margin(lab, "HIP vs fp64 oracle |dp| mm", dp, tol_p)
        record_margin(lab, "HIP vs fp64 oracle |dR|,|ddq|", ds, tol_s)
        record_margin(lab, "HIP vs fp64 oracle |d thrust|", np.abs(out[0] - o64[0]).max(), TOL_T)
        record_margin(lab, "HIP vs fp64 oracle |d moment| / max(2e-2,1e-3|u|)",
                      (np.abs(out[1:3] - o64[1:3]) / tol_tau(o64[1:3])).max(), 1.0)
        record_margin(lab, "HIP vs fp64 oracle |d accdes|", np.abs(out[3:] - o64[3:]).max(), TOL_A)
        assert dp <= tol_p and ds <= tol_s, (label, dp, ds, tol_p, tol_s)
        assert np.abs(out[0] - o64[0]).max() <= TOL_T
        assert np.all(np.abs(out[1:3] - o64[1:3]) <= tol_tau(o64[1:3]))
        assert np.abs(out[3:] - o64[3:]).max() <= TOL_A
    # the whole batch hovers
    s = mpc.state.cpu().numpy().astype(np.float64)
    assert np.isfinite(s).all() and np.linalg.norm(s[0:3], axis=0).max() < 0.05


def test_nan_and_cold_start_branch_matches_reference(torch_cuda):
    """the `!has_solution` branch (OSQP_NAN = the NUMBER 2143289344 stored into the solution, iterates
    cold-started; auxil.c:539-564, constants.h:96) as the reference itself takes it -- tests/golden/nan_branch.npz,
    generated from the compiled reference: calls 4 and 9 see a state of 1e33 (residual > OSQP_INFTY -> status -7),
    the calls after them restart from zero iterates with actualT0 overriding the thrust accumulator."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC
    seq = golden("nan_branch.npz")
    n = len(seq["p0"])
    mpc = BatchUprightMPC(n, torch.float32)
    _load_seq_into(mpc, seq, torch)
    mpc.update()
    out = mpc.out.cpu().numpy().astype(np.float64)
    ctrl = mpc.ctrl.cpu().numpy()
    status = mpc.status.cpu().numpy()
    bad = np.nonzero(seq["status"] == -7)[0]
    assert list(bad) == [4, 9]
    np.testing.assert_array_equal(status[bad], -7)
    nanv = np.float32(2143289344.0)
    for k in bad:
        np.testing.assert_array_equal(out[1:3, k], nanv)                           # moments: the number itself
        np.testing.assert_array_equal(out[:3, k].astype(np.float32), seq["uquad"][k])
        np.testing.assert_allclose(out[3:, k], seq["accdes"][k], rtol=1e-6)        # (OSQP_NAN - dq0) / dt
        np.testing.assert_array_equal(ctrl[:123, k], 0)                            # cold start (auxil.c:563)
        np.testing.assert_array_equal(ctrl[123, k], seq["T0"][k])                  # T0 += OSQP_NAN, as the reference
    good = np.setdiff1d(np.arange(n), bad)
    assert set(np.unique(status[good])).issubset({1, 2, -2})
    _check_outputs(out, seq, n, "nan_branch.npz (other calls)", sel=good)


def test_nonpositive_weights_are_rejected(torch_cuda):
    """The Ruiz scaling D is recovered from the equilibrated diagonal of P, which needs every weight > 0: a zero
    weight is refused with an error instead of a silent NaN (include/umpc_mi355x.h)."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC
    with pytest.raises(RuntimeError, match="weights must be > 0"):
        BatchUprightMPC(64, torch.float32, wpr=0.0)
    m = BatchUprightMPC(64, torch.float32)
    w = torch.ones((8, 64), device="cuda")
    w[7, 13] = 0.0
    with pytest.raises(RuntimeError, match="weights must be > 0"):
        m.set_weights(w)
    w[7, 13] = 1e-2
    m.set_weights(w)


def test_dropin_survives_a_copied_pod_and_honours_T0_edits(torch_cuda):
    """The reference's pybind class holds UprightMPC_t by value and the POD's T0 is the accumulator of record: a
    host that copies the struct, or edits T0 between calls, keeps working (the controller id travels in the POD)."""
    import ctypes as C
    from robobee3d_amd import _lib
    L = _lib.lib()
    seq = golden("seq_iter50.npz")
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    Ib = f32([3333.0, 3333.0, 1000.0])
    prm = (5.0, 9.81e-3, 2.0, 1e1, 1e3, 1.0, 5.0, 1e3, 2e3, 1e-1, 1e-2)

    def call(up, k, aT0):
        uq, ac = np.zeros(3, np.float32), np.zeros(6, np.float32)
        args = [f32(seq["p0"][k]), f32(seq["R0"][k].T.ravel()), f32(seq["dq0"][k]), f32(seq["pdes"][k]),
                f32(seq["dpdes"][k]), f32(seq["sdes"][k])]
        rc = L.umpcUpdate(C.byref(up), fp(uq), fp(ac), *[fp(a) for a in args], C.c_float(aT0))
        assert rc == 0
        return uq, ac
    a, b = _lib.UprightMPC_t(), _lib.UprightMPC_t()
    for up in (a, b):
        L.umpcInit(C.byref(up), *[C.c_float(v) for v in prm], fp(Ib), C.c_int(50))
    ra = [call(a, k, -1.0) for k in range(3)]
    moved = _lib.UprightMPC_t()
    for k in range(3):                       # b is copied to a new address before every call
        C.memmove(C.byref(moved), C.byref(b), C.sizeof(b))
        rb = call(moved, k, -1.0)
        C.memmove(C.byref(b), C.byref(moved), C.sizeof(b))
        assert np.array_equal(ra[k][0], rb[0]) and np.array_equal(ra[k][1], rb[1]), k
    # editing T0 in the POD == passing actualT0
    a.T0 = 0.0123
    u1, _ = call(a, 3, -1.0)
    u2, _ = call(b, 3, 0.0123)
    assert np.array_equal(u1, u2)
    L.umpcRelease(C.byref(a)); L.umpcRelease(C.byref(b))


def test_assembly_kernel_and_cpp_kernel_agree(torch_cuda):
    """The all-assembly fp32 step kernel (asmstep.py) against the C++ kernel around the assembly ADMM loop (umpc_step.h)
    on the same inputs: the iterations are the same instruction stream and phase A mirrors the C++ operation order, so
    iterates agree to round-off; statuses come from differently associated residual norms (unscaled in the assembly
    kernel) and may flip at a tolerance boundary."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    B, K = 4096, 10
    st, ref = hover_initial_conditions(B, 11, np.float32)
    res = {}
    for mode in ("auto", "cpp"):
        m = BatchUprightMPC(B, torch.float32, plant_mode=1)
        m.set_step_kernel(mode)
        m.set_state(st, ref)
        m.rollout(K)
        res[mode] = [t.cpu().numpy().astype(np.float64) for t in (m.state, m.out, m.ctrl, m.stats)] + [m.status.cpu().numpy()]
    a, c = res["auto"], res["cpp"]
    # two fp32 evaluation orders of the same closed loop: the K = 12 band of test_closed_loop_rollout_matches_oracle
    lab = "assembly vs C++ step kernel (B=4096, K=10)"
    dp, ds = np.abs(a[0][0:3] - c[0][0:3]).max(), np.abs(a[0][3:] - c[0][3:]).max()
    record_margin(lab, "|dp| mm", dp, 1.5e-3)
    record_margin(lab, "|dR|,|ddq|", ds, 3e-4)
    assert dp <= 1.5e-3 and ds <= 3e-4
    for name, k in (("out", 1), ("ctrl", 2), ("stats", 3)):
        d = np.abs(a[k] - c[k]) / (1e-3 + np.abs(c[k]).max(axis=1, keepdims=True))
        record_margin(lab, name + ": |d| / (1e-3 + max|row|)", d.max(), 2e-2)
        assert d.max() <= 2e-2, (name, d.max())
    flips = int(np.count_nonzero(a[4] != c[4]))
    record_margin("assembly vs C++ step kernel (B=4096, K=10)", "status flips of 4096", flips, 400)
    assert flips <= 400 and set(np.unique(a[4])).issubset({1, 2, -2})


def test_fp64_assembly_loop_and_cpp_loop_agree(torch_cuda, margin):
    """BASELINE config 2's kernel: the fp64 step with its ADMM phase as generated assembly (asmgen64.py, the default for
    batches of at most 256 waves) against the same step with the C++ loop (`set_step_kernel("cpp")`), closed loop, for
    the iteration counts that take different paths through the program (1: the first iteration only; 2: first + last;
    3: one pass of the loop; 50) and a ragged batch (exec-masked lanes in the last wave). Same operations; the backward
    solve associates each column in the opposite order and the dynamics rows use delta_y = alpha (nu - y)."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    for B, iters, K in ((100, 1, 3), (100, 2, 3), (100, 3, 3), (4096, 50, 6)):
        st, ref = hover_initial_conditions(B, 5, np.float64)
        res = {}
        for mode in ("auto", "cpp"):
            m = BatchUprightMPC(B, torch.float64, plant_mode=0, maxIter=iters)
            m.set_step_kernel(mode)
            m.set_state(st, ref)
            m.rollout(K)
            res[mode] = [t.cpu().numpy() for t in (m.state, m.out, m.ctrl)] + [m.status.cpu().numpy()]
        a, c = res["auto"], res["cpp"]
        lab = "fp64 assembly loop vs C++ loop B=%d maxIter=%d K=%d: " % (B, iters, K)
        margin(lab + "state", np.abs(a[0] - c[0]).max(), 1e-10)
        margin(lab + "outputs", np.abs(a[1] - c[1]).max(), 1e-10)
        margin(lab + "iterates / (1 + max|row|)", (np.abs(a[2] - c[2]) / (1 + np.abs(c[2]).max(axis=1, keepdims=True))).max(), 1e-10)
        assert np.array_equal(a[3], c[3])


def test_assembly_kernel_ragged_batches_and_monte_carlo(torch_cuda, oracle_built):
    """The all-assembly kernel with exec-masked lanes (B = 1, 63, 65, 131: the last wave is partial) and with
    per-robot inertia / thrust gain (config 5 inputs) in fp32, RK4 plant: partial batches are bit-identical slices of
    a larger one; the Monte-Carlo rollout matches the fp64 oracle within the fp32 closed-loop band."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions, monte_carlo_draws
    from robobee3d_amd import _lib
    assert b"asm" in _lib.lib().umpcKernelName(0, 1)
    st, ref = hover_initial_conditions(131, 7)
    big = BatchUprightMPC(131, torch.float32, plant_mode=1)
    big.set_state(st, ref)
    big.rollout(3)
    for n in (1, 63, 65):
        m = BatchUprightMPC(n, torch.float32, plant_mode=1)
        m.set_state(st[:, :n].copy(), ref[:, :n].copy())
        m.rollout(3)
        np.testing.assert_array_equal(m.state.cpu().numpy(), big.state[:, :n].cpu().numpy())
        np.testing.assert_array_equal(m.out.cpu().numpy(), big.out[:, :n].cpu().numpy())
        np.testing.assert_array_equal(m.status.cpu().numpy(), big.status[:n].cpu().numpy())
    # K launches of one step == one launch of K steps
    one = BatchUprightMPC(131, torch.float32, plant_mode=1)
    one.set_state(st, ref)
    for _ in range(3):
        one.rollout(1)
    np.testing.assert_array_equal(one.state.cpu().numpy(), big.state.cpu().numpy())
    np.testing.assert_array_equal(one.stats.cpu().numpy(), big.stats.cpu().numpy())
    # config 5 inputs
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    B, K = 320, 8
    st, ref = hover_initial_conditions(B, 20201120, np.float64)
    Ib, gain = monte_carlo_draws(B, 20201120, np.float64)
    ctrl = np.zeros((127, B)); ctrl[124:] = 1
    s_o = st.copy()
    out_o, stats_o, _ = oracle_built.batch_rollout(s_o, ctrl, ref, K, dtype=np.float64, perm=perm, Ib=Ib, gain=gain, plant_mode=1)
    m = BatchUprightMPC(B, torch.float32, plant_mode=1)
    m.set_state(st.astype(np.float32), ref.astype(np.float32))
    m.Ib = torch.as_tensor(Ib.astype(np.float32)).cuda()
    m.gain = torch.as_tensor(gain.astype(np.float32)).cuda()
    m.rollout(K)
    s = m.state.cpu().numpy().astype(np.float64)
    dp, ds = np.abs(s[0:3] - s_o[0:3]).max(), np.abs(s[3:] - s_o[3:]).max()
    record_margin("assembly kernel, config-5 inputs (B=320, K=8) vs fp64 oracle", "|dp| mm", dp, 1.5e-3)
    record_margin("assembly kernel, config-5 inputs (B=320, K=8) vs fp64 oracle", "|dR|,|ddq|", ds, 3e-4)
    assert dp <= 1.5e-3 and ds <= 3e-4
    np.testing.assert_allclose(m.stats.cpu().numpy(), stats_o, rtol=1e-3)


def test_bench_rccl_path_rehearsal_one_rank(torch_cuda):
    """The multi-GPU code of bench.py on real hardware as far as a one-GPU box allows: torch.distributed.run with ONE
    rank and UMPC_FORCE_DIST=1 initialises RCCL ("nccl"), runs the barriers, the max-over-ranks all_reduce and the
    end-of-run all_gather of the per-robot statistics on device tensors, and prints the contract's JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from robobee3d_amd import shard
    env = dict(os.environ, UMPC_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(shard.free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--batch", "4096", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["collectives"] == {"backend": "nccl", "world_size": 1, "gathered_robots": 4096}
    assert line["n_gpus"] == 1 and line["value"] > 1e6 and line["check"]["nonfinite_state_values"] == 0


def test_reference_import_line_and_calls_in_a_fresh_process(torch_cuda):
    """VERDICT r3 item 7, on the GPU: a fresh interpreter with only the repository root on sys.path runs the reference's
    import lines (template/template_controllers.py:5, robobee_test_controllers.py:9), constructs the controller exactly as
    createMPC does (template_controllers.py:279), makes the seven-argument call of robobee_test_controllers.py:138 on the
    first call of the reference fixture (result inside the single-step band) and then the SIX-argument call of the
    reference's harness (template/uprightmpc2.py:139: `mdl.update(p, Rb, dq, pdes, dpdes, sdes)`, no actualT0)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys
sys.path.append(%r)
import numpy as np
from uprightmpc2py import UprightMPC2C # C version
from uprightmpc2py import UprightMPC2C, WLCon
dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom = 5, 9.81e-3, 2, 1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2
Ib = np.diag(np.diag([3333, 3333, 1000]))
cver = UprightMPC2C(dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, Ib, 50)
seq = np.load(%r)
u, acc = cver.update(seq["p0"][0], seq["R0"][0], seq["dq0"][0], seq["pdes"][0], seq["dpdes"][0], seq["sdes"][0], seq["actualT0"][0])
print(" ".join("%%.9g" %% v for v in list(u) + list(acc)))
u, acc = cver.update(seq["p0"][1], seq["R0"][1], seq["dq0"][1], seq["pdes"][1], seq["dpdes"][1], seq["sdes"][1])
print(" ".join("%%.9g" %% v for v in list(u) + list(acc)))
""" % (root, os.path.join(root, "tests", "golden", "seq_iter50.npz"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/",
                       env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()[-2:]
    seq = golden("seq_iter50.npz")
    assert np.abs(seq["pre_x"][0]).max() == 0 and seq["pre_T0"][0] == 0      # call 0 of the fixture starts pristine
    for k, ln in enumerate(lines):
        vals = np.array([float(t) for t in ln.split()])
        assert vals.shape == (9,) and np.all(np.isfinite(vals))
        if k == 0 or float(seq["actualT0"][1]) < 0:     # the six-argument call keeps the accumulator, as fixture call 1 does
            sc = 1.0 if k == 0 else 3.0
            assert abs(vals[0] - seq["uquad"][k][0]) <= TOL_T * sc
            assert np.all(np.abs(vals[1:3] - seq["uquad"][k][1:]) <= sc * tol_tau(seq["uquad"][k][1:]))
            assert np.abs(vals[3:] - seq["accdes"][k]).max() <= TOL_A * sc


def test_quad_form_of_the_stream_on_the_gpu(torch_cuda, oracle_built, margin):
    """Round 4: the one-robot-per-lane-quad form of the all-assembly stream (asmquad.py), the form
    batches of <= 16 384 robots and the B = 1 drop-in take. (a) it is what "auto" dispatches there; (b) against the lane
    form on the same 4 096 robots over 6 closed-loop steps: equal up to the rounding of a different summation order,
    every status word equal or both at the tolerance boundary family; (c) robot i computes in a ragged batch (37
    robots: a wave with 5 of its 16 quads live) and alone (B = 1) bit for bit what it computes among 4 096 -- quads are
    independent, dead quads stay dead through the EXEC switches of the entry transposition; (d) the quad form at the
    headline size equals its own small-batch run bit for bit too (forced: throughput there is the lane form's job);
    (e) against the fp64 oracle on a sample, the quad form is inside the closed-loop fp32 band."""
    torch = torch_cuda
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    B, K = 4096, 6
    st, ref = hover_initial_conditions(B, 20201118)
    res = {}
    for form in ("auto", "lane", "quad"):
        m = BatchUprightMPC(B, torch.float32, plant_mode=1)
        m.set_step_kernel(form)
        m.set_state(st, ref)
        m.rollout(K // 2)
        m.rollout(K - K // 2)
        assert m.kernel_name == ("umpc_rollout_asm_kernel" if form == "lane" else "umpc_rollout_asm_quad_kernel")
        res[form] = [t.cpu().numpy().astype(np.float64) for t in (m.state, m.out, m.stats, m.ctrl)] + [m.status.cpu().numpy()]
    for k in range(5):
        assert np.array_equal(res["auto"][k], res["quad"][k])
    a, c = res["quad"], res["lane"]
    lab = "quad vs lane form, B = 4096, K = 6: "
    margin(lab + "|dp| mm", float(np.abs(a[0][0:3] - c[0][0:3]).max()), 5e-4)
    margin(lab + "|dR|, |ddq|", float(np.abs(a[0][3:] - c[0][3:]).max()), 6e-5)
    margin(lab + "|d thrust|", float(np.abs(a[1][0] - c[1][0]).max()), 1e-5)
    margin(lab + "|d moment| / max(2e-2, 1e-3|u|)", float((np.abs(a[1][1:3] - c[1][1:3]) / np.maximum(2e-2, 1e-3 * np.abs(c[1][1:3]))).max()), 0.6)
    margin(lab + "stats relative", float(np.max(np.abs(a[2] - c[2]) / (1e-6 + np.abs(c[2])))), 1e-3)
    margin(lab + "status words that differ (of 4096)", float(np.sum(a[4] != c[4])), 64)
    assert set(np.unique(a[4])).issubset({1, 2, -2})
    # (c) ragged and single
    for n, off in ((37, 1000), (1, 4095), (16, 0), (17, 2048)):
        sub = BatchUprightMPC(n, torch.float32, plant_mode=1)
        sub.set_state(st[:, off:off + n], ref[:, off:off + n])
        sub.rollout(K // 2)
        sub.rollout(K - K // 2)
        assert sub.kernel_name == "umpc_rollout_asm_quad_kernel"
        assert np.array_equal(sub.state.cpu().numpy().astype(np.float64), a[0][:, off:off + n]), (n, off)
        assert np.array_equal(sub.out.cpu().numpy().astype(np.float64), a[1][:, off:off + n])
        assert np.array_equal(sub.ctrl.cpu().numpy().astype(np.float64), a[3][:, off:off + n])
        assert np.array_equal(sub.status.cpu().numpy(), a[4][off:off + n])
    # (d) forced quad form at the headline size
    Bb = 65536
    stb, refb = hover_initial_conditions(Bb, 20201118)
    big = BatchUprightMPC(Bb, torch.float32, plant_mode=1)
    big.set_step_kernel("quad")
    big.set_state(stb, refb)
    big.rollout(K // 2)
    big.rollout(K - K // 2)
    assert big.kernel_name == "umpc_rollout_asm_quad_kernel"
    assert np.array_equal(big.state[:, :B].cpu().numpy().astype(np.float64), a[0])
    assert bool(torch.isfinite(big.state).all())
    # (e) oracle on a sample
    perm = np.array(__import__("robobee3d_amd._lib", fromlist=["lib"]).lib().umpcKKTPerm().contents)
    sel = np.arange(0, B, 128)
    s_o = np.ascontiguousarray(st[:, sel], np.float64)
    ctrl = np.zeros((127, len(sel))); ctrl[124:] = 1
    out_o, _, _ = oracle_built.batch_rollout(s_o, ctrl, np.ascontiguousarray(ref[:, sel], np.float64), K, dtype=np.float64,
                                             perm=perm, plant_mode=1)
    margin("quad form vs fp64 oracle (32 robots, K = 6): |dp| mm", float(np.abs(a[0][0:3, sel] - s_o[0:3]).max()), 1e-3)
    margin("quad form vs fp64 oracle: |dR|, |ddq|", float(np.abs(a[0][3:, sel] - s_o[3:]).max()), 1e-4)
    margin("quad form vs fp64 oracle: |d thrust|", float(np.abs(a[1][0, sel] - out_o[0]).max()), TOL_T)
    margin("quad form vs fp64 oracle: |d moment| / max(2e-2, 1e-3|u|)",
           float((np.abs(a[1][1:3, sel] - out_o[1:3]) / tol_tau(out_o[1:3])).max()), 1.0)


def test_dropin_runs_the_quad_form(torch_cuda):
    """umpcUpdate (B = 1) dispatches the quad form; UMPC_QUAD=0 in a fresh process restores the lane form, and the two
    agree inside the single-step band on the first call of the reference fixture."""
    import os
    import subprocess
    import sys
    from robobee3d_amd import _lib
    from robobee3d_amd.uprightmpc2py import UprightMPC2C
    seq = golden("seq_iter50.npz")
    upc = UprightMPC2C(5, 9.81e-3, 2, 1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2, np.array([3333., 3333., 1000.]), 50)
    uq, ac = upc.update(seq["p0"][0], seq["R0"][0], seq["dq0"][0], seq["pdes"][0], seq["dpdes"][0], seq["sdes"][0],
                        float(seq["actualT0"][0]))
    assert abs(uq[0] - seq["uquad"][0][0]) <= TOL_T and np.all(np.abs(uq[1:] - seq["uquad"][0][1:]) <= tol_tau(seq["uquad"][0][1:]))
    assert np.abs(ac - seq["accdes"][0]).max() <= TOL_A
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys
sys.path.append(%r)
import numpy as np
from robobee3d_amd.uprightmpc2py import UprightMPC2C
seq = np.load(%r)
upc = UprightMPC2C(5, 9.81e-3, 2, 1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2, np.array([3333., 3333., 1000.]), 50)
u, a = upc.update(seq["p0"][0], seq["R0"][0], seq["dq0"][0], seq["pdes"][0], seq["dpdes"][0], seq["sdes"][0], float(seq["actualT0"][0]))
print(" ".join("%%.9g" %% v for v in list(u) + list(a)))
""" % (root, os.path.join(root, "tests", "golden", "seq_iter50.npz"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/", env=dict(os.environ, UMPC_QUAD="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    lane = np.array([float(t) for t in r.stdout.strip().splitlines()[-1].split()])
    assert abs(lane[0] - uq[0]) <= TOL_T and np.all(np.abs(lane[1:3] - uq[1:]) <= tol_tau(uq[1:]))
    assert np.abs(lane[3:] - ac).max() <= TOL_A


def test_fp64_quad_form_on_the_gpu(torch_cuda, oracle_built, margin):
    """Round 4: the fp64 ADMM phase with one robot per lane quad (asmquad64.py), what BASELINE config 2's 4 096 robots take.
    (a) "auto" dispatches it for B <= 4 096; (b) against the lane form over 6 closed-loop steps: 1e-9 (summation order);
    (c) against the fp64 oracle: the fp64 bound of the path, 1e-9; (d) a ragged batch (37 robots) and one robot alone
    compute bit for bit what they compute among 4 096."""
    torch = torch_cuda
    from robobee3d_amd import _lib
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    B, K = 4096, 6
    st, ref = hover_initial_conditions(B, 20201117, np.float64)
    res = {}
    for form in ("auto", "lane", "quad"):
        m = BatchUprightMPC(B, torch.float64, plant_mode=0)
        m.set_step_kernel(form)
        m.set_state(st, ref)
        m.rollout(K // 2)
        m.rollout(K - K // 2)
        assert m.kernel_name == ("umpc_rollout_kernel<double, LDSF, ASM64>" if form == "lane" else
                                 "umpc_rollout_kernel<double, LDSF, ASM64, QUAD>")
        res[form] = [t.cpu().numpy() for t in (m.state, m.out, m.stats, m.ctrl, m.status)]
    for k in range(5):
        assert np.array_equal(res["auto"][k], res["quad"][k])
    a, c = res["quad"], res["lane"]
    margin("fp64 quad vs lane form, B = 4096, K = 6: |d state|", float(np.abs(a[0] - c[0]).max()), 1e-9)
    margin("fp64 quad vs lane form: |d out| relative", float((np.abs(a[1] - c[1]) / (1e-3 + np.abs(c[1]))).max()), 1e-8)
    margin("fp64 quad vs lane form: status words that differ", float(np.sum(a[4] != c[4])), 8)
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    sel = np.arange(0, B, 64)
    s_o = np.ascontiguousarray(st[:, sel])
    ctrl = np.zeros((127, len(sel))); ctrl[124:] = 1
    out_o, _, _ = oracle_built.batch_rollout(s_o, ctrl, np.ascontiguousarray(ref[:, sel]), K, dtype=np.float64, perm=perm, plant_mode=0)
    margin("fp64 quad form vs fp64 oracle (64 robots, K = 6): |d state|", float(np.abs(a[0][:, sel] - s_o).max()), 1e-9)
    margin("fp64 quad form vs fp64 oracle: |d out| relative", float((np.abs(a[1][:, sel] - out_o) / (1e-3 + np.abs(out_o))).max()), 1e-8)
    for n, off in ((37, 1000), (1, 4095)):
        sub = BatchUprightMPC(n, torch.float64, plant_mode=0)
        sub.set_state(np.ascontiguousarray(st[:, off:off + n]), np.ascontiguousarray(ref[:, off:off + n]))
        sub.rollout(K // 2)
        sub.rollout(K - K // 2)
        assert sub.kernel_name == "umpc_rollout_kernel<double, LDSF, ASM64, QUAD>"
        assert np.array_equal(sub.state.cpu().numpy(), a[0][:, off:off + n]) and np.array_equal(sub.out.cpu().numpy(), a[1][:, off:off + n])
        assert np.array_equal(sub.ctrl.cpu().numpy(), a[3][:, off:off + n])


def test_compiled_binding_equals_the_ctypes_binding(torch_cuda):
    """The compiled module (csrc/uprightmpc2py_ext.cpp; what `from uprightmpc2py import UprightMPC2C, WLCon` names) and the
    ctypes classes call the same C symbols: a warm-started 12-call sequence through each gives the same bits -- outputs,
    debug fields (vectors / matrices), status -- including the R0 row-major -> column-major conversion and the six-argument
    call of template/uprightmpc2.py:139."""
    from robobee3d_amd import uprightmpc2py as w
    seq = golden("seq_iter50.npz")
    assert w.binding() == "pybind11"
    prm = (5, 9.81e-3, 2, 1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2)
    Ib = np.array([3333., 3333., 1000.])
    a, b = w.UprightMPC2C(*prm, Ib, 50), w.UprightMPC2C_ctypes(*prm, Ib, 50)
    for k in range(12):
        args = (seq["p0"][k], seq["R0"][k], seq["dq0"][k], seq["pdes"][k], seq["dpdes"][k], seq["sdes"][k])
        ra = a.update(*args, float(seq["actualT0"][k])) if k % 2 else a.update(*args)
        rb = b.update(*args, float(seq["actualT0"][k])) if k % 2 else b.update(*args)
        assert ra[0].dtype == np.float32 and ra[0].shape == (3,) and ra[1].shape == (6,)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and a.status() == b.status()
        for x, y in zip(a.vectors() + a.matrices(), b.vectors() + b.matrices()):
            assert np.array_equal(x, y) and x.dtype == y.dtype
    # a transposed (non-contiguous) view is taken by its logical layout, as the Eigen caster would
    R = np.asfortranarray(seq["R0"][3])
    assert np.array_equal(a.update(seq["p0"][3], R, *[seq[n][3] for n in ("dq0", "pdes", "dpdes", "sdes")])[0],
                          b.update(seq["p0"][3], R, *[seq[n][3] for n in ("dq0", "pdes", "dpdes", "sdes")])[0])
    g = golden("wl_step.npz")
    wa = w.WLCon(g["u0"], g["umin"], g["umax"], g["dumax"], g["Qw"], float(g["controlRate"]), g["popts"])
    wb = w.WLCon_ctypes(g["u0"], g["umin"], g["umax"], g["dumax"], g["Qw"], float(g["controlRate"]), g["popts"])
    for k in range(4):
        ua, ub = wa.update(g["h0"][k], g["pdotdes"][k]), wb.update(g["h0"][k], g["pdotdes"][k])
        assert np.array_equal(ua[0], ub[0]) and np.array_equal(ua[1], ub[1])

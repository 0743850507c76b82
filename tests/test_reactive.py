"""the reactive baseline of the reference's gain sweeps. reactiveController
(template/template_controllers.py:282-296) and its closed loop controlTest(useMPC=False)
(template/uprightmpc2.py:121-151). Oracle pinned by tests/golden/reactive.npz (the reference's python, run by
tests/golden/make_golden.py); the HIP kernel is checked against the oracle."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import golden


def test_oracle_reactive_controller_matches_reference(oracle_built):
    g = golden("reactive.npz")
    for k in range(len(g["p"])):
        u = oracle_built.reactive(g["p"][k], g["R"][k], g["dq"][k], g["pdes"][k], g["k"][k])
        assert np.allclose(u, g["u"][k], rtol=1e-12, atol=1e-13), k


def _hover_state():
    st = np.zeros((18, 1))
    st[3:12, 0] = Rotation.from_euler('xyz', [0.5, -0.5, 0]).as_matrix().T.ravel()   # column-major
    st[12, 0] = 0.1
    ref = np.zeros((9, 1)); ref[8] = 1
    return st, ref


def test_oracle_reproduces_reference_reactive_log(oracle_built):
    """The reference's own controlTest(None, 100, useMPC=False, taulim=10, ks=[15, 120]) log and logMetric pair."""
    g = golden("reactive.npz")
    st, ref = _hover_state()
    gains = np.array([[5e-3], [5e-1], [1e-1], [1e0], [g["log_ks"][0]], [g["log_ks"][1]]])
    n = len(g["log_t"])
    _, stats, log = oracle_built.reactive_rollout(st, ref, n, 1, gains, taulim=10.0, log_robot=0)
    assert np.allclose(log[:, :12], g["log_y"], rtol=1e-9, atol=1e-10)
    assert np.allclose(log[:, 12:], g["log_u"], rtol=1e-9, atol=1e-11)
    assert np.allclose(stats[:, 0] / n, g["metric"], rtol=1e-9)


@pytest.mark.gpu
def test_gpu_reactive_rollout_matches_oracle(oracle_built):
    """A (ks0, ks1) grid like gainTuningSims(useMPC=False) (template/uprightmpc2.py:272-303) as ONE batch."""
    import torch
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    B, n = 100, 400
    k1, k2 = np.meshgrid(np.linspace(5, 20, 10), np.linspace(50, 200, 10), indexing="ij")
    gains = np.tile(np.array([[5e-3], [5e-1], [1e-1], [1e0], [0], [0]]), (1, B))
    gains[4], gains[5] = k1.ravel(), k2.ravel()
    for tdt, ndt, tol in ((torch.float64, np.float64, 1e-9), (torch.float32, np.float32, 2e-3)):
        st, ref = hover_initial_conditions(B, 7, ndt)
        mpc = BatchUprightMPC(B, tdt, taulim=10.0)
        mpc.set_state(st, ref)
        mpc.reactive_rollout(n, gains)
        mpc.reactive_rollout(n, gains, every=3)
        torch.cuda.synchronize()
        so = st.astype(np.float64)
        ro = np.ascontiguousarray(ref, np.float64)
        _, s1, _ = oracle_built.reactive_rollout(so, ro, n, 1, gains, taulim=10.0)
        out, s2, _ = oracle_built.reactive_rollout(so, ro, n, 3, gains, taulim=10.0, t0=n * 0.2)
        got = mpc.state.cpu().numpy().astype(np.float64)
        scale = np.maximum(1.0, np.abs(so))
        assert np.max(np.abs(got - so) / scale) < tol, tdt
        assert np.allclose(mpc.out[:3].cpu().numpy(), out, rtol=max(tol, 1e-8) * 10, atol=tol)
        assert np.allclose(mpc.stats.cpu().numpy(), s1 + s2, rtol=max(tol, 1e-8) * 10)
        assert mpc.time_ms == pytest.approx(2 * n * 0.2)


@pytest.mark.gpu
def test_gpu_reactive_log_equals_the_reference_log():
    """control_test_log(use_mpc=False) in fp64 against the reference's own controlTest(useMPC=False) log."""
    import torch
    from robobee3d_amd.batch import BatchUprightMPC
    g = golden("reactive.npz")
    B = 4
    st, ref = _hover_state()
    mpc = BatchUprightMPC(B, torch.float64, taulim=10.0)
    mpc.set_state(np.repeat(st, B, 1), np.repeat(ref, B, 1))
    gains = np.tile(np.array([[5e-3], [5e-1], [1e-1], [1e0], [g["log_ks"][0]], [g["log_ks"][1]]]), (1, B))
    logs = mpc.control_test_log(100, robots=(0, 3), use_mpc=False, gains=gains)
    for r in (0, 3):
        lg = logs[r]
        assert np.allclose(lg["t"], g["log_t"]) and lg["y"].shape == g["log_y"].shape
        assert np.allclose(lg["y"], g["log_y"], rtol=1e-8, atol=1e-9)
        assert np.allclose(lg["u"], g["log_u"], rtol=1e-8, atol=1e-10)
        assert np.allclose(lg["metric"], g["metric"], rtol=1e-8)
        assert set(lg) >= {"t", "y", "u", "pdes", "accdes"}      # the keys viewControlTestLog reads


@pytest.mark.gpu
def test_gpu_mpc_log_matches_oracle_loop(oracle_built):
    """control_test_log(use_mpc=True), fp64, helix task: the same loop stepped on the CPU with the oracle."""
    import torch
    from robobee3d_amd import _lib
    from robobee3d_amd.batch import BatchUprightMPC
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    st, ref = _hover_state()
    mpc = BatchUprightMPC(2, torch.float64)
    mpc.set_state(np.repeat(st, 2, 1), np.repeat(ref, 2, 1))
    mpc.ref[0:3] = 0
    mpc.set_task("helix", trajAmp=20, trajFreq=2, dz=0.1, useY=False)
    nsub, Nt = 25, 120
    lg = mpc.control_test_log(Nt * 0.2, robots=(1,))[1]
    o = oracle_built.Oracle(np.float64, perm=perm)
    o.set_canonical(True, np.ones(3))
    p, R, dq = st[0:3, 0].copy(), st[3:12, 0].reshape(3, 3).T.copy(), st[12:18, 0].copy()
    u = np.zeros(3)
    for ti in range(Nt):
        t = ti * 0.2
        rf = oracle_built.task_reference(1, (20, 2, 0.1, 0), t, np.zeros(3))
        if ti % nsub == 0:
            u, acc = o.update(p, R, dq, rf[0:3], rf[3:6], rf[6:9])
            u = u.copy(); u[1:3] = np.clip(u[1:3], -100, 100)
            assert np.allclose(lg["accdes"][ti], acc, rtol=1e-6, atol=1e-9)
        p, R, dq = oracle_built.plant_step(p, R, dq, u, 0.2)
        assert np.allclose(lg["pdes"][ti], rf[0:3], rtol=1e-12, atol=1e-12)
        assert np.allclose(lg["u"][ti], u, rtol=1e-6, atol=1e-9)
        assert np.allclose(lg["y"][ti], np.hstack((p, R[:, 2], dq)), rtol=1e-6, atol=1e-8), ti

"""reference generators (template/flight_tasks.py) evaluated on device at every
MPC fire, and per-robot objective weights (gain sweeps, template/uprightmpc2.py:272-303).
CPU: the oracle's generators are pinned to golden vectors produced by importing the reference's
flight_tasks.py. GPU: closed loops with tasks / weight tables match the oracle."""
import numpy as np
import pytest

from conftest import golden, record_margin

# (task id, parameter tuple in the C ABI's order, golden key); defaults of flight_tasks.py
CASES = [
    (1, (0.0, 0.0, 0.1, 0.0), "hover"),        # helix(trajAmp=0, trajFreq=0, dz=.1, useY=False): the hover harness call
    (1, (80.0, 1.0, 0.15, 1.0), "helix"),
    (2, (500.0, 2.0), "straightAcc"),
    (3, (100.0, 200.0), "flip"),
    (4, (500.0, 100.0, 450.0, 0.2), "perch"),
]


def test_oracle_task_generators_match_reference_python(oracle_built):
    g = golden("flight_tasks.npz")
    for task, tp, key in CASES:
        for k, t in enumerate(g["t"]):
            r = oracle_built.task_reference(task, tp, float(t), g["p0"], np.float64)
            np.testing.assert_allclose(r, g[key][k], rtol=1e-13, atol=1e-13, err_msg="%s t=%g" % (key, t))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES[1:], ids=[c[2] for c in CASES[1:]])
def test_closed_loop_with_task_matches_oracle(oracle_built, case):
    import torch
    from robobee3d_amd import _lib
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    task, tp, key = case
    names = {1: "helix", 2: "straightAcc", 3: "flip", 4: "perch"}
    kwn = BatchUprightMPC.TASKS[names[task]][1]
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    B, K, t0 = 128, 8, 35.0
    st, ref = hover_initial_conditions(B, 11, np.float64, tilt=0.2)
    rng = np.random.default_rng(5)
    ref[:] = 0
    ref[0:3] = rng.normal(size=(3, B))           # initialPos per robot
    st[0:3] = ref[0:3]
    ctrl = np.zeros((127, B)); ctrl[124:] = 1
    s_o = st.copy()
    out_o, stats_o, _ = oracle_built.batch_rollout(s_o, ctrl, ref, K, dtype=np.float64, perm=perm, task=task,
                                                   task_p=tp, t0=t0)
    m = BatchUprightMPC(B, torch.float64)
    m.set_state(st, ref)
    m.set_task(names[task], t_ms=t0, **dict(zip(kwn, tp)))
    m.rollout(K // 2)
    assert m.time_ms == t0 + (K // 2) * 25 * 0.2
    m.rollout(K - K // 2)                        # time carries over between launches
    np.testing.assert_allclose(m.state.cpu().numpy(), s_o, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(m.out.cpu().numpy(), out_o, rtol=1e-6, atol=1e-8)
    # fp32 follows within the closed-loop fp32 band
    m32 = BatchUprightMPC(B, torch.float32)
    m32.set_state(st.astype(np.float32), ref.astype(np.float32))
    m32.set_task(names[task], t_ms=t0, **dict(zip(kwn, tp)))
    m32.rollout(K)
    assert m32.kernel_name in ("umpc_rollout_asm_kernel", "umpc_rollout_asm_quad_kernel")      # the task generators are an option of the all-assembly stream
    s32 = m32.state.cpu().numpy().astype(np.float64)
    # straightAcc commands a 2 m/s velocity step: moments saturate at the clip and positions reach 40 mm,
    # so the fp32 band is relative there
    # (saturated, fast transients amplify fp32 round-off). Self-calibrating band: the kernel may be at most
    # 4x as far from the fp64 trajectory as the fp32 CPU oracle (the reference's arithmetic) itself is.
    s_o32 = st.astype(np.float32)
    c32 = np.zeros((127, B), np.float32); c32[124:] = 1
    oracle_built.batch_rollout(s_o32, c32, ref.astype(np.float32), K, dtype=np.float32, perm=perm, task=task,
                               task_p=tp, t0=t0)
    band_p = max(2e-3, 4 * np.abs(s_o32[0:3].astype(np.float64) - s_o[0:3]).max())
    band_r = max(3e-4, 4 * np.abs(s_o32[3:].astype(np.float64) - s_o[3:]).max())
    assert np.abs(s32[0:3] - s_o[0:3]).max() <= band_p, (np.abs(s32[0:3] - s_o[0:3]).max(), band_p)
    assert np.abs(s32[3:] - s_o[3:]).max() <= band_r, (np.abs(s32[3:] - s_o[3:]).max(), band_r)


@pytest.mark.gpu
def test_gain_sweep_per_robot_weights_match_oracle(oracle_built):
    """A (wpr, wvr) grid like gainTuningSims (template/uprightmpc2.py:272-303) as ONE batch."""
    import torch
    from robobee3d_amd import _lib
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    perm = np.array(_lib.lib().umpcKKTPerm().contents)
    g1, g2 = np.meshgrid(np.logspace(-2, 1, 10), np.logspace(1, 4, 10))
    B, K = g1.size, 6
    W = np.tile(np.array([1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2])[:, None], (1, B))
    W[2], W[4] = g1.ravel(), g2.ravel()          # wpr, wvr per robot
    st, ref = hover_initial_conditions(B, 3, np.float64)
    st[:, :] = st[:, :1]                         # same initial condition everywhere: only the gains differ
    ctrl = np.zeros((127, B)); ctrl[124:] = 1
    s_o = st.copy()
    out_o, stats_o, _ = oracle_built.batch_rollout(s_o, ctrl, ref, K, dtype=np.float64, perm=perm, weights=W)
    m = BatchUprightMPC(B, torch.float64)
    m.set_state(st, ref)
    m.set_weights(W)
    m.rollout(K)
    np.testing.assert_allclose(m.state.cpu().numpy(), s_o, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(m.stats.cpu().numpy(), stats_o, rtol=1e-7)
    assert np.ptp(m.stats.cpu().numpy()[0]) > 0  # the gains do change the tracking metric
    # batch-constant weights through the table == no table
    m2 = BatchUprightMPC(B, torch.float64)
    m2.set_state(st, ref)
    m2.rollout(K)
    m3 = BatchUprightMPC(B, torch.float64)
    m3.set_state(st, ref)
    m3.set_weights(np.tile(np.array([1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2])[:, None], (1, B)))
    m3.rollout(K)
    np.testing.assert_array_equal(m2.state.cpu().numpy(), m3.state.cpu().numpy())
    # fp32: the sweep runs on the all-assembly kernel (per-robot weights are an option of that stream); closed-loop fp32
    # band against the fp64 oracle, self-calibrated like the task test (extreme gains amplify round-off)
    m32 = BatchUprightMPC(B, torch.float32)
    m32.set_state(st.astype(np.float32), ref.astype(np.float32))
    m32.set_weights(W.astype(np.float32))
    m32.rollout(K)
    assert m32.kernel_name in ("umpc_rollout_asm_kernel", "umpc_rollout_asm_quad_kernel")
    s32 = m32.state.cpu().numpy().astype(np.float64)
    s_o32 = st.astype(np.float32)
    c32 = np.zeros((127, B), np.float32); c32[124:] = 1
    oracle_built.batch_rollout(s_o32, c32, ref.astype(np.float32), K, dtype=np.float32, perm=perm, weights=W.astype(np.float32))
    band_p = max(2e-3, 4 * np.abs(s_o32[0:3].astype(np.float64) - s_o[0:3]).max())
    band_r = max(3e-4, 4 * np.abs(s_o32[3:].astype(np.float64) - s_o[3:]).max())
    record_margin("gain sweep 10x10 fp32 (assembly kernel) K=6", "|dp| mm vs fp64 oracle", np.abs(s32[0:3] - s_o[0:3]).max(), band_p)
    record_margin("gain sweep 10x10 fp32 (assembly kernel) K=6", "|dR|,|ddq| vs fp64 oracle", np.abs(s32[3:] - s_o[3:]).max(), band_r)
    assert np.abs(s32[0:3] - s_o[0:3]).max() <= band_p and np.abs(s32[3:] - s_o[3:]).max() <= band_r
    np.testing.assert_allclose(m32.stats.cpu().numpy(), stats_o, rtol=2e-3)


def test_save_viewlog_writes_the_reference_format(tmp_path):
    """batch.save_viewlog: gzip-pickle dict with the keys / shapes / 1 ms sampling of template/viewlog.py:8-33 (CPU:
    the writer only reshapes a log dict)."""
    import gzip
    import pickle
    from robobee3d_amd.batch import save_viewlog
    Nt = 26
    rng = np.random.default_rng(0)
    from scipy.spatial.transform import Rotation
    Rm = Rotation.from_rotvec(rng.normal(size=(Nt, 3)) * 0.3).as_matrix()
    log = {"t": np.arange(Nt) * 0.2, "y": rng.normal(size=(Nt, 12)), "u": rng.normal(size=(Nt, 3)), "pdes": rng.normal(size=(Nt, 3)),
           "accdes": rng.normal(size=(Nt, 6)), "R": Rm.transpose(0, 2, 1).reshape(Nt, 9)}
    fname = save_viewlog(str(tmp_path / "mpc"), log, timestamp="20201117000000")
    assert fname.endswith("mpc_20201117000000.zip")
    with gzip.GzipFile(fname, "rb") as zf:
        d = pickle.load(zf)
    assert set(d) == {"t", "q", "dq", "u", "accdes", "posdes"}
    assert np.allclose(np.diff(d["t"]), 1.0) and d["q"].shape == (6, 7) and d["dq"].shape == (6, 6) and d["accdes"].shape == (6, 6)
    k = np.arange(0, Nt, 5)
    np.testing.assert_allclose(Rotation.from_quat(d["q"][:, 3:]).as_matrix(), Rm[k], atol=1e-12)
    np.testing.assert_array_equal(d["q"][:, :3], log["y"][k, :3])
    np.testing.assert_array_equal(d["posdes"], log["pdes"][k])

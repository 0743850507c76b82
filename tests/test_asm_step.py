"""CPU check of the ALL-ASSEMBLY fp32 step kernel (robobee3d_amd/asmstep.py -> csrc/umpc_step_asm.h): the emitted gfx950
instruction stream -- phase A, the ADMM loop, phase C, RK4 plant, the K-step loop -- is interpreted on numpy float32
(asmstep.simulate, one lane) and compared with the oracle, so register reuse (freed registers are poisoned), loop
control, masks and the memory rows are verified before the kernel reaches a GPU."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from robobee3d_amd import asmgen, asmstep
from robobee3d_amd.batch import hover_initial_conditions, monte_carlo_draws


@pytest.fixture(scope="module")
def program():
    g = asmstep.StepGen()
    return g, g.program()


def _arrays(st, ref, b, Ib=None, gain=None, aT0=None, weights=None, taskf=None, wl=None, wlu=None, wlw=None):
    a = dict(state=st[:, b].astype(np.float32), ctrl=np.zeros(127, np.float32), ref=ref[:, b].astype(np.float32),
             ws=np.zeros(asmgen.WS_ROWS, np.float32), out=np.zeros(9, np.float32), stats=np.zeros(2, np.float32),
             status=np.zeros(1, np.int32), info=np.zeros(2, np.float32),
             Ib=None if Ib is None else Ib[:, b].astype(np.float32), gain=None if gain is None else gain[b:b + 1].astype(np.float32),
             aT0=None if aT0 is None else aT0[b:b + 1].astype(np.float32),
             weights=None if weights is None else weights[:, b].astype(np.float32),
             taskf=None if taskf is None else np.ascontiguousarray(taskf, np.float32).ravel(),
             wl=None if wl is None else np.ascontiguousarray(wl, np.float32),
             wlu=None if wlu is None else wlu[:, b].astype(np.float32), wlw=None if wlw is None else wlw[:, b].astype(np.float32))
    a["ctrl"][124:] = 1
    return a


def test_register_budget_and_size(program):
    g, ins = program
    real = [t for t in ins if t[0] not in ("kill", "label")]
    assert g.pool.peak <= 254 and len(real) < 12000
    # the stream is position-independent text: every branch target is a local numeric label
    assert all(t[1][-1] in "fb" for t in ins if t[0].startswith("s_cbranch") or t[0] == "s_branch")


def test_stream_assembles_for_gfx950(program, tmp_path):
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not present")
    _, ins = program
    src = tmp_path / "step.s"
    src.write_text("\n".join(asmstep.fmt(t) for t in ins if t[0] != "kill") + "\n")
    r = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", str(src), "-o", os.devnull], capture_output=True, text=True)
    assert r.returncode == 0 and "error" not in r.stderr, r.stderr[:2000]


def test_closed_loop_steps_match_the_oracle(program, oracle_built):
    """3 robots x 2 closed-loop steps (50 iterations, 25 RK4 substeps) against the fp64 oracle in the kernel's
    elimination order: fp32 band of the path (tests/test_gpu_parity.py), equal status codes."""
    g, ins = program
    B, K = 3, 2
    st, ref = hover_initial_conditions(B, 20201118, np.float32)
    fl = asmstep.host_floats()
    got = []
    for b in range(B):
        a = _arrays(st, ref, b)
        n = asmstep.simulate(ins, a, dict(K=K, maxIter=50, nsub=25, plant=1), fl)
        assert 100000 < n < 140000
        got.append(a)
    s64 = st.astype(np.float64)
    c64 = np.zeros((127, B)); c64[124:] = 1
    out_o, stats_o, status_o = oracle_built.batch_rollout(s64, c64, ref.astype(np.float64), K, dtype=np.float64,
                                                          perm=g.s.perm, plant_mode=1)
    for b in range(B):
        a = got[b]
        assert np.abs(a["state"][0:3] - s64[0:3, b]).max() < 1e-4 and np.abs(a["state"][3:] - s64[3:, b]).max() < 3e-5
        assert abs(a["out"][0] - out_o[0, b]) < 3e-5
        assert np.all(np.abs(a["out"][1:3] - out_o[1:3, b]) <= np.maximum(2e-2, 1e-3 * np.abs(out_o[1:3, b])))
        assert np.abs(a["out"][3:] - out_o[3:, b]).max() < 3e-5
        assert int(a["status"][0]) == int(status_o[b])
        sc = 1e-3 + np.abs(c64[:123, b]).max()
        assert np.abs(a["ctrl"][:123] - c64[:123, b]).max() / sc < 1e-3
        np.testing.assert_allclose(a["ctrl"][123:], c64[123:, b], rtol=2e-5, atol=3e-5)
        np.testing.assert_allclose(a["stats"], stats_o[:, b], rtol=1e-4)
        assert np.isfinite(a["info"]).all() and a["info"][0] > 0


def test_options_monte_carlo_actualT0_and_controller_only(program, oracle_built):
    """per-robot Ib / thrust gain (config 5), actualT0 on the first step, nsub = 0 (= umpcUpdate), 1..3 iterations"""
    g, ins = program
    B = 2
    st, ref = hover_initial_conditions(B, 7, np.float32)
    Ib, gain = monte_carlo_draws(B, 20201120, np.float64)
    for b in range(B):
        a = _arrays(st, ref, b, Ib=Ib, gain=gain)
        asmstep.simulate(ins, a, dict(K=1, maxIter=50, nsub=25, plant=1), asmstep.host_floats())
        s64 = np.ascontiguousarray(st[:, b:b + 1]).astype(np.float64)
        c64 = np.zeros((127, 1)); c64[124:] = 1
        out_o, _, status_o = oracle_built.batch_rollout(s64, c64, np.ascontiguousarray(ref[:, b:b + 1]).astype(np.float64), 1,
                                                        dtype=np.float64, perm=g.s.perm, plant_mode=1,
                                                        Ib=np.ascontiguousarray(Ib[:, b:b + 1]), gain=gain[b:b + 1].copy())
        assert np.abs(a["state"] - s64[:, 0]).max() < 1e-4 and abs(a["out"][0] - out_o[0, 0]) < 3e-5
        assert int(a["status"][0]) == int(status_o[0])
    # actualT0 + controller only + few iterations
    o = oracle_built.Oracle(np.float64, perm=g.s.perm)
    for it in (1, 2, 3):
        a = _arrays(st, ref, 0, aT0=np.array([0.0123, -1.0]))
        before = a["state"].copy()
        asmstep.simulate(ins, a, dict(K=1, maxIter=it, nsub=0, plant=1), asmstep.host_floats())
        assert np.array_equal(a["state"], before)                      # nsub = 0: the state is only read
        o = oracle_built.Oracle(np.float64, perm=g.s.perm, maxIter=it)
        o.set_canonical(True)
        R = st[3:12, 0].reshape(3, 3).T
        uq, ac = o.update(st[0:3, 0], R, st[12:18, 0], ref[0:3, 0], ref[3:6, 0], ref[6:9, 0], 0.0123)
        sc = 15.0 if it == 1 else 1.0
        assert abs(a["out"][0] - uq[0]) < 3e-5 * sc and np.abs(a["out"][3:] - ac).max() < 3e-5 * sc
        assert np.all(np.abs(a["out"][1:3] - uq[1:]) <= np.maximum(2e-2, 1e-3 * np.abs(uq[1:])) * sc)
        assert abs(a["ctrl"][123] - uq[0]) < 3e-5 * sc


def test_limit_scaling_exact_path(program, oracle_built):
    """Weights far outside [1e-4, 1e4] engage the exact limit_scaling branch of the Ruiz passes (scaling.c:7-14) that
    the wave-wide min / max test normally skips."""
    g, ins = program
    st, ref = hover_initial_conditions(1, 3, np.float32)
    kw = dict(wvf=5e5, wmom=2e-5)
    a = _arrays(st, ref, 0)
    asmstep.simulate(ins, a, dict(K=1, maxIter=50, nsub=0, plant=1), asmstep.host_floats(**kw))
    o = oracle_built.Oracle(np.float64, perm=g.s.perm, **kw)
    o.set_canonical(True)
    R = st[3:12, 0].reshape(3, 3).T
    uq, ac = o.update(st[0:3, 0], R, st[12:18, 0], ref[0:3, 0], ref[3:6, 0], ref[6:9, 0], -1.0)
    assert abs(a["out"][0] - uq[0]) < 1e-4 and np.abs(a["out"][3:] - ac).max() < 1e-4     # extreme weights: worse conditioning
    assert np.all(np.abs(a["out"][1:3] - uq[1:]) <= np.maximum(5e-2, 2e-3 * np.abs(uq[1:])))
    np.testing.assert_allclose(a["ctrl"][124:], np.asarray(o.get("E"))[36:39], rtol=1e-4)


def test_reference_euler_expm_plant_mode(program, oracle_built):
    """plant = 0: the reference's own step (template/genqp.py:32-41, Euler + SO(3) exponential) inside the assembly
    kernel, against the fp64 oracle's restatement of it (which reproduces scipy's expm to 1e-12, tests/test_oracle_golden.py)."""
    g, ins = program
    B, K = 2, 2
    st, ref = hover_initial_conditions(B, 5, np.float32)
    for b in range(B):
        a = _arrays(st, ref, b)
        asmstep.simulate(ins, a, dict(K=K, maxIter=50, nsub=25, plant=0), asmstep.host_floats())
        s64 = np.ascontiguousarray(st[:, b:b + 1]).astype(np.float64)
        c64 = np.zeros((127, 1)); c64[124:] = 1
        out_o, stats_o, _ = oracle_built.batch_rollout(s64, c64, np.ascontiguousarray(ref[:, b:b + 1]).astype(np.float64), K,
                                                       dtype=np.float64, perm=g.s.perm, plant_mode=0)
        assert np.abs(a["state"][0:3] - s64[0:3, 0]).max() < 1e-4 and np.abs(a["state"][3:] - s64[3:, 0]).max() < 3e-5
        R = a["state"][3:12].reshape(3, 3)
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-5
        np.testing.assert_allclose(a["stats"], stats_o[:, 0], rtol=1e-4)


# ----------------------------------------------------------------------------------------------------------------
# SURVEY 8(f) options of the all-assembly stream: per-robot weights, task table, fused WL step
# ----------------------------------------------------------------------------------------------------------------
def _check_band(a, s64, out_o, b, k_state=(1e-4, 3e-5)):
    assert np.abs(a["state"][0:3] - s64[0:3, b]).max() < k_state[0] and np.abs(a["state"][3:] - s64[3:, b]).max() < k_state[1]
    assert abs(a["out"][0] - out_o[0, b]) < 3e-5
    assert np.all(np.abs(a["out"][1:3] - out_o[1:3, b]) <= np.maximum(2e-2, 1e-3 * np.abs(out_o[1:3, b])))
    assert np.abs(a["out"][3:] - out_o[3:, b]).max() < 3e-5


def test_per_robot_weights(program, oracle_built):
    """A (wpr, wvr) gain sweep (template/uprightmpc2.py:272-303): the weights table is read per robot, 1 / w by
    v_rcp + Newton, both parked in AGPRs across the Ruiz passes and the loop."""
    g, ins = program
    B, K = 3, 2
    st, ref = hover_initial_conditions(B, 3, np.float32)
    W = np.tile(np.array([1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2])[:, None], (1, B))
    W[2] = [0.02, 1.0, 7.0]          # wpr
    W[4] = [30.0, 1e3, 8e3]          # wvr
    s64 = st.astype(np.float64)
    c64 = np.zeros((127, B)); c64[124:] = 1
    out_o, stats_o, status_o = oracle_built.batch_rollout(s64, c64, ref.astype(np.float64), K, dtype=np.float64,
                                                          perm=g.s.perm, plant_mode=1, weights=W)
    for b in range(B):
        a = _arrays(st, ref, b, weights=W)
        asmstep.simulate(ins, a, dict(K=K, maxIter=50, nsub=25, plant=1), asmstep.host_floats())
        _check_band(a, s64, out_o, b)
        np.testing.assert_allclose(a["stats"], stats_o[:, b], rtol=3e-4)     # (sum of squared moments: extreme gains amplify round-off)
    # the table with the batch constants == no table (same weights; reciprocal by Newton vs the host's division)
    a0, a1 = _arrays(st, ref, 0), _arrays(st, ref, 0, weights=np.tile(np.array([1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2])[:, None], (1, B)))
    for a in (a0, a1):
        asmstep.simulate(ins, a, dict(K=1, maxIter=50, nsub=25, plant=1), asmstep.host_floats())
    assert np.abs(a0["state"] - a1["state"]).max() < 2e-6 and np.abs(a0["out"] - a1["out"]).max() < 2e-3


def _task_table(oracle_built, task, tp, t0, K, nsub=25, dtsim=0.2):
    """What the host side hands the kernel: per step (dp[3], dpdes[3], sdes_x, sdes_z) of the task at the fire time,
    evaluated with initialPos = 0 in float32 (umpc_taskf_kernel)."""
    f = np.float32
    tab = np.zeros((K, 8), f)
    for k in range(K):
        t = f(t0) + f(k) * (f(nsub) * f(dtsim))
        r = oracle_built.task_reference(task, tp, float(t), np.zeros(3), np.float32)
        assert r[7] == 0
        tab[k] = [r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[8]]
    return tab


@pytest.mark.parametrize("task,tp", [(1, (80.0, 1.0, 0.15, 1.0)), (3, (100.0, 200.0)), (4, (500.0, 100.0, 450.0, 0.2))],
                         ids=["helix", "flip", "perch"])
def test_task_table(program, oracle_built, task, tp):
    """template/flight_tasks.py generators: the time-dependent part of the reference is one scalar load per step; rows
    0..2 of ref are the robot's initialPos."""
    g, ins = program
    B, K, t0 = 2, 3, 110.0
    st, ref = hover_initial_conditions(B, 11, np.float32, tilt=0.2)
    rng = np.random.default_rng(5)
    ref[:] = 0
    ref[0:3] = rng.normal(size=(3, B))
    st[0:3] = ref[0:3]
    s64 = st.astype(np.float64)
    c64 = np.zeros((127, B)); c64[124:] = 1
    out_o, _, _ = oracle_built.batch_rollout(s64, c64, ref.astype(np.float64), K, dtype=np.float64, perm=g.s.perm,
                                             plant_mode=1, task=task, task_p=tp, t0=t0)
    tab = _task_table(oracle_built, task, tp, t0, K)
    for b in range(B):
        a = _arrays(st, ref, b, taskf=tab)
        asmstep.simulate(ins, a, dict(K=K, maxIter=50, nsub=25, plant=1), asmstep.host_floats())
        _check_band(a, s64, out_o, b, k_state=(3e-4, 1e-4))


def _wldev_words(args, Mdiag):
    """struct umpc::WLDev (csrc/umpc_step.h) as the host builds it from wlConInit's WLCon_t (funapprox.c:35-51,102-116)"""
    u0, umin, umax, dumax, Qw, rate, popts = args
    f = np.float32
    w = np.zeros(150, f)
    w[0:4], w[4:8] = f(umin), f(umax)
    w[8:12] = f(dumax) / f(rate)
    w[12:18] = f(Qw)
    for i in range(6):
        p = f(popts)[15 * i:15 * i + 15]
        w[18 + i] = p[0]
        w[24 + 4 * i:28 + 4 * i] = p[1:5]
        A2 = np.zeros(16, f)
        kk = 0
        for r in range(4):
            for c in range(r, 4):
                A2[r + 4 * c] = A2[c + 4 * r] = p[5 + kk]
                kk += 1
        w[48 + 16 * i:64 + 16 * i] = A2
    w[144:150] = f(Mdiag)
    return w


def test_fused_wl_step(program, oracle_built):
    """MPC -> WL -> actualT0 (template/robobee_test_controllers.py:162-171) fused behind every controller step, against
    the oracle's coupled rollout: state, WL input state u4, wrench w0 and the fed-back thrust."""
    from conftest import golden
    from test_wl_step import _args, _loop_inputs
    g, ins = program
    gl = golden("mpc_wl_loop.npz")
    K = 4
    Md = (100.0, 100.0, 100.0, 3333.0, 3333.0, 1000.0)
    st, ref, ctrl, u4 = _loop_inputs(gl, np.float64, 1)
    wlo = oracle_built.WLOracle(*_args(gl), dtype=np.float64)
    w64 = np.zeros((6, 1))
    s64, c64, u64 = st.copy(), ctrl.copy(), u4.copy()
    out_o, _, _ = oracle_built.batch_rollout(s64, c64, ref.copy(), K, dtype=np.float64, perm=g.s.perm, plant_mode=1,
                                             wl=wlo, wl_u=u64, wl_w=w64)
    a = _arrays(st.astype(np.float32), ref.astype(np.float32), 0, wl=_wldev_words(_args(gl), Md),
                wlu=u4.astype(np.float32), wlw=np.zeros((6, 1), np.float32))
    asmstep.simulate(ins, a, dict(K=K, maxIter=50, nsub=25, plant=1), asmstep.host_floats())
    _check_band(a, s64, out_o, 0, k_state=(2e-4, 5e-5))
    np.testing.assert_allclose(a["wlw"], w64[:, 0], rtol=1e-4, atol=5e-4)
    assert np.all(np.abs(a["wlu"] - u64[:, 0]) <= 2e-2 * np.array([5.0, 0.01, 0.01, 0.01]) + 1e-6)       # 2 % of the rate limits
    np.testing.assert_allclose(a["ctrl"][123], c64[123, 0], rtol=1e-4)          # accumulator := actualT0 = w0[2] / M0[2,2]
    assert abs(a["ctrl"][123] - a["out"][0]) > 1e-6                                # ... which is not T0 + u0

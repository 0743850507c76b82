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


def _arrays(st, ref, b, Ib=None, gain=None, aT0=None):
    a = dict(state=st[:, b].astype(np.float32), ctrl=np.zeros(127, np.float32), ref=ref[:, b].astype(np.float32),
             ws=np.zeros(asmgen.WS_ROWS, np.float32), out=np.zeros(9, np.float32), stats=np.zeros(2, np.float32),
             status=np.zeros(1, np.int32), info=np.zeros(2, np.float32),
             Ib=None if Ib is None else Ib[:, b].astype(np.float32), gain=None if gain is None else gain[b:b + 1].astype(np.float32),
             aT0=None if aT0 is None else aT0[b:b + 1].astype(np.float32))
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

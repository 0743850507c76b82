"""CPU check of the ONE-ROBOT-PER-LANE-QUAD form of the all-assembly fp32 step stream (robobee3d_amd/asmquad.py inside
asmstep.StepGen(quad=True) -> csrc/umpc_step_asm_quad.h): the emitted gfx950 instructions are interpreted -- the phases
around the loop on one lane (asmstep.simulate; the four lanes of a quad run them redundantly), the quad section on the
four lanes of a quad with DPP selects, EXEC masks, AGPRs and each lane's LDS slice (asmquad.simulate) -- and compared
with the fp64 oracle and with the one-lane form of the same stream before anything reaches a GPU."""
import os
import subprocess

import numpy as np
import pytest

from robobee3d_amd import asmgen, asmquad, asmstep
from robobee3d_amd.batch import hover_initial_conditions
from test_asm_step import _arrays


@pytest.fixture(scope="module")
def programs():
    g1 = asmstep.StepGen()
    g4 = asmstep.StepGen(quad=True)
    return g1, g1.program(), g4, g4.program()


def _section(ins, name=None):
    """the ADMM section (name None) or a named one ("ruiz")"""
    qb = ins.index(("quad_begin",) if name is None else ("quad_begin", name))
    qe = next(k for k in range(qb, len(ins)) if ins[k][0] == "quad_end")
    return ins[qb + 1:qe]


def test_plan_covers_every_entry_once_and_no_lane_reads_what_its_instruction_writes(programs):
    _, _, g4, _ = programs
    pl = asmquad.QuadPlan(g4.st)
    s = g4.s
    nL = len(s.L_i)
    for sched, fwd in ((pl.fwd, True), (pl.bwd, False)):
        seen = []
        final_after = {}
        for q, ins in enumerate(sched):
            dsts = {o[0] for o in ins["ops"].values()}
            for ln, (d, sr, j) in ins["ops"].items():
                assert pl.home[d] == (ln, ins["d"]) and pl.home[sr][1] == ins["s"] and ins["perm"][ln] == pl.home[sr][0]
                assert sr not in dsts, "a lane reads an unknown another lane of the same instruction writes"
                r, c = s.L_i[j], next(c for c in range(s.nk) if s.L_p[c] <= j < s.L_p[c + 1])
                assert (d, sr) == ((r, c) if fwd else (c, r))
                seen.append(j)
                final_after[d] = q
        assert sorted(seen) == list(range(nL))
        # a source is final (no later instruction writes it) when it is read
        for q, ins in enumerate(sched):
            for (_, sr, _) in ins["ops"].values():
                assert final_after.get(sr, -1) < q
    assert len(pl.fwd) <= 96 and len(pl.bwd) <= 102
    assert sum(1 for kind, _ in pl.cloc if kind == "v") <= asmquad.NCV


def test_stream_size_and_exec_discipline(programs):
    _, ins1, g4, ins4 = programs
    assert g4.pool.peak <= 254
    sec = _section(ins4)
    labs = [k for k, t in enumerate(sec) if t[0] == "label"]
    assert len(labs) == 2
    body = [t for t in sec[labs[0]:labs[1]] if t[0] != "label"]
    # one quad iteration against the one-lane body (804 instructions): VERDICT r3 item 1
    assert len(body) <= 330, len(body)
    assert not any(t[0] == "s_mov_b64" and t[1] == "exec" for t in body), "EXEC must stay full inside the iterations (DPP)"
    # the only memory instructions of an iteration: the ~20 float4 reads of the coefficient words that found no VGPR home
    assert not any(t[0].startswith("global_") or t[0].startswith("ds_write") for t in body)
    assert sum(1 for t in body if t[0] == "ds_read_b128") <= 20 and not any(t[0] == "v_accvgpr_read_b32" for t in body)
    # EXEC masks only around plain moves: no DPP instruction between a mask switch and the restore
    masked = False
    for t in sec:
        if t[0] == "s_mov_b64" and t[1] == "exec":
            masked = t[2] != "s[%d:%d]" % (asmquad.S_EXEC, asmquad.S_EXEC + 1)
        assert not (masked and t[0].endswith("_dpp")), t
    real4 = [t for t in ins4 if t[0] not in asmstep.PSEUDO and t[0] != "label"]
    assert len(real4) < 12000
    # the Ruiz section: one pass <= 460 instructions (one-lane: 730), EXEC masks only around plain moves / maxima
    rz = _section(ins4, "ruiz")
    lab = [k for k, t in enumerate(rz) if t[0] == "label"][0]
    loop_end = [k for k, t in enumerate(rz) if t[0] == "s_cbranch_scc1"][-1]
    assert loop_end - lab <= 460, loop_end - lab
    masked = False
    for t in rz:
        if t[0] == "s_mov_b64" and t[1] == "exec":
            masked = t[2] != "s[%d:%d]" % (asmquad.S_EXEC, asmquad.S_EXEC + 1)
        assert not (masked and t[0].endswith("_dpp")), t


def test_quad_stream_assembles_for_gfx950(programs, tmp_path):
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        pytest.skip("llvm-mc not present")
    _, _, _, ins4 = programs
    src = tmp_path / "quad.s"
    src.write_text("\n".join(asmstep.fmt(t) for t in ins4 if t[0] not in asmstep.PSEUDO) + "\n")
    r = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", str(src), "-o", os.devnull], capture_output=True, text=True)
    assert r.returncode == 0 and "error" not in r.stderr, r.stderr[:2000]


def test_interpreter_rejects_a_dpp_read_of_a_masked_off_lane():
    """gfx9: a DPP source lane that EXEC masks off is an invalid lane and the reader's write is dropped. The generator must
    never depend on it; the interpreter raises instead of modelling it."""
    e = asmgen.Emit()
    e("quad_begin",)
    e("s_mov_b64", "s[36:37]", "exec")
    e("s_mov_b32", "s30", 0x11111111)
    e("s_mov_b32", "s31", 0x11111111)
    e("s_and_b64", "s[30:31]", "s[30:31]", "s[36:37]")
    e("s_mov_b64", "exec", "s[30:31]")
    e("v_mov_b32_dpp", "v5", "v6", asmquad.qperm([1, 1, 1, 1]))
    e("s_mov_b64", "exec", "s[36:37]")
    e("quad_end",)
    V, A, L = np.zeros((4, 256), np.uint32), np.zeros((4, 256), np.uint32), np.zeros((4, asmgen.NLDS), np.uint32)
    with pytest.raises(AssertionError, match="masked-off lane"):
        asmquad.simulate(e.ins, 0, V, A, L, {})


def test_quad_form_matches_the_oracle_and_the_lane_form(programs, oracle_built, margin):
    """4 robots x 2 closed-loop steps (50 iterations, 25 RK4 substeps): the quad form inside the fp32 band of the fp64 oracle
    (the bounds of tests/test_asm_step.py), equal status words, and no further from the oracle than the lane form is (the
    two differ only in the order in which an unknown's updates are added up)."""
    g1, ins1, g4, ins4 = programs
    B, K = 4, 2
    st, ref = hover_initial_conditions(B, 20201118, np.float32)
    fl = asmstep.host_floats()
    s64 = st.astype(np.float64)
    c64 = np.zeros((127, B)); c64[124:] = 1
    out_o, stats_o, status_o = oracle_built.batch_rollout(s64, c64, ref.astype(np.float64), K, dtype=np.float64,
                                                          perm=g4.s.perm, plant_mode=1)
    err = {1: [], 4: []}
    for b in range(B):
        a1, a4 = _arrays(st, ref, b), _arrays(st, ref, b)
        n1 = asmstep.simulate(ins1, a1, dict(K=K, maxIter=50, nsub=25, plant=1), fl)
        n4 = asmstep.simulate(ins4, a4, dict(K=K, maxIter=50, nsub=25, plant=1), fl)
        sec = asmstep.simulate.last_quad_sections
        # per step: one-lane 57 k instructions, quad form <= 31.5 k: <= 18.5 k in the ADMM section (entry, 48 + 1 iterations,
        # exit), <= 4.4 k in the ten Ruiz passes (one-lane: 7.3 k), <= 3.5 k in the 25 RK4 substeps (one-lane: 5.8 k)
        assert n4 < 0.55 * n1 and 30000 < sec["admm"] < 37000 and 6000 < sec["ruiz"] < 8800 and 5000 < sec["plant"] < 7000, (n1, n4, sec)
        for a, key in ((a1, 1), (a4, 4)):
            assert np.abs(a["state"][0:3] - s64[0:3, b]).max() < 1e-4 and np.abs(a["state"][3:] - s64[3:, b]).max() < 3e-5
            assert abs(a["out"][0] - out_o[0, b]) < 3e-5
            assert np.all(np.abs(a["out"][1:3] - out_o[1:3, b]) <= np.maximum(2e-2, 1e-3 * np.abs(out_o[1:3, b])))
            assert np.abs(a["out"][3:] - out_o[3:, b]).max() < 3e-5
            assert int(a["status"][0]) == int(status_o[b])
            sc = 1e-3 + np.abs(c64[:123, b]).max()
            assert np.abs(a["ctrl"][:123] - c64[:123, b]).max() / sc < 1e-3
            np.testing.assert_allclose(a["ctrl"][123:], c64[123:, b], rtol=2e-5, atol=3e-5)
            np.testing.assert_allclose(a["stats"], stats_o[:, b], rtol=1e-4 if key == 1 else 4e-4)
            err[key].append([abs(a["out"][0] - out_o[0, b]), np.abs(a["out"][1:3] - out_o[1:3, b]).max(),
                             np.abs(a["state"] - s64[:, b]).max()])
    e1, e4 = np.array(err[1]).mean(0), np.array(err[4]).mean(0)
    margin("quad form: mean |d thrust| vs fp64 oracle / the lane form's", e4[0] / e1[0], 3.0)
    margin("quad form: mean |d moment| vs fp64 oracle / the lane form's", e4[1] / e1[1], 3.0)
    margin("quad form: mean |d state| vs fp64 oracle / the lane form's", e4[2] / e1[2], 3.0)


@pytest.mark.parametrize("iters", [1, 2, 3, 7])
def test_iteration_counts_that_take_different_paths(programs, oracle_built, iters):
    """maxIter = 1: the quad section is skipped (the one-lane first iteration is the capturing one); 2: entry + capturing
    body + exit; 3, 7: the hardware loop in between. Controller only (nsub = 0 = umpcUpdate), actualT0 on."""
    _, ins1, g4, ins4 = programs
    st, ref = hover_initial_conditions(2, 7, np.float32)
    a1 = _arrays(st, ref, 0, aT0=np.array([0.0123, -1.0]))
    a4 = _arrays(st, ref, 0, aT0=np.array([0.0123, -1.0]))
    asmstep.simulate(ins1, a1, dict(K=1, maxIter=iters, nsub=0, plant=1), asmstep.host_floats())
    asmstep.simulate(ins4, a4, dict(K=1, maxIter=iters, nsub=0, plant=1), asmstep.host_floats())
    assert (asmstep.simulate.last_quad_sections.get("admm", 0) == 0) == (iters == 1)
    if iters == 1:
        for k in ("out", "ctrl", "info", "status"):
            assert np.array_equal(a1[k], a4[k]), k
        return
    o = oracle_built.Oracle(np.float64, perm=g4.s.perm, maxIter=iters)
    o.set_canonical(True)
    R = st[3:12, 0].reshape(3, 3).T
    uq, ac = o.update(st[0:3, 0], R, st[12:18, 0], ref[0:3, 0], ref[3:6, 0], ref[6:9, 0], 0.0123)
    for a in (a1, a4):
        assert abs(a["out"][0] - uq[0]) < 3e-5 and np.abs(a["out"][3:] - ac).max() < 3e-5
        assert np.all(np.abs(a["out"][1:3] - uq[1:]) <= np.maximum(2e-2, 1e-3 * np.abs(uq[1:])))
    sc = 1e-3 + np.abs(a1["ctrl"][:123]).max()
    assert np.abs(a1["ctrl"][:123] - a4["ctrl"][:123]).max() / sc < 2e-5
    assert int(a1["status"][0]) == int(a4["status"][0])


def test_quad_ruiz_limit_scaling_exact_path_and_per_robot_weights(programs, oracle_built):
    """the Ruiz passes on the quad with weights far outside [1e-4, 1e4] (the exact limit_scaling branch, scaling.c:7-14)
    handed over as a per-robot weights row (the gain-sweep option of the stream), against the fp64 oracle"""
    _, _, g4, ins4 = programs
    st, ref = hover_initial_conditions(1, 3, np.float32)
    kw = dict(wvf=5e5, wmom=2e-5)
    base = dict(ws=1e1, wds=1e3, wpr=1.0, wpf=5.0, wvr=1e3, wvf=2e3, wthrust=1e-1, wmom=1e-2)
    base.update(kw)
    W = np.array([[base[n]] for n in ("ws", "wds", "wpr", "wpf", "wvr", "wvf", "wthrust", "wmom")], np.float32)
    for weights in (None, W):
        a = _arrays(st, ref, 0, weights=weights)
        fl = asmstep.host_floats(**kw) if weights is None else asmstep.host_floats()
        asmstep.simulate(ins4, a, dict(K=1, maxIter=50, nsub=0, plant=1), fl)
        o = oracle_built.Oracle(np.float64, perm=g4.s.perm, **kw)
        o.set_canonical(True)
        R = st[3:12, 0].reshape(3, 3).T
        uq, ac = o.update(st[0:3, 0], R, st[12:18, 0], ref[0:3, 0], ref[3:6, 0], ref[6:9, 0], -1.0)
        assert abs(a["out"][0] - uq[0]) < 1e-4 and np.abs(a["out"][3:] - ac).max() < 1e-4
        assert np.all(np.abs(a["out"][1:3] - uq[1:]) <= np.maximum(5e-2, 2e-3 * np.abs(uq[1:])))
        np.testing.assert_allclose(a["ctrl"][124:], np.asarray(o.get("E"))[36:39], rtol=1e-4)


def test_ruiz_and_plant_sections_reproduce_the_lane_stream_bit_for_bit(programs):
    """With ONE ADMM iteration the quad stream has no quad ADMM section: what differs from the lane stream is the Ruiz passes
    and the RK4 substeps on the quad. Both go through the one-lane sequence of operations per element (the cost
    normalisation's sum of P_jj is the exception, and is exact on these weights), so 3 closed-loop steps of 2 robots, plant
    and statistics included, come out BIT-identical -- registers, lane masks, DPP selections and the write-back of every
    word of both sections checked at once."""
    _, ins1, _, ins4 = programs
    st, ref = hover_initial_conditions(2, 11, np.float32)
    fl = asmstep.host_floats()
    for b in range(2):
        a1, a4 = _arrays(st, ref, b), _arrays(st, ref, b)
        asmstep.simulate(ins1, a1, dict(K=3, maxIter=1, nsub=25, plant=1), fl)
        asmstep.simulate(ins4, a4, dict(K=3, maxIter=1, nsub=25, plant=1), fl)
        sec = asmstep.simulate.last_quad_sections
        assert sec.get("admm", 0) == 0 and sec["ruiz"] > 0 and sec["plant"] > 0
        for k in ("state", "out", "ctrl", "stats", "info", "status"):
            assert np.array_equal(a1[k], a4[k], equal_nan=True), (b, k)
        # the parked scalings: the lane form keeps c and its first nine D words in spare AGPRs (round 5), the quad form in rows
        parked = np.ones(asmgen.WS_ROWS, bool)
        parked[:asmstep.N_SPARE_D] = False
        parked[asmgen.WS_C - asmgen.WS_DS] = False
        assert np.array_equal(a1["ws"][parked], a4["ws"][parked], equal_nan=True), b
        assert not a1["ws"][~parked].any() and a4["ws"][~parked].all()

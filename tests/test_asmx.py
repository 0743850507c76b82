"""The two-lanes-per-robot ADMM iteration of the closed experiment tools/asmx.py (outside the product package: measured,
not built, DESIGN.md "Two lanes per robot"): the generated stream, interpreted on a lane pair, equals a numpy statement of the OSQP iteration
(reference: osqp 0.6.0 src/osqp.c:osqp_solve loop body -- update_xz_tilde / update_x / update_z / update_y,
auxil.c:64-140 -- with the LDL' solve of lin_sys/direct/qdldl/qdldl_interface.c:solve_linsys_qdldl)."""
import os
import sys

import numpy as np


def _asmx():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import asmx
    return asmx


def test_two_lane_iteration_equals_the_unsplit_one():
    asmx = _asmx()
    from robobee3d_amd.asmgen import Emit
    p = asmx.Plan()
    s = p.s
    nx, nc, neq, nk = s.nx, s.nc, p.neq, s.nk
    # budget of the experiment: 96 coefficient registers + 9 LDS quads per lane hold every word
    assert len(p.coef) <= asmx.NLREG + 4 * (asmx.LQ_END - asmx.LQ_L)
    # every L entry is applied exactly once per solve
    for sol in (p.fwd, p.bwd):
        assert sorted(o[2] for ins in sol for o in ins["ops"].values()) == list(range(s.L_p[nk]))
    for seed in range(3):
        rng = np.random.default_rng(seed)
        d = dict(x=rng.normal(size=nx), y=rng.normal(size=nc), z3=rng.normal(size=3), q=rng.normal(size=nx),
                 lo=rng.normal(size=neq), lo3=-np.full(3, 0.5), up3=np.full(3, 0.5), rho3=np.full(3, 0.1),
                 rinv3=np.full(3, 10.0), L=rng.normal(size=s.L_p[nk]) * 0.3, Dinv=rng.normal(size=nk))
        d = {k: v.astype(np.float32).astype(np.float64) for k, v in d.items()}
        V, lds = asmx.load_pair(p, d)
        e = Emit()
        asmx.body(e, p)
        asmx.simulate(e.ins, V, lds)
        x, y, z3 = asmx.reference_iteration(p, d)
        gx = np.array([V[p.xs[j] & 1, asmx.VX + (p.xs[j] >> 1)] for j in range(nx)])
        gy = np.array([V[p.zs[i] & 1, asmx.VY + (p.zs[i] >> 1)] for i in range(nc)])
        gz = np.array([V[p.zs[neq + k] & 1, asmx.VZT + (p.zs[neq + k] >> 1) - neq // 2] for k in range(3)])
        assert np.abs(gx - x).max() <= 2e-6 * np.abs(x).max()
        assert np.abs(gy - y).max() <= 2e-6 * np.abs(y).max()
        assert np.abs(gz - z3).max() <= 2e-6


def test_two_lane_stream_assembles():
    import os, shutil, subprocess, tempfile
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    if not os.path.exists(mc):
        import pytest
        pytest.skip("llvm-mc not available")
    asmx = _asmx()
    from robobee3d_amd.asmgen import Emit
    e = Emit()
    asmx.body(e, asmx.Plan())
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write("\n".join(asmx.fmt(t) for t in e.ins) + "\n")
    try:
        r = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-filetype=obj", "-o", os.devnull, f.name],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[:2000]
    finally:
        os.unlink(f.name)

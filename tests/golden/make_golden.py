#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE itself (run in the build
container only; the reference never travels and is never copied).

Sources of truth:
  * oracle/_ref/libumpc_ref.so — the reference's own C (template/uprightmpc2/*.c)
    compiled in place by oracle/Makefile, driven through ctypes (oracle/refbind.py);
  * the reference's pure-numpy Python (template/genqp.py, template_controllers.py,
    uprightmpc2.py, flight_tasks.py), imported from /root/reference/template with
    EMPTY placeholder modules for the three imports that are absent here and are
    never called on these code paths (`osqp`, `progressbar`, and `uprightmpc2py`
    whose `UprightMPC2C` is replaced by the ctypes wrapper of the compiled C).

Outputs are DATA only (inputs + expected outputs).

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import refbind  # noqa: E402

REF_T = "/root/reference/template"


def rand_rot(rng, tilt):
    from scipy.spatial.transform import Rotation
    return Rotation.from_rotvec(rng.normal(size=3) * tilt).as_matrix()


def gen_inputs(rng, n):
    ins = []
    for k in range(n):
        p0 = rng.normal(size=3) * 5.0
        R0 = rand_rot(rng, 0.4)
        dq0 = np.hstack((rng.normal(size=3) * 0.05, rng.normal(size=3) * 0.01))
        pdes = p0 + rng.normal(size=3) * 3.0
        dpdes = rng.normal(size=3) * 0.02
        sdes = np.array([0, 0, 1.0]) if k % 3 else rand_rot(rng, 0.3)[:, 2]
        aT0 = -1.0 if k % 5 else float(rng.uniform(0, 0.02))
        ins.append((p0, R0, dq0, pdes, dpdes, sdes, aT0))
    return ins


def structure():
    r = refbind.RefUMPC()
    d = {k: r.iarr(v) for k, v in dict(
        perm="linsys_solver_P", A_p="Adata_p", A_i="Adata_i", K_p="linsys_solver_KKT_p",
        K_i="linsys_solver_KKT_i", PtoKKT="linsys_solver_PtoKKT", AtoKKT="linsys_solver_AtoKKT",
        rhotoKKT="linsys_solver_rhotoKKT", etree="linsys_solver_etree", Lnz="linsys_solver_Lnz",
        L_p="linsys_solver_L_p", L_i="linsys_solver_L_i", constr_type0="work_constr_type").items()}
    d["A_x0"] = r.farr("Adata_x")
    d["P_x0"] = r.farr("Pdata_x")
    d["K_x0"] = r.farr("linsys_solver_KKT_x")
    d["l0"], d["u0"], d["q0"] = r.farr("ldata"), r.farr("udata"), r.farr("qdata")
    s = r.scaling()
    d["D0"], d["E0"], d["c0"] = s["D"], s["E"], s["c"]
    d["rho_vec0"] = r.farr("work_rho_vec")
    d["Ax_idx"] = r.struct_vectors()[5]
    np.savez_compressed(os.path.join(HERE, "structure.npz"), **d)
    print("structure.npz: nnzA=%d nnzKKT=%d nnzL=%d" % (len(d["A_i"]), len(d["K_i"]), len(d["L_i"])))


def sequence(seed, n, maxIter, fname, full_every=1, inputs=None, **ctor):
    """One controller from the pristine workspace through n calls (F1/F2)."""
    rng = np.random.default_rng(seed)
    r = refbind.RefUMPC(maxIter=maxIter, **ctor)
    rec = {k: [] for k in ("p0 R0 dq0 pdes dpdes sdes actualT0 pre_x pre_y pre_z pre_T0 pre_E3 "
                           "l u q Px Ax c D E rho_vec constr_type Lx Dinv x y z sol_x sol_y uquad accdes "
                           "status pri_res dua_res T0 ret").split()}
    for (p0, R0, dq0, pdes, dpdes, sdes, aT0) in (inputs if inputs is not None else gen_inputs(rng, n)):
        x, y, z = r.iterates()
        rec["pre_x"].append(x); rec["pre_y"].append(y); rec["pre_z"].append(z)
        rec["pre_T0"].append(np.float32(r.up.T0))
        rec["pre_E3"].append(r.scaling()["E"][36:39])
        uq, ac = r.update(p0, R0, dq0, pdes, dpdes, sdes, aT0)
        for k, v in zip(("p0", "R0", "dq0", "pdes", "dpdes", "sdes"), (p0, R0, dq0, pdes, dpdes, sdes)):
            rec[k].append(np.asarray(v, np.float32))
        rec["actualT0"].append(np.float32(aT0))
        l, u, q, Px, Ax, _ = r.struct_vectors()
        s = r.scaling()
        info = r.info()
        x, y, z = r.iterates()
        for k, v in dict(l=l, u=u, q=q, Px=Px, Ax=Ax, c=s["c"], D=s["D"], E=s["E"],
                         rho_vec=r.farr("work_rho_vec"), constr_type=r.iarr("work_constr_type"),
                         Lx=r.farr("linsys_solver_L_x"), Dinv=r.farr("linsys_solver_Dinv"),
                         x=x, y=y, z=z, sol_x=r.farr("xsolution"), sol_y=r.farr("ysolution"),
                         uquad=uq, accdes=ac, status=np.int32(info["status_val"]),
                         pri_res=info["pri_res"], dua_res=info["dua_res"], T0=np.float32(r.up.T0),
                         ret=np.int32(r.ret)).items():
            rec[k].append(v)
    out = {k: np.stack(v) for k, v in rec.items()}
    out["maxIter"] = np.int32(maxIter)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    st, cnt = np.unique(out["status"], return_counts=True)
    print(fname, "calls", n, "status histogram", dict(zip(st.tolist(), cnt.tolist())))


def nan_branch():
    """the branch of the reference that stores OSQP_NAN and cold-starts the iterates (auxil.c:539-564)
    behind a `!has_solution` status. The QP of this path is feasible and strictly convex for every input, so the
    infeasibility certificates never fire; what does reach the branch in the reference is a residual beyond
    OSQP_INFTY (osqp.c:536-541 -> OSQP_NON_CVX), e.g. from a state of 1e33. Sequence: normal calls, one such call,
    then recovery calls that override the (now garbage) thrust accumulator through actualT0."""
    rng = np.random.default_rng(14)
    ins = gen_inputs(rng, 12)
    for k, big in ((4, 1e33), (9, 1e35)):   # primal residual 5e30 and 6e31: clear of the 1e30 threshold in fp32
        p0, R0, dq0, pdes, dpdes, sdes, aT0 = ins[k]
        ins[k] = (p0 * 0 + np.array([big, -2 * big, 0.0]), R0, dq0, pdes, dpdes, sdes, aT0)
        p0, R0, dq0, pdes, dpdes, sdes, aT0 = ins[k + 1]
        ins[k + 1] = (p0, R0, dq0, pdes, dpdes, sdes, 0.0098)   # T0 <- actualT0 (uprightmpc2.c:215-216)
    sequence(0, len(ins), 50, "nan_branch.npz", inputs=ins)


def bounds_reject():
    """osqp_update_bounds' reject path (template/uprightmpc2/osqp.c:801-808): TtoWmax < 0 crosses the thrust rows'
    bounds (l = -T0 > u = Tmax - T0), every bounds update returns 1 without touching the workspace, and umpcUpdate
    (which drops that return value, uprightmpc2.c:246) keeps solving with the code-generated placeholder bounds."""
    sequence(20201117, 12, 50, "bounds_reject.npz", TtoWmax=-2.0)


def import_reference_python():
    for name in ("osqp", "progressbar", "uprightmpc2py"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["uprightmpc2py"].UprightMPC2C = object  # only the name is imported
    sys.path.insert(0, REF_T)
    import genqp, template_controllers, flight_tasks, uprightmpc2  # noqa
    return genqp, template_controllers, flight_tasks, uprightmpc2


class _CtrlForHarness:
    """What template/uprightmpc2.py:139 calls: mdl.update(p, Rb, dq, pdes, dpdes, sdes)."""

    def __init__(self):
        self.r = refbind.RefUMPC()
        self.calls = []

    def update(self, p, Rb, dq, pdes, dpdes, sdes, actualT0=-1.0):
        self.calls.append((np.array(p), np.array(Rb), np.array(dq)))
        uq, ac = self.r.update(p, Rb, dq, pdes, dpdes, sdes, actualT0)
        self.status = getattr(self, "status", []) + [self.r.info()["status_val"]]
        return uq.astype(np.float64), ac.astype(np.float64)


def closed_loop(mods):
    genqp, tc, ft, um2 = mods
    ctrl = _CtrlForHarness()
    log = um2.controlTest(ctrl, 500, useMPC=True, hlInterval=5, showPlots=False)
    fire = np.nonzero(np.any(log["accdes"] != 0, axis=1))[0]
    y = log["y"].copy()
    err, eff = um2.logMetric({k: (v.copy() if hasattr(v, "copy") else v) for k, v in log.items()})
    np.savez_compressed(os.path.join(HERE, "closed_loop_hover.npz"), t=log["t"], y=y, u=log["u"],
                        accdes=log["accdes"], pdes=log["pdes"], fire=fire.astype(np.int32),
                        ncalls=np.int32(len(ctrl.calls)), status=np.array(ctrl.status, np.int32),
                        metric=np.array([err, eff]))
    print("closed_loop_hover.npz: %d substeps, %d fires (first %d), final p=%s metric=(%.4f, %.4f)"
          % (len(log["t"]), len(ctrl.calls), fire[0], np.round(y[-1, :3], 4), err, eff))


def assembly_fp64(mods, N=3, fname="assembly_fp64.npz"):
    """F4: the pure-Python fp64 assembly for the inputs of sequence A (pins the
    '#OK' identities of template_controllers.py:321-326). N = 5 pins the general-horizon path."""
    genqp, tc, ft, um2 = mods
    seq = np.load(os.path.join(HERE, "seq_iter50.npz"))
    ny, nu = 6, 3
    nx, nc = N * (2 * ny + nu), 2 * N * ny + N
    A, P = tc.initConstraint(N, nx, nc)
    Ibi = np.diag(1 / genqp.Ib.diagonal())
    ws, wds, wpr, wvr, wpf, wvf, wthrust, wmom = 1e1, 1e3, 1, 1e3, 5, 2e3, 1e-1, 1e-2
    Wts = [np.hstack((np.full(3, wpr), np.full(3, ws))), np.hstack((np.full(3, wpf), np.full(3, ws))),
           np.hstack((np.full(3, wvr), np.full(3, wds))), np.hstack((np.full(3, wvf), np.full(3, wds))),
           np.hstack((wthrust, np.full(2, wmom)))]
    out = {k: [] for k in "l u q Px Adata Axidx".split()}
    n = 64
    for k in range(n):
        R0 = seq["R0"][k].astype(np.float64)
        dq0 = seq["dq0"][k].astype(np.float64)
        T0 = float(seq["actualT0"][k]) if seq["actualT0"][k] >= 0 else float(seq["pre_T0"][k])
        s0 = R0[:, 2].copy()
        Btau = (-R0 @ tc.e3h @ Ibi)[:, :2]
        ds0 = -R0 @ tc.e3h @ dq0[3:6]
        y0 = np.hstack((seq["p0"][k].astype(np.float64), s0))
        dy0 = np.hstack((dq0[:3], ds0))
        ydes = np.hstack((seq["pdes"][k], seq["sdes"][k])).astype(np.float64)
        dydes = np.hstack((seq["dpdes"][k].astype(np.float64), 0, 0, 0))
        A, l, u, Axidx = tc.updateConstraint(N, A, 5.0, T0, [s0] * N, [Btau] * N, y0, dy0, 9.81e-3, 2 * 9.81e-3)
        Pdata, q = tc.updateObjective(N, *Wts, ydes, dydes)
        for key, v in zip(out, (l, u, q, Pdata, A.data.copy(), Axidx)):
            out[key].append(np.array(v))
    np.savez_compressed(os.path.join(HERE, fname), n=np.int32(n),
                        A_indices=A.indices, A_indptr=A.indptr, **{k: np.stack(v) for k, v in out.items()})
    print(fname, n, "cases; A.nnz =", A.nnz)


def plant(mods):
    genqp = mods[0]
    rng = np.random.default_rng(5)
    rec = {k: [] for k in "p R dq u dt p2 R2 dq2".split()}
    for k in range(128):
        p = rng.normal(size=3) * 10
        R = rand_rot(rng, 1.0)
        dq = np.hstack((rng.normal(size=3) * 0.2, rng.normal(size=3) * (0.05 if k % 4 else 1e-5)))
        u = np.array([rng.uniform(0, 0.02), rng.normal() * 50, rng.normal() * 50])
        dt = [0.2, 0.1, 1.0, 5.0][k % 4]
        p2, R2, dq2 = genqp.quadrotorNLDyn(p, R, dq, u, dt)
        for key, v in zip(rec, (p, R, dq, u, dt, p2, R2, dq2)):
            rec[key].append(np.asarray(v, np.float64))
    np.savez_compressed(os.path.join(HERE, "plant.npz"), **{k: np.stack(v) for k, v in rec.items()})
    print("plant.npz: 128 cases")


def models():
    """the reference's own ca6 wrench map and (M, h) terms (template/ca6dynamics.py:35-50) and its
    ThrustStrokeDev vector field (template/FlappingModels3D.py:19-38), evaluated by IMPORTING those two modules from
    where they lie. What they import and this image lacks computes nothing here:
      * `autograd.numpy` is numpy's function set wrapped for differentiation -- forward values are numpy's own; the
        name is bound to the installed numpy (no gradient is taken by the functions called);
      * `controlutils.py.model.Model` is only the base class of ThrustStrokeDev (an un-vendored submodule; `dynamics`
        uses nothing of it) and `controlutils.py.kinematics` is imported but unused: empty placeholders;
      * `Rotation.as_dcm` is scipy's pre-1.4 name of `Rotation.as_matrix` (same function, renamed; removed in 1.6):
        answered with `as_matrix` while the fixture is generated.
    The inputs are the cases of tests/test_models.py."""
    import numpy
    from scipy.spatial.transform import Rotation
    ag = types.ModuleType("autograd")
    ag.numpy = numpy
    sys.modules.setdefault("autograd", ag)
    sys.modules.setdefault("autograd.numpy", numpy)
    cu, cupy, cumodel, cukin = (types.ModuleType(n) for n in ("controlutils", "controlutils.py", "controlutils.py.model",
                                                               "controlutils.py.kinematics"))
    cumodel.Model = type("Model", (), {})
    cu.py, cupy.model, cupy.kinematics = cupy, cumodel, cukin
    for m in (cu, cupy, cumodel, cukin):
        sys.modules.setdefault(m.__name__, m)
    sys.path.insert(0, REF_T)
    import ca6dynamics
    import FlappingModels3D
    if not hasattr(Rotation, "as_dcm"):
        # (the scipy type is immutable: the module's own name `Rotation` is pointed at a forwarder whose rotation objects
        # answer `as_dcm()` with scipy's `as_matrix()`)
        class _OldNameRotation:
            @staticmethod
            def from_rotvec(v):
                return types.SimpleNamespace(as_dcm=Rotation.from_rotvec(v).as_matrix)
        FlappingModels3D.Rotation = _OldNameRotation
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    from test_models import _cases
    n = 64
    y6, u6, y12, u4 = _cases(n, 1)
    w = np.stack([ca6dynamics.wrenchMap(u6[k]) for k in range(n)])
    Ms, hs = [], []
    for k in range(n):
        Rb = y6[k, 3:12].reshape(3, 3).T                              # column-major in the state vector
        q = np.hstack((y6[k, :3], Rotation.from_matrix(Rb).as_quat()))
        M, h = ca6dynamics.dynamicsTerms(q, y6[k, 12:])
        Ms.append(np.array(M, float)); hs.append(np.array(h, float))
    tsd = FlappingModels3D.ThrustStrokeDev()
    yd = np.stack([tsd.dynamics(y12[k], u4[k]) for k in range(n)])
    np.savez_compressed(os.path.join(HERE, "models.npz"), y6=y6, u6=u6, y12=y12, u4=u4, ca6_w=w, ca6_M=np.stack(Ms),
                        ca6_h=np.stack(hs), tsd_ydot=yd, ca6_const=np.array([ca6dynamics.ycp, ca6dynamics.mb, ca6dynamics.g]),
                        tsd_const=np.array([tsd.m, tsd.ycp, FlappingModels3D.g]), tsd_Ib=np.array(tsd.Ib, float))
    print("models.npz", w.shape, yd.shape)


def tasks(mods):
    ft = mods[2]
    ts = np.linspace(0, 1200, 49)
    p0 = np.array([1.0, -2.0, 3.0])
    out = {}
    for name, fn, kw in (("hover", ft.helix, dict(trajAmp=0, trajFreq=0, dz=0.1, useY=False)),
                         ("helix", ft.helix, dict(trajAmp=80, trajFreq=1, dz=0.15, useY=True)),
                         ("straightAcc", ft.straightAcc, {}), ("flip", ft.flip, {}), ("perch", ft.perch, {})):
        out[name] = np.stack([np.hstack(fn(t, p0, **kw)) for t in ts])
    np.savez_compressed(os.path.join(HERE, "flight_tasks.npz"), t=ts, p0=p0, **out)
    print("flight_tasks.npz")


def wl_step():
    """§8f-1: wlConInit / wlConUpdate of the reference (template/uprightmpc2/funapprox.c:118-165)
    with the constructor arguments and the fitted wrench-map coefficients of WaypointHover
    (template/robobee_test_controllers.py:71-76), read from the reference's source text as DATA."""
    import ctypes as C
    import re
    src = open(os.path.join(REF_T, "robobee_test_controllers.py")).read()
    m = re.search(r"\n\s+popts = \[(.*?)\]", src, re.S)
    popts = np.array([float(v) for v in m.group(1).replace("\n", " ").split(",")], np.float32)
    assert popts.shape == (90,)
    r = refbind.RefUMPC(do_init=False)
    wl = (C.c_float * 184)()  # WLCon_t: u0,umin,umax,dumax (16) + Qw (36) + 6 x {k,a0,a1[4],A2[16]} (132)
    f = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    u0 = np.array([140.0, 0, 0, 0], np.float32)
    umin, umax = np.array([90, -0.5, -0.2, -0.1], np.float32), np.array([240, 0.5, 0.2, 0.1], np.float32)
    dumax, Qw = np.array([5e3, 10, 10, 10], np.float32), np.array([1, 1, 1, 0.1, 0.1, 0.1], np.float32)
    r.lib.wlConInit(wl, f(u0), f(umin), f(umax), f(dumax), f(Qw), C.c_float(1000.0), f(popts))
    rng = np.random.default_rng(77)
    rec = {k: [] for k in "pre_u0 h0 pdotdes u1 w0".split()}
    M0 = np.array([100, 100, 100, 3333, 3333, 1000.0])
    for k in range(96):
        Rb = rand_rot(rng, 0.3)
        h0 = np.hstack((Rb.T @ np.array([0, 0, 100 * 9.81e-3]), np.zeros(3))).astype(np.float32)
        acc = np.hstack((rng.normal(size=3) * 5e-3, rng.normal(size=3) * 2e-4))
        pd = (M0 * acc).astype(np.float32)
        if k % 17 == 16:
            pd = pd * 50  # drive the rate / box limits
        rec["pre_u0"].append(np.array(wl[0:4], np.float32))
        u1, w0 = np.zeros(4, np.float32), np.zeros(6, np.float32)
        r.lib.wlConUpdate(wl, f(u1), f(w0), f(h0), f(pd))
        for key, v in zip(("h0", "pdotdes", "u1", "w0"), (h0, pd, u1, w0)):
            rec[key].append(np.array(v))
    np.savez_compressed(os.path.join(HERE, "wl_step.npz"), popts=popts, u0=u0, umin=umin, umax=umax, dumax=dumax,
                        Qw=Qw, controlRate=np.float32(1000.0), **{k: np.stack(v) for k, v in rec.items()})
    print("wl_step.npz: 96 updates, final u =", np.array(wl[0:4]))



def mpc_wl_loop(mods):
    """SURVEY 8f-1, the coupling: accController of template/robobee_test_controllers.py:162-171 (= conn_MPC_WL.m:2-10)
    around the COMPILED reference -- per call: (uquad, accdes) = umpcUpdate(..., actualT0); h0 = (Rb'(0,0,mb g), 0),
    pdotdes = M0 accdes with M0 = diag(100,100,100,3333,3333,1000) (dynamicsTerms, template/ca6dynamics.py:5-10,
    44-50, read as text: the module needs autograd); (u4, w0) = wlConUpdate(h0, pdotdes); actualT0 = w0[2]/M0[2,2].
    Between calls the state advances like controlTest (template/uprightmpc2.py:148-151): 25 substeps of the
    reference's quadrotorNLDyn (template/genqp.py:32-41), moments clipped at +-100. WL constructor arguments and
    popts as in wl_step()."""
    import ctypes as C
    import re
    from scipy.spatial.transform import Rotation
    genqp = mods[0]
    src = open(os.path.join(REF_T, "robobee_test_controllers.py")).read()
    m = re.search(r"\n\s+popts = \[(.*?)\]", src, re.S)
    popts = np.array([float(v) for v in m.group(1).replace("\n", " ").split(",")], np.float32)
    r = refbind.RefUMPC()
    wl = (C.c_float * 184)()
    f = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    u0 = np.array([140.0, 0, 0, 0], np.float32)
    umin, umax = np.array([90, -0.5, -0.2, -0.1], np.float32), np.array([240, 0.5, 0.2, 0.1], np.float32)
    dumax, Qw = np.array([5e3, 10, 10, 10], np.float32), np.array([1, 1, 1, 0.1, 0.1, 0.1], np.float32)
    r.lib.wlConInit(wl, f(u0), f(umin), f(umax), f(dumax), f(Qw), C.c_float(1000.0), f(popts))
    M0 = np.array([100, 100, 100, 3333, 3333, 1000.0])
    mb, g = 100.0, 9.81e-3
    # hover start of controlTest (template/uprightmpc2.py:101-103)
    p, Rb = np.zeros(3), Rotation.from_euler("xyz", [0.5, -0.5, 0]).as_matrix()
    dq = np.zeros(6); dq[0] = 0.1
    pdes, dpdes, sdes = np.zeros(3), np.zeros(3), np.array([0, 0, 1.0])
    aT0 = -1.0
    rec = {k: [] for k in "p0 R0 dq0 actualT0 pre_x pre_y pre_z pre_T0 pre_E3 pre_u4 uquad accdes h0 pdotdes u4 w0 T0 status".split()}
    for k in range(64):
        x, y, z = r.iterates()
        for key, v in dict(p0=p, R0=Rb, dq0=dq, actualT0=aT0, pre_x=x, pre_y=y, pre_z=z, pre_T0=np.float32(r.up.T0),
                           pre_E3=r.scaling()["E"][36:39], pre_u4=np.array(wl[0:4], np.float32)).items():
            rec[key].append(np.array(v))
        uq, ac = r.update(p, Rb, dq, pdes, dpdes, sdes, aT0)
        h0 = np.hstack((np.asarray(Rb, np.float32).T @ np.array([0, 0, np.float32(mb) * np.float32(g)], np.float32),
                        np.zeros(3, np.float32))).astype(np.float32)
        pd = (M0.astype(np.float32) * ac).astype(np.float32)
        u1, w0 = np.zeros(4, np.float32), np.zeros(6, np.float32)
        r.lib.wlConUpdate(wl, f(u1), f(w0), f(h0), f(pd))
        aT0 = float(np.float32(w0[2]) / np.float32(M0[2]))
        for key, v in dict(uquad=uq, accdes=ac, h0=h0, pdotdes=pd, u4=u1, w0=w0, T0=np.float32(r.up.T0),
                           status=np.int32(r.info()["status_val"])).items():
            rec[key].append(np.array(v))
        u = uq.astype(np.float64)
        u[1:] = np.clip(u[1:], -100, 100)
        for _ in range(25):
            p, Rb, dq = genqp.quadrotorNLDyn(p, Rb, dq, u, 0.2)
    np.savez_compressed(os.path.join(HERE, "mpc_wl_loop.npz"), popts=popts, u0=u0, umin=umin, umax=umax, dumax=dumax,
                        Qw=Qw, controlRate=np.float32(1000.0), Mdiag=M0, final_p=p, final_R=Rb, final_dq=dq,
                        **{k: np.stack(v) for k, v in rec.items()})
    print("mpc_wl_loop.npz: 64 coupled calls; actualT0 range %.5f..%.5f, final |p| = %.4f mm, u4 = %s"
          % (min(rec["actualT0"][1:]), max(rec["actualT0"][1:]), np.linalg.norm(p), np.array(wl[0:4])))


class _RecordingOSQP:
    """Placeholder for the absent pip `osqp` on the two assembly-only paths below: records what the reference
    hands to setup()/update() and computes NOTHING; solve() aborts the caller."""

    class Stop(Exception):
        pass

    def __init__(self):
        self.setup_args, self.updates = None, []

    def setup(self, *a, **kw):
        self.setup_args = (a, kw)

    def update(self, **kw):
        self.updates.append({k: np.array(v) for k, v in kw.items()})

    def solve(self):
        raise _RecordingOSQP.Stop()


def v1_qp(mods):
    """the data genqp.UprightMPC.update (template/genqp.py:132-158) assembles, for random
    arguments, N = 3; plus its dynamics / dynamicsNLVF one-liners (:170-187)."""
    genqp = mods[0]
    sys.modules["osqp"].OSQP = _RecordingOSQP
    rng = np.random.default_rng(22)
    N = 3
    up = genqp.UprightMPC(N)
    rec = {k: [] for k in "q0 qdes Qf Rd smin smax dt snom vT0 Pdata Adata q l u dyn_u dyn_out nlvf_out".split()}
    for _ in range(32):
        q0 = np.hstack((rng.normal(size=3), rng.normal(size=3) * 0.2 + np.array([0, 0, 1.0])))
        qdes = np.hstack((rng.normal(size=3), [0, 0, 1.0]))
        Qf = rng.uniform(0.5, 50, 6)
        Rd = rng.uniform(0.1, 5, 3)
        smin = -rng.uniform(1.2, 2, 3)
        smax = rng.uniform(1.2, 2, 3)
        dt = float(rng.uniform(1, 5))
        snom = [rng.normal(size=3) * 0.2 + np.array([0, 0, 1.0]) for _ in range(N)]
        vT0 = float(rng.normal() * 0.05)
        try:
            up.update(q0, qdes, Qf, Rd, smin, smax, dt, snom, vT0)
        except _RecordingOSQP.Stop:
            pass
        uu = rng.normal(size=3)
        for k, v in dict(q0=q0, qdes=qdes, Qf=Qf, Rd=Rd, smin=smin, smax=smax, dt=dt, snom=np.hstack(snom), vT0=vT0,
                         Pdata=up.P.data, Adata=up.A.data, q=up.q, l=up.l, u=up.u, dyn_u=uu,
                         dyn_out=up.dynamics(q0, uu, dt, snom[0], vT0), nlvf_out=up.dynamicsNLVF(q0, uu)).items():
            rec[k].append(np.array(v, dtype=np.float64))
    np.savez_compressed(os.path.join(HERE, "v1_qp.npz"), N=N, A_indices=up.A.indices, A_indptr=up.A.indptr,
                        P_indices=up.P.indices, P_indptr=up.P.indptr, Axidx=np.array(up.Axidx),
                        **{k: np.stack(v) for k, v in rec.items()})
    print("v1_qp.npz: nnzA=%d nnzP=%d" % (up.A.nnz, up.P.nnz))


def planar_p5f():
    """planar/mpc_osqp_p5f.py is a script whose module body stops with a ValueError under this
    scipy (block_diag is handed a list, :116), so it cannot be imported whole. Its model constants (:33-43) and
    getLin (:45-85) are self-contained: those statements alone are compiled from the file where it lies and
    evaluated. Fixture = getLin samples + the constants."""
    import ast
    path = "/root/reference/planar/mpc_osqp_p5f.py"
    tree = ast.parse(open(path).read(), path)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "getLin":
            keep.append(node)
            break
        if isinstance(node, ast.Assign) and all(isinstance(t, ast.Name) for t in node.targets):
            keep.append(node)
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    rng = np.random.default_rng(21)
    us, sg, ph = rng.normal(size=64) * 10, rng.normal(size=64) * 0.3, rng.normal(size=64) * 0.5
    us[:4] = [0.0, -0.0, 1e-3, -1e-3]
    Ads, Bds = [], []
    for a, b, c in zip(us, sg, ph):
        Ad, Bd = ns["getLin"](a, b, c)
        Ads.append(np.asarray(Ad))
        Bds.append(np.asarray(Bd).ravel())
    # F6: the QP data of :87-147. Every top-level statement between getLin and the tick loop is executed where it
    # lies, one at a time; the three that cannot run here are skipped and reported: `P = sp.linalg.block_diag([..])`
    # (:116, ValueError under this scipy: a list is passed), `prob = osqp.OSQP()` and `prob.setup(...)` (pip osqp
    # absent). P's diagonal follows from the evaluated Q, QN, R (kron(eye(N), Q) | QN | kron(eye(N), R), :116-117).
    import scipy as sp
    import scipy.linalg  # noqa: F401
    ns2 = dict(ns)
    ns2.update(sp=sp, sys=sys)
    seen_getlin, skipped = False, []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "getLin":
            seen_getlin = True
            continue
        if not seen_getlin or isinstance(node, (ast.Import, ast.ImportFrom)):
            continue
        if isinstance(node, ast.For):
            break
        try:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns2)
        except Exception as ex:
            skipped.append("line %d: %s" % (node.lineno, type(ex).__name__))
    qp = {k: np.asarray(ns2[k], np.float64) for k in ("Q", "QN", "R", "y0", "yr", "q", "leq", "ueq", "lineq", "uineq",
                                                        "l", "u", "Aineq", "Ax", "Bu", "Aeq", "A", "xmin", "xmax",
                                                        "umin", "umax")}
    assert "P" not in ns2 and len(skipped) == 3, skipped
    np.savez_compressed(os.path.join(HERE, "planar_p5f.npz"), dt=ns["dt"], tf=ns["tf"], mb=ns["mb"], ib=ns["ib"],
                        lin_u=us, lin_sigma=sg, lin_phi=ph, lin_Ad=np.stack(Ads), lin_Bd=np.stack(Bds),
                        N=np.int32(ns2["N"]), nx=np.int32(ns2["nx"]), nu=np.int32(ns2["nu"]),
                        skipped=np.array(skipped), **{"qp_" + k: v for k, v in qp.items()})
    print("  F6: q %s l %s u %s A %s; skipped %s" % (qp["q"].shape, qp["l"].shape, qp["u"].shape, qp["A"].shape, skipped))
    print("planar_p5f.npz: %d getLin samples; Ad non-zeros at" % len(us),
          sorted(set(zip(*[a.tolist() for a in np.nonzero(np.abs(np.stack(Ads)).sum(0))]))))



def planar_p5f_stroke():
    """SURVEY 8(f-4)'s SECOND planar structure: planar/mpc_osqp_p5f_stroke.py (stroke model: nx = 7, nu = 2 inputs (u, tf),
    N = 1 => n = 16, m = 30; LTV blocks from getLin :38-86; input box umin = (-10, 0.1), umax = (10, 100), states free). The
    script needs pip `osqp` (absent) at import and solves with it (eps 1e-2, :196), so its solves are unpinnable; what it
    COMPUTES without osqp is pinned here by executing its own statements where they lie, one at a time: getLin, the QP
    data :139-193 (P through scipy.sparse.block_diag, which drops Q's zero diagonal entries), and for every tick of its loop
    :204-221 the nominal-input bookkeeping, the open-loop state, and getCondensed's (A, l, u) -- the statements of the loop
    body up to `prob.update`, in the script's own namespace (ympc is never advanced: the script's own update line :247 is
    commented out). Skipped statements are recorded."""
    import ast
    import scipy as sp
    import scipy.linalg  # noqa: F401
    import scipy.sparse as sparse
    path = "/root/reference/planar/mpc_osqp_p5f_stroke.py"
    tree = ast.parse(open(path).read(), path)
    ns = {"np": np, "sp": sp, "sparse": sparse, "sys": sys}
    skipped, loop = [], None
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            continue
        if isinstance(node, ast.For):
            loop = node
            break
        try:
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
        except Exception as ex:
            skipped.append("line %d: %s" % (node.lineno, type(ex).__name__))
    assert loop is not None and len(skipped) == 2, skipped          # prob = osqp.OSQP(); prob.setup(...)
    # the loop body up to (not including) the first statement that touches `prob`
    body = []
    for node in loop.body:
        if "prob" in {n_.id for n_ in ast.walk(node) if isinstance(n_, ast.Name)}:
            break
        body.append(node)
    assert isinstance(body[-1], ast.Assign) and "getCondensed(" in ast.unparse(body[-1])
    rec = {k: [] for k in ("u0", "tf0", "y_prev", "ympc_prev", "y_ol", "A", "l", "u", "Ad0", "Bd0")}
    for ti in ns["tvec"]:
        ns["ti"] = ti
        try:
            for node in body:
                if isinstance(node, ast.If) and ast.unparse(node.test) == "ti == 0":
                    if ti == 0:
                        raise StopIteration      # the script's `continue`
                    continue
                exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
        except StopIteration:
            continue
        rec["u0"].append(float(ns["u0"])); rec["tf0"].append(float(ns["tf0"]))
        rec["y_prev"].append(np.array(ns["y"][ti - 1], np.float64)); rec["ympc_prev"].append(np.array(ns["ympc"][ti - 1], np.float64))
        rec["y_ol"].append(np.array(ns["y"][ti], np.float64).ravel())
        rec["A"].append(np.asarray(ns["A"], np.float64)); rec["l"].append(np.array(ns["l"], np.float64)); rec["u"].append(np.array(ns["u"], np.float64))
        rec["Ad0"].append(np.asarray(ns["Ad0"], np.float64)); rec["Bd0"].append(np.asarray(ns["Bd0"], np.float64))
    # more getLin samples, away from the script's own trajectory (random inputs, both signs of u0, u0 = 0)
    rng = np.random.default_rng(22)
    lu0 = np.concatenate(([0.0, 5.0, -5.0], rng.normal(size=45) * 8))
    ltf = np.concatenate(([0.0, 5.0, 5.0], rng.uniform(0.1, 20, 45)))
    ly = rng.normal(size=(48, 7)) * np.array([10, 1, 1, 0.5, 1, 1, 1])
    lAd, lBd = [], []
    for a, b, c in zip(lu0, ltf, ly):
        Ad, Bd = ns["getLin"](a, b, c)
        lAd.append(np.asarray(Ad, np.float64)); lBd.append(np.asarray(Bd, np.float64))
    P = ns["P"].tocsc()
    np.savez_compressed(os.path.join(HERE, "planar_p5f_stroke.npz"), N=np.int32(ns["N"]), nx=np.int32(ns["nx"]), nu=np.int32(ns["nu"]),
                        P_indices=P.indices, P_indptr=P.indptr, P_data=P.data, q=np.asarray(ns["q"], np.float64),
                        umin=ns["umin"], umax=ns["umax"], A_setup=np.asarray(ns["A"].todense() if hasattr(ns["A"], "todense") else ns["A"], np.float64) if False else
                        np.asarray(sparse.vstack([ns["Aeq"], ns["Aineq"]]).todense(), np.float64),
                        lin_u0=lu0, lin_tf0=ltf, lin_y=ly, lin_Ad=np.stack(lAd), lin_Bd=np.stack(lBd),
                        skipped=np.array(skipped), **{"tick_" + k: np.stack(v) for k, v in rec.items()})
    print("planar_p5f_stroke.npz: %d ticks, A %s, P nnz %d, skipped %s" % (len(rec["A"]), rec["A"][0].shape, P.nnz, skipped))


def planar_code():
    """The reference's SECOND generated controller, planar/code (emosqp of OSQP 0.5.0, EMBEDDED 1, DFLOAT, n = 46,
    m = 82, scaling 0, rho 5.694, check_termination 25, max_iter 50: planar/code/include/workspace.h:1068-1071; the output of
    planar/mpc_thrust_strokedev.py:199-200 for its 'hover' parameters :31), compiled
    where it lies (oracle/Makefile, oracle/planar_ref_host.cpp) and driven the way its host does (planar/mcuqp/main.cpp:
    131: osqp_solve on the workspace, warm start carried) plus the vector updates of its API (osqp.c:756,785). A box
    MPC in the (Aeq; I) form of planar/mpc_osqp.py:84-100 -- the family of config 4 (planar/mpc_osqp_p5f.py:120-128).
    Pins the general-structure solver (oracle/osqp_table.py, then the product's batch QP) on a second structure,
    permutation and settings. Row classes stay as generated (EMBEDDED 1 never re-types rho_vec: auxil.c, `#if EMBEDDED
    != 1` around update_rho_vec in osqp_update_bounds), so the updates keep loose rows loose and equalities equal."""
    import planarbind
    rng = np.random.default_rng(20201121)
    r0 = planarbind.PlanarRef()
    rec = dict(n=r0.n, m=r0.m, A_p=r0.A_p, A_i=r0.A_i, P_i=r0.P_i, perm=r0.perm, L_p=r0.L_p, L_i=r0.L_i, P_x=r0.P_x, A_x=r0.A_x,
               q0=r0.q, l0=r0.l, u0=r0.u, rho_vec=r0.rho_vec, L_x=r0.L_x, Dinv=r0.Dinv,
               settings=np.array([r0.rho, r0.sigma, r0.alpha, r0.eps_abs, r0.eps_rel, r0.eps_prim_inf, r0.eps_dual_inf], np.float32),
               isettings=np.array([r0.max_iter, r0.check_termination, r0.scaling, r0.warm_start, r0.scaled_termination], np.int32))
    calls = []

    def call(r, ctrl, chk, **kw):
        pre = dict(x0=kw.pop("wx"), y0=kw.pop("wy"), z0=kw.pop("wz"))
        o = r.solve(**kw)
        calls.append(dict(ctrl=ctrl, chk=chk, q=kw.get("q", r0.q), l=kw.get("l", r0.l), u=kw.get("u", r0.u), **pre,
                          **{k: o[k] for k in ("sol_x", "sol_y", "x", "y", "z", "pri_res", "dua_res", "status", "iter", "rc_update")}))
        return o

    zero = lambda: dict(wx=np.zeros(r0.n, np.float32), wy=np.zeros(r0.m, np.float32), wz=np.zeros(r0.m, np.float32))
    carry = lambda o: dict(wx=o["x"], wy=o["y"], wz=o["z"])
    # controller A: what main.cpp does -- osqp_solve on the generated vectors, four times
    w = zero()
    for k in range(4):
        o = call(r0, 0, 25, **w)
        w = carry(o)
    # controller B: 36 calls with new vectors (state, cost, the finite boxes), warm start carried; every sixth call
    # with check_termination 0 (exactly 50 iterations, the setting of the uprightmpc2 path)
    rB = planarbind.PlanarRef()
    fin = np.where((np.abs(r0.l) < 1e19) & (r0.l != r0.u))[0]          # rows 38, 44, ..., 68 (tilt) and 72..81 (inputs)
    w = zero()
    for k in range(36):
        chk = 0 if k % 6 == 5 else 25
        rB.set(50, chk)
        a = 0.02 if k < 18 else 1.0                      # small moves (warm starts that converge) and large ones
        q = (r0.q * (1 + 0.5 * a * rng.normal(size=r0.n))).astype(np.float32)
        l, u = r0.l.copy(), r0.u.copy()
        l[:6] = u[:6] = (r0.l[:6] * (1 + 0.4 * a * rng.normal(size=6)) + 1e-3 * a * rng.normal(size=6)).astype(np.float32)
        lo = np.abs(rng.normal(size=len(fin))) * np.where(fin < 72, 4e-3, 6e-2)
        hi = np.abs(rng.normal(size=len(fin))) * np.where(fin < 72, 4e-3, 6e-2)
        if k % 3:                                         # two calls in three with boxes tight enough to bind
            l[fin], u[fin] = (-lo).astype(np.float32), hi.astype(np.float32)
        o = call(rB, 1, chk, q=q, l=l, u=u, **w)
        w = carry(o)
    # controller C: a tilt box that contradicts the dynamics -> primal infeasibility certificate, OSQP_NAN in the
    # solution and a cold start (auxil.c store_solution), then a feasible call from that cold start
    rC = planarbind.PlanarRef()
    l, u = r0.l.copy(), r0.u.copy()
    l[44], u[44] = 1.0, 2.0
    w = zero()
    for k in range(3):
        o = call(rC, 2, 25, l=l, u=u, **w)
        w = carry(o)
    o = call(rC, 2, 25, l=r0.l.copy(), u=r0.u.copy(), **w)
    for k in calls[0]:
        rec["c_" + k] = np.array([c[k] for c in calls])
    st = rec["c_status"]
    print("planar_code: %d calls; status counts %s; iterations %s" % (len(calls), {int(v): int((st == v).sum()) for v in np.unique(st)},
          {int(v): int((rec["c_iter"] == v).sum()) for v in np.unique(rec["c_iter"])}))
    np.savez_compressed(os.path.join(HERE, "planar_code.npz"), **rec)


def reactive(mods):
    """reactiveController samples (template_controllers.py:282-296) and the reference's own
    controlTest(None, 100, useMPC=False, taulim=10) log (uprightmpc2.py:87-159, the call of gainTuningSims :297)."""
    genqp, tc, ft, um2 = mods
    rng = np.random.default_rng(33)
    rec = {k: [] for k in "p R dq pdes k u".split()}
    for _ in range(64):
        p, R, dq = rng.normal(size=3) * 20, rand_rot(rng, 0.6), np.hstack((rng.normal(size=3) * 0.3, rng.normal(size=3) * 0.02))
        pdes = p + rng.normal(size=3) * 50
        k = rng.uniform(0.2, 3, 6) * np.array([5e-3, 5e-1, 1e-1, 1e0, 10e0, 1e2])
        u = tc.reactiveController(p, R, dq, pdes, kpos=list(k[0:2]), kz=list(k[2:4]), ks=list(k[4:6]))
        for key, v in zip(rec, (p, R, dq, pdes, k, u)):
            rec[key].append(np.array(v, np.float64))
    log = um2.controlTest(None, 100, useMPC=False, showPlots=False, taulim=10, ks=[15.0, 1.2e2])
    err, eff = um2.logMetric({k: (v.copy() if hasattr(v, "copy") else v) for k, v in log.items()})
    np.savez_compressed(os.path.join(HERE, "reactive.npz"), log_t=log["t"], log_y=log["y"], log_u=log["u"],
                        log_ks=np.array([15.0, 1.2e2]), metric=np.array([err, eff]),
                        **{k: np.stack(v) for k, v in rec.items()})
    print("reactive.npz: 64 samples, log", log["y"].shape, "metric", err, eff)


if __name__ == "__main__":
    assert refbind.available(), "build oracle/_ref first: make -C oracle ref"
    if len(sys.argv) > 1 and sys.argv[1] == "wlloop":
        mpc_wl_loop(import_reference_python())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "nan":
        nan_branch()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "models":
        models()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "reject":
        bounds_reject()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "planarcode":
        planar_code()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wl":
        wl_step()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "reactive":
        reactive(import_reference_python())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "qp":
        assembly_fp64(import_reference_python(), 5, "assembly_fp64_N5.npz")
        v1_qp(import_reference_python())
        planar_p5f()
        planar_p5f_stroke()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stroke":
        planar_p5f_stroke()
        sys.exit(0)
    structure()
    sequence(20201117, 256, 50, "seq_iter50.npz")
    for k in (1, 2, 10):
        sequence(20201117 + k, 24, k, "seq_iter%d.npz" % k)
    mods = import_reference_python()
    closed_loop(mods)
    assembly_fp64(mods)
    plant(mods)
    tasks(mods)
    wl_step()
    assembly_fp64(mods, 5, "assembly_fp64_N5.npz")
    reactive(mods)
    v1_qp(mods)
    planar_p5f()
    planar_p5f_stroke()
    nan_branch()
    bounds_reject()
    models()
    planar_code()
    mpc_wl_loop(mods)

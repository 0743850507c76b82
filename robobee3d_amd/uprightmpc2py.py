"""Drop-in for the reference's pybind11 module `uprightmpc2py`
(template/uprightmpc2/py/uprightmpc2py.cpp:30-52): same class name, same
constructor arguments, same `update / vectors / matrices` methods and return
shapes, backed by libumpc_mi355x.so's umpcInit / umpcUpdate (the reference's own
C symbols, here executing on the MI355X).

    from robobee3d_amd.uprightmpc2py import UprightMPC2C
    upc = UprightMPC2C(dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, Ib, 50)
    uquad, accdes = upc.update(p0, R0, dq0, pdes, dpdes, sdes, actualT0)

`actualT0` defaults to -1 because the reference's own harness calls update with
six arguments (template/uprightmpc2.py:139).

Two bindings of the same C symbols: the COMPILED module `_uprightmpc2py` (csrc/uprightmpc2py_ext.cpp, pybind11 like the
reference's own extension; built in-tree by _lib.build()) is what `UprightMPC2C` / `WLCon` name when it is present; the
ctypes classes below (`UprightMPC2C_ctypes`, `WLCon_ctypes`) are the portable binding and what is used under
UMPC_LIB (A/B builds of the library) or UMPC_PY_BINDING=ctypes. `binding()` says which one is active.
"""
import ctypes as C
import os

import numpy as np

from . import _lib


COMPAT_BOUNDS_REJECT = 1     # include/umpc_mi355x.h UMPC_COMPAT_BOUNDS_REJECT


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class UprightMPC2C_ctypes:
    def __init__(self, dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, Ib, maxIter):
        self._L = _lib.lib()
        self.umpc = _lib.UprightMPC_t()
        ib = np.ascontiguousarray(Ib, np.float32)
        assert ib.shape == (3,)
        self._L.umpcInit(C.byref(self.umpc), *[C.c_float(v) for v in
                         (dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom)], _fp(ib), C.c_int(maxIter))

    def __del__(self):
        try:
            self._L.umpcRelease(C.byref(self.umpc))
        except Exception:
            pass

    def update(self, p0, R0, dq0, pdes, dpdes, sdes, actualT0=-1.0):
        f = lambda a, n: np.ascontiguousarray(np.asarray(a, np.float32).reshape(n))
        p0, dq0, pdes, dpdes, sdes = f(p0, 3), f(dq0, 6), f(pdes, 3), f(dpdes, 3), f(sdes, 3)
        # pybind11/eigen converts a row-major numpy 3x3 into a column-major Matrix3f
        R0c = np.ascontiguousarray(np.asarray(R0, np.float32).reshape(3, 3).T.ravel())
        uquad = np.zeros(3, np.float32)
        accdes = np.zeros(6, np.float32)
        rc = self._L.umpcUpdate(C.byref(self.umpc), _fp(uquad), _fp(accdes), _fp(p0), _fp(R0c), _fp(dq0),
                                _fp(pdes), _fp(dpdes), _fp(sdes), C.c_float(actualT0))
        if rc:
            raise RuntimeError("umpcUpdate failed (no GPU / not initialised)")
        return uquad, accdes

    def vectors(self):
        u = self.umpc
        return (np.array(u.l, np.float32), np.array(u.u, np.float32), np.array(u.q, np.float32))

    def matrices(self):
        u = self.umpc
        return (np.array(u.Px_data, np.float32), np.array(u.Ax_data, np.float32), np.array(u.Ax_idx, np.int32))

    def status(self):
        """OSQP status code of the last update (extension; the reference drops it)."""
        return int(self._L.umpcLastStatus(C.byref(self.umpc)))

    def set_compat(self, flags):
        """umpcSetCompat (extension): opt-in reference compatibility switches, e.g. COMPAT_BOUNDS_REJECT = the reference's
        handling of crossed bounds (osqp.c:801-808 + uprightmpc2.c:246). Returns the previous flags."""
        rc = int(self._L.umpcSetCompat(C.byref(self.umpc), C.c_int(int(flags))))
        if rc < 0:
            raise RuntimeError("umpcSetCompat: no live controller")
        return rc


class WLCon_ctypes:
    """Mirror of the pybind `WLCon` class (template/uprightmpc2/py/uprightmpc2py.cpp:54-68):
    WLCon(u0, umin, umax, dumax, Qw, controlRate, popts).update(h0, pdotdes) -> (u1[4], w0[6]),
    over the reference's own C symbols wlConInit / wlConUpdate (funapprox.h:43-45) in libumpc_mi355x.so."""

    def __init__(self, u0, umin, umax, dumax, Qw, controlRate, popts):
        self._L = _lib.lib()
        self.wl = _lib.WLCon_t()
        f = lambda a, n: np.ascontiguousarray(np.asarray(a, np.float32).reshape(n))
        self._L.wlConInit(C.byref(self.wl), _fp(f(u0, 4)), _fp(f(umin, 4)), _fp(f(umax, 4)), _fp(f(dumax, 4)),
                          _fp(f(Qw, 6)), C.c_float(controlRate), _fp(f(popts, 90)))

    def update(self, h0, pdotdes):
        f = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).reshape(6))
        u1, w0 = np.zeros(4, np.float32), np.zeros(6, np.float32)
        self._L.wlConUpdate(C.byref(self.wl), _fp(u1), _fp(w0), _fp(f(h0)), _fp(f(pdotdes)))
        return u1, w0


try:
    if os.environ.get("UMPC_LIB") or os.environ.get("UMPC_PY_BINDING") == "ctypes":
        raise ImportError("ctypes binding requested")
    from . import _uprightmpc2py as _native
except ImportError:
    _native = None
UprightMPC2C = _native.UprightMPC2C if _native is not None else UprightMPC2C_ctypes
WLCon = _native.WLCon if _native is not None else WLCon_ctypes


def binding():
    """"pybind11" (the compiled module) or "ctypes": which binding `UprightMPC2C` / `WLCon` are"""
    return "pybind11" if _native is not None else "ctypes"


class UprightMPC2:
    """Twin of the pure-Python class template_controllers.UprightMPC2(N, dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf,
    wthrust, wmom, Ib) (template/template_controllers.py:170-258): any horizon N, fp64, one robot, on the
    general-structure solver (robobee3d_amd.batchqp.UprightMPC2N with B = 1).

    Solve semantics = the reference's: `osqp.OSQP().setup(..., eps_rel=1e-4, eps_abs=1e-4)` + `solve()`
    (template_controllers.py:190-191, 216-219), i.e. pip-osqp defaults -- iterate until the termination criteria
    hold at eps 1e-4, tested every 25 iterations, at most 4000, rho adapted (adapt_rho, auxil.c:62-82) -- and a
    message when the status is not "solved" (:218-219). pip osqp derives its rho-adaptation interval from wall-clock
    timings (a multiple of 25); 25 is used here, so the iterate path is ONE of the reference's possible paths and
    the returned solution is the QP optimum to eps either way.
    maxIter = k gives the embedded C path's semantics instead (exactly k iterations, uprightmpc2.c:116-117)."""

    def __init__(self, N, dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, Ib, maxIter=None):
        import torch
        from .batchqp import UprightMPC2N
        self.N, self._torch = N, torch
        st = dict(max_iter=maxIter) if maxIter is not None else dict(max_iter=4000, check_termination=25, adaptive_rho_interval=25)
        self._mpc = UprightMPC2N(1, N, dt=dt, g=g, TtoWmax=TtoWmax, ws=ws, wds=wds, wpr=wpr, wpf=wpf, wvr=wvr, wvf=wvf,
                                 wthrust=wthrust, wmom=wmom, Ib=tuple(float(v) for v in Ib), dtype=torch.float64, **st)

    @property
    def T0(self):
        return float(self._mpc.T0[0].item())

    def update(self, p0, R0, dq0, pdes, dpdes, sdes, actualT0=-1.0):
        torch = self._torch
        st = np.concatenate((np.asarray(p0, np.float64).ravel(), np.asarray(R0, np.float64).reshape(3, 3).T.ravel(),
                             np.asarray(dq0, np.float64).ravel()))[:, None]
        rf = np.concatenate((np.asarray(pdes, np.float64).ravel(), np.asarray(dpdes, np.float64).ravel(),
                             np.asarray(sdes, np.float64).ravel()))[:, None]
        dev = self._mpc.dev
        aT0 = torch.full((1,), float(actualT0), dtype=torch.float64, device=dev)
        out = self._mpc.update(torch.as_tensor(np.ascontiguousarray(st)).to(dev), torch.as_tensor(np.ascontiguousarray(rf)).to(dev),
                               aT0).cpu().numpy()[:, 0]
        self.prevsol = self._mpc.qp.sol_x.cpu().numpy()[:, 0]
        self.status_val = int(self._mpc.qp.status[0].item())
        self.iterations = int(self._mpc.qp.info[4, 0].item())
        if self.status_val not in (1, 2):     # template_controllers.py:218-219
            print({-2: "maximum iterations reached", -3: "primal infeasible", 3: "primal infeasible inaccurate",
                   -4: "dual infeasible", 4: "dual infeasible inaccurate", -7: "problem non convex"}.get(
                       self.status_val, "status %d" % self.status_val))
        return out[0:3].copy(), out[3:9].copy()


def createMPC(N=3, ws=1e1, wds=1e3, wpr=1, wvr=1e3, wpf=5, wvf=2e3, wthrust=1e-1, wmom=1e-2, TtoWmax=2, **kwargs):
    """template/template_controllers.py:260-280, same defaults and the same (pyver, cver) return pair: the Python
    twin at horizon N and the compiled-C twin (horizon 3, uprightmpc2.h:20)."""
    dt, g = 5, 9.81e-3
    Ib = np.array([3333.0, 3333.0, 1000.0])
    up = UprightMPC2(N, dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, Ib)
    upc = UprightMPC2C(dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, Ib, 50)
    return up, upc

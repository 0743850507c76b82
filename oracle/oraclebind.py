"""TEST INFRASTRUCTURE — ctypes view of oracle/liboracle_f32.so / liboracle_f64.so
(the CPU restatement in oracle/umpc_oracle.c). Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NX, NC, NK, NCTRL = 45, 39, 84, 127

DEFAULTS = dict(dt=5.0, g=9.81e-3, TtoWmax=2.0, ws=1e1, wds=1e3, wpr=1.0, wpf=5.0,
                wvr=1e3, wvf=2e3, wthrust=1e-1, wmom=1e-2)  # createMPC, template_controllers.py:260-263
IB = (3333.0, 3333.0, 1000.0)  # template/genqp.py:22


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    need = force or not all(os.path.exists(os.path.join(HERE, f))
                            for f in ("liboracle_f32.so", "liboracle_f64.so"))
    if need or not os.path.exists(os.path.join(HERE, "_ref", "libumpc_ref.so")):
        subprocess.run(["make", "-s", "-C", HERE, "all"], check=True,
                       stdout=subprocess.DEVNULL)


_libs = {}


def lib(dtype):
    dtype = np.dtype(dtype)
    key = "f32" if dtype == np.float32 else "f64"
    if key not in _libs:
        path = os.path.join(HERE, "liboracle_%s.so" % key)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.umpc_oracle_sizeof.restype = C.c_size_t
        L.umpc_oracle_get.restype = C.c_void_p
        L.umpc_oracle_update.restype = C.c_int
        assert L.umpc_oracle_real_size() == dtype.itemsize
        _libs[key] = L
    return _libs[key]


class Params(C.Structure):
    pass


def _params_struct(ct):
    class P(C.Structure):
        _fields_ = [(k, ct) for k in ("dt", "g", "TtoWmax", "ws", "wds", "wpr", "wpf", "wvr",
                                      "wvf", "wthrust", "wmom")] + \
                   [("Ib", ct * 3), ("maxIter", C.c_int), ("dtsim", ct), ("taulim", ct),
                    ("nsub", C.c_int), ("plant_mode", C.c_int)]
    return P


class Oracle:
    """One controller instance (umpcInit/umpcUpdate twin)."""

    def __init__(self, dtype=np.float32, perm=None, Ib=IB, maxIter=50, **kw):
        self.dtype = np.dtype(dtype)
        self.ct = C.c_float if self.dtype == np.float32 else C.c_double
        self.L = lib(self.dtype)
        self.buf = C.create_string_buffer(self.L.umpc_oracle_sizeof())
        prm = dict(DEFAULTS)
        prm.update(kw)
        ib = np.asarray(Ib, self.dtype)
        pp = None if perm is None else np.ascontiguousarray(perm, np.int32)
        self.L.umpc_oracle_init(self.buf, *[self.ct(prm[k]) for k in DEFAULTS], self._p(ib),
                                C.c_int(maxIter),
                                None if pp is None else pp.ctypes.data_as(C.POINTER(C.c_int)))

    def _p(self, a):
        return a.ctypes.data_as(C.POINTER(self.ct))

    def update(self, p0, R0, dq0, pdes, dpdes, sdes, actualT0=-1.0):
        f = lambda a: np.ascontiguousarray(np.asarray(a, self.dtype).ravel())
        p0, dq0, pdes, dpdes, sdes = map(f, (p0, dq0, pdes, dpdes, sdes))
        R0c = np.ascontiguousarray(np.asarray(R0, self.dtype).T.ravel())  # -> column-major
        uquad = np.zeros(3, self.dtype)
        accdes = np.zeros(6, self.dtype)
        self.ret = self.L.umpc_oracle_update(self.buf, self._p(uquad), self._p(accdes), self._p(p0),
                                             self._p(R0c), self._p(dq0), self._p(pdes),
                                             self._p(dpdes), self._p(sdes), self.ct(actualT0))
        return uquad, accdes

    def get(self, name):
        n, isint = C.c_int(0), C.c_int(0)
        ptr = self.L.umpc_oracle_get(self.buf, name.encode(), C.byref(n), C.byref(isint))
        if not ptr:
            raise KeyError(name)
        ty = C.c_int if isint.value else self.ct
        arr = np.array(C.cast(ptr, C.POINTER(ty * n.value)).contents)
        return arr.astype(np.int32 if isint.value else self.dtype)

    def set_iterates(self, x, y, z):
        x, y, z = (np.ascontiguousarray(v, self.dtype) for v in (x, y, z))
        self.L.umpc_oracle_set_iterates(self.buf, self._p(x), self._p(y), self._p(z))

    def set_T0(self, T0):
        self.L.umpc_oracle_set_T0(self.buf, self.ct(T0))

    def set_max_iter(self, k):
        self.L.umpc_oracle_set_max_iter(self.buf, C.c_int(k))

    def set_canonical(self, on=True, Eprev3=None):
        e = None if Eprev3 is None else np.ascontiguousarray(Eprev3, self.dtype)
        self.L.umpc_oracle_set_canonical(self.buf, C.c_int(int(on)), None if e is None else self._p(e))


def plant_step(p, R, dq, u, dt, Ib=IB, gain=1.0, mode=0, dtype=np.float64, force_double=False):
    """R is a 3x3 matrix (row-major numpy); returns new (p, R, dq)."""
    dtype = np.dtype(dtype)
    L = lib(np.float32 if (dtype == np.float32) else np.float64)
    ct = C.c_float if dtype == np.float32 else C.c_double
    f = lambda a: np.array(np.asarray(a, dtype).ravel())
    p, dq, u, ib = f(p), f(dq), f(u), f(Ib)
    Rc = np.array(np.asarray(R, dtype).T.ravel())
    P = lambda a: a.ctypes.data_as(C.POINTER(ct))
    fn = L.umpc_oracle_plant_step
    fn(P(p), P(Rc), P(dq), P(u), ct(dt), P(ib), ct(gain), C.c_int(mode))
    return p, Rc.reshape(3, 3).T.copy(), dq


def plant_step_d(L32, p, R, dq, u, dt, Ib=IB, gain=1.0, mode=0):
    """fp64 plant exported by either library (umpc_oracle_plant_step_d)."""
    f = lambda a: np.array(np.asarray(a, np.float64).ravel())
    p, dq, u, ib = f(p), f(dq), f(u), f(Ib)
    Rc = np.array(np.asarray(R, np.float64).T.ravel())
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    L32.umpc_oracle_plant_step_d(P(p), P(Rc), P(dq), P(u), C.c_double(dt), P(ib),
                                 C.c_double(gain), C.c_int(mode))
    return p, Rc.reshape(3, 3).T.copy(), dq


def batch_rollout(state, ctrl, ref, K, dtype=np.float32, perm=None, Ib=None, gain=None,
                  Ib_nom=IB, maxIter=50, dtsim=0.2, taulim=100.0, nsub=25, plant_mode=0,
                  nthreads=0, weights=None, task=0, task_p=None, t0=0.0, wl=None, wl_u=None, wl_w=None,
                  Mdiag=(100.0, 100.0, 100.0, 3333.0, 3333.0, 1000.0), **kw):
    """SoA arrays state[18,B], ctrl[127,B], ref[9,B] (modified in place).
    wl: a WLOracle (parameters) switches the MPC -> WL -> actualT0 coupling on; wl_u [4,B] is the per-robot WL
    state (in/out), wl_w [6,B] receives w0 of the last step.
    Returns (out[9,B], stats[2,B], status[B])."""
    dtype = np.dtype(dtype)
    L = lib(dtype)
    ct = C.c_float if dtype == np.float32 else C.c_double
    PS = _params_struct(ct)
    prm = dict(DEFAULTS)
    prm.update(kw)
    ps = PS()
    for k in DEFAULTS:
        setattr(ps, k, prm[k])
    for i in range(3):
        ps.Ib[i] = Ib_nom[i]
    ps.maxIter, ps.dtsim, ps.taulim, ps.nsub, ps.plant_mode = maxIter, dtsim, taulim, nsub, plant_mode
    B = state.shape[1]
    for a, rows in ((state, 18), (ctrl, NCTRL), (ref, 9)):
        assert a.dtype == dtype and a.shape == (rows, B) and a.flags.c_contiguous
    out = np.zeros((9, B), dtype)
    stats = np.zeros((2, B), dtype)
    status = np.zeros(B, np.int32)
    P = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(ct))
    pp = None if perm is None else np.ascontiguousarray(perm, np.int32)
    if Ib is not None:
        Ib = np.ascontiguousarray(Ib, dtype)
        assert Ib.shape == (3, B)
    if gain is not None:
        gain = np.ascontiguousarray(gain, dtype)
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype)
        assert weights.shape == (8, B)
    tp = np.zeros(4, dtype)
    if task_p is not None:
        tp[:len(task_p)] = task_p
    md = np.ascontiguousarray(Mdiag, dtype)
    if wl is not None:
        assert wl.dtype == dtype and wl_u.dtype == dtype and wl_u.shape == (4, B) and wl_u.flags.c_contiguous
        assert wl_w is None or (wl_w.dtype == dtype and wl_w.shape == (6, B) and wl_w.flags.c_contiguous)
    L.umpc_oracle_batch_rollout3(C.byref(ps), None if pp is None else pp.ctypes.data_as(C.POINTER(C.c_int)),
                                 C.c_int(B), C.c_int(K), P(state), P(ctrl), P(ref), P(Ib), P(gain), P(weights),
                                 C.c_int(task), P(tp), ct(t0), P(out), P(stats),
                                 status.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(nthreads),
                                 None if wl is None else wl.buf, P(md), P(wl_u), P(wl_w))
    return out, stats, status


def reactive(p, R, dq, pdes, k=(5e-3, 5e-1, 1e-1, 1e0, 10e0, 1e2), dtype=np.float64):
    """reactiveController (template/template_controllers.py:282-296) as restated in the oracle; R is a 3x3 matrix."""
    dtype = np.dtype(dtype)
    L = lib(dtype)
    ct = C.c_float if dtype == np.float32 else C.c_double
    f = lambda a: np.ascontiguousarray(np.asarray(a, dtype).ravel())
    P = lambda a: a.ctypes.data_as(C.POINTER(ct))
    u = np.zeros(3, dtype)
    L.umpc_oracle_reactive(P(f(p)), P(f(np.asarray(R).T)), P(f(dq)), P(f(pdes)), P(f(k)), P(u))
    return u


def reactive_rollout(state, ref, nsteps, every=1, gains=None, dtype=np.float64, Ib=None, gain=None, Ib_nom=IB,
                     dtsim=0.2, taulim=100.0, plant_mode=0, task=0, task_p=None, t0=0.0, log_robot=-1):
    """controlTest(useMPC=False) twin: state [18,B] modified in place; returns (out[3,B], stats[2,B], log or None)."""
    dtype = np.dtype(dtype)
    L = lib(dtype)
    ct = C.c_float if dtype == np.float32 else C.c_double
    ps = _params_struct(ct)()
    for k in DEFAULTS:
        setattr(ps, k, DEFAULTS[k])
    for i in range(3):
        ps.Ib[i] = Ib_nom[i]
    ps.maxIter, ps.dtsim, ps.taulim, ps.nsub, ps.plant_mode = 50, dtsim, taulim, 25, plant_mode
    B = state.shape[1]
    assert state.dtype == dtype and state.flags.c_contiguous and ref.dtype == dtype and ref.flags.c_contiguous
    P = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(ct))
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype)
    gains, Ib, gain = c(gains), c(Ib), c(gain)
    out, stats = np.zeros((3, B), dtype), np.zeros((2, B), dtype)
    log = np.zeros((nsteps, 15), dtype) if log_robot >= 0 else None
    tp = np.zeros(4, dtype)
    if task_p is not None:
        tp[:len(task_p)] = task_p
    L.umpc_oracle_reactive_rollout(C.byref(ps), C.c_int(B), C.c_int(nsteps), C.c_int(every), P(state), P(ref), P(gains),
                                   P(Ib), P(gain), C.c_int(task), P(tp), ct(t0), P(out), P(stats), P(log),
                                   C.c_int(log_robot))
    return out, stats, log


def task_reference(task, task_p, t, initial_pos, dtype=np.float64):
    """(pdes, dpdes, sdes) of template/flight_tasks.py as restated in the oracle."""
    dtype = np.dtype(dtype)
    L = lib(dtype)
    ct = C.c_float if dtype == np.float32 else C.c_double
    tp = np.zeros(4, dtype)
    tp[:len(task_p)] = task_p
    r = np.zeros(9, dtype)
    r[:3] = initial_pos
    L.umpc_oracle_task_reference(C.c_int(task), tp.ctypes.data_as(C.POINTER(ct)), ct(t),
                                 r.ctypes.data_as(C.POINTER(ct)))
    return r


class WLOracle:
    """wlConInit / wlConUpdate twin (oracle/umpc_oracle.c, funapprox.c restated)."""

    def __init__(self, u0, umin, umax, dumax, Qw, controlRate, popts, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.ct = C.c_float if self.dtype == np.float32 else C.c_double
        self.L = lib(self.dtype)
        self.L.umpc_oracle_wl_sizeof.restype = C.c_size_t
        self.buf = C.create_string_buffer(self.L.umpc_oracle_wl_sizeof())
        a = lambda v, n: np.ascontiguousarray(np.asarray(v, self.dtype).reshape(n))
        P = lambda v: v.ctypes.data_as(C.POINTER(self.ct))
        self._keep = [a(u0, 4), a(umin, 4), a(umax, 4), a(dumax, 4), a(Qw, 6), a(popts, 90)]
        k = self._keep
        self.L.umpc_oracle_wl_init(self.buf, P(k[0]), P(k[1]), P(k[2]), P(k[3]), P(k[4]), self.ct(controlRate), P(k[5]))

    def set_u0(self, u0):
        u0 = np.ascontiguousarray(u0, self.dtype)
        self.L.umpc_oracle_wl_set_u0(self.buf, u0.ctypes.data_as(C.POINTER(self.ct)))

    def update(self, h0, pdotdes):
        P = lambda v: v.ctypes.data_as(C.POINTER(self.ct))
        h0 = np.ascontiguousarray(h0, self.dtype)
        pd = np.ascontiguousarray(pdotdes, self.dtype)
        u1, w0 = np.zeros(4, self.dtype), np.zeros(6, self.dtype)
        self.L.umpc_oracle_wl_update(self.buf, P(u1), P(w0), P(h0), P(pd))
        return u1, w0

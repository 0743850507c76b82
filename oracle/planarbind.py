"""TEST INFRASTRUCTURE -- ctypes view of oracle/_ref/libplanar_ref.so: the reference's generated planar controller
(planar/code: emosqp of OSQP 0.5.0, EMBEDDED 1, fp32, n = 46, m = 82, box-constrained MPC in the (Aeq; I) form of
planar/mpc_osqp.py:84-100) compiled where it lies by oracle/Makefile, behind oracle/planar_ref_host.cpp. One loaded
image == one workspace (its globals carry the warm start), so every PlanarRef() loads a private copy.
Used by tests/golden/make_golden.py (tests/golden/planar_code.npz) and the CPU tests that pin oracle/osqp_table.py
on this second structure."""
import ctypes as C
import os
import shutil
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libplanar_ref.so")
_f, _i = C.POINTER(C.c_float), C.POINTER(C.c_int)


def available():
    return os.path.exists(REF_SO)


def _fp(a):
    return None if a is None else a.ctypes.data_as(_f)


class PlanarRef:
    def __init__(self):
        self._tmp = tempfile.mkdtemp(prefix="planar_ref_")
        path = os.path.join(self._tmp, "libplanar_ref.so")
        shutil.copy(REF_SO, path)
        self.lib = lib = C.CDLL(path)
        d = np.zeros(16, np.int32)
        lib.planar_dims(d.ctypes.data_as(_i))
        (self.n, self.m, self.nnzP, self.nnzA, self.nk, self.nnzL, self.max_iter, self.check_termination, self.scaling,
         self.warm_start, self.scaled_termination) = (int(v) for v in d[:11])
        s = np.zeros(8, np.float32)
        lib.planar_settings(_fp(s))
        self.rho, self.sigma, self.eps_abs, self.eps_rel, self.eps_prim_inf, self.eps_dual_inf, self.alpha = s[:7]
        ia = lambda k: np.zeros(k, np.int32)
        fa = lambda k: np.zeros(k, np.float32)
        self.P_p, self.P_i, self.A_p, self.A_i = ia(self.n + 1), ia(self.nnzP), ia(self.n + 1), ia(self.nnzA)
        self.perm, self.L_p, self.L_i = ia(self.nk), ia(self.nk + 1), ia(self.nnzL)
        lib.planar_pattern(*(a.ctypes.data_as(_i) for a in (self.P_p, self.P_i, self.A_p, self.A_i, self.perm, self.L_p,
                                                            self.L_i)))
        self.P_x, self.A_x, self.q, self.l, self.u = fa(self.nnzP), fa(self.nnzA), fa(self.n), fa(self.m), fa(self.m)
        self.rho_vec, self.L_x, self.Dinv = fa(self.m), fa(self.nnzL), fa(self.nk)
        lib.planar_values(*(_fp(a) for a in (self.P_x, self.A_x, self.q, self.l, self.u, self.rho_vec, self.L_x,
                                             self.Dinv)))

    def set(self, max_iter, check_termination):
        return self.lib.planar_set(int(max_iter), int(check_termination))

    def solve(self, q=None, l=None, u=None, x0=None, y0=None, z0=None):
        c = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        q, l, u, x0, y0, z0 = (c(a) for a in (q, l, u, x0, y0, z0))
        sx, sy = np.zeros(self.n, np.float32), np.zeros(self.m, np.float32)
        wx, wy, wz = np.zeros(self.n, np.float32), np.zeros(self.m, np.float32), np.zeros(self.m, np.float32)
        inf, st = np.zeros(3, np.float32), np.zeros(4, np.int32)
        self.lib.planar_solve(_fp(q), _fp(l), _fp(u), _fp(x0), _fp(y0), _fp(z0), _fp(sx), _fp(sy), _fp(wx), _fp(wy),
                              _fp(wz), _fp(inf), st.ctypes.data_as(_i))
        return dict(sol_x=sx, sol_y=sy, x=wx, y=wy, z=wz, pri_res=inf[0], dua_res=inf[1], obj_val=inf[2],
                    status=int(st[0]), iter=int(st[1]), rc_update=int(st[2]), rc_solve=int(st[3]))

    def __del__(self):
        shutil.rmtree(getattr(self, "_tmp", ""), ignore_errors=True)

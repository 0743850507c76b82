"""TEST INFRASTRUCTURE — ctypes view of oracle/_ref/libumpc_ref.so.

That library is the REFERENCE's own C (template/uprightmpc2/*.c, OSQP 0.6.0
embedded, fp32) compiled in place by oracle/Makefile in the authoring
container (the reference's sources exist only there); the prebuilt, git-ignored
file travels to the GPU box with the tree, so `available()` may be true there
as well (nothing under /root/reference is read at run time).  This module is used by
tools/make_golden.py to generate tests/golden/*.npz and by the CPU tests that
pin oracle/umpc_oracle.c against the live reference when it is present.

The reference keeps ONE global OSQP `workspace` (template/uprightmpc2/
workspace.c:2620) and file-static scratch, so one loaded image == one
controller.  `RefUMPC()` therefore dlopen()s a private temp copy of the .so per
instance to get pristine globals.
"""
import ctypes as C
import os
import shutil
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libumpc_ref.so")

NX, NC, NKKT, NNZA, NNZL, NNZKKT, NADATA = 45, 39, 84, 111, 213, 195, 48


def available():
    return os.path.exists(REF_SO)


class UprightMPC_t(C.Structure):
    # layout: template/uprightmpc2/uprightmpc2.h:27-43
    _fields_ = [
        ("dt", C.c_float), ("g", C.c_float), ("Tmax", C.c_float),
        ("Qyr", C.c_float * 6), ("Qyf", C.c_float * 6),
        ("Qdyr", C.c_float * 6), ("Qdyf", C.c_float * 6), ("R", C.c_float * 3),
        ("smin", C.c_float * 3), ("smax", C.c_float * 3),
        ("e3h", C.c_float * 9), ("e3hIbi", C.c_float * 9),
        ("l", C.c_float * NC), ("u", C.c_float * NC), ("q", C.c_float * NX),
        ("Px_data", C.c_float * NX), ("Ax_data", C.c_float * NADATA),
        ("Ax_idx", C.c_int * NADATA), ("nAxT0dt", C.c_int), ("nAxdt", C.c_int),
        ("c0", C.c_float * 6), ("T0", C.c_float),
    ]


class _Scaling(C.Structure):
    # template/uprightmpc2/types.h OSQPScaling
    _fields_ = [("c", C.c_float), ("D", C.POINTER(C.c_float)),
                ("E", C.POINTER(C.c_float)), ("cinv", C.c_float),
                ("Dinv", C.POINTER(C.c_float)), ("Einv", C.POINTER(C.c_float))]


class _Info(C.Structure):
    # template/uprightmpc2/types.h OSQPInfo with EMBEDDED=2, no PROFILING
    _fields_ = [("iter", C.c_int), ("status", C.c_char * 32),
                ("status_val", C.c_int), ("obj_val", C.c_float),
                ("pri_res", C.c_float), ("dua_res", C.c_float),
                ("rho_updates", C.c_int), ("rho_estimate", C.c_float)]


_F = {  # exported float arrays (template/uprightmpc2/workspace.h)
    "Pdata_x": 45, "Adata_x": 111, "qdata": 45, "ldata": 39, "udata": 39,
    "Dscaling": 45, "Dinvscaling": 45, "Escaling": 39, "Einvscaling": 39,
    "linsys_solver_L_x": 213, "linsys_solver_Dinv": 84, "linsys_solver_D": 84,
    "linsys_solver_KKT_x": 195, "linsys_solver_rho_inv_vec": 39,
    "work_rho_vec": 39, "work_rho_inv_vec": 39,
    "work_x": 45, "work_y": 39, "work_z": 39, "work_xz_tilde": 84,
    "work_x_prev": 45, "work_z_prev": 39, "xsolution": 45, "ysolution": 39,
}
_I = {  # exported int arrays
    "Pdata_i": 45, "Pdata_p": 46, "Adata_i": 111, "Adata_p": 46,
    "linsys_solver_L_i": 213, "linsys_solver_L_p": 85, "linsys_solver_P": 84,
    "linsys_solver_KKT_i": 195, "linsys_solver_KKT_p": 85,
    "linsys_solver_PtoKKT": 45, "linsys_solver_AtoKKT": 111,
    "linsys_solver_rhotoKKT": 39, "linsys_solver_Pdiag_idx": 45,
    "linsys_solver_etree": 84, "linsys_solver_Lnz": 84, "work_constr_type": 39,
}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class RefUMPC:
    """One private image of the reference controller (fp32)."""

    def __init__(self, dt=5.0, g=9.81e-3, TtoWmax=2.0, ws=1e1, wds=1e3, wpr=1.0,
                 wpf=5.0, wvr=1e3, wvf=2e3, wthrust=1e-1, wmom=1e-2,
                 Ib=(3333.0, 3333.0, 1000.0), maxIter=50, do_init=True):
        if not available():
            raise RuntimeError("oracle/_ref/libumpc_ref.so not built (reference absent?)")
        fd, self._tmp = tempfile.mkstemp(suffix=".so", prefix="umpc_ref_")
        os.close(fd)
        shutil.copyfile(REF_SO, self._tmp)
        self.lib = C.CDLL(self._tmp)
        os.unlink(self._tmp)  # image stays mapped
        self.up = UprightMPC_t()
        self.lib.umpcUpdate.restype = C.c_int
        # work_x/y/z/x_prev/z_prev are reached through swapped POINTERS in
        # `workspace`; read them through the struct, not by array name.
        if do_init:
            ib = np.asarray(Ib, np.float32)
            self.lib.umpcInit(C.byref(self.up), *[C.c_float(v) for v in
                              (dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom)],
                              _fp(ib), C.c_int(maxIter))

    # -- the boundary -------------------------------------------------
    def update(self, p0, R0, dq0, pdes, dpdes, sdes, actualT0=-1.0):
        """R0 is a 3x3 matrix (numpy row-major); the C ABI wants column-major
        (template/uprightmpc2/uprightmpc2.c:219)."""
        f = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).ravel())
        p0, dq0, pdes, dpdes, sdes = map(f, (p0, dq0, pdes, dpdes, sdes))
        R0c = np.ascontiguousarray(np.asarray(R0, np.float32).T.ravel())
        uquad = np.zeros(3, np.float32)
        accdes = np.zeros(6, np.float32)
        ret = self.lib.umpcUpdate(C.byref(self.up), _fp(uquad), _fp(accdes), _fp(p0),
                                  _fp(R0c), _fp(dq0), _fp(pdes), _fp(dpdes), _fp(sdes),
                                  C.c_float(actualT0))
        self.ret = ret
        return uquad, accdes

    # -- inspection ---------------------------------------------------
    def farr(self, name):
        return np.array((C.c_float * _F[name]).in_dll(self.lib, name), np.float32)

    def iarr(self, name):
        return np.array((C.c_int * _I[name]).in_dll(self.lib, name), np.int32)

    def _wsptr(self, idx, n):
        # OSQPWorkspace (EMBEDDED=2) is a struct of pointers; slots:
        # 0 data,1 linsys,2 rho_vec,3 rho_inv_vec,4 constr_type,5 x,6 y,7 z,
        # 8 xz_tilde,9 x_prev,10 z_prev,...  (template/uprightmpc2/types.h)
        ws = (C.c_void_p * 26).in_dll(self.lib, "workspace")
        return np.array(C.cast(ws[idx], C.POINTER(C.c_float * n)).contents, np.float32)

    def iterates(self):
        return self._wsptr(5, NX), self._wsptr(6, NC), self._wsptr(7, NC)

    def set_iterates(self, x, y, z):
        ws = (C.c_void_p * 26).in_dll(self.lib, "workspace")
        for idx, v, n in ((5, x, NX), (6, y, NC), (7, z, NC)):
            dst = C.cast(ws[idx], C.POINTER(C.c_float * n)).contents
            dst[:] = list(np.asarray(v, np.float32))

    def scaling(self):
        s = _Scaling.in_dll(self.lib, "scaling")
        return dict(c=np.float32(s.c), cinv=np.float32(s.cinv), D=self.farr("Dscaling"),
                    E=self.farr("Escaling"), Dinv=self.farr("Dinvscaling"),
                    Einv=self.farr("Einvscaling"))

    def info(self):
        i = _Info.in_dll(self.lib, "info")
        return dict(iter=i.iter, status_val=i.status_val, pri_res=np.float32(i.pri_res),
                    dua_res=np.float32(i.dua_res), obj_val=np.float32(i.obj_val))

    def set_max_iter(self, k):
        ws = (C.c_void_p * 26).in_dll(self.lib, "workspace")
        self.lib.osqp_update_max_iter(C.c_void_p(C.addressof(ws)), C.c_int(k))

    def struct_vectors(self):
        u = self.up
        return (np.array(u.l, np.float32), np.array(u.u, np.float32), np.array(u.q, np.float32),
                np.array(u.Px_data, np.float32), np.array(u.Ax_data, np.float32),
                np.array(u.Ax_idx, np.int32))

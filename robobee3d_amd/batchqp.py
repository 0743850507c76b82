"""Host side of the general-structure batch QP solver (csrc/umpc_bqp.hip, include/umpc_mi355x.h Part 5) and the
reference MPC formulations built on it:

  BatchQP          B independent QPs of one sparsity structure; mirrors the osqp.OSQP() setup / update / solve
                   usage of the reference's Python MPCs (template/genqp.py:206-209,157-158;
                   planar/mpc_osqp_p5f.py:131-147,172) with the embedded-C step of template/uprightmpc2/.
  PlanarP5fMPC     planar/mpc_osqp_p5f.py: stroke-plane MPC, nx = 7, nu = 1, N = 10 (config 4).
  PlanarP5fStrokeMPC  planar/mpc_osqp_p5f_stroke.py: the stroke model, nx = 7, nu = 2, LTV blocks, N = 1 in the script.
  UprightMPCv1     template/genqp.py:43-168: the v1 template QP (nq = 6, nu = 3).

torch is device memory / streams only; every computation is a kernel of libumpc_mi355x.so. Arrays are SoA
[rows, B], robot index fastest. There is no CPU path.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, qpstruct

_DT = {torch.float32: _lib.UMPC_F32, torch.float64: _lib.UMPC_F64}
OSQP_INFTY = 1e30   # the osqp Python wrapper clips +-inf bounds to this before setup


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchQP:
    def __init__(self, n, m, A_p, A_i, P_cols, B, dtype=torch.float32, device="cuda", perm=None, **settings):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchQP needs a HIP device; there is no CPU path")
        self.L = _lib.lib()
        self.s = qpstruct.analyse_qp(n, m, A_p, A_i, P_cols, perm=perm)
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.n, self.m, self.B, self.dtype, self.device = n, m, int(B), dtype, dev
        st = _lib.QPSettings()
        self.L.umpcQPDefaultSettings(C.byref(st))
        for k, v in settings.items():
            if not hasattr(st, k):
                raise TypeError("unknown setting %r" % k)
            setattr(st, k, v)
        self.settings = st
        blob = np.ascontiguousarray(self.s.blob, np.int32)
        with torch.cuda.device(self.device):
            self.h = self.L.umpcQPCreate(blob.ctypes.data_as(C.c_void_p), int(blob.size), self.B, _DT[dtype],
                                         C.byref(st))
        if not self.h:
            raise RuntimeError(self.L.umpcLastError().decode())
        z = lambda r, dt=dtype: torch.zeros((max(r, 1), self.B), dtype=dt, device=self.device)
        self.x, self.y, self.z = z(n), z(m), z(m)
        self.Eprev = torch.ones((m, self.B), dtype=dtype, device=self.device)
        self.sol_x, self.sol_y, self.info = z(n), z(m), z(6)
        self.status = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        if os.environ.get("UMPC_QP_KERNEL"):      # diagnostics: wave | lane | tables
            self.set_kernel(os.environ["UMPC_QP_KERNEL"])

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.L.umpcQPDestroy(h)

    KERNELS = {"wave": 0, "lane": 1, "tables": 2, "lane_cpp": 3}

    def set_kernel(self, mode):
        """"lane" (default; one lane per robot: the build-time specialisation if the structure has one, else the tables),
        "tables" (lane per robot, tables), "wave" (one wavefront per robot, working set in LDS, level-scheduled),
        "lane_cpp" ("lane" without the assembly loop of the fp32 p5f specialisation)."""
        if self.L.umpcQPSetKernel(self.h, self.KERNELS[mode]) != 0:
            raise RuntimeError(self.L.umpcLastError().decode())

    def use_tables(self, on=True):
        """Force the table-driven lane-per-robot kernel (True) or the lane-per-robot specialisation (False)."""
        self.set_kernel("tables" if on else "lane")

    @property
    def kernel_name(self):
        return self.L.umpcQPKernelName(self.h).decode()

    def reset(self):
        """Cold start (x = y = z = 0) and E = 1, the state of a freshly set-up solver."""
        for t in (self.x, self.y, self.z):
            t.zero_()
        self.Eprev.fill_(1)

    def _chk(self, t, rows, name):
        if t is None:
            if rows == 0:
                return None
            raise ValueError("%s is required" % name)
        if t.dtype != self.dtype or t.device != self.device or tuple(t.shape) != (rows, self.B) or not t.is_contiguous():
            raise ValueError("%s must be a contiguous [%d, %d] %s tensor on %s" % (name, rows, self.B, self.dtype, self.device))
        return t

    def set_termination(self, check_every=25, max_iter=4000, adaptive_rho_interval=0):
        """pip-osqp semantics (template_controllers.py:190-191,216-219): test the termination criteria at the exact
        tolerances every `check_every` iterations and stop a robot when one is met; adapt rho every
        `adaptive_rho_interval` iterations (0 = fixed rho). check_every = 0 restores the embedded reference's fixed
        iteration count. `self.info[4]` / `[5]` report the iterations each robot ran and its rho updates."""
        if self.L.umpcQPSetCheckTermination(self.h, int(check_every)) != 0 or \
                self.L.umpcQPSetMaxIter(self.h, int(max_iter)) != 0 or \
                self.L.umpcQPSetAdaptiveRho(self.h, int(adaptive_rho_interval)) != 0:
            raise RuntimeError(self.L.umpcLastError().decode())

    def solve(self, Pv, Av, q, l, u, max_iter=None):
        """One canonical-restart step on raw data; returns (sol_x, sol_y, status) (views of this object's buffers)."""
        s = self.s
        Pv = self._chk(Pv, s.nnzP, "Pv"); Av = self._chk(Av, s.nnzA, "Av")
        q = self._chk(q, s.n, "q"); l = self._chk(l, s.m, "l"); u = self._chk(u, s.m, "u")
        if max_iter is not None:
            self.L.umpcQPSetMaxIter(self.h, int(max_iter))
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self.L.umpcQPSolve(self.h, _ptr(Pv), _ptr(Av), _ptr(q), _ptr(l), _ptr(u), _ptr(self.x), _ptr(self.y),
                                _ptr(self.z), _ptr(self.Eprev), _ptr(self.sol_x), _ptr(self.sol_y),
                                _ptr(self.status), _ptr(self.info), stream)
        if rc != 0:
            raise RuntimeError(self.L.umpcLastError().decode())
        return self.sol_x, self.sol_y, self.status

    def gather(self, cst, src, par, out, update=False):
        """out[k] = cst[k] if src[k] < 0 else cst[k] * par[src[k]] (umpcQPGather); update: only the entries with
        src[k] >= 0 -- `out` holds the constants from an earlier full gather (umpcQPGatherUpdate)."""
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        fn = self.L.umpcQPGatherUpdate if update else self.L.umpcQPGather
        rc = fn(self.B, _DT[self.dtype], int(cst.numel()), _ptr(cst), _ptr(src), _ptr(par), _ptr(out), stream)
        if rc != 0:
            raise RuntimeError(self.L.umpcLastError().decode())
        return out


# ---------------------------------------------------------------------------------------------------------
# planar/mpc_osqp_p5f.py
# ---------------------------------------------------------------------------------------------------------
qp_components = qpstruct.qp_components


def p5f_analysis(N=10):
    """(p5f_structure(N, grouped=True), its qpstruct.analyse_qp): what PlanarP5fMPC and the build-time specialisation
    (codegen_qp) both use"""
    st = p5f_structure(N, grouped=True)
    # elimination order: the two horizon chains cut in the middle (qpstruct.bisect_ordering: an elimination tree of two halves
    # under a two-vertex separator instead of one spine; +14 entries of L at N = 10) -- asmqp.LoopSplit gives the halves to two
    # wavefronts. UMPC_QP_ORDERING=minfill: the plain min-fill order of rounds 2-4.
    if os.environ.get("UMPC_QP_ORDERING") == "minfill":
        from . import symbolic
        perm = symbolic.min_fill_ordering(qpstruct.kkt_adjacency(st["n"], st["m"], st["A_p"], st["A_i"]), hold=st["hold"])
        return st, qpstruct.analyse_qp(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"], perm=perm)
    perm = qpstruct.bisect_ordering(st["n"], st["m"], st["A_p"], st["A_i"], parts=st["parts"], hold=st["hold"])
    return st, qpstruct.analyse_qp(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"], perm=perm, parts=st["parts"])


def p5f_hold(st):
    """symbolic.min_fill_ordering's tie-break for the p5f QP: a variable WITHOUT a cost term waits (among candidates of equal
    fill) for one of the dynamics rows that carry it with a constant unit coefficient. Eliminated before all of them its pivot
    is sigma + 1/rho_min^-1 = 2e-6 and its column of L 5e5: still an exact factorisation, but in fp32 the solves lose five
    digits (measured on the reference's linearisations: 7.6e-2 of the iterates against 7.5e-7 with the tie-break; plain
    min-fill on the script's labelling happens to walk the horizon forwards and never does it -- any ordering that walks a
    stretch of the horizon backwards does)."""
    n, m = st["n"], st["m"]
    neq = m - n
    costless = set(range(n)) - set(int(j) for j in st["P_cols"])
    hold = {}
    for j in sorted(costless):
        rows = {n + st["A_i"][q] for q in range(st["A_p"][j], st["A_p"][j + 1])
                if st["A_i"][q] < neq and st["src"][q] < 0 and abs(abs(float(st["cst"][q])) - 1.0) < 1e-12}
        if rows:
            hold[j] = rows
    return hold


def p5f_structure(N=10, grouped=False):
    """The QP of planar/mpc_osqp_p5f.py:87-147: x = (y(0..N) [7 each], u(0..N-1)); rows = (N+1)*7 dynamics
    equalities + identity box rows. Returns dict(n, m, A_p, A_i, P_cols, cst, src, Pv, q, l, u) where
    A[k] = cst[k] (src < 0) or cst[k] * lin[src[k]] with lin = (Ad43, Ad53, Bd4, Bd5, Bd6) (getLin :45-85).
    Structural pattern of (Ad, Bd): the entries the symbolic expressions of getLin can make non-zero.
    grouped: the SAME problem with variables and rows relabelled so that each connected component (qp_components: the
    script's Ad has no diagonal, so the horizon falls apart into two interleaved chains and five one- or two-variable
    pieces) is a contiguous stretch -- variables by component, the dynamics rows by component, the box row of variable j
    still row neq + j. var_order / row_order give the script's index of every relabelled variable / row. This is the order
    PlanarP5fMPC hands to the solver (whose build-time specialisation gives each wavefront of a workgroup whole components)."""
    if grouped:
        st = p5f_structure(N)
        n, m = st["n"], st["m"]
        neq = m - n
        vc, rc = qp_components(n, m, st["A_p"], st["A_i"])
        # ... and inside a large component, the two halves that qpstruct.bisect_ordering's separator cuts it into (the separator
        # itself with the first): a wavefront that owns a half owns a contiguous stretch of variables and rows (aligned pairs
        # of neighbours stay whole)
        hold0 = p5f_hold(st)
        parts = qpstruct.bisect_parts(n, m, st["A_p"], st["A_i"], hold=hold0)
        half = [0 if h == 2 else h for h in parts]
        var_order = sorted(range(n), key=lambda j: (vc[j], half[j], j))
        row_order = sorted(range(neq), key=lambda i: (rc[i], half[n + i], i)) + [neq + j for j in var_order]
        vnew = {j: t for t, j in enumerate(var_order)}
        rnew = {i: t for t, i in enumerate(row_order)}
        A_p, A_i, cst, src = [0], [], [], []
        for j in var_order:
            col = sorted((rnew[st["A_i"][q]], q) for q in range(st["A_p"][j], st["A_p"][j + 1]))
            for i, q in col:
                A_i.append(i)
                cst.append(st["cst"][q])
                src.append(st["src"][q])
            A_p.append(len(A_i))
        Pfull = np.zeros(n)
        Pfull[st["P_cols"]] = st["Pv"]
        Pfull = Pfull[var_order]
        P_cols = [j for j in range(n) if Pfull[j] != 0.0]
        return dict(N=N, n=n, m=m, A_p=A_p, A_i=A_i, P_cols=P_cols, cst=np.array(cst), src=np.array(src, np.int32),
                    Pv=Pfull[P_cols], q=st["q"][var_order], l=st["l"][row_order], u=st["u"][row_order],
                    var_order=np.array(var_order), row_order=np.array(row_order),
                    parts=[parts[j] for j in var_order] + [parts[n + i] for i in row_order],
                    hold={vnew[j]: {n + rnew[r - n] for r in rows} for j, rows in hold0.items()})
    nx, nu = 7, 1
    n = (N + 1) * nx + N * nu
    neq = (N + 1) * nx
    m = neq + n
    ent = {}
    for k in range(N + 1):
        for i in range(nx):
            ent[(k * nx + i, k * nx + i)] = (-1.0, -1)                 # kron(eye(N+1), -eye(nx))
        if k > 0:
            c0 = (k - 1) * nx
            for (r, c, v, sidx) in ((1, 4, 1.0, -1), (2, 5, 1.0, -1), (3, 6, 1.0, -1), (4, 3, 1.0, 0), (5, 3, 1.0, 1)):
                ent[(k * nx + r, c0 + c)] = (v, sidx)                  # kron(eye(N+1, k=-1), Ad)
            cu = (N + 1) * nx + (k - 1)
            for (r, v, sidx) in ((0, 1.0, -1), (4, 1.0, 2), (5, 1.0, 3), (6, 1.0, 4)):
                ent[(k * nx + r, cu)] = (v, sidx)                      # kron(vstack(0, eye(N)), Bd)
    for j in range(n):
        ent[(neq + j, j)] = (1.0, -1)                                  # Aineq = eye
    A_p, A_i, cst, src = [0], [], [], []
    for j in range(n):
        for i in sorted(r for (r, c) in ent if c == j):
            A_i.append(i)
            cst.append(ent[(i, j)][0])
            src.append(ent[(i, j)][1])
        A_p.append(len(A_i))
    Qd = np.array([0., 10., 10., 10., 0., 0., 0.])
    Pfull = np.hstack([np.tile(Qd, N + 1), np.full(N * nu, 0.1)])
    P_cols = [j for j in range(n) if Pfull[j] != 0.0]                  # scipy/osqp keep the non-zeros of P
    yr = np.array([0., 0., 1., 0., 0., 0., 0.])
    q = np.hstack([np.tile(-Qd * yr, N + 1), np.zeros(N * nu)])
    l = np.hstack([np.zeros(neq), np.full(n, -OSQP_INFTY)])
    u = np.hstack([np.zeros(neq), np.full(n, OSQP_INFTY)])
    return dict(N=N, n=n, m=m, A_p=A_p, A_i=A_i, P_cols=P_cols, cst=np.array(cst), src=np.array(src, np.int32),
                Pv=Pfull[P_cols], q=q, l=l, u=u)


class PlanarP5fMPC:
    """B copies of the planar stroke-plane MPC loop of planar/mpc_osqp_p5f.py:151-176. Per tick and robot:
    getLin about (unom(t), sigma, phi) of the previous state, rebuild A, one fixed-iteration QP step (the
    reference calls prob.update(Ax=A) and never solve(); the step is the build's), then the
    reference's plant tick y += (Ad y + Bd unom) dt."""

    def __init__(self, B, dtype=torch.float32, device="cuda", N=10, dt=0.002, **settings):
        st, s = p5f_analysis(N)           # (the labelling AND the elimination order the build-time specialisation was made for)
        self.st, self.B, self.dt, self.dtype = st, int(B), float(dt), dtype
        self.qp = BatchQP(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"], B, dtype, device, perm=s.perm, **settings)
        if not os.environ.get("UMPC_QP_KERNEL") and dtype != torch.float32:
            self.qp.set_kernel("wave")     # fp64: 10.8 ms per tick (fp32 figure) against 13.0 ms for the C++ lane specialisation;
            # fp32 keeps the default "lane": the specialisation with its middle iterations in assembly (asmqp.py)
        dev = self.qp.device
        col = lambda v: torch.as_tensor(np.repeat(np.asarray(v, np.float64)[:, None], B, 1)).to(dev, dtype).contiguous()
        self.Pv, self.q, self.l, self.u = col(st["Pv"]), col(st["q"]), col(st["l"]), col(st["u"])
        self.cst = torch.as_tensor(st["cst"]).to(dev, dtype)
        self.src = torch.as_tensor(st["src"]).to(dev)
        self.Av = torch.zeros((len(st["A_i"]), B), dtype=dtype, device=dev)
        self.lin = torch.zeros((5, B), dtype=dtype, device=dev)
        self.y = torch.zeros((7, B), dtype=dtype, device=dev)
        self.u_nom = torch.zeros(B, dtype=dtype, device=dev)
        self._av_constants_written = False
        self.fused = dtype == torch.float32 and not os.environ.get("UMPC_P5F_UNFUSED")   # tick(): try umpcP5fTick first
        self.L = self.qp.L

    def linearise(self, u):
        """getLin at (u, sigma = y[0], phi = y[3]) -> lin [5, B] and the assembled A values."""
        stream = C.c_void_p(torch.cuda.current_stream(self.qp.device).cuda_stream)
        if torch.is_tensor(u):
            self.u_nom.copy_(u)
        # one launch: getLin -> lin and the entries of A (after the first tick only the state-dependent ones)
        rc = self.L.umpcP5fLinearise(self.B, _DT[self.dtype], _ptr(self.u_nom) if torch.is_tensor(u) else None,
                                     0.0 if torch.is_tensor(u) else float(u), _ptr(self.y), _ptr(self.lin), int(self.cst.numel()),
                                     _ptr(self.cst), _ptr(self.src), _ptr(self.Av), 1 if self._av_constants_written else 0, stream)
        if rc != 0:
            raise RuntimeError(self.L.umpcLastError().decode())
        self._av_constants_written = True      # (Av is this object's: nobody else writes its constant entries)
        return self.lin

    def tick(self, t, solve=True):
        """One loop body of mpc_osqp_p5f.py:157-176 at time t (unom = 15 sin(2 pi 170 t), :157)."""
        unom = 15.0 * np.sin(2 * np.pi * 170 * t)
        if solve and self.fused and self._av_constants_written:
            # round 5: getLin, the A update and the plant tick are the prologue of the QP kernel (umpcP5fTick): one launch
            # instead of three, the same numbers (tests/test_bqp.py::test_gpu_p5f_fused_tick_equals_the_three_launches)
            qp = self.qp
            stream = C.c_void_p(torch.cuda.current_stream(qp.device).cuda_stream)
            rc = self.L.umpcP5fTick(qp.h, _ptr(self.Pv), _ptr(self.Av), _ptr(self.q), _ptr(self.l), _ptr(self.u), _ptr(qp.x),
                                    _ptr(qp.y), _ptr(qp.z), _ptr(qp.Eprev), _ptr(qp.sol_x), _ptr(qp.sol_y), _ptr(qp.status),
                                    _ptr(qp.info), float(unom), self.dt, _ptr(self.y), _ptr(self.lin), int(self.cst.numel()),
                                    _ptr(self.cst), _ptr(self.src), stream)
            if rc == 0:
                self.u_nom_scalar = float(unom)
                return self.y
            if rc != -2:
                raise RuntimeError(self.L.umpcLastError().decode())
            self.fused = False          # this handle does not dispatch the assembly kernel: the three calls from now on
        self.linearise(unom)
        if solve:
            self.qp.solve(self.Pv, self.Av, self.q, self.l, self.u)
        self._p5f_step(1, unom, None)
        return self.y

    def solution(self):
        """sol_x [n, B] of the last solve in the SCRIPT's variable order (the solver works on the relabelled problem of
        p5f_structure(grouped=True))"""
        out = torch.empty_like(self.qp.sol_x)
        out[torch.as_tensor(self.st["var_order"]).to(out.device)] = self.qp.sol_x
        return out

    def _p5f_step(self, mode, u, lin):
        """umpcP5fStep with a per-robot input tensor, umpcP5fStepU with the reference's scalar (no [B] array filled per tick)"""
        stream = C.c_void_p(torch.cuda.current_stream(self.qp.device).cuda_stream)
        if torch.is_tensor(u):
            self.u_nom.copy_(u)
            rc = self.L.umpcP5fStep(self.B, _DT[self.dtype], mode, self.dt, _ptr(self.u_nom), _ptr(self.y),
                                    _ptr(lin) if lin is not None else None, stream)
        else:
            self.u_nom_scalar = float(u)
            rc = self.L.umpcP5fStepU(self.B, _DT[self.dtype], mode, self.dt, float(u), _ptr(self.y),
                                     _ptr(lin) if lin is not None else None, stream)
        if rc != 0:
            raise RuntimeError(self.L.umpcLastError().decode())


# ---------------------------------------------------------------------------------------------------------
# planar/mpc_osqp_p5f_stroke.py: the stroke model (two inputs (u, tf) per stroke, LTV blocks), SURVEY 8(f-4)
# ---------------------------------------------------------------------------------------------------------
STROKE_NPAR = 14        # per stage: tf0 | Ad[4,3] Ad[5,3] Ad[6,0] | u0 dx dz dphi | Bd[4,0] Bd[4,1] Bd[5,0] Bd[5,1] Bd[6,0] Bd[6,1]


def stroke_getlin(u0, tf0, y):
    """getLin of planar/mpc_osqp_p5f_stroke.py:38-86 for arrays u0, tf0 [B] and y [7, B] (y = (sigma, x, z, phi, dx, dz,
    dphi)): the STROKE_NPAR entries of (Ad, Bd) that depend on the linearisation point, [14, B] float64 (host side:
    the script evaluates it once per stage and tick; the structure's other entries are the constants 1 and -1)."""
    u0, tf0, y = np.asarray(u0, np.float64), np.asarray(tf0, np.float64), np.asarray(y, np.float64)
    sigma0, phi, dx, dz, dphi = y[0], y[3], y[4], y[5], y[6]
    d, kaero2 = 2, 0.091875
    u02, u03 = u0 ** 2, u0 ** 3
    cphi, sphi, su0 = np.cos(phi), np.sin(phi), np.sign(u0)
    one = np.ones_like(u0 * tf0 * phi)
    return np.stack([
        tf0 * one,
        (tf0 * (-(cphi * kaero2 * u02) + kaero2 * sphi * su0 * u02)) / 100.,
        (tf0 * (-(kaero2 * sphi * u02) - cphi * kaero2 * su0 * u02)) / 100.,
        (kaero2 * tf0 * u02) / 7200. * one,
        u0 * one, dx * one, dz * one, dphi * one,
        (tf0 * (-2 * kaero2 * sphi * u0 - 2 * cphi * kaero2 * su0 * u0)) / 100.,
        (-(kaero2 * sphi * u02) - cphi * kaero2 * su0 * u02) / 100.,
        (tf0 * (2 * cphi * kaero2 * u0 - 2 * kaero2 * sphi * su0 * u0)) / 100.,
        (-0.9800000000000001 + cphi * kaero2 * u02 - kaero2 * sphi * su0 * u02) / 100.,
        (tf0 * (2 * d * kaero2 * su0 * u0 + 2 * kaero2 * u0 * (sigma0 + (tf0 * u0) / 2.) + (kaero2 * tf0 * u02) / 2.)) / 7200.,
        (d * kaero2 * su0 * u02 + kaero2 * (sigma0 + (tf0 * u0) / 2.) * u02) / 7200. + (kaero2 * tf0 * u03) / 14400.])


def p5f_stroke_structure(N=1):
    """The QP of planar/mpc_osqp_p5f_stroke.py:139-193 with the LTV blocks of getCondensed :88-124: x = (y(0..N) [7 each],
    (u, tf)(0..N-1)); rows = (N+1)*7 dynamics equalities (-I on the diagonal, stage k's Ad below it, its Bd in the input
    columns) + identity box rows (states free, umin = (-10, 0.1), umax = (10, 100) :130-133). Returns dict(n, m, A_p, A_i,
    P_cols, cst, src, Pv, q, l, u): A[k] = cst[k] (src < 0) or cst[k] * par[src[k]] with par [STROKE_NPAR * N, B] = the stages'
    stroke_getlin rows. The pattern holds every entry getLin's expressions can make non-zero (13 of Ad, 11 of Bd); P is what
    scipy's block_diag keeps of diag(Q .. QN, R) as non-zeros (Q = diag(0, 10, 10, 10, 0, 0, 0), R = 10 I :136-138)."""
    nx, nu = 7, 2
    n = (N + 1) * nx + N * nu
    neq = (N + 1) * nx
    m = neq + n
    ent = {}
    AD = ((0, 0, -1), (1, 1, -1), (1, 4, 0), (2, 2, -1), (2, 5, 0), (3, 3, -1), (3, 6, 0), (4, 3, 1), (4, 4, -1), (5, 3, 2),
          (5, 5, -1), (6, 0, 3), (6, 6, -1))
    BD = ((0, 0, 0), (0, 1, 4), (1, 1, 5), (2, 1, 6), (3, 1, 7), (4, 0, 8), (4, 1, 9), (5, 0, 10), (5, 1, 11), (6, 0, 12), (6, 1, 13))
    for k in range(N + 1):
        for i in range(nx):
            ent[(k * nx + i, k * nx + i)] = (-1.0, -1)
        if k > 0:
            for (r, c, sidx) in AD:
                ent[(k * nx + r, (k - 1) * nx + c)] = (1.0, -1 if sidx < 0 else STROKE_NPAR * (k - 1) + sidx)
            for (r, c, sidx) in BD:
                ent[(k * nx + r, neq + nu * (k - 1) + c)] = (1.0, STROKE_NPAR * (k - 1) + sidx)
    for j in range(n):
        ent[(neq + j, j)] = (1.0, -1)
    A_p, A_i, cst, src = [0], [], [], []
    for j in range(n):
        for i in sorted(r for (r, c) in ent if c == j):
            A_i.append(i)
            cst.append(ent[(i, j)][0])
            src.append(ent[(i, j)][1])
        A_p.append(len(A_i))
    Qd = np.array([0., 10., 10., 10., 0., 0., 0.])
    Pfull = np.hstack([np.tile(Qd, N + 1), np.full(N * nu, 10.0)])
    P_cols = [j for j in range(n) if Pfull[j] != 0.0]
    yr = np.zeros(nx)                                                   # :143
    q = np.hstack([np.tile(-Qd * yr, N + 1), np.zeros(N * nu)])
    l = np.hstack([np.zeros(neq), np.full(neq, -OSQP_INFTY), np.tile([-10.0, 0.1], N)])
    u = np.hstack([np.zeros(neq), np.full(neq, OSQP_INFTY), np.tile([10.0, 100.0], N)])
    return dict(N=N, n=n, m=m, A_p=A_p, A_i=A_i, P_cols=P_cols, cst=np.array(cst), src=np.array(src, np.int32),
                Pv=Pfull[P_cols], q=q, l=l, u=u, npar=STROKE_NPAR * N)


def stroke_dense_A(st, par):
    """dense A [m, n] of ONE robot from the structure and its parameter column (host side; the tests compare it with the
    script's own getCondensed output)"""
    A = np.zeros((st["m"], st["n"]))
    for j in range(st["n"]):
        for k in range(st["A_p"][j], st["A_p"][j + 1]):
            A[st["A_i"][k], j] = st["cst"][k] * (1.0 if st["src"][k] < 0 else par[st["src"][k]])
    return A


class PlanarP5fStrokeMPC:
    """B copies of the QP step of planar/mpc_osqp_p5f_stroke.py:204-225 on the general-structure solver: per tick the N stage
    linearisations (stroke_getlin along the nominal alternating-stroke rollout, getCondensed :88-124) fill A through
    umpcQPGather, the first seven bounds pin the initial state (l[:nx] = u[:nx] = -y0, :120-122) and the QP is solved with
    the script's tolerances (eps 1e-2, :196) in pip-osqp termination semantics, or for a fixed iteration count."""

    def __init__(self, B, dtype=torch.float64, device="cuda", N=1, **settings):
        st = p5f_stroke_structure(N)
        self.st, self.B, self.N, self.dtype = st, int(B), N, dtype
        settings.setdefault("eps_abs", 1e-2)
        settings.setdefault("eps_rel", 1e-2)
        self.qp = BatchQP(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"], B, dtype, device, **settings)
        dev = self.qp.device
        col = lambda v: torch.as_tensor(np.repeat(np.asarray(v, np.float64)[:, None], B, 1)).to(dev, dtype).contiguous()
        self.Pv, self.q, self.l, self.u = col(st["Pv"]), col(st["q"]), col(st["l"]), col(st["u"])
        self.cst = torch.as_tensor(st["cst"]).to(dev, dtype)
        self.src = torch.as_tensor(st["src"]).to(dev)
        self.par = torch.zeros((st["npar"], B), dtype=dtype, device=dev)
        self.Av = torch.zeros((len(st["A_i"]), B), dtype=dtype, device=dev)
        self.L = self.qp.L

    def rollout_parameters(self, u0, tf0, y0):
        """getCondensed's forward simulation (:97-112): stage j linearises about (u_j, tf0, y_j) with u_j alternating in sign
        and y_{j+1} = Ad y_j + Bd (u_j, tf0). u0, tf0 [B], y0 [7, B] (host arrays) -> par [14 N, B]."""
        u0, tf0 = np.asarray(u0, np.float64) * np.ones(self.B), np.asarray(tf0, np.float64) * np.ones(self.B)
        y = np.array(y0, np.float64).reshape(7, -1) * np.ones((7, self.B))
        rows, up = [], u0
        for _ in range(self.N):
            pr = stroke_getlin(up, tf0, y)
            rows.append(pr)
            tf, a43, a53, a60, bu, bdx, bdz, bdphi, b40, b41, b50, b51, b60, b61 = pr
            y = np.stack([y[0] + tf * up + bu * tf0,                       # row 0: Ad = 1; Bd = (tf0, u0)
                          y[1] + tf * y[4] + bdx * tf0, y[2] + tf * y[5] + bdz * tf0, y[3] + tf * y[6] + bdphi * tf0,
                          a43 * y[3] + y[4] + b40 * up + b41 * tf0, a53 * y[3] + y[5] + b50 * up + b51 * tf0,
                          a60 * y[0] + y[6] + b60 * up + b61 * tf0])
            up = -up
        return np.vstack(rows)

    def update(self, par, y0):
        """A <- gather(par), l[:7] = u[:7] = -y0; par [14 N, B], y0 [7, B] (host arrays or device tensors)"""
        dev = self.qp.device
        self.par.copy_(torch.as_tensor(np.asarray(par, np.float64) if not torch.is_tensor(par) else par).to(dev, self.dtype))
        y0 = torch.as_tensor(np.asarray(y0, np.float64) if not torch.is_tensor(y0) else y0).to(dev, self.dtype)
        self.l[:7] = -y0
        self.u[:7] = -y0
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = self.L.umpcQPGather(self.B, _DT[self.dtype], int(self.cst.numel()), _ptr(self.cst), _ptr(self.src), _ptr(self.par),
                                 _ptr(self.Av), stream)
        if rc != 0:
            raise RuntimeError(self.L.umpcLastError().decode())

    def solve(self, max_iter=None):
        return self.qp.solve(self.Pv, self.Av, self.q, self.l, self.u, max_iter=max_iter)


# ---------------------------------------------------------------------------------------------------------
# template/genqp.py UprightMPC (v1)
# ---------------------------------------------------------------------------------------------------------
def v1_structure(N=3):
    """Pattern of genqp.UprightMPC.__init__ (template/genqp.py:49-112) and the positions `Axidx` of its
    state-dependent A entries (saveAxidx :120-135). x = (q_1..q_N [6 each], u_0..u_{N-1} [3 each]); rows =
    6N dynamics + 3N s-limits. Returns dict(n, m, A_p, A_i, P_cols, A0, Axidx)."""
    nq, nu = 6, 3
    nx = N * (nq + nu)
    ncon = N * (nq + 3)
    A = np.zeros((ncon, nx))
    Bs = lambda s: np.block([[np.reshape(s, (3, 1)), np.zeros((3, 2))], [np.zeros((2, 1)), np.eye(2)],
                             [np.zeros((1, 1)), -s[None, :2] / s[2]]]) * 2.0
    A0 = np.eye(nq) + np.diag(np.ones(3), k=3)
    for k in range(N):
        A[k * nq:(k + 1) * nq, k * nq:(k + 1) * nq] = -np.eye(nq)
        if k > 0:
            A[k * nq:(k + 1) * nq, (k - 1) * nq:k * nq] = A0
        A[k * nq:(k + 1) * nq, N * nq + k * nu:N * nq + (k + 1) * nu] = Bs(np.ones(3))
        A[N * nq + 3 * k:N * nq + 3 * (k + 1), k * nq + 3:k * nq + 6] = np.eye(3)
    A_p, A_i = qpstruct.csc_pattern(A != 0)
    Adata = np.array([A[i, j] for j in range(nx) for i in A_i[A_p[j]:A_p[j + 1]]])
    A1nnz = 18 * N - 9
    Axidx = [18 * k + i for k in range(N - 1) for i in (7, 11, 15)] + list(range(A1nnz, A1nnz + 7 * N))
    P_cols = list(range((N - 1) * nq, N * nq)) + list(range(N * nq, nx))
    return dict(N=N, n=nx, m=ncon, A_p=A_p, A_i=A_i, P_cols=P_cols, Adata=Adata, Axidx=Axidx)


class UprightMPCv1:
    """B copies of genqp.UprightMPC (template/genqp.py:43-168). update() takes per-robot [rows, B] tensors (or
    broadcastable numpy) of the reference's arguments and returns (x [n, B], uu [3, B]); the nominal-trajectory
    bookkeeping (snom, vT0; :163-166) is kept per robot on the device."""

    def __init__(self, B, N=3, dtype=torch.float64, device="cuda", **settings):
        st = v1_structure(N)
        self.st, self.N, self.B, self.dtype = st, N, int(B), dtype
        self.qp = BatchQP(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"], B, dtype, device, **settings)
        dev = self.qp.device
        self.dev = dev
        nnzA = len(st["A_i"])
        # A = constants except at Axidx: parameter rows par = [dt*vT0 | dt*(sx,sy,sz,1,-sx/sz,1,-sy/sz) per stage]
        cst = np.array(st["Adata"], np.float64)
        src = np.full(nnzA, -1, np.int32)
        for r, k in enumerate(st["Axidx"]):
            cst[k] = 1.0
            src[k] = 0 if r < 3 * (N - 1) else 1 + (r - 3 * (N - 1))
        self.cst = torch.as_tensor(cst).to(dev, dtype)
        self.src = torch.as_tensor(src).to(dev)
        self.par = torch.zeros((1 + 7 * N, B), dtype=dtype, device=dev)
        self.Av = torch.zeros((nnzA, B), dtype=dtype, device=dev)
        self.Pv = torch.zeros((len(st["P_cols"]), B), dtype=dtype, device=dev)
        self.q = torch.zeros((st["n"], B), dtype=dtype, device=dev)
        self.l = torch.zeros((st["m"], B), dtype=dtype, device=dev)
        self.u = torch.zeros((st["m"], B), dtype=dtype, device=dev)
        self.resetNominal()

    def resetNominal(self):
        self.snom = torch.zeros((3 * self.N, self.B), dtype=self.dtype, device=self.dev)
        self.snom[2::3] = 1.0
        self.vT0 = torch.zeros(self.B, dtype=self.dtype, device=self.dev)

    def _t(self, v, rows):
        t = torch.as_tensor(np.asarray(v, np.float64)) if not torch.is_tensor(v) else v
        t = t.to(self.dev, self.dtype)
        if t.dim() == 1:
            t = t[:, None].expand(rows, self.B)
        return t

    def assemble(self, q0, qdes, Qfdiag, Rdiag, smin, smax, dt, snom=None, vT0=None):
        """The data updates of genqp.py:132-155 (device tensor arithmetic is row copies / scalings only; the
        A values go through umpcQPGather)."""
        N, nq = self.N, 6
        q0, qdes, Qf, Rd = self._t(q0, 6), self._t(qdes, 6), self._t(Qfdiag, 6), self._t(Rdiag, 3)
        smin, smax = self._t(smin, 3), self._t(smax, 3)
        snom = self.snom if snom is None else self._t(snom, 3 * N)
        if vT0 is None:
            vT0 = self.vT0
        elif torch.is_tensor(vT0):
            vT0 = vT0.to(self.dev, self.dtype).expand(self.B)
        else:
            vT0 = torch.as_tensor(np.broadcast_to(np.asarray(vT0, np.float64), (self.B,)).copy()).to(self.dev, self.dtype)
        self.vT0 = vT0.clone()
        dtv = dt * vT0
        a0q0 = q0.clone()
        a0q0[:3] += dtv * q0[3:]                                      # A0 @ q0, A0 = I + dt vT0 diag(1,1,1; k=3)
        self.l.zero_(); self.u.zero_()
        self.l[:nq] = -a0q0; self.u[:nq] = -a0q0
        self.l[N * nq:] = smin.repeat(N, 1); self.u[N * nq:] = smax.repeat(N, 1)
        self.q.zero_()
        self.q[(N - 1) * nq:N * nq] = -(Qf * qdes)
        self.Pv[:nq] = Qf
        self.Pv[nq:] = Rd.repeat(N, 1)
        self.par[0] = dtv
        for k in range(N):
            s = snom[3 * k:3 * k + 3]
            blk = self.par[1 + 7 * k:8 + 7 * k]
            blk[0], blk[1], blk[2] = dt * s[0], dt * s[1], dt * s[2]
            blk[3] = dt
            blk[4] = dt * (-s[0] / s[2])
            blk[5] = dt
            blk[6] = dt * (-s[1] / s[2])
        self.qp.gather(self.cst, self.src, self.par, self.Av)

    def update(self, q0, qdes, Qfdiag, Rdiag, smin, smax, dt, snom=None, vT0=None, max_iter=None):
        self.assemble(q0, qdes, Qfdiag, Rdiag, smin, smax, dt, snom, vT0)
        x, _, _ = self.qp.solve(self.Pv, self.Av, self.q, self.l, self.u, max_iter=max_iter)
        N, nq = self.N, 6
        uu = x[N * nq:N * nq + 3].clone()
        self.snom = torch.cat([x[i * nq + 3:i * nq + 6] for i in range(N)], 0).clone()   # genqp.py:164
        self.vT0 = self.vT0 + uu[0]                                                       # genqp.py:165
        return x, uu


# ---------------------------------------------------------------------------------------------------------
# template/template_controllers.py UprightMPC2 at any horizon N
# ---------------------------------------------------------------------------------------------------------
def uprightmpc2_structure(N):
    """initConstraint(N, nx, nc) (template/template_controllers.py:28-63): CSC pattern of A and, per entry, its
    constant or the umpcNAssemble parameter row that scales it."""
    from . import symbolic
    A_p, A_i, A_tag = symbolic.build_A(N)
    nx, nc = symbolic.dims(N)
    cst, src = [], []
    for tag in A_tag:
        if tag[0] == 'c':
            cst.append(tag[1]); src.append(-1)
        elif tag[0] == 'dt':
            cst.append(None); src.append(-1)          # filled with dt by the caller
        elif tag[0] == 'T0dt':
            cst.append(1.0); src.append(0)
        elif tag[0] == 's0':
            cst.append(1.0); src.append(1 + tag[1])
        else:
            cst.append(1.0); src.append(4 + tag[1])
    return dict(N=N, n=nx, m=nc, A_p=A_p, A_i=A_i, P_cols=list(range(nx)), cst=cst, src=np.array(src, np.int32))


class UprightMPC2N:
    """B copies of template_controllers.UprightMPC2(N, dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, Ib)
    (:170-258) for any horizon N >= 2, with the embedded-C step (fixed max_iter ADMM iterations) as the solve.
    update(state, ref, actualT0=None) -> out [9, B] = (uquad, accdes); T0 is carried per robot."""

    def __init__(self, B, N, dt=5.0, g=9.81e-3, TtoWmax=2.0, ws=1e1, wds=1e3, wpr=1.0, wpf=5.0, wvr=1e3, wvf=2e3,
                 wthrust=1e-1, wmom=1e-2, Ib=(3333.0, 3333.0, 1000.0), dtype=torch.float32, device="cuda", **settings):
        st = uprightmpc2_structure(N)
        self.st, self.N, self.B, self.dtype = st, N, int(B), dtype
        self.qp = BatchQP(st["n"], st["m"], st["A_p"], st["A_i"], st["P_cols"], B, dtype, device, **settings)
        dev = self.qp.device
        self.dev, self.L = dev, self.qp.L
        self.prm = _lib.NParams(dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, (C.c_double * 3)(*Ib))
        self.cst = torch.as_tensor(np.array([dt if c is None else c for c in st["cst"]], np.float64)).to(dev, dtype)
        self.src = torch.as_tensor(st["src"]).to(dev)
        z = lambda r: torch.zeros((r, self.B), dtype=dtype, device=dev)
        self.Pv, self.q, self.l, self.u = z(st["n"]), z(st["n"]), z(st["m"]), z(st["m"])
        self.Av, self.par, self.out = z(len(st["A_i"])), z(10), z(9)
        self.T0 = torch.zeros(self.B, dtype=dtype, device=dev)

    def assemble(self, state, ref, actualT0=None):
        self.qp._chk(state, 18, "state")
        self.qp._chk(ref, 9, "ref")
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        rc = self.L.umpcNAssemble(self.B, _DT[self.dtype], self.N, C.byref(self.prm), _ptr(state), _ptr(ref), _ptr(self.T0),
                                  _ptr(actualT0), _ptr(self.Pv), _ptr(self.q), _ptr(self.l), _ptr(self.u), _ptr(self.par),
                                  stream)
        if rc != 0:
            raise RuntimeError(self.L.umpcLastError().decode())
        self.qp.gather(self.cst, self.src, self.par, self.Av)

    def update(self, state, ref, actualT0=None):
        self.assemble(state, ref, actualT0)
        sol_x, _, _ = self.qp.solve(self.Pv, self.Av, self.q, self.l, self.u)
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        rc = self.L.umpcNExtract(self.B, _DT[self.dtype], self.N, float(self.prm.dt), _ptr(state), _ptr(sol_x),
                                 _ptr(self.T0), _ptr(self.out), stream)
        if rc != 0:
            raise RuntimeError(self.L.umpcLastError().decode())
        return self.out

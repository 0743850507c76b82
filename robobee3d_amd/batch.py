"""Batched uprightmpc2 controller + plant on one MI355X (host side, Python).

torch is used only for device memory, streams and (elsewhere) torch.distributed;
all compute is in libumpc_mi355x.so, reached through the C ABI with raw device
pointers. Array convention: SoA [rows, B] contiguous, robot index fastest
(include/umpc_mi355x.h).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_DT = {torch.float32: _lib.UMPC_F32, torch.float64: _lib.UMPC_F64}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def hover_initial_conditions(B, seed, dtype=np.float32, tilt=0.5, index_offset=0):
    """Random-tilt hover start (reference single IC at
    template/uprightmpc2.py:101-103 is a=0.5, b=-0.5): p=0, Rb = Rx(a) Ry(b),
    a,b ~ U(-tilt, tilt), dq = (0.1,0,0,0,0,0). Streams are keyed by the GLOBAL
    robot index so a sharded run draws the same numbers as a single-GPU run.
    Returns state[18,B] (R column-major) and ref[9,B] (pdes=0, dpdes=0, sdes=e3)."""
    idx = np.arange(index_offset, index_offset + B, dtype=np.uint64)
    # counter-based: two uniforms per robot from a hash of (seed, index)
    def u01(salt):
        v = (idx + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(salt)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        v ^= v >> np.uint64(30); v *= np.uint64(0xBF58476D1CE4E5B9)
        v ^= v >> np.uint64(27); v *= np.uint64(0x94D049BB133111EB)
        v ^= v >> np.uint64(31)
        return (v >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    with np.errstate(over="ignore"):
        a = (2 * u01(1) - 1) * tilt
        b = (2 * u01(2) - 1) * tilt
    ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    # R = Rx(a) @ Ry(b)
    R = np.empty((B, 3, 3))
    R[:, 0, 0] = cb; R[:, 0, 1] = 0; R[:, 0, 2] = sb
    R[:, 1, 0] = sa * sb; R[:, 1, 1] = ca; R[:, 1, 2] = -sa * cb
    R[:, 2, 0] = -ca * sb; R[:, 2, 1] = sa; R[:, 2, 2] = ca * cb
    state = np.zeros((18, B), dtype)
    state[3:12] = R.transpose(2, 1, 0).reshape(9, B)  # column-major: row r + 3*col c
    state[12] = 0.1
    ref = np.zeros((9, B), dtype)
    ref[8] = 1.0
    return state, ref


def _u01(seed, idx, salt):
    """Counter-based uniform in [0, 1): a hash of (seed, global robot index, salt)."""
    with np.errstate(over="ignore"):
        v = (idx + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(salt)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        v ^= v >> np.uint64(30); v *= np.uint64(0xBF58476D1CE4E5B9)
        v ^= v >> np.uint64(27); v *= np.uint64(0x94D049BB133111EB)
        v ^= v >> np.uint64(31)
    return (v >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def monte_carlo_draws(B, seed, dtype=np.float32, spread=0.2, index_offset=0):
    """BASELINE configs[4] inputs: Ib = Ib0 (1 + d), d ~ U(-spread, spread)^3 for controller and
    plant, plant thrust gain 1 + U(-spread, spread); keyed by the GLOBAL robot index like hover_initial_conditions, so
    a sharded run draws exactly what the single-GPU run draws. Returns Ib [3, B], gain [B]."""
    idx = np.arange(index_offset, index_offset + B, dtype=np.uint64)
    Ib0 = np.array([3333.0, 3333.0, 1000.0])   # template/genqp.py:22
    Ib = np.stack([Ib0[i] * (1 + (2 * _u01(seed, idx, 11 + i) - 1) * spread) for i in range(3)])
    gain = 1 + (2 * _u01(seed, idx, 14) - 1) * spread
    return Ib.astype(dtype), gain.astype(dtype)


def _u01_device(seed, idx, salt):
    """_u01 on the device: the same counter hash in int64 two's-complement arithmetic (wrap-around multiplies, logical
    right shifts spelled as arithmetic shift + mask), bit-identical to the numpy uint64 version."""
    def c(v):        # 64-bit constant as a signed int
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >> 63 else v

    def lsr(v, k):
        return (v >> k) & ((1 << (64 - k)) - 1)
    v = idx + c(int(seed) * 0x9E3779B97F4A7C15) + int(salt)
    v = v ^ lsr(v, 30)
    v = v * c(0xBF58476D1CE4E5B9)
    v = v ^ lsr(v, 27)
    v = v * c(0x94D049BB133111EB)
    v = v ^ lsr(v, 31)
    return lsr(v, 11).to(torch.float64) / float(1 << 53)


def monte_carlo_draws_device(B, seed, dtype=torch.float32, spread=0.2, index_offset=0, device="cuda"):
    """monte_carlo_draws generated ON the device (an 8-rank config-5 start is then 8 tiny kernels, not 8
    host-side numpy passes over 2^17 x 4 draws + 8 uploads): same hash, same fp64 arithmetic, bit-identical values.
    Returns Ib [3, B], gain [B] as device tensors."""
    idx = torch.arange(index_offset, index_offset + B, dtype=torch.int64, device=device)
    Ib0 = (3333.0, 3333.0, 1000.0)   # template/genqp.py:22
    Ib = torch.stack([Ib0[i] * (1 + (2 * _u01_device(seed, idx, 11 + i) - 1) * spread) for i in range(3)])
    gain = 1 + (2 * _u01_device(seed, idx, 14) - 1) * spread
    return Ib.to(dtype).contiguous(), gain.to(dtype).contiguous()


def hover_initial_conditions_device(B, seed, dtype=torch.float32, tilt=0.5, index_offset=0, device="cuda"):
    """hover_initial_conditions generated ON the device: identical tilt angles (same hash, bit for bit); the rotation
    entries come from the device's fp64 sin / cos, which may differ from numpy's in the last fp64 bit before the cast."""
    idx = torch.arange(index_offset, index_offset + B, dtype=torch.int64, device=device)
    a = (2 * _u01_device(seed, idx, 1) - 1) * tilt
    b = (2 * _u01_device(seed, idx, 2) - 1) * tilt
    ca, sa, cb, sb = torch.cos(a), torch.sin(a), torch.cos(b), torch.sin(b)
    z = torch.zeros_like(a)
    state = torch.zeros((18, B), dtype=torch.float64, device=device)
    # R = Rx(a) Ry(b), column-major rows 3..11: entry r + 3 c
    cols = ((cb, sa * sb, -ca * sb), (z, ca, sa), (sb, -sa * cb, ca * cb))
    for cidx, col in enumerate(cols):
        for r in range(3):
            state[3 + r + 3 * cidx] = col[r]
    state[12] = 0.1
    ref = torch.zeros((9, B), dtype=torch.float64, device=device)
    ref[8] = 1.0
    return state.to(dtype).contiguous(), ref.to(dtype).contiguous(), (a, b)


class BatchUprightMPC:
    """B independent uprightmpc2 controllers (+ plants), one GPU lane each."""

    def __init__(self, B, dtype=torch.float32, device="cuda", global_batch=None, **params):
        """global_batch: the size of the whole job when this object holds one block of a sharded batch (shard.py,
        SURVEY 8e). The automatic lane / quad choice of the step kernel is made from it, so every block runs the
        instruction stream the undivided batch would and the partition cannot change a bit of the result."""
        if not torch.cuda.is_available():
            raise RuntimeError("BatchUprightMPC needs a HIP device; there is no CPU path")
        self.L = _lib.lib()
        self.B, self.dtype, self.device = int(B), dtype, torch.device(device)
        self.prm = _lib.default_params()
        for k, v in params.items():
            if k == "Ib":
                for i in range(3):
                    self.prm.Ib[i] = float(v[i])
            else:
                if not hasattr(self.prm, k):
                    raise TypeError("unknown parameter %r" % k)
                setattr(self.prm, k, v)
        with torch.cuda.device(self.device):
            self.h = self.L.umpcBatchCreate(C.byref(self.prm), self.B, _DT[dtype])
        if not self.h:
            raise RuntimeError(self.L.umpcLastError().decode())
        if global_batch is not None:
            self._check(self.L.umpcBatchSetGlobalBatch(self.h, int(global_batch)))
        z = lambda r, dt=dtype: torch.zeros((r, self.B), dtype=dt, device=self.device)
        self.state, self.ctrl, self.ref = z(_lib.STATE_ROWS), z(_lib.CTRL_ROWS), z(_lib.REF_ROWS)
        self.out, self.stats, self.info = z(_lib.OUT_ROWS), z(_lib.STAT_ROWS), z(2)
        self.status = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self.Ib = None
        self.gain = None
        self.actualT0 = None
        self.reset_controller()

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.umpcBatchDestroy(self.h)
                self.h = None
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        if rc:
            raise RuntimeError(self.L.umpcLastError().decode())

    def reset_controller(self):
        self._check(self.L.umpcBatchInitCtrl(self.h, _ptr(self.ctrl), self._stream()))
        self.stats.zero_()

    def set_state(self, state, ref=None):
        self.state.copy_(torch.as_tensor(state, dtype=self.dtype))
        if ref is not None:
            self.ref.copy_(torch.as_tensor(ref, dtype=self.dtype))

    TASKS = {"ref": (0, ()), "helix": (1, ("trajAmp", "trajFreq", "dz", "useY")),
             "straightAcc": (2, ("tduration", "vdes")), "flip": (3, ("tstart", "tend")),
             "perch": (4, ("tend", "trotstart", "trotend", "vdes"))}
    TASK_DEFAULTS = {"trajAmp": 80, "trajFreq": 1, "dz": 0.15, "useY": True, "tduration": 500, "vdes": None,
                     "tstart": 100, "tend": None, "trotstart": 100, "trotend": 450}

    def set_task(self, name, t_ms=0.0, **kw):
        """On-device reference generator (template/flight_tasks.py, same keyword names and defaults).
        With a task other than "ref", rows 0..2 of `self.ref` are the robots' initialPos."""
        tid, names = self.TASKS[name]
        dflt = dict(self.TASK_DEFAULTS)
        dflt["vdes"] = {"straightAcc": 2, "perch": 0.2}.get(name, 0)
        dflt["tend"] = {"flip": 200, "perch": 500}.get(name, 0)
        vals = [float(kw.pop(n, dflt[n])) for n in names] + [0.0] * (4 - len(names))
        if kw:
            raise TypeError("unknown task parameter(s) %r" % sorted(kw))
        arr = (C.c_double * 4)(*vals)
        self._task = (tid, arr)
        self._check(self.L.umpcBatchSetTask(self.h, tid, arr, float(t_ms)))

    def set_weights(self, weights):
        """Per-robot objective weights [8, B] = (ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom), or None."""
        if weights is None:
            self._weights = None
            self._check(self.L.umpcBatchSetWeights(self.h, None))
            return
        w = torch.as_tensor(weights, dtype=self.dtype).to(self.device).contiguous()
        assert w.shape == (8, self.B)
        self._weights = w  # keep alive: the library stores the pointer
        self._check(self.L.umpcBatchSetWeights(self.h, _ptr(w)))

    def set_step_kernel(self, mode):
        """"auto" (default: the all-assembly fp32 kernel / the fp64 kernel with the assembly ADMM loop where they apply)
        or "cpp" (fp32: the C++ kernel around the assembly ADMM loop; fp64: the C++ loop): ablation and cross-checks.
        "lane" / "quad" pin the form of the assembly path (one lane / one lane quad per robot; "auto" takes the quad form for
        B <= 16 384 in fp32, B <= 4 096 in fp64, counted on `global_batch`, the size of the WHOLE job, so that a shard takes the
        form the undivided batch takes): equal up to rounding, not bit for bit."""
        self._check(self.L.umpcBatchSetStepKernel(self.h, {"auto": 0, "cpp": 1, "lane": 2, "quad": 3}[mode]))

    M0_CA6 = (100.0, 100.0, 100.0, 3333.0, 3333.0, 1000.0)   # dynamicsTerms, template/ca6dynamics.py:5-10

    def set_wl(self, wl, Mdiag=M0_CA6):
        """Fuse the wrench-linearisation step into every MPC step (robobee_test_controllers.py:162-171):
        accdes -> (u4, w0) = wlConUpdate(h0, M0 accdes) -> actualT0 = w0[2] / M0[2,2] for the next step.
        wl: a BatchWLCon of the same B / dtype / device (its `u` [4,B] is the per-robot WL state, its `w0` [6,B]
        receives the wrench), or None to switch the coupling off."""
        if wl is None:
            self._wl = None
            self._check(self.L.umpcBatchSetWL(self.h, None, None, None, None))
            return
        assert wl.B == self.B and wl.dtype == self.dtype and wl.u.is_contiguous() and wl.w0.is_contiguous()
        self._wl = wl   # keeps u / w0 alive: the library stores the pointers
        md = (C.c_double * 6)(*[float(v) for v in Mdiag])
        with torch.cuda.device(self.device):
            self._check(self.L.umpcBatchSetWL(self.h, C.byref(wl.wl), md, _ptr(wl.u), _ptr(wl.w0)))

    @property
    def kernel_name(self):
        """The kernel the last rollout() / update() of this handle dispatched (umpcBatchKernelName)."""
        return self.L.umpcBatchKernelName(self.h).decode()

    @property
    def time_ms(self):
        return float(self.L.umpcBatchTime(self.h))

    def rollout(self, K=1):
        """K closed-loop MPC steps (QP + nsub plant substeps each) in one launch."""
        with torch.cuda.device(self.device):
            self._check(self.L.umpcBatchRollout(self.h, int(K), _ptr(self.state), _ptr(self.ctrl), _ptr(self.ref),
                                                _ptr(self.actualT0), _ptr(self.Ib), _ptr(self.gain), _ptr(self.out),
                                                _ptr(self.stats), _ptr(self.status), _ptr(self.info), self._stream()))

    def reactive_rollout(self, nsteps, gains=None, every=1):
        """controlTest(useMPC=False) (template/uprightmpc2.py:121-151) for every robot: `nsteps` plant substeps with
        reactiveController (template/template_controllers.py:282-296) evaluated every `every` substeps. gains:
        [6, B] tensor (kpos0, kpos1, kz0, kz1, ks0, ks1) or None for the reference's defaults. Last command in
        self.out[0:3]; statistics accumulate in self.stats."""
        if gains is not None:
            gains = torch.as_tensor(gains, dtype=self.dtype, device=self.device).contiguous()
            assert gains.shape == (6, self.B)
        with torch.cuda.device(self.device):
            self._check(self.L.umpcBatchReactive(self.h, int(nsteps), int(every), _ptr(self.state), _ptr(self.ref),
                                                 _ptr(gains), _ptr(self.Ib), _ptr(self.gain), _ptr(self.out),
                                                 _ptr(self.stats), self._stream()))

    def task_reference(self, t_ms):
        """(pdes, dpdes, sdes) [9, B] of the current task at time t_ms (template/flight_tasks.py:6-49)."""
        out = torch.empty((9, self.B), dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.L.umpcBatchTaskReference(self.h, float(t_ms), _ptr(self.ref), _ptr(out), self._stream()))
        return out

    def control_test_log(self, tend, robots=(0,), use_mpc=True, gains=None, fire=None):
        """The log of controlTest (template/uprightmpc2.py:113,150-159) for the selected robots, in the reference's
        layout -- a dict {'t' [Nt], 'y' [Nt, 12] = (p, Rb[:, 2], dq), 'u' [Nt, 3], 'pdes' [Nt, 3], 'accdes' [Nt, 6]}
        per robot that viewControlTestLog / logMetric (:14-84, :161-175) take as is -- plus 'metric'. The loop runs
        from the current state, one launch per substep: this is the logging path, not the throughput path.
        fire: None = the fixed schedule (an MPC step every nsub substeps, the first at substep 0), or the substep
        indices at which the MPC fires -- the reference fires when `tt[ti] - thlPrev > hlInterval` (:136), which in
        floating point gives gaps of 25 / 26 substeps starting at substep 26; before the first fire the command is
        zero (:118) and between fires it is held. Returns {robot: log}."""
        nsub, dts = int(self.prm.nsub), float(self.prm.dtsim)
        Nt = int(np.ceil(tend / dts - 1e-9))
        idx = torch.as_tensor(list(robots), device=self.device)
        rec = {k: [] for k in ("y", "u", "pdes", "accdes", "R")}
        t_start = self.time_ms
        acc_now = torch.zeros((6, len(robots)), dtype=self.dtype, device=self.device)
        tl = float(self.prm.taulim)
        fire_at = None if fire is None else {int(i) for i in np.asarray(fire).ravel()}
        u = torch.zeros((3, self.B), dtype=self.dtype, device=self.device)      # uquad = np.zeros(3), :118
        for ti in range(Nt):
            t = t_start + ti * dts
            fire = (ti % nsub == 0) if fire_at is None else (ti in fire_at)
            if use_mpc and fire:
                self._check(self.L.umpcBatchSetTask(self.h, self._task_id(), self._task_params(), t))
                self.update()
                u = self.out[0:3].clone()
                acc_now = self.out[3:9][:, idx].clone()
            pd = self.task_reference(t)[0:3][:, idx]
            if use_mpc:
                u[1:3].clamp_(-tl, tl)
                self.plant(u, 1)
                ulog = u[:, idx]
            else:
                self.reactive_rollout(1, gains)
                ulog = self.out[0:3][:, idx]
            st = self.state[:, idx]
            rec["y"].append(torch.cat((st[0:3], st[9:12], st[12:18]), 0).T.cpu().numpy().copy())
            rec["R"].append(st[3:12].T.cpu().numpy().copy())          # column-major Rb (for save_viewlog's quaternion)
            rec["u"].append(ulog.T.cpu().numpy().copy())
            rec["pdes"].append(pd.T.cpu().numpy().copy())
            rec["accdes"].append((acc_now if (use_mpc and fire) else torch.zeros_like(acc_now)).T.cpu().numpy().copy())
        if use_mpc:   # leave the handle's clock where the loop ended
            self._check(self.L.umpcBatchSetTask(self.h, self._task_id(), self._task_params(), t_start + Nt * dts))
        logs = {}
        tt = np.arange(Nt) * dts
        for c, r in enumerate(robots):
            lg = {"t": tt.copy()}
            for k in rec:
                lg[k] = np.stack([a[c] for a in rec[k]]).astype(np.float64)
            perr, tau = lg["y"][:, :3], lg["u"][:, 1:3]
            lg["metric"] = (float((perr ** 2).sum() / Nt), float((tau ** 2).sum() / Nt))
            logs[r] = lg
        return logs

    def _task_id(self):
        return getattr(self, "_task", (0, (C.c_double * 4)()))[0]

    def _task_params(self):
        return getattr(self, "_task", (0, (C.c_double * 4)()))[1]

    def update(self):
        """One controller step on the current state (= umpcUpdate for every robot); no plant."""
        with torch.cuda.device(self.device):
            self._check(self.L.umpcBatchUpdate(self.h, _ptr(self.state), _ptr(self.ctrl), _ptr(self.ref),
                                               _ptr(self.actualT0), _ptr(self.Ib), _ptr(self.out), _ptr(self.status),
                                               _ptr(self.info), self._stream()))

    def plant(self, u, nsub=1):
        u = torch.as_tensor(u, dtype=self.dtype, device=self.device).contiguous()
        with torch.cuda.device(self.device):
            self._check(self.L.umpcBatchPlant(self.h, int(nsub), _ptr(self.state), _ptr(u), _ptr(self.Ib),
                                              _ptr(self.gain), self._stream()))

    def assemble(self):
        """Raw QP data (l, u, q, Px, Ax) for the current state; parity/debug."""
        z = lambda r: torch.empty((r, self.B), dtype=self.dtype, device=self.device)
        l, u, q, Px, Ax = z(39), z(39), z(45), z(45), z(48)
        with torch.cuda.device(self.device):
            self._check(self.L.umpcBatchAssemble(self.h, _ptr(self.state), _ptr(self.ctrl), _ptr(self.ref),
                                                 _ptr(self.Ib), _ptr(l), _ptr(u), _ptr(q), _ptr(Px), _ptr(Ax),
                                                 self._stream()))
        return l, u, q, Px, Ax

    def metrics(self, nsteps):
        """logMetric pair per robot (template/uprightmpc2.py:161-175): mean |p|^2, mean |tau|^2
        over the nsteps*nsub plant substeps accumulated so far."""
        n = max(1, int(nsteps) * int(self.prm.nsub))
        return self.stats / n


def save_viewlog(f1, log, timestamp=None):
    """Writes one robot's control_test_log() as the file template/viewlog.py reads (saveLog :24-33, readFile :50-56):
    gzip-pickled dict {'t','q','dq','u','accdes','posdes'} of arrays, one row per >= 0.999 ms like appendLog (:11-22),
    q = (p, quaternion xyzw of Rb), named `<f1>_<YYYYmmddHHMMSS>.zip`. Returns the file name."""
    import gzip
    import pickle
    import time
    from scipy.spatial.transform import Rotation
    t = np.asarray(log["t"], np.float64)
    keep, last = [], -np.inf
    for i, ti in enumerate(t):
        if ti - last >= 0.999:
            keep.append(i)
            last = ti
    keep = np.array(keep, int)
    Rb = np.asarray(log["R"])[keep].reshape(-1, 3, 3).transpose(0, 2, 1)          # stored column-major
    y = np.asarray(log["y"])[keep]
    data = {"t": t[keep], "q": np.hstack((y[:, 0:3], Rotation.from_matrix(Rb).as_quat())), "dq": y[:, 6:12],
            "u": np.asarray(log["u"])[keep], "accdes": np.asarray(log["accdes"])[keep],
            "posdes": np.asarray(log["pdes"])[keep]}
    fname = "%s_%s.zip" % (f1, timestamp or time.strftime("%Y%m%d%H%M%S", time.localtime()))
    with gzip.GzipFile(fname, "wb") as zf:
        pickle.dump(data, zf)
    return fname


MODELS = {"ca6": (0, 18, 6, 30), "ThrustStrokeDev": (1, 12, 4, 12)}  # id, state rows, input rows, vf output rows


def model_vector_field(model, y, u):
    """ydot of the reference's ca6 / ThrustStrokeDev models (template/ca6dynamics.py:35-50,
    template/FlappingModels3D.py:19-38) for a batch: y [ny,B], u [nu,B] CUDA tensors. ca6 also returns the
    wrench and the bias h: rows [ydot 18 | w 6 | h 6]."""
    mid, ny, nu, nout = MODELS[model]
    L = _lib.lib()
    B = y.shape[1]
    assert y.shape == (ny, B) and u.shape == (nu, B) and y.is_cuda and y.dtype == u.dtype
    y, u = y.contiguous(), u.contiguous()
    aux = torch.empty((nout, B), dtype=y.dtype, device=y.device)
    with torch.cuda.device(y.device):
        rc = L.umpcBatchModel(mid, B, _DT[y.dtype], 0, 0.0, _ptr(y), _ptr(u), _ptr(aux),
                              C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
    if rc:
        raise RuntimeError(L.umpcLastError().decode())
    return aux


def model_rk4(model, y, u, dt, nsub=1):
    """Advance y in place by nsub RK4 steps of dt under constant u (build-defined integrator)."""
    mid, ny, nu, _ = MODELS[model]
    L = _lib.lib()
    B = y.shape[1]
    assert y.shape == (ny, B) and u.shape == (nu, B) and y.is_cuda and y.is_contiguous() and nsub >= 1
    u = u.contiguous()
    with torch.cuda.device(y.device):
        rc = L.umpcBatchModel(mid, B, _DT[y.dtype], int(nsub), float(dt), _ptr(y), _ptr(u), None,
                              C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
    if rc:
        raise RuntimeError(L.umpcLastError().decode())
    return y


class BatchWLCon:
    """B wrench-linearisation controllers (the step that consumes accdes, SURVEY 8f-1;
    template/uprightmpc2/funapprox.c:118-165), one lane each. `u` [4,B] is the input state."""

    def __init__(self, B, u0, umin, umax, dumax, Qw, controlRate, popts, dtype=torch.float32, device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchWLCon needs a HIP device; there is no CPU path")
        self.L = _lib.lib()
        self.B, self.dtype, self.device = int(B), dtype, torch.device(device)
        self.wl = _lib.WLCon_t()
        f = lambda a, n: np.ascontiguousarray(np.asarray(a, np.float32).reshape(n))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        self.L.wlConInit(C.byref(self.wl), fp(f(u0, 4)), fp(f(umin, 4)), fp(f(umax, 4)), fp(f(dumax, 4)),
                         fp(f(Qw, 6)), C.c_float(controlRate), fp(f(popts, 90)))
        self.u = torch.as_tensor(np.asarray(u0, np.float64)).to(dtype).to(self.device)[:, None].repeat(1, self.B).contiguous()
        self.w0 = torch.zeros((6, self.B), dtype=dtype, device=self.device)

    def update(self, h0, pdotdes):
        h0 = torch.as_tensor(h0, dtype=self.dtype, device=self.device).contiguous()
        pd = torch.as_tensor(pdotdes, dtype=self.dtype, device=self.device).contiguous()
        assert h0.shape == (6, self.B) and pd.shape == (6, self.B)
        with torch.cuda.device(self.device):
            rc = self.L.umpcBatchWLUpdate(C.byref(self.wl), self.B, _DT[self.dtype], _ptr(self.u), _ptr(h0), _ptr(pd),
                                          _ptr(self.w0), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc:
            raise RuntimeError(self.L.umpcLastError().decode())
        return self.u, self.w0

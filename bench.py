#!/usr/bin/env python3
"""Headline benchmark: closed-loop MPC steps/sec (QP + plant), BASELINE.json metric.

Workload (BASELINE.json configs[2], the one the >=1e6 steps/s target is quoted
on; SURVEY 8d "Config 3"): B = 65536 robots PER GPU, fp32, horizon N = 3,
random-tilt hover start (seed 20201118), closed loop. One "step" = one pass of
the hot path over the whole batch = for every robot one umpcUpdate-equivalent
(assembly + 10 Ruiz passes + LDL' + 50 ADMM iterations + status + extraction)
followed by 25 plant substeps of 0.2 ms, moments clipped at +-100. By default the
K timed steps run as ONE persistent launch of the step kernel (the closed loop is
a rollout; --steps-per-launch 1 gives one launch per step); inputs are resident
in HBM before the timed region and every step round-trips its state through HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Robots are independent: ranks shard the batch (weak scaling, 65536 robots per
GPU, RNG keyed by global robot index), there is no collective on the data path;
one RCCL all_gather of the per-robot trajectory statistics runs AFTER the timed
region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_STEP_FP32 = 1208  # (18 plant + 124 controller + 9 ref) read + (18+124+9) written, x4 B
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def profile_json(name):
    """A committed rocprofv3 summary under profiles/ (PMC counters cannot be read live from inside the process)."""
    for rnd in ("r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s.json" % (rnd, name))
        if os.path.exists(path):
            try:
                j = json.load(open(path))
                j["_file"] = os.path.relpath(path, ROOT)
                return j
            except Exception:
                pass
    return None


def self_launch(argv, n):
    """`python bench.py --gpus N` outside torchrun: start N ranks as CHILD processes of torch.distributed.run
    (127.0.0.1 rendezvous on a free port) before this process has touched the GPU, pass their output through and
    exit with their code. Nothing is exec'd from a process that has initialised HIP."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def reference_anchor():
    """The reference's OWN umpcUpdate (oracle/_ref/libumpc_ref.so: its C sources compiled where they lie by
    oracle/Makefile in the authoring container; the prebuilt file travels with the tree) timed on this host, single
    thread, arguments marshalled once -- reported next to the port when the file is present."""
    try:
        import ctypes as C
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import refbind
        if not refbind.available():
            return None
        r = refbind.RefUMPC()
        rng = np.random.default_rng(1)
        n = 2000
        ab = rng.uniform(-0.5, 0.5, (n, 2))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        uq, ac = np.zeros(3, np.float32), np.zeros(6, np.float32)
        z3, e3 = np.zeros(3, np.float32), np.array([0, 0, 1], np.float32)
        dq = np.zeros(6, np.float32); dq[0] = 0.1
        Rs = []
        for a, b in ab:   # Rx(a) Ry(b), column-major (uprightmpc2.c:219)
            ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
            Rs.append(np.array([cb, sa * sb, -ca * sb, 0, ca, sa, sb, -sa * cb, ca * cb], np.float32))
        args = [(C.byref(r.up), fp(uq), fp(ac), fp(z3), fp(R), fp(dq), fp(z3), fp(z3), fp(e3), C.c_float(-1.0)) for R in Rs]
        f = r.lib.umpcUpdate
        for k in range(50):
            f(*args[k])
        t0 = time.perf_counter()
        for k in range(n):
            f(*args[k])
        dt_call = (time.perf_counter() - t0) / n
        g = r.lib.osqp_update_max_iter   # a trivial exported function: the cost of one ctypes call with arguments
        ws = (C.c_void_p * 26).in_dll(r.lib, "workspace")
        t0 = time.perf_counter()
        for k in range(n):
            g(C.c_void_p(C.addressof(ws)), C.c_int(50))
        over = (time.perf_counter() - t0) / n
        return {"umpcUpdate_us": dt_call * 1e6, "ctypes_call_us": over * 1e6, "calls": n,
                "what": "the reference C umpcUpdate (OSQP 0.6.0 embedded, fp32, 50 it), 1 thread, random-tilt hover states; "
                        "controller step only, no plant"}
    except Exception as ex:  # the anchor is optional: never let it break the bench line
        return {"error": repr(ex)[:200]}


def valu_issue(valu_per_wave_step, B, spl, kern_ms, src, robots_per_wave=64):
    """Vector-instruction issue rate of the step kernel against BOTH peaks: the hardware's (MI355X_MICROARCH.md constants
    table: a wave64 VALU op occupies a SIMD-32 for 2 cycles, reached with >= 2 waves per SIMD) and the lone-wave rate
    (one wave per SIMD issues a VALU op every 4 cycles; this kernel's 512-register waves run one per SIMD by design)."""
    ach = valu_per_wave_step * (B / robots_per_wave) * spl / (kern_ms * 1e-3) / 1e9
    p2, p4 = 1024 * 2.4 / 2, 1024 * 2.4 / 4
    return {"valu_instr_per_wave_step": valu_per_wave_step, "robots_per_wave": robots_per_wave, "achieved": ach, "unit": "G wave-instr/s",
            "peak": p2, "frac": ach / p2, "peak_simd32": p2, "frac_simd32": ach / p2,
            "peak_lone_wave": p4, "frac_lone_wave": ach / p4, "source": src,
            "note": "VALU instructions only (SQ_INSTS_VALU per wave-step, profile-derived). peak / frac = the SIMD-32 issue "
                    "peak, 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 op; *_lone_wave = 1024 SIMDs x 2.4 GHz / 4, the rate "
                    "ONE wave per SIMD can issue at (the kernel's occupancy: 512 registers + 40 KiB LDS per wave)"}


class DryRunMPC:
    """--dry-run stand-in for BatchUprightMPC on CPU: no kernel, statistics = the global robot index, so the
    launcher / rendezvous / gather path of bench.py can run under gloo in the CPU test-suite."""

    def __init__(self, B, state, rank):
        import torch
        self.B, self.state, self.rank = B, state, rank
        self.status = torch.ones(B, dtype=torch.int32)
        self.Ib = self.gain = None
        self.nsteps = 0

    def rollout(self, K):
        self.nsteps += K
        time.sleep(1e-3 * (1 + self.rank))      # ranks finish at different times: the line reports the maximum

    def metrics(self, nsteps):
        import torch
        assert nsteps == self.nsteps
        idx = torch.arange(self.rank * self.B, (self.rank + 1) * self.B, dtype=torch.float32)
        return torch.stack((idx, torch.zeros_like(idx)))


def cpu_baseline(args, plant_mode):
    """The oracle (CPU restatement, OpenMP over robots) on a bounded sample of the
    same workload, timed on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oraclebind
    from robobee3d_amd.batch import hover_initial_conditions
    oraclebind.build()
    # the one-GPU box exposes every host thread but grants a 16-CPU share: use that many threads
    ncores = min(len(os.sched_getaffinity(0)), int(os.environ.get("UMPC_CPU_THREADS", "16")))
    Bs, Ks = args.cpu_robots, args.cpu_steps
    st, ref = hover_initial_conditions(Bs, 20201118, np.float32)
    ctrl = np.zeros((127, Bs), np.float32)
    ctrl[124:] = 1
    # warm (page-in, thread pool)
    oraclebind.batch_rollout(st.copy(), ctrl.copy(), ref, 1, dtype=np.float32, plant_mode=plant_mode, nthreads=ncores)
    t0 = time.perf_counter()
    oraclebind.batch_rollout(st, ctrl, ref, Ks, dtype=np.float32, plant_mode=plant_mode, nthreads=ncores)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    n1 = max(1, Bs // 64)
    st1, ref1 = hover_initial_conditions(n1, 20201118, np.float32)
    c1 = np.zeros((127, n1), np.float32); c1[124:] = 1
    oraclebind.batch_rollout(st1, c1, ref1, Ks, dtype=np.float32, plant_mode=plant_mode, nthreads=1)
    dt1 = time.perf_counter() - t1
    # the same sample on the fp64 build of the oracle, single thread (fp32 and fp64, 1 thread and OpenMP)
    st2, ref2 = hover_initial_conditions(n1, 20201118, np.float64)
    c2 = np.zeros((127, n1), np.float64); c2[124:] = 1
    oraclebind.batch_rollout(st2.copy(), c2.copy(), ref2, 1, dtype=np.float64, plant_mode=plant_mode, nthreads=1)  # load + warm
    t2 = time.perf_counter()
    oraclebind.batch_rollout(st2, c2, ref2, Ks, dtype=np.float64, plant_mode=plant_mode, nthreads=1)
    dt2 = time.perf_counter() - t2
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": Bs * Ks / dt, "unit": "closed-loop MPC steps/s", "cores": ncores, "kind": "port",
            "reference": reference_anchor(),
            "cpu_model": model, "host_threads_visible": len(os.sched_getaffinity(0)),
            "sample": "%d robots x %d steps of the same workload (oracle/umpc_oracle.c, fp32, OpenMP over robots)"
                      % (Bs, Ks),
            "single_thread_value": n1 * Ks / dt1, "fp64_single_thread_value": n1 * Ks / dt2}


def _timed(launch, steps, warmup, dev, spl=None):
    """W untimed steps, then EXACTLY K timed ones: barrier-less one-GPU form of the main protocol (synchronise, wall clock
    and HIP events on the launch stream around the K steps, synchronise). launch(k) enqueues k steps. The FIRST pass is the
    figure of record (the literal W / K protocol). Like the headline, a pass shorter than the chip's clock ramp (< 25 ms) is
    repeated behind >= 30 ms of the same work and that second pass is reported BESIDE it as the loaded-clock figure.
    Returns (seconds, mean milliseconds per launch from the events, launches, loaded-clock pass seconds or None)."""
    import torch
    spl = steps if not spl else spl
    n = steps // spl

    def one_pass():
        w = warmup
        while w > 0:
            launch(min(spl, w))
            w -= min(spl, w)
        torch.cuda.synchronize(dev)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        t0 = time.perf_counter()
        for k in range(n):
            evs[k][0].record()
            launch(spl)
            evs[k][1].record()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, float(np.mean([a.elapsed_time(b) for a, b in evs]))
    dt, kms = one_pass()
    loaded = None
    if dt * (1 + warmup / max(1, steps)) < 25e-3:
        launch(int(np.ceil(30e-3 / (dt / steps))))
        loaded, _ = one_pass()
    return dt, kms, n, loaded


def _guard(out, key, fn):
    """One side configuration: whatever goes wrong in it is recorded under its key and never costs the headline
    measurement, which has been taken by then."""
    try:
        fn()
    except Exception as ex:     # noqa: BLE001 -- recorded, not swallowed
        out[key] = {"error": repr(ex)[:400]}


def gain_grid_weights(B):
    """template/uprightmpc2.py:272-303 (gainTuningSims): the 10 x 10 (wpr, wvr) grid, tiled over B robots; rows in the order
    of umpcBatchSetWeights (ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom)."""
    g1, g2 = np.meshgrid(np.logspace(-2, 1, 10), np.logspace(1, 4, 10))
    W = np.tile(np.array([1e1, 1e3, 1, 5, 1e3, 2e3, 1e-1, 1e-2])[:, None], (1, B))
    W[2], W[4] = np.resize(g1.ravel(), B), np.resize(g2.ravel(), B)
    return W


def side_configs(dev, steps, warmup):
    """The other single-GPU configurations of BASELINE.json, driver-timed by the SAME command as the headline (after its
    timed region): config 2 (configs[1]: fp64, B = 4096, the reference's Euler + expm plant), config 4 (configs[3]:
    planar p5f, N = 10, B = 16384) and one GPU's shard of config 5 (configs[4]: B = 2^17, per-robot inertia + thrust
    gain). Same protocol: W untimed steps, K timed steps, inputs resident, value = robots x K / wall time; the
    roofline record prices the dominant kernel's ALGORITHMIC bytes against 8 TB/s from the HIP-event duration, with
    the committed counter traffic of that kernel (per unit, labelled) beside it."""
    import torch
    from robobee3d_amd.batch import (BatchUprightMPC, hover_initial_conditions_device, monte_carlo_draws_device)
    from robobee3d_amd.batchqp import PlanarP5fMPC
    out = {}

    def roof(alg_per_unit, units, kern_ms, prof, key):
        ach = alg_per_unit * units / (kern_ms * 1e-3) / 1e9
        tr = src = None
        if prof is not None:
            per = prof[key]
            tr = (per["read_corrected"] + per["written"]) * units
            src = "%s: %.0f B read + %.0f B written per unit (rocprofv3 --pmc passes, gfx950 correction applied), x %d units" \
                  % (prof["_file"], per["read_corrected"], per["written"], units)
        return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": tr, "traffic_source": src, "alg_bytes_per_launch": alg_per_unit * units}

    # ---- config 2: uprightmpc2 hover, B = 4096 random tilts, fp64 (seed 20201117) ----
    def cfg0():
        B = 4096
        m = BatchUprightMPC(B, torch.float64, device=dev, plant_mode=0)
        st, ref, _ = hover_initial_conditions_device(B, 20201117, torch.float64, device=dev)
        m.set_state(st, ref)
        dt, kms, n, loaded = _timed(m.rollout, steps, warmup, dev)
        out["config2_fp64_B4096"] = {
            "workload": "BASELINE configs[1]: uprightmpc2 hover, batch=4096 random initial tilts, fp64, Euler+expm plant, closed loop",
            "value": B * steps / dt, "unit": "steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "loaded_clocks_ms_per_step": None if loaded is None else loaded / steps * 1e3,
            "dtype": "f64", "robots": B, "kernel": m.kernel_name, "kernel_ms": kms, "steps_per_launch": steps,
            "roofline": roof(2 * ALG_BYTES_PER_STEP_FP32, B * steps, kms, profile_json("pmc_config2_f64"), "per_unit_bytes"),
            "check": {"nonfinite_state_values": int((~torch.isfinite(m.state)).sum().item()),
                      "status_solved_frac": float((m.status > 0).float().mean().item())}}
    _guard(out, "config2_fp64_B4096", cfg0)
    torch.cuda.empty_cache()
    # ---- config 4: planar p5f, N = 10, B = 16384, fp32 (seed 20201119) ----
    def cfg1():
        B = 16384
        mp = PlanarP5fMPC(B, torch.float32, device=dev)
        pert = np.random.default_rng(20201119).uniform(-0.1, 0.1, (2, B))
        mp.y[0] = torch.as_tensor(pert[0]).to(mp.y)
        mp.y[3] = torch.as_tensor(pert[1]).to(mp.y)
        tick = [2]

        def p5f_launch(k):
            for _ in range(k):
                mp.tick(0.002 * tick[0]); tick[0] += 1
        dt, kms, n, loaded = _timed(p5f_launch, steps, warmup, dev)
        sq = mp.qp.s
        alg = (2 * (sq.n + 2 * sq.m) + 15) * 4
        kn = mp.qp.kernel_name
        out["config4_p5f_B16384"] = {
            "workload": "BASELINE configs[3]: planar/mpc_osqp_p5f stroke-plane MPC, N=10 (n=87, m=164), 50 ADMM it, 10 Ruiz, "
                        "LDL' refactor per tick + Euler plant tick",
            "value": B * steps / dt, "unit": "steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "loaded_clocks_ms_per_step": None if loaded is None else loaded / steps * 1e3,
            "dtype": "f32", "robots": B,
            "kernel": "bqp_fixed_%s_asm_kernel" % kn[:-4] if kn.endswith("+asm") else kn,
            "kernel_ms": kms / steps, "kernel_ms_note": "HIP events around the %d ticks (gather, getLin, QP, plant kernels of a tick) / ticks" % steps,
            "roofline": roof(alg, B, kms / steps, profile_json("pmc_config4_p5f") if kn.endswith("+asm") else None, "per_unit_bytes"),
            "check": {"nonfinite_state_values": int((~torch.isfinite(mp.y)).sum().item()),
                      "status_solved_frac": float((mp.qp.status > 0).float().mean().item())}}
    _guard(out, "config4_p5f_B16384", cfg1)
    torch.cuda.empty_cache()
    # ---- config 5, one GPU's shard: B = 2^17, per-robot inertia and thrust gain, fp32, RK4 (seed 20201120) ----
    def cfg2():
        B = 131072
        m = BatchUprightMPC(B, torch.float32, device=dev, plant_mode=1)
        st, ref, _ = hover_initial_conditions_device(B, 20201118, torch.float32, device=dev)
        m.set_state(st, ref)
        m.Ib, m.gain = monte_carlo_draws_device(B, 20201120, torch.float32, device=dev)
        dt, kms, n, loaded = _timed(m.rollout, steps, warmup, dev)
        out["config5_shard_B131072"] = {
            "workload": "BASELINE configs[4], one GPU's shard: Monte-Carlo mass/inertia sweep, 2^17 robots (of 2^20 over 8 GPUs), "
                        "per-robot Ib (controller + plant) and thrust gain +-20 %, fp32, RK4 plant",
            "value": B * steps / dt, "unit": "steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "loaded_clocks_ms_per_step": None if loaded is None else loaded / steps * 1e3,
            "dtype": "f32", "robots": B, "kernel": m.kernel_name, "kernel_ms": kms, "steps_per_launch": steps,
            "roofline": roof(ALG_BYTES_PER_STEP_FP32 + 16, B * steps, kms, profile_json("pmc_config5_shard"), "per_unit_bytes"),
            "check": {"nonfinite_state_values": int((~torch.isfinite(m.state)).sum().item()),
                      "status_solved_frac": float((m.status > 0).float().mean().item())}}
    _guard(out, "config5_shard_B131072", cfg2)
    torch.cuda.empty_cache()
    # ---- the reference's 10 x 10 (wpr, wvr) gain-tuning grid (template/uprightmpc2.py:272-303) as ONE batch
    # of the headline size: per-robot objective weights, same kernel as the headline ----
    def cfg3():
        B = 65536
        m = BatchUprightMPC(B, torch.float32, device=dev, plant_mode=1)
        st, ref, _ = hover_initial_conditions_device(B, 20201118, torch.float32, device=dev)
        m.set_state(st, ref)
        m.set_weights(gain_grid_weights(B))
        dt, kms, n, loaded = _timed(m.rollout, steps, warmup, dev)
        out["f3_gain_sweep_B65536"] = {
            "workload": "SURVEY 8(f-3): 10x10 (wpr, wvr) gain grid of gainTuningSims tiled over 65536 robots, per-robot objective "
                        "weights, closed loop, fp32, RK4 plant",
            "value": B * steps / dt, "unit": "steps/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "loaded_clocks_ms_per_step": None if loaded is None else loaded / steps * 1e3,
            "dtype": "f32", "robots": B, "kernel": m.kernel_name, "kernel_ms": kms, "steps_per_launch": steps,
            "roofline": roof(ALG_BYTES_PER_STEP_FP32 + 32, B * steps, kms, profile_json("pmc_f3_gain_sweep"), "per_unit_bytes"),
            "check": {"nonfinite_state_values": int((~torch.isfinite(m.state)).sum().item()),
                      "status_solved_frac": float((m.status > 0).float().mean().item())}}
    _guard(out, "f3_gain_sweep_B65536", cfg3)
    torch.cuda.empty_cache()
    return out


def main_p5f(args):
    """BASELINE configs[3]: one step = one tick of planar/mpc_osqp_p5f.py:157-176 for every
    robot = getLin + A rebuild + one 50-iteration QP step (10 Ruiz passes, LDL' refactor) + the reference's plant
    tick. Results are parity-unpinned in the reference (it never solves); tests/test_bqp.py checks the kernel
    against the table oracle."""
    import torch
    import torch.distributed as dist
    from robobee3d_amd import shard
    from robobee3d_amd.batchqp import PlanarP5fMPC
    rank, world, local_rank = shard.world()
    dev = torch.device("cuda", local_rank)
    if args.gpus > 1 or world > 1:
        assert world == args.gpus
        torch.cuda.set_device(local_rank)
        shard.init("nccl", device=dev)
    B = args.batch if args.batch != 65536 else 16384
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    mpc = PlanarP5fMPC(B, tdt, device=dev, max_iter=args.max_iter)
    lo, _ = shard.robot_range(B, rank)
    rng = np.random.default_rng(20201119)
    pert = rng.uniform(-0.1, 0.1, (2, world * B))[:, lo:lo + B]       # sigma, phi per robot
    mpc.y[0] = torch.as_tensor(pert[0]).to(mpc.y)
    mpc.y[3] = torch.as_tensor(pert[1]).to(mpc.y)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
    ti = 2
    for _ in range(args.warmup):
        mpc.tick(0.002 * ti); ti += 1
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        mpc.tick(0.002 * ti); ti += 1
    e1.record()
    torch.cuda.synchronize(dev)
    local = time.perf_counter() - t0
    barrier()
    elapsed = shard.max_over_ranks(local, device=dev)
    s = mpc.qp.s
    esz = 4 if args.dtype == "f32" else 8
    # algorithmic bytes per robot-tick: iterates x, y, z read + written, state 7 + 7, nominal input 1
    alg = (2 * (s.n + 2 * s.m) + 15) * esz
    if rank == 0:
        kern_ms = e0.elapsed_time(e1) / args.steps
        # HBM bytes: the committed rocprofv3 --pmc passes of this kernel (per robot-tick, scaled to this batch), when this run
        # is the profiled kernel configuration
        traffic, traffic_src = None, None
        j = profile_json("pmc_config4_p5f")
        if j is not None and args.dtype == "f32" and args.max_iter == 50 and mpc.qp.kernel_name.endswith("+asm"):
            per = j["per_unit_bytes"]
            traffic = (per["read_corrected"] + per["written"]) * B
            traffic_src = "%s: %.0f B read + %.0f B written per robot-tick (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at B = %d, " \
                          "gfx950 correction applied; the loop re-reads its read-only stream every iteration) x %d robots" \
                          % (j["_file"], per["read_corrected"], per["written"], j["batch"], B)
        print(json.dumps({
            "metric": "closed-loop MPC steps/sec (QP+dyn)", "value": world * B * args.steps / elapsed, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: planar/mpc_osqp_p5f stroke-plane MPC, N=10 (n=87, m=164, "
                                   "nnz(L)=%d), %d ADMM it, 10 Ruiz, LDL' refactor per tick + Euler plant tick" % (s.nnzL, args.max_iter),
                       "robots_per_gpu": B, "global_batch": world * B, "horizon": 10,
                       "parallelism": "robots sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": alg * B / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg * B / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": {"tables": "bqp_solve_kernel", "wave": "bqp_wave_kernel"}.get(
                             mpc.qp.kernel_name, "bqp_fixed_%s_asm_kernel" % mpc.qp.kernel_name[:-4] if mpc.qp.kernel_name.endswith("+asm")
                             else "bqp_fixed_%s_kernel" % mpc.qp.kernel_name),
                         "kernel_ms": kern_ms, "alg_bytes_per_launch": alg * B},
            "check": {"nonfinite_state_values": int((~torch.isfinite(mpc.y)).sum().item()),
                      "status_solved_frac": float((mpc.qp.status > 0).float().mean().item())}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--batch", type=int, default=65536, help="robots per GPU")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--plant", default="rk4", choices=["euler", "rk4"],
                    help="rk4 = RK4 substeps of the reference's vector field (what BASELINE configs[2] names; "
                         "build-defined, SURVEY finding 2); euler = the reference's own Euler+expm step (parity mode)")
    ap.add_argument("--max-iter", type=int, default=50, help="ADMM iterations (50 = the reference; other values are diagnostics)")
    ap.add_argument("--nsub", type=int, default=25, help="plant substeps per MPC step (25 = the metric; 0 = QP only)")
    ap.add_argument("--steps-per-launch", type=int, default=0,
                    help="closed-loop steps carried by one kernel launch (0 = all K timed steps in one launch; "
                         "1 = one launch per step)")
    ap.add_argument("--workload", default="uprightmpc2", choices=["uprightmpc2", "p5f"],
                    help="uprightmpc2 = the headline metric (BASELINE configs[2]); p5f = BASELINE configs[3], the planar "
                         "stroke-plane MPC (N = 10, n = 87, m = 164) on the general-structure solver, default batch 16384")
    ap.add_argument("--monte-carlo", action="store_true",
                    help="BASELINE configs[4]: per-robot inertia (controller + plant) and plant thrust gain, +-20 %%")
    ap.add_argument("--gain-sweep", action="store_true",
                    help="SURVEY 8(f-3): per-robot objective weights, the 10 x 10 (wpr, wvr) grid of gainTuningSims tiled over the batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-precondition", action="store_true",
                    help="skip the second, loaded-clock pass (`loaded_clocks`); `value` is the first W + K pass either way")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the driver-timed side configurations (BASELINE configs[1], [3], [4]'s shard) that a default "
                         "one-GPU run of the headline workload appends under \"configs\"")
    ap.add_argument("--side-steps", type=int, default=20)
    ap.add_argument("--side-warmup", type=int, default=5)
    ap.add_argument("--cpu-robots", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=100)
    ap.add_argument("--dry-run", action="store_true",
                    help="TEST ONLY (tests/test_shard_gloo.py): run the launcher, rendezvous (gloo), sharding, barriers, "
                         "max-over-ranks timing and the statistics gather on CPU with NO kernel; the line says dry_run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (children only; this process never touches the GPU)
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    if args.workload == "p5f":
        return main_p5f(args)
    import torch
    import torch.distributed as dist
    from robobee3d_amd.batch import hover_initial_conditions
    from robobee3d_amd import _lib

    from robobee3d_amd import shard
    rank, world, local_rank = shard.world()
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    if args.dry_run:
        dev = torch.device("cpu")
        shard.init("gloo")
    else:
        from robobee3d_amd.batch import BatchUprightMPC
        dev = torch.device("cuda", local_rank)
        force = os.environ.get("UMPC_FORCE_DIST") == "1"    # rehearse the RCCL path with one rank on a one-GPU box
        if world > 1 or force:
            torch.cuda.set_device(local_rank)
            shard.init("nccl", device=dev, force=force)   # "nccl" is RCCL on ROCm
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    ndt = np.float32 if args.dtype == "f32" else np.float64
    plant_mode = 0 if args.plant == "euler" else 1
    B = args.batch

    # synthetic inputs, resident in HBM before the timed region
    lo, _hi = shard.robot_range(B, rank)   # weak scaling: B robots per GPU, RNG keyed by the global index
    if args.dry_run:
        st, ref = hover_initial_conditions(B, 20201118, ndt, index_offset=lo)
        mpc = DryRunMPC(B, torch.as_tensor(st), rank)
    else:
        # inputs are generated ON the device from the counter hash: no host-side pass over the batch per rank
        from robobee3d_amd.batch import hover_initial_conditions_device, monte_carlo_draws_device
        # global_batch: the lane / quad form of the step kernel is chosen from the size of the WHOLE job, so a shard runs
        # the instruction stream the undivided batch would (results are partition-invariant)
        mpc = BatchUprightMPC(B, tdt, device=dev, global_batch=world * B, plant_mode=plant_mode, maxIter=args.max_iter,
                              nsub=args.nsub)
        st, ref, _ = hover_initial_conditions_device(B, 20201118, tdt, index_offset=lo, device=dev)
        mpc.set_state(st, ref)
    if args.monte_carlo:  # Ib = Ib0 (1 + d), d ~ U(-0.2, 0.2)^3, thrust gain 1 + U(-0.2, 0.2)
        if args.dry_run:
            from robobee3d_amd.batch import monte_carlo_draws
            Ib, gain = monte_carlo_draws(B, 20201120, ndt, index_offset=lo)   # keyed by the GLOBAL robot index
            mpc.Ib, mpc.gain = torch.as_tensor(Ib), torch.as_tensor(gain)
        else:
            mpc.Ib, mpc.gain = monte_carlo_draws_device(B, 20201120, tdt, index_offset=lo, device=dev)

    if args.gain_sweep and not args.dry_run:
        mpc.set_weights(gain_grid_weights(B))

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    spl = args.steps if args.steps_per_launch <= 0 else max(1, min(args.steps_per_launch, args.steps))
    assert args.steps % spl == 0, "--steps must be a multiple of --steps-per-launch"
    # EXACTLY `warmup` untimed steps, as multi-step launches like the timed ones (full launches of spl steps, then
    # the remainder as one shorter launch): the code object, the workspace pages and the clocks are warm and no
    # warm-up step runs in a shape the timed region does not use
    nlaunch = args.steps // spl
    cuda = dev.type == "cuda"

    def protocol():
        """W untimed steps, then EXACTLY K timed steps bracketed by barrier + synchronize on both sides; returns this
        rank's seconds, the maximum over ranks, every rank's seconds and the mean HIP-event milliseconds per launch."""
        w = args.warmup
        while w > 0:
            mpc.rollout(min(spl, w))
            w -= min(spl, w)
        barrier()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nlaunch)] if cuda else []
        t0 = time.perf_counter()
        for k in range(nlaunch):
            if cuda:
                evs[k][0].record()      # on the stream the kernel is launched on (torch's current stream)
            mpc.rollout(spl)        # ONE launch = spl closed-loop steps of all B robots
            if cuda:
                evs[k][1].record()
        if cuda:
            torch.cuda.synchronize(dev)
        local = time.perf_counter() - t0          # this rank's K steps, start barrier -> local completion
        kms = float(np.mean([a.elapsed_time(b) for a, b in evs])) if cuda else local / nlaunch * 1e3
        # The measurement exists from here on. The two small collectives that turn it into the job's figure (max over ranks,
        # every rank's time) are guarded: if the fabric fails NOW, rank 0 still prints what it measured itself, marked as such.
        try:
            barrier()
            return local, shard.max_over_ranks(local, device=dev), shard.gather_scalars(local, device=dev), kms
        except Exception as ex:     # noqa: BLE001 -- recorded under collectives.error, the line still prints
            coll_err.append("timing collectives: " + repr(ex)[:300])
            return local, local, [local], kms

    coll_err = []      # collective failures after the timed region: recorded in the line, never fatal to it
    local, elapsed, per_rank_s, kern_ms = protocol()
    nsteps_done = args.warmup + args.steps
    loaded, precond = None, None
    # `value` / `ms_per_step` are ALWAYS this first pass: the literal --warmup W / --steps K protocol, nothing run before it.
    # The chip needs ~25 ms of sustained load to reach its loaded clock (measured: profiles/r03_clock_ramp.txt -- behind a
    # long launch a 20-step launch runs at the 500-step rate, after idle it is 15-20 % slower and consecutive launches
    # ramp for ~22 ms), and W + K steps of 0.12 ms are far shorter than that ramp. So when the whole protocol took < 25 ms
    # the same W + K pass is run a SECOND time behind >= 30 ms of the same step kernel on the same batch and reported
    # BESIDE the value of record, under `loaded_clocks` (compare loaded with loaded and first-pass with first-pass across
    # rounds). --no-precondition skips the second pass.
    if not args.no_precondition and not args.dry_run and not coll_err and elapsed * (1 + args.warmup / max(1, args.steps)) < 25e-3:
        P = int(np.ceil(30e-3 / (elapsed / args.steps)))
        t0 = time.perf_counter()
        mpc.rollout(P)
        barrier()
        precond = {"steps": P, "ms": (time.perf_counter() - t0) * 1e3,
                   "what": "untimed launch of the same kernel on the same batch between the pass of record and the loaded_clocks pass"}
        _l2, e2, pr2, k2 = protocol()
        loaded = {"value": world * B * args.steps / e2, "ms_per_step": e2 / args.steps * 1e3, "kernel_ms": k2,
                  "ms_per_step_per_rank": [t / args.steps * 1e3 for t in pr2],
                  "effective_warmup_steps": 2 * args.warmup + args.steps + P,
                  "what": "the same W + K protocol repeated behind the preconditioning launch (chip at its loaded clock); "
                          "NOT the value of record"}
        nsteps_done = 2 * (args.warmup + args.steps) + P

    # end-of-run trajectory statistics: the only exchange of the path (RCCL all_gather over xGMI,
    # outside the timed region; 2 floats per robot)
    local_metric = mpc.metrics(nsteps_done)
    try:
        metric = shard.gather_stats(local_metric)
    except Exception as ex:     # noqa: BLE001 -- the headline measurement is already taken: report it with rank 0's own statistics
        coll_err.append("gather_stats: " + repr(ex)[:300])
        metric = local_metric
    status = mpc.status
    nbad = int((~torch.isfinite(mpc.state)).sum().item())
    if args.dry_run and rank == 0 and not coll_err:
        # the gather returned every rank's block in global robot order
        assert metric.shape[1] == world * B and torch.equal(metric[0], torch.arange(world * B, dtype=metric.dtype))

    if rank == 0:
        total_steps = world * B * args.steps
        value = total_steps / elapsed
        bps = ALG_BYTES_PER_STEP_FP32 * (2 if args.dtype == "f64" else 1)
        achieved = bps * B * spl / (kern_ms * 1e-3) / 1e9
        # PMC counters cannot be read from inside the process: `traffic` and the VALU instruction count are the
        # committed rocprofv3 --pmc measurements of this kernel (profiles/), kept PER ROBOT-STEP and scaled to this
        # launch (B x steps-per-launch); used only when the kernel configuration is the profiled one
        traffic, traffic_src, valu_per_wave_step, valu_src = None, None, None, None
        # ... and the KERNEL that ran (a batch <= 16 384 dispatches the quad form -- 16 robots per wavefront, other
        # counters -- which has its own committed passes; any other kernel gets no traffic / VALU figure rather than the lane form's)
        kname = mpc.kernel_name if not args.dry_run else None
        quad_form = kname == "umpc_rollout_asm_quad_kernel"
        robots_per_wave = 16 if quad_form else 64
        same_kernel = lambda j: (j is not None and j.get("dtype", "f32") == args.dtype and j.get("plant", "rk4") == args.plant
                                 and j.get("max_iter", 50) == args.max_iter and j.get("nsub", 25) == args.nsub
                                 and j.get("kernel") == kname)
        j = profile_json("pmc_config5_shard" if args.monte_carlo else "pmc_f3_gain_sweep" if args.gain_sweep
                         else "pmc_quad_B16384" if quad_form else "pmc_traffic")
        if j is not None and "per_unit_bytes" in j:
            j["per_robot_step_bytes"] = j["per_unit_bytes"]
        if same_kernel(j) and not (args.monte_carlo and args.gain_sweep):
            per = j["per_robot_step_bytes"]
            traffic = (per["read_corrected"] + per["written"]) * B * spl
            traffic_src = "%s: %.0f B read + %.0f B written per robot-step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, " \
                          "gfx950 correction applied) x %d robots x %d steps" % (j["_file"], per["read_corrected"], per["written"], B, spl)
        j = profile_json("pmc_config2_f64")      # the fp64 assembly-loop kernel (config 2) has its own counter passes
        if (traffic is None and j is not None and args.dtype == "f64" and args.plant == "euler" and args.max_iter == 50
                and args.nsub == 25 and not args.monte_carlo):
            per = j["per_unit_bytes"]
            traffic = (per["read_corrected"] + per["written"]) * B * spl
            traffic_src = "%s: %.0f B read + %.0f B written per robot-step (rocprofv3 --pmc passes at B = %d, gfx950 correction " \
                          "applied) x %d robots x %d steps" % (j["_file"], per["read_corrected"], per["written"], j["batch"], B, spl)
        j = profile_json("sq_counters_quad_B16384" if quad_form else "sq_counters")
        if j is not None and quad_form:       # (tools/profile_summary_options.py writes the kernel and batch, not the options)
            j.setdefault("dtype", "f32"); j.setdefault("plant", "rk4")
        if same_kernel(j):
            valu_per_wave_step = j["per_wave_step"]["valu_instructions"]
            valu_src = j["_file"]
        line = {
            "metric": "closed-loop MPC steps/sec (QP+dyn)", "value": value, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "ms_per_step_per_rank": [t / args.steps * 1e3 for t in per_rank_s],
            "loaded_clocks": loaded, "preconditioning": precond,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: closed-loop uprightmpc2 (N=3, 50 ADMM it, 10 Ruiz, LDL' "
                                   "refactor per step) + 25 plant substeps, random-tilt hover, seed 20201118",
                       "robots_per_gpu": B, "global_batch": world * B, "horizon": 3,
                       "plant": "Euler+expm (reference step)" if plant_mode == 0 else "RK4 (build-defined)",
                       "parallelism": "robots sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname,   # what the handle's last launch dispatched
                         "kernel_ms": kern_ms, "steps_per_launch": spl, "alg_bytes_per_launch": bps * B * spl,
                         # SURVEY 8d asks for all three rooflines; the one that binds is vector issue
                         "flops": {"achieved": 1.1e5 * B * spl / (kern_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                                   "peak": 157.3, "note": "~1.1e5 flop per robot-step (SURVEY 8d); MI355X fp32 vector peak"},
                         "lds": {"achieved": (args.max_iter * 76 * 16 + 2 * 640) * B * spl / (kern_ms * 1e-3) / 1e9 if args.dtype == "f32" else None,
                                 "unit": "GB/s", "peak": 256 * 128 * 2.4,
                                 "note": "76 ds_read_b128 per ADMM iteration per lane (asmstep.py's loop) + hand-off; peak 128 B/clk/CU"},
                         "valu_issue": valu_issue(valu_per_wave_step, B, spl, kern_ms, valu_src, robots_per_wave) if valu_per_wave_step else None,
                         "note": "path is VALU-issue bound, not HBM bound (DESIGN.md): ~1.1e5 flop per 1208 B"},
            "check": {"nonfinite_state_values": nbad,
                      "mean_pos_err_mm2": float(metric[0].mean().item()),
                      "status_solved_frac": float((status > 0).float().mean().item())},
        }
        if args.dry_run:
            line["dry_run"] = True
            # no kernel ran: nothing was measured. What the line WOULD carry (from the ranks' sleeps) is kept beside it so that
            # the CPU tests can see that a failed collective does not cost the figure
            line["dry_run_unmeasured"] = {"value": line["value"], "ms_per_step": line["ms_per_step"]}
            line["value"] = line["ms_per_step"] = None
            line["roofline"] = None
        default_headline = (args.dtype == "f32" and B == 65536 and args.max_iter == 50 and args.nsub == 25
                            and args.plant == "rk4" and not args.monte_carlo and not args.gain_sweep)
        if world == 1 and not args.dry_run and not args.no_side_configs and default_headline:
            del mpc
            torch.cuda.empty_cache()
            line["configs"] = side_configs(dev, args.side_steps, args.side_warmup)
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            line["cpu_baseline"] = cpu_baseline(args, plant_mode)
        if dist.is_initialized():
            line["collectives"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                   "gathered_robots": int(metric.shape[1])}
            if coll_err:
                # `value` then is rank 0's own time over world x B robots (weak scaling: every rank ran the same launch)
                line["collectives"]["error"] = coll_err
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        try:
            if not coll_err:
                dist.barrier()
            dist.destroy_process_group()
        except Exception:       # noqa: BLE001 -- shutting down a broken group: the line is out, exit code below tells
            pass
    if any(e_.startswith("timing") for e_ in coll_err):
        # the line was printed, but its value is ONE rank's time, not the maximum over ranks the protocol asks for: a
        # non-zero code tells the launcher. (A failed statistics gather alone leaves the measurement whole: exit 0.)
        sys.exit(3)


if __name__ == "__main__":
    main()

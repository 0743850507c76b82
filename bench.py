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
            "cpu_model": model, "host_threads_visible": len(os.sched_getaffinity(0)),
            "sample": "%d robots x %d steps of the same workload (oracle/umpc_oracle.c, fp32, OpenMP over robots)"
                      % (Bs, Ks),
            "single_thread_value": n1 * Ks / dt1, "fp64_single_thread_value": n1 * Ks / dt2}


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
        print(json.dumps({
            "metric": "closed-loop MPC steps/sec (QP+dyn)", "value": world * B * args.steps / elapsed, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: planar/mpc_osqp_p5f stroke-plane MPC, N=10 (n=87, m=164, "
                                   "nnz(L)=%d), %d ADMM it, 10 Ruiz, LDL' refactor per tick + Euler plant tick" % (s.nnzL, args.max_iter),
                       "robots_per_gpu": B, "global_batch": world * B, "horizon": 10,
                       "parallelism": "robots sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": alg * B / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg * B / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": {"tables": "bqp_solve_kernel", "wave": "bqp_wave_kernel"}.get(mpc.qp.kernel_name, "bqp_fixed_%s_kernel" % mpc.qp.kernel_name),
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-robots", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=100)
    args = ap.parse_args()

    if args.workload == "p5f":
        return main_p5f(args)
    import torch
    import torch.distributed as dist
    from robobee3d_amd.batch import BatchUprightMPC, hover_initial_conditions
    from robobee3d_amd import _lib

    from robobee3d_amd import shard
    rank, world, local_rank = shard.world()
    dev = torch.device("cuda", local_rank)
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        torch.cuda.set_device(local_rank)
        shard.init("nccl", device=dev)   # "nccl" is RCCL on ROCm
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    ndt = np.float32 if args.dtype == "f32" else np.float64
    plant_mode = 0 if args.plant == "euler" else 1
    B = args.batch

    # synthetic inputs, resident in HBM before the timed region
    lo, _hi = shard.robot_range(B, rank)   # weak scaling: B robots per GPU, RNG keyed by the global index
    st, ref = hover_initial_conditions(B, 20201118, ndt, index_offset=lo)
    mpc = BatchUprightMPC(B, tdt, device=dev, plant_mode=plant_mode, maxIter=args.max_iter, nsub=args.nsub)
    mpc.set_state(st, ref)
    if args.monte_carlo:  # Ib = Ib0 (1 + d), d ~ U(-0.2, 0.2)^3, thrust gain 1 + U(-0.2, 0.2)
        rng = np.random.default_rng(20201120 + rank)
        mpc.Ib = torch.as_tensor((np.array([3333.0, 3333.0, 1000.0])[:, None] *
                                  (1 + rng.uniform(-0.2, 0.2, (3, B)))).astype(ndt)).to(dev)
        mpc.gain = torch.as_tensor((1 + rng.uniform(-0.2, 0.2, B)).astype(ndt)).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    spl = args.steps if args.steps_per_launch <= 0 else max(1, min(args.steps_per_launch, args.steps))
    assert args.steps % spl == 0, "--steps must be a multiple of --steps-per-launch"
    # warm-up steps use the timed launch shape when they divide into it (so a rocprofv3 --stats average over
    # all dispatches of the step kernel is the timed launch's duration), one-step launches otherwise
    wspl = spl if args.warmup % spl == 0 else 1
    for _ in range(args.warmup // wspl):
        mpc.rollout(wspl)
    barrier()
    nlaunch = args.steps // spl
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nlaunch)]
    t0 = time.perf_counter()
    for k in range(nlaunch):
        evs[k][0].record()
        mpc.rollout(spl)        # ONE launch = spl closed-loop steps of all B robots
        evs[k][1].record()
    torch.cuda.synchronize(dev)
    local = time.perf_counter() - t0          # this rank's K steps, start barrier -> local completion
    barrier()
    elapsed = shard.max_over_ranks(local, device=dev)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    # end-of-run trajectory statistics: the only exchange of the path (RCCL all_gather over xGMI,
    # outside the timed region; 2 floats per robot)
    metric = shard.gather_stats(mpc.metrics(args.warmup + args.steps))
    status = mpc.status
    nbad = int((~torch.isfinite(mpc.state)).sum().item())

    if rank == 0:
        total_steps = world * B * args.steps
        value = total_steps / elapsed
        bps = ALG_BYTES_PER_STEP_FP32 * (2 if args.dtype == "f64" else 1)
        # instructions one wavefront issues per closed-loop step (fp32 kernel; counted in the ISA):
        # phase A 1.2k + 10 Ruiz passes x 0.94k + LDL'/hand-off 3.0k, 835 per ADMM iteration, phase C 5k, plant substeps
        ninstr = 13600 + 835 * args.max_iter + 5000 + args.nsub * (250 if plant_mode == 1 else 110)
        achieved = bps * B * spl / (kern_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                # PMC counters cannot be read live; this is the committed rocprofv3 --pmc measurement of THIS
                # command (profiles/r01_pmc_traffic.json), used only when the configuration matches it
                if (j.get("batch") == B and j.get("dtype") == args.dtype and j.get("plant") == args.plant
                        and j.get("steps_per_launch") == spl and args.max_iter == 50 and args.nsub == 25):
                    traffic = j["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        line = {
            "metric": "closed-loop MPC steps/sec (QP+dyn)", "value": value, "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: closed-loop uprightmpc2 (N=3, 50 ADMM it, 10 Ruiz, LDL' "
                                   "refactor per step) + 25 plant substeps, random-tilt hover, seed 20201118",
                       "robots_per_gpu": B, "global_batch": world * B, "horizon": 3,
                       "plant": "Euler+expm (reference step)" if plant_mode == 0 else "RK4 (build-defined)",
                       "parallelism": "robots sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": _lib.lib().umpcKernelName(0 if args.dtype == "f32" else 1, plant_mode).decode(),
                         "kernel_ms": kern_ms, "steps_per_launch": spl, "alg_bytes_per_launch": bps * B * spl,
                         # SURVEY 8d asks for all three rooflines; the one that binds is vector issue
                         "flops": {"achieved": 1.1e5 * B * spl / (kern_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                                   "peak": 157.3, "note": "~1.1e5 flop per robot-step (SURVEY 8d); MI355X fp32 vector peak"},
                         "lds": {"achieved": (args.max_iter * 78 * 16 + 2 * 640) * B * spl / (kern_ms * 1e-3) / 1e9 if args.dtype == "f32" else None,
                                 "unit": "GB/s", "peak": 256 * 128 * 2.4,
                                 "note": "78 ds_read_b128 per ADMM iteration per lane + hand-off; peak 128 B/clk/CU"},
                         "valu_issue": {"instr_per_wave_step": ninstr, "achieved": ninstr * (B / 64) * spl / (kern_ms * 1e-3) / 1e9,
                                        "peak": 1024 * 2.4 / 4, "unit": "G wave-instr/s",
                                        "note": "static instruction counts x loop trips (DESIGN.md 2); peak = 1024 SIMDs, one "
                                                "wave64 VALU op per 4 cycles at 2.4 GHz"} if args.dtype == "f32" else None,
                         "note": "path is VALU-issue bound, not HBM bound (DESIGN.md): ~1.1e5 flop per 1208 B"},
            "check": {"nonfinite_state_values": nbad,
                      "mean_pos_err_mm2": float(metric[0].mean().item()),
                      "status_solved_frac": float((status > 0).float().mean().item())},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, plant_mode)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

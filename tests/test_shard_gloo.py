"""N > 1 path on CPU: world_size-2 gloo job. Sharded run == single-process run (robots are
independent, RNG is keyed by the global robot index), the end-of-run gather returns the
statistics in global robot order, timing is the max over ranks."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo_shard_equals_single_process(tmp_path, oracle_built):
    per_rank, K = 48, 3
    out = str(tmp_path / "gathered.npz")
    from robobee3d_amd import shard
    port = str(shard.free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "tests", "_shard_worker.py"), out, str(per_rank), str(K)]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=300, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    g = np.load(out)
    from robobee3d_amd.batch import hover_initial_conditions
    st, ref = hover_initial_conditions(2 * per_rank, 20201118, np.float32)
    ctrl = np.zeros((127, 2 * per_rank), np.float32)
    ctrl[124:] = 1
    _, stats, _ = oracle_built.batch_rollout(st, ctrl, ref, K, dtype=np.float32, nthreads=2)
    np.testing.assert_array_equal(g["state"], st)                       # same trajectories ...
    np.testing.assert_array_equal(g["metric"], stats / (K * 25))        # ... same statistics, global order
    assert float(g["tmax"]) == 2.0                                      # max over ranks
    np.testing.assert_array_equal(g["uneven"], np.array([[0, 0, 0, 1, 1, 1, 1]], np.float32))


def test_split_ranges_cover_everything():
    from robobee3d_amd import shard
    for total, ws in ((1 << 20, 8), (1000, 3), (7, 8)):
        r = [shard.split_range(total, k, ws) for k in range(ws)]
        assert r[0][0] == 0 and r[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert shard.robot_range(65536, 3) == (196608, 262144)


def _bench(args, env=None, timeout=300):
    e = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, cwd=ROOT, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus N` (no torchrun around it, the driver's N = 1 command form) starts N ranks itself,
    before touching any device. --dry-run keeps the kernel out (no GPU here): launcher, gloo rendezvous, sharding,
    barriers, max-over-ranks and the statistics gather in global robot order all run; rank 0 prints ONE line."""
    import json
    r = _bench(["--gpus", "2", "--steps", "4", "--warmup", "3", "--batch", "96", "--dry-run", "--monte-carlo"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["steps"] == 4 and j["warmup"] == 3
    assert j["config"]["global_batch"] == 192 and j["scaling"] == "weak" and j["value"] is None
    # every rank's own time for the K steps (the line's ms_per_step is their maximum) and the process group's facts
    assert len(j["ms_per_step_per_rank"]) == 2 and all(t > 0 for t in j["ms_per_step_per_rank"])
    assert j["ms_per_step_per_rank"][1] > j["ms_per_step_per_rank"][0]      # the dry-run ranks sleep 1 + rank ms per launch
    assert j["collectives"] == {"backend": "gloo", "world_size": 2, "gathered_robots": 192}


def test_config5_shape_dry_run_on_eight_ranks():
    """BASELINE configs[4]'s exact shape -- 2^17 robots per rank x 8 ranks = 2^20, Monte-Carlo draws keyed
    by the global robot index -- through bench.py's own launcher on eight gloo ranks (CPU, no kernel): ONE line,
    world size 8, 2^20 robots gathered in global order (bench.py asserts the order on rank 0), one time per rank. This
    rehearses the rendezvous / sharding / gather of the 8-GPU run; it measures nothing and claims nothing about hardware."""
    import json
    r = _bench(["--gpus", "8", "--monte-carlo", "--batch", "131072", "--steps", "2", "--warmup", "1", "--dry-run"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["dry_run"] is True and j["value"] is None
    assert j["config"]["global_batch"] == 2 ** 20 and j["config"]["robots_per_gpu"] == 2 ** 17
    assert j["collectives"] == {"backend": "gloo", "world_size": 8, "gathered_robots": 2 ** 20}
    assert len(j["ms_per_step_per_rank"]) == 8
    assert j["ms_per_step"] is None          # dry run: nothing was measured (the per-rank entries are the ranks' sleeps)


def test_rendezvous_ports_are_never_fixed():
    """no literal rendezvous port left in the product, the bench or the tests (a collision risk on a shared node)"""
    import re
    for rel in ("robobee3d_amd/shard.py", "bench.py", "tests/test_gpu_parity.py", "tests/test_shard_gloo.py"):
        txt = open(os.path.join(ROOT, rel)).read()
        assert not re.search(r"[\"']295\d\d[\"']", txt), rel


def test_bench_launcher_propagates_failure():
    r = _bench(["--gpus", "2", "--steps", "5", "--steps-per-launch", "2", "--dry-run"])   # 5 % 2 != 0: every rank asserts
    assert r.returncode != 0


def test_monte_carlo_draws_are_partition_invariant():
    from robobee3d_amd.batch import monte_carlo_draws
    Ib, gain = monte_carlo_draws(300, 20201120, np.float64)
    Ib2, gain2 = monte_carlo_draws(100, 20201120, np.float64, index_offset=150)
    np.testing.assert_array_equal(Ib[:, 150:250], Ib2)
    np.testing.assert_array_equal(gain[150:250], gain2)
    assert np.all(np.abs(Ib / np.array([[3333.0], [3333.0], [1000.0]]) - 1) <= 0.2) and np.all(np.abs(gain - 1) <= 0.2)
    assert Ib.std(axis=1).min() > 50 and abs(gain.mean() - 1) < 0.02


def test_device_draws_equal_the_host_draws():
    """batch.monte_carlo_draws_device / hover_initial_conditions_device (what bench.py uses on the GPU: the draws are
    generated on the device) run the same counter hash in int64 arithmetic: bit-identical to the numpy
    versions (here on the CPU device; tests/test_r3_parity_evidence.py repeats it on the MI355X)."""
    import torch
    from robobee3d_amd import batch
    Ib, g = batch.monte_carlo_draws(5000, 20201120, np.float32, index_offset=(1 << 20) - 2500)
    Ibd, gd = batch.monte_carlo_draws_device(5000, 20201120, torch.float32, index_offset=(1 << 20) - 2500, device="cpu")
    assert np.array_equal(Ib, Ibd.numpy()) and np.array_equal(g, gd.numpy())
    st, ref = batch.hover_initial_conditions(3000, 20201118, np.float64, index_offset=77)
    std, refd, (a, b) = batch.hover_initial_conditions_device(3000, 20201118, torch.float64, index_offset=77, device="cpu")
    assert np.allclose(st, std.numpy(), rtol=0, atol=1e-15) and np.array_equal(ref, refd.numpy())
    assert np.abs(a.numpy()).max() <= 0.5 and np.abs(b.numpy()).max() <= 0.5


def test_failed_statistics_gather_does_not_cost_the_measurement():
    """the N > 1 path has never met two RCCL ranks, and the statistics gather sits between the timed
    region and the print. A gather that raises (injected on every rank, the way a broken collective surfaces in Python)
    must leave ONE line with the figure the ranks had already measured, the failure recorded under collectives.error, and
    a clean exit (the timing itself went through its own collectives)."""
    import json
    r = _bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "64", "--dry-run"], env={"UMPC_TEST_FAIL_GATHER": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and len(j["ms_per_step_per_rank"]) == 2
    assert j["dry_run_unmeasured"]["value"] > 0 and j["dry_run_unmeasured"]["ms_per_step"] > 0     # what `value` would carry
    assert j["collectives"]["world_size"] == 2 and "injected gather failure" in j["collectives"]["error"][0]
    assert j["collectives"]["gathered_robots"] == 64          # rank 0's own block only


def test_process_group_has_an_explicit_timeout():
    """shard.init passes timeout= to init_process_group (the backend default is half an hour of silence)"""
    import inspect
    from robobee3d_amd import shard
    src = inspect.getsource(shard.init)
    assert "timeout" in src and shard.INIT_TIMEOUT_S <= 600

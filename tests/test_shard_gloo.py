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
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533",
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

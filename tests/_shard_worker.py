"""Worker for tests/test_shard_gloo.py: one rank of a world_size-2 gloo job on CPU.
Each rank runs ITS block of robots through the CPU oracle (the checker standing in for
the GPU kernel, which cannot run here), then the ranks gather statistics exactly like
bench.py does."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from robobee3d_amd import shard  # noqa: E402
from robobee3d_amd.batch import hover_initial_conditions  # noqa: E402
import oraclebind  # noqa: E402

out_path, per_rank, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rank, ws, _ = shard.init("gloo")
lo, hi = shard.robot_range(per_rank, rank)
st, ref = hover_initial_conditions(per_rank, 20201118, np.float32, index_offset=lo)
ctrl = np.zeros((127, per_rank), np.float32)
ctrl[124:] = 1
out, stats, status = oraclebind.batch_rollout(st, ctrl, ref, K, dtype=np.float32, nthreads=2)
g = shard.gather_stats(torch.from_numpy(stats / (K * 25)))
gs = shard.gather_stats(torch.from_numpy(st))
tmax = shard.max_over_ranks(1.0 + rank)
uneven = shard.gather_stats(torch.full((1, 3 + rank), float(rank)))
if rank == 0:
    np.savez(out_path, metric=g.numpy(), state=gs.numpy(), tmax=tmax, uneven=uneven.numpy())
torch.distributed.barrier()
torch.distributed.destroy_process_group()

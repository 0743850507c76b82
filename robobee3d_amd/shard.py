"""Multi-GPU layout of the batched controller: robots are independent (no coupling
term anywhere on the path), so ranks own contiguous blocks of robots and there is
NO collective on the data path. The only exchange is the end-of-run gather of the
per-robot trajectory statistics. One process per GPU; backend "nccl"
(= RCCL over xGMI) on the MI355X node, "gloo" in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def world():
    """(rank, world_size, local_rank) from the torchrun environment (1 process: 0, 1, 0)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def free_port():
    """A TCP port that is free on 127.0.0.1 right now (rendezvous ports are never fixed numbers: a shared node may run
    several jobs)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


INIT_TIMEOUT_S = float(os.environ.get("UMPC_DIST_TIMEOUT_S", "300"))


def init(backend=None, device=None, force=False, timeout_s=None):
    """force: initialise the process group even for world_size 1 (rehearsal of the RCCL path on a one-GPU box).
    timeout_s: rendezvous AND collective timeout of the group (default INIT_TIMEOUT_S = 300 s, UMPC_DIST_TIMEOUT_S in the
    environment): a rank that never arrives fails the job in minutes with an error instead of hanging it for the backend's
    default half hour."""
    rank, ws, local_rank = world()
    if (ws > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # a launcher (torchrun, bench.py's self-launch) always provides the port; only a lone process that forces a
            # group (world_size 1) gets here, and it may pick any free one
            if ws > 1:
                raise RuntimeError("MASTER_PORT is not set: start the ranks with torch.distributed.run (or `python bench.py "
                                   "--gpus N`, which picks a free port itself)")
            os.environ["MASTER_PORT"] = str(free_port())
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        import datetime
        kw = {"timeout": datetime.timedelta(seconds=float(timeout_s if timeout_s is not None else INIT_TIMEOUT_S))}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=ws, **kw)
    return rank, ws, local_rank


def robot_range(per_rank, rank):
    """Weak scaling: every rank owns `per_rank` robots; global ids [rank*per_rank, (rank+1)*per_rank)."""
    return rank * per_rank, (rank + 1) * per_rank


def split_range(total, rank, ws):
    """Strong scaling: `total` robots split into ws contiguous blocks (first blocks one larger)."""
    base, rem = divmod(total, ws)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_scalars(value, device="cpu"):
    """One float per rank -> the list over ranks, in rank order (every rank gets it)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if not dist.is_initialized():
        return [float(t.item())]
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def gather_stats(stats):
    """stats [rows, B_local] on every rank -> [rows, sum B_local] in global robot order (all ranks)."""
    if not dist.is_initialized():
        return stats
    if os.environ.get("UMPC_TEST_FAIL_GATHER") == "1":
        # TEST ONLY (tests/test_shard_gloo.py): every rank fails here the way a broken RCCL gather would surface in
        # Python, so that the CPU suite can check that bench.py still prints the measurement it has already taken
        raise RuntimeError("injected gather failure (UMPC_TEST_FAIL_GATHER=1)")
    ws = dist.get_world_size()
    n = torch.tensor([stats.shape[1]], dtype=torch.int64, device=stats.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((stats.shape[0], m), dtype=stats.dtype, device=stats.device)
    pad[:, :stats.shape[1]] = stats
    parts = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad.contiguous())
    return torch.cat([p[:, :s] for p, s in zip(parts, sizes)], dim=1)

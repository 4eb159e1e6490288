"""Multi-GPU sharding: one process per GPU, independent env shards, no data-path
collective.  The only exchange is an all-gather of the fixed-size rollout-metrics
vector at the end of a rollout (RCCL over xGMI when the backend is "nccl"; the same
code runs over gloo on CPU tensors in the tests).

The reference has no distributed code at all; this is new.
"""
import os
from typing import Dict, Tuple

import torch

METRIC_NAMES = ["env_steps", "episodes", "successes", "reward_sum", "completed_subtasks_sum",
                "errors", "reserved6", "reserved7"]


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous env-index range [start, start+count) of `rank`; sizes differ by <= 1."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(int(total_envs), world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def rank_seed(seed: int, rank: int) -> int:
    """Independent action stream per shard."""
    return int(seed) + 1000003 * int(rank)


def gather_rollout_metrics(local: torch.Tensor, elapsed_s: float, group=None) -> Dict:
    """all_gather the int64[8] metrics vector of every rank and MAX-reduce the elapsed
    time.  Returns {'per_rank': [[...]], 'total': {name: sum}, 'elapsed_s': max}."""
    import torch.distributed as dist
    if local.dtype != torch.int64 or local.numel() != 8:
        raise ValueError("metrics must be int64[8]")
    if not (dist.is_available() and dist.is_initialized()):
        per_rank = [local.cpu().tolist()]
        el = float(elapsed_s)
    else:
        world = dist.get_world_size(group)
        bufs = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(bufs, local.contiguous(), group=group)
        per_rank = [b.cpu().tolist() for b in bufs]
        t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=local.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        el = float(t.item())
    total = {name: sum(r[i] for r in per_rank) for i, name in enumerate(METRIC_NAMES)}
    return {"per_rank": per_rank, "total": total, "elapsed_s": el}


def reduce_max(values, device=None, group=None):
    """MAX over ranks of a short list of floats (the values themselves when no process group
    is up).  `device`: where the exchanged tensor lives (a CUDA device for backend nccl/RCCL,
    None = host for gloo)."""
    import torch.distributed as dist
    vals = [float(v) for v in values]
    if not (dist.is_available() and dist.is_initialized()):
        return vals
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.cpu().tolist()


def whole_job_rate(total_env_steps: int, elapsed_s: float) -> float:
    return float(total_env_steps) / float(elapsed_s)

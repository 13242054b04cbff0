"""Multi-GPU layout of the hot path: envs are independent worlds, so a batch is cut into
contiguous shards, one handle (one process, one GPU) per shard, with no data-path collective.  The only exchange
is the end-of-step metrics all-gather: RCCL over xGMI from the C shim (``Env.metrics_allgather``) on GPUs, or any
object exposing ``all_gather(np.ndarray) -> list[np.ndarray]`` (the CPU tests use torch.distributed / gloo)."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from . import capi


def shard_range(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous env range [start, start + count) owned by ``rank``: GPU g owns envs [g*N/G, (g+1)*N/G)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    start = rank * total_envs // world
    end = (rank + 1) * total_envs // world
    return start, end - start


def make_shard(lib: capi.CLib, track, total_envs: int, rank: int, world: int, **env_kwargs) -> capi.Env:
    """The handle for this rank's slice of a ``total_envs`` batch (same seed on every rank: identity comes from env_base)."""
    start, count = shard_range(total_envs, rank, world)
    return capi.Env(lib, track, n_envs=count, env_base=start, **env_kwargs)


def reduce_metrics(records: np.ndarray) -> Dict[str, float]:
    """[world][FTGP_METRIC_DOUBLES] per-rank records -> whole-job totals."""
    r = np.asarray(records, dtype=np.float64).reshape(-1, capi.METRIC_DOUBLES)
    out = {k: float(r[:, i].sum()) for i, k in enumerate(capi.METRIC_FIELDS[:6])}
    out["min_lap_time"] = float(r[:, 6].min())
    out["max_lap_time"] = float(r[:, 7].max())
    out["ranks"] = int(r.shape[0])
    return out


class GlooGather:
    """all_gather of small float64 records over an initialised torch.distributed (gloo) group -- CPU tests only."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist

    def all_gather(self, rec: np.ndarray) -> np.ndarray:
        t = self.torch.from_numpy(np.ascontiguousarray(rec, dtype=np.float64))
        outs = [self.torch.empty_like(t) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(outs, t)
        return np.stack([o.numpy() for o in outs])


def gather_metrics(env: capi.Env, comm: Optional[object] = None) -> np.ndarray:
    """Per-rank metrics records of the whole job.  comm=None: the handle's own RCCL communicator (or world 1)."""
    if comm is None:
        return env.metrics_allgather()
    return comm.all_gather(env.metrics_local())

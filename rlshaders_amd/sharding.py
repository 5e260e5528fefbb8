"""Multi-GPU sharding of a shading-point batch: one process per GPU, contiguous index ranges, no
collective on the data path.

Every shading point is independent (the reference relies on the same fact: Arnold's render threads
share nothing), so a batch of ``total`` points splits into ``world`` ranges
``[g*total/world, (g+1)*total/world)``; each rank generates / receives and processes its own range
on its own GPU.  The only communication is control-plane: a barrier around the timed region, the
max-over-ranks of the elapsed time, and (validation only) a gather of per-shard 64-bit checksums.
torch.distributed supplies it -- backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import datetime
import os
from typing import List, Optional, Tuple

import torch


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first index, count) of rank's contiguous shard; the shards tile [0, total) exactly."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    return lo, hi - lo


class Ranks:
    """Thin wrapper over torch.distributed that degrades to a no-op for a single process."""

    DEFAULT_TIMEOUT_S = 120.0        # rendezvous + every collective (RLS_DIST_TIMEOUT_S overrides)

    def __init__(self, backend: Optional[str] = None, device: Optional[torch.device] = None, launched: bool = False,
                 timeout_s: Optional[float] = None):
        """``launched``: opt in to initialising the process group for a ONE-rank run as well (bench.py and the tests pass it
        when a launcher started them: the control path of an N-rank run -- init_process_group("nccl", device_id), barrier,
        device all_reduce -- is then the one a single-GPU box exercises too).  Without it a single process never touches
        torch.distributed, whatever RANK / MASTER_PORT variables its environment happens to export.  A default process
        group the caller has already created is reused, never re-initialised, and left alone by close()."""
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device if device is not None else torch.device("cpu")
        self.dist = None
        self._owns_group = False
        self.timeout_s = float(timeout_s if timeout_s is not None else os.environ.get("RLS_DIST_TIMEOUT_S", self.DEFAULT_TIMEOUT_S))
        if self.world > 1 or launched:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.world, self.rank = dist.get_world_size(), dist.get_rank()
            else:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                if backend is None:
                    backend = "nccl" if self.device.type == "cuda" else "gloo"
                kw = {"device_id": self.device} if backend == "nccl" else {}
                # A rank that never arrives (a dead GPU, a child that failed before this line) must not hold the others for
                # torch's default 10-30 minutes: the rendezvous, and every collective after it, gives up after `timeout_s`.
                dist.init_process_group(backend, timeout=datetime.timedelta(seconds=self.timeout_s), **kw)
                self._owns_group = True
            self.dist = dist

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, values: List[float]) -> List[float]:
        if self.dist is None:
            return list(values)
        t = torch.tensor(values, dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def gather_u64(self, value: int) -> List[int]:
        """every rank's 64-bit value, in rank order (validation: checksum of checksums)"""
        if self.dist is None:
            return [int(value)]
        # two 32-bit halves in int64 slots: no signed-overflow games
        mine = torch.tensor([value & 0xFFFFFFFF, (value >> 32) & 0xFFFFFFFF], dtype=torch.int64, device=self.device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [int(o[0]) | (int(o[1]) << 32) for o in out]

    def gather_objects(self, obj) -> list:
        """every rank's (picklable) object, in rank order -- device identities, per-rank timings"""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self) -> None:
        if self.dist is not None:
            self.dist.barrier()
            if self._owns_group:
                self.dist.destroy_process_group()
            self.dist = None

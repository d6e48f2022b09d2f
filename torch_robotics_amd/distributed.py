"""Multi-GPU use of the hot path: one process per GPU, batch sharding, one tiny collective.

Every (batch, horizon) sample is independent for FK and for the collision / EE objectives, and the terms that
couple time steps stay inside one trajectory, so the batch dimension is split in contiguous blocks of whole
trajectories.  Models, cost tables and the SDF grid are replicated.  The only exchange
step is the sum of the per-rank cost scalars (and, if a caller wants them, other packed partial sums): one small
all-reduce -- RCCL over xGMI on the GPU box (`backend="nccl"`), gloo in the CPU tests.  Per-sample outputs
(cost, gradient, link positions) stay sharded.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_batch(n_trajectories: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) block of trajectories owned by `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(int(n_trajectories), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_reduce_sum_(packed: torch.Tensor, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
    """In-place sum of a small packed buffer of partial sums over all ranks (no-op without a process group)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class CostSumReducer:
    """Folds the fused kernel's per-wavefront cost sums into one scalar per evaluation and all-reduces it on a
    side stream, so the launch stream never waits for the collective."""

    def __init__(self, device, n_slots: int = 64):
        from . import ops
        self._ops = ops
        self.device = torch.device(device)
        self.slots = torch.zeros(n_slots, device=self.device, dtype=torch.float32)
        self.side = torch.cuda.Stream(self.device)
        self._k = 0

    def submit(self, block_sums: torch.Tensor) -> torch.Tensor:
        """Returns a 1-element view that holds the global cost sum once the side stream has run."""
        k = self._k % self.slots.numel()
        self._k += 1
        out = self.slots[k:k + 1]
        self._ops.reduce_sum(block_sums, out=out)              # deterministic, on the current stream
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            all_reduce_sum_(out)
        return out

    def wait(self):
        torch.cuda.current_stream(self.device).wait_stream(self.side)

"""Point-chunk sharded MSM over the ranks of a torch.distributed group (north_star: "MSM shards by
point-chunk across the 8 GPUs of one node with a final RCCL all-reduce of partial Jacobian sums over xGMI").

One process per GPU.  Rank r keeps bases [lo_r, hi_r) of the SRS resident in its own HBM (a `backend.Srs`) and is handed the
matching chunk of every scalar vector; its partial sum is one `uzk_msm_g1_device` call.  Group addition is not a collective's
reduce operator, so the "all-reduce" is what it reduces to for a 96-byte operand: an all-gather of the N partial sums (RCCL over
xGMI with backend "nccl", gloo on CPU for the tests) and the same fold on every rank through the C ABI's host-side
`uzk_g1_fold` -- N - 1 group additions, identical bytes everywhere.  There is no other data-path exchange: bases and scalars
never move between ranks.

`bench.py --gpus N` and tests/test_distributed_fold.py both run THIS class; torch is imported lazily so that the package
itself does not depend on it."""
from __future__ import annotations

import time

import numpy as np

from . import backend as b


def chunk_bounds(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous point chunk of rank `rank`: [lo, hi).  Chunks differ by at most one point."""
    return rank * n_total // world, (rank + 1) * n_total // world


class ShardedCommitter:
    """`srs`: this rank's chunk of the bases (backend.Srs, device resident; None = exchange only).  `group`: a torch.distributed process group
    (None = the default group; not initialised or world size 1 = no exchange).  `device`: the torch device of this rank's GPU
    (needed by the "nccl" backend, whose collectives take device tensors)."""

    def __init__(self, srs, group=None, device=None, force_collective: bool = False):
        self.srs = srs
        self.group = group
        self.world, self.rank, self.backend = 1, 0, None
        self._dist = None
        try:
            import torch.distributed as dist  # type: ignore
            if dist.is_available() and dist.is_initialized():
                self._dist = dist
                self.world = dist.get_world_size(group)
                self.rank = dist.get_rank(group)
                self.backend = dist.get_backend(group)
        except ImportError:
            pass
        self.exchanging = self._dist is not None and (self.world > 1 or force_collective)
        self.msm_s = 0.0          # wall clock spent in this rank's MSMs / in the exchange (incl. waiting for the slowest rank)
        self.exchange_s = 0.0
        if self.exchanging:
            import torch  # type: ignore
            if self.backend == "nccl" and device is None:
                raise ValueError("ShardedCommitter: the nccl backend needs this rank's torch device")
            coll = device if self.backend == "nccl" else torch.device("cpu")
            self._send = torch.zeros(96, dtype=torch.uint8, device=coll)
            self._recv = torch.zeros(96 * self.world, dtype=torch.uint8, device=coll)
            self._torch = torch

    def exchange(self, partial: np.ndarray) -> np.ndarray:
        """This rank's Jacobian partial sum (12 x u64, wire format) -> the sum over all ranks, the same bytes on every rank."""
        part = np.ascontiguousarray(partial, dtype=np.uint64).reshape(12)
        if not self.exchanging:
            return part
        t0 = time.perf_counter()
        self._send.copy_(self._torch.from_numpy(part.view(np.uint8)))
        self._dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
        allp = self._recv.cpu().numpy().view(np.uint64).reshape(self.world, 12)
        total = b.g1_fold(allp)
        self.exchange_s += time.perf_counter() - t0
        return total

    def commit_device(self, d_scalars: int, n_local: int) -> np.ndarray:
        """MSM of this rank's scalar chunk (device pointer, `n_local` elements matching the rank's bases), then the exchange."""
        if self.srs is None:
            raise ValueError("ShardedCommitter: made without bases (exchange-only); commit_device needs this rank's Srs")
        t0 = time.perf_counter()
        part = b.msm_device(self.srs, d_scalars, n_local)
        self.msm_s += time.perf_counter() - t0
        return self.exchange(part)

    def reset_clocks(self) -> None:
        self.msm_s = self.exchange_s = 0.0

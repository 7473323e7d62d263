"""Multi-GPU path on CPU: world_size-2 / -3 gloo rendezvous of the product's sharded commit
(uzkge_amd/sharded.py, the class bench.py --gpus N runs: all_gather of each rank's 96-byte Jacobian partial,
then a local fold through the C ABI's host-side `uzk_g1_fold`).  The partials come from the oracle here (no GPU); on the GPU box the
same code path runs with RCCL and partials from `uzk_msm_g1_device`."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bn254_py as opy
import oracle_c as oc
from util import load_srs, rand_fr_wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uzkge_amd import backend as b
    wire, _ = load_srs("lagrange-srs-4096.bin")
    scal = rand_fr_wire(n, 4242)
    lo, hi = rank * n // world, (rank + 1) * n // world          # contiguous point chunk per rank
    part = oc.msm_pippenger(wire[lo:hi], scal[lo:hi], 0, 1)      # stand-in for the rank's GPU partial
    from uzkge_amd.sharded import ShardedCommitter, chunk_bounds
    assert (lo, hi) == chunk_bounds(n, rank, world)
    sc = ShardedCommitter(None)                                    # product code: all-gather of the partials + C ABI host fold
    assert (sc.world, sc.rank, sc.backend, sc.exchanging) == (world, rank, "gloo", True)
    folded = sc.exchange(part)
    assert sc.exchange_s > 0.0
    q.put((rank, oc.jac_to_affine_ints(folded)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_and_fold_matches_single_msm(world):
    n = 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    wire, _ = load_srs("lagrange-srs-4096.bin")
    want = oc.jac_to_affine_ints(oc.msm_pippenger(wire[:n], rand_fr_wire(n, 4242), 0, 2))
    assert want is not None
    for _, got in results:
        assert got == want       # every rank folds to the same commitment

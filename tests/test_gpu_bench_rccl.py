"""bench.py's exchange path over RCCL on real hardware, as far as one GPU allows: a process group of one rank
(UZK_BENCH_FORCE_DIST=1) initialises RCCL on the device and runs the 96-byte all-gather, the max-over-ranks
all-reduce and the barriers that the driver's N > 1 runs use; the folded result must be the rank's own partial."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_runs_its_collective_over_rccl_with_one_rank():
    env = dict(os.environ, UZK_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "16", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["dist_world_size"] == 1 and line["collective_backend"] == "nccl"
    assert line["n_gpus"] == 1 and line["value"] > 0

    env.pop("UZK_BENCH_FORCE_DIST")
    r2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    plain = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    assert plain["collective_backend"] is None
    assert plain["result_affine_sha256"] == line["result_affine_sha256"]      # all-gather + fold of one partial = that partial

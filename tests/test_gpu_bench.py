"""GPU: bench.py's multi-rank machinery on a one-GPU box -- the RCCL call path with a single rank
(--force-collective), and two self-launched ranks sharing the GPU over gloo."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--workload", "uniform10m", "--no-cpu-baseline", "--generate", "device"]


def _bench(extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + extra, capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_rccl_path_runs_with_one_rank():
    """Communicator set-up, the int64 all_reduce of the count, barrier, all_gather of the rank rows and teardown,
    through backend "nccl" (= RCCL) with world size 1."""
    plain = _bench([])
    one = _bench(["--force-collective"])
    assert one["config"]["backend"] == "nccl" and one["n_gpus"] == 1
    assert one["all_reduce_ms"] is not None and 0.0 < one["all_reduce_ms"] < 50.0
    assert one["pairs"] == plain["pairs"] > 0 and one["tests"] == plain["tests"]
    assert plain["all_reduce_ms"] is None


def test_two_self_launched_ranks_share_the_gpu():
    plain = _bench([])
    two = _bench(["--gpus", "2", "--backend", "gloo"])
    assert two["n_gpus"] == 2 and len(two["ranks"]) == 2
    assert two["pairs"] == plain["pairs"] and two["tests"] == plain["tests"]
    assert [r["x_lo"] for r in two["ranks"]] == [0, 200] and two["all_reduce_ms"] is not None

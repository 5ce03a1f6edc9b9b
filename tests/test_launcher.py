"""CPU: the parent process of a self-launched multi-rank bench (bench.py --gpus N without a rank environment).
It must not outlive a rank that dies at start-up, and it must hand on that rank's status and stderr."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_ranks_stops_the_others_when_a_rank_dies(capfd):
    import bench
    child = ("import os, sys, time\n"
             "if os.environ['RANK'] == '1':\n"
             "    sys.exit('rank 1: no such device')\n"
             "time.sleep(300)\n")                      # the survivors: stuck as if in a barrier
    t0 = time.time()
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(3, cmd=[sys.executable, "-c", child], grace_s=2.0)
    assert e.value.code == 1
    assert time.time() - t0 < 30
    err = capfd.readouterr().err
    assert "rank 1 exited with status 1" in err and "no such device" in err


def test_spawn_ranks_relays_rank_zero_and_every_status(capfd):
    import bench
    child = ("import os, sys\n"
             "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
             "print('line of rank', os.environ['RANK'], 'local', os.environ['LOCAL_RANK'])\n")
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(2, cmd=[sys.executable, "-c", child])
    assert e.value.code == 0
    assert capfd.readouterr().out == "line of rank 0 local 0\n"
    # a rank that fails only at the very end still fails the run
    late = "import os, sys\nsys.exit(7 if os.environ['RANK'] == '1' else 0)\n"
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(2, cmd=[sys.executable, "-c", late])
    assert e.value.code == 7


def test_spawn_ranks_gives_up_after_its_time_limit(capfd):
    import bench
    t0 = time.time()
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(2, cmd=[sys.executable, "-c", "import time; time.sleep(300)"], timeout_s=1.0, grace_s=1.0)
    assert e.value.code == 1 and time.time() - t0 < 20
    assert "no result within" in capfd.readouterr().err


def test_bench_command_line_with_a_failing_rank():
    """The real command line: rank 1 is told to fail before it touches anything; the run ends in seconds, not
    at a collective's timeout, with a non-zero status."""
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0", "--workload", "uniform10m", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, PAIRCOUNT_TEST_FAIL_RANK="1"))
    assert r.returncode != 0 and r.stdout == ""
    assert "the other ranks were stopped" in r.stderr
    assert time.time() - t0 < 120

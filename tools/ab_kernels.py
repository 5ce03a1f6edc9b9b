#!/usr/bin/env python3
"""Per-kernel times of the whole-path call for several builds of the library (A/B measurements).

    python3 tools/ab_kernels.py [--workload uniform100m] [--filter pc_c] lib1.so lib2.so ...

Each library runs in a child process of its own (PAIRCOUNT_LIB); prints the average launch duration of the
kernels whose name contains the filter, and the step time."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="uniform100m")
ap.add_argument("--filter", default="pc_c")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--extra", default="", help="more bench.py arguments")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
for lib in a.libs:
    env = dict(os.environ, PAIRCOUNT_LIB=os.path.abspath(lib)) if lib != "default" else dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--generate", "device", "--steps", str(a.steps),
                        "--event-period", "1", "--workload", a.workload] + a.extra.split(), env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        print(lib, "FAILED", r.stderr[-300:], flush=True)
        continue
    d = json.loads(r.stdout.strip().splitlines()[-1])
    ks = {k: round(v["avg_ms"], 4) for k, v in d["kernels"].items() if a.filter in k and v["avg_ms"] > 0.008}
    print(lib, "step %.3f bin %.3f pair %.3f pairs %d" % (d["ms_per_step"], d["binning_ms"], d["pair_test_ms"], d["pairs"]), ks, flush=True)

#!/usr/bin/env python3
"""Step times of several builds of the library, each run several times in turn (A/B of whole steps on one box).

    python3 tools/ab_steps.py [--rounds 3] [--extra "--workload uniform10m"] default exp/lib_x.so ...

Prints ms per step of every run and the median per library; `default` is the in-tree build."""
import argparse, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--extra", default="", help="more bench.py arguments")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
seen = {lib: [] for lib in a.libs}
for r in range(a.rounds):
    for lib in a.libs:
        env = dict(os.environ, PAIRCOUNT_LIB=os.path.abspath(lib)) if lib != "default" else dict(os.environ)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-e2e-full", "--generate", "device",
                            "--steps", str(a.steps), "--event-period", "1000000"] + a.extra.split(), env=env, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            print(lib, "FAILED", p.stderr[-300:], flush=True)
            sys.exit(1)
        d = json.loads(p.stdout.strip().splitlines()[-1])
        seen[lib].append(d["ms_per_step"])
        print("round %d %-24s %.4f ms/step  pairs %d" % (r, lib, d["ms_per_step"], d["pairs"]), flush=True)
for lib, v in seen.items():
    print("%-24s median %.4f  min %.4f  (%s)" % (lib, statistics.median(v), min(v), a.extra))

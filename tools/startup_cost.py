#!/usr/bin/env python3
"""Wall time of the host program outside its own phase timers: dynamic loading before main() and teardown after the count is printed."""
import importlib
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
ht = importlib.import_module("particle_simulation_amd.hosttools")


def wall(cmd, env=None, reps=5):
    best = 1e9
    for _ in range(reps):
        t = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        best = min(best, time.time() - t)
    return best, r


print("true: %.1f ms" % (wall(["/bin/true"])[0] * 1e3))
print("main -h (dynamic loading only): %.1f ms" % (wall([ht.MAIN_PATH, "-h"])[0] * 1e3))
path = "/tmp/u1m.xyz"
ht.generate_text(path, ht.UNIFORM, 7, 1_000_000)
w, r = wall([ht.MAIN_PATH, "-f", path, "-p", "16"], env=dict(os.environ, PAIRCOUNT_TIMER="1"))
ph = {m.group(1): int(m.group(2)) / 1e3 for m in re.finditer(r"(\w[\w()]*):\s+(\d+) us", r.stderr)}
inside = ph["init"] + ph["parsing"] + ph["finish(wall)"]
print("1 M points: wall %.1f ms, phases %s, inside timers %.1f ms, outside %.1f ms" % (w * 1e3, ph, inside, w * 1e3 - inside))
os.remove(path)

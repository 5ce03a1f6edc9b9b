#!/usr/bin/env python3
"""Phase times of the C host program on a generated 10 M-point file, for several
parser thread counts, alone and while this process holds a HIP context."""
import importlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ht, path, p):
    env = dict(os.environ, PAIRCOUNT_TIMER="1")
    t = time.time()
    r = subprocess.run([ht.MAIN_PATH, "-f", path, "-p", str(p)], capture_output=True, text=True, env=env)
    print(p, "wall %.3f" % (time.time() - t), r.stdout.strip(), " ".join(r.stderr.split()), flush=True)


def main():
    from __graft_entry__ import load_package
    load_package()
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    path = "/tmp/u10m.xyz"
    ht.generate_text(path, ht.UNIFORM, 7, 10_000_000)
    print("file", os.path.getsize(path))
    for p in (1, 4, 16):
        run(ht, path, p)
    import torch
    x = torch.zeros(1 << 28, device="cuda:0")
    torch.cuda.synchronize()
    print("with a HIP context and 1 GiB held by the parent:")
    for p in (16, 16):
        run(ht, path, p)
    os.remove(path)


if __name__ == "__main__":
    main()

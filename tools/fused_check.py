#!/usr/bin/env python3
"""Builder's check of the fused pair test (count variant 7) against variants 5 and 0 on a few inputs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from __graft_entry__ import load_package
psa = load_package()

def run(st, t, lo, hi, v):
    psa.set_variant("count", v)
    st.count_pairs(t, lo, hi)
    return st.pairs()

rng = np.random.default_rng(5)
cases = []
# uniform slab of 6 layers + halo at ~1.5 points per cell
def slab(n, lo, hi, seed):
    r = np.random.default_rng(seed)
    p = r.uniform(-9.9999, 9.9999, (n, 3))
    p[:, 0] = r.uniform(lo * 0.05 - 10.0 + 1e-5, (min(hi, 399) + 1) * 0.05 - 10.0 - 1e-5, n)
    return p.astype(np.float32)
cases.append(("slab6 1.5/cell", slab(int(7 * 160000 * 1.5), 100, 106, 1), 100, 106))
cases.append(("slab2 1.9/cell", slab(int(2 * 160000 * 1.9), 398, 400, 2), 398, 400))
cases.append(("slab3 at 0", slab(int(4 * 160000 * 1.0), 0, 3, 3), 0, 3))
cases.append(("sparse whole", rng.uniform(-9.9999, 9.9999, (1_000_000, 3)).astype(np.float32), 0, 400))
b = rng.normal(0, 0.2, (300_000, 3)).astype(np.float32)
cases.append(("blob (classic takes over)", b, 0, 400))
st = psa.DeviceStages(2_200_000, "cuda:0")
ok = True
for name, pts, lo, hi in cases:
    t = torch.from_numpy(pts).to("cuda:0")
    got = {v: run(st, t, lo, hi, v) for v in (5, 7, 0, 4)}
    st.count_tests(lo, hi)
    print(name, len(pts), got, "tests", st.tests(), flush=True)
    ok &= len(set(got.values())) == 1
# context path: pc_finish through the fused kernel, tests from the kernel
psa.set_variant("count", 7)
pts = cases[0][1]
pc = psa.PairCounter(0, len(pts), slab=(100, 106))
pc.append(pts)
print("finish", pc.finish(), "tests", pc.tests(), pc.phase_ms(), flush=True)
pc.close()
psa.set_variant("count", 4)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

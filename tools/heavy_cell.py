#!/usr/bin/env python3
"""Pair-test time when many points share ONE cell (the case no work item can split): n points uniform inside a single cell."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
psa = load_package()
for n in (20_000, 50_000, 100_000, 200_000):
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    xyz = (torch.rand((n, 3), device="cuda:0", generator=g) * 0.049 + 0.0005).contiguous()   # cell (200,200,200)
    st = psa.DeviceStages(n, "cuda:0")
    for variant in (3, 0):
        psa.set_variant("count", variant)
        st.count_pairs(xyz); torch.cuda.synchronize()
        psa.profile_enable(True)
        t = time.time(); st.count_pairs(xyz); torch.cuda.synchronize(); dt = time.time() - t
        psa.profile_enable(False)
        st.count_tests()
        print("n %7d variant %d: %.1f ms, pairs %d, tests %d (%.2e tests/s)" % (n, variant, dt * 1e3, st.pairs(), st.tests(), st.tests() / dt), flush=True)
        print("      " + ", ".join("%s %.3f" % (k, v[2]) for k, v in psa.profile_read().items() if v[2] > 0.02), flush=True)
psa.set_variant("count", 4)

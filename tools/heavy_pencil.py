#!/usr/bin/env python3
"""Binning time when all points share ONE pencil (a single workgroup sorts it) and pair-test time of a pencil of over-full cells."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
psa = load_package()
for n in (1_000_000, 5_000_000, 20_000_000):
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    xyz = torch.rand((n, 3), device="cuda:0", generator=g)
    xyz[:, 0] = xyz[:, 0] * 0.049 + 0.0005
    xyz[:, 1] = xyz[:, 1] * 0.049 + 0.0005
    xyz[:, 2] = xyz[:, 2] * 19.99 - 9.995
    xyz = xyz.contiguous()
    st = psa.DeviceStages(n, "cuda:0")
    st.bin(xyz); torch.cuda.synchronize()
    t = time.time(); st.bin(xyz); torch.cuda.synchronize(); dt = time.time() - t
    print("n %9d in one pencil: binning %.2f ms" % (n, dt * 1e3), flush=True)
    if n <= 5_000_000:
        for variant in (3, 0):
            psa.set_variant("count", variant)
            t = time.time(); st.count_sorted(n); torch.cuda.synchronize(); dt = time.time() - t
            st.count_tests()
            print("    variant %d: pair test %.1f ms, pairs %d, tests %d (%.2e tests/s)" % (variant, dt * 1e3, st.pairs(), st.tests(), st.tests() / dt), flush=True)
        psa.set_variant("count", 4)

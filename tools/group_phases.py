#!/usr/bin/env python3
"""Where the waves of the group kernel (pc_c3_count_groups) spend their time: runs the whole-path call on a workload
with a build of the library made with -DPC_G_TRACE (per-part shader-clock sums, see pc_count_group.h) and prints the
share of each part and the cycles per window of its kind.

    tools/build_variant.sh gtrace -DPC_G_TRACE
    python3 tools/group_phases.py exp/lib_gtrace.so [--workload blobs100m] [--sigma-micro S] [--count-variant 3]
"""
import argparse
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PARTS = ["table reads + window bookkeeping", "copy into LDS (list, DMA, ranges)", "steps, 2-lane groups", "steps, 4-lane groups",
         "steps, 8-lane groups", "dense window: register blocks", "(unused)", "cell that does not fit: set-up",
         "  the cell, piece by piece (pc_g_cell)"]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--workload", default="blobs100m")
    ap.add_argument("--sigma-micro", type=int, default=None)
    ap.add_argument("--count-variant", type=int, default=None)
    ap.add_argument("--as-rank", type=int, nargs=2, default=None)
    args = ap.parse_args()
    os.environ["PAIRCOUNT_LIB"] = os.path.abspath(args.lib)
    import torch
    from __graft_entry__ import load_package
    import bench
    psa = load_package()
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    sh = importlib.import_module("particle_simulation_amd.sharding")
    kind_name, points, sigma, scaling = bench.WORKLOADS[args.workload]
    if args.sigma_micro is not None:
        sigma = args.sigma_micro
    kind = ht.UNIFORM if kind_name == "uniform" else ht.BLOBS
    x_lo, x_hi = 0, 400
    if args.as_rank:
        r, w = args.as_rank
        if scaling == "weak":
            points *= w
        cuts = sh.equal_cuts(w)
        x_lo, x_hi = cuts[r], cuts[r + 1]
        xyz = psa.generate_device(kind, 7, points, sigma, device="cuda:0", x_lo=x_lo, x_hi=min(x_hi, 399))
    else:
        xyz = psa.generate_device(kind, 7, points, sigma, device="cuda:0")
    if args.count_variant is not None:
        psa.set_variant("count", args.count_variant)
    st = psa.DeviceStages(xyz.shape[0], "cuda:0")
    st.count_pairs(xyz, x_lo, x_hi)
    out = (ctypes.c_ulonglong * 32)()
    raw = ctypes.CDLL(os.path.abspath(args.lib))
    assert raw.pc_group_trace_read(out) == 0          # (clears the counters of the warm-up)
    reps = 3
    psa.profile_enable(1)
    for _ in range(reps):
        st.count_pairs(xyz, x_lo, x_hi)
    assert raw.pc_group_trace_read(out) == 0
    prof = psa.profile_read()
    psa.profile_enable(False)
    for k, (timed, launches, ms) in prof.items():
        if k.startswith("pc_c3_count") and timed:
            print("%s: %.3f ms per launch in this build" % (k, ms / timed))
    print("pairs %d" % st.pairs())
    v = [int(x) for x in out]
    total = sum(v[:16])
    for i, name in enumerate(PARTS):
        print("   %-40s %5.1f %% of the waves' cycles  %10d times per run  %8.0f cycles each"
              % (name, 100.0 * v[i] / max(total, 1), v[16 + i] // reps, v[i] / max(v[16 + i], 1)))


if __name__ == "__main__":
    main()

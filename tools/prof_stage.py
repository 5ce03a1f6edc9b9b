#!/usr/bin/env python3
"""Runs one stage of the hot path a few times on synthetic input; meant to sit
behind `rocprofv3 --pmc ... -- python3 tools/prof_stage.py ...` so that the
counter pass sees only a handful of dispatches.

    python3 tools/prof_stage.py [--workload uniform100m] [--stage count|bin|all] [--reps 3]
"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="uniform100m")
    ap.add_argument("--stage", default="all", choices=["count", "bin", "all", "pairs"])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--bin-variant", type=int, default=None)
    ap.add_argument("--count-variant", type=int, default=None)
    ap.add_argument("--lib", default=None, help="another build of libpaircount.so (A/B experiments)")
    ap.add_argument("--as-rank", type=int, nargs=2, metavar=("R", "W"), default=None,
                    help="the slab rank R of W would hold (equal cuts; weak-scaling workloads use W times the points)")
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    import bench
    psa = load_package()
    if args.lib:
        psa.LIB_PATH = os.path.abspath(args.lib)
        psa._lib = None
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    kind_name, points, sigma, _ = bench.WORKLOADS[args.workload]
    kind = ht.UNIFORM if kind_name == "uniform" else ht.BLOBS
    if args.bin_variant is not None:
        psa.set_variant("bin", args.bin_variant)
    if args.count_variant is not None:
        psa.set_variant("count", args.count_variant)
    x_lo, x_hi = 0, 400
    if args.as_rank:
        r, w = args.as_rank
        if bench.WORKLOADS[args.workload][3] == "weak":
            points *= w
        x_lo, x_hi = 400 * r // w, 400 * (r + 1) // w
        host = ht.generate(kind, 7, points, sigma, x_lo=x_lo, x_hi=min(x_hi, 399))
    else:
        host = ht.generate(kind, 7, points, sigma)
    xyz = torch.from_numpy(host).to("cuda:0")
    st = psa.DeviceStages(xyz.shape[0], "cuda:0")
    n = st.bin(xyz)
    torch.cuda.synchronize()
    for _ in range(args.reps):
        if args.stage in ("bin", "all"):
            st.bin(xyz)
        if args.stage in ("count", "all"):
            st.count_sorted(n, x_lo, x_hi)
        if args.stage == "pairs":                 # the whole-path entry point (what bench.py times)
            st.count_pairs(xyz, x_lo, x_hi)
    torch.cuda.synchronize()
    print("pairs", st.pairs())


if __name__ == "__main__":
    main()

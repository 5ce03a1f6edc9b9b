#!/usr/bin/env python3
"""A/B timing of differently built copies of libpaircount.so in one process.

    python3 tools/exp_libs.py [--workload uniform100m] [--reps 10] LIB [LIB ...]

Each LIB is a path to a build of the library (e.g. made with extra -D flags into
particle-simulation_amd/build/).  For every one the input is binned once, then the
binning and the pair test are timed with events on the launch stream; the pair
count is printed so that a variant which changes results is seen at once.
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
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--kernels", action="store_true", help="print per-kernel event times")
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    import bench
    psa = load_package()
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    kind_name, points, sigma, _ = bench.WORKLOADS[args.workload]
    kind = ht.UNIFORM if kind_name == "uniform" else ht.BLOBS
    xyz = torch.from_numpy(ht.generate(kind, 7, points, sigma)).to("cuda:0")
    for path in args.libs:
        psa.LIB_PATH = os.path.abspath(path)
        psa._lib = None
        st = psa.DeviceStages(points, "cuda:0")
        n = st.bin(xyz)
        st.count_sorted(n)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tb = tc = 0.0
        psa.profile_enable(True)
        for _ in range(args.reps):
            ev[0].record()
            st.bin(xyz)
            ev[1].record()
            st.count_sorted(n)
            ev[2].record()
            torch.cuda.synchronize()
            tb += ev[0].elapsed_time(ev[1])
            tc += ev[1].elapsed_time(ev[2])
        prof = psa.profile_read()
        psa.profile_enable(False)
        print("%-60s bin %.3f ms  count %.3f ms  pairs %d" % (os.path.basename(path), tb / args.reps, tc / args.reps,
                                                            st.pairs()), flush=True)
        if args.kernels:
            for k, (timed, launches, ms) in sorted(prof.items()):
                if timed:
                    print("    %-28s %.4f ms" % (k, ms / timed))
        del st
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

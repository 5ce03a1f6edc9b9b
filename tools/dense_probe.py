#!/usr/bin/env python3
"""Pair test of well-filled grids: the dense kernel against variant 3's for the segment list.

    python3 tools/dense_probe.py [--lib LIB] [--points N ...] [--rank R W]

For each N uniform points (or the slab rank R of W holds of W*N points) the input is binned once and
the pair test is timed with pc_set_variant("dense", 0) and ("dense", 1); the counts must agree.
"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--points", type=int, nargs="+", default=[125_000_000])
    ap.add_argument("--rank", type=int, nargs=2, default=None)
    ap.add_argument("--kind", default="uniform", choices=["uniform", "blobs"])
    ap.add_argument("--sigma", type=int, default=0, help="blob width in micro-units")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    psa = load_package()
    if args.lib:
        psa.LIB_PATH = os.path.abspath(args.lib)
        psa._lib = None
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    kind = ht.UNIFORM if args.kind == "uniform" else ht.BLOBS
    for points in args.points:
        x_lo, x_hi = 0, 400
        if args.rank:
            r, w = args.rank
            x_lo, x_hi = 400 * r // w, 400 * (r + 1) // w
            host = ht.generate(kind, 7, points * w, args.sigma, x_lo=x_lo, x_hi=min(x_hi, 399))
        else:
            host = ht.generate(kind, 7, points, args.sigma)
        xyz = torch.from_numpy(host).to("cuda:0")
        st = psa.DeviceStages(xyz.shape[0], "cuda:0")
        n = st.bin(xyz)
        for dense in (0, 1, 0, 1):
            psa.set_variant("dense", dense)
            st.count_sorted(n, x_lo, x_hi)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(args.reps):
                st.count_sorted(n, x_lo, x_hi)
            ev[1].record()
            torch.cuda.synchronize()
            print("points held %d layers [%d,%d) dense=%d: pair test %.3f ms, pairs %d" %
                  (xyz.shape[0], x_lo, x_hi, dense, ev[0].elapsed_time(ev[1]) / args.reps, st.pairs()), flush=True)
        psa.set_variant("dense", 1)
        del st, xyz
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where the workgroups of the partition kernels spend their time: runs the binning of a workload with a build of the
library made with -DPC_L_TRACE (per-phase shader-clock sums, see pc_bin_lds.h) and prints the average cycles per
workgroup and phase.

    hipcc ... -DPC_L_TRACE -shared -o exp/lib_trace.so paircount.hip
    python3 tools/partition_phases.py exp/lib_trace.so [--workload uniform100m]
"""
import argparse
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PHASES = ["load the tile", "bin + rank (LDS atomics)", "scan of the bins", "claims issued, tile scattered to LDS",
          "claims answered + barrier", "write-out issued", "stores acknowledged"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--workload", default="uniform100m")
    args = ap.parse_args()
    import torch
    from __graft_entry__ import load_package
    import bench
    psa = load_package()
    psa.LIB_PATH = os.path.abspath(args.lib)
    psa._lib = None
    ht = importlib.import_module("particle_simulation_amd.hosttools")
    kind_name, points, sigma, _ = bench.WORKLOADS[args.workload]
    kind = ht.UNIFORM if kind_name == "uniform" else ht.BLOBS
    xyz = torch.from_numpy(ht.generate(kind, 7, points, sigma)).to("cuda:0")
    st = psa.DeviceStages(points, "cuda:0")
    st.bin(xyz)
    out = (ctypes.c_ulonglong * 16)()
    raw = ctypes.CDLL(psa.LIB_PATH)
    assert raw.pc_trace_read(out) == 0          # (clears the counters of the warm-up)
    reps = 3
    psa.profile_enable(1)
    for _ in range(reps):
        st.bin(xyz)
    assert raw.pc_trace_read(out) == 0
    prof = psa.profile_read()
    psa.profile_enable(False)
    for k in ("pc_l3_partition_x", "pc_l6_partition_y"):
        timed, launches, ms = prof[k]
        print("%s: %.3f ms per launch in this build" % (k, ms / timed))
    for level, name in enumerate(["level X (pc_l3_partition_x)", "level Y (pc_l6_partition_y)"]):
        v = [int(x) for x in out[8 * level:8 * level + 8]]
        n = max(v[7], 1)
        total = sum(v[:7]) / n
        print("%s: %d workgroups per run, %.0f cycles each" % (name, n // reps, total))
        for ph, c in zip(PHASES, v[:7]):
            print("   %-40s %8.0f cycles  %5.1f %%" % (ph, c / n, 100.0 * c / n / total))


if __name__ == "__main__":
    main()

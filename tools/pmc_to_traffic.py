#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (csv) into per-kernel HBM-side traffic per launch.

Read pass:  TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
Write pass: TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum

The L2's memory-side requests come in three read sizes and two write sizes, each with a counter of its own on
gfx950, so the bytes need no correction factor:
    read  = 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B      (RDREQ = their sum: checked here)
    write = 64 * WRREQ_64B + 32 * (WRREQ - WRREQ_64B)
(FETCH_SIZE, the derived counter, tallies 64 bytes for every request whatever its size: half the bytes of a
streaming kernel, whose requests are 128-byte ones, and about right for a gather of 32- and 64-byte ones.)

    python3 tools/pmc_to_traffic.py <read_dir> <write_dir> <out.json> [workload]
"""
import collections
import csv
import glob
import importlib.util
import json
import os
import sys


def device_source_hash():
    """The hash bench.py checks before it quotes this file (hosttools.device_source_hash)."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "particle-simulation_amd", "hosttools.py")
    spec = importlib.util.spec_from_file_location("pc_hosttools_for_hash", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.device_source_hash()

NAMES = {"pc_l36_partition<0>": "pc_l3_partition_x", "pc_l36_partition<1>": "pc_l6_partition_y",
         "pc_l7_sort_pencils<768>": "pc_l7_sort_pencils", "pc_l7_sort_pencils<192>": "pc_l7_sort_pencils"}


def per_kernel(directory):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            vals[NAMES.get(k, k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in vals.items()}


def main():
    rd, wr = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    out = {"workload": sys.argv[4] if len(sys.argv) > 4 else "uniform100m",
           "device_source_hash": device_source_hash(),
           "method": "rocprofv3 --pmc, one pass for the read request counters (TCC_EA0_RDREQ, _32B, _64B, _128B) and one "
                     "for the write request counters (TCC_EA0_WRREQ, _64B); bytes = sum of request sizes; averages per "
                     "launch over the launches a pass saw (the dispatches of 'pc_c3_count_groups' are two gated launches "
                     "per call, one of which returns at once)",
           "kernels": {}}
    for k in sorted(set(rd) | set(wr)):
        if not k.startswith("pc_"):
            continue
        r, w = rd.get(k, {}), wr.get(k, {})
        n32, n64, n128 = (r.get("TCC_EA0_RDREQ_%s_sum" % s, 0.0) for s in ("32B", "64B", "128B"))
        rb = 32 * n32 + 64 * n64 + 128 * n128
        w64, wall = w.get("TCC_EA0_WRREQ_64B_sum", 0.0), w.get("TCC_EA0_WRREQ_sum", 0.0)
        wb = 64 * w64 + 32 * (wall - w64)
        out["kernels"][k] = {"read_bytes": rb, "write_bytes": wb, "hbm_bytes": rb + wb,
                             "read_requests": {"32B": n32, "64B": n64, "128B": n128, "all": r.get("TCC_EA0_RDREQ_sum", 0.0)},
                             "write_requests": {"64B": w64, "all": wall}}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in out["kernels"].items():
        q = v["read_requests"]
        print("%-24s read %.3f GB (32B %.3g, 64B %.3g, 128B %.3g of %.3g requests)  write %.3f GB"
              % (k, v["read_bytes"] / 1e9, q["32B"], q["64B"], q["128B"], q["all"], v["write_bytes"] / 1e9))


if __name__ == "__main__":
    main()

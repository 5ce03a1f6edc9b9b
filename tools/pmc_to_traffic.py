#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv) into per-kernel HBM
traffic per launch.  gfx950 correction: FETCH_SIZE tallies 64 B per 128 B request
on streaming reads, so read bytes = 2 x FETCH_SIZE (KB -> bytes x 1024);
WRITE_SIZE is exact for streaming stores.

    python3 tools/pmc_to_traffic.py <fetch_dir> <write_dir> <out.json> [workload]
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(directory, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(directory + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"workload": sys.argv[4] if len(sys.argv) > 4 else "uniform100m",
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; read bytes = 2*FETCH_SIZE*1024 "
                     "(gfx950 half-count correction), write bytes = WRITE_SIZE*1024; per-launch averages",
           "kernels": {}}
    names = {"pc_l36_partition<0>": "pc_l3_partition_x", "pc_l36_partition<1>": "pc_l6_partition_y",
             "pc_l7_sort_pencils<768>": "pc_l7_sort_pencils", "pc_l7_sort_pencils<192>": "pc_l7_sort_pencils"}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("pc_"):
            continue
        rb, wb = 2 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        out["kernels"][names.get(k, k)] = {"read_bytes": rb, "write_bytes": wb, "hbm_bytes": rb + wb}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in out["kernels"].items():
        print("%-24s read %.3f GB  write %.3f GB" % (k, v["read_bytes"] / 1e9, v["write_bytes"] / 1e9))


if __name__ == "__main__":
    main()

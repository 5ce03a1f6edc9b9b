#!/usr/bin/env python3
"""Prints per-kernel averages of every counter found in rocprofv3 --pmc csv output.

    python3 tools/pmc_dump.py <dir> [kernel-substring]
"""
import collections
import csv
import glob
import sys


def main():
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            vals[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    want = sys.argv[2] if len(sys.argv) > 2 else "pc_"
    for k in sorted(vals):
        if want in k:
            print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(vals[k].items())))


if __name__ == "__main__":
    main()

#!/bin/bash
# The four bench lines that a change of the kernels is judged on: 100 M uniform points, 10 M, the slab of rank 3 of 8
# of the 100 M-point run, and the clustered 100 M points.  Usage: tools/bench_set.sh <output directory>
set -e
out=${1:?output directory}
mkdir -p "$out"
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-e2e-full --generate device"
$B --steps 30 > "$out/bench_uniform100m.json" 2> "$out/bench_uniform100m.err"
$B --steps 50 --workload uniform10m > "$out/bench_uniform10m.json" 2> "$out/bench_uniform10m.err"
$B --steps 50 --as-rank 3 8 > "$out/bench_uniform100m_rank3of8.json" 2> "$out/bench_uniform100m_rank3of8.err"
$B --steps 10 --workload blobs100m > "$out/bench_blobs100m.json" 2> "$out/bench_blobs100m.err"
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "ms/step %.4f binning %.4f pair test %.4f pairs %d" % (d["ms_per_step"], d["binning_ms"], d["pair_test_ms"], d["pairs"]))
    print("   ", " ".join("%s=%.4f" % (k.replace("pc_", ""), v["avg_ms"]) for k, v in d["kernels"].items()))
PY

set -e
tag=$1
mkdir -p gpurun_out
bash tools/profile_round.sh $tag uniform100m > gpurun_out/${tag}_profile_round.log 2>&1
for w in uniform10m blobs100m; do
  timeout -k 10 400 python3 bench.py --workload $w --no-e2e-full > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err
done
timeout -k 10 300 python3 bench.py --workload uniform1b --as-rank 3 8 --generate device --no-cpu-baseline --steps 10 > gpurun_out/${tag}_bench_1b_rank3.json 2> gpurun_out/${tag}_bench_1b_rank3.err
for rw in "0 2" "1 4" "3 8"; do
  set -- $rw
  timeout -k 10 300 python3 bench.py --as-rank $1 $2 --generate device --no-cpu-baseline --steps 50 > gpurun_out/${tag}_bench_uniform100m_rank$1of$2_preview.json 2> gpurun_out/${tag}_preview.err
done
python3 - $tag <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % sys.argv[1])):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "ms/step %.4f binning %.4f pair %.4f pairs %d frac %.3f binfrac %.3f" % (d["ms_per_step"], d["binning_ms"], d["pair_test_ms"], d["pairs"], d["roofline"]["frac"], d["binning_roofline"]["frac"]))
PY

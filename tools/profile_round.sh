#!/bin/bash
# One measurement set for the record: bench line, rocprofv3 kernel stats of the
# same command, HBM-side traffic (two PMC passes) and the pair-test SQ counters.
#   tools/profile_round.sh <tag> [workload]     (writes gpurun_out/<tag>_*)
set -e
tag=$1
wl=${2:-uniform100m}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python3 bench.py --workload $wl > gpurun_out/${tag}_bench_$wl.json 2> gpurun_out/${tag}_bench_$wl.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --workload $wl --steps 20 --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d gpurun_out/${tag}_rd -- python3 tools/prof_stage.py --workload $wl --stage all --reps 2 > gpurun_out/${tag}_rd.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d gpurun_out/${tag}_wr -- python3 tools/prof_stage.py --workload $wl --stage all --reps 2 > gpurun_out/${tag}_wr.log 2>&1
python3 tools/pmc_to_traffic.py gpurun_out/${tag}_rd gpurun_out/${tag}_wr gpurun_out/${tag}_traffic_$wl.json $wl > gpurun_out/${tag}_traffic_$wl.txt
bash tools/pmc_count.sh gpurun_out/${tag}_sq --workload $wl > gpurun_out/${tag}_sq_counters_pair_test.txt
find gpurun_out/${tag}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_${wl}_kernel_stats.csv \;
tail -c 400 gpurun_out/${tag}_bench_$wl.json

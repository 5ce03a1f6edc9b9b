#!/bin/bash
# One measurement set for the record: bench line, rocprofv3 kernel stats of the
# same command, HBM traffic (two PMC passes) and the pair-test SQ counters.
#   tools/profile_round.sh <tag>      (writes gpurun_out/<tag>_*)
set -e
tag=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 20 --no-cpu-baseline > gpurun_out/${tag}_stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 tools/prof_stage.py --stage all --reps 2 > gpurun_out/${tag}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 tools/prof_stage.py --stage all --reps 2 > gpurun_out/${tag}_write.log 2>&1
bash tools/pmc_count.sh gpurun_out/${tag}_sq
tail -c 600 gpurun_out/${tag}_bench.json

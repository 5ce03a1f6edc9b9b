#!/bin/bash
# Third SQ counter pass over the pair-test kernel: lane use of the vector instructions, branch and LDS waits.
#   tools/pmc_count3.sh <out-dir> [prof_stage.py arguments]
set -e
out=$1; shift
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_LDS \
    --output-format csv -d $out/sq3 -- python3 tools/prof_stage.py --stage count --reps 2 "$@" > $out.sq3.log 2>&1
python3 tools/pmc_dump.py $out pc_c

#!/bin/bash
# Three L2 (TCC) counter passes over the binning kernels (rocprofv3 --pmc, one pass per run, four TCC counters each):
# busy and stall cycles of the L2 towards memory, and the average time a read or write request spends outstanding.
#   tools/pmc_l2.sh <out-dir> [prof_stage.py arguments]
set -e
out=$1; shift
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCC_CYCLE_sum TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum \
    --output-format csv -d $out/l2a -- python3 tools/prof_stage.py --stage bin --reps 2 "$@" > $out.l2a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum \
    --output-format csv -d $out/l2b -- python3 tools/prof_stage.py --stage bin --reps 2 "$@" > $out.l2b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum \
    --output-format csv -d $out/l2c -- python3 tools/prof_stage.py --stage bin --reps 2 "$@" > $out.l2c.log 2>&1
python3 tools/pmc_dump.py $out pc_l

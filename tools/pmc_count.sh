#!/bin/bash
# Two SQ counter passes over the pair-test kernel (rocprofv3 --pmc, one pass per run).
#   tools/pmc_count.sh <out-dir> [prof_stage.py arguments]
set -e
out=$1; shift
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM \
    --output-format csv -d $out/sq1 -- python3 tools/prof_stage.py --stage ${PC_STAGE:-count} --reps 2 "$@" > $out.sq1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $out/sq2 -- python3 tools/prof_stage.py --stage ${PC_STAGE:-count} --reps 2 "$@" > $out.sq2.log 2>&1
python3 tools/pmc_dump.py $out pc_c

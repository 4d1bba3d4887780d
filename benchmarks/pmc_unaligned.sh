#!/bin/bash
# LDS alignment counters of any script: bash benchmarks/pmc_unaligned.sh <kernel-name-substring> <python script> [args...]
PAT=$1; shift
REPO=$(pwd)
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pu
timeout 300 rocprofv3 --pmc SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d /tmp/pu -- python $SCRIPT "$@" > /tmp/pu.log 2>&1
python $REPO/benchmarks/pmcq.py $(find /tmp/pu -name "*.db" | head -1) $PAT

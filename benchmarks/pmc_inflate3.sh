#!/bin/bash
# SQ counters for the wave-per-stream inflate kernel: bash benchmarks/pmc_inflate3.sh [members] [variant]
NB=${1:-16384}; VAR=${2:-10}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
    i=$((i+1)); rm -rf /tmp/p3_$i
    timeout 200 rocprofv3 --pmc $set -d /tmp/p3_$i -- python $REPO/benchmarks/inflate_profile.py $NB $VAR > /tmp/p3_$i.log 2>&1
    db=$(find /tmp/p3_$i -name "*.db" | head -1)
    python $REPO/benchmarks/pmcq.py $db inflate3 2>&1 | awk '{print $(NF-2), $(NF-1), $NF}'
done

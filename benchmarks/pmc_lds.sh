#!/bin/bash
# LDS/VALU/SALU busy counters for one LZ4 decode variant (run on the GPU box via gpurun): bash benchmarks/pmc_lds.sh text 10
KIND=${1:-text}; VAR=${2:-0}
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i+1))
    rm -rf /tmp/pl_$i
    rocprofv3 --pmc $set -d /tmp/pl_$i -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-others --kind $KIND --variant $VAR > /tmp/pl_$i.log 2>&1
    db=$(find /tmp/pl_$i -name "*.db" | head -1)
    python $REPO/benchmarks/pmcq.py $db lz4_decode 2>&1 | cut -c1-20,40- || tail -5 /tmp/pl_$i.log
done

#!/bin/bash
# SQ counters of the range coder's kernels in the pipeline (benchmarks/pipeline_stages.py at $1 of 10^9 bytes), one rocprofv3 --pmc
# pass per set, no trace domains.  Output: gpurun_out/pmc_ari.txt (averages per dispatch)
SCALE=${1:-0.25}
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $REPO/gpurun_out/pmc_ari.txt
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
    i=$((i+1))
    rm -rf /tmp/pmca_$i
    timeout 600 rocprofv3 --pmc $set -d /tmp/pmca_$i -- python $REPO/benchmarks/pipeline_stages.py $SCALE > /tmp/pmca_$i.log 2>&1
    db=$(find /tmp/pmca_$i -name "*.db" | head -1)
    for k in k_ari_byte_quad k_dc_decode; do
        python $REPO/benchmarks/pmcq.py $db $k >> $REPO/gpurun_out/pmc_ari.txt 2>&1 || tail -5 /tmp/pmca_$i.log >> $REPO/gpurun_out/pmc_ari.txt
    done
done
cat $REPO/gpurun_out/pmc_ari.txt

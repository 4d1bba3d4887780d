#!/bin/bash
# PMC passes over the LZ4 decode bench (run on the GPU box via gpurun).  Each pass is its own rocprofv3 run
# with --pmc only (no trace domains).  Output: gpurun_out/pmc_<n>.txt
KIND=${1:-text}
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH"; do
    i=$((i+1))
    rm -rf /tmp/pmc_$i
    rocprofv3 --pmc $set -d /tmp/pmc_$i -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-others --kind $KIND > /tmp/pmc_$i.log 2>&1
    db=$(find /tmp/pmc_$i -name "*.db" | head -1)
    python $REPO/benchmarks/pmcq.py $db lz4_decode > $REPO/gpurun_out/pmc_$i.txt 2>&1 || tail -5 /tmp/pmc_$i.log > $REPO/gpurun_out/pmc_$i.txt
done
cat $REPO/gpurun_out/pmc_*.txt

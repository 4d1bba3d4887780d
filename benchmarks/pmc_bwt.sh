#!/bin/bash
# SQ counters of the forward BWT's kernels (config 4, one distribution: $1 = text | dna4), one rocprofv3 --pmc pass per set,
# no trace domains.  Output: gpurun_out/pmc_bwt_<kind>.txt (averages per dispatch)
KIND=${1:-text}
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $REPO/gpurun_out/pmc_bwt_$KIND.txt
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
    i=$((i+1))
    rm -rf /tmp/pmcb_$i
    timeout 600 rocprofv3 --pmc $set -d /tmp/pmcb_$i -- python $REPO/benchmarks/bench_configs.py --configs 4 --kinds $KIND --once > /tmp/pmcb_$i.log 2>&1
    db=$(find /tmp/pmcb_$i -name "*.db" | head -1)
    python $REPO/benchmarks/pmcq.py $db k_bws_ >> $REPO/gpurun_out/pmc_bwt_$KIND.txt 2>&1 || tail -5 /tmp/pmcb_$i.log >> $REPO/gpurun_out/pmc_bwt_$KIND.txt
done
grep "local_wg<unsigned long>\|k_bws_dense<unsigned int>\|k_bws_gather" $REPO/gpurun_out/pmc_bwt_$KIND.txt

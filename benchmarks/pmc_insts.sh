#!/bin/bash
# Instruction-mix counters of the LZ4 decode kernels (one rocprofv3 --pmc run per variant): bash benchmarks/pmc_insts.sh "0 17"
# Variants other than 0 / 11 need the A/B library (RCX_AB=1).  Output: gpurun_out/pmc_insts_v<variant>.txt
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in ${1:-0}; do
    rm -rf /tmp/pi_$V
    RCX_AB=1 RCX_BENCH_EXPERIMENT_NOCHECK=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES -d /tmp/pi_$V -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-others --variant $V > /tmp/pi_$V.log 2>&1
    db=$(find /tmp/pi_$V -name "*.db" | head -1)
    python $REPO/benchmarks/pmcq.py $db lz4_decode > $REPO/gpurun_out/pmc_insts_v$V.txt 2>&1 || tail -5 /tmp/pi_$V.log > $REPO/gpurun_out/pmc_insts_v$V.txt
    echo "== variant $V"; cat $REPO/gpurun_out/pmc_insts_v$V.txt
done

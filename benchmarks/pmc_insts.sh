#!/bin/bash
# Instruction-mix counters of the LZ4 decode kernels (one rocprofv3 --pmc run per variant): bash benchmarks/pmc_insts.sh "0 23"
# Variants other than the shipped ones need the A/B library (RCX_AB=1).  Output: gpurun_out/pmc_insts_v<variant>.json
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in ${1:-0}; do
    rm -rf /tmp/pi_$V
    RCX_AB=1 RCX_BENCH_EXPERIMENT_NOCHECK=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES -d /tmp/pi_$V -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-others --no-dists --variant $V > /tmp/pi_$V.log 2>&1
    db=$(find /tmp/pi_$V -name "*.db" | head -1)
    python $REPO/benchmarks/pmc_insts.py $db lz4_decode 268435456 $REPO/gpurun_out/pmc_insts_v$V.json "bench.py default workload (4096 x 64 KiB G-text, 5510 sequences a block), LZ4 decode variant $V" 2>&1 || tail -5 /tmp/pi_$V.log
done

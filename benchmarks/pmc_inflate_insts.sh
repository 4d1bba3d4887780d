#!/bin/bash
# Instruction mix of the DEFLATE kernel variants on BASELINE config 3: bash benchmarks/pmc_inflate_insts.sh "10 12"
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in ${1:-0}; do
    rm -rf /tmp/pii_$V
    RCX_INFLATE_VARIANT=$V timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES -d /tmp/pii_$V -- python $REPO/benchmarks/bench_configs.py --configs 3 --once > /tmp/pii_$V.log 2>&1
    db=$(find /tmp/pii_$V -name "*.db" | head -1)
    python $REPO/benchmarks/pmc_insts.py $db k_inflate3 1073741824 $REPO/gpurun_out/pmc_insts_inflate_v$V.json "BASELINE config 3 (65536 x 16 KiB zlib members), DEFLATE variant $V" 2>&1 | cut -c1-900 || tail -5 /tmp/pii_$V.log
done

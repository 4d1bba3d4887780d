#!/bin/bash
# Kernel-trace stats of BASELINE configs 3, 4, 5 (run on the GPU box via gpurun).  One rocprofv3 run per config,
# --kernel-trace --stats only.  Outputs gpurun_out/cfg<N>_kernel_stats.csv + the bench lines.
TAG=${1:-r02}
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in 3 4 5; do
    rm -rf /tmp/kt$cfg
    SCALE=1.0
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$cfg -- python $REPO/benchmarks/bench_configs.py --configs $cfg --scale $SCALE > /tmp/kt$cfg.log 2>&1
    grep "^{\"config\"" /tmp/kt$cfg.log > $REPO/gpurun_out/${TAG}_cfg${cfg}_lines.jsonl
    f=$(find /tmp/kt$cfg -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $REPO/gpurun_out/${TAG}_cfg${cfg}_kernel_stats.csv
    tail -3 /tmp/kt$cfg.log
done
head -12 $REPO/gpurun_out/${TAG}_cfg*_kernel_stats.csv | cut -c1-160

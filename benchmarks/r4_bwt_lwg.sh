#!/bin/bash
# the LDS sorter alone: time (kernel-trace stats) and vector instructions (one --pmc pass) of k_bws_local_wg / k_bws_local_wave, config 4 text
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktl -- python $REPO/benchmarks/bench_configs.py --configs 4 --kinds text > /tmp/ktl.log 2>&1
f=$(find /tmp/ktl -name "*kernel_stats.csv" | head -1); grep "k_bws_local" $f | cut -d, -f1-4
rm -rf /tmp/pml; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pml -- python $REPO/benchmarks/bench_configs.py --configs 4 --kinds text --once > /tmp/pml.log 2>&1
python $REPO/benchmarks/pmcq.py $(find /tmp/pml -name "*.db" | head -1) k_bws_local_wg

#!/bin/bash
# Counters of one kernel of one of the benchmarks/ scripts (one rocprofv3 --pmc pass):
#   bash benchmarks/pmc_kernel.sh "<counters>" <kernel substring> <script and args ...>
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
C="$1"; K="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout 600 rocprofv3 --pmc $C -d /tmp/pk -- python $REPO/benchmarks/$@ > /tmp/pk.log 2>&1
db=$(find /tmp/pk -name "*.db" | head -1)
python $REPO/benchmarks/pmcq.py $db "$K" 2>&1 | tee $REPO/gpurun_out/pmc_kernel.txt || tail -5 /tmp/pk.log

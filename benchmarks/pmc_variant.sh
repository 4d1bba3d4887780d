#!/bin/bash
V=$1
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf /tmp/pw
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu --no-e2e --no-others --variant $V > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu --no-e2e --no-others --variant $V > /tmp/pw.log 2>&1
python $REPO/benchmarks/pmc_traffic.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) lz4_decode 366370784 /tmp/pmc_v$V.json
grep -E "hbm_read|hbm_write|hbm_bytes" /tmp/pmc_v$V.json

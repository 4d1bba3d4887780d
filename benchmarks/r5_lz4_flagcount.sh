#!/bin/bash
# Instruction counters (per batch of the headline: per CU / 1568) and time of the headline kernel for builds with extra flags:
#   FLAGSETS="-DA=1|-DB=2 -DC=3" bash benchmarks/r5_lz4_flagcount.sh      ('|' separates builds)
REPO=$(pwd)
IFS='|' read -ra SETS <<< "${FLAGSETS:--DRCX_NONE=1}"
for flags in "${SETS[@]}"; do
    cd $REPO
    RCX_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
    T=$(RCX_EXTRA_FLAGS="$flags" RCX_BENCH_EXPERIMENT_NOCHECK=1 python bench.py --no-cpu --no-e2e --no-others --no-dists --steps 40 --warmup 10 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_ms_avg'])")
    cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fc
    RCX_EXTRA_FLAGS="$flags" RCX_BENCH_EXPERIMENT_NOCHECK=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -d /tmp/fc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-others --no-dists ${BENCH_ARGS} > /tmp/fc.log 2>&1
    db=$(find /tmp/fc -name "*.db" | head -1)
    python - "$db" "$flags" "$T" <<'P'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
r = {c: v for c, v in cur.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%lz4_decode%' group by counter_name")}
print("%-44s %s ms | per batch: SALU %.0f  VALU %.0f  LDS %.0f  BRANCH %.0f" % (sys.argv[2], sys.argv[3], r.get("SQ_INSTS_SALU", 0) / 256 / 1568, r.get("SQ_INSTS_VALU", 0) / 256 / 1568, r.get("SQ_INSTS_LDS", 0) / 256 / 1568, r.get("SQ_INSTS_BRANCH", 0) / 256 / 1568))
P
done
cd $REPO
python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1

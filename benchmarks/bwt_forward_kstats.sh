#!/bin/bash
# Per-kernel time of one BWT forward batch (rocprofv3 --kernel-trace --stats): bash benchmarks/bwt_forward_kstats.sh [kind] [nblocks]
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kf
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kf -- python $REPO/benchmarks/bwt_forward_profile.py ${1:-text} ${2:-1024} ${3:-262144} > /tmp/kf.log 2>&1
f=$(find /tmp/kf -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    t = float(r["TotalDurationNs"]) / 2e6           # the script runs the batch twice
    tot += t
    print("%-60s calls %5d  %.2f ms" % (r["Name"][:60], int(r["Calls"]) // 2, t))
print("total %.2f ms per batch" % tot)
PY

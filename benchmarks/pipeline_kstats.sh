#!/bin/bash
# Per-kernel time of the BWT -> DC -> Ari pipeline's launches (rocprofv3 --kernel-trace --stats): bash benchmarks/pipeline_kstats.sh [scale] [filter]
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python $REPO/benchmarks/pipeline_stages.py ${1:-1.0} > /tmp/kp.log 2>&1
tail -8 /tmp/kp.log
f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1)
python3 - "$f" "${2:-}" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"] and float(r["TotalDurationNs"]) > 2e5:
        print("%-80s calls %5d  avg %.3f ms  total %.2f ms" % (r["Name"][:80], int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY

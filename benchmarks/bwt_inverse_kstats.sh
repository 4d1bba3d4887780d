#!/bin/bash
# Per-kernel time of the inverse BWT's launches (rocprofv3 --kernel-trace --stats): bash benchmarks/bwt_inverse_kstats.sh [--sweep]
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ki
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ki -- python $REPO/benchmarks/bwt_inverse_rate.py "$@" > /tmp/ki.log 2>&1
tail -25 /tmp/ki.log
f=$(find /tmp/ki -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "bwti" in r["Name"].lower() or "bwt_inverse" in r["Name"]:
        print("%-90s calls %5d  avg %.3f ms" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"]) / 1e6))
PY

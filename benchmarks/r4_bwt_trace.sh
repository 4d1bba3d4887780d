#!/bin/bash
# every dispatch of the forward BWT in order with its duration (one forward transform of config 4, $1: text | dna4)
KIND=${1:-text}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktt
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktt -- python $REPO/benchmarks/bench_configs.py --configs 4 --kinds $KIND --once > /tmp/ktt.log 2>&1
f=$(find /tmp/ktt -name "*kernel_trace.csv" | head -1)
python - <<P
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None; last_end = None
for r in rows:
    n = r["Kernel_Name"]
    if not ("bws" in n or "bwtf" in n): continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    gap = (s - last_end) / 1e3 if last_end else 0
    last_end = e
    print("%8.3f ms  +%7.1f us gap  %8.1f us  grid %-8s %s" % ((s - t0) / 1e6, gap, (e - s) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "")), n[:60]))
P

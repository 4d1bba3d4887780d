#!/bin/bash
# every kernel and copy of one host-memory LZ4 decode call in time order (is the piecewise path overlapping its copies?)
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hpt
export RCX_HP_ONLY=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/hpt -- python $REPO/benchmarks/host_path_rate.py > /tmp/hpt.log 2>&1
tail -3 /tmp/hpt.log
python - <<P
import csv, glob
ev = []
for f in glob.glob("/tmp/hpt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lz4_decode" in r["Kernel_Name"]: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel"))
for f in glob.glob("/tmp/hpt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", "copy"))))
ev.sort()
# the last call of the run: events after the last gap of > 20 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - ev[i - 1][1] > 5_000_000: cut = i
ev = ev[cut:]
t0 = ev[0][0]
for s, e, k in ev[:120]:
    print("%9.3f ms .. %9.3f ms  %8.1f us  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, k))
P

#!/bin/bash
# per-kernel times of the forward BWT, one distribution at a time ($1: text | dna4): rocprofv3 --kernel-trace --stats of config 4
KIND=${1:-text}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktb -- python $REPO/benchmarks/bench_configs.py --configs 4 --kinds $KIND > /tmp/ktb.log 2>&1
grep forward_ms /tmp/ktb.log | cut -c1-200
f=$(find /tmp/ktb -name "*kernel_stats.csv" | head -1)
cp $f $REPO/gpurun_out/r4_bwt_kernels_$KIND.csv
python - <<P
import csv
rows = list(csv.DictReader(open("$f")))
fw = [r for r in rows if "bws" in r["Name"] or "bwtf" in r["Name"]]
npass = max(int(r["Calls"]) for r in fw if "k_bws_first" in r["Name"])
tot = 0
for r in fw:
    ms = int(r["TotalDurationNs"]) / npass / 1e6; tot += ms
    print("%-50s calls/pass %5.1f  ms/pass %.3f" % (r["Name"][:50], int(r["Calls"]) / npass, ms))
print("forward kernels per pass: %.2f ms (%d passes)" % (tot, npass))
P

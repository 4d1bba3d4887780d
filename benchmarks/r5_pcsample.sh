#!/bin/bash
# PC sampling (host trap, beta) of the headline kernel: where do the waves of k_lz4_decode_v8 spend their time?
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pcs
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval ${1:-1} --output-format csv -d /tmp/pcs -- \
    python $REPO/bench.py --no-cpu --no-e2e --no-others --no-dists --steps 300 --warmup 5 > /tmp/pcs.log 2>&1
echo "rc=$?"
tail -5 /tmp/pcs.log | cut -c1-300
find /tmp/pcs -type f | head -20
f=$(find /tmp/pcs -name "*pc_sampling*.csv" | head -1)
[ -n "$f" ] && { head -5 "$f"; wc -l "$f"; mkdir -p $REPO/gpurun_out/pcs; python - "$f" $REPO/gpurun_out/pcs/r5_pcs_hist.txt <<'P'
import csv, sys, collections
f, out = sys.argv[1], sys.argv[2]
rows = csv.DictReader(open(f))
hist = collections.Counter()
cols = None
n = 0
for r in rows:
    if cols is None: cols = list(r.keys())
    key = (r.get("Instruction", ""), r.get("Instruction_Comment", ""), r.get("Code_Object_Offset", r.get("Code_Object_Id", "")))
    hist[key] += 1; n += 1
with open(out, "w") as fh:
    fh.write("columns: %s\nsamples: %d\n" % (cols, n))
    for (ins, com, off), c in hist.most_common(400):
        fh.write("%7d %6.2f%%  %s  | %s | %s\n" % (c, 100.0 * c / n, off, ins, com))
print(open(out).read()[:3000])
P
}

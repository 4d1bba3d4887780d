#!/usr/bin/env python3
"""Per-dispatch view of a rocprofv3 --kernel-trace database: python ktrace.py <dir> [run-start kernel substring]
Prints, for the LONGEST run (dispatches from one start kernel to the next), every dispatch >= 20 us in order and per-kernel totals."""
import sqlite3, sys, glob, collections
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
mark = sys.argv[2] if len(sys.argv) > 2 else "k_bwtf_hist"
con = sqlite3.connect(db); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
runs = []
for n, a, b in rows:
    if mark in n: runs.append([])
    if runs: runs[-1].append((n, a, b))
run = max(runs, key=lambda r: sum(b - a for _, a, b in r))
tot = collections.OrderedDict()
for n, a, b in run:
    short = n.split("(")[0][:48]
    tot.setdefault(short, [0, 0.0]); tot[short][0] += 1; tot[short][1] += (b - a) / 1e3
    if b - a >= 20000: print("%-50s %9.1f us" % (short, (b - a) / 1e3))
print("---- totals: wall %.1f ms, kernels %.1f ms" % ((run[-1][2] - run[0][1]) / 1e6, sum(v[1] for v in tot.values()) / 1e3))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]): print("%-50s x%-3d %9.1f us" % (k, v[0], v[1]))

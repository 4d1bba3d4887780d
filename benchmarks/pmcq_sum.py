#!/usr/bin/env python3
"""Per kernel: dispatches and the SUM of a counter over them (rocprofv3 --pmc database): pmcq_sum.py <db> <kernel substring>"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", ('%' + sys.argv[2] + '%',)))
rows.sort(key=lambda r: -r[2])
tot = {}
for r in rows:
    print("%-70s %-12s sum %.4g  n %d" % (r[0][:70], r[1], r[2], r[3]))
    tot[r[1]] = tot.get(r[1], 0) + r[2]
print("total", {k: "%.4g" % v for k, v in tot.items()})

#!/usr/bin/env python3
"""Instruction mix of one kernel from a rocprofv3 --pmc SQ_INSTS_* pass -> JSON (per launch, per CU, per decoded byte).
usage: pmc_insts.py <db> <kernel substring> <decoded bytes per launch> <out.json> [note]"""
import datetime, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as BC
db, kern, nbytes, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
cur = sqlite3.connect(db).cursor()
res = {"kernel": None, "kernel_source_hash": BC.source_hash(), "date": datetime.date.today().isoformat(),
       "how": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES (one pass, no trace domains); averages over the dispatches",
       "note": sys.argv[5] if len(sys.argv) > 5 else ""}
for k, c, v, n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", ("%" + kern + "%",)):
    res["kernel"] = k
    res[c] = {"per_launch": v, "per_CU": v / 256.0, "per_decoded_byte": v / nbytes, "dispatches": n}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:1200])

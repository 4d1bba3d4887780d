#!/usr/bin/env python3
"""HBM traffic of the headline kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass,
no trace domains) -> JSON.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request -> x2 for read bytes.
usage: pmc_traffic.py <fetch.db> <write.db> <kernel-substring> <alg_bytes_per_launch> <out.json>"""
import datetime, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def avg(db, counter, kern):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like ? group by kernel_name",
                            (counter, "%" + kern + "%")))
    if not rows:
        raise SystemExit("no rows for %s / %s in %s" % (counter, kern, db))
    rows.sort(key=lambda r: -r[2])
    return rows[0]


def main():
    fdb, wdb, kern, alg, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    kn, fetch, nf = avg(fdb, "FETCH_SIZE", kern)
    _, write, nw = avg(wdb, "WRITE_SIZE", kern)
    rd, wr = int(2 * fetch * 1024), int(write * 1024)
    json.dump({
        "kernel": kn, "workload": "bench.py default: 4096 x 64 KiB G-text",
        "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write, "dispatches": [nf, nw],
        "correction": "MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE counts 64 B per 128-B request -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 taken as is",
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": alg,
        "kernel_source_hash": __import__("bench").kernel_source_hash(), "date": datetime.date.today().isoformat(),
        "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) with rocprofv3; reads above the algorithmic bytes are the old-match (HBM -> LDS) 16-byte gathers, served mostly from L2",
    }, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()

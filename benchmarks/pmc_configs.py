#!/usr/bin/env python3
"""HBM traffic of BASELINE configs 3, 4, 5 from two rocprofv3 PMC passes each (FETCH_SIZE, WRITE_SIZE: one counter per pass, no
trace domains) over `bench_configs.py --configs N --once` (every launch exactly once) -> profiles-style JSON.
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts 64 B per 128-B request -> x 2 for read bytes; WRITE_SIZE taken as is (uncalibrated: ratios between kernels hold).
usage: pmc_configs.py <cfg> <fetch.db> <write.db> <out.json> [<label> <fetch.db> <write.db>]...   (config 4: one pair per kind)"""
import datetime, json, os, re, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as BC


def rows(db, counter):
    """[(kernel_name, value)] in dispatch order"""
    con = sqlite3.connect(db); cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    order = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else "rowid")
    try:
        return list(cur.execute("select kernel_name, sum(value) from counters_collection where counter_name = ? group by %s, kernel_name order by %s" % (order, order), (counter,)))
    except sqlite3.OperationalError:
        return list(cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)))


def product(name):
    return re.match(r"^(void )?k_", name) is not None


def short(name):
    return re.sub(r"\(.*", "", name.replace("void ", ""))


def table(fdb, wdb):
    per = {}
    seq = []
    for counter, db, key in (("FETCH_SIZE", fdb, "f"), ("WRITE_SIZE", wdb, "w")):
        for i, (k, v) in enumerate(rows(db, counter)):
            if not product(k):
                continue
            d = per.setdefault(short(k), {"dispatches": 0, "f": 0.0, "w": 0.0})
            d[key] += v
            if key == "f":
                d["dispatches"] += 1
                seq.append((short(k), v))
    wseq = [(short(k), v) for k, v in rows(wdb, "WRITE_SIZE") if product(k)]
    return per, seq, wseq


def bytes_of(f_kb, w_kb):
    rd, wr = int(2 * f_kb * 1024), int(w_kb * 1024)
    return {"FETCH_SIZE_KB_raw": round(f_kb, 1), "WRITE_SIZE_KB_raw": round(w_kb, 1), "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
            "hbm_bytes_per_launch": rd + wr}


def group(per, pred):
    f = sum(d["f"] for k, d in per.items() if pred(k))
    w = sum(d["w"] for k, d in per.items() if pred(k))
    g = bytes_of(f, w)
    g["kernel_names"] = sorted(k for k in per if pred(k))
    return g


def main():
    cfg, out = sys.argv[1], sys.argv[4]
    res = {"config": cfg, "kernel_source_hash": BC.source_hash(cfg), "date": datetime.date.today().isoformat(),
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, over `python benchmarks/bench_configs.py --configs %s --once` (every launch once)" % cfg,
           "correction": "MI355X_MICROARCH.md HBM section: KiB units; gfx950 FETCH_SIZE counts 64 B per 128-B request -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 as is"}
    fwd = lambda k: k.startswith("k_bws") or k.startswith("k_bwtf")
    inv = lambda k: k.startswith("k_bwt_inverse") or k.startswith("k_bwti")
    if cfg == "4":
        extra = sys.argv[5:]
        sets = [("text", sys.argv[2], sys.argv[3])] + [(extra[i], extra[i + 1], extra[i + 2]) for i in range(0, len(extra) - 2, 3)]
        for label, fdb, wdb in sets:
            per, _, _ = table(fdb, wdb)
            res["forward_" + label] = group(per, fwd)
            res["inverse_" + label] = group(per, inv)
            res["per_kernel_" + label] = {k: dict(dispatches=d["dispatches"], **bytes_of(d["f"], d["w"])) for k, d in sorted(per.items(), key=lambda kv: -(kv[1]["f"] + kv[1]["w"]))}
    else:
        per, seq, wseq = table(sys.argv[2], sys.argv[3])
        res["per_kernel"] = {k: dict(dispatches=d["dispatches"], **bytes_of(d["f"], d["w"])) for k, d in sorted(per.items(), key=lambda kv: -(kv[1]["f"] + kv[1]["w"]))}
        res.update(group(per, lambda k: True))
        if cfg == "5":
            # the range coder's kernel runs once per direction: its first dispatch is the encoder's
            ari_f = [v for k, v in seq if k.startswith("k_ari_byte")]
            ari_w = [v for k, v in wseq if k.startswith("k_ari_byte")]
            enc = group(per, lambda k: fwd(k) or k.startswith("k_dc_encode") or k.startswith("k_dcx"))
            dec = group(per, lambda k: inv(k) or k.startswith("k_dc_decode"))
            if len(ari_f) == 2 and len(ari_w) == 2:
                for g, i in ((enc, 0), (dec, 1)):
                    b = bytes_of(ari_f[i], ari_w[i])
                    for key in ("hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "hbm_bytes_per_launch"):
                        g[key] += b[key]
                    g["kernel_names"].append("k_ari_byte (dispatch %d of 2)" % (i + 1))
            res["encode"], res["decode"] = enc, dec
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("per_kernel")})[:1500])


if __name__ == "__main__":
    main()

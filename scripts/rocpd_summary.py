#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd (.db) outputs into the text/JSON files committed under profiles/.
  kernel stats:  python scripts/rocpd_summary.py stats  <trace.db>  > profiles/<name>_kernel_stats.txt
  PMC:           python scripts/rocpd_summary.py pmc <pmc.db> <COUNTER> [kernel-substring]
"""
import json
import sqlite3
import sys


def stats(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-78s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for r in rows:
        print("%-78s %8d %14d %12.0f %12d %12d %6.2f%%" % (r[0][:78], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total))


def pmc(db, counter, sub=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    rows = cur.execute("select * from counters_collection").fetchall()
    out = {}
    ni = cols.index("kernel_name") if "kernel_name" in cols else None
    ci = cols.index("counter_name") if "counter_name" in cols else None
    vi = cols.index("value") if "value" in cols else None
    if None in (ni, ci, vi):
        print(cols)
        return
    for r in rows:
        if r[ci] != counter:
            continue
        if sub and sub not in r[ni]:
            continue
        out.setdefault(r[ni], []).append(r[vi])
    res = {k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in out.items()}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)

#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd (.db) outputs into the text/JSON files committed under profiles/.
  kernel stats:  python scripts/rocpd_summary.py stats  <trace.db>  > profiles/<name>_kernel_stats.txt
  PMC:           python scripts/rocpd_summary.py pmc <pmc.db> <COUNTER> [kernel-substring]
  gaps:          python scripts/rocpd_summary.py gaps <trace.db>      (device idle time between consecutive kernels, by the kernel that follows the gap)
  segments:      python scripts/rocpd_summary.py segments <trace.db> [split_ms] [min_kernels]
                 (the trace cut wherever the device idles longer than split_ms — the target script sleeps around its timed region — and every piece with at
                  least min_kernels launches summarised: span, busy, idle by gap size, launches at the floor, the ten kernels in front of which the stream idles most)
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


def gaps(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    if not rows:
        print("no kernels")
        return
    short = lambda n: n.replace("pgo::", "").replace("void ", "").split("(")[0][:44]
    busy = sum(r[2] - r[1] for r in rows)
    span = rows[-1][2] - rows[0][1]
    by = {}
    prev_end = rows[0][2]
    for name, st, en in rows[1:]:
        g = max(0, st - prev_end)
        d = by.setdefault(short(name), [0, 0, 0, 0, 0, 0])
        d[0] += 1; d[1] += g; d[2] = max(d[2], g); d[3] += en - st
        if en - st < 3000:      # early exits of a stopped PCG (and other launches at the floor): what they hold the stream for
            d[4] += 1; d[5] += (en - st) + min(g, 3000)
        prev_end = max(prev_end, en)
    print("%d kernels, first start to last end %.3f ms, busy %.3f ms, idle %.3f ms" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    print("%-46s %7s %12s %10s %10s %12s %9s %14s" % ("kernel that follows the gap", "calls", "idle_total_us", "mean_us", "max_us", "busy_total_us", "under_3us", "their_span_us"))
    for k, d in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("%-46s %7d %12.1f %10.2f %10.1f %12.1f %9d %14.1f" % (k, d[0], d[1] / 1e3, d[1] / 1e3 / d[0], d[2] / 1e3, d[3] / 1e3, d[4], d[5] / 1e3))
    print("launches under 3 us: %d, holding the stream for %.3f ms in all" % (sum(d[4] for d in by.values()), sum(d[5] for d in by.values()) / 1e6))


def segments(db, split_ms=50.0, min_kernels=500):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    short = lambda n: n.replace("pgo::", "").replace("void ", "").split("(")[0][:44]
    segs, cur_seg = [], []
    for r in rows:
        if cur_seg and r[1] - max(x[2] for x in cur_seg[-4:]) > split_ms * 1e6:
            segs.append(cur_seg); cur_seg = []
        cur_seg.append(r)
    if cur_seg:
        segs.append(cur_seg)
    for si, sg in enumerate(segs):
        if len(sg) < min_kernels:
            continue
        span = max(x[2] for x in sg) - sg[0][1]
        busy = 0; prev_end = sg[0][1]
        bins = [[0, 0], [0, 0], [0, 0], [0, 0]]      # gaps < 3 us, 3-20 us, 20-200 us, > 200 us
        by = {}
        floor_n = floor_t = 0
        for name, st, en in sg:
            g = max(0, st - prev_end)
            busy += max(0, en - max(st, prev_end))
            b = 0 if g < 3000 else 1 if g < 20000 else 2 if g < 200000 else 3
            bins[b][0] += 1; bins[b][1] += g
            d = by.setdefault(short(name), [0, 0, 0])
            d[0] += 1; d[1] += g; d[2] += en - st
            if en - st < 3000:
                floor_n += 1; floor_t += en - st
            prev_end = max(prev_end, en)
        print("segment %d: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms (%.1f %%)" % (si, len(sg), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
        print("  gaps  <3 us: %d = %.3f ms | 3-20 us: %d = %.3f ms | 20-200 us: %d = %.3f ms | >200 us: %d = %.3f ms" %
              (bins[0][0], bins[0][1] / 1e6, bins[1][0], bins[1][1] / 1e6, bins[2][0], bins[2][1] / 1e6, bins[3][0], bins[3][1] / 1e6))
        print("  launches shorter than 3 us: %d, %.3f ms of kernel time (early exits of a stopped PCG and other launches at the floor)" % (floor_n, floor_t / 1e6))
        print("  %-46s %7s %12s %12s" % ("kernel that follows the gap", "calls", "idle_ms", "busy_ms"))
        for k, d in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
            print("  %-46s %7d %12.3f %12.3f" % (k, d[0], d[1] / 1e6, d[2] / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "gaps":
        gaps(sys.argv[2])
    elif sys.argv[1] == "segments":
        segments(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 50.0, int(sys.argv[4]) if len(sys.argv) > 4 else 500)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)

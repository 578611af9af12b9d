"""Where the GPU idles inside a solve: from a rocprofv3 --kernel-trace rocpd database, the gaps (> threshold us) between consecutive kernels
with the names of the kernels on both sides, and the total of all gaps.  python scripts/research/trace_gaps.py <trace.db> [threshold_us] [first_n]"""
import sqlite3, sys
db = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0; first_n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
con = sqlite3.connect(db); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
tot_gap = 0.0; tot_busy = 0.0; big = []
for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
    g = (s1 - e0) / 1e3
    tot_busy += (e0 - s0) / 1e3
    if g > 0: tot_gap += g
    if g > thr: big.append((g, n0[:50], n1[:50], s1))
print('kernels %d, busy %.1f ms, gaps %.1f ms, gaps > %.0f us: %d totalling %.1f ms' % (len(rows), tot_busy / 1e3, tot_gap / 1e3, thr, len(big), sum(b[0] for b in big) / 1e3))
from collections import Counter
c = Counter(); t = Counter()
for g, a, b, _ in big: c[(a, b)] += 1; t[(a, b)] += g
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:first_n]: print('%8.1f us total  %4d x  %-50s -> %s' % (v, c[k], k[0], k[1]))

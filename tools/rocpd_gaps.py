"""Idle time between consecutive kernels of a rocprofv3 (rocpd SQLite) kernel trace, attributed to the (previous, next) pair:
where does the timeline of a step go that no kernel accounts for?   python tools/rocpd_gaps.py trace.db [top] [skip_fraction]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5  # steady state: the second half of the run
rows = rows[int(len(rows) * skip):]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
gaps, busy = collections.defaultdict(lambda: [0, 0.0]), 0.0
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    busy += e0 - s0
    g = max(0, s1 - e0)
    k = (short(n0), short(n1))
    gaps[k][0] += 1
    gaps[k][1] += g
span = rows[-1][2] - rows[0][1]
idle = sum(v[1] for v in gaps.values())
print(f"{len(rows)} dispatches, span {span / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms, idle between kernels {idle / 1e6:.2f} ms ({100 * idle / span:.1f} %)")
hist = collections.Counter()
for (a, b), (c, t) in gaps.items():
    hist[min(int(t / c / 1e3) // 5 * 5, 100)] += c
print("gap length histogram (us, count):", sorted(hist.items()))
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{t / 1e6:8.3f} ms  {c:5d} x {t / c / 1e3:7.1f} us   {a:48s} -> {b}")

"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / average duration, like --stats CSV."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for n, c, t, a, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{n[:90]:90s} {c:6d} {t/1e6:10.3f} {a/1e3:10.1f} {mn/1e3:10.1f} {mx/1e3:10.1f} {100*t/total:6.2f}")
print(f"TOTAL kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")

"""Timeline of one training step from a rocprofv3 (rocpd SQLite) kernel trace: every kernel longer than a threshold with its queue,
start and end relative to the step's first kernel -- shows which kernels of different streams actually overlap.
    python tools/rocpd_timeline.py trace.db [min_us] [anchor kernel substring] [step index]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in cols if "queue" in c or "stream" in c), None)
rows = cur.execute(f"select {name_col}, start, end, {qcol or 0} from kernels order by start").fetchall()
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
anchor = sys.argv[3] if len(sys.argv) > 3 else "gather_batch"
starts = [i for i, r in enumerate(rows) if anchor in r[0]]
k = int(sys.argv[4]) if len(sys.argv) > 4 else 6  # which step of the run (default: inside the timed region of a --warmup 3 run)
i0, i1 = starts[k], starts[k + 1]
t0 = rows[i0][1]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
print(f"step of {(rows[i1][1] - t0) / 1e6:.2f} ms, columns: {cols}")
queues = {}
for n, s, e, q in rows[i0:i1]:
    if (e - s) / 1e3 < min_us:
        continue
    queues.setdefault(q, len(queues))
    print(f"q{queues[q]}  {(s - t0) / 1e6:7.3f} -> {(e - t0) / 1e6:7.3f}  ({(e - s) / 1e3:7.1f} us)  {short(n)}")

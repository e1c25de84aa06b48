"""Per-kernel averages of PMC counters from rocprofv3 rocpd databases (one db per counter pass)."""
import sqlite3
import sys
from collections import defaultdict

want = sys.argv[1]  # substring of kernel names to report, comma separated
dbs = sys.argv[2:]
res = defaultdict(dict)
for path in dbs:
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        print("no counters view in", path, [t for t in tabs if "pmc" in t or "counter" in t])
        continue
    cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
    kn = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "kernel" in c][0]
    cn = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vn = "value" if "value" in cols else "counter_value"
    for k, c, n, avg in cur.execute(f"select {kn}, {cn}, count(*), avg({vn}) from {view} group by {kn}, {cn}"):
        for w in want.split(","):
            if w in k:
                res[k[:70]][c] = (n, avg)
for k, d in res.items():
    print(k)
    for c, (n, avg) in sorted(d.items()):
        print(f"    {c:36s} launches {n:4d}  avg {avg:18.1f}")

"""Per-kernel duration summary from a rocprofv3 rocpd database (kernel trace)."""
import glob
import sqlite3
import sys

for path in sys.argv[1:]:
    for db_path in sorted(glob.glob(path)):
        db = sqlite3.connect(db_path)
        rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                          "from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print(f"== {db_path}")
        print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
        for name, n, s, a, mn, mx in rows[:12]:
            print(f"{name[:70]:70s} {n:7d} {s/1e6:10.3f} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")

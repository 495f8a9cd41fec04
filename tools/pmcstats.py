"""Per-kernel average of each collected PMC counter from a rocprofv3 rocpd database."""
import glob
import sqlite3
import sys

for pat in sys.argv[1:]:
    for db_path in sorted(glob.glob(pat)):
        db = sqlite3.connect(db_path)
        cols = [r[1] for r in db.execute("pragma table_info('pmc_events')")]
        print("==", db_path)
        name_col = "name" if "name" in cols else "kernel_name"
        ctr_col = "counter_name" if "counter_name" in cols else ("pmc_name" if "pmc_name" in cols else None)
        val_col = "value" if "value" in cols else "counter_value"
        if ctr_col is None:
            print("columns:", cols)
            continue
        q = (f"select {name_col}, {ctr_col}, count(*), avg({val_col}), min({val_col}), max({val_col}) "
             f"from pmc_events group by {name_col}, {ctr_col} order by sum({val_col}) desc")
        for name, ctr, n, avg, mn, mx in db.execute(q).fetchall()[:48]:
            print(f"{str(name)[:64]:64s} {ctr:22s} n={n:6d} avg={avg:14.2f} min={mn:14.2f} max={mx:14.2f}")

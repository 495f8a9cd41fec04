"""Per-launch start/duration of the pass kernels from a rocprofv3 rocpd database, in launch order (last N)."""
import glob, sqlite3, sys
path, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 260
for db_path in sorted(glob.glob(path)):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info('kernels')")]
    rows = db.execute("select name, start, end, duration from kernels order by start").fetchall()
    rows = [r for r in rows if "ts_" in r[0]][-last:]
    t0 = rows[0][1]
    prev_end = None
    for name, st, en, du in rows:
        short = "first" if "true" in name else "plain" if "ts_pass" in name else name.split("(")[0][-12:]
        gap = (st - prev_end) / 1e3 if prev_end else 0.0
        print(f"{(st - t0) / 1e3:10.1f} us  {short:8s} {du / 1e3:8.2f} us  gap {gap:7.2f}")
        prev_end = en

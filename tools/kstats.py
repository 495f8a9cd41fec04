"""Per-kernel duration summary from a rocprofv3 rocpd database (kernel trace).
avg_busy = average over the launches that did real work (duration >= half the kernel's median): the
state-machine sequence contains launches that only carry state forward (after an early
convergence, past the end of a schedule, tsamd_prepare's dry replays)."""
import glob
import sqlite3
import statistics
import sys

for path in sys.argv[1:]:
    for db_path in sorted(glob.glob(path)):
        db = sqlite3.connect(db_path)
        per = {}
        for name, d in db.execute("select name, duration from kernels"):
            per.setdefault(name, []).append(d)
        tot = sum(sum(v) for v in per.values()) or 1
        print(f"== {db_path}")
        print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'busy':>6s} {'avg_busy':>9s}")
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:12]:
            med = statistics.median(v)
            busy = [x for x in v if x >= 0.5 * med]
            print(f"{name[:70]:70s} {len(v):7d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:9.2f} {min(v)/1e3:9.2f} {max(v)/1e3:9.2f} "
                  f"{100*sum(v)/tot:6.2f} {len(busy):6d} {sum(busy)/len(busy)/1e3:9.2f}")
            if any(x in name for x in ("ts_schedule", "ts_hybrid", "ts_holblock")) and len(v) <= 24:  # one launch per schedule: the launches differ in length, list them
                print("    each launch (us): " + " ".join(f"{x/1e3:.1f}" for x in v))

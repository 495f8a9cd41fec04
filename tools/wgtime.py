"""Summary of tools/wgtime.sh output: per-workgroup start and finish times (us, relative to the earliest start),
grouped by dispatch round and by XCD (workgroup index mod 8)."""
import re
import statistics as st
import sys

for kind in ("wgtime", "wgplain"):
    t = {}
    for line in open(sys.argv[1]):
        m = re.match(kind + r" (\d+) (\d+) (\d+)", line)
        if m:
            t[int(m.group(1))] = (int(m.group(2)), int(m.group(3)))
    if not t:
        continue
    t0 = min(v[0] for v in t.values())
    print(f"== {kind}: {len(t)} workgroups; kernel span {(max(v[1] for v in t.values()) - t0) / 100:.1f} us")
    for lo, hi in ((0, 256), (256, 4096)):
        for x in range(8):
            v = [((s - t0) / 100, (e - t0) / 100) for b, (s, e) in t.items() if lo <= b < hi and b % 8 == x]
            if v:
                print(f"  wg [{lo},{hi}) xcd {x}: n {len(v):3d}  start med {st.median(a for a, _ in v):5.1f}  "
                      f"finish min/med/max {min(e for _, e in v):5.1f} {st.median(e for _, e in v):5.1f} {max(e for _, e in v):5.1f}")

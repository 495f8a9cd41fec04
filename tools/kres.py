#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in a device assembly file (hipcc -save-temps) or object.
usage: tools/kres.py file.s [name filter]"""
import re
import subprocess
import sys


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    txt = open(path, errors="replace").read()
    for blk in txt.split("  - .agpr_count:")[1:]:
        def f(key):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        name = f("name")
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        except Exception:
            pass
        if flt and flt not in name:
            continue
        agpr = blk.split("\n")[0].strip()
        print(f"{name:60s} vgpr {f('vgpr_count'):>4s} agpr {agpr:>4s} vspill {f('vgpr_spill_count'):>4s} sgpr {f('sgpr_count'):>4s} "
              f"sspill {f('sgpr_spill_count'):>4s} lds {f('group_segment_fixed_size'):>7s} scratch {f('private_segment_fixed_size'):>6s}")


if __name__ == "__main__":
    main()

"""Builds profiles/pass_kernel_pmc.json from the --pmc summaries that tools/profile_round.sh wrote
(profiles/r02_*_pmc_{fetch,write}_size.txt: per-kernel avg/min/max of FETCH_SIZE / WRITE_SIZE in KB).
bench.py reads `traffic` from it.   usage: python3 tools/pmc_record.py [dir with the r02_* files, default profiles/]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
LINE = re.compile(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)\s+min=\s*([\d.]+)\s+max=\s*([\d.]+)")


def maxima(path):
    out = {}
    for ln in open(path):
        m = LINE.match(ln)
        if m:
            out[m.group(1).strip()] = float(m.group(6))
    return out


def pick(d, *needles):
    for name, v in d.items():
        if all(n in name for n in needles):
            return v
    return None


def kb(fetch, write):
    """bytes: FETCH_SIZE x 2 (gfx950 correction, see `correction`) + WRITE_SIZE, KB of 1024 bytes"""
    return int(round((2.0 * fetch + write) * 1024))


N = 1_000_000
records = []
f, w = maxima(f"{D}/r02_k8_pmc_fetch_size.txt"), maxima(f"{D}/r02_k8_pmc_write_size.txt")
fs, ws = pick(f, "ts_schedule<8"), pick(w, "ts_schedule<8")
if fs is not None:
    upd = 200  # tools/profile_round.sh: the largest launch of the counter runs is the 200-update schedule
    records.append({
        "mode": "schedule", "kernel": "ts_schedule<8,false,0>", "n": N, "k": 8, "n_gpus": 1, "updates_in_launch": upd,
        "FETCH_SIZE_KB_max": fs, "WRITE_SIZE_KB_max": ws,
        "hbm_bytes_per_launch": kb(fs, ws), "hbm_bytes_per_update": kb(fs, ws) / upd,
        "note": ("per update the kernel streams the gamma rows and c_n of every second item (K = 8: the other half stays in "
                 "LDS) and one 2-bit column: 16NK/2 + 8N/2 + N/4 = 68.25 MB algorithmic at N = 1M; the WRITE half (35.8 MB) "
                 "arrives at the fabric in full, of the READ half only what misses the XCDs' L2s (the rows a workgroup wrote "
                 "90 us earlier are partly still there: 8 x 4 MB of L2 against 33.5 MB of streamed gamma)"),
        "source_files": ["profiles/r02_k8_pmc_fetch_size.txt", "profiles/r02_k8_pmc_write_size.txt", "profiles/r02_k8_kernel_trace.txt"],
    })
f, w = maxima(f"{D}/r02_k8_per_snp_pmc_fetch_size.txt"), maxima(f"{D}/r02_k8_per_snp_pmc_write_size.txt")
if pick(f, "ts_resident<8>") is not None:
    records.append({
        "mode": "snp", "kernel": "ts_resident<8> / ts_pass<8,true,256,1>", "n": N, "k": 8, "n_gpus": 1,
        "FETCH_SIZE_KB_max": pick(f, "ts_resident<8>"), "WRITE_SIZE_KB_max": pick(w, "ts_resident<8>"),
        "first_pass_FETCH_SIZE_KB_max": pick(f, "ts_pass<8, true"), "first_pass_WRITE_SIZE_KB_max": pick(w, "ts_pass<8, true"),
        "hbm_bytes_per_launch": kb(pick(f, "ts_resident<8>"), pick(w, "ts_resident<8>")),
        "first_pass_hbm_bytes_per_launch": kb(pick(f, "ts_pass<8, true"), pick(w, "ts_pass<8, true")),
        "algorithmic_bytes_per_launch": 9 * 64_250_000, "first_pass_algorithmic_bytes_per_launch": 264_500_000,
        "note": "TSAMD_PERSISTENT=0: the resident kernel reads the weights once per SNP (64 MB) for its 9 passes",
        "source_files": ["profiles/r02_k8_per_snp_pmc_fetch_size.txt", "profiles/r02_k8_per_snp_pmc_write_size.txt",
                         "profiles/r02_k8_per_snp_kernel_trace.txt"],
    })
f, w = maxima(f"{D}/r02_k20_pmc_fetch_size.txt"), maxima(f"{D}/r02_k20_pmc_write_size.txt")
if pick(f, "ts_pass<20, false") is not None:
    records.append({
        "mode": "pass", "kernel": "ts_pass<20,false,256,2> / ts_pass<20,true,256,1>", "n": N, "k": 20, "n_gpus": 1,
        "FETCH_SIZE_KB_max": pick(f, "ts_pass<20, false"), "WRITE_SIZE_KB_max": pick(w, "ts_pass<20, false"),
        "first_pass_FETCH_SIZE_KB_max": pick(f, "ts_pass<20, true"), "first_pass_WRITE_SIZE_KB_max": pick(w, "ts_pass<20, true"),
        "hbm_bytes_per_launch": kb(pick(f, "ts_pass<20, false"), pick(w, "ts_pass<20, false")),
        "first_pass_hbm_bytes_per_launch": kb(pick(f, "ts_pass<20, true"), pick(w, "ts_pass<20, true")),
        "algorithmic_bytes_per_launch": 160_250_000, "first_pass_algorithmic_bytes_per_launch": 648_500_000,
        "source_files": ["profiles/r02_k20_pmc_fetch_size.txt", "profiles/r02_k20_pmc_write_size.txt", "profiles/r02_k20_kernel_trace.txt"],
    })
out = {
    "records": records,
    "correction": ("gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM "
                   "section); calibrated in round 1 on ts_refresh_w, which reads exactly 8*1000448*8 B = 64.03 MB and reports "
                   "31339.4 KB (x2 = 64.2 MB) and writes the same amount, reported as 62528 KB WRITE_SIZE = 64.03 MB (no correction)."),
    "collected": ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (tools/pmc.sh via "
                  "tools/profile_round.sh), per-launch MAXIMUM over the launches (the sequences contain launches that only carry "
                  "state forward, and ts_schedule launches of different lengths; the maximum is a full launch)"),
    "note": ("the counters sit on the fabric side of the L2s and include Infinity Cache hits: they say what was (not) re-read, "
             "not that the bytes came from DRAM"),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "pass_kernel_pmc.json"), "w"), indent=1)
print(json.dumps([{k: r[k] for k in r if "bytes" in k or k in ("mode", "k")} for r in records], indent=1))

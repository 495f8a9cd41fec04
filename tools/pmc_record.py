"""Builds profiles/pass_kernel_pmc.json from the --pmc summaries that tools/profile_round3.sh wrote
(profiles/{RND}_*_pmc_*.txt: per-kernel avg/min/max of each counter; FETCH_SIZE / WRITE_SIZE in KB) and the in-kernel
timers of the diagnostic build (profiles/{RND}_sched_timers.txt).  bench.py reads `traffic`, the fp64 flops per update
and the exchange time per update from it.   usage: python3 tools/pmc_record.py [dir with the {RND}_* files, default profiles/]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
RND = sys.argv[2] if len(sys.argv) > 2 else "r04"   # file prefix of the round whose summaries are read
LINE = re.compile(r"^(.*?)\s+([A-Z][A-Z0-9_]+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)\s+min=\s*([\d.]+)\s+max=\s*([\d.]+)")
UPD = 200  # tools/profile_round3.sh: the largest launch of the counter runs is the 200-update schedule


def maxima(name):
    """{(kernel, counter): (max over launches, entries per launch group)}"""
    out = {}
    path = os.path.join(D, name)
    if not os.path.exists(path):
        return out
    for ln in open(path):
        m = LINE.match(ln)
        if m:
            out[(m.group(1).strip(), m.group(2))] = float(m.group(6))
    return out


def pick(d, counter, *needles):
    for (name, ctr), v in d.items():
        if ctr == counter and all(n in name for n in needles):
            return v
    return None


def kb(fetch, write):
    """bytes: FETCH_SIZE x 2 (gfx950 correction, see `correction`) + WRITE_SIZE, KB of 1024 bytes"""
    return int(round((2.0 * fetch + write) * 1024))


def timers(tag):
    """exchange / gamma / sweep microseconds per SNP from the diagnostic build's line for `tag` (e.g. 'N=1000000 K=8')"""
    path = os.path.join(D, f"{RND}_sched_timers.txt")
    if not os.path.exists(path):
        return None
    for ln in open(path):
        if ln.startswith(tag + ":"):
            m = re.search(r"gamma ([\d.]+) first pass ([\d.]+) later passes ([\d.]+).*in exchanges ([\d.]+), in folds ([\d.]+), "
                          r"in epilogues ([\d.]+), in sweeps ([\d.]+) \| whole launch ([\d.]+) us", ln)
            n = re.search(r"n=(\d+) exchanges=(\d+)", ln)
            if m and n:
                nn = int(n.group(1))
                return {"gamma_us": float(m.group(1)), "exchange_us": float(m.group(4)), "fold_us": float(m.group(5)),
                        "epilogue_us": float(m.group(6)), "sweep_us": float(m.group(7)), "exchanges": int(n.group(2)) / nn,
                        "launch_us_per_snp": float(m.group(8)) / nn}
    return None


def schedule_record(prefix, n, k, kernel_needle):
    f, w, c = maxima(f"{RND}_{prefix}_pmc_fetch_size.txt"), maxima(f"{RND}_{prefix}_pmc_write_size.txt"), maxima(f"{RND}_{prefix}_pmc_f64.txt")
    fs, ws = pick(f, "FETCH_SIZE", kernel_needle), pick(w, "WRITE_SIZE", kernel_needle)
    if fs is None or ws is None:
        return None
    rec = {
        "mode": "schedule", "kernel": kernel_needle.strip("<") + ">", "n": n, "k": k, "n_gpus": 1, "updates_in_launch": UPD,
        "FETCH_SIZE_KB_max": fs, "WRITE_SIZE_KB_max": ws,
        "hbm_bytes_per_launch": kb(fs, ws), "hbm_bytes_per_update": kb(fs, ws) / UPD,
        "source_files": [f"profiles/{RND}_{prefix}_pmc_fetch_size.txt", f"profiles/{RND}_{prefix}_pmc_write_size.txt"],
    }
    fma, mul, add, trn = (pick(c, "SQ_INSTS_VALU_" + x + "_F64", kernel_needle) for x in ("FMA", "MUL", "ADD", "TRANS"))
    if None not in (fma, mul, add, trn):
        # SQ_INSTS_VALU_* count wave-level instructions per shader-engine entry (the summary's max is the 200-update launch
        # of ONE of the 32 entries, which cover all workgroups evenly): x 32 entries x 64 lanes; an FMA is two flops
        scale = 32.0 * 64.0 / UPD
        rec.update({
            "SQ_INSTS_VALU_FMA_F64_max": fma, "SQ_INSTS_VALU_MUL_F64_max": mul, "SQ_INSTS_VALU_ADD_F64_max": add,
            "SQ_INSTS_VALU_TRANS_F64_max": trn,
            "fp64_flops_per_update": (2.0 * fma + mul + add + trn) * scale,
            "fp64_instructions_per_wave_and_update": (fma + mul + add + trn) / UPD / 32.0,  # (an entry covers the 32 waves of 8 CUs)
            "flops_source_files": [f"profiles/{RND}_{prefix}_pmc_f64.txt"],
            "flops_note": ("(2 FMA + MUL + ADD + TRANS) wave instructions x 64 lanes, summed over the 32 shader-engine entries of the "
                           "200-update launch, / 200; lanes of partially filled waves and the transcendental estimates count as "
                           "one flop per lane"),
        })
    tm = None if "ts_hybrid" in kernel_needle else timers(f"N={n} K={k}")
    if tm:
        rec.update({"exchange_us_per_update": tm["exchange_us"], "exchanges_per_update": tm["exchanges"],
                    "gamma_us_per_update": tm["gamma_us"], "sweep_us_per_update": tm["sweep_us"],
                    "epilogue_us_per_update": tm["epilogue_us"],
                    "exchange_source": ("in-kernel timers of the diagnostic build (-DTSAMD_SCHED_TIME, workgroup 0, 2 000-SNP launch; "
                                        f"the timers themselves cost about 2 us per SNP): profiles/{RND}_sched_timers.txt")})
    return rec


records = []
for prefix, n, k, needle in (("k8", 1_000_000, 8, "ts_schedule<8"), ("k16_n500k", 500_000, 16, "ts_schedule<16"),
                             ("k20_n125k", 125_000, 20, "ts_schedule<20"),
                             ("k20_n1m", 1_000_000, 20, "ts_hybrid<20")):   # (round 4: config 5 on one GPU runs ts_hybrid)
    r = schedule_record(prefix, n, k, needle)
    if r:
        records.append(r)

N = 1_000_000
f, w = maxima(f"{RND}_k8_per_snp_pmc_fetch_size.txt"), maxima(f"{RND}_k8_per_snp_pmc_write_size.txt")
if pick(f, "FETCH_SIZE", "ts_resident<8>") is not None:
    records.append({
        "mode": "snp", "kernel": "ts_resident<8> / ts_pass<8,true,256,1>", "n": N, "k": 8, "n_gpus": 1,
        "FETCH_SIZE_KB_max": pick(f, "FETCH_SIZE", "ts_resident<8>"), "WRITE_SIZE_KB_max": pick(w, "WRITE_SIZE", "ts_resident<8>"),
        "first_pass_FETCH_SIZE_KB_max": pick(f, "FETCH_SIZE", "ts_pass<8, true"),
        "first_pass_WRITE_SIZE_KB_max": pick(w, "WRITE_SIZE", "ts_pass<8, true"),
        "hbm_bytes_per_launch": kb(pick(f, "FETCH_SIZE", "ts_resident<8>"), pick(w, "WRITE_SIZE", "ts_resident<8>")),
        "first_pass_hbm_bytes_per_launch": kb(pick(f, "FETCH_SIZE", "ts_pass<8, true"), pick(w, "WRITE_SIZE", "ts_pass<8, true")),
        "algorithmic_bytes_per_launch": 64_250_000, "first_pass_algorithmic_bytes_per_launch": 264_500_000,
        "note": "TSAMD_PERSISTENT=0: the resident kernel reads the weights once per SNP (64 MB) for its 9 passes",
        "source_files": [f"profiles/{RND}_k8_per_snp_pmc_fetch_size.txt", f"profiles/{RND}_k8_per_snp_pmc_write_size.txt"],
    })
f, w = maxima(f"{RND}_k20_n1m_pmc_fetch_size.txt"), maxima(f"{RND}_k20_n1m_pmc_write_size.txt")
if pick(f, "FETCH_SIZE", "ts_pass<20, false") is not None:
    records.append({
        "mode": "pass", "kernel": "ts_pass<20,false,256,2> / ts_pass<20,true,256,1>", "n": N, "k": 20, "n_gpus": 1,
        "FETCH_SIZE_KB_max": pick(f, "FETCH_SIZE", "ts_pass<20, false"), "WRITE_SIZE_KB_max": pick(w, "WRITE_SIZE", "ts_pass<20, false"),
        "first_pass_FETCH_SIZE_KB_max": pick(f, "FETCH_SIZE", "ts_pass<20, true"),
        "first_pass_WRITE_SIZE_KB_max": pick(w, "WRITE_SIZE", "ts_pass<20, true"),
        "hbm_bytes_per_launch": kb(pick(f, "FETCH_SIZE", "ts_pass<20, false"), pick(w, "WRITE_SIZE", "ts_pass<20, false")),
        "first_pass_hbm_bytes_per_launch": kb(pick(f, "FETCH_SIZE", "ts_pass<20, true"), pick(w, "WRITE_SIZE", "ts_pass<20, true")),
        "algorithmic_bytes_per_launch": 160_250_000, "first_pass_algorithmic_bytes_per_launch": 648_500_000,
        "source_files": [f"profiles/{RND}_k20_n1m_pmc_fetch_size.txt", f"profiles/{RND}_k20_n1m_pmc_write_size.txt"],
    })
sys.path.insert(0, ROOT)
from terastructure_amd.build import kernel_sources_sha  # noqa: E402

out = {
    # the device sources these counters were collected from (csrc/tsamd_device.h, tsamd_kernels.h, tsamd_resident_kernels.h):
    # bench.py ignores the records -- and says so -- when the tree's sources no longer hash to this
    "kernel_sources_sha": kernel_sources_sha(),
    "records": records,
    "correction": ("gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM "
                   "section); calibrated in round 1 on ts_refresh_w, which reads exactly 8*1000448*8 B = 64.03 MB and reports "
                   "31339.4 KB (x2 = 64.2 MB) and writes the same amount, reported as 62528 KB WRITE_SIZE = 64.03 MB (no correction)."),
    "collected": ("rocprofv3 --kernel-trace --pmc <counters> in separate runs per counter group (tools/pmc.sh via "
                  "tools/profile_round3.sh), per-launch MAXIMUM over the launches (the sequences contain launches that only carry "
                  "state forward, and ts_schedule launches of different lengths; the maximum is a full launch)"),
    "note": ("the FETCH/WRITE counters sit on the fabric side of the L2s and include Infinity Cache hits: they say what was (not) "
             "re-read, not that the bytes came from DRAM"),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "pass_kernel_pmc.json"), "w"), indent=1)
print(json.dumps([{k: r[k] for k in r if "bytes" in k or "flops_per" in k or "exchange_us" in k or k in ("mode", "k", "n")} for r in records], indent=1))

"""bench.py prints ONE JSON line with the keys the driver reads (small workload here; the
default workload is the N=1M, L=1M, K=8 configuration)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.spawns]


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "60", "--warmup", "20",
                        "--individuals", "30000", "--snps", "1000", "--pops", "8", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
        assert isinstance(d[key], typ), key
    assert d["steps"] == 60 and d["warmup"] == 20 and d["n_gpus"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["unit"] == "updates/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-3          # value = steps / time
    rf = d["roofline"]
    # one GPU, K = 8: the whole schedule is one launch of ts_schedule, bound by fp64 vector issue (one wave per SIMD);
    # its memory side and the reference-dataflow equivalent are secondary objects; the kernels of the launch-per-SNP
    # sequence are timed beside it
    assert rf["bound"] == "fp64_valu" and rf["unit"] == "TFLOP/s" and rf["peak"] == 78.6
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf and 0.0 < rf["frac"] < 1.0
    assert "N=30000 K=8" in d["metric"]                                       # names the workload that ran
    assert "ts_schedule" in rf["kernel"] and rf["updates_per_launch"] == 60 and rf["launches_timed"] == 1
    assert rf["flops_per_update"] > 0 and "algorithmic" in rf["flops_source"] and "flops_per_update" in rf["executed"]
    hb = rf["hbm"]
    assert hb["bound"] == "hbm" and hb["peak"] == 8000.0 and hb["unit"] == "GB/s" and 0.0 < hb["moved_frac"] < 1.0
    # (K = 8: gamma and c_n of 9 of a thread's 16 individuals stay in LDS)
    assert hb["moved_bytes_per_update"] == (16.0 * 30000 * 8 + 8.0 * 30000) * 7 / 16 + 30000 / 4.0
    eq = rf["algorithmic_bandwidth_equiv"]
    assert eq["bytes_per_update"] > hb["moved_bytes_per_update"] and "frac" not in eq
    assert rf["latency"]["exchanges_per_update"] > 0
    assert "update_hbm_frac_of_peak" not in d
    assert abs(rf["per_update_us"] * rf["updates_per_launch"] - rf["avg_launch_us"]) < 0.1 * rf["avg_launch_us"]
    ps = rf["launch_per_snp"]
    assert "ts_resident" in ps["kernel"] and ps["probe_read_us"] > 0 and "tsamd_probe_stream" in ps["ceiling_note"]
    assert ps["passes_per_launch"] > 1.0 and abs(ps["frac"] - ps["achieved"] / ps["peak"]) < 1e-3
    fp = rf["first_pass"]
    assert fp["bound"] == "hbm" and abs(fp["frac"] - fp["achieved"] / fp["peak"]) < 1e-3 and fp["probe_rmw_us"] > 0
    assert fp["algorithmic_bytes_per_launch"] == 32.0 * 30000 * 8 + 8.0 * 30000 + 30000 / 2.0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] and cb["sample"]
    assert cb["value_1_thread"] > 0
    pv = d["parity_vs_cpu_baseline"]                                         # GPU vs the oracle on the timed updates ...
    assert pv["ok"] and pv["c_n_equal"] and pv["lambda_rel_err"] < 1e-9 and pv["gamma_rel_err"] < 1e-9
    om = pv["other_launch_modes"]                                            # ... in every launch mode the line publishes timings of
    assert set(om) == {"launch_per_snp", "launch_per_pass"} and all(o["ok"] and o["c_n_equal"] for o in om.values())
    assert pv["kernels_per_snp"] == 0 and om["launch_per_snp"]["kernels_per_snp"] == 2 and om["launch_per_pass"]["kernels_per_snp"] == 10
    assert sum(d["inner_passes_histogram"].values()) == 60
    assert isinstance(rf["counter_records"], str) and rf["counter_records"]      # fresh, absent for this shape, or stale (then: no `executed` figures)
    vb = d["validation_block"]                                                # floor(0.005 L) = 5 locations x N / 100 held-out individuals
    assert vb["locations"] == 5 and vb["heldout_per_location"] == 300 and vb["kernel"].startswith("ts_holblock<8>: 16 locations")
    assert vb["seconds_per_report"] > 0 and vb["entry_by_entry_seconds_per_report"] > 0 and vb["evaluation_only_seconds"] >= 0
    assert 0 < vb["heldout_entries"] <= 5 * 300 and vb["mean_loglik"] < 0
    assert all(v == "ok" for v in d["legs"].values()), d["legs"]              # every secondary leg ran (each in its own try)
    assert {"roofline_timed_kernel", "roofline_other_launch_modes", "cpu_baseline", "parity_vs_cpu_baseline", "validation_block"} <= set(d["legs"])
    assert d["recoveries"] == 0                                               # (the timed kernel is the one the line names: nothing was replayed)


def test_bench_short_run_matches_long_run():
    """The driver times `--steps 20 --warmup 5`: graphs are built before the timed region and a
    schedule of any length replays without padding, so the short run's per-update time is the long
    run's (small workload here, where fixed costs weigh most: within 25 %)."""
    vals = {}
    for steps, warm in ((20, 5), (600, 50)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps),
                            "--warmup", str(warm), "--individuals", "200000", "--snps", "300", "--pops", "8",
                            "--cpu-seconds", "0", "--no-profile"], capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        vals[steps] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["value"]
    assert vals[20] > 0.75 * vals[600], vals


def test_bench_two_ranks_json_contract():
    """`bench.py --gpus 2` launched exactly as the driver launches it (torch.distributed.run, one process per rank), both
    ranks on device 0 (TSAMD_BENCH_DEVICE=0: a functional rehearsal, the rate means nothing): the line for N > 1 is
    complete -- the timed kernel's roofline from the timed mode alone, cpu_baseline and parity_vs_cpu_baseline from rank 0
    with the shards' states gathered, the validation block summed over the ranks, every leg's verdict, and what RCCL said
    when it refused two ranks on one device."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, TSAMD_BENCH_DEVICE="0", GPU_MAX_HW_QUEUES="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TSAMD_LIB", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--individuals", "60000", "--pops", "8",
                        "--snps", "2000", "--steps", "40", "--warmup", "10", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 40 and d["warmup"] == 10 and d["scaling"] == "strong" and d["dtype"] == "f64"
    assert d["unit"] == "updates/s" and d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-3
    st = d["config"]["exchange_selftest"]
    assert d["config"]["exchange"] == st["chosen"] and st["valid"][st["chosen"]] is True
    assert st["valid"].get("p2p") is True and st["valid"].get("p2p_schedule") is True      # both peer-to-peer forms match the oracle
    assert st["rel_err_vs_oracle"]["p2p_schedule"] < 1e-9
    if not st["valid"].get("rccl"):                                 # (two ranks on one device: RCCL refuses -- its words are in the line)
        assert isinstance(st.get("rccl_error"), str) and st["rccl_error"]
    legs = d["legs"]
    assert all(v == "ok" for v in legs.values()), legs
    rf = d["roofline"]                                              # the timed kernel, timed in the timed mode
    assert rf is not None and rf["avg_launch_us"] > 0 and 0.0 < rf["frac"] < 1.0
    if d["config"]["exchange"].startswith("p2p_schedule"):
        assert "ts_schedule<8>" in rf["kernel"] and rf["bound"] == "fp64_valu" and rf["launches_timed"] == 1 and rf["updates_per_launch"] == 40
        assert rf["launch_per_snp"] is None                         # (no mode switch on a sharded context)
    else:
        assert rf["bound"] == "hbm" and "ts_pass<8" in rf["kernel"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    pv = d["parity_vs_cpu_baseline"]
    assert pv["ok"] and pv["c_n_equal"] and pv["lambda_rel_err"] < 1e-9 and pv["gamma_rel_err"] < 1e-9
    vb = d["validation_block"]
    assert vb["locations"] == 10 and vb["heldout_per_location"] == 600 and 0 < vb["heldout_entries"] <= 6000 and vb["mean_loglik"] < 0
    assert sum(d["inner_passes_histogram"].values()) == 40
    # round 6: an N > 1 line says what it is to be read against (no one-GPU rate is on record for this shape: null)
    assert "predicted_1gpu_equiv" in d and d["predicted_1gpu_equiv"] is None and "dependent exchanges" in d["scaling_note"]
    assert d["recoveries"] == 0                                    # (no launch of the timed context was replayed)

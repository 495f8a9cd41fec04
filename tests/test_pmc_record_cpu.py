"""The committed counter records (profiles/pass_kernel_pmc.json) carry a hash of the device sources they were
collected from; bench.py uses their traffic / flops / latency figures only while the tree's sources still hash to it
(a kernel change without re-profiling must not silently skew roofline.frac: the counter-based figures are dropped and the line says so)."""
import json
import os
import shutil

from conftest import ROOT


def test_kernel_sources_sha_follows_the_sources(tmp_path, monkeypatch):
    from terastructure_amd import build

    a = build.kernel_sources_sha()
    assert len(a) == 16 and a == build.kernel_sources_sha()
    csrc = tmp_path / "csrc"
    shutil.copytree(build.CSRC, csrc)
    monkeypatch.setattr(build, "CSRC", str(csrc))
    assert build.kernel_sources_sha() == a
    with open(csrc / "tsamd_device.h", "a") as f:
        f.write("// touched\n")
    assert build.kernel_sources_sha() != a


def test_committed_records_carry_a_hash():
    doc = json.load(open(os.path.join(ROOT, "profiles", "pass_kernel_pmc.json")))
    assert isinstance(doc.get("kernel_sources_sha"), str) and len(doc["kernel_sources_sha"]) == 16
    assert doc["records"], "no counter records"
    for rec in doc["records"]:
        assert {"mode", "n", "k", "n_gpus"} <= set(rec)


def test_roofline_frac_is_algorithmic_and_the_counters_stand_beside_it():
    """roofline.achieved / frac price the FORMULATION's flops of the run's own pass count; what the counters of a profiled
    launch saw (wave instructions x 64 lanes: the per-wave epilogues at full width) is reported as `executed`, and only
    while a record for the shape is committed -- the fraction itself never depends on the record."""
    import sys
    sys.path.insert(0, ROOT)
    import bench

    n, k, nsteps, ran = 1_000_000, 8, 2000, 20_000                # ten passes per update
    prs = {"pass_launches": 1, "pass_ms": 144.8, "first_launches": 0, "first_ms": 0.0}
    geo = {"workgroups": 256, "indivs_per_thread": 16, "on_chip_per_thread": 16}
    hand = 10.0 * n * (8 * k + 12) + n * (92 * k + 25)            # (the full-size K <= 8 instantiation keeps the literal step)
    rec = {"fp64_flops_per_update": 1.699e9, "flops_source_files": ["profiles/r05_k8_pmc_f64.txt"], "hbm_bytes_per_update": 43.0e6,
           "exchange_us_per_update": 23.9}
    with_rec = bench.schedule_roofline(prs, ran, nsteps, rec, [None], k, n, geo, None)
    without = bench.schedule_roofline(prs, ran, nsteps, {}, ["stale: other sources"], k, n, geo, None)
    for rf in (with_rec, without):
        assert rf["bound"] == "fp64_valu" and rf["flops_per_update"] == hand and "algorithmic" in rf["flops_source"]
        assert abs(rf["frac"] - hand * nsteps / 0.1448 / 1e12 / 78.6) < 1e-3 and abs(rf["per_update_us"] - 72.4) < 1e-9
    ex = with_rec["executed"]
    assert ex["flops_per_update"] == 1.699e9 and ex["frac"] > with_rec["frac"] and "r05_k8_pmc_f64" in ex["source"]
    assert abs(ex["frac"] - 1.699e9 * nsteps / 0.1448 / 1e12 / 78.6) < 1e-3
    assert without["executed"]["flops_per_update"] is None and without["executed"]["frac"] is None
    assert without["executed"]["source"] == "stale: other sources" and without["traffic"] is None

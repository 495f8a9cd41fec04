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

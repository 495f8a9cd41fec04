"""Register budget of the built kernels, from the code-object metadata (terastructure_amd/resources.py) -- not from a
hand-copied tools/kcompile.sh run.  The whole-launch kernels hold a shard's weights in the register file for a whole
schedule: an instantiation that spills to scratch re-reads its spills ten times per SNP.  Round 5's tree carried such
instantiations unnoticed (ts_schedule<8, ., 32> -- what BASELINE config 4 runs on 8 GPUs -- 20-28 bytes,
ts_holblock<8, 32> 92 bytes) while DESIGN.md said "no scratch anywhere"."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rows():
    from terastructure_amd import build, resources

    build.build()
    return resources.table()


def test_no_whole_launch_kernel_the_launchers_can_select_uses_scratch(rows):
    """every instantiation launch_schedule_k / launch_holblock_k / launch_hybrid_k / launch_hybhol_k can select
    (csrc/tsamd_sched.hip, tsamd_hol.hip, tsamd_hyb.hip, tsamd_hhol.hip): K = 1 ... 32, every PARTIAL / WR / STREAM form"""
    fams = ("ts_schedule<", "ts_holblock<", "ts_hybrid<", "ts_hybhol<")
    sel = [r for r in rows if r["name"].startswith(fams)]
    count = {f: sum(1 for r in sel if r["name"].startswith(f)) for f in fams}
    # ts_schedule: K <= 8 with and without the skip branches x WR in {0, 8, 16, 32}, above only with them; ts_holblock: WR in {0, 8, 32};
    # ts_hybrid: WR in {0, 8, 16} x with / without streamed items; ts_hybhol: WR in {0, 8, 16}
    assert count == {"ts_schedule<": 8 * 8 + 24 * 4, "ts_holblock<": 32 * 3, "ts_hybrid<": 32 * 6, "ts_hybhol<": 32 * 3}, count
    bad = [(r["name"], r["private_segment_fixed_size"]) for r in sel if r["private_segment_fixed_size"] != 0]
    assert not bad, f"whole-launch kernels with scratch: {bad}"
    for r in sel:  # one wave per SIMD: the whole 512-entry file, nothing beyond it
        assert r["vgpr_count"] <= 512 and r["group_segment_fixed_size"] <= 160 * 1024, r
    # spills that stay in the register file (VGPR -> AGPR copies) are allowed but bounded: today at most 6 per kernel
    assert max(r["vgpr_spill_count"] for r in sel) <= 8, sorted((r["vgpr_spill_count"], r["name"]) for r in sel)[-5:]


def test_known_scratch_users_are_exactly_the_documented_ones(rows):
    """What still touches scratch is stated, not discovered: ts_resident<K> (launch mode "per SNP", not the default) at
    K = 14, 24, 32 and launch-per-pass kernels in geometries the host never selects or at wide K."""
    res = sorted(int(re.search(r"<(\d+)", r["name"]).group(1)) for r in rows
                 if r["name"].startswith("ts_resident<") and r["private_segment_fixed_size"] > 0)
    assert res == [14, 24, 32], res
    for r in rows:
        if r["name"].startswith("ts_resident<"):
            assert r["private_segment_fixed_size"] <= 36, r
    # the launch-per-pass kernels the default geometry runs (configure_launch, csrc/tsamd.hip): first pass ts_pass<K, true, 256, 1>,
    # plain pass ts_pass<K, false, 512, 2> up to K = 16 and <K, false, 256, 2> above -- scratch-free up to K = 22 (first pass) / 29
    for r in rows:
        m = re.match(r"ts_pass<(\d+), (true|false), (\d+), (\d+)>", r["name"])
        if not m:
            continue
        k, first, block, vec = int(m.group(1)), m.group(2) == "true", int(m.group(3)), int(m.group(4))
        default = (first and block == 256 and vec == 1) or (not first and vec == 2 and block == (512 if k <= 16 else 256))
        if default and ((first and k <= 22) or (not first and k <= 29)):
            assert r["private_segment_fixed_size"] == 0, r


def test_committed_resource_table_is_the_builds(rows):
    """profiles/r06_kernel_resources.txt (what DESIGN.md quotes) was generated from these kernel sources"""
    from terastructure_amd import build, resources

    path = os.path.join(ROOT, "profiles", "r06_kernel_resources.txt")
    text = open(path).read()
    m = re.search(r"kernel sources sha (\w+)", text)
    assert m and m.group(1) == build.kernel_sources_sha(), "regenerate: python -m terastructure_amd.resources --write"
    assert resources.format_table(rows) in text

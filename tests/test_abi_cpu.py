"""No-GPU checks of the C-ABI library: it builds, loads and exports every
symbol include/tsamd.h declares; without a GPU it refuses to run (no CPU path)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def ts():
    import terastructure_amd as t
    from terastructure_amd import build

    build.build()
    return t


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "tsamd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(tsamd_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_all_exported(ts):
    lib = C.CDLL(ts.lib_path())
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/tsamd.h but not exported"
    # and the binding covers exactly the header
    from terastructure_amd import _lib

    assert sorted(_lib.SYMBOLS) == names


def test_header_compiles_as_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "tsamd.h"\nint main(void){tsamd_config c; tsamd_default_config(&c,1,1,1); return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-c", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(tmp_path / "t.o")])


def test_shard_range_properties(ts):
    for n in (1, 3, 200, 1003, 10_000, 1_000_000):
        for world in (1, 2, 3, 4, 8):
            covered = 0
            for r in range(world):
                b, c = ts.shard_range(n, r, world)
                assert b % 4 == 0 or c == 0          # byte-aligned slices of a .bed column
                assert b == min(covered, n) or c == 0
                covered = max(covered, b + c)
            assert covered == n
    assert ts.shard_range(1_000_000, 7, 8) == (875_000, 125_000)


def test_default_config_matches_reference_constants(ts):
    from terastructure_amd import _lib

    h = ts.load()
    cfg = _lib.Config()
    h.tsamd_default_config(C.byref(cfg), 200, 10000, 3)
    # src/env.hh:200-249, src/snpsamplinge.cc:16, :702
    assert (cfg.n, cfg.l, cfg.k) == (200, 10000, 3)
    assert cfg.alpha == 1.0 / 3 and cfg.eta0 == 1.0 and cfg.eta1 == 1.0
    assert cfg.nodetau0 == 2.0 and cfg.nodekappa == 0.5
    assert cfg.max_inner == 10 and cfg.conv_thresh == 1e-3 and cfg.gamma_scale == 10000.0
    assert cfg.struct_size == C.sizeof(_lib.Config)


def test_no_gpu_fails_loudly(ts):
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(ts.TsamdError) as ei:
        ts.Engine(100, 10, 3)
    assert "no CPU path" in str(ei.value)
    # bad config is rejected before any device work
    with pytest.raises(ts.TsamdError):
        ts.Engine(100, 10, 129)


def test_resident_geometry_mirrors_agree(tmp_path):
    """The resident kernels' geometry (individuals per thread, items whose gamma stays in LDS, capacity per GPU) is stated in
    csrc/tsamd_resident_kernels.h and mirrored by bench.py (bytes the kernel must move) and include/tsamd.h (capacity
    table): compile the header's constexpr functions for the host and compare."""
    import sys

    src = tmp_path / "geom.hip"
    src.write_text('#include <cstdio>\n#include "tsamd_resident_kernels.h"\nint main() {\n'
                   '  for (int k = 1; k <= 32; ++k)\n'
                   '    printf("%d %d %d %d %d\\n", k, tsamd::resident_vec(k), tsamd::resident_items(k),\n'
                   '           tsamd::sched_lds_items(k, tsamd::resident_items(k), tsamd::resident_vec(k)), tsamd::resident_capacity(k));\n'
                   '  return 0;\n}\n')
    exe = tmp_path / "geom"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "terastructure_amd", "csrc"), str(src), "-o", str(exe)])
    rows = [tuple(int(x) for x in ln.split()) for ln in subprocess.check_output([str(exe)], text=True).splitlines()]
    sys.path.insert(0, ROOT)
    import bench

    for k, vec, items, lds, cap in rows:
        assert bench.resident_geometry(k) == (vec, items, lds), k
        assert cap == 256 * items * vec and 0 < lds <= items
        assert items * k <= 128                                     # at most 256 registers of weights per thread
    by_k = {r[0]: r for r in rows}
    # the capacity table of include/tsamd.h (tsamd_launch_info) and DESIGN.md section 4
    assert by_k[8][4] * 256 == 1_048_576 and by_k[16][4] * 256 == 524_288 and by_k[20][4] * 256 == 327_680
    assert by_k[24][4] * 256 == 262_144 and by_k[32][4] * 256 == 196_608 and by_k[9][4] * 256 == 917_504

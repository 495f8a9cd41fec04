"""The host front end's text writers (a16: save_gamma / save_beta, src/snpsamplinge.cc:546-576, :761-798) print with
fprintf("%.8f\\t"); the MI355X host formats with host/fast_format.h on a writer thread (host/model_writer.h).  Both must
produce the reference's BYTES: tests/host_format_check.cpp compares the formatter with snprintf on > 10^7 random and
edge-case doubles (dyadic ties, neighbours of decimal ties, values >= 10^6, any bit pattern) and the writer with the
fprintf loop file against file."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_formatter_and_writer_produce_printf_bytes(tmp_path):
    exe = tmp_path / "host_format_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "host"),
                           os.path.join(HERE, "host_format_check.cpp"), "-o", str(exe), "-lpthread"])
    out = subprocess.run([str(exe), "10000000", "200003", "7", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"formatter: (\d+) values, (\d+) mismatches", out.stdout)
    assert m and int(m.group(1)) > 10_000_000 and int(m.group(2)) == 0, out.stdout
    assert "files identical" in out.stdout, out.stdout
    # the writer thread is several times faster than the loop it replaces (7.35 s per report at N = 1M, K = 8 with fprintf)
    m = re.search(r"fprintf loop ([\d.]+) s, writer thread ([\d.]+) s", out.stdout)
    assert m and float(m.group(2)) * 3 < float(m.group(1)), out.stdout

"""The C++ command-line host (host/terastructure) over libtsamd: the reference's
data/run.sh, verbatim except for the binary path, against the oracle."""
import ctypes as C
import itertools
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

import oracle_py as op
from conftest import REF_DATA, ROOT

HOST = os.path.join(ROOT, "host", "terastructure")


@pytest.fixture(scope="module")
def host_bin():
    from terastructure_amd import build

    build.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    return HOST


def test_help_and_argument_errors(host_bin, tmp_path):
    r = subprocess.run([host_bin, "-help"], capture_output=True, text=True)
    assert r.returncode == 0 and "-file <name>" in r.stdout and "-rfreq" in r.stdout
    r = subprocess.run([host_bin, "-frobnicate"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode != 0 and "unknown option" in r.stdout
    r = subprocess.run([host_bin, "-file", "x.bed", "-k", "3"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode != 0 and "-n, -l and -k are required" in r.stderr
    assert os.listdir(tmp_path) == []


def test_without_gpu_fails_before_writing(host_bin, tmp_path):
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    r = subprocess.run([host_bin, "-file", os.path.join(REF_DATA, "test.bed"), "-n", "200", "-l", "10000", "-k", "3"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode != 0 and "no CPU path" in r.stderr
    assert os.listdir(tmp_path) == []


def _read_matrix(path):
    return np.array([[float(x) for x in line.split()] for line in open(path)])


@pytest.mark.gpu
def test_run_sh_end_to_end(host_bin, tmp_path):
    data = tmp_path / "data"
    data.mkdir()
    for f in ("test.bed", "test.bim", "test.fam"):
        shutil.copy(os.path.join(REF_DATA, f), data / f)
    # data/run.sh line 1
    cmd1 = [host_bin, "-file", "test.bed", "-n", "200", "-l", "10000", "-k", "3", "-stochastic",
            "-nthreads", "1", "-rfreq", "1000", "-seed", "1234", "-label", "test"]
    r = subprocess.run(cmd1, cwd=data, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    run = data / "n200-k3-l10000-test-seed1234"
    for f in ("infer.log", "param.txt", "validation.txt", "gamma.txt", "theta.txt"):
        assert (run / f).exists(), f
    # symlink to the data file exactly as given on the command line (src/env.hh:309-311)
    assert os.path.islink(run / "network.dat") and os.readlink(run / "network.dat") == "test.bed"
    param = open(run / "param.txt").read()
    assert "validation locations: 50" in param and "validation snps per location: 20" in param
    assert "missing snps: 0" in param and "GSL seed: 1234.000000000" in param
    # an existing directory without -force is refused (src/log.cc:113-116)
    r2 = subprocess.run(cmd1, cwd=data, capture_output=True, text=True, timeout=60)
    assert r2.returncode != 0 and "already exists" in r2.stderr

    # the oracle on the same stream
    orc = op.Oracle(200, 10000, 3)
    orc.read_bed_file(os.path.join(REF_DATA, "test.bed"))
    res = orc.run(seed=1234, reportfreq=1000)
    val = [line.split("\t") for line in open(run / "validation.txt").read().splitlines()]
    assert len(val) == len(res["lines"]) == 17
    assert val[0][0] == "0" and val[0][2] == "-1.169339294" and val[0][3] == "1000"
    assert val[1][0] == "1050" and val[1][2] == "-0.732008912"
    assert val[-1][0] == "16050"
    for got, (it, ll, cnt) in zip(val, res["lines"]):
        assert int(got[0]) == it and int(got[3]) == cnt
        assert abs(float(got[2]) - ll) < 1e-8
    theta = _read_matrix(run / "theta.txt")
    gamma = _read_matrix(run / "gamma.txt")
    assert theta.shape == (200, 3)
    assert np.max(np.abs(theta - orc.theta())) <= 1e-6          # stated tolerance
    assert np.max(np.abs(gamma - orc.gamma()) / orc.gamma()) <= 1e-6
    raw = open(run / "theta.txt").read()
    assert raw.count("\n") == 200 and raw.split("\n")[0].endswith("\t")   # "%.8f\t" * K + "\n"

    # data/run.sh lines 2-3: -compute-beta from inside the run directory
    with open(run / "ids.txt", "w") as f:       # -idfile: labels echoed into gammasave.txt (src/snp.cc:255-276)
        f.write("\n".join(f"ind{i}" for i in range(150)) + "\n")
    cmd2 = [host_bin, "-file", "../test.bed", "-n", "200", "-l", "10000", "-k", "3", "-stochastic",
            "-nthreads", "1", "-compute-beta", "-idfile", "ids.txt"]
    r = subprocess.run(cmd2, cwd=run, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    bdir = run / "n200-k3-l10000-xx"
    beta = _read_matrix(bdir / "beta.txt")
    assert beta.shape == (10000, 4) and np.array_equal(beta[:, 0], np.arange(10000))
    gs = open(bdir / "gammasave.txt").read().splitlines()
    assert len(gs) == 200 and gs[0].split("\t")[:2] == ["0", "ind0"] and gs[149].split("\t")[1] == "ind149"
    assert gs[150].split("\t")[1] == "unknown"
    # same sweep in the oracle: gamma re-read from the %.8f text, fresh default-seed validation sample
    o2 = op.Oracle(200, 10000, 3)
    o2.read_bed_file(os.path.join(REF_DATA, "test.bed"))
    rng = op.gsl_mt19937(0)
    op.lib().orc_set_validation_sample(o2.s, C.byref(rng))
    o2.set_gamma(gamma)
    o2.compute_all_lambda()
    assert np.max(np.abs(beta[:, 1:] - o2.ebeta())) <= 1e-6
    # and against the reference's ground truth (allele coding flipped)
    truth_t = np.loadtxt(os.path.join(REF_DATA, "oracle_theta.txt"))
    perm = min(itertools.permutations(range(3)), key=lambda p: np.mean((theta[:, list(p)] - truth_t) ** 2))
    truth_b = np.loadtxt(os.path.join(REF_DATA, "oracle_beta.txt"))
    assert np.sqrt(np.mean(((1 - beta[:, 1:][:, list(perm)]) - truth_b) ** 2)) <= 0.05


@pytest.mark.gpu
def test_run_sh_sharded_over_devices(host_bin, tmp_path):
    """The same command with the individuals sharded (-devices; all shards on device 0 here):
    same validation trajectory and stop iteration, theta within the stated tolerance of the
    single-shard run's oracle."""
    data = tmp_path / "data"
    data.mkdir()
    for f in ("test.bed", "test.bim", "test.fam"):
        shutil.copy(os.path.join(REF_DATA, f), data / f)
    cmd = [host_bin, "-file", "test.bed", "-n", "200", "-l", "10000", "-k", "3", "-stochastic",
           "-nthreads", "1", "-rfreq", "1000", "-seed", "1234", "-label", "test", "-devices", "0,0,0"]
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")  # shards sharing a device need a hardware queue each
    r = subprocess.run(cmd, cwd=data, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    run = data / "n200-k3-l10000-test-seed1234"
    orc = op.Oracle(200, 10000, 3)
    orc.read_bed_file(os.path.join(REF_DATA, "test.bed"))
    res = orc.run(seed=1234, reportfreq=1000)
    val = [line.split("\t") for line in open(run / "validation.txt").read().splitlines()]
    assert len(val) == len(res["lines"]) == 17 and val[-1][0] == "16050"
    for got, (it, ll, cnt) in zip(val, res["lines"]):
        assert int(got[0]) == it and int(got[3]) == cnt
        assert abs(float(got[2]) - ll) < 1e-8
    theta = _read_matrix(run / "theta.txt")
    assert theta.shape == (200, 3)
    assert np.max(np.abs(theta - orc.theta())) <= 1e-6
    # -compute-beta, sharded as well
    cmd2 = [host_bin, "-file", "../test.bed", "-n", "200", "-l", "10000", "-k", "3", "-stochastic",
            "-nthreads", "1", "-compute-beta", "-devices", "0,0"]
    r = subprocess.run(cmd2, cwd=run, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    beta = _read_matrix(run / "n200-k3-l10000-xx" / "beta.txt")
    gamma = _read_matrix(run / "gamma.txt")
    o2 = op.Oracle(200, 10000, 3)
    o2.read_bed_file(os.path.join(REF_DATA, "test.bed"))
    rng = op.gsl_mt19937(0)
    op.lib().orc_set_validation_sample(o2.s, C.byref(rng))
    o2.set_gamma(gamma)
    o2.compute_all_lambda()
    assert np.max(np.abs(beta[:, 1:] - o2.ebeta())) <= 1e-6


@pytest.mark.gpu
def test_text_012_input_equals_bed(host_bin, tmp_path):
    """SNP::read's other branch (src/snp.cc:16-92): the same genotypes as a .012 text file give
    byte-identical outputs to the .bed run (same packed columns on the device)."""
    from helpers import unpack_bed

    data = tmp_path / "data"
    data.mkdir()
    for f in ("test.bed", "test.bim", "test.fam"):
        shutil.copy(os.path.join(REF_DATA, f), data / f)
    raw = np.fromfile(data / "test.bed", dtype=np.uint8)[3:].reshape(10000, 50)
    y = unpack_bed(raw, 200)
    chars = np.array(list("012-"))
    with open(data / "test.012", "w") as f:
        for row in y:
            f.write("".join(chars[row]) + "\n")
    outs = {}
    for name in ("test.bed", "test.012"):
        cmd = [host_bin, "-file", name, "-n", "200", "-l", "10000", "-k", "3", "-stochastic", "-nthreads", "1",
               "-rfreq", "1000", "-seed", "77", "-label", name[-3:], "-max-iter", "3000"]
        r = subprocess.run(cmd, cwd=data, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
        run = data / f"n200-k3-l10000-{name[-3:]}-seed77"
        outs[name] = {f: open(run / f).read() for f in ("theta.txt", "gamma.txt", "validation.txt", "param.txt")}
    b, t = outs["test.bed"], outs["test.012"]
    assert b["theta.txt"] == t["theta.txt"] and b["gamma.txt"] == t["gamma.txt"]
    assert [ln.split("\t")[0::2] for ln in b["validation.txt"].splitlines()] == \
           [ln.split("\t")[0::2] for ln in t["validation.txt"].splitlines()]      # iteration, log likelihood, likelihood
    cnt = lambda txt, key: int([ln for ln in txt.splitlines() if ln.startswith(key)][0].split(": ")[1])  # noqa: E731
    assert cnt(t["param.txt"], "missing snps") == cnt(b["param.txt"], "missing snps") == int((y == 3).sum())
    assert cnt(t["param.txt"], "0s snps") == int((y == 0).sum()) == cnt(b["param.txt"], "2s snps")   # (.bed labels are swapped in the reference)
    assert cnt(t["param.txt"], "2s snps") == int((y == 2).sum())
    # unknown extensions are refused like the reference does
    r = subprocess.run([host_bin, "-file", "test.txt", "-n", "200", "-l", "10000", "-k", "3"], cwd=data,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "unrecognized file extension" in r.stderr


@pytest.mark.gpu
def test_locations_file_branch(host_bin, tmp_path):
    """-compute-beta -locations-file (compute_and_save_beta, src/snpsamplinge.cc:385-413, flag
    src/main.cc:176): only the listed locations are optimised, in file order (first integer of each
    line, the rest of the line ignored), with the 100-pass cap and the gamma steps the reference
    keeps applying; beta.txt lists exactly those rows."""
    data = tmp_path / "data"
    data.mkdir()
    for f in ("test.bed", "test.bim", "test.fam"):
        shutil.copy(os.path.join(REF_DATA, f), data / f)
    rng = np.random.default_rng(12)
    gamma = rng.gamma(2.0, 1.5, size=(200, 3)) + 0.05
    with open(data / "gamma.txt", "w") as f:       # load_gamma reads ./gamma.txt of the cwd (:804)
        for row in gamma:
            f.write("".join("%.8f\t" % v for v in row) + "\n")
    gamma = np.array([[float("%.8f" % v) for v in row] for row in gamma])
    locs = [int(x) for x in rng.integers(0, 10000, size=40)]
    locs[7] = locs[6]                               # the same location twice in a row
    locs[20] = locs[3]                              # and revisited later
    with open(data / "locs.txt", "w") as f:
        for i, loc in enumerate(locs):
            f.write(f"{loc}\trs{i}\tsome annotation {i}\n")
    cmd = [host_bin, "-file", "test.bed", "-n", "200", "-l", "10000", "-k", "3", "-stochastic", "-nthreads", "1",
           "-compute-beta", "-locations-file", "locs.txt", "-label", "locs"]
    r = subprocess.run(cmd, cwd=data, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    beta = _read_matrix(data / "n200-k3-l10000-locs" / "beta.txt")
    assert beta.shape == (40, 4) and [int(x) for x in beta[:, 0]] == locs
    orc = op.Oracle(200, 10000, 3, online_iterations=100)
    orc.read_bed_file(os.path.join(REF_DATA, "test.bed"))
    r0 = op.gsl_mt19937(0)
    op.lib().orc_set_validation_sample(orc.s, C.byref(r0))
    orc.set_gamma(gamma)
    its = [orc.snp_update(loc) for loc in locs]
    assert max(its) > 10                           # the 100-pass cap is what is in force
    assert np.max(np.abs(beta[:, 1:] - orc.ebeta()[locs])) <= 1e-6


@pytest.mark.gpu
def test_streaming_ingest_rate(host_bin, tmp_path):
    """SURVEY 8f.4 at scale: an 8 GB PLINK .bed (N = 1M individuals x 32 768 SNPs) through the
    CLI's ingest pipeline (reader threads -> pinned double buffer -> strided DMA into HBM), with the
    genotype tallies taken on the device.  Reports GB/s; the floor asserted here is deliberately
    low (shared test boxes), the measured figure is recorded in DESIGN.md."""
    n, l = 1_000_000, 32768
    bps = n // 4
    free = shutil.disk_usage(tmp_path).free
    if free < 10 * (1 << 30):
        pytest.skip(f"needs 10 GB of scratch space, {free >> 30} GB free")
    rng = np.random.default_rng(5)
    block_cols = 256                                # one random 64 MB block, written 128 times
    block = rng.integers(0, 256, size=(block_cols, bps), dtype=np.uint8)
    with open(tmp_path / "big.bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        for _ in range(l // block_cols):
            f.write(block.tobytes())
    for ext, count in ((".bim", l), (".fam", n)):
        with open(tmp_path / ("big" + ext), "w") as f:
            f.write("x\n" * count)
    cmd = [host_bin, "-file", "big.bed", "-n", str(n), "-l", str(l), "-k", "8", "-label", "ingest", "-ingest-only"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("+ ingest:")][0]
    gbps = float(line.split("(")[1].split(" GB/s")[0])
    print("\n" + line)
    param = open(tmp_path / f"n{n}-k8-l{l}-ingest" / "param.txt").read()
    cnt = lambda key: int([ln for ln in param.splitlines() if ln.startswith(key)][0].split(": ")[1])  # noqa: E731
    codes = np.stack([(block >> (2 * j)) & 3 for j in range(4)])
    want = np.bincount(codes.ravel(), minlength=4) * (l // block_cols)
    assert cnt("missing snps") == want[1] and cnt("0s snps") == want[3]      # (labels swapped like the reference)
    assert cnt("1s snps") == want[2] and cnt("2s snps") == want[0]
    assert gbps > 4.0, line


@pytest.mark.gpu
def test_individual_major_bed_equals_snp_major(host_bin, tmp_path):
    """A PLINK individual-major .bed (magic 6c 1b 00), which the reference refuses, gives byte-identical
    outputs to the SNP-major file of the same genotypes (transposed on the device at ingest; the
    validation sample then reads its columns back from HBM)."""
    from helpers import pack_bed, unpack_bed

    data = tmp_path / "data"
    data.mkdir()
    for f in ("test.bed", "test.bim", "test.fam"):
        shutil.copy(os.path.join(REF_DATA, f), data / f)
    raw = np.fromfile(data / "test.bed", dtype=np.uint8)[3:].reshape(10000, 50)
    y = unpack_bed(raw, 200)                                   # [l][n]
    with open(data / "im.bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x00]))
        f.write(pack_bed(y.T.copy()).tobytes())               # [n][ceil(l/4)]
    for ext in (".bim", ".fam"):
        shutil.copy(data / ("test" + ext), data / ("im" + ext))
    outs = {}
    for name in ("test.bed", "im.bed"):
        cmd = [host_bin, "-file", name, "-n", "200", "-l", "10000", "-k", "3", "-stochastic", "-nthreads", "1",
               "-rfreq", "1000", "-seed", "99", "-label", name[:2], "-max-iter", "2100"]
        r = subprocess.run(cmd, cwd=data, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
        run = data / f"n200-k3-l10000-{name[:2]}-seed99"
        outs[name] = {f: open(run / f).read() for f in ("theta.txt", "gamma.txt", "validation.txt", "param.txt")}
    a, b = outs["test.bed"], outs["im.bed"]
    assert a["theta.txt"] == b["theta.txt"] and a["gamma.txt"] == b["gamma.txt"]
    assert [ln.split("\t")[0::2] for ln in a["validation.txt"].splitlines()] == \
           [ln.split("\t")[0::2] for ln in b["validation.txt"].splitlines()]
    keys = ("missing snps", "0s snps", "1s snps", "2s snps", "total validation snps")
    pick = lambda txt: [ln for ln in txt.splitlines() if ln.startswith(keys)]  # noqa: E731
    assert pick(a["param.txt"]) == pick(b["param.txt"])


@pytest.mark.gpu
def test_cli_end_to_end_at_scale(host_bin, tmp_path):
    """The drop-in binary at a BASELINE number of individuals (5 s; the full-size runs are profiles/r06_cli_end_to_end.txt): N = 100 000 individuals (config 3's), L = 20 000, K = 8 -- a synthetic PSD .bed generated on
    the GPU (tools/make_synth_bed.py), two report periods.  Checks what the reference's files promise -- theta rows sum to 1,
    gamma.txt / theta.txt have N rows of K "%.8f" values with a trailing tab, the held-out log-likelihood improves from its initial
    value -- and what round 6 added: save_model is off the critical path (timing.txt) and every file is complete when the process ends."""
    n, l, k = 100_000, 20_000, 8
    free = shutil.disk_usage(tmp_path).free
    if free < 2 * (1 << 30):
        pytest.skip(f"needs 2 GB of scratch space, {free >> 30} GB free")
    prefix = str(tmp_path / "e2e")
    g = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_bed.py"), prefix, str(n), str(l), str(k)],
                       capture_output=True, text=True, timeout=600)
    assert g.returncode == 0, g.stderr[-2000:]
    cmd = [host_bin, "-file", "e2e.bed", "-n", str(n), "-l", str(l), "-k", str(k), "-stochastic", "-nthreads", "1", "-label", "e2e",
           "-rfreq", "5000", "-max-iter", "10200"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2000:])
    run = tmp_path / f"n{n}-k{k}-l{l}-e2e"
    for name in ("gamma.txt", "theta.txt"):
        rows = open(run / name).read().split("\n")
        assert rows[-1] == "" and len(rows) == n + 1                       # complete: N lines, the last one terminated
        assert all(ln.endswith("\t") and ln.count("\t") == k for ln in rows[:-1:997])
    theta = _read_matrix(run / "theta.txt")
    assert theta.shape == (n, k) and np.max(np.abs(theta.sum(axis=1) - 1.0)) < 1e-6
    val = [ln.split("\t") for ln in open(run / "validation.txt").read().splitlines()]
    assert len(val) == 3 and float(val[-1][2]) > float(val[0][2])            # initial + two reports; the fit improves
    tm = dict(ln.split(": ", 1) for ln in open(run / "timing.txt").read().splitlines() if ": " in ln and not ln.startswith("#"))
    training = float(tm["training"])
    blocked = float(tm["save_model, main thread blocked"].split(" ")[0])
    assert training > 0 and blocked < 0.5 * training, tm                     # (the writer thread does the formatting)

"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Tolerances (fp64; the device evaluates the reference's softmax in the linear
domain and reduces in a different, fixed order):
  single pass / single gamma step, known answer: rel 1e-12
  trajectories of tens of SNP updates:           rel 1e-9 on lambda and gamma
  config-1 run (thousands of updates):           |dtheta| <= 1e-6
Integer state (c_n, inner pass counts, bed bytes) is compared exactly.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_py as op
from conftest import REF_DATA
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err, unpack_bed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ts():
    import terastructure_amd as t

    t.load()
    return t


def make_pair(ts, n, l, k, seed, missing=0.02, held=True, flags=0, **cfg):
    """Engine + Oracle loaded with the same genotypes, gamma and held-out set."""
    y, _, _ = psd_genotypes(n, l, k, seed, missing)
    payload = pack_bed(y)
    eng = ts.Engine(n, l, k, flags=flags, **cfg)
    ocfg = {}
    if "max_inner" in cfg:
        ocfg["online_iterations"] = cfg["max_inner"]
    orc = op.Oracle(n, l, k, **ocfg)
    eng.upload_bed(payload)
    orc.load_bed_payload(payload)
    g = init_gamma(n, k, seed + 1)
    eng.set_gamma(g)
    orc.set_gamma(g)
    if held:
        rng = np.random.default_rng(seed + 2)
        for loc in rng.choice(l, size=max(1, l // 8), replace=False):
            cand = np.nonzero(y[loc] != 3)[0]
            pick = rng.choice(cand, size=min(len(cand), max(1, n // 50)), replace=False)
            eng.set_heldout(int(loc), pick)
            orc.set_heldout(int(loc), pick)
    return eng, orc, y


def assert_state_close(eng, orc, tol, what=""):
    assert rel_err(eng.get_lambda(), orc.lambda_()) < tol, what + " lambda"
    assert rel_err(eng.get_gamma(), orc.gamma()) < tol, what + " gamma"
    assert np.array_equal(eng.get_counts(), orc.c_indiv()), what + " c_n"


def test_bed_roundtrip_and_padding(ts):
    n, l, k = 1003, 9, 3  # n % 4 != 0 exercises the partial last byte
    y, _, _ = psd_genotypes(n, l, k, 5, 0.1)
    payload = pack_bed(y)
    with ts.Engine(n, l, k) as eng:
        # before upload every genotype is missing
        assert np.all(unpack_bed(eng.download_bed(0)[None, :], n) == 3)
        eng.upload_bed(payload[:4], first_loc=0)
        eng.upload_bed(payload[4:], first_loc=4)
        for loc in range(l):
            got = unpack_bed(eng.download_bed(loc)[None, :], n)[0]
            assert np.array_equal(got, y[loc])
        # held-out entries read back as missing, everything else untouched
        ok = np.nonzero(y[2] != 3)[0][:7]
        eng.set_heldout(2, ok)
        got = unpack_bed(eng.download_bed(2)[None, :], n)[0]
        want = y[2].copy()
        want[ok] = 3
        assert np.array_equal(got, want)


def test_initial_state_matches_init_lambda(ts):
    n, l, k = 300, 5, 4
    with ts.Engine(n, l, k) as eng:
        orc = op.Oracle(n, l, k)
        assert np.array_equal(eng.get_lambda(), orc.lambda_())
        assert rel_err(eng.get_elogbeta(), orc.elogbeta()) < 1e-13  # psi(1) - psi(2) = -1
        assert np.allclose(eng.get_ebeta(), 0.5, rtol=0, atol=0)
        g = init_gamma(n, k, 3)
        eng.set_gamma(g)
        orc.set_gamma(g)
        assert np.array_equal(eng.get_gamma(), g)
        assert rel_err(eng.get_theta(), orc.theta()) < 1e-14
        assert rel_err(eng.get_elogtheta(), orc.elogtheta()) < 1e-12


@pytest.mark.parametrize("k", [1, 2, 3, 4, 6, 8, 12, 20, 32, 33, 40, 100, 128])
def test_single_pass_known_answer(ts, k):
    """One pass of phi + lambda_t + epilogue (max_inner = 1): rel 1e-12."""
    n, l = 2500, 6
    eng, orc, _ = make_pair(ts, n, l, k, 100 + k, max_inner=1)
    with eng:
        for loc in (0, 3, 5):
            it = eng.snp_update(loc)
            assert it == orc.snp_update(loc) == 1
            lam_d = eng.get_lambda(loc, 1)[0]
            lam_o = orc.lambda_()[loc]
            assert rel_err(lam_d, lam_o) < 1e-12
            assert rel_err(eng.get_ebeta(loc, 1)[0], orc.ebeta()[loc]) < 1e-12
            assert rel_err(eng.get_elogbeta(loc, 1)[0], orc.elogbeta()[loc]) < 1e-11


@pytest.mark.parametrize("k", [3, 8])
def test_deferred_gamma_step_known_answer(ts, k):
    """The step of SNP t is applied at the start of SNP t+1 with the phi of t's LAST
    pass (src/snpsamplinge.cc:660-668): gamma/theta/Elogtheta/c_n after it, rel 1e-12."""
    n, l = 1500, 8
    eng, orc, _ = make_pair(ts, n, l, k, 7 + k)
    with eng:
        g0 = eng.get_gamma()
        assert eng.snp_update(2) == orc.snp_update(2)
        # nothing applied yet (pending)
        assert np.array_equal(eng.get_gamma(), g0)
        assert not eng.get_counts().any()
        assert eng.snp_update(5) == orc.snp_update(5)
        assert rel_err(eng.get_gamma(), orc.gamma()) < 1e-12
        assert rel_err(eng.get_theta(), orc.theta()) < 1e-12
        assert rel_err(eng.get_elogtheta(), orc.elogtheta()) < 1e-11
        assert np.array_equal(eng.get_counts(), orc.c_indiv())
        assert eng.get_counts().max() == 1


def test_hol_mode_suppresses_step(ts):
    """A SNP run in hol mode leaves no pending step; the step of the last training SNP
    before it IS applied (src/snpsamplinge.cc:664)."""
    n, l, k = 800, 10, 4
    eng, orc, _ = make_pair(ts, n, l, k, 21)
    with eng:
        seq = [(1, 0), (4, 1), (6, 1), (2, 0), (3, 0)]
        for loc, hol in seq:
            assert eng.snp_update(loc, hol) == orc.snp_update(loc, hol)
        assert_state_close(eng, orc, 1e-10, "hol sequence")
        assert eng.get_counts().max() == 2  # steps of loc 1 and loc 2 only


@pytest.mark.parametrize("n,l,k", [(200, 40, 3), (1000, 64, 6), (5000, 48, 8), (3001, 32, 20), (70000, 24, 5), (2000, 16, 31),
                                   (3000, 24, 48), (9001, 16, 70)])
def test_trajectory_matches_oracle(ts, n, l, k):
    eng, orc, _ = make_pair(ts, n, l, k, 1000 + n)
    rng = np.random.default_rng(n)
    locs = rng.integers(0, l, size=40)
    with eng:
        its_d = [eng.snp_update(int(loc)) for loc in locs]
        its_o = [orc.snp_update(int(loc)) for loc in locs]
        assert its_d == its_o
        assert_state_close(eng, orc, 1e-9, f"n={n}")
        assert eng.total_passes() == sum(its_o)
        assert np.array_equal(eng.pass_histogram(), np.bincount(its_o, minlength=128).astype(np.uint64))


@pytest.mark.parametrize("n,l,k", [(4000, 64, 8), (90000, 40, 5), (600000, 40, 8)])
def test_run_schedule_equals_snp_updates_bitwise(ts, n, l, k, monkeypatch):
    """run_schedule == n x snp_update, bit for bit, and is reproducible run to run (fixed reduction order) -- also
    at sizes where a thread of the resident kernels owns many individuals.  From 4M weights per GPU on, single-entry
    calls are routed through the launch-per-SNP kernels by default (faster for that shape, include/tsamd.h): there
    the bitwise statement holds with TSAMD_SINGLE_ROUTE=0, and the routed calls agree to rounding."""
    rng = np.random.default_rng(9)
    locs = rng.integers(0, l, size=37).astype(np.uint32)
    routed = n * k >= 4 << 20
    outs = []
    for mode in ("eager", "schedule", "schedule", "nograph") + (("eager-routed",) if routed else ()):
        flags = ts.FLAG_NO_GRAPH if mode == "nograph" else 0
        monkeypatch.setenv("TSAMD_SINGLE_ROUTE", "1" if mode == "eager-routed" else "0")
        eng, _, _ = make_pair(ts, n, l, k, 77, flags=flags)
        with eng:
            if mode.startswith("eager"):
                for loc in locs:
                    eng.snp_update(int(loc))
            else:
                eng.run_schedule(locs)
                eng.synchronize()
            outs.append((eng.get_lambda(), eng.get_gamma(), eng.get_counts(), eng.total_passes()))
    for o in outs[1:4]:
        assert np.array_equal(o[0], outs[0][0])
        assert np.array_equal(o[1], outs[0][1])
        assert np.array_equal(o[2], outs[0][2])
        assert o[3] == outs[0][3]
    if routed:
        o = outs[4]
        assert not np.array_equal(o[0], outs[0][0]), "the routed calls ran the same kernels?"
        assert rel_err(o[0], outs[0][0]) < 1e-10 and rel_err(o[1], outs[0][1]) < 1e-10
        assert np.array_equal(o[2], outs[0][2]) and o[3] == outs[0][3]


@pytest.mark.parametrize("n,l,k", [(3000, 32, 6), (90000, 32, 6)])
def test_split_epilogue_path_bitwise(ts, n, l, k):
    """The sharded kernel sequence (pass -> row sum -> exchange) with one shard gives the
    same bits as the single-GPU sequence."""
    locs = np.random.default_rng(4).integers(0, l, size=20).astype(np.uint32)
    res = []
    for flags in (0, ts.FLAG_SPLIT_EPILOGUE):
        eng, _, _ = make_pair(ts, n, l, k, 55, flags=flags)
        with eng:
            eng.set_launch_mode(ts.LAUNCH_PER_PASS)  # (the resident kernels add the partial rows in another order)
            eng.run_schedule(locs)
            eng.synchronize()
            res.append((eng.get_lambda(), eng.get_gamma()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_rccl_single_rank_allreduce(ts):
    """world = 1 communicator: exercises the RCCL plumbing (dlopen, unique id,
    ncclCommInitRank, ncclAllReduce on the engine's stream) on one GPU."""
    n, l, k = 2000, 16, 8
    locs = np.random.default_rng(6).integers(0, l, size=12).astype(np.uint32)
    ref, _, _ = make_pair(ts, n, l, k, 88)
    with ref:
        ref.set_launch_mode(ts.LAUNCH_PER_PASS)  # (same summation order as the sharded sequence)
        ref.run_schedule(locs)
        ref.synchronize()
        want = (ref.get_lambda(), ref.get_gamma())
    eng, _, _ = make_pair(ts, n, l, k, 88, flags=ts.FLAG_SPLIT_EPILOGUE)
    with eng:
        eng.comm_init(eng.comm_unique_id())
        eng.run_schedule(locs)
        eng.synchronize()
        assert np.array_equal(eng.get_lambda(), want[0])
        assert np.array_equal(eng.get_gamma(), want[1])


def test_heldout_loglik(ts):
    n, l, k = 1200, 24, 5
    eng, orc, _ = make_pair(ts, n, l, k, 31)
    with eng:
        for loc in (1, 7, 3):
            eng.snp_update(loc)
            orc.snp_update(loc)
        for loc in orc.heldout_locs():
            eng.snp_update(int(loc), 1)
            orc.snp_update(int(loc), 1)
            sd, cd = eng.heldout_loglik(int(loc))
            so, co = orc.heldout_loglik(int(loc))
            assert cd == co and co > 0
            assert abs(sd - so) <= 1e-10 * abs(so)


def test_sharded_contexts_slice_columns(ts):
    """Two shard contexts (world = 2) take the right byte ranges, ids and rows."""
    n, l, k = 1003, 6, 3
    y, _, _ = psd_genotypes(n, l, k, 8, 0.05)
    payload = pack_bed(y)
    for rank in (0, 1):
        with ts.Engine(n, l, k, rank=rank, world=2) as eng:
            b, c = ts.shard_range(n, rank, 2)
            assert (b, c) == (eng.shard_begin, eng.shard_count) and b % 4 == 0
            eng.upload_bed(payload)
            for loc in range(l):
                got = unpack_bed(eng.download_bed(loc)[None, :], c)[0]
                assert np.array_equal(got, y[loc, b:b + c])
            ok = np.nonzero(y[1] != 3)[0]
            eng.set_heldout(1, ok[::5])  # global ids; the shard keeps its own
            got = unpack_bed(eng.download_bed(1)[None, :], c)[0]
            want = y[1].copy()
            want[ok[::5]] = 3
            assert np.array_equal(got, want[b:b + c])
            # a sharded context without a communicator must refuse to run
            with pytest.raises(ts.TsamdError):
                eng.run_schedule(np.array([0], dtype=np.uint32))


def test_error_paths(ts):
    with pytest.raises(ts.TsamdError):
        ts.Engine(100, 10, 129)  # K above compiled maximum
    with pytest.raises(ts.TsamdError):
        ts.Engine(0, 10, 3)
    with ts.Engine(100, 10, 3) as eng:
        with pytest.raises(ts.TsamdError):
            eng.snp_update(10)  # loc out of range
        with pytest.raises(ts.TsamdError):
            eng.set_gamma(np.zeros((100, 3)))  # gamma must be positive
        with pytest.raises(ts.TsamdError):
            eng.upload_bed(np.zeros((1, 7), dtype=np.uint8))  # wrong bytes_per_snp
        assert "bytes_per_snp" in eng.h.tsamd_last_error(eng.ctx).decode()


def test_config1_reference_fixture(ts):
    """data/run.sh on data/test.bed: same RNG stream as the oracle (validation sample,
    initial gamma, location draws), first report period (1000 training + 50 validation
    updates).  |dtheta| <= 1e-6, validation log-likelihood to 1e-9."""
    n, l, k = 200, 10000, 3
    with open(os.path.join(REF_DATA, "test.bed"), "rb") as f:
        raw = f.read()
    assert raw[:3] == bytes([0x6C, 0x1B, 0x01])
    payload = np.frombuffer(raw, dtype=np.uint8, offset=3).reshape(l, (n + 3) // 4)
    L = op.lib()
    orc = op.Oracle(n, l, k)
    orc.load_bed_payload(payload)
    r = op.gsl_mt19937(1234)
    L.orc_set_validation_sample(orc.s, C.byref(r))
    L.orc_init_gamma(orc.s, C.byref(r))
    with ts.Engine(n, l, k) as eng:
        eng.upload_bed(payload)
        for loc in orc.heldout_locs():
            eng.set_heldout(int(loc), orc.heldout_indivs(int(loc)))
        eng.set_gamma(orc.gamma())
        locs = np.array([L.orc_rng_uniform_int(C.byref(r), l) for _ in range(1000)], dtype=np.uint32)
        eng.run_schedule(locs)
        for loc in locs:
            orc.snp_update(int(loc))
        # the validation block in one call (tsamd_heldout_eval): hol-mode schedule over the
        # validation locations + one evaluation kernel; the oracle goes location by location
        vlocs = orc.heldout_locs()
        p0 = eng.total_passes()
        sd, cnt, sums_d, cnts_d = eng.heldout_eval(vlocs)
        so, its_o = 0.0, 0
        for j, loc in enumerate(vlocs):
            its_o += orc.snp_update(int(loc), 1)
            b, c2 = orc.heldout_loglik(int(loc))
            assert cnts_d[j] == c2 and abs(sums_d[j] - b) <= 1e-9 * abs(b)
            so += b
        assert eng.total_passes() - p0 == its_o
        assert cnt == 1000
        assert "%.9f" % (so / cnt) == "-0.732008912"  # validation.txt line 2 of the reference run
        assert abs(sd / cnt - so / cnt) < 1e-9
        assert np.max(np.abs(eng.get_theta() - orc.theta())) <= 1e-6
        assert rel_err(eng.get_gamma(), orc.gamma()) < 1e-7
        # and against the committed golden vectors of the same run (tests/golden/make_golden.py)
        from conftest import GOLDEN

        g = np.load(os.path.join(GOLDEN, "config1_oracle.npz"))
        assert np.array_equal(g["locs"], locs) and np.array_equal(g["held_locs"], orc.heldout_locs())
        assert np.max(np.abs(eng.get_theta() - g["theta1050"])) <= 1e-6
        assert rel_err(eng.get_gamma(), g["gamma1050"]) < 1e-7
        assert rel_err(eng.get_lambda()[locs[:8]], g["lam_sample"]) < 1e-7


def test_full_size_invariants(ts):
    """BASELINE size (N = 1M individuals, K = 8): properties that need no oracle.
    Every phi row sums to 1, so per pass  sum_k lambda_t[k][0] = sum_n y_n  and
    sum_k lambda_t[k][1] = sum_n (2 - y_n)  over observed genotypes; one gamma step maps a
    row sum S to (1-rho) S + rho (K alpha + 2 L) with rho = (2 + c_n)^-1/2; theta rows sum
    to 1; and a re-run is bit-identical."""
    n, l, k = 1_000_000, 8, 8
    rng = np.random.default_rng(123)
    theta = rng.dirichlet(np.full(k, 0.2), size=n)
    beta = rng.uniform(0.05, 0.95, size=(l, k))
    g0 = rng.gamma(100.0, 0.01, size=(n, k))
    outs = []
    for rep in range(2):
        with ts.Engine(n, l, k) as eng:
            eng.synth_genotypes(theta, beta, seed=99, missing_rate=0.01)
            eng.set_gamma(g0)
            cols = [unpack_bed(eng.download_bed(j)[None, :], n)[0] for j in (2, 5)]
            assert eng.snp_update(2) == 10
            lam = eng.get_lambda(2, 1)[0]
            y = cols[0].astype(np.int64)
            ok = y != 3
            assert abs((lam[:, 0] - 1.0).sum() - y[ok].sum()) <= 1e-9 * y[ok].sum()
            assert abs((lam[:, 1] - 1.0).sum() - (2 - y[ok]).sum()) <= 1e-9 * (2 - y[ok]).sum()
            assert 0.005 < (~ok).mean() < 0.015                  # ~1 % missing as requested
            eng.snp_update(5)                                      # applies the step of location 2
            g1 = eng.get_gamma()
            rho = 2.0 ** -0.5
            want = np.where(ok, (1 - rho) * g0.sum(1) + rho * (k * (1.0 / k) + 2 * l), g0.sum(1))
            assert np.max(np.abs(g1.sum(1) - want) / want) < 1e-12
            assert np.array_equal(eng.get_counts(), ok.astype(np.uint32))
            th = eng.get_theta()
            assert np.max(np.abs(th.sum(1) - 1.0)) < 1e-12 and th.min() > 0
            eb = eng.get_ebeta()
            assert eb.min() > 0 and eb.max() < 1
            outs.append((eng.get_lambda(), g1))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_wide_k_graph_replay_bitwise(ts):
    """K above TSAMD_SPECIALIZED_K (run-time-K fallback kernels): graph replay == eager, and the
    held-out likelihood matches the oracle."""
    n, l, k = 2500, 32, 37
    locs = np.random.default_rng(5).integers(0, l, size=21).astype(np.uint32)
    outs = []
    for flags in (0, ts.FLAG_NO_GRAPH):
        eng, orc, _ = make_pair(ts, n, l, k, 66, flags=flags)
        with eng:
            eng.run_schedule(locs)
            eng.synchronize()
            outs.append((eng.get_lambda(), eng.get_gamma()))
            if flags == 0:
                for loc in locs:
                    orc.snp_update(int(loc))
                assert_state_close(eng, orc, 1e-9, "wide K")
                hl = int(orc.heldout_locs()[0])
                eng.snp_update(hl, 1)
                orc.snp_update(hl, 1)
                a, c = eng.heldout_loglik(hl)
                b, c2 = orc.heldout_loglik(hl)
                assert c == c2 and abs(a - b) <= 1e-10 * abs(b)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_heldout_eval_equals_per_location_path_bitwise(ts):
    """tsamd_heldout_eval (one schedule + one kernel) == snp_update(loc, hol) + heldout_loglik per
    location, bit for bit, including locations without held-out entries and the
    evaluate-only mode used for the initial likelihood."""
    n, l, k = 5000, 40, 6
    train = np.random.default_rng(2).integers(0, l, size=25).astype(np.uint32)
    outs = []
    for mode in ("block", "per-location"):
        eng, orc, _ = make_pair(ts, n, l, k, 404)
        with eng:
            vl = [int(x) for x in orc.heldout_locs()] + [0 if 0 not in orc.heldout_locs() else 1]
            vl = np.array(sorted(set(vl)), dtype=np.uint32)
            s0 = eng.heldout_eval(vl, run_updates=False)            # initial likelihood: no updates
            assert s0[1] == sum(len(orc.heldout_indivs(int(x))) for x in orc.heldout_locs())
            eng.run_schedule(train)
            eng.synchronize()
            if mode == "block":
                s, c, sums, cnts = eng.heldout_eval(vl)
            else:
                sums, cnts = np.zeros(len(vl)), np.zeros(len(vl), dtype=np.uint32)
                for j, loc in enumerate(vl):
                    eng.snp_update(int(loc), 1)
                    sums[j], cnts[j] = eng.heldout_loglik(int(loc))
                s, c = 0.0, 0
                for j in range(len(vl)):
                    s += sums[j]
                    c += int(cnts[j])
            outs.append((s, c, sums.copy(), cnts.copy(), eng.get_lambda(), eng.get_gamma(), s0[0]))
    a, b = outs
    assert a[0] == b[0] and a[1] == b[1] and a[6] == b[6]
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
    assert (a[3] == 0).any() and (a[3] > 0).any()

"""GPU parity on the edge cases of the path: degenerate columns (all missing, monomorphic,
everybody held out), shards smaller than a wavefront, K = 1, extreme gamma values, the pass
caps the reference uses (1 ... 100), the same location many times in a row.  Tolerances as in
test_gpu_parity.py; integer state is compared exactly.
"""
import numpy as np
import pytest

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err
from test_gpu_parity import assert_state_close, ts  # noqa: F401

pytestmark = pytest.mark.gpu


def pair_from_y(ts, y, k, seed, gamma=None, **cfg):
    l, n = y.shape
    payload = pack_bed(y)
    eng = ts.Engine(n, l, k, **cfg)
    ocfg = {"online_iterations": cfg["max_inner"]} if "max_inner" in cfg else {}
    orc = op.Oracle(n, l, k, **ocfg)
    eng.upload_bed(payload)
    orc.load_bed_payload(payload)
    g = init_gamma(n, k, seed) if gamma is None else gamma
    eng.set_gamma(g)
    orc.set_gamma(g)
    return eng, orc


def run_both(eng, orc, locs, tol=1e-9, what=""):
    with eng:
        for loc in locs:
            assert eng.snp_update(int(loc)) == orc.snp_update(int(loc)), what
        assert_state_close(eng, orc, tol, what)


def test_degenerate_columns(ts):
    """All-missing, all-0, all-1, all-2 columns between ordinary ones."""
    n, l, k = 1003, 12, 4
    y, _, _ = psd_genotypes(n, l, k, 5, 0.01)
    y[2, :] = 3
    y[4, :] = 0
    y[6, :] = 2
    y[8, :] = 1
    eng, orc = pair_from_y(ts, y, k, 6)
    with eng:
        for loc in [0, 2, 2, 3, 4, 5, 6, 7, 8, 9, 2, 1]:
            assert eng.snp_update(loc) == orc.snp_update(loc)
        assert_state_close(eng, orc, 1e-9, "degenerate columns")
        # an all-missing column leaves lambda at eta and steps nobody
        lam = eng.get_lambda()[2]
        assert np.array_equal(lam, np.ones_like(lam))


def test_everybody_held_out_at_a_location(ts):
    n, l, k = 600, 8, 3
    y, _, _ = psd_genotypes(n, l, k, 15)
    eng, orc = pair_from_y(ts, y, k, 16)
    everybody = np.arange(n, dtype=np.uint32)
    eng.set_heldout(3, everybody)
    orc.set_heldout(3, everybody)
    run_both(eng, orc, [1, 3, 2, 3, 3, 4], what="all held out")


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 64, 65, 511, 512, 513])
def test_tiny_shards(ts, n):
    l, k = 10, 3
    y, _, _ = psd_genotypes(n, l, k, 100 + n, 0.05)
    eng, orc = pair_from_y(ts, y, k, 200 + n)
    run_both(eng, orc, np.random.default_rng(n).integers(0, l, size=15), what=f"n={n}")


def test_single_population(ts):
    """K = 1: phi is 1 for every observed parent copy."""
    n, l = 700, 10
    y, _, _ = psd_genotypes(n, l, 2, 3)
    eng, orc = pair_from_y(ts, y, 1, 4)
    run_both(eng, orc, [0, 1, 2, 1, 1, 5, 9], what="K=1")


@pytest.mark.parametrize("max_inner", [1, 2, 100])
def test_pass_caps(ts, max_inner):
    """-compute-beta runs with a cap of 100 passes (src/snpsamplinge.cc:84), the cap is the
    loop bound of optimize_lambda (:327)."""
    n, l, k = 900, 10, 5
    y, _, _ = psd_genotypes(n, l, k, 21)
    eng, orc = pair_from_y(ts, y, k, 22, max_inner=max_inner)
    locs = np.random.default_rng(1).integers(0, l, size=12)
    with eng:
        its = [eng.snp_update(int(loc)) for loc in locs]
        assert its == [orc.snp_update(int(loc)) for loc in locs]
        assert max(its) <= max_inner
        assert_state_close(eng, orc, 1e-9, f"cap {max_inner}")


@pytest.mark.parametrize("max_inner", [1, 2, 3, 10])
@pytest.mark.parametrize("n", [900, 40_000])
def test_schedules_with_pass_caps(ts, max_inner, n):
    """Whole schedules (graph replay and eager launches) under small pass caps: with a cap of 1 every
    launch is a first pass that finishes its predecessor; repeated locations take the slow path of
    the first pass (values still in flight), everything else the fast path (NextSnp)."""
    l, k = 12, 5
    y, _, _ = psd_genotypes(n, l, k, 61, 0.03)
    locs = np.array([3, 3, 7, 1, 7, 7, 7, 0, 2, 2, 5, 9, 11, 4, 4, 6, 8, 10, 3, 1, 0, 0, 5], dtype=np.uint32)
    outs = []
    for flags in (0, ts.FLAG_NO_GRAPH):
        eng, orc = pair_from_y(ts, y, k, 62, flags=flags, max_inner=max_inner)
        with eng:
            eng.run_schedule(locs[:9])
            eng.run_schedule(locs[9:10], 1)     # one validation-mode update in between
            eng.run_schedule(locs[10:])
            eng.synchronize()
            if flags == 0:
                its = [orc.snp_update(int(x), 1 if i == 9 else 0) for i, x in enumerate(locs)]
                assert eng.total_passes() == sum(its) and max(its) <= max_inner
                assert_state_close(eng, orc, 1e-9, f"cap {max_inner}")
            outs.append((eng.get_lambda(), eng.get_gamma(), eng.get_counts()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_extreme_gamma_values(ts):
    """gamma from 1e-3 (psi = -1000: the weight underflows next to its neighbours) to 1e7."""
    n, l, k = 512, 8, 6
    y, _, _ = psd_genotypes(n, l, k, 31)
    rng = np.random.default_rng(32)
    g = 10.0 ** rng.uniform(-3, 7, size=(n, k))
    g[:8, :] = 1e-3      # every population tiny: only the per-individual scaling keeps w finite
    g[8:16, :] = 1e7
    eng, orc = pair_from_y(ts, y, k, 0, gamma=g)
    with eng:
        for loc in [0, 1, 2, 3, 1]:
            assert eng.snp_update(loc) == orc.snp_update(loc)
        assert np.all(np.isfinite(eng.get_lambda())) and np.all(np.isfinite(eng.get_gamma()))
        assert rel_err(eng.get_lambda(), orc.lambda_()) < 1e-9
        assert rel_err(eng.get_gamma(), orc.gamma()) < 1e-9
        th = eng.get_theta()
        assert np.allclose(th.sum(axis=1), 1.0, atol=1e-12)


def test_tiniest_accepted_gamma(ts):
    """gamma = 1e-8 (the smallest non-zero value of a %.8f gamma.txt; psi = -1e8) next to components of order 1 and on its
    own: the exponent of exp(psi(gamma) - max) is read off the low word of d / ln 2 + 1.5 * 2^52 and must not wrap; anything
    smaller is refused at the boundary (gamma never falls below min(gamma, alpha), so the bound holds for the whole run).
    Tolerance 1e-7 here, not 1e-9: it is the REFERENCE's formulation that loses digits at this extreme -- it adds Elogbeta
    (order 1) to Elogtheta = psi(1e-8) - psi(sum) (order -1e8, one ulp = 1.5e-8) before the softmax, so an individual whose
    every gamma is 1e-8 gets phi to ~1e-8 relative; the device's linear-domain weights are exact there (w_k = 10 exp(0))."""
    n, l, k = 512, 8, 5
    y, _, _ = psd_genotypes(n, l, k, 41)
    rng = np.random.default_rng(42)
    g = rng.gamma(100.0, 0.01, size=(n, k))
    g[rng.random((n, k)) < 0.3] = 1e-8
    g[:8, :] = 1e-8
    g[8:16, 1:] = 1e-8
    eng, orc = pair_from_y(ts, y, k, 0, gamma=g)
    with eng:
        for loc in [0, 1, 2, 3, 1, 0]:
            assert eng.snp_update(loc) == orc.snp_update(loc)
        assert np.all(np.isfinite(eng.get_lambda())) and np.all(np.isfinite(eng.get_gamma()))
        assert rel_err(eng.get_lambda(), orc.lambda_()) < 1e-7
        assert rel_err(eng.get_gamma(), orc.gamma()) < 1e-7
        bad = g.copy()
        bad[3, 2] = 9e-9
        with pytest.raises(ts.TsamdError):
            eng.set_gamma(bad)
    with pytest.raises(ts.TsamdError):
        ts.Engine(n, l, k, alpha=1e-9)


def test_same_location_many_times(ts):
    n, l, k = 2000, 4, 4
    y, _, _ = psd_genotypes(n, l, k, 41)
    eng, orc = pair_from_y(ts, y, k, 42)
    locs = np.array([1] * 12 + [2] + [1] * 5, dtype=np.uint32)
    with eng:
        eng.run_schedule(locs)
        eng.synchronize()
        its = [orc.snp_update(int(loc)) for loc in locs]
        assert eng.total_passes() == sum(its)
        assert_state_close(eng, orc, 1e-9, "repeats")


def test_schedule_of_every_location_in_order(ts):
    """compute_all_lambda order (src/snpsamplinge.cc:368-377): 0, 1, ..., L-1."""
    n, l, k = 1500, 40, 3
    y, _, _ = psd_genotypes(n, l, k, 51, 0.02)
    eng, orc = pair_from_y(ts, y, k, 52, max_inner=100)
    with eng:
        eng.run_schedule(np.arange(l, dtype=np.uint32))
        eng.synchronize()
        for loc in range(l):
            orc.snp_update(loc)
        assert_state_close(eng, orc, 1e-9, "in order")


@pytest.mark.parametrize("n,l", [(1003, 517), (4096, 64), (70, 5000)])
def test_individual_major_upload_equals_snp_major(ts, n, l):
    """PLINK's individual-major layout (third magic byte 0; the reference refuses it, src/snp.cc:176-178)
    transposed on the device gives the same 2-bit columns as the SNP-major upload, for sizes that are not
    multiples of the transpose tile, in several batches; the device-side genotype tallies agree with numpy."""
    k = 3
    y, _, _ = psd_genotypes(n, l, k, 17 + n, 0.07)
    cols = pack_bed(y)          # [l][ceil(n/4)]  SNP-major
    rows = pack_bed(y.T.copy())  # [n][ceil(l/4)]  individual-major: location j at bits 2(j%4) of byte j/4
    with ts.Engine(n, l, k) as a, ts.Engine(n, l, k) as b:
        a.upload_bed(cols)
        cut = min(n, 48) // 16 * 16 or n   # first batch: a multiple of 16 individuals
        b.upload_bed_indiv_major(rows[:cut], 0)
        if cut < n:
            mid = cut + (n - cut) // 2 // 16 * 16
            b.upload_bed_indiv_major(rows[cut:mid], cut)
            b.upload_bed_indiv_major(rows[mid:], mid)
        for loc in range(l):
            assert np.array_equal(a.download_bed(loc), b.download_bed(loc)), loc
        want = np.array([(y == 0).sum(), (y == 3).sum(), (y == 1).sum(), (y == 2).sum()], dtype=np.uint64)  # codes 00 01 10 11
        assert np.array_equal(a.genotype_counts(), want) and np.array_equal(b.genotype_counts(), want)
        assert np.array_equal(a.genotype_counts(3, 2), np.array([(y[3:5] == v).sum() for v in (0, 3, 1, 2)], dtype=np.uint64))
        with pytest.raises(ts.TsamdError):
            b.upload_bed_indiv_major(rows[:8], 8)      # batches start on multiples of 16 individuals

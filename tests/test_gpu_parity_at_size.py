"""GPU parity at the BENCHMARKED launch geometry: the HIP path (through the C ABI) against the
CPU oracle at sizes where a thread of the plain pass owns several items (two-stage software
pipeline, backwards sweeps, 512-thread workgroups, several batches of partial rows) and the
first pass walks several individuals per thread -- BASELINE configs 4 and 5 on one GPU.

The oracle (oracle/ts_oracle.c, the reference's arithmetic: logsum softmax, k-outer/n-inner
sums) runs with all usable host cores in the reference's work partition
(src/snpsamplinge.cc:298-318, :337-352); its per-thread partial sums are added in chunk order,
which moves results by rounding only.

Tolerances: lambda and gamma rel 1e-9 after ~20 updates; inner-pass counts and c_n exact.
"""
import os

import numpy as np
import pytest

import oracle_py as op
from helpers import init_gamma, pack_bed, psd_genotypes, rel_err, usable_cores

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ts():
    import terastructure_amd as t

    t.load()
    return t


def _held_sets(y, seed, per_loc):
    """{loc: held-out individuals}: two locations, observed genotypes only"""
    rng = np.random.default_rng(seed)
    out = {}
    for loc in (1, 5):
        cand = np.nonzero(y[loc] != 3)[0]
        out[loc] = np.sort(rng.choice(cand, size=per_loc, replace=False)).astype(np.uint32)
    return out


# training updates (consecutive repeats included), one validation-mode update of a held-out
# location, training again: 17 + 1 + 4 updates = graphs of 16 + 1, 1, 4 SNPs
CALLS = [([3, 3, 5, 1, 6, 1, 2, 0, 7, 7, 4, 5, 3, 6, 2, 1, 0], 0), ([5], 1), ([6, 1, 6, 0], 0)]


def _launch_modes(ts, eng):
    """every launch mode the context qualifies for (the default one first)"""
    modes = [None]
    for m in (ts.LAUNCH_PER_SNP, ts.LAUNCH_PER_PASS):
        try:
            eng.set_launch_mode(m)
            modes.append(m)
        except ts.TsamdError:
            pass
    return modes


@pytest.mark.parametrize("n,k,thresh", [(100_000, 8, None), (300_000, 8, None), (1_000_000, 8, None), (125_000, 20, None),
                                        (300_000, 8, 20.0),   # ((1M, 20) -- config 5 on one GPU, ts_hybrid -- is tests/test_gpu_hybrid.py's, mode switches included)
                                        (600_000, 8, None), (600_000, 8, 30.0), (1_000_000, 8, 30.0),
                                        (500_000, 16, None), (250_000, 12, 30.0)])
def test_benchmarked_geometry_matches_oracle(ts, n, k, thresh):
    """Every kernel sequence the library can run at the benchmarked sizes, against the oracle: for each
    launch mode the context qualifies for -- one launch per schedule (ts_schedule, the default where the
    shard's weights fit the register file), per SNP (first pass + ts_resident: 6 items per thread at
    N = 600K, 8 at N = 1M) and per pass (ts_pass<K,true,256,1> + ts_pass<K,false,512,2>, hipGraph
    replay) -- run_schedule calls and eager snp_update calls.  `thresh` raises meanchangethresh so that
    SNPs stop after differing numbers of passes (1 ... 10).  N = 100K, K = 8 is BASELINE config 3's
    shape, N = 1M those of configs 4 and 5 on one GPU."""
    l = 8
    y, _, _ = psd_genotypes(n, l, k, 4000 + k, 0.02)
    payload = pack_bed(y)
    g = init_gamma(n, k, 4001 + k)
    held = _held_sets(y, 4002 + k, n // 100)
    del y
    over_o = {} if thresh is None else {"meanchangethresh": thresh}
    over_d = {} if thresh is None else {"conv_thresh": thresh}

    orc = op.Oracle(n, l, k, nthreads=usable_cores(), **over_o)
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    for loc, ids in held.items():
        orc.set_heldout(loc, ids)
    its = [orc.snp_update(loc, hol) for locs, hol in CALLS for loc in locs]
    lam_o, gam_o, cnt_o = orc.lambda_(), orc.gamma(), orc.c_indiv()
    orc.close()
    if thresh is None:
        assert set(its) == {10}
    else:
        assert len(set(its)) >= 3, f"pass counts do not vary: {sorted(set(its))}"

    with ts.Engine(n, l, k) as probe:
        launch_modes = _launch_modes(ts, probe)
    if k <= 8 and n <= 1_000_000:
        assert len(launch_modes) == 3, "K <= 8 at these sizes qualifies for every launch mode"
    per_mode = {}
    for launch in launch_modes:
        res = {}
        for mode in ("graph", "eager"):
            with ts.Engine(n, l, k, flags=0 if mode == "graph" else ts.FLAG_NO_GRAPH, **over_d) as eng:
                if launch is not None:
                    eng.set_launch_mode(launch)
                eng.upload_bed(payload)
                eng.set_gamma(g)
                for loc, ids in held.items():
                    eng.set_heldout(loc, ids)
                if mode == "graph":
                    for locs, hol in CALLS:
                        eng.run_schedule(np.array(locs, dtype=np.uint32), hol)
                    eng.synchronize()
                else:
                    its_d = [eng.snp_update(loc, hol) for locs, hol in CALLS for loc in locs]
                    assert its_d == its, (launch, mode)
                assert eng.total_passes() == sum(its)
                assert np.array_equal(eng.pass_histogram(), np.bincount(its, minlength=128).astype(np.uint64))
                lam_d, gam_d, cnt_d = eng.get_lambda(), eng.get_gamma(), eng.get_counts()
            assert np.array_equal(cnt_d, cnt_o), (launch, mode, "c_n")
            e_lam, e_gam = rel_err(lam_d, lam_o), rel_err(gam_d, gam_o)
            assert e_lam < 1e-9 and e_gam < 1e-9, (launch, mode, e_lam, e_gam)
            res[mode] = (lam_d, gam_d)
        # within a launch mode: one call per schedule == one call per update, bit for bit -- except in the default mode
        # from 4M weights per GPU on, where single-entry calls are routed through the launch-per-SNP kernels (tsamd.h)
        if launch is None and len(launch_modes) == 3 and n * k >= 4 << 20:
            assert rel_err(res["graph"][0], res["eager"][0]) < 1e-10 and rel_err(res["graph"][1], res["eager"][1]) < 1e-10
        else:
            assert np.array_equal(res["graph"][0], res["eager"][0]) and np.array_equal(res["graph"][1], res["eager"][1]), launch
        per_mode[launch] = res["graph"]
    # the modes differ by the order in which the workgroups' partial rows are added
    for launch, got in per_mode.items():
        assert rel_err(got[0], per_mode[None][0]) < 1e-10 and rel_err(got[1], per_mode[None][1]) < 1e-10, launch


def test_config3_full_size_properties(ts):
    """BASELINE config 3 at its full size -- N = 100 000 individuals x L = 500 000 SNPs, K = 8, the whole
    2-bit matrix (12.5 GB) resident -- where the oracle cannot follow: 3 000 updates over random locations,
    then size-independent properties of the path:
      * sum_{k,t} (lambda[loc][k][t] - eta_t) = 2 x (observed genotypes at loc) for every visited location
        (phi_mom and phi_dad each sum to 1 over k, src/snpsamplinge.cc:742-759), untouched locations keep eta;
      * c_n = the number of updates in which individual n was observed (update_rho_indiv, :705-719);
      * every gamma row sum follows S <- (1 - rho) S + rho (K alpha + 2 L), replayed
        on the host from c_n alone;
    and the oracle itself on 8 of the matrix' own columns: the same 24 updates from the same state."""
    n, l, k = 100_000, 500_000, 8
    rng = np.random.default_rng(303)
    theta = rng.dirichlet(np.full(k, 0.2), size=n)
    g0 = rng.gamma(100.0, 0.01, size=(n, k))
    with ts.Engine(n, l, k) as eng:
        assert eng.launch_info()["kernels_per_snp"] == 0   # ts_schedule
        chunk = 1 << 16
        brng = np.random.default_rng(304)
        for l0 in range(0, l, chunk):
            eng.synth_genotypes(theta, brng.uniform(0.05, 0.95, size=(min(chunk, l - l0), k)), first_loc=l0, seed=9,
                                missing_rate=0.01)
        eng.set_gamma(g0)
        locs = rng.integers(0, l, size=3000).astype(np.uint32)
        eng.run_schedule(locs)
        eng.synchronize()
        assert eng.total_passes() == 10 * len(locs)
        gam, cn = eng.get_gamma(), eng.get_counts()
        # lambda of visited / unvisited locations
        visited = np.unique(locs)
        for loc in visited[:: max(1, len(visited) // 40)]:
            cnt = eng.genotype_counts(int(loc), 1)
            observed = int(cnt[0] + cnt[2] + cnt[3])
            lam = eng.get_lambda(int(loc), 1)[0]
            assert abs(float(np.sum(lam - 1.0)) - 2.0 * observed) < 1e-7 * observed, loc
        unvisited = np.setdiff1d(np.arange(0, l, 9973), visited)[:20]
        for loc in unvisited:
            assert np.array_equal(eng.get_lambda(int(loc), 1)[0], np.ones((k, 2)))
        # c_n: observed updates per individual -- all but the last SNP's step have been applied (deferred)
        want_cn = np.zeros(n, dtype=np.uint32)
        for loc in locs[:-1]:
            bits = np.unpackbits(eng.download_bed(int(loc)), bitorder="little")[: 2 * n].reshape(n, 2)
            want_cn += ~((bits[:, 0] == 1) & (bits[:, 1] == 0))        # PLINK 01 (low bit first) = missing
        assert np.array_equal(cn, want_cn)
        # gamma row sums from c_n alone: rho_j = (tau0 + j)^-kappa for the j-th observed update
        target = k * (1.0 / k) + 2.0 * l
        s = g0.sum(axis=1)
        for j in range(int(cn.max())):
            rho = (2.0 + j) ** -0.5
            s = np.where(cn > j, (1.0 - rho) * s + rho * target, s)
        assert rel_err(gam.sum(axis=1), s) < 1e-10
        # the oracle on 8 of these columns, same state, same updates (gamma_scale = L like the engine)
        ls = 8
        sample = np.stack([eng.download_bed(j) for j in range(ls)])
        orc = op.Oracle(n, ls, k, nthreads=usable_cores(), gamma_scale=float(l))
        orc.load_bed_payload(sample)
        orc.set_gamma(g0)
        seq = [0, 3, 3, 1, 7, 5, 2, 6, 4, 0, 1, 1, 2, 7, 3, 5, 6, 4, 0, 2, 5, 7, 1, 3]
        its = [orc.snp_update(j) for j in seq]
        for j in range(ls):
            eng.set_lambda(j, np.ones((k, 2)))
        eng.set_gamma(g0)
        eng.set_counts(np.zeros(n, dtype=np.uint32))
        eng.clear_pending()
        p0 = eng.total_passes()
        eng.run_schedule(np.array(seq, dtype=np.uint32))
        eng.synchronize()
        assert eng.total_passes() - p0 == sum(its)
        assert rel_err(eng.get_lambda(0, ls), orc.lambda_()) < 1e-9
        assert rel_err(eng.get_gamma(), orc.gamma()) < 1e-9
        assert np.array_equal(eng.get_counts(), orc.c_indiv())
        orc.close()


@pytest.mark.parametrize("k,block", [(1, 256), (3, 512), (8, 256), (8, 512), (8, 1024), (12, 512), (16, 512), (17, 256),
                                     (20, 256), (32, 256)])
def test_multi_item_loops_every_instantiation(ts, k, block, monkeypatch):
    """Four workgroups for 20 000 individuals (TSAMD_GRID / TSAMD_GRID_FIRST = 4): every thread of
    the plain pass owns 5-10 items and of the first pass 20, at each workgroup size the library
    can pick -- the pipelined loop, its clamped prefetch and the backwards sweep of every
    K-specialised instantiation meet the oracle at a size the oracle finishes in a second."""
    monkeypatch.setenv("TSAMD_GRID", "4")
    monkeypatch.setenv("TSAMD_GRID_FIRST", "4")
    monkeypatch.setenv("TSAMD_BLOCK", str(block))
    n, l = 20_000, 12
    y, _, _ = psd_genotypes(n, l, k, 700 + k, 0.03)
    payload = pack_bed(y)
    g = init_gamma(n, k, 701 + k)
    orc = op.Oracle(n, l, k, nthreads=usable_cores())
    orc.load_bed_payload(payload)
    orc.set_gamma(g)
    locs = np.random.default_rng(k).integers(0, l, size=21).astype(np.uint32)
    locs[3] = locs[2]
    its = [orc.snp_update(int(loc)) for loc in locs]
    with ts.Engine(n, l, k) as eng:
        eng.upload_bed(payload)
        eng.set_gamma(g)
        eng.run_schedule(locs)
        eng.synchronize()
        assert eng.total_passes() == sum(its)
        assert np.array_equal(eng.get_counts(), orc.c_indiv())
        assert rel_err(eng.get_lambda(), orc.lambda_()) < 1e-9
        assert rel_err(eng.get_gamma(), orc.gamma()) < 1e-9


def test_clear_pending_then_schedule_equals_eager_bitwise(ts):
    """A single extra kernel between schedules (tsamd_clear_pending) shifts the launch parity of
    everything after it; the sweep direction of a pass follows its index within the SNP, not the
    launch parity, so graph replay still equals the eager sequence bit for bit (and an odd
    max_inner, where every SNP shifts the parity, does too)."""
    n, l, k = 600_000, 16, 8
    y, _, _ = psd_genotypes(n, l, k, 31, 0.02)
    payload = pack_bed(y)
    g = init_gamma(n, k, 32)
    del y
    locs_a = np.array([2, 9, 9, 4, 11], dtype=np.uint32)
    locs_b = np.random.default_rng(3).integers(0, l, size=19).astype(np.uint32)
    for max_inner in (10, 7):
        outs = []
        for flags in (0, ts.FLAG_NO_GRAPH):
            with ts.Engine(n, l, k, flags=flags, max_inner=max_inner) as eng:
                eng.upload_bed(payload)
                eng.set_gamma(g)
                eng.run_schedule(locs_a)
                eng.clear_pending()
                eng.run_schedule(locs_b)
                if flags == 0:
                    eng.prepare()  # dry replay of every graph with a gamma step pending: changes nothing
                eng.run_schedule(locs_b[:5])
                eng.clear_pending()
                eng.clear_pending()
                eng.run_schedule(locs_a[:3])
                eng.synchronize()
                outs.append((eng.get_lambda(), eng.get_gamma(), eng.get_counts(), eng.total_passes()))
        for a, b in zip(outs[0], outs[1]):
            assert np.array_equal(a, b), f"max_inner={max_inner}"


def test_config2_recovers_the_simulated_truth(ts):
    """BASELINE config 2 (synthetic PSD, N = 10 000 individuals x L = 100 000 SNPs, K = 6) end to end on
    the device: 60 000 SNP-minibatch updates from the reference's initialisation recover the simulated
    admixture proportions (best column permutation; the paper's Supp. Table 2 reports a median
    per-individual KL of 0.009-0.020 at this N) and allele frequencies; every gamma row sum has reached
    K alpha + 2 L (each step maps the sum S to (1-rho) S + rho (K alpha + 2 L))."""
    import itertools

    n, l, k = 10_000, 100_000, 6
    rng = np.random.default_rng(2024)
    theta = rng.dirichlet(np.full(k, 0.2), size=n)
    beta = rng.uniform(0.05, 0.95, size=(l, k))
    with ts.Engine(n, l, k) as eng:
        eng.synth_genotypes(theta, beta, seed=5)
        eng.set_gamma(rng.gamma(100.0, 0.01, size=(n, k)))
        eng.run_schedule(rng.integers(0, l, size=60_000).astype(np.uint32))
        eng.synchronize()
        th, gam = eng.get_theta(), eng.get_gamma()
        # beta of the locations visited last (their lambda saw the converged theta)
        eb = eng.get_ebeta(0, 2000)
    assert np.max(np.abs(gam.sum(1) - (k * (1.0 / k) + 2 * l))) < 1e-6 * 2 * l
    perm = min(itertools.permutations(range(k)), key=lambda p: np.mean((th[:, list(p)] - theta) ** 2))
    thp = th[:, list(perm)]
    rmse = float(np.sqrt(np.mean((thp - theta) ** 2)))
    kl = np.sum(theta * (np.log(theta + 1e-12) - np.log(thp + 1e-12)), axis=1)
    assert rmse < 0.03, rmse
    assert float(np.median(kl)) < 0.03, float(np.median(kl))
    visited = np.abs(eb - 0.5).sum(1) > 1e-9                      # locations updated at least once
    assert visited.sum() > 500
    assert float(np.sqrt(np.mean((eb[visited][:, list(perm)] - beta[:2000][visited]) ** 2))) < 0.08

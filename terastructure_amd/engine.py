"""Engine: one tsamd_ctx (one GPU, one shard of individuals).

Method names follow include/tsamd.h; array shapes follow the reference
(gamma/theta [n][k], lambda [l][k][2], Ebeta [l][k]), all float64 row-major.
"""
import ctypes as C

import numpy as np

from . import _lib


def shard_range(n, rank, world):
    """(begin, count) of rank's individuals -- tsamd_shard_range."""
    h = _lib.load()
    b, c = C.c_uint32(0), C.c_uint32(0)
    h.tsamd_shard_range(n, rank, world, C.byref(b), C.byref(c))
    return b.value, c.value


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def host_alloc(nbytes):
    """numpy uint8 view of pinned host memory (tsamd_host_alloc); free it with host_free(arr)"""
    h = _lib.load()
    ptr = C.c_void_p()
    rc = h.tsamd_host_alloc(C.byref(ptr), nbytes)
    if rc != 0:
        raise _lib.TsamdError(rc, h.tsamd_last_error(None).decode())
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))
    return arr


def host_free(arr):
    _lib.load().tsamd_host_free(C.c_void_p(arr.ctypes.data))


class Engine:
    def __init__(self, n, l, k, device=0, rank=0, world=1, flags=0, **overrides):
        self.h = _lib.load()
        cfg = _lib.Config()
        self.h.tsamd_default_config(C.byref(cfg), n, l, k)
        cfg.device, cfg.rank, cfg.world, cfg.flags = device, rank, world, flags
        for key, val in overrides.items():
            if not hasattr(cfg, key):
                raise AttributeError(f"tsamd_config has no field {key}")
            setattr(cfg, key, val)
        self.cfg = cfg
        self.n, self.l, self.k = n, l, k
        self.rank, self.world = rank, world
        self.shard_begin, self.shard_count = shard_range(n, rank, world)
        ctx = C.c_void_p()
        rc = self.h.tsamd_create(C.byref(cfg), C.byref(ctx))
        if rc != 0:
            raise _lib.TsamdError(rc, self.h.tsamd_last_error(None).decode())
        self.ctx = ctx

    # -- plumbing -----------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise _lib.TsamdError(rc, self.h.tsamd_last_error(self.ctx).decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.h.tsamd_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- genotypes ----------------------------------------------------------
    def upload_bed(self, payload, first_loc=0):
        """payload: uint8 [n_locs][ceil(n/4)] raw PLINK SNP-major bytes (global n)."""
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        if payload.ndim != 2:
            raise ValueError("payload must be [n_locs][bytes_per_snp]")
        self._check(self.h.tsamd_upload_bed(self.ctx, payload.ctypes.data, payload.shape[1], first_loc,
                                            payload.shape[0]))

    def upload_bed_async(self, ptr, bytes_per_snp, first_loc, n_locs):
        """payload at address ptr in pinned memory (host_alloc); valid until the next synchronize()"""
        self._check(self.h.tsamd_upload_bed_async(self.ctx, ptr, bytes_per_snp, first_loc, n_locs))

    def upload_bed_indiv_major(self, payload, first_indiv=0):
        """payload: uint8 [n_indivs][ceil(l/4)] PLINK individual-major rows (global individuals from first_indiv)"""
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        if payload.ndim != 2:
            raise ValueError("payload must be [n_indivs][bytes_per_indiv]")
        self._check(self.h.tsamd_upload_bed_indiv_major(self.ctx, payload.ctypes.data, payload.shape[1], first_indiv,
                                                        payload.shape[0]))

    def genotype_counts(self, first_loc=0, n_locs=None):
        """counts of the PLINK codes (00, 01 = missing, 10, 11) over the shard's individuals"""
        n_locs = self.l - first_loc if n_locs is None else n_locs
        out = np.zeros(4, dtype=np.uint64)
        self._check(self.h.tsamd_genotype_counts(self.ctx, first_loc, n_locs, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def download_bed(self, loc):
        out = np.zeros((self.shard_count + 3) // 4, dtype=np.uint8)
        self._check(self.h.tsamd_download_bed(self.ctx, loc, out.ctypes.data, out.size))
        return out

    def set_heldout(self, loc, indivs):
        a = np.ascontiguousarray(indivs, dtype=np.uint32)
        self._check(self.h.tsamd_set_heldout(self.ctx, loc, _up(a), a.size))

    def synth_genotypes(self, theta, beta, first_loc=0, seed=1, missing_rate=0.0):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        beta = np.ascontiguousarray(beta, dtype=np.float64)
        assert theta.shape == (self.shard_count, self.k) and beta.shape[1] == self.k
        self._check(self.h.tsamd_synth_genotypes(self.ctx, _dp(theta), _dp(beta), first_loc, beta.shape[0],
                                                 seed, missing_rate))

    # -- per-individual state -------------------------------------------------
    def set_gamma(self, gamma):
        g = np.ascontiguousarray(gamma, dtype=np.float64)
        if g.shape != (self.shard_count, self.k):
            raise ValueError(f"gamma must be [{self.shard_count}][{self.k}]")
        self._check(self.h.tsamd_set_gamma(self.ctx, _dp(g)))

    def _get_nk(self, fn):
        out = np.empty((self.shard_count, self.k), dtype=np.float64)
        self._check(fn(self.ctx, _dp(out)))
        return out

    def get_gamma(self):
        return self._get_nk(self.h.tsamd_get_gamma)

    def get_theta(self):
        return self._get_nk(self.h.tsamd_get_theta)

    def get_elogtheta(self):
        return self._get_nk(self.h.tsamd_get_elogtheta)

    def set_counts(self, c):
        a = np.ascontiguousarray(c, dtype=np.uint32)
        assert a.shape == (self.shard_count,)
        self._check(self.h.tsamd_set_counts(self.ctx, _up(a)))

    def get_counts(self):
        out = np.empty(self.shard_count, dtype=np.uint32)
        self._check(self.h.tsamd_get_counts(self.ctx, _up(out)))
        return out

    # -- per-location state ---------------------------------------------------
    def set_lambda(self, loc, lam):
        a = np.ascontiguousarray(lam, dtype=np.float64)
        if a.shape != (self.k, 2):
            raise ValueError("lambda must be [k][2]")
        self._check(self.h.tsamd_set_lambda(self.ctx, loc, _dp(a)))

    def get_lambda(self, first_loc=0, n_locs=None):
        n_locs = self.l - first_loc if n_locs is None else n_locs
        out = np.empty((n_locs, self.k, 2), dtype=np.float64)
        self._check(self.h.tsamd_get_lambda(self.ctx, first_loc, n_locs, _dp(out)))
        return out

    def get_ebeta(self, first_loc=0, n_locs=None):
        n_locs = self.l - first_loc if n_locs is None else n_locs
        out = np.empty((n_locs, self.k), dtype=np.float64)
        self._check(self.h.tsamd_get_ebeta(self.ctx, first_loc, n_locs, _dp(out)))
        return out

    def get_elogbeta(self, first_loc=0, n_locs=None):
        n_locs = self.l - first_loc if n_locs is None else n_locs
        out = np.empty((n_locs, self.k, 2), dtype=np.float64)
        self._check(self.h.tsamd_get_elogbeta(self.ctx, first_loc, n_locs, _dp(out)))
        return out

    # -- the hot path -----------------------------------------------------------
    def snp_update(self, loc, hol_mode=0):
        it = C.c_uint32(0)
        self._check(self.h.tsamd_snp_update(self.ctx, loc, int(hol_mode), C.byref(it)))
        return it.value

    def run_schedule(self, locs, hol_mode=0):
        a = np.ascontiguousarray(locs, dtype=np.uint32)
        self._check(self.h.tsamd_run_schedule(self.ctx, _up(a), a.size, int(hol_mode)))

    def synchronize(self):
        self._check(self.h.tsamd_synchronize(self.ctx))

    def prepare(self):
        """capture + instantiate the replayed hipGraphs now (otherwise the first run_schedule does)"""
        self._check(self.h.tsamd_prepare(self.ctx))

    def total_passes(self):
        v = C.c_uint64(0)
        self._check(self.h.tsamd_total_passes(self.ctx, C.byref(v)))
        return v.value

    def pass_histogram(self):
        """completed SNP updates by inner passes run (index = passes; last bin = that many or more)"""
        h = np.zeros(_lib.PASS_HIST_BINS, dtype=np.uint64)
        self._check(self.h.tsamd_pass_histogram(self.ctx, h.ctypes.data_as(C.POINTER(C.c_uint64))))
        return h

    def clear_pending(self):
        self._check(self.h.tsamd_clear_pending(self.ctx))

    def heldout_loglik(self, loc):
        s, c = C.c_double(0), C.c_uint32(0)
        self._check(self.h.tsamd_heldout_loglik(self.ctx, loc, C.byref(s), C.byref(c)))
        return s.value, c.value

    def heldout_eval(self, locs, run_updates=True):
        """(sum, count, per-location sums, per-location counts) of the held-out log-likelihood over
        locs -- tsamd_heldout_eval: the validation block of compute_likelihood in one call."""
        a = np.ascontiguousarray(locs, dtype=np.uint32)
        sums = np.zeros(a.size, dtype=np.float64)
        cnts = np.zeros(a.size, dtype=np.uint32)
        s, c = C.c_double(0), C.c_uint32(0)
        self._check(self.h.tsamd_heldout_eval(self.ctx, _up(a), a.size, int(run_updates), _dp(sums), _up(cnts),
                                              C.byref(s), C.byref(c)))
        return s.value, c.value, sums, cnts

    # -- multi-GPU --------------------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        rc = self.h.tsamd_comm_unique_id(buf)
        if rc != 0:
            raise _lib.TsamdError(rc, self.h.tsamd_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, uid):
        assert len(uid) == _lib.COMM_ID_BYTES
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(uid)
        self._check(self.h.tsamd_comm_init(self.ctx, buf))

    def p2p_export(self):
        buf = (C.c_uint8 * _lib.P2P_HANDLE_BYTES)()
        self._check(self.h.tsamd_p2p_export(self.ctx, buf))
        return bytes(buf)

    def p2p_connect(self, handles):
        """handles: list of world byte strings in rank order (tsamd_p2p_export of every rank)."""
        blob = b"".join(handles)
        assert len(blob) == self.world * _lib.P2P_HANDLE_BYTES
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        self._check(self.h.tsamd_p2p_connect(self.ctx, buf))

    @staticmethod
    def p2p_connect_local(engines):
        """Peer-to-peer exchange between engines of THIS process (ranks 0..world-1, one each).
        Afterwards enqueue the same run_schedule on every engine, then synchronize each;
        snp_update would wait for peers that have not been enqueued."""
        arr = (C.c_void_p * len(engines))(*[e.ctx for e in engines])
        rc = engines[0].h.tsamd_p2p_connect_local(arr, len(engines))
        engines[0]._check(rc)

    @staticmethod
    def run_schedule_all(engines, locs, hol_mode=0):
        """run_schedule on every engine of a p2p_connect_local group (bounded interleaving)."""
        a = np.ascontiguousarray(locs, dtype=np.uint32)
        arr = (C.c_void_p * len(engines))(*[e.ctx for e in engines])
        engines[0]._check(engines[0].h.tsamd_run_schedule_all(arr, len(engines), _up(a), a.size, int(hol_mode)))

    # -- measurement ------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.h.tsamd_profile_enable(self.ctx, int(on)))

    def profile_read(self):
        pn, fn = C.c_uint64(0), C.c_uint64(0)
        pm, fm = C.c_double(0), C.c_double(0)
        self._check(self.h.tsamd_profile_read(self.ctx, C.byref(pn), C.byref(pm), C.byref(fn), C.byref(fm)))
        return dict(pass_launches=pn.value, pass_ms=pm.value, first_launches=fn.value, first_ms=fm.value)

    def probe_stream(self, reps=50):
        """(read_us, rmw_us): bare streaming read of w / read-modify-write of w and gamma per launch"""
        a, b = C.c_double(0), C.c_double(0)
        self._check(self.h.tsamd_probe_stream(self.ctx, reps, C.byref(a), C.byref(b)))
        return a.value, b.value

    def launch_info(self):
        """dict(kernels_per_snp, plain_grid, first_grid) -- tsamd_launch_info"""
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._check(self.h.tsamd_launch_info(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return dict(kernels_per_snp=a.value, plain_grid=b.value, first_grid=c.value)

    def schedule_geometry(self, mode=_lib.LAUNCH_PER_SCHEDULE):
        """dict(workgroups, indivs_per_thread, exchange_levels, on_chip_per_thread) of the resident kernel of `mode` --
        tsamd_schedule_geometry (on_chip_per_thread < indivs_per_thread: ts_hybrid, part of the weights streamed)"""
        a, b, c, d = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._check(self.h.tsamd_schedule_geometry(self.ctx, int(mode), C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(workgroups=a.value, indivs_per_thread=b.value, exchange_levels=c.value, on_chip_per_thread=d.value)

    def holblock_info(self):
        """dict(batch, launches, locations) of the batched validation block -- tsamd_holblock_info"""
        a, b, c = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self.h.tsamd_holblock_info(self.ctx, C.byref(a), C.byref(b), C.byref(c)))
        return dict(batch=a.value, launches=b.value, locations=c.value)

    def set_launch_mode(self, mode):
        """LAUNCH_PER_PASS / LAUNCH_PER_SNP / LAUNCH_PER_SCHEDULE -- tsamd_set_launch_mode"""
        self._check(self.h.tsamd_set_launch_mode(self.ctx, int(mode)))

    def recoveries(self):
        """times a resident launch found its workgroups not all resident and the schedule was replayed launch per pass"""
        v = C.c_uint32(0)
        self._check(self.h.tsamd_recoveries(self.ctx, C.byref(v)))
        return v.value

    def debug_occupy(self, workgroups, milliseconds):
        """test aid: hold `workgroups` compute units for `milliseconds` with a kernel on a second stream"""
        self._check(self.h.tsamd_debug_occupy(self.ctx, int(workgroups), int(milliseconds)))

    def last_error(self):
        return self.h.tsamd_last_error(self.ctx).decode()

    def mem_info(self):
        f, t = C.c_uint64(0), C.c_uint64(0)
        self._check(self.h.tsamd_mem_info(self.ctx, C.byref(f), C.byref(t)))
        return f.value, t.value

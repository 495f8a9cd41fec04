"""ctypes view of oracle/libts_oracle.so (test infrastructure, never used by the product).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "libts_oracle.so")


class Config(C.Structure):
    _fields_ = [
        ("n", C.c_uint32), ("l", C.c_uint32), ("k", C.c_uint32),
        ("alpha", C.c_double), ("eta0", C.c_double), ("eta1", C.c_double),
        ("nodetau0", C.c_double), ("nodekappa", C.c_double),
        ("meanchangethresh", C.c_double), ("online_iterations", C.c_uint32),
        ("gamma_scale", C.c_double), ("nthreads", C.c_int),
    ]


class Rng(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("mti", C.c_int)]


class ValLine(C.Structure):
    _fields_ = [("iter", C.c_uint32), ("mean_ll", C.c_double), ("count", C.c_uint32)]


class RunParams(C.Structure):
    _fields_ = [
        ("seed", C.c_ulong), ("reportfreq", C.c_uint32), ("stop_threshold", C.c_double),
        ("max_iter", C.c_uint32), ("lines", C.POINTER(ValLine)), ("lines_cap", C.c_uint32),
        ("n_lines", C.c_uint32), ("final_iter", C.c_uint32), ("stopped", C.c_int),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, u32, u64, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double
    pd = C.POINTER(C.c_double)
    pu = C.POINTER(C.c_uint32)
    sig = {
        "orc_rng_seed": (None, [C.POINTER(Rng), C.c_ulong]),
        "orc_rng_get": (u32, [C.POINTER(Rng)]),
        "orc_rng_uniform_int": (u32, [C.POINTER(Rng), u32]),
        "orc_rng_uniform": (dbl, [C.POINTER(Rng)]),
        "orc_ran_gamma": (dbl, [C.POINTER(Rng), dbl, dbl]),
        "orc_digamma": (dbl, [dbl]),
        "orc_default_config": (None, [C.POINTER(Config), u32, u32, u32]),
        "orc_create": (vp, [C.POINTER(Config)]),
        "orc_destroy": (None, [vp]),
        "orc_load_bed_payload": (u64, [vp, vp, u64, u32, u32]),
        "orc_read_bed_file": (C.c_int, [vp, C.c_char_p]),
        "orc_y": (C.c_uint8, [vp, u32, u32]),
        "orc_set_heldout": (None, [vp, u32, pu, u32]),
        "orc_kv_ok": (C.c_int, [vp, u32, u32]),
        "orc_init_gamma": (None, [vp, C.POINTER(Rng)]),
        "orc_set_gamma": (None, [vp, pd]),
        "orc_init_lambda": (None, [vp]),
        "orc_set_lambda": (None, [vp, u32, pd]),
        "orc_set_validation_sample": (u32, [vp, C.POINTER(Rng)]),
        "orc_snp_update": (u32, [vp, u32, C.c_int]),
        "orc_pass_partial": (None, [vp, u32, u32, u32, pd]),
        "orc_epilogue": (dbl, [vp, u32, pd]),
        "orc_gamma_step": (None, [vp, u32]),
        "orc_heldout_loglik": (dbl, [vp, u32, pu]),
        "orc_estimate_beta": (None, [vp, u32]),
        "orc_gamma": (pd, [vp]), "orc_elogtheta": (pd, [vp]), "orc_etheta": (pd, [vp]),
        "orc_lambda": (pd, [vp]), "orc_elogbeta": (pd, [vp]), "orc_ebeta": (pd, [vp]),
        "orc_c_indiv": (pu, [vp]),
        "orc_n_heldout_locs": (u32, [vp]),
        "orc_heldout_locs": (u32, [vp, pu, u32]),
        "orc_heldout_indivs": (u32, [vp, u32, pu, u32]),
        "orc_run": (C.c_int, [vp, C.POINTER(RunParams)]),
        "orc_compute_all_lambda": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class Oracle:
    """One orc_state. Arrays returned are copies (row-major, reference shapes)."""

    def __init__(self, n, l, k, nthreads=1, **overrides):
        self.L = lib()
        self.cfg = Config()
        self.L.orc_default_config(C.byref(self.cfg), n, l, k)
        self.cfg.nthreads = nthreads
        for key, val in overrides.items():
            setattr(self.cfg, key, val)
        self.n, self.l, self.k = n, l, k
        self.s = self.L.orc_create(C.byref(self.cfg))

    def close(self):
        if self.s:
            self.L.orc_destroy(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # genotypes -----------------------------------------------------------
    def load_bed_payload(self, payload, first_loc=0):
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        n_locs, bps = payload.shape
        assert bps == (self.n + 3) // 4
        return self.L.orc_load_bed_payload(self.s, payload.ctypes.data, bps, first_loc, n_locs)

    def read_bed_file(self, path):
        rc = self.L.orc_read_bed_file(self.s, path.encode())
        if rc != 0:
            raise IOError(f"orc_read_bed_file({path}) failed")

    def set_heldout(self, loc, indivs):
        a = np.ascontiguousarray(indivs, dtype=np.uint32)
        self.L.orc_set_heldout(self.s, loc, _up(a), len(a))

    def heldout_locs(self):
        m = self.L.orc_n_heldout_locs(self.s)
        a = np.zeros(max(m, 1), dtype=np.uint32)
        self.L.orc_heldout_locs(self.s, _up(a), m)
        return a[:m]

    def heldout_indivs(self, loc):
        a = np.zeros(self.n, dtype=np.uint32)
        c = self.L.orc_heldout_indivs(self.s, loc, _up(a), self.n)
        return a[:c].copy()

    # state ----------------------------------------------------------------
    def set_gamma(self, g):
        g = np.ascontiguousarray(g, dtype=np.float64)
        assert g.shape == (self.n, self.k)
        self.L.orc_set_gamma(self.s, _dp(g))

    def set_lambda(self, loc, lam):
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        assert lam.shape == (self.k, 2)
        self.L.orc_set_lambda(self.s, loc, _dp(lam))

    def _arr(self, fn, shape):
        p = fn(self.s)
        return np.ctypeslib.as_array(p, shape=shape).copy()

    def gamma(self):
        return self._arr(self.L.orc_gamma, (self.n, self.k))

    def elogtheta(self):
        return self._arr(self.L.orc_elogtheta, (self.n, self.k))

    def theta(self):
        return self._arr(self.L.orc_etheta, (self.n, self.k))

    def lambda_(self):
        return self._arr(self.L.orc_lambda, (self.l, self.k, 2))

    def elogbeta(self):
        return self._arr(self.L.orc_elogbeta, (self.l, self.k, 2))

    def ebeta(self):
        return self._arr(self.L.orc_ebeta, (self.l, self.k))

    def c_indiv(self):
        p = self.L.orc_c_indiv(self.s)
        return np.ctypeslib.as_array(p, shape=(self.n,)).copy()

    # path -----------------------------------------------------------------
    def snp_update(self, loc, hol_mode=0):
        return self.L.orc_snp_update(self.s, loc, int(hol_mode))

    def pass_partial(self, loc, begin, end):
        out = np.zeros((self.k, 2), dtype=np.float64)
        self.L.orc_pass_partial(self.s, loc, begin, end, _dp(out))
        return out

    def epilogue(self, loc, lambdat):
        lt = np.ascontiguousarray(lambdat, dtype=np.float64)
        return self.L.orc_epilogue(self.s, loc, _dp(lt))

    def gamma_step(self, loc):
        self.L.orc_gamma_step(self.s, loc)

    def heldout_loglik(self, loc):
        c = C.c_uint32(0)
        v = self.L.orc_heldout_loglik(self.s, loc, C.byref(c))
        return v, c.value

    def run(self, seed, reportfreq, stop_threshold=1e-5, max_iter=0, lines_cap=4096):
        lines = (ValLine * lines_cap)()
        p = RunParams(seed=seed, reportfreq=reportfreq, stop_threshold=stop_threshold,
                      max_iter=max_iter, lines=lines, lines_cap=lines_cap)
        self.L.orc_run(self.s, C.byref(p))
        out = [(lines[i].iter, lines[i].mean_ll, lines[i].count) for i in range(min(p.n_lines, lines_cap))]
        return dict(lines=out, final_iter=p.final_iter, stopped=bool(p.stopped))

    def compute_all_lambda(self):
        self.L.orc_compute_all_lambda(self.s)


def gsl_mt19937(seed):
    r = Rng()
    lib().orc_rng_seed(C.byref(r), seed)
    return r

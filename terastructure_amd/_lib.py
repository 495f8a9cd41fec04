"""ctypes binding of include/tsamd.h (libtsamd.so).  No CPU fallback: if the
library is missing or no MI355X is visible, calls raise."""
import ctypes as C
import os

from . import build as _build

_HANDLE = None

COMM_ID_BYTES = 128
P2P_HANDLE_BYTES = 64
FLAG_SPLIT_EPILOGUE = 1
FLAG_NO_GRAPH = 2
FLAG_TEST_HOOKS = 4
LAUNCH_PER_PASS, LAUNCH_PER_SNP, LAUNCH_PER_SCHEDULE = 0, 1, 2  # tsamd_set_launch_mode
PASS_HIST_BINS = 128


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n", C.c_uint32), ("l", C.c_uint32), ("k", C.c_uint32),
        ("alpha", C.c_double), ("eta0", C.c_double), ("eta1", C.c_double),
        ("nodetau0", C.c_double), ("nodekappa", C.c_double),
        ("max_inner", C.c_uint32), ("conv_thresh", C.c_double), ("gamma_scale", C.c_double),
        ("device", C.c_int32), ("rank", C.c_uint32), ("world", C.c_uint32), ("flags", C.c_uint32),
    ]


class TsamdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"tsamd error {code}: {msg}")
        self.code = code


# every symbol include/tsamd.h declares: (restype, argtypes)
_vp, _u32, _u64, _dbl, _int = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double, C.c_int
_pd, _pu32, _pu8, _pu64 = C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
SYMBOLS = {
    "tsamd_abi_version": (_int, []),
    "tsamd_default_config": (None, [C.POINTER(Config), _u32, _u32, _u32]),
    "tsamd_shard_range": (None, [_u32, _u32, _u32, _pu32, _pu32]),
    "tsamd_create": (_int, [C.POINTER(Config), C.POINTER(_vp)]),
    "tsamd_destroy": (None, [_vp]),
    "tsamd_last_error": (C.c_char_p, [_vp]),
    "tsamd_upload_bed": (_int, [_vp, _vp, _u64, _u32, _u32]),
    "tsamd_upload_bed_async": (_int, [_vp, _vp, _u64, _u32, _u32]),
    "tsamd_upload_bed_indiv_major": (_int, [_vp, _vp, _u64, _u32, _u32]),
    "tsamd_host_alloc": (_int, [C.POINTER(_vp), _u64]),
    "tsamd_host_free": (None, [_vp]),
    "tsamd_genotype_counts": (_int, [_vp, _u32, _u32, _pu64]),
    "tsamd_download_bed": (_int, [_vp, _u32, _vp, _u64]),
    "tsamd_set_heldout": (_int, [_vp, _u32, _pu32, _u32]),
    "tsamd_set_gamma": (_int, [_vp, _pd]),
    "tsamd_get_gamma": (_int, [_vp, _pd]),
    "tsamd_get_theta": (_int, [_vp, _pd]),
    "tsamd_get_elogtheta": (_int, [_vp, _pd]),
    "tsamd_set_counts": (_int, [_vp, _pu32]),
    "tsamd_get_counts": (_int, [_vp, _pu32]),
    "tsamd_set_lambda": (_int, [_vp, _u32, _pd]),
    "tsamd_get_lambda": (_int, [_vp, _u32, _u32, _pd]),
    "tsamd_get_ebeta": (_int, [_vp, _u32, _u32, _pd]),
    "tsamd_get_elogbeta": (_int, [_vp, _u32, _u32, _pd]),
    "tsamd_snp_update": (_int, [_vp, _u32, _int, _pu32]),
    "tsamd_run_schedule": (_int, [_vp, _pu32, _u32, _int]),
    "tsamd_synchronize": (_int, [_vp]),
    "tsamd_prepare": (_int, [_vp]),
    "tsamd_total_passes": (_int, [_vp, _pu64]),
    "tsamd_pass_histogram": (_int, [_vp, _pu64]),
    "tsamd_clear_pending": (_int, [_vp]),
    "tsamd_heldout_loglik": (_int, [_vp, _u32, _pd, _pu32]),
    "tsamd_heldout_eval": (_int, [_vp, _pu32, _u32, _int, _pd, _pu32, _pd, _pu32]),
    "tsamd_comm_unique_id": (_int, [_pu8]),
    "tsamd_comm_init": (_int, [_vp, _pu8]),
    "tsamd_p2p_export": (_int, [_vp, _pu8]),
    "tsamd_p2p_connect": (_int, [_vp, _pu8]),
    "tsamd_p2p_connect_local": (_int, [C.POINTER(_vp), _u32]),
    "tsamd_run_schedule_all": (_int, [C.POINTER(_vp), _u32, _pu32, _u32, _int]),
    "tsamd_synth_genotypes": (_int, [_vp, _pd, _pd, _u32, _u32, _u64, _dbl]),
    "tsamd_profile_enable": (_int, [_vp, _int]),
    "tsamd_profile_read": (_int, [_vp, _pu64, _pd, _pu64, _pd]),
    "tsamd_probe_stream": (_int, [_vp, _u32, _pd, _pd]),
    "tsamd_launch_info": (_int, [_vp, _pu32, _pu32, _pu32]),
    "tsamd_schedule_geometry": (_int, [_vp, _int, _pu32, _pu32, _pu32, _pu32]),
    "tsamd_holblock_info": (_int, [_vp, _pu32, _pu64, _pu64]),
    "tsamd_set_launch_mode": (_int, [_vp, _int]),
    "tsamd_recoveries": (_int, [_vp, _pu32]),
    "tsamd_debug_occupy": (_int, [_vp, _u32, _u32]),
    "tsamd_mem_info": (_int, [_vp, _pu64, _pu64]),
}


def lib_path():
    """libtsamd.so of this tree (TSAMD_LIB overrides it: kernel-variant experiments, tools/variant.sh)."""
    return os.environ.get("TSAMD_LIB") or _build.LIB_PATH


def load():
    """Loads libtsamd.so.  torch (if importable) is imported first so that one
    process holds one HIP runtime and one RCCL (torch ships its own copies with
    the same SONAMEs)."""
    global _HANDLE
    if _HANDLE is not None:
        return _HANDLE
    if not os.environ.get("TSAMD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build it with `python -m terastructure_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback.")
    h = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(h, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if h.tsamd_abi_version() != 1:
        raise RuntimeError("libtsamd ABI version mismatch")
    _HANDLE = h
    return h

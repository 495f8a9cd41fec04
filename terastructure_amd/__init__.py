"""MI355X-native SNP-minibatch SVI engine for TeraStructure's SNPSamplingE path.

The product is libtsamd.so (HIP kernels behind the C ABI in include/tsamd.h) and
the C++ host under host/.  This package is the ctypes view of that ABI used by
tests/, bench.py and __graft_entry__.py.
"""
from ._lib import (Config, TsamdError, load, lib_path, FLAG_SPLIT_EPILOGUE, FLAG_NO_GRAPH, FLAG_TEST_HOOKS,  # noqa: F401
                   LAUNCH_PER_PASS, LAUNCH_PER_SNP, LAUNCH_PER_SCHEDULE)
from .engine import Engine, shard_range  # noqa: F401

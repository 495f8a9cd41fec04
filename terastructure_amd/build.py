"""Builds libtsamd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libtsamd.so")
SOURCES = ["tsamd.hip"]
DEPS = ["tsamd.hip", "tsamd_kernels.h", "tsamd_device.h", os.path.join(ROOT, "include", "tsamd.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for d in DEPS:
        path = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.getmtime(path) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

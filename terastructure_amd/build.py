"""Builds libtsamd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

The K-specialised kernels are compiled as one translation unit per K
(csrc/tsamd_inst.hip, csrc/tsamd_sched.hip, csrc/tsamd_hol.hip, csrc/tsamd_hyb.hip and csrc/tsamd_hhol.hip with -DTSAMD_K=k), in parallel; objects are cached under
terastructure_amd/lib/obj and rebuilt when a source they include changes.
"""
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libtsamd.so")
MAX_K = 32
SCHED_MAX_K = 32  # kResidentMaxK (csrc/tsamd_resident_kernels.h)
HEADERS = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [os.path.join(ROOT, "include", "tsamd.h")]
# -amdgpu-kernarg-preload-count: the leading scalar kernel arguments (ts_pass: control block, partial rows, weights,
# geometry) arrive in SGPRs with the wave instead of through a kernel-argument load (gfx950 supports it)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("TSAMD_EXTRA_HIPCC_FLAGS", "").split()


def kernel_sources_sha():
    """sha256 (16 hex digits) of the device sources the measured kernels are compiled from.  Stored with every counter
    record in profiles/pass_kernel_pmc.json (tools/pmc_record.py); bench.py only uses a record's traffic / flops / latency
    figures when the hash still matches -- a kernel change without re-profiling falls back to the hand count, visibly."""
    import hashlib

    h = hashlib.sha256()
    # every device header, the units that pick kernels and launch geometries (csrc/*.hip) and the ABI header -- the advisor's
    # round-5 finding: tsamd_generic_kernels.h / tsamd_wide_kernels.h were not covered
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".hip")))
    for path in [os.path.join(CSRC, f) for f in names] + [os.path.join(ROOT, "include", "tsamd.h")]:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _units():
    """(object path, source path, extra flags)"""
    units = [(os.path.join(OBJ_DIR, "tsamd.o"), os.path.join(CSRC, "tsamd.hip"), [])]
    for k in range(1, MAX_K + 1):
        units.append((os.path.join(OBJ_DIR, f"inst_k{k}.o"), os.path.join(CSRC, "tsamd_inst.hip"),
                      [f"-DTSAMD_K={k}"]))
    # the whole-schedule kernel: its own units, without machine LICM (see csrc/tsamd_sched.hip)
    for k in range(1, SCHED_MAX_K + 1):
        units.append((os.path.join(OBJ_DIR, f"sched_k{k}.o"), os.path.join(CSRC, "tsamd_sched.hip"),
                      [f"-DTSAMD_K={k}", "-mllvm", "-disable-machine-licm"]))
    # the batched validation-mode kernel (csrc/tsamd_hol.hip): same structure, same reason
    for k in range(1, SCHED_MAX_K + 1):
        units.append((os.path.join(OBJ_DIR, f"hol_k{k}.o"), os.path.join(CSRC, "tsamd_hol.hip"),
                      [f"-DTSAMD_K={k}", "-mllvm", "-disable-machine-licm"]))
    # the above-capacity whole-schedule kernel (csrc/tsamd_hyb.hip)
    for k in range(1, SCHED_MAX_K + 1):
        units.append((os.path.join(OBJ_DIR, f"hyb_k{k}.o"), os.path.join(CSRC, "tsamd_hyb.hip"),
                      [f"-DTSAMD_K={k}", "-mllvm", "-disable-machine-licm"]))
    # the batched validation-mode kernel for shards above the register capacity (csrc/tsamd_hhol.hip)
    for k in range(1, SCHED_MAX_K + 1):
        units.append((os.path.join(OBJ_DIR, f"hhol_k{k}.o"), os.path.join(CSRC, "tsamd_hhol.hip"),
                      [f"-DTSAMD_K={k}", "-mllvm", "-disable-machine-licm"]))
    return units


_DEPS = {}


def _deps(src):
    """src and the project headers it includes, transitively (#include "..." lines): a change to the hybrid kernel's header
    rebuilds the hybrid units and tsamd.hip, not all 129 units"""
    if src in _DEPS:
        return _DEPS[src]
    seen, todo = {src}, [src]
    while todo:
        f = todo.pop()
        try:
            text = open(f).read()
        except OSError:
            # a dependency that cannot be read: depend on every header rather than on none
            seen.update(HEADERS)
            continue
        for line in text.splitlines():
            m = re.match(r'\s*#\s*include\s*"([^"]+)"', line)
            if m is None:
                if re.match(r"\s*#\s*include\s+[A-Za-z_]", line):  # an include through a macro: cannot be resolved here
                    seen.update(HEADERS)
                continue
            name = m.group(1)
            for d in (CSRC, os.path.join(ROOT, "include")):
                cand = os.path.join(d, name)
                if os.path.exists(cand):
                    if cand not in seen:
                        seen.add(cand)
                        todo.append(cand)
                    break
            else:  # a project-style include that resolves to no file (generated header?): be conservative
                seen.update(HEADERS)
    _DEPS[src] = sorted(seen)
    return _DEPS[src]


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in _deps(src))


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(_stale(o, s) or os.path.getmtime(o) > t for o, s, _ in _units())


def build(force=False, verbose=False, jobs=None):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    jobs = jobs or min(8, os.cpu_count() or 1)
    todo = [(o, s, x) for o, s, x in _units() if force or _stale(o, s)]

    def compile_one(unit):
        obj, src, extra = unit
        cmd = [hipcc, "-c"] + FLAGS + extra + ["-o", obj, src]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout)
        return obj

    if verbose:
        print(f"[tsamd build] compiling {len(todo)} unit(s) with {jobs} job(s) for gfx950", file=sys.stderr)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + [o for o, _, _ in _units()] + ["-ldl"]
    if verbose:
        print("[tsamd build] linking " + LIB_PATH, file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

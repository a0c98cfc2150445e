"""Builds the native pieces in-tree with hipcc (cross-compiles for gfx950 without a GPU).

    python -m rpsmf_amd.build            # libpsmf_hip.so  (only what is stale)
    python -m rpsmf_amd.build --force    # everything from source

The library is three translation units, compiled side by side and linked: the C ABI with the two-launch and blocked engines
(psmf_capi.hip and what it includes), the persistent per-step engine (psmf_pstep.hip), and the build identity (psmf_buildid.cpp:
`psmf_build_id()` returns the SHA-256 of every source file and of the compiler flags the library was built from).  `build_library`
rebuilds whenever that hash differs from the sources on disk -- modification times play no part -- and `source_hash()` /
`library_build_id()` let `__graft_entry__.build()` prove that the shipped binary is the compiled form of the shipped sources.
"""

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libpsmf_hip.so")
HEADER = os.path.join(ROOT, "include", "psmf_hip.h")

# -amdgpu-mfma-vgpr-form: MFMAs with VGPR accumulators in every kernel.  By default the compiler takes that form only
# when a wave is limited to 256 registers (512-thread workgroups); the 256-thread kernels got accumulators in AGPRs and a
# copy in and out around every dependent MFMA.  With the flag they keep all 512 registers AND the short form, and what
# does not fit the 256 VGPRs is parked in AGPRs (one instruction) instead of scratch memory.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form"]
FLAGS += os.environ.get("PSMF_CXXFLAGS", "").split()      # diagnostic builds (e.g. -DPSTEP_PROF); part of the build id
LINK = ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]

_PSTEP_FILES = ("psmf_pstep.hip", "psmf_pstep.h", "psmf_ns.hip", "psmf_device.h")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))


def _hash_files(paths, extra=""):
    h = hashlib.sha256()
    h.update((" ".join(FLAGS + LINK) + extra).encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def source_hash():
    """SHA-256 over every file of rpsmf_amd/csrc, include/psmf_hip.h and the compiler flags."""
    return _hash_files(_sources() + [HEADER])


def _units():
    """(object name, source, files whose content decides whether the object is stale, extra flags)"""
    srcs = _sources()
    pstep_deps = [os.path.join(CSRC, f) for f in _PSTEP_FILES]
    capi_deps = [p for p in srcs if os.path.basename(p) not in ("psmf_pstep.hip", "psmf_buildid.cpp")] + [HEADER]
    pstep = os.path.join(CSRC, "psmf_pstep.hip")
    return [
        ("psmf_capi.o", os.path.join(CSRC, "psmf_capi.hip"), capi_deps, []),
        ("psmf_pstep.o", pstep, pstep_deps, []),          # persistent per-step kernel (+ its host entry points)
    ]


def library_build_id():
    """what the library on disk says it was built from ('' if it is missing or predates psmf_build_id); read from the file's
    bytes, not through dlopen: a process that has already mapped a stale library would be handed the same image again"""
    if not os.path.exists(LIB_PATH):
        return ""
    with open(LIB_PATH, "rb") as f:
        blob = f.read()
    i = blob.find(b"PSMF_BUILD_ID=")
    if i < 0:
        return ""
    return blob[i + 14:i + 14 + 64].decode("ascii", "replace")


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_library(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -> rpsmf_amd/lib/libpsmf_hip.so; returns its path.  Compiles nothing when the library's
    build id equals the hash of the sources on disk."""
    want = source_hash()
    if not force and library_build_id() == want:
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    inc = ["-I", os.path.join(ROOT, "include")]
    jobs = []
    for obj, src, deps, extra in _units():
        out = os.path.join(OBJ_DIR, obj)
        stamp = out + ".sha256"
        dep_hash = _hash_files(deps, " ".join(extra))
        fresh = (not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == dep_hash)
        if not fresh:
            jobs.append((out, stamp, dep_hash, [hipcc] + FLAGS + extra + inc + ["-c", src, "-o", out]))

    def compile_one(job):
        out, stamp, dep_hash, cmd = job
        if os.path.exists(stamp):
            os.remove(stamp)
        _run(cmd, verbose)
        with open(stamp, "w") as f:
            f.write(dep_hash)

    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as ex:
        list(ex.map(compile_one, jobs))
    idobj = os.path.join(OBJ_DIR, "psmf_buildid.o")
    _run([hipcc, "-O2", "-std=c++17", "-fPIC", "-x", "c++", f'-DPSMF_BUILD_ID="{want}"'] + inc + ["-c", os.path.join(CSRC, "psmf_buildid.cpp"), "-o", idobj], verbose)
    objs = [os.path.join(OBJ_DIR, u[0]) for u in _units()] + [idobj]
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + LINK, verbose)
    got = library_build_id()
    if got != want:
        raise RuntimeError(f"{LIB_PATH} reports build id {got!r}, expected {want!r}")
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(f"{LIB_PATH}: build id {library_build_id()[:16]}...")

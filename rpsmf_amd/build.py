"""Builds the native pieces in-tree with hipcc (cross-compiles for gfx950 without a GPU).

    python -m rpsmf_amd.build            # libpsmf_hip.so
"""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpsmf_hip.so")
SOURCES = [os.path.join(HERE, "csrc", "psmf_capi.hip")]
DEPS = [os.path.join(HERE, "csrc", f) for f in ("psmf_capi.hip", "psmf_kernels.hip", "psmf_block.hip", "psmf_impute.hip", "psmf_impute3.hip", "psmf_ns.hip", "psmf_blk3.hip", "psmf_blk4.hip", "psmf_blk16.hip", "psmf_blk32.hip", "psmf_masked.hip", "psmf_wave16.hip", "psmf_rotate.hip", "psmf_bulk.hip", "psmf_dyn.hip", "psmf_device.h")] + [
    os.path.join(ROOT, "include", "psmf_hip.h")
]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in DEPS if os.path.exists(p))


def build_library(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -shared -> rpsmf_amd/lib/libpsmf_hip.so"""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form: MFMAs with VGPR accumulators in every kernel.  By default the compiler takes that form only
    # when a wave is limited to 256 registers (512-thread workgroups); the 256-thread kernels got accumulators in AGPRs and a
    # copy in and out around every dependent MFMA.  With the flag they keep all 512 registers AND the short form, and what
    # does not fit the 256 VGPRs is parked in AGPRs (one instruction) instead of scratch memory.
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wno-unused-value",
           "-mllvm", "-amdgpu-mfma-vgpr-form", "-I", os.path.join(ROOT, "include"), "-o", LIB_PATH] + SOURCES + ["-L/opt/rocm/lib", "-lrccl",
           "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)

"""
Build the native pieces in-tree:

    python -m nellie_amd.build            # libnellie_hip.so for gfx950 (hipcc cross-compiles without a GPU)

`hipcc --offload-arch=gfx950 -O3 -ffp-contract=off`: contraction is OFF on purpose -- the
kernels reproduce numpy/scipy rounding points, an fma where numpy does mul-then-add changes bits.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnellie_hip.so")
SOURCES = ["nellie_hip.hip"]
# every include of the translation unit: a stale library after editing one of them would silently test old kernels
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".inc", ".h"))) + [os.path.join("..", "..", "include", "nellie_amd.h")]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the ROCm toolchain is required to build libnellie_hip.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
           "-fno-slp-vectorize",      # packed-float32 pairs cost the Hessian walk 27 register moves per voxel-plane (-1.1 ms/step)
           "-fPIC", "-shared", "-Wno-unused-value",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-ldl"]          # RCCL is dlopen()ed on first use (nl_comm_*), never linked
    if verbose:
        print("[nellie_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

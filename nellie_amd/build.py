"""
Build the native pieces in-tree:

    python -m nellie_amd.build            # libnellie_hip.so for gfx950 (hipcc cross-compiles without a GPU)
    python -m nellie_amd.build --variant NAME [-D...]   # nellie_amd/variants/libnellie_hip_NAME.so with extra flags (A/B builds)

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
# translation unit -> extra flags.  The pair walk has its own unit because it wants the ILP-first instruction scheduler, which costs the
# fused Gaussian pass 15 % (csrc/hv_launch.h); everything else is nellie_hip.hip.
SOURCES = {"nellie_hip.hip": [], "nellie_hv.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
# every include of the translation units: a stale library after editing one of them would silently test old kernels
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".inc", ".h"))) + [os.path.join("..", "..", "include", "nellie_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
         "-fno-slp-vectorize",      # packed-float32 pairs cost the Hessian walk 27 register moves per voxel-plane (-1.1 ms/step)
         "-fPIC", "-Wno-unused-value"]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the ROCm toolchain is required to build libnellie_hip.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in list(SOURCES) + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True, extra_flags=(), out: str = None) -> str:
    """Compiles the translation units side by side (objects under csrc/.obj, git-ignored) and links them.  extra_flags / out: A/B builds
    (tools/build_variant.sh)."""
    out = out or LIB
    if not force and out == LIB and not needs_build():
        return LIB
    objdir = os.path.join(CSRC, ".obj" if out == LIB else ".obj_" + os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src, flags in SOURCES.items():
        obj = os.path.join(objdir, src + ".o")
        cmd = [hipcc_path()] + FLAGS + list(flags) + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[nellie_amd.build]", " ".join(cmd), flush=True)
        procs.append((cmd, obj, subprocess.Popen(cmd)))
    objs = []
    for cmd, obj, proc in procs:
        if proc.wait() != 0:
            for _, _, other in procs:
                if other.poll() is None:
                    other.wait()
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        objs.append(obj)
    link = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs + ["-ldl"]   # RCCL is dlopen()ed on first use, never linked
    if verbose:
        print("[nellie_amd.build]", " ".join(link), flush=True)
    subprocess.check_call(link)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:
        k = sys.argv.index("--variant")
        name, extra = sys.argv[k + 1], sys.argv[k + 2:]
        vdir = os.path.join(HERE, "variants")
        os.makedirs(vdir, exist_ok=True)
        print(build(force=True, extra_flags=extra, out=os.path.join(vdir, f"libnellie_hip_{name}.so")))
    else:
        build(force="--force" in sys.argv)
        print(LIB)

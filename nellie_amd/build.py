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
# translation unit -> extra flags, compiled side by side.  The pair walk has its own unit because it wants the ILP-first instruction
# scheduler, which costs the fused Gaussian pass 15 % (csrc/hv_launch.h); Filter + comm (nellie_hip.hip), Label / Network / streaming
# (nellie_label.hip) and Markers (nellie_markers.hip) are separate so that an edit rebuilds one of them (nl_host.h holds what they share).
SOURCES = {"nellie_hip.hip": [], "nellie_gauss.hip": [], "nellie_gzyx.hip": [], "nellie_label.hip": [], "nellie_markers.hip": [],
           "nellie_hv.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
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


def _unit_fresh(obj: str, cmd) -> bool:
    """An object is reused when it is newer than every file its compiler-written dependency list (-MD) names and was made by the same
    command: an edit recompiles the translation units that include the edited file, nothing else."""
    dep, cmdf = obj + ".d", obj + ".cmd"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(cmdf)):
        return False
    if open(cmdf).read() != " ".join(cmd):
        return False
    t = os.path.getmtime(obj)
    words = open(dep).read().replace("\\\n", " ").split()
    files = [w for w in words[1:] if not w.endswith(":")]
    return all(os.path.exists(f) and os.path.getmtime(f) <= t for f in files)


def build(force: bool = False, verbose: bool = True, extra_flags=(), out: str = None) -> str:
    """Compiles the translation units side by side (objects under csrc/.obj, git-ignored) and links them.  extra_flags / out: A/B builds
    (tools/build_variant.sh)."""
    out = out or LIB
    if not force and out == LIB and not needs_build():
        return LIB
    objdir = os.path.join(CSRC, ".obj") if out == LIB else os.path.join(os.path.dirname(out), ".obj_" + os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    procs, objs_all = [], []
    for src, flags in SOURCES.items():
        obj = os.path.join(objdir, src + ".o")
        objs_all.append(obj)
        cmd = [hipcc_path()] + FLAGS + list(flags) + list(extra_flags) + ["-MD", "-MF", obj + ".d", "-c", os.path.join(CSRC, src), "-o", obj]
        if not force and _unit_fresh(obj, cmd):
            continue
        if verbose:
            print("[nellie_amd.build]", " ".join(cmd), flush=True)
        with open(obj + ".cmd", "w") as f:
            f.write(" ".join(cmd))
        procs.append((cmd, obj, subprocess.Popen(cmd)))
    objs = []
    for cmd, obj, proc in procs:
        if proc.wait() != 0:
            for _, _, other in procs:
                if other.poll() is None:
                    other.wait()
            if os.path.exists(obj):
                os.remove(obj)
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    objs = objs_all
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

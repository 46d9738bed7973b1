"""Build recipe for the HIP library: plain ``hipcc`` for gfx950, output in-tree (``lib/``).

``python -m gsplat_amd.build`` or ``gsplat_amd.build.build()``.  Cross-compiles without a GPU.
"""

from __future__ import annotations

import os
import subprocess
import sys

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # 3d-gaussian-splatting_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_DIR = os.path.join(PKG_ROOT, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgsplat_amd.so")
SOURCES = ["abi.hip", "project.hip", "sort.hip", "composite.hip", "train_step.hip", "view.hip"]
# -fno-slp-vectorize: the SLP pass pairs scalar fp32 ops into v_pk_* forms; on gfx950 a packed fp32 op issues
# at the cost of two scalar ones and the register shuffling around it (v_mov) is pure loss -- measured 8 % per step
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function", "-I" + os.path.join(REPO_ROOT, "include"), "-I" + CSRC]


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source to an object and link ``lib/libgsplat_amd.so``; returns its path."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("GSA_EXTRA_HIPCC_FLAGS", "").split()  # developer tuning, e.g. -DGSA_FWD_WAVES_PER_SIMD=6
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(REPO_ROOT, "include", f) for f in os.listdir(os.path.join(REPO_ROOT, "include"))]
    objs, relink = [], force
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        if force or _newer(s, o) or any(_newer(h, o) for h in headers):
            cmd = [hipcc, *FLAGS, *extra, "-c", s, "-o", o]
            if verbose:
                print("[gsplat_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            relink = True
        objs.append(o)
    if relink or not os.path.exists(LIB_PATH):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
        if verbose:
            print("[gsplat_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)

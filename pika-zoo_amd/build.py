"""Build libpikazoo_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

    python pika-zoo_amd/build.py [--force]

The shared library is plain HIP + a C ABI (include/pikazoo_hip.h); it links only against the
HIP runtime, not against torch.  hipcc cross-compiles without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_ROOT = Path(__file__).resolve().parent
REPO = PKG_ROOT.parent
CSRC = PKG_ROOT / "csrc"
INCLUDE = REPO / "include"
LIB_DIR = PKG_ROOT / "lib"
LIB = LIB_DIR / "libpikazoo_hip.so"
SOURCES = [CSRC / "pz_kernels.hip"]
DEPS = SOURCES + [CSRC / "pz_physics.hpp", INCLUDE / "pikazoo_hip.h"]
ARCH = "gfx950"
# the step kernels' five leading arguments (10 dwords) are preloaded into SGPRs at wave launch (pz_kernels.hip: HotArgs)
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-mllvm", "-amdgpu-kernarg-preload-count=10"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(d.stat().st_mtime > t for d in DEPS)


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> Path:
    if not force and not needs_build():
        return LIB
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    cmd = [hipcc_path(), *FLAGS, "-shared", "-fPIC",
           f"-I{INCLUDE}", f"-I{CSRC}", *extra_flags, "-o", str(LIB), *map(str, SOURCES)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

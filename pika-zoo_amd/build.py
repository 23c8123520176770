"""Build libpikazoo_hip.so -- and beside it libpikazoo_diag.so -- for gfx950 with hipcc (in-tree, no JIT cache).

    python pika-zoo_amd/build.py [--force]

The product library is plain HIP + a C ABI (include/pikazoo_hip.h); it links only against the
HIP runtime, not against torch.  hipcc cross-compiles without a GPU present.  The diagnostics library
(include/pikazoo_diag.h: pz_probe_launch, pz_selftest_predictor) is compiled from the product's own headers and is
loaded by bench.py and tests/ only (pika-zoo_amd/diag.py); it carries the same build id.
"""
from __future__ import annotations

import hashlib
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
DIAG_LIB = LIB_DIR / "libpikazoo_diag.so"
SOURCES = [CSRC / "pz_kernels.hip"]
DIAG_SOURCES = [CSRC / "pz_diag.hip"]
DEPS = SOURCES + DIAG_SOURCES + [CSRC / "pz_physics.hpp", CSRC / "pz_packed.hpp", CSRC / "pz_memory.hpp", CSRC / "pz_diagnostic.hpp",
                                 INCLUDE / "pikazoo_hip.h", INCLUDE / "pikazoo_diag.h"]
ARCH = "gfx950"
# the step kernels' six leading arguments (11 dwords) are preloaded into SGPRs at wave launch (pz_kernels.hip: HotArgs)
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-mllvm", "-amdgpu-kernarg-preload-count=11"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def source_id(extra_flags=()) -> str:
    """Digest of the sources and compiler flags the library is built from; baked into the library as
    ``pz_build_id()`` so that a stale prebuilt .so (git-ignored, but shipped to the GPU box) is never run.
    `extra_flags` are part of it: a variant compiled with other flags and copied to the product path carries
    another id than the product and is refused by ``_native.load()`` / rebuilt by ``needs_build()``."""
    h = hashlib.sha256()
    for d in DEPS:
        h.update(d.name.encode() + b"\0" + d.read_bytes() + b"\0")
    h.update(" ".join([*FLAGS, *extra_flags]).encode())
    return h.hexdigest()[:16]


_ID_MARKER = b"pz_build_id:"


def library_id(lib: Path = LIB):
    """The build id baked into a built library, or None (missing file / pre-build-id library).  Read from
    the file's bytes (the library stores it behind the marker ``pz_build_id:``), not through dlopen: a
    stale library mapped into this process would shadow the rebuilt one of the same name."""
    if not lib.exists():
        return None
    data = lib.read_bytes()
    at = data.find(_ID_MARKER)
    if at < 0:
        return None
    end = data.find(b"\0", at)
    return data[at + len(_ID_MARKER):end].decode(errors="replace")


def needs_build() -> bool:
    return library_id() != source_id() or library_id(DIAG_LIB) != source_id()


def _compile(out: Path, sources, extra_flags, verbose: bool) -> None:
    # compile to a temporary name and rename: other ranks / processes never see a half-written library
    tmp = out.with_suffix(f".so.tmp{os.getpid()}")
    cmd = [hipcc_path(), *FLAGS, "-shared", "-fPIC", f'-DPZ_BUILD_ID="{source_id(tuple(extra_flags))}"',
           f"-I{INCLUDE}", f"-I{CSRC}", *extra_flags, "-o", str(tmp), *map(str, sources)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, out)


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> Path:
    """The product library (returned) and the diagnostics library beside it (a few seconds)."""
    if any("PZ_DIAGNOSTIC_BUILD" in f for f in extra_flags):
        # csrc/pz_diagnostic.hpp: the one compile-time switch of the kernels -- tools/ab.py builds such variants into
        # tools/bin/; the product path never holds one (and _native.load() would refuse its build id "diagnostic")
        raise ValueError("a diagnostic build (-DPZ_DIAGNOSTIC_BUILD) is never written to the product library's path")
    want = source_id(tuple(extra_flags))
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    if force or extra_flags or library_id(DIAG_LIB) != want:
        _compile(DIAG_LIB, DIAG_SOURCES, extra_flags, verbose)
    if force or extra_flags or library_id() != want:
        _compile(LIB, SOURCES, extra_flags, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

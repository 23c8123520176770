#!/usr/bin/env python3
"""Digest of a kernel's instruction stream in a built library (diagnostic).

    python tools/kernel_digest.py                     # every step kernel of the product library: digest, instructions
    python tools/kernel_digest.py 'step_pair_kernel<false, false, false, false>'

The digest is the SHA-256 (16 hex digits) of the kernel's disassembled instruction texts -- mnemonics and operands,
without addresses, encodings or symbol offsets -- so two builds whose sources differ elsewhere (a comment, another
kernel, the host code) give the SAME digest for a kernel the change did not touch.  profiles/traffic.json stores it
beside the build id of the library the PMC counters were taken on; bench.py accepts a counter figure taken on another
build only if the kernel's digest is unchanged (`roofline.traffic_stale` otherwise).
"""
import hashlib
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
DEFAULT_LIB = REPO / "pika-zoo_amd" / "lib" / "libpikazoo_hip.so"
_cache = {}


def available() -> bool:
    return (LLVM / "llvm-objdump").exists()


def kernels(lib: Path = DEFAULT_LIB):
    """{demangled kernel name without its argument list: (digest, instruction count)} of the gfx950 code object."""
    lib = Path(lib).resolve()
    key = (str(lib), lib.stat().st_mtime_ns)
    if key in _cache:
        return _cache[key]
    with tempfile.TemporaryDirectory() as t:
        copy = shutil.copy(lib, Path(t) / "lib.so")  # the code objects are written next to the input file
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(copy)], check=True, cwd=t, capture_output=True)
        objs = sorted(Path(t).glob("*gfx950*"))
        if not objs:
            raise RuntimeError(f"no gfx950 code object in {lib}")
        asm = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--demangle", str(objs[0])], check=True,
                             capture_output=True, text=True).stdout
    out, name, h, count = {}, None, None, 0

    def close():
        if name is not None:
            out[name] = (h.hexdigest()[:16], count)

    for line in asm.splitlines():
        head = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if head:
            close()
            name = head.group(1).split("(")[0].replace("void ", "")
            h, count = hashlib.sha256(), 0
            continue
        m = re.match(r"^\s+([a-z_0-9]+(?: .*?)?)\s*//", line)
        if m and name is not None:
            text = re.sub(r"\s*<[^>]*>\s*$", "", m.group(1))  # (a branch target's symbol + offset)
            h.update(text.encode() + b"\n")
            count += 1
    close()
    _cache[key] = out
    return out


def digest(kernel: str, lib: Path = DEFAULT_LIB):
    """Digest of the kernel named exactly `kernel` (e.g. 'pz::step_pair_kernel<false, false, false, false>'), or None."""
    hit = kernels(lib).get(kernel)
    return hit[0] if hit else None


def main():
    table = kernels()
    want = sys.argv[1] if len(sys.argv) > 1 else None
    for name, (d, n) in sorted(table.items()):
        if (want is None and name.startswith(("pz::step_", "pz::rollout_pair"))) or (want is not None and want in name):
            print(f"{d}  {n:6d}  {name}")


if __name__ == "__main__":
    main()

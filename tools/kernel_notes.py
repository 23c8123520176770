#!/usr/bin/env python3
"""Register / LDS / spill / scratch figures of every kernel in a built library's gfx950 code object (diagnostic).

    python tools/kernel_notes.py [--lib path] [filter text ...]

One line per kernel from the code object's metadata note: VGPRs, SGPRs, SGPR spills, VGPR spills, private segment
(scratch) bytes, LDS bytes.  `notes(lib)` is what tests/test_cabi_and_host.py imports for its scratch / spill check.
"""
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
KEYS = (".vgpr_count", ".sgpr_count", ".sgpr_spill_count", ".vgpr_spill_count", ".private_segment_fixed_size",
        ".group_segment_fixed_size")


def notes(lib: Path):
    """[(demangled kernel name, {key: int})] of the library's gfx950 code object."""
    with tempfile.TemporaryDirectory() as t:
        copy = shutil.copy(lib, Path(t) / "lib.so")  # the code objects are written next to the input file
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(copy)], check=True, cwd=t, capture_output=True)
        objs = sorted(Path(t).glob("*gfx950*"))
        if not objs:
            raise SystemExit(f"no gfx950 code object in {lib}")
        text = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(objs[0])], check=True, capture_output=True,
                              text=True).stdout
    # one YAML list item per kernel: "- .agpr_count: ..." up to the next one
    blocks = re.split(r"\n\s*- \.agpr_count:", text)[1:]
    syms, rows = [], []
    for b in blocks:
        m = re.search(r"\.name:\s+'?([^'\n]+)'?\n", b)
        if not m:
            continue
        syms.append(m.group(1).strip())
        rows.append({k: int(v.group(1)) if (v := re.search(re.escape(k) + r":\s+(\d+)", b)) else 0 for k in KEYS})
    dem = subprocess.run(["c++filt", *syms], capture_output=True, text=True).stdout.splitlines()
    return list(zip(dem, rows))


def main():
    args = sys.argv[1:]
    lib = REPO / "pika-zoo_amd" / "lib" / "libpikazoo_hip.so"
    if "--lib" in args:
        i = args.index("--lib")
        lib = Path(args[i + 1])
        del args[i:i + 2]
    print(f"{'vgpr':>5} {'sgpr':>5} {'s-spill':>7} {'v-spill':>7} {'scratch':>7} {'lds':>6}  kernel")
    for name, r in notes(lib.resolve()):
        short = name.split("(")[0].replace("void pz::", "")
        if args and not any(a in short for a in args):
            continue
        print(f"{r['.vgpr_count']:5d} {r['.sgpr_count']:5d} {r['.sgpr_spill_count']:7d} {r['.vgpr_spill_count']:7d} "
              f"{r['.private_segment_fixed_size']:7d} {r['.group_segment_fixed_size']:6d}  {short}")


if __name__ == "__main__":
    main()

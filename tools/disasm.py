#!/usr/bin/env python3
"""Disassemble the gfx950 code object of a built library and print one kernel (diagnostic).

    python tools/disasm.py                       # list the kernels (demangled) with their sizes
    python tools/disasm.py 'step_kernel<false, false, 2' [--lib path] [--out file]
                                                 # the first kernel whose demangled name contains the text

Also prints, for the chosen kernel, the s_waitcnt vmcnt(...) sites and the register / LDS figures
of its kernel descriptor note (what decides the occupancy).
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")


def code_object(lib: Path, tmp: Path) -> Path:
    import shutil

    copy = shutil.copy(lib, tmp / "lib.so")  # the code objects are written next to the input file
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(copy)], check=True, cwd=tmp, capture_output=True)
    objs = sorted(tmp.glob("*gfx950*"))
    if not objs:
        raise SystemExit(f"no gfx950 code object in {lib}")
    return objs[0]


def main():
    args = sys.argv[1:]
    lib = REPO / "pika-zoo_amd" / "lib" / "libpikazoo_hip.so"
    out = None
    if "--lib" in args:
        i = args.index("--lib")
        lib = Path(args[i + 1])
        del args[i:i + 2]
    if "--out" in args:
        i = args.index("--out")
        out = Path(args[i + 1])
        del args[i:i + 2]
    with tempfile.TemporaryDirectory() as t:
        obj = code_object(lib.resolve(), Path(t))
        asm = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--demangle", str(obj)], check=True, capture_output=True,
                             text=True).stdout
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(obj)], capture_output=True, text=True).stdout
    # split into functions
    funcs = {}
    name = None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            name = m.group(1)
            funcs[name] = []
        elif name is not None:
            funcs[name].append(line)
    if not args:
        for nm, body in funcs.items():
            print(f"{len(body):7d}  {nm}")
        return
    want = args[0]
    hits = [nm for nm in funcs if want in nm]
    if not hits:
        raise SystemExit(f"no kernel matches {want!r}")
    nm = hits[0]
    body = funcs[nm]
    text = "\n".join(body)
    if out is not None:
        out.write_text(f"{nm}\n{text}\n")
    print(f"kernel: {nm}\ninstructions: {len(body)}")
    for key in ("s_waitcnt vmcnt", "s_barrier", "buffer_store", "buffer_load", "ds_write", "ds_read", "v_mad_u64_u32",
                "s_cbranch", "scratch_"):
        print(f"  {key:18s} {sum(key in ln for ln in body)}")
    # kernel descriptor facts from the code object's metadata note
    syms = re.findall(r"\.name:\s+'?([^'\n]+)'?\n", notes)
    dem = subprocess.run(["c++filt", *syms], capture_output=True, text=True).stdout.splitlines() if syms else []
    for sym, d in zip(syms, dem):
        if d == nm or d.split("(")[0] == nm.split("(")[0]:
            at = notes.index(sym)
            lo = notes.rfind("- .agpr_count", 0, at)
            hi = notes.find("- .agpr_count", at)
            block = notes[lo:hi if hi > 0 else len(notes)]
            for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".group_segment_fixed_size",
                        ".private_segment_fixed_size", ".vgpr_spill_count", ".sgpr_spill_count"):
                m = re.search(re.escape(key) + r":\s+(\d+)", block)
                if m:
                    print(f"  {key:28s} {m.group(1)}")
            break

if __name__ == "__main__":
    main()

"""ctypes binding of libpikazoo_diag.so (include/pikazoo_diag.h): DIAGNOSTICS for bench.py and tests/, not product.

Lives beside build.py, outside the pikazoo_amd package: nothing the package imports loads this library.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

PKG_ROOT = Path(__file__).resolve().parent
LIB_PATH = PKG_ROOT / "lib" / "libpikazoo_diag.so"

_P = C.c_void_p
SIGNATURES = {
    "pz_diag_build_id": (C.c_char_p, []),
    "pz_probe_launch": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "pz_selftest_predictor": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, _P, _P, _P]),
}
_lib = None


def load():
    """Load the diagnostics library (once); refuses one built from other sources than the product library."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python pika-zoo_amd/build.py`")
    lib = C.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    from pikazoo_amd import _native

    have, want = lib.pz_diag_build_id().decode(), _native.build_id()
    if have != want:
        raise RuntimeError(f"{LIB_PATH} was built from sources {have}, the product library from {want}: rebuild both with "
                           "`python pika-zoo_amd/build.py`")
    _lib = lib
    return lib

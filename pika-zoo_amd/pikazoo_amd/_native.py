"""ctypes binding of libpikazoo_hip.so (C ABI declared in include/pikazoo_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If it is missing or a
call fails, the error is raised to the caller.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

PKG_ROOT = Path(__file__).resolve().parent.parent  # .../pika-zoo_amd
LIB_PATH = PKG_ROOT / "lib" / "libpikazoo_hip.so"

ABI_VERSION = 10
PACKED_BYTES_PER_GAME = 36
SCENERY_WORDS = 75
STATE_WORDS = 44
OBS_DIM = 35
SERVE_MODES = {"winner": 0, "alternate": 1, "random": 2}
# pz_action_format: the element type of the action vectors a step launch reads as they are
ACTION_FORMATS = {"int32": 0, "int64": 1, "uint8": 2, "int16": 3}


class PzConfig(C.Structure):
    """`pz_config` (include/pikazoo_hip.h): the kwargs of pikazoo_v0.env (pikazoo_env.py:79-86),
    the two fusable wrappers, and the batched-env additions."""

    _fields_ = [
        ("winning_score", C.c_int32),
        ("serve_mode", C.c_int32),
        ("p1_computer", C.c_int32),
        ("p2_computer", C.c_int32),
        ("simplify_action", C.c_int32),
        ("ballpos_reward", C.c_int32),
        ("x_line", C.c_int32),
        ("y_line", C.c_int32),
        ("additional_reward", C.c_float * 8),
        ("auto_reset", C.c_int32),
        ("packed_state", C.c_int32),
        ("normal_state_mode", C.c_int32),
        ("normal_state_reward", C.c_float),
        ("normalize_obs", C.c_int32),
        ("episode_stats_mode", C.c_int32),
        ("seed", C.c_uint64),
        ("env_id_base", C.c_int64),
        ("action_faults", C.c_void_p),  # NULL or a device uint64 counter of out-of-range actions (pikazoo_env.py:182)
        ("action_format", C.c_int32),   # ACTION_FORMATS: element type of the action vectors
        ("reserved0", C.c_int32),
    ]


class PzFlightTables(C.Structure):
    """`pz_flight_tables` (include/pikazoo_hip.h): device pointers of the computer player's flight tables."""

    _fields_ = [("landing", C.c_void_p), ("power_hit", C.c_void_p)]  # either may be NULL


class PikazooNativeError(RuntimeError):
    pass


_P = C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "pz_abi_version": (C.c_int, []),
    "pz_state_words": (C.c_int, []),
    "pz_obs_dim": (C.c_int, []),
    "pz_config_bytes": (C.c_int, []),
    "pz_error_string": (C.c_char_p, [C.c_int]),
    "pz_build_id": (C.c_char_p, []),
    "pz_flight_table_bytes": (C.c_int64, [C.c_int32]),
    "pz_build_flight_tables": (C.c_int, [_P, _P, _P]),
    "pz_init": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(PzConfig), _P]),
    "pz_reset": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(PzConfig), _P, _P, _P, _P, _P]),
    "pz_observe": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P]),
    "pz_packed_state_bytes": (C.c_int64, [C.c_int64]),
    "pz_pack_state": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, _P, _P]),
    "pz_unpack_state": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, _P, _P]),
    # (the argument before the stream is `const pz_flight_tables*`: a byref(PzFlightTables) or None)
    "pz_step": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(PzConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pz_step_bound_bytes": (C.c_int64, []),
    "pz_step_bind": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.POINTER(PzConfig), _P, _P, _P, _P, _P, _P, _P]),
    "pz_step_bound": (C.c_int, [_P, _P, _P, _P]),
    "pz_count_packed_misfits": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "pz_probe_write": (C.c_int, [_P, _P, C.c_int64, _P]),
    "pz_probe_frame_bytes": (C.c_int64, []),
    "pz_step_random": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(PzConfig), C.c_uint64, C.c_uint64,
                                 C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pz_rollout_random": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(PzConfig), C.c_uint64, C.c_uint64,
                                    C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pz_step_many": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(PzConfig), _P, C.c_int32, _P, _P, _P, _P, _P, _P,
                               _P, _P, _P]),
    "pz_random_actions": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int32, _P]),
    "pz_scenery_init": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.POINTER(PzConfig), _P]),
    "pz_scenery_track": (C.c_int, [_P, _P, C.c_int64, C.c_int64, C.POINTER(PzConfig), C.c_int32, _P]),
    # (cfg may be None when no scenery is passed: ctypes passes NULL for None)
    "pz_render": (C.c_int, [_P, C.c_int64, C.c_int64, C.POINTER(PzConfig), _P, C.c_int64, _P, _P, _P, _P, _P, _P]),
}

_lib = None


def _pz_build():
    """pika-zoo_amd/build.py as a module (it lives beside the package, not inside it)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("pz_build", PKG_ROOT / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load():
    """Load the shared library (once).  Raises if it has not been built, or if it was built from other
    sources than the ones in this tree (the library is git-ignored but travels with the working tree,
    so a stale one would otherwise be run silently)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise PikazooNativeError(
            f"{LIB_PATH} is missing: build it with `python pika-zoo_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    if (PKG_ROOT / "csrc" / "pz_kernels.hip").exists():  # a source tree: the library must match it
        b = _pz_build()
        have, want = b.library_id(LIB_PATH), b.source_id()
        if have != want:
            raise PikazooNativeError(
                f"{LIB_PATH} is stale: built from sources {have}, the tree holds {want}; rebuild it with "
                "`python pika-zoo_amd/build.py`")
    lib = C.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.pz_abi_version() != ABI_VERSION:
        raise PikazooNativeError(f"ABI mismatch: library {lib.pz_abi_version()} != binding {ABI_VERSION}")
    if lib.pz_config_bytes() != C.sizeof(PzConfig) or lib.pz_state_words() != STATE_WORDS \
            or lib.pz_obs_dim() != OBS_DIM:
        raise PikazooNativeError("pz_config / state layout of the library does not match this binding")
    _lib = lib
    return lib


def build_id() -> str:
    """Source digest the loaded library was compiled from (``pz_build_id``)."""
    return load().pz_build_id().decode()


def exported_names():
    return list(_SIGNATURES)


def check(code: int, what: str):
    if code != 0:
        msg = load().pz_error_string(code)
        raise PikazooNativeError(f"{what} failed: {msg.decode() if msg else code} (code {code})")

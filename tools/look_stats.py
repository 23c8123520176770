#!/usr/bin/env python3
"""Diagnostic: how many of a config-3 launch's deciding lanes / waves are served by their look-ahead rows.
Build: python tools/ab.py --build --common "-DPZ_DEV_SUBSET=721" lookstats="-DPZ_LOOKAHEAD=1 -DPZ_LOOK_STATS"
Run on the GPU box: python tools/look_stats.py [lib name, default lookstats]"""
import ctypes as C
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "pika-zoo_amd"))
import torch
from pikazoo_amd import _native

name = sys.argv[1] if len(sys.argv) > 1 else "lookstats"
lib = C.CDLL(str(REPO / "pika-zoo_amd" / "lib" / f"ab_{name}.so"))
P = C.c_void_p
lib.pz_init.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P]
lib.pz_reset.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P, P, P, P, P]
lib.pz_step.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P, P, P, P, P, P, P, P, P, P]
lib.pz_flight_table_bytes.restype = C.c_int64
lib.pz_flight_table_bytes.argtypes = [C.c_int32]
lib.pz_build_flight_tables.argtypes = [P, P, P]
lib.pz_look_stats.argtypes = [P]
n = 65536
dev = torch.device("cuda:0")
cfg = _native.PzConfig()
cfg.winning_score, cfg.auto_reset, cfg.seed, cfg.p2_computer = 15, 1, 0, 1
state = torch.zeros((44, n), dtype=torch.int32, device=dev)
obs = [torch.zeros((n, 35), dtype=torch.int32, device=dev) for _ in range(2)]
rew = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
term = torch.zeros(n, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
t_land = torch.empty(lib.pz_flight_table_bytes(0), dtype=torch.uint8, device=dev)
t_hit = torch.empty(lib.pz_flight_table_bytes(1), dtype=torch.uint8, device=dev)
assert lib.pz_build_flight_tables(t_land.data_ptr(), t_hit.data_ptr(), stream) == 0
tables = _native.PzFlightTables(t_land.data_ptr(), t_hit.data_ptr())
assert lib.pz_init(state.data_ptr(), n, n, C.byref(cfg), stream) == 0
assert lib.pz_reset(state.data_ptr(), n, n, C.byref(cfg), None, obs[0].data_ptr(), obs[1].data_ptr(), None, stream) == 0
acts = torch.randint(0, 18, (64, 2, n), dtype=torch.int32, device=dev)
out = (C.c_ulonglong * 8)()
for phase, steps in (("first frame", 1), ("frames 2-600", 599), ("frames 601-1600", 1000)):
    for t in range(steps):
        a = acts[t % 64]
        assert lib.pz_step(state.data_ptr(), n, n, C.byref(cfg), a[0].data_ptr(), a[1].data_ptr(), obs[0].data_ptr(),
                           obs[1].data_ptr(), rew[0].data_ptr(), rew[1].data_ptr(), term.data_ptr(), None, C.byref(tables),
                           stream) == 0
    torch.cuda.synchronize()
    lib.pz_look_stats(out)
    lanes, served, ok, full, waves = out[0], out[1], out[2], out[3], out[4]
    print(f"{phase}: deciding lanes {lanes}, rows matching {ok / max(lanes, 1):.6f}, served {served / max(lanes, 1):.6f}; "
          f"waves fully served {full} of {waves} = {full / max(waves, 1):.4f}")

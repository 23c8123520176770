#!/usr/bin/env python3
"""Per-wave timeline of one step launch (diagnostic build, pz_diagnostic.hpp bit 0 = stamps; not product).

    python tools/stamps.py --build ; python tools/stamps.py [--ai [--tables]] [--packed] [--n N]     (second on the GPU box)

Stamps (100 MHz s_memrealtime, 10 ns ticks), lane 0 of every workgroup:
 0 start | 1 state loaded (vmcnt(0)) | 2 frame computed | 3 state/reward stores issued |
 4 observations staged in LDS | 5 observation stores issued | 6 all stores complete
Prints, relative to the earliest start of the launch, the median / p95 / max over waves of each stamp.
"""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "pika-zoo_amd"))
from build import FLAGS as PRODUCT_FLAGS  # noqa: E402  (same compiler flags as the product library)
LIB = REPO / "tools" / "bin" / "stamps.so"


def main():
    args = sys.argv[1:]
    if "--build" in args:
        LIB.parent.mkdir(parents=True, exist_ok=True)  # (tools/bin/ does not travel to the GPU box: build there)
        subprocess.check_call(["hipcc", *PRODUCT_FLAGS, "-shared", "-fPIC", "-DPZ_DIAGNOSTIC_BUILD=1u",
                               f"-I{REPO / 'include'}", f"-I{REPO / 'pika-zoo_amd' / 'csrc'}", "-o", str(LIB),
                               str(REPO / "pika-zoo_amd" / "csrc" / "pz_kernels.hip")])
        return
    import torch
    from pikazoo_amd import _native

    ai = "--ai" in args
    n = int(args[args.index("--n") + 1]) if "--n" in args else 65536
    lib = C.CDLL(str(LIB))
    P = C.c_void_p
    lib.pz_init.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P]
    lib.pz_reset.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P, P, P, P, P]
    lib.pz_step.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P, P, P, P, P, P, P, P, P, P]
    lib.pz_debug_read_stamps.argtypes = [P, C.c_int64]
    tb = None
    if "--tables" in args:
        lib.pz_flight_table_bytes.restype = C.c_int64
        lib.pz_flight_table_bytes.argtypes = [C.c_int32]
        lib.pz_build_flight_tables.argtypes = [P, P, P]
    dev = torch.device("cuda:0")
    cfg = _native.PzConfig()
    cfg.winning_score, cfg.auto_reset, cfg.seed, cfg.p2_computer = 15, 1, 0, int(ai)
    packed = "--packed" in args  # the 36-byte state format (n must be a multiple of 4 here: stride = n)
    cfg.packed_state = int(packed)
    state = torch.zeros(36 * n, dtype=torch.uint8, device=dev) if packed else torch.zeros((44, n), dtype=torch.int32, device=dev)
    obs = [torch.zeros((n, 35), dtype=torch.int32, device=dev) for _ in range(2)]
    rew = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
    term = torch.zeros(n, dtype=torch.uint8, device=dev)
    acts = torch.randint(0, 18, (64, 2, n), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    if "--tables" in args:
        t_land = torch.empty(lib.pz_flight_table_bytes(0), dtype=torch.uint8, device=dev)
        t_hit = torch.empty(lib.pz_flight_table_bytes(1), dtype=torch.uint8, device=dev)
        assert lib.pz_build_flight_tables(t_land.data_ptr(), t_hit.data_ptr(), stream) == 0
        tables = _native.PzFlightTables(t_land.data_ptr(), t_hit.data_ptr())
        tb = C.byref(tables)
    lib.pz_init(state.data_ptr(), n, n, C.byref(cfg), stream)
    lib.pz_reset(state.data_ptr(), n, n, C.byref(cfg), None, obs[0].data_ptr(), obs[1].data_ptr(), None, stream)

    def run(steps):
        for t in range(steps):
            a = acts[t % 64]
            lib.pz_step(state.data_ptr(), n, n, C.byref(cfg), a[0].data_ptr(), a[1].data_ptr(), obs[0].data_ptr(),
                        obs[1].data_ptr(), rew[0].data_ptr(), rew[1].data_ptr(), term.data_ptr(), None, tb, stream)

    run(800)
    torch.cuda.synchronize()
    pair = n < 393216 and (not ai or "--tables" in args)  # the two-wave launch: slot = 2 * workgroup + role
    waves = min((n + 63) // 64 * (2 if pair else 1), 8192)
    buf = np.zeros(8192 * 8, np.uint64)
    names = ["start", "loaded", "computed", "state stores issued", "obs staged", "obs stores issued", "all stores done"]
    for rep in range(3):
        run(20)  # back-to-back launches; the stamps of the last one survive
        torch.cuda.synchronize()
        assert lib.pz_debug_read_stamps(buf.ctypes.data, 8192 * 8) == 0
        st = buf.reshape(8192, 8)[:waves, :7].astype(np.int64)
        t0 = st[:, 0].min()
        rel = (st - t0) * 0.01  # microseconds
        print(f"launch sample {rep}: n={n} ai={ai} packed={packed} waves={waves}; microseconds since the first wave started")
        for k, nm in enumerate(names):
            c = rel[:, k]
            print(f"  {nm:22s} min {c.min():6.2f}  median {np.median(c):6.2f}  p95 {np.percentile(c, 95):6.2f}  max {c.max():6.2f}")
        d = np.diff(rel, axis=1)
        print("  per-wave phase durations (median): " + ", ".join(f"{names[k + 1]} {np.median(d[:, k]):.2f}" for k in range(6)))
        if pair:
            for role in (0, 1):
                dr = d[role::2]
                print(f"    wave of player {role + 1} (median): " + ", ".join(f"{names[k + 1]} {np.median(dr[:, k]):.2f}" for k in range(6)))
        if pair and "--where" in args:
            # which waves share a SIMD (HW_ID / XCC_ID of every wave, slot 7), and what that does to the frame
            where = buf.reshape(8192, 8)[:waves, 7].astype(np.uint64)
            hw, xcc = (where & np.uint64(0xFFFFFFFF)).astype(np.int64), (where >> np.uint64(32)).astype(np.int64) & 0xF
            simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
            home = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
            cuid = home // 4
            roles = np.arange(waves) % 2
            frame = rel[:, 2] - rel[:, 1]  # loaded -> computed
            import collections
            by_home = collections.defaultdict(list)
            for w in range(waves):
                by_home[int(home[w])].append(w)
            comp = collections.Counter(tuple(sorted(int(roles[w]) for w in ws)) for ws in by_home.values())
            print(f"    SIMDs in use {len(by_home)}, CUs {len(set(cuid.tolist()))}, XCCs {len(set(xcc.tolist()))}; "
                  f"roles sharing a SIMD: {dict(comp)}")
            same_wg = sum(1 for ws in by_home.values() for a in ws for b in ws if a < b and a // 2 == b // 2)
            print(f"    pairs of one workgroup on one SIMD: {same_wg}")
            stats = collections.defaultdict(list)
            for ws in by_home.values():
                for w in ws:
                    mates = tuple(sorted(int(roles[v]) for v in ws if v != w))
                    stats[(int(roles[w]), mates)].append(frame[w])
            for key in sorted(stats):
                v = np.array(stats[key])
                print(f"    role {key[0]} with SIMD mates {key[1]}: {len(v):5d} waves, loaded->computed median {np.median(v):5.2f} "
                      f"p95 {np.percentile(v, 95):5.2f} max {v.max():5.2f} us")
            wg_cu = collections.Counter(int(cuid[2 * g]) for g in range(waves // 2))
            print(f"    workgroups per CU: {dict(collections.Counter(wg_cu.values()))}")
            sample = [(g, int(xcc[2 * g]), int(se[2 * g]), int(cu[2 * g]), int(simd[2 * g]), int(simd[2 * g + 1])) for g in range(24)]
            print("    first workgroups (wg, xcc, se, cu, simd of wave 0, simd of wave 1): " + " ".join(map(str, sample)))
        fb = np.zeros(8192 * 8, np.uint64)
        lib.pz_debug_read_frame_stamps.argtypes = [P, C.c_int64]
        assert lib.pz_debug_read_frame_stamps(fb.ctypes.data, 8192 * 8) == 0
        fs = fb.reshape(8192, 8)[:(waves // 2 if pair else waves)].astype(np.int64)  # pair: lane 0 of the player-1 wave
        fd = np.diff(fs, axis=1)  # shader-clock cycles (s_memtime)
        fnames = (["round start", "action decode", "ball-world", "decision + own move", "exchange", "collisions + scoring",
                   "tail"] if pair else
                  ["round start", "action decode", "ball-world", "AI1 + player 1", "AI2 + player 2", "collisions", "scoring"])
        print("  frame sub-phases, shader cycles per wave (median / p95): " +
              ", ".join(f"{fnames[k]} {int(np.median(fd[:, k]))}/{int(np.percentile(fd[:, k], 95))}" for k in range(7)) +
              f"; whole frame {int(np.median(fs[:, 7] - fs[:, 0]))}")


if __name__ == "__main__":
    main()

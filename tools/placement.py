#!/usr/bin/env python3
"""Does the k-frame rollout's rate depend on WHERE its trajectory tensors live (diagnostic)?  One process, one box:
pz_rollout_random (65 536 games) into (a) tensors allocated one by one (torch.empty: what the env API does),
(b) the same tensors carved out of ONE large arena, at several arena sizes / alignments, (c) again one by one,
each through a hipGraph of >= 2 048 frames, timed for `--seconds`.

    python tools/placement.py [--k 32] [--seconds 0.5] [--p2-computer]
"""
import argparse
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "pika-zoo_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, nargs="+", default=[32, 128])
    ap.add_argument("--seconds", type=float, default=0.5)
    ap.add_argument("--num-envs", type=int, default=65536)
    ap.add_argument("--p2-computer", action="store_true")
    ap.add_argument("--arena-mib", type=int, default=6144)
    ap.add_argument("--bases", type=int, nargs="+", default=None, help="only: the arena set at these byte offsets")
    ap.add_argument("--tables-align", action="store_true", help="only (with --p2-computer): pz_step with the flight tables in "
                    "plain allocations vs at a 1 GiB boundary of one large allocation (larger page-table fragments)")
    ap.add_argument("--tables-rank", action="store_true", help="only (with --p2-computer): the k-frame launch with the flight "
                    "tables in the rank of obs1, of obs2, and in the third rank")
    ap.add_argument("--step-spread", action="store_true", help="only: pz_step (single frame) with state / obs1 / obs2 in one "
                    "rank of the device memory vs spread over three, at --num-envs")
    ap.add_argument("--idle", type=float, default=0.0, help="with --product: seconds of host-side pause in front of every "
                    "allocation (the probe must not be fooled by the low clocks after an idle spell)")
    ap.add_argument("--obs16", action="store_true", help="int16 observations (with --product)")
    ap.add_argument("--product", action="store_true", help="only: the env's own allocation, with and without placement")
    ap.add_argument("--many", type=int, default=0, help="only: this many separately allocated sets per k, all kept alive")
    ap.add_argument("--replays", type=int, default=0, help="a fixed number of timed replays per case (under rocprofv3 --pmc: "
                    "tools/pmc_cases.py then groups the dispatches by case), instead of --seconds")
    ap.add_argument("--gaps", type=int, nargs="+",
                    default=[0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 10 << 20, 16 << 20,
                             (2 << 20) + 4096, (32 << 20) + 256, 34 << 20])
    args = ap.parse_args()
    import torch
    from pikazoo_amd import _native, pikazoo_v0

    dev = torch.device("cuda:0")
    env = pikazoo_v0.env(num_envs=args.num_envs, device="cuda:0", seed=0, is_player2_computer=args.p2_computer)
    raw = env.unwrapped
    env.reset()
    lib = _native.load()
    n = raw.num_envs
    keep = []  # every allocation stays alive: later cases get other memory

    def shapes(k):
        return [("actions", (k, 2, n), torch.int32), ("obs1", (k, n, 35), torch.int32), ("obs2", (k, n, 35), torch.int32),
                ("rew1", (k, n), torch.int32), ("rew2", (k, n), torch.int32), ("term", (k, n), torch.uint8)]

    def separate(k):
        t = {nm: torch.empty(shp, dtype=dt, device=dev) for nm, shp, dt in shapes(k)}
        keep.append(t)
        return t

    arena = torch.empty(args.arena_mib << 20, dtype=torch.uint8, device=dev)
    arena_off0 = (-arena.data_ptr()) % (1 << 30)  # carve from a 1 GiB boundary

    def carved(k, gap, order=None, base=0):
        """the six tensors one behind the other in the arena, `gap` bytes (plus rounding to 256) between them"""
        off = arena_off0 + base
        t = {}
        sh = shapes(k)
        if order:
            sh = [sh[i] for i in order]
        for nm, shp, dt in sh:
            nbytes = torch.empty((), dtype=dt).element_size()
            for d in shp:
                nbytes *= d
            assert off + nbytes <= arena.numel(), "arena too small"
            t[nm] = arena[off:off + nbytes].view(dt).view(shp)
            off = (off + nbytes + gap + 255) // 256 * 256
        return t

    def time_case(tag, k, t, raw=raw):
        # (the tensors must be the row format of `raw`'s configuration: int16 rows are half the size of int32 ones)
        assert t["obs1"].dtype == raw._traj_obs_dtype() and t["obs1"].shape == (k, raw.num_envs, 35)
        n = raw.num_envs
        launches = max(64, 2048 // k)
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                cs = torch.cuda.current_stream().cuda_stream
                for j in range(launches):
                    rc = lib.pz_rollout_random(raw._state_ptr, n, raw._stride, raw._cfg_ref, 1, j * k, k,
                                               t["actions"].data_ptr(), t["obs1"].data_ptr(), t["obs2"].data_ptr(),
                                               t["rew1"].data_ptr(), t["rew2"].data_ptr(), t["term"].data_ptr(), None,
                                               raw._episodes.data_ptr(), raw._tables_ref, cs)
                    assert rc == 0, rc
            g.replay()
            side.synchronize()
            best, total, reps = 1e9, 0.0, 0
            while (reps < args.replays) if args.replays else (total < args.seconds * 1e3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                g.replay()
                e1.record(side)
                side.synchronize()
                ms = e0.elapsed_time(e1)
                total += ms
                reps += 1
                best = min(best, ms)
        us = total * 1e3 / (reps * launches * k)
        ptrs = " ".join(f"{nm}@{t[nm].data_ptr() & 0xFFFFFFFFFF:010x}" for nm in ("obs1", "obs2", "term"))
        print(f"{tag:34s} k={k:4d}  {us:6.3f} us/frame (best replay {best * 1e3 / (launches * k):6.3f})  {ptrs}", flush=True)

    if args.tables_align:
        import ctypes as C

        land_b, hit_b = int(lib.pz_flight_table_bytes(0)), int(lib.pz_flight_table_bytes(1))
        cs0 = torch.cuda.current_stream().cuda_stream
        sets = {}
        t_land, t_hit = torch.empty(land_b, dtype=torch.uint8, device=dev), torch.empty(hit_b, dtype=torch.uint8, device=dev)
        sets["two plain allocations"] = (t_land, t_hit)
        big = torch.empty(3 << 30, dtype=torch.uint8, device=dev)
        off = (-big.data_ptr()) % (1 << 30)
        sets["1 GiB boundary of a 3 GiB allocation"] = (big[off:off + land_b], big[off + (1 << 30):off + (1 << 30) + hit_b])
        tabs = {}
        for kind, (tl, th) in sets.items():
            assert lib.pz_build_flight_tables(tl.data_ptr(), th.data_ptr(), cs0) == 0
            tabs[kind] = _native.PzFlightTables(tl.data_ptr(), th.data_ptr())
        torch.cuda.synchronize()
        acts = torch.randint(0, 18, (64, 2, n), dtype=torch.int32, device=dev)
        p_ = raw._ptrs
        snap = raw.state.clone()
        for rnd in range(3):
            for kind in tabs:
                raw.set_state(snap)
                side = torch.cuda.Stream()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        cs = torch.cuda.current_stream().cuda_stream
                        for t in range(512):
                            a = acts[t % 64]
                            rc = lib.pz_step(raw._state_ptr, n, raw._stride, raw._cfg_ref, a[0].data_ptr(), a[1].data_ptr(),
                                             raw._obs[0].data_ptr(), raw._obs[1].data_ptr(), raw._rew_raw[0].data_ptr(),
                                             raw._rew_raw[1].data_ptr(), raw._term_u8.data_ptr(), None, C.byref(tabs[kind]), cs)
                            assert rc == 0, rc
                    g.replay()
                    side.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                    for _ in range(20):
                        g.replay()
                    e1.record(side)
                    side.synchronize()
                print(f"  pz_step, player 2 = computer, tables in {kind:40s}: {e0.elapsed_time(e1) * 1e3 / (20 * 512):6.3f} us per launch", flush=True)
        return
    if args.tables_rank:
        import ctypes as C
        from pikazoo_amd import placement

        k = args.k[0]
        out = raw.rollout_random(1, k, t0=0)
        print("observation tensors:", raw.trajectory_placement, flush=True)
        o1, o2 = out["_obs"]
        land_b, hit_b = int(lib.pz_flight_table_bytes(0)), int(lib.pz_flight_table_bytes(1))
        blocks, spacers = {}, []
        while len(blocks) < 3 and len(spacers) < 40:
            c = torch.empty((land_b + hit_b + 4095) // 4096 * 4096 + (64 << 20), dtype=torch.uint8, device=dev)
            r1, r2 = placement.pair_ratio(o1, c), placement.pair_ratio(o2, c)
            kind = "rank of obs1" if r1 >= 0.9 and r2 < 0.9 else "rank of obs2" if r2 >= 0.9 and r1 < 0.9 else "third rank" if r1 < 0.9 and r2 < 0.9 else None
            if kind and kind not in blocks:
                blocks[kind] = c
            else:
                spacers.append(c)
            spacers.append(torch.empty(4 << 30, dtype=torch.uint8, device=dev))
        print("table blocks found:", sorted(blocks), "after", len(spacers), "other allocations", flush=True)
        tabs = {}
        cs0 = torch.cuda.current_stream().cuda_stream
        for kind, c in blocks.items():
            t_land = c[:land_b]
            t_hit = c[(land_b + 4095) // 4096 * 4096:][:hit_b]
            assert lib.pz_build_flight_tables(t_land.data_ptr(), t_hit.data_ptr(), cs0) == 0
            tabs[kind] = _native.PzFlightTables(t_land.data_ptr(), t_hit.data_ptr())
        torch.cuda.synchronize()
        ptrs = (out["_obs"][0].data_ptr(), out["_obs"][1].data_ptr(), out["_rew"][0].data_ptr(), out["_rew"][1].data_ptr(),
                out["_term"].data_ptr())
        snap = raw.state.clone()
        for rnd in range(2):
            for kind in sorted(tabs):
                raw.set_state(snap)
                launches = max(64, 2048 // k)
                side = torch.cuda.Stream()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        cs = torch.cuda.current_stream().cuda_stream
                        for j in range(launches):
                            rc = lib.pz_rollout_random(raw._state_ptr, n, raw._stride, raw._cfg_ref, 1, j * k, k,
                                                       out["actions"].data_ptr(), *ptrs, None, raw._episodes.data_ptr(),
                                                       C.byref(tabs[kind]), cs)
                            assert rc == 0, rc
                    g.replay()
                    side.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                    for _ in range(30):
                        g.replay()
                    e1.record(side)
                    side.synchronize()
                print(f"  flight tables in the {kind:13s}: {e0.elapsed_time(e1) * 1e3 / (30 * launches * k):6.3f} us per frame", flush=True)
        return
    if args.step_spread:
        from pikazoo_amd import placement

        # three 1 GiB blocks in three different ranks (walk the allocator, probe pairwise), and a fourth in the rank of the first
        blocks, spacers = [torch.empty(1 << 30, dtype=torch.uint8, device=dev)], []
        same_as_first = []
        while len(blocks) < 3 and len(spacers) < 40:
            spacers.append(torch.empty(4 << 30, dtype=torch.uint8, device=dev))
            c = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            r = [placement.pair_ratio(b, c) for b in blocks]
            if all(x < 0.9 for x in r):
                blocks.append(c)
            elif r[0] >= 0.9:
                same_as_first.append(c)
            else:
                spacers.append(c)
        print(f"{len(blocks)} mutually distinct blocks after {len(spacers)} spacers; {len(same_as_first)} more in the first one's rank", flush=True)
        if len(blocks) < 3:
            return
        words = 44
        P = __import__("ctypes").c_void_p

        def carve(block, off, nbytes, dt, shape):
            return block[off:off + nbytes].view(dt).view(shape)

        for place in ("one rank", "three ranks", "one rank", "three ranks"):
            b_state, b_o1, b_o2 = (blocks[0], blocks[0], blocks[0]) if place == "one rank" else blocks
            st = carve(b_state, 0, words * n * 4, torch.int32, (words, n))
            o1 = carve(b_o1, 256 << 20, n * 140, torch.int32, (n, 35))
            o2 = carve(b_o2, 512 << 20, n * 140, torch.int32, (n, 35))
            small = blocks[0]
            rew = [carve(small, (768 << 20) + i * (8 << 20), n * 4, torch.int32, (n,)) for i in range(2)]
            term = carve(small, (800 << 20), n, torch.uint8, (n,))
            acts = torch.randint(0, 18, (64, 2, n), dtype=torch.int32, device=dev)
            cfg = raw._cfg
            stream0 = torch.cuda.current_stream().cuda_stream
            assert lib.pz_init(st.data_ptr(), n, n, raw._cfg_ref, stream0) == 0
            assert lib.pz_reset(st.data_ptr(), n, n, raw._cfg_ref, None, o1.data_ptr(), o2.data_ptr(), None, stream0) == 0
            side = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    cs = torch.cuda.current_stream().cuda_stream
                    for t in range(256):
                        a = acts[t % 64]
                        rc = lib.pz_step(st.data_ptr(), n, n, raw._cfg_ref, a[0].data_ptr(), a[1].data_ptr(), o1.data_ptr(),
                                         o2.data_ptr(), rew[0].data_ptr(), rew[1].data_ptr(), term.data_ptr(), None,
                                         raw._tables_ref, cs)
                        assert rc == 0, rc
                g.replay()
                side.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                reps = 20 if n <= 65536 else 4
                for _ in range(reps):
                    g.replay()
                e1.record(side)
                side.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * 256)
            print(f"pz_step n={n}: state / obs1 / obs2 in {place:12s} {us:8.3f} us per launch  ({649 * n / us / 1e3:7.1f} GB/s of the 649 B contract)", flush=True)
        return
    if args.product:
        for k in args.k:
            for rnd in range(3):
                for place in (False, True):
                    e2 = pikazoo_v0.env(num_envs=args.num_envs, device="cuda:0", seed=0, is_player2_computer=args.p2_computer,
                                        place_trajectories=place, observation_dtype=torch.int16 if args.obs16 else torch.int32)
                    e2.reset()
                    torch.cuda.synchronize()
                    if args.idle:
                        __import__("time").sleep(args.idle)
                    out = e2.unwrapped.rollout_random(1, k, t0=0)
                    keep.append(out)
                    t = {"actions": out["actions"], "obs1": out["_obs"][0], "obs2": out["_obs"][1], "rew1": out["_rew"][0],
                         "rew2": out["_rew"][1], "term": out["_term"]}
                    info = e2.unwrapped.trajectory_placement
                    time_case(f"env alloc, place={place} {({x: (round(v, 3) if isinstance(v, float) else v) for x, v in info.items() if x != 'bytes'})}"[:110], k, t,
                              raw=e2.unwrapped)
        return
    if args.bases is not None:
        for k in args.k:
            for b in args.bases:
                time_case(f"arena + {b / (1 << 20):.6g} MiB", k, carved(k, 0, base=b))
        return
    if args.many:
        for r in range(args.many):
            for k in args.k:
                time_case(f"separate set {r}", k, separate(k))
        return
    for k in args.k:
        time_case("separate tensors (torch.empty)", k, separate(k))
        for gap in args.gaps:
            time_case(f"arena, gap {gap} B ({gap / (1 << 20):.4g} MiB)", k, carved(k, gap))
        time_case("separate tensors again", k, separate(k))
    print(torch.cuda.memory_summary(abbreviated=True))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B timing of kernel build variants (diagnostic; cdna_hip_programming.md rule 24: interleaved
rounds in ONE process, median and min reported).

    python tools/ab.py --build [--subset 705] name1=BITS name2=BITS ...    # here: cross-compile variants
    python tools/ab.py [--ai] [--n 65536] [--rollout K] name1 name2 ...    # on the GPU box: time them
                                                                           # (--rollout: pz_rollout_random, K frames/launch)
    a name with the suffix "+t" runs that library WITH both flight look-up tables (pz_flight_tables), "+q" with the
    power-hit table alone, "+p" on the packed state format, "+h" with int16 observations, combined at will,
    e.g. `python tools/ab.py --ai base base+t base+q base+tph`: run-time variants of ONE library.

Compile-time variants are DIAGNOSTIC builds (pika-zoo_amd/csrc/pz_diagnostic.hpp: the one switch of the kernel
sources): BITS are the low 16 bits of -DPZ_DIAGNOSTIC_BUILD (1 stamps, 2 no pair kernel, 4 no rollout pair kernel,
8 no scout wave, 16 early stores without the hand-shake, 32 no early stores, N << 8 hold the computer's wave back),
--subset the kernel families to instantiate (bits 16-29: a variant then builds in seconds).  A variant named "base"
is always built with BITS = 0.  Libraries go to tools/bin/ab_<name>.so (git-ignored, shipped by gpurun); the product
path pika-zoo_amd/lib/ never holds one.
"""
import ctypes as C
import statistics
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "pika-zoo_amd"))
from build import FLAGS as PRODUCT_FLAGS  # noqa: E402  (same compiler flags as the product library)
LIBDIR = REPO / "tools" / "bin"
LIB_ALIASES = {}


def diag_bits(bits=0, subset=0):
    """The value of -DPZ_DIAGNOSTIC_BUILD (pz_diagnostic.hpp): switches in the low 16 bits, kernel subset in bits 16-29."""
    return (int(subset) << 16) | (int(bits) & 0xFFFF)


def build(name, bits, subset):
    out = LIBDIR / f"ab_{name}.so"
    cmd = ["hipcc", *PRODUCT_FLAGS, "-shared", "-fPIC", f"-DPZ_DIAGNOSTIC_BUILD={diag_bits(bits, subset)}u",
           f"-I{REPO / 'include'}", f"-I{REPO / 'pika-zoo_amd' / 'csrc'}", "-o", str(out),
           str(REPO / "pika-zoo_amd" / "csrc" / "pz_kernels.hip")]
    subprocess.check_call(cmd)
    print("built", out.name, f"bits={bits} subset={subset}")


def main():
    args = sys.argv[1:]
    if args and args[0] == "--build":
        # --subset bits: for every variant incl. base (pz_kernels.hip dev_keep: only the kernel families the run will
        # launch -- seconds instead of 100 s per variant); the variants compile in parallel
        from concurrent.futures import ThreadPoolExecutor

        LIBDIR.mkdir(exist_ok=True)
        specs = args[1:]
        subset = 0
        if "--subset" in specs:
            at = specs.index("--subset")
            subset = int(specs[at + 1], 0)
            del specs[at:at + 2]
        jobs = [("base", 0, subset)] + [(name, int(bits, 0), subset) for name, _, bits in (s.partition("=") for s in specs)
                                        if name != "base"]
        with ThreadPoolExecutor(max_workers=6) as pool:
            list(pool.map(lambda j: build(*j), jobs))
        return
    import torch
    from pikazoo_amd import _native

    ai = "--ai" in args
    wrappers = "--wrappers" in args
    n = 65536
    if "--n" in args:
        n = int(args[args.index("--n") + 1])
    rollout = int(args[args.index("--rollout") + 1]) if "--rollout" in args else 0
    tape = "--tape" in args  # with --rollout K: pz_step_many on an action tape instead of pz_rollout_random
    names = [a for a in args if not a.startswith("--") and not a.isdigit()]
    if "--lib" in args:  # e.g. --lib product=pika-zoo_amd/lib/libpikazoo_hip.so: a library by path under a variant name
        at = args.index("--lib")
        alias, _, path = args[at + 1].partition("=")
        LIB_ALIASES[alias] = Path(path)
        names = [a for a in names if a != args[at + 1]]
    no_check = "--no-check" in args  # variants that change the stored state legitimately
    # the variant every other one is compared with: "base", or the first "base+..." given (e.g. base+t: with the tables)
    ref = "base" if "base" in names else next((nm for nm in names if nm.startswith("base+")), None)
    if ref is None:
        names = ["base"] + names
        ref = "base"
    dev = torch.device("cuda:0")
    P = C.c_void_p
    libs = {}
    loaded = {}
    for nm in names:
        file = nm.partition("+")[0]
        if file in loaded:
            libs[nm] = loaded[file]
            continue
        lib = loaded[file] = C.CDLL(str(LIB_ALIASES.get(file, LIBDIR / f"ab_{file}.so")))
        lib.pz_init.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P]
        lib.pz_reset.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P, P, P, P, P]
        lib.pz_step.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P, P, P, P, P, P, P, P, P, P]
        lib.pz_rollout_random.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), C.c_uint64, C.c_uint64,
                                          C.c_int32, P, P, P, P, P, P, P, P, P, P]
        lib.pz_step_many.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P, C.c_int32, P, P, P, P, P, P, P,
                                     P, P]
        lib.pz_flight_table_bytes.restype = C.c_int64
        lib.pz_flight_table_bytes.argtypes = [C.c_int32]
        lib.pz_build_flight_tables.argtypes = [P, P, P]
        libs[nm] = lib
    cfg = _native.PzConfig()
    cfg.winning_score, cfg.auto_reset, cfg.seed, cfg.p2_computer = 15, 1, 0, int(ai)
    if wrappers:
        cfg.simplify_action, cfg.ballpos_reward, cfg.x_line, cfg.y_line = 1, 1, 216, 176
        for i, v in enumerate((0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01)):
            cfg.additional_reward[i] = v
    def mods(nm):
        return nm.partition("+")[2]

    cfgs = {}
    for nm in names:
        c = cfgs[nm] = _native.PzConfig.from_buffer_copy(cfg)
        c.packed_state = int("p" in mods(nm))
        if "h" in mods(nm):
            c.normalize_obs = 2  # int16 observations
    # every variant owns its state buffer and initialises it itself: variants may differ in the state's layout
    states = {nm: (torch.zeros(36 * n, dtype=torch.uint8, device=dev) if "p" in mods(nm)
                   else torch.zeros((44, n), dtype=torch.int32, device=dev)) for nm in names}
    obs32 = [torch.zeros((n, 35), dtype=torch.int32, device=dev) for _ in range(2)]
    obs16 = [torch.zeros(((n + 1) // 2 * 2, 35), dtype=torch.int16, device=dev) for _ in range(2)]
    obs = obs32
    rew = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
    term = torch.zeros(n, dtype=torch.uint8, device=dev)
    # --slices N: distinct action slices cycled through (default 64 = 32 MB at 65 536 games, cache-resident; bench.py's
    # default run streams 2 000 slices = 1 GB cold from HBM)
    slices = int(args[args.index("--slices") + 1]) if "--slices" in args else 64
    acts = torch.randint(0, 13 if wrappers else 18, (slices, 2, n), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    base = libs[ref]
    tables = tables_hit = None
    if any("t" in mods(nm) or "q" in mods(nm) for nm in names):
        t_land = torch.empty(base.pz_flight_table_bytes(0), dtype=torch.uint8, device=dev)
        t_hit = torch.empty(base.pz_flight_table_bytes(1), dtype=torch.uint8, device=dev)
        assert base.pz_build_flight_tables(t_land.data_ptr(), t_hit.data_ptr(), stream) == 0
        tables = _native.PzFlightTables(t_land.data_ptr(), t_hit.data_ptr())
        tables_hit = _native.PzFlightTables(None, t_hit.data_ptr())  # "+q": the power-hit table alone

    for nm in names:
        assert libs[nm].pz_init(states[nm].data_ptr(), n, n, C.byref(cfgs[nm]), stream) == 0
        obs = obs16 if "h" in mods(nm) else obs32
        assert libs[nm].pz_reset(states[nm].data_ptr(), n, n, C.byref(cfgs[nm]), None, obs[0].data_ptr(), obs[1].data_ptr(),
                                 None, stream) == 0

    if rollout:
        k = rollout
        # the two observation tensors in different ranks of the HBM (pikazoo_amd/placement.py), like the env's own:
        # in one rank the memory system caps every variant at ~3.6 us per frame and hides what the kernels differ in
        from pikazoo_amd import placement
        t_obs = list(placement.alloc_pair((k, n, 35), torch.int32, dev))
        print("  observation tensors:", {x: (round(v, 3) if isinstance(v, float) else v) for x, v in placement.last_info.items()})
        t_rew = [torch.zeros((k, n), dtype=torch.int32, device=dev) for _ in range(2)]
        t_term = torch.zeros((k, n), dtype=torch.uint8, device=dev)
        t_act = torch.zeros((k, 2, n), dtype=torch.int32, device=dev)
        # --tapes N: distinct tapes cycled through (default 16 = 268 MB at k = 32: streamed from HBM; 1 = cache-resident)
        n_tapes = int(args[args.index("--tapes") + 1]) if "--tapes" in args else 16
        tapes = torch.randint(0, 13 if wrappers else 18, (n_tapes, k, 2, n), dtype=torch.int32, device=dev) if tape else None

    def run(nm, steps):
        lib = libs[nm]
        state = states[nm]
        tb = C.byref(tables) if "t" in mods(nm) else (C.byref(tables_hit) if "q" in mods(nm) else None)
        cfg = cfgs[nm]
        obs = obs16 if "h" in mods(nm) else obs32
        if rollout:
            for j in range(max(1, steps // rollout)):
                if tape:
                    rc = lib.pz_step_many(state.data_ptr(), n, n, C.byref(cfg), tapes[j % n_tapes].data_ptr(), rollout,
                                          t_obs[0].data_ptr(), t_obs[1].data_ptr(), t_rew[0].data_ptr(), t_rew[1].data_ptr(),
                                          t_term.data_ptr(), None, None, tb, stream)
                    assert rc == 0, rc
                    continue
                rc = lib.pz_rollout_random(state.data_ptr(), n, n, C.byref(cfg), 7, j * rollout, rollout,
                                           t_act.data_ptr(), t_obs[0].data_ptr(), t_obs[1].data_ptr(),
                                           t_rew[0].data_ptr(), t_rew[1].data_ptr(), t_term.data_ptr(), None, None,
                                           tb, stream)
                assert rc == 0, rc
            return max(1, steps // rollout) * rollout
        for t in range(steps):
            a = acts[t % slices]
            rc = lib.pz_step(state.data_ptr(), n, n, C.byref(cfg), a[0].data_ptr(), a[1].data_ptr(), obs[0].data_ptr(),
                             obs[1].data_ptr(), rew[0].data_ptr(), rew[1].data_ptr(), term.data_ptr(), None, tb, stream)
            assert rc == 0, (nm, rc)  # (-3: a PZ_DEV_SUBSET build without this launch's kernel family)
        return steps

    snapshots = {}
    for nm in names:
        run(nm, 700)
        snapshots[nm] = states[nm].clone()
    torch.cuda.synchronize()

    def restore(nm):
        states[nm].copy_(snapshots[nm])

    # every variant must produce the same trajectory as base (compared on the outputs; the state only between
    # variants of the same layout)
    finals = {}
    for nm in names:
        restore(nm)
        run(nm, 128)
        torch.cuda.synchronize()
        o = obs16 if "h" in mods(nm) else obs32
        finals[nm] = ((t_obs[0].clone(), t_obs[1].clone(), t_rew[0].clone(), t_term.clone()) if rollout
                      else (o[0][:n].to(torch.int32), o[1][:n].to(torch.int32), rew[0].clone(), term.clone()))
    for nm in names:
        same = all(torch.equal(a, b) for a, b in zip(finals[nm], finals[ref]))
        print(f"  {nm}: trajectory (observations, rewards, terminations) identical to {ref}: {same}")
    # The K launches of a round are captured once per variant in a hipGraph and replayed: an eager ctypes launch
    # costs the host ~7 us, which would hide every kernel faster than that ("--eager" keeps the direct calls).
    # (a k-frame round must be long enough for the clocks to settle: 12 launches = 1 ms was not -- 3 200 frames by default)
    K, rounds = (3200 if rollout else max(400, slices)), 9
    if "--frames" in args:
        K = int(args[args.index("--frames") + 1])
    eager = "--eager" in args
    side = torch.cuda.Stream()
    graphs = {}
    if not eager:
        stream_holder = [stream]
        for nm in names:
            restore(nm)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    stream = torch.cuda.current_stream().cuda_stream  # run() launches on the capture stream
                    frames_per_graph = run(nm, K)
            stream = stream_holder[0]
            graphs[nm] = g
    times = {nm: [] for nm in names}
    for _ in range(rounds):
        if not eager and "--no-lead-in" not in args:
            # an untimed replay in front of every round: whatever runs first behind the pause between two rounds reads
            # slow (with a computer player by ~7 %: the flight tables' lines have to come back into the caches)
            with torch.cuda.stream(side):
                graphs[names[-1]].replay()
            torch.cuda.synchronize()
        for nm in names:
            restore(nm)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if eager:
                e0.record()
                frames = run(nm, K)
                e1.record()
            else:
                with torch.cuda.stream(side):
                    e0.record(side)
                    graphs[nm].replay()
                    e1.record(side)
                frames = frames_per_graph
            torch.cuda.synchronize()
            times[nm].append(e0.elapsed_time(e1) * 1e3 / frames)
    print(f"n={n} p2_computer={ai} wrappers={wrappers} rollout={rollout} {'eager' if eager else 'hipGraph'}: us per "
          f"{'frame' if rollout else 'launch'}, median / min over {rounds} interleaved rounds of {K}")
    for nm in names:
        print(f"  {nm:28s} {statistics.median(times[nm]):7.3f} {min(times[nm]):7.3f}"
              + ("   [" + " ".join(f"{t:.3f}" for t in times[nm]) + "]" if "--samples" in args else ""))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Timing-only ablation of the step kernel (diagnostic, not product, not a test).

Builds csrc with -DPZ_ABLATE into pika-zoo_amd/lib/libpikazoo_hip_ablate.so and times the
human-vs-human step at 65 536 games with traffic classes redirected to one workgroup's span or the
frame skipped (bits 3.. of cfg.packed_state, see pz_kernels.hip).  Interleaved rounds in one process
(cdna_hip_programming.md rule 24); prints median / min microseconds per launch from HIP events.

    python tools/ablate.py --build      # here (cross-compile)
    python tools/ablate.py [N] [--ai] [--tables]   # on the GPU box (--tables: computer player on the look-up tables)
"""
import ctypes as C
import statistics
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "pika-zoo_amd"))
from build import FLAGS as PRODUCT_FLAGS  # noqa: E402  (same compiler flags as the product library)
LIB = REPO / "pika-zoo_amd" / "lib" / "libpikazoo_hip_ablate.so"


def build():
    cmd = ["hipcc", *PRODUCT_FLAGS, "-shared", "-fPIC", "-DPZ_ABLATE=1",
           f"-I{REPO / 'include'}", f"-I{REPO / 'pika-zoo_amd' / 'csrc'}", "-o", str(LIB),
           str(REPO / "pika-zoo_amd" / "csrc" / "pz_kernels.hip")]
    subprocess.check_call(cmd)


def main():
    if "--build" in sys.argv:
        build()
        return
    import torch
    from pikazoo_amd import _native

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    p2ai = "--ai" in sys.argv
    lib = C.CDLL(str(LIB))
    dev = torch.device("cuda:0")
    cfg = _native.PzConfig()
    cfg.winning_score, cfg.auto_reset, cfg.seed, cfg.p2_computer = 15, 1, 0, int(p2ai)
    state = torch.zeros((44, n), dtype=torch.int32, device=dev)
    obs = [torch.zeros((n, 35), dtype=torch.int32, device=dev) for _ in range(2)]
    rew = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
    term = torch.zeros(n, dtype=torch.uint8, device=dev)
    acts = torch.randint(0, 18, (64, 2, n), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    P = C.c_void_p
    lib.pz_init.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P]
    lib.pz_reset.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P, P, P, P, P]
    lib.pz_step.argtypes = [P, C.c_int64, C.c_int64, C.POINTER(_native.PzConfig), P, P, P, P, P, P, P, P, P, P]
    tb = None
    if "--tables" in sys.argv:
        lib.pz_flight_table_bytes.restype = C.c_int64
        lib.pz_flight_table_bytes.argtypes = [C.c_int32]
        lib.pz_build_flight_tables.argtypes = [P, P, P]
        t_land = torch.empty(lib.pz_flight_table_bytes(0), dtype=torch.uint8, device=dev)
        t_hit = torch.empty(lib.pz_flight_table_bytes(1), dtype=torch.uint8, device=dev)
        assert lib.pz_build_flight_tables(t_land.data_ptr(), t_hit.data_ptr(), stream) == 0
        tables = _native.PzFlightTables(t_land.data_ptr(), t_hit.data_ptr())
        tb = C.byref(tables)
    assert lib.pz_init(state.data_ptr(), n, n, C.byref(cfg), stream) == 0
    assert lib.pz_reset(state.data_ptr(), n, n, C.byref(cfg), None, obs[0].data_ptr(), obs[1].data_ptr(), None, stream) == 0

    def run(flags, steps):
        cfg.packed_state = flags  # bits 3.. of this field are the timing-only switches (bit 0: packed state format)
        for t in range(steps):
            a = acts[t % 64]
            lib.pz_step(state.data_ptr(), n, n, C.byref(cfg), a[0].data_ptr(), a[1].data_ptr(), obs[0].data_ptr(),
                        obs[1].data_ptr(), rew[0].data_ptr(), rew[1].data_ptr(), term.data_ptr(),
                        scratch.data_ptr() if (flags & 4096) else None, tb, stream)

    run(0, 600)  # desynchronise the games so divergence is realistic
    snapshot = state.clone()
    variants = {"baseline": 0, "no_state_stores": 1024, "no_obs_stores": 2048, "no_stores": 3072,
                "wide_fake_state_stores": 1024 + 4096, "wide_fake_state_stores_no_obs": 1024 + 2048 + 4096}
    scratch = torch.zeros(96 * n, dtype=torch.uint8, device=dev)
    if p2ai:
        variants = {"baseline": 0, "no_frame": 8, "no_landing_A": 32, "no_candidates": 64, "no_landing_B": 128,
                    "no_A_no_B": 160, "no_predictors": 224}
        if tb is not None:
            variants.update({"no_predraw": 256, "no_predictors_no_predraw": 224 + 256, "no_decision": 512,
                             "no_decision_no_predictors_no_predraw": 512 + 256 + 224, "no_obs": 16})
    times = {k: [] for k in variants}
    K = 300
    # one hipGraph of K launches per variant (the flag bits travel by value in the kernel arguments): an eager ctypes
    # launch costs the host ~7 us and would hide every variant faster than that
    side = torch.cuda.Stream()
    graphs = {}
    default_stream = stream
    for name, flags in variants.items():
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                stream = torch.cuda.current_stream().cuda_stream
                run(flags, K)
        stream = default_stream
        graphs[name] = g
    for rnd in range(7):
        for name, flags in variants.items():
            state.copy_(snapshot)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(side):
                e0.record(side)
                graphs[name].replay()
                e1.record(side)
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 1e3 / K)
    print(f"n={n} p2_computer={p2ai}: microseconds per launch (median / min over 7 interleaved rounds of {K})")
    for name in variants:
        print(f"  {name:24s} {statistics.median(times[name]):7.2f} {min(times[name]):7.2f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Diagnostic: ONE 65 536-game batch stepped as C independent sub-batch chains inside ONE hipGraph (fork / join
at the graph's ends), against the single chain.  Each step of sub-batch c is one pz_step launch on lane range
[c*N/C, (c+1)*N/C) of the same state / observation / reward tensors (pointer offsets, full column pitch).

    python tools/chains.py [--n 65536] [--steps 512] [--ai] [--random] [--separate]

--random: every step is pz_step_random(k = 1) -- the uniform random policy drawn inside the step launch -- instead of
pz_step on a pre-generated action tape.
"""
import ctypes as C
import statistics
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "pika-zoo_amd"))
import torch  # noqa: E402

from pikazoo_amd import _native, pikazoo_v0  # noqa: E402


def main():
    args = sys.argv[1:]
    n = int(args[args.index("--n") + 1]) if "--n" in args else 65536
    K = int(args[args.index("--steps") + 1]) if "--steps" in args else 512
    ai = "--ai" in args
    fused_policy = "--random" in args
    lib = _native.load()
    dev = torch.device("cuda:0")
    results = {}
    final = {}
    for chains in (1, 2, 4):
        env = pikazoo_v0.env(num_envs=n, device=dev, seed=0, validate_actions=False, is_player2_computer=ai)
        env.reset()
        env.step_random(2, t0=0, k=1024)
        acts = torch.empty((K, 2, n), dtype=torch.int32, device=dev)
        for t in range(K):
            a = env.random_actions(1, t)
            acts[t, 0].copy_(a["player_1"]); acts[t, 1].copy_(a["player_2"])
        torch.cuda.synchronize()
        sub = n // chains
        main_s = torch.cuda.Stream(device=dev)
        side = [torch.cuda.Stream(device=dev) for _ in range(chains - 1)]
        streams = [main_s] + side

        def launch(c, t, stream):
            lo = c * sub
            cfg = _native.PzConfig.from_buffer_copy(env._cfg)
            cfg.env_id_base = env.env_id_base + lo
            cfgs.append(cfg)
            if fused_policy:
                rc = lib.pz_step_random(env.state.data_ptr() + 4 * lo, sub, env._stride, C.byref(cfg), 1, t, 1,
                                        env._obs[0].data_ptr() + 140 * lo, env._obs[1].data_ptr() + 140 * lo,
                                        env._rew_raw[0].data_ptr() + 4 * lo, env._rew_raw[1].data_ptr() + 4 * lo,
                                        env._term_u8.data_ptr() + lo, None, None, env._tables_ref, stream.cuda_stream)
            else:
                rc = lib.pz_step(env.state.data_ptr() + 4 * lo, sub, env._stride, C.byref(cfg),
                                 acts[t, 0].data_ptr() + 4 * lo, acts[t, 1].data_ptr() + 4 * lo,
                                 env._obs[0].data_ptr() + 140 * lo, env._obs[1].data_ptr() + 140 * lo,
                                 env._rew_raw[0].data_ptr() + 4 * lo, env._rew_raw[1].data_ptr() + 4 * lo,
                                 env._term_u8.data_ptr() + lo, None, env._tables_ref, stream.cuda_stream)
            assert rc == 0, rc

        cfgs = []
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(main_s):
            with torch.cuda.graph(graph, stream=main_s, capture_error_mode="thread_local"):
                cur = torch.cuda.current_stream(dev)
                fork = torch.cuda.Event()
                fork.record(cur)
                for s in side:
                    s.wait_event(fork)
                for t in range(K):
                    for c in range(chains):
                        launch(c, t, cur if c == 0 else side[c - 1])
                for s in side:
                    e = torch.cuda.Event()
                    e.record(s)
                    cur.wait_event(e)
            graph.replay()
            main_s.synchronize()
            final[chains] = env.state.clone()
            times = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main_s)
                for _ in range(4):
                    graph.replay()
                e1.record(main_s)
                main_s.synchronize()
                times.append(e0.elapsed_time(e1) * 1e3 / (4 * K))
        results[chains] = (statistics.median(times), min(times))
    if "--separate" in args:
        # the same sub-batch chains as SEPARATE hipGraphs, one per stream (different priorities = different hardware
        # queues), replayed side by side from this one thread
        for chains in (2, 4):
            env = pikazoo_v0.env(num_envs=n, device=dev, seed=0, validate_actions=False, is_player2_computer=ai)
            env.reset()
            env.step_random(2, t0=0, k=1024)
            acts = torch.empty((K, 2, n), dtype=torch.int32, device=dev)
            for t in range(K):
                a = env.random_actions(1, t)
                acts[t, 0].copy_(a["player_1"]); acts[t, 1].copy_(a["player_2"])
            torch.cuda.synchronize()
            sub = n // chains
            streams = [torch.cuda.Stream(device=dev, priority=-(c % 2)) for c in range(chains)]
            cfgs, graphs = [], []
            for c in range(chains):
                lo = c * sub
                cfg = _native.PzConfig.from_buffer_copy(env._cfg)
                cfg.env_id_base = env.env_id_base + lo
                cfgs.append(cfg)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(streams[c]):
                    with torch.cuda.graph(g, stream=streams[c], capture_error_mode="thread_local"):
                        cur = torch.cuda.current_stream(dev)
                        for t in range(K):
                            if fused_policy:
                                rc = lib.pz_step_random(env.state.data_ptr() + 4 * lo, sub, env._stride, C.byref(cfg), 1, t, 1,
                                                        env._obs[0].data_ptr() + 140 * lo, env._obs[1].data_ptr() + 140 * lo,
                                                        env._rew_raw[0].data_ptr() + 4 * lo, env._rew_raw[1].data_ptr() + 4 * lo,
                                                        env._term_u8.data_ptr() + lo, None, None, env._tables_ref,
                                                        cur.cuda_stream)
                            else:
                                rc = lib.pz_step(env.state.data_ptr() + 4 * lo, sub, env._stride, C.byref(cfg),
                                                 acts[t, 0].data_ptr() + 4 * lo, acts[t, 1].data_ptr() + 4 * lo,
                                                 env._obs[0].data_ptr() + 140 * lo, env._obs[1].data_ptr() + 140 * lo,
                                                 env._rew_raw[0].data_ptr() + 4 * lo, env._rew_raw[1].data_ptr() + 4 * lo,
                                                 env._term_u8.data_ptr() + lo, None, env._tables_ref, cur.cuda_stream)
                            assert rc == 0, rc
                graphs.append(g)
            for c in range(chains):
                with torch.cuda.stream(streams[c]):
                    graphs[c].replay()
            torch.cuda.synchronize()
            same = torch.equal(env.state, final[1])
            times = []
            for _ in range(7):
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                ends = [torch.cuda.Event(enable_timing=True) for _ in range(chains)]
                e0.record(streams[0])
                for c in range(1, chains):
                    streams[c].wait_event(e0)
                for _ in range(4):
                    for c in range(chains):
                        with torch.cuda.stream(streams[c]):
                            graphs[c].replay()
                for c in range(chains):
                    ends[c].record(streams[c])
                torch.cuda.synchronize()
                times.append(max(e0.elapsed_time(e) for e in ends) * 1e3 / (4 * K))
            med = statistics.median(times)
            print(f"n={n} ai={ai} policy_in_step={fused_policy} {chains} separate graphs on {chains} streams: {med:.3f} us per step (min {min(times):.3f}) -> "
                  f"{n / med / 1e3:.2f} G env-steps/s; same trajectory as one chain: {same}")
    for c, (med, mn) in results.items():
        print(f"n={n} ai={ai} policy_in_step={fused_policy} chains={c}: {med:.3f} us per step (min {mn:.3f})  -> {n / med / 1e3:.2f} G env-steps/s; "
              f"same trajectory as one chain: {torch.equal(final[c], final[1])}")


if __name__ == "__main__":
    main()

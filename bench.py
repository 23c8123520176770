#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused HIP step path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (N=1): 65 536 independent games on one MI355X, both players driven by the uniform random
policy, winning_score=15, serve="winner" (BASELINE.json configs: the per-GPU shard of the
524 288-game job; the metric is quoted on "random policy, 65 536 envs").  A *step* is one frame of
every game = ONE launch of the fused kernel through the C ABI (``pz_step``): it reads the 44 state
columns and the two action vectors from HBM and writes state, both 35-dim observations, both
rewards and the terminated flags.  The actions of all W+K steps are generated on device by the
policy kernel BEFORE the timed region, so every input is resident in HBM when timing starts.

Multi-GPU (torchrun, one rank per GPU): weak scaling, rank r owns global games
[r*65536, (r+1)*65536); no collective on the step path, one SUM all-reduce of the counters after
the timed region (RCCL).  Timing: barrier + synchronize on both sides, MAX over ranks.

The single JSON line also carries
  roofline     -- algorithmic HBM bytes per launch (649 B/game-step) / average launch
                  duration measured with HIP events on the launch stream, against 8 TB/s;
  cpu_baseline -- the CPU oracle (oracle/pz_oracle.c, a C port of the reference's Python step,
                  kind="port") timed on this host's cores on a bounded sample of the same workload,
                  and used to check the GPU trajectories bit-for-bit on a lane subset.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "pika-zoo_amd"))
sys.path.insert(0, str(REPO))

import torch  # noqa: E402

from pikazoo_amd import _native, dist, pikazoo_v0  # noqa: E402

BYTES_PER_ENV_STEP = 8 * 44 + 297  # rd+wr state, 2 actions, 2x35 obs, 2 rewards, 1 flag
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
ACTION_SEED = 1


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--num-envs", type=int, default=65536, help="games per GPU")
    ap.add_argument("--launch", choices=["graph", "cabi", "api"], default="graph",
                    help="graph: the K launches captured once in a hipGraph and replayed; cabi: K direct "
                         "C-ABI calls; api: K env.step() calls")
    ap.add_argument("--p2-computer", action="store_true", help="config 3: rule-based AI on player 2")
    ap.add_argument("--p1-computer", action="store_true", help="rule-based AI on player 1 (not a BASELINE config)")
    ap.add_argument("--wrappers", action="store_true", help="config 5: fused SimplifyAction+RewardByBallPosition")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall budget of the CPU baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=16, help="upper bound on CPU baseline threads")
    ap.add_argument("--check-lanes", type=int, default=2048, help="lanes replayed on the CPU oracle for parity")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dist-backend", default=None,
                    help="torch.distributed backend (default nccl = RCCL). 'gloo' + PZ_BENCH_ONE_DEVICE=1 rehearses "
                         "the N>1 path with every rank on cuda:0 of a 1-GPU box")
    ap.add_argument("--extra", action="store_true", help="also time configs 3 and 5 and report them under 'extra'")
    return ap.parse_args()


def make_env(args, shard, p2_computer, wrappers, device):
    from pikazoo_amd.wrappers import RewardByBallPosition, SimplifyAction

    env = pikazoo_v0.env(winning_score=15, serve="winner", is_player2_computer=p2_computer,
                         is_player1_computer=args.p1_computer,
                         num_envs=shard.n_local, device=device, seed=0, env_id_base=shard.env_id_base,
                         auto_reset=True, validate_actions=False)
    if wrappers:
        env = SimplifyAction(env)
        env = RewardByBallPosition(env, (0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01), 216, 176)
    return env


def pregenerate_actions(raw, total_steps):
    """[total_steps, 2, n] int32 in HBM from the device policy kernel (Philox stream ACTION_SEED)."""
    lib = _native.load()
    acts = torch.empty((total_steps, 2, raw.num_envs), dtype=torch.int32, device=raw.device)
    s = torch.cuda.current_stream(raw.device).cuda_stream
    for t in range(total_steps):
        _native.check(lib.pz_random_actions(acts[t, 0].data_ptr(), acts[t, 1].data_ptr(), raw.num_envs,
                                            raw.env_id_base, ACTION_SEED, t, raw.n_actions, s), "pz_random_actions")
    return acts


def run_gpu(args, env, acts, warmup, steps, launch):
    """W untimed + K timed launches.  Returns (wall seconds of the K steps, event ms of the K steps)."""
    raw = env.unwrapped
    lib = _native.load()
    n = raw.num_envs
    st, cfg = raw.state.data_ptr(), raw._cfg_ref
    o1, o2 = raw._obs[0].data_ptr(), raw._obs[1].data_ptr()
    r1, r2, tm = raw._rew_raw[0].data_ptr(), raw._rew_raw[1].data_ptr(), raw._term_u8.data_ptr()
    a_ptr = acts.data_ptr()
    a_stride = 2 * n * 4

    def launch_range(t_lo, t_hi, stream):
        s = stream.cuda_stream
        for t in range(t_lo, t_hi):
            rc = lib.pz_step(st, n, raw._stride, cfg, a_ptr + t * a_stride, a_ptr + t * a_stride + n * 4, o1, o2, r1, r2, tm,
                             None, s)
            if rc:
                _native.check(rc, "pz_step")

    stream = torch.cuda.Stream(device=raw.device)
    graph = None
    with torch.cuda.stream(stream):
        if launch == "api":
            names = raw.possible_agents
            # the per-step action dicts a policy would hand over (views built outside the timed loop)
            feed = [{names[0]: acts[t, 0], names[1]: acts[t, 1]} for t in range(warmup + steps)]

            def timed(t_lo, t_hi):
                step = env.step
                for t in range(t_lo, t_hi):
                    step(feed[t])
            timed(0, warmup)
        else:
            launch_range(0, warmup, stream)
            if launch == "graph":
                stream.synchronize()
                state_before = raw.state.clone()
                graph = torch.cuda.CUDAGraph()
                # thread_local: the RCCL watchdog thread of a multi-rank run may poll events while we capture
                with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                    launch_range(warmup, warmup + steps, torch.cuda.current_stream(raw.device))
                graph.replay()                 # untimed first replay (graph upload), then rewind
                raw.state.copy_(state_before)

                def timed(t_lo, t_hi):
                    graph.replay()
            else:
                def timed(t_lo, t_hi):
                    launch_range(t_lo, t_hi, stream)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        stream.synchronize()
        dist.barrier()
        torch.cuda.synchronize(raw.device)
        t0 = time.perf_counter()
        ev0.record(stream)
        timed(warmup, warmup + steps)
        ev1.record(stream)
        torch.cuda.synchronize(raw.device)
        dist.barrier()
        wall = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1)
    raw.steps_done = warmup + steps
    return wall, ev_ms


def usable_cores(requested: int) -> int:
    """Threads for the CPU baseline: the cores this process may actually run on (affinity mask and
    cgroup CPU quota), capped at `requested` (the GPU box gives one GPU's job a 16-core share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, requested))


def cpu_baseline(args, raw_gpu, p2_computer, wrappers, total_steps):
    """Oracle timed on the host cores (bounded sample) + bit-exact check of a GPU lane subset."""
    from oracle import pz_oracle as po

    po.build()
    cores = usable_cores(args.cpu_threads)
    table = (0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01) if wrappers else None
    n = raw_gpu.num_envs

    def cfg(base):
        return po.make_config(winning_score=15, serve="winner", is_player2_computer=p2_computer,
                              is_player1_computer=args.p1_computer,
                              simplify_action=wrappers, additional_reward=table, seed=0, env_id_base=base)

    # timing sample: the same 65 536-game batch, as many 250-step chunks as fit the budget
    env = po.OracleEnv(n, cfg(raw_gpu.env_id_base), nthreads=cores)
    env.reset()
    env.rollout_random(ACTION_SEED, 0, 10)  # touch pages / spin up the thread pool
    done, t_spent, chunk = 10, 0.0, 250
    while t_spent < args.cpu_seconds and done < 100000:
        t0 = time.perf_counter()
        env.rollout_random(ACTION_SEED, done, chunk)
        t_spent += time.perf_counter() - t0
        done += chunk
    sample_steps = done - 10
    value = n * sample_steps / t_spent
    # single-core figure on a smaller sample (scalar port)
    env1 = po.OracleEnv(4096, cfg(raw_gpu.env_id_base), nthreads=1)
    env1.reset()
    t0 = time.perf_counter()
    env1.rollout_random(ACTION_SEED, 0, 300)
    one_core = 4096 * 300 / (time.perf_counter() - t0)
    # BASELINE.json configs[0]: ONE env, 10 000 steps, as the reference's own scalar loop would run it
    env0 = po.OracleEnv(1, cfg(raw_gpu.env_id_base), nthreads=1)
    env0.reset()
    t0 = time.perf_counter()
    for t in range(10000):
        a1, a2 = po.random_actions(1, raw_gpu.env_id_base, ACTION_SEED, t, raw_gpu.n_actions)
        env0.step(a1, a2)
    one_env = 10000 / (time.perf_counter() - t0)
    # parity: replay the first check-lanes games for every step the GPU ran
    k = min(args.check_lanes, n)
    chk = po.OracleEnv(k, cfg(raw_gpu.env_id_base), nthreads=cores)
    chk.reset()
    chk.rollout_random(ACTION_SEED, 0, total_steps)
    gpu_state = raw_gpu.state[:, :k].cpu().numpy()
    parity = bool((gpu_state == chk.state).all())
    return {
        "value": value, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{n} games x {sample_steps} steps of the same workload, OpenMP static lane partition "
                  f"over {cores} threads ({t_spent:.1f} s)",
        "one_core_value": one_core,
        "config1_one_env_steps_per_s": one_env,  # 1 game stepped call by call from Python (ctypes overhead-bound)
        "parity_lanes_checked": k, "parity_steps_checked": total_steps, "parity_bit_exact": parity,
    }


def measure(args, shard, device, p2_computer, wrappers, launch, with_cpu):
    env = make_env(args, shard, p2_computer, wrappers, device)
    raw = env.unwrapped
    env.reset()
    total = args.warmup + args.steps
    acts = pregenerate_actions(raw, total)
    torch.cuda.synchronize(device)
    wall, ev_ms = run_gpu(args, env, acts, args.warmup, args.steps, launch)
    cdev = dist.collective_device(device)  # counters live on the GPU under nccl (RCCL), on the CPU under gloo
    wall = dist.all_reduce_max(wall, device=cdev)
    terminated_now = int(raw._term_u8.sum().item())
    n_total, = dist.all_reduce_sum([raw.num_envs], device=cdev)
    res = {
        "wall_s": wall, "event_ms": ev_ms, "n_total": n_total,
        "value": n_total * args.steps / wall,
        "launch_us": ev_ms * 1e3 / args.steps,
        "terminated_in_last_frame": terminated_now,
    }
    if with_cpu:
        res["cpu"] = cpu_baseline(args, raw, p2_computer, wrappers, total)
    del acts
    return res


def measure_rollout(args, shard, device, k, tape=False, p2_computer=False):
    """pz_rollout_random (or, tape=True, pz_step_many on a pre-generated action tape): k frames per
    launch, every frame's outputs written to trajectory tensors (state in registers, read/written once
    per launch).  Honest bytes per game-step of THESE kernels: 297 (8 of them action words written
    resp. read) + 352/k."""
    env = make_env(args, shard, p2_computer, False, device)
    raw = env.unwrapped
    env.reset()
    launches = max(1, args.steps // k)
    tapes = None
    if tape:
        acts = pregenerate_actions(raw, (launches + 1) * k).view(launches + 1, k, 2, raw.num_envs)
        tapes = [acts[j] for j in range(launches + 1)]
        out = raw.step_many(tapes[0])
    else:
        out = raw.rollout_random(ACTION_SEED, k)          # allocate + warm up
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for j in range(launches):
        out = raw.step_many(tapes[j + 1], out=out) if tape else raw.rollout_random(ACTION_SEED, k, out=out)
    ev1.record()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    us_per_frame = ev0.elapsed_time(ev1) * 1e3 / (launches * k)
    bytes_per_step = 297 + 352.0 / k
    gbps = bytes_per_step * raw.num_envs / (us_per_frame * 1e-6) / 1e9
    return {"value": raw.num_envs * launches * k / wall, "us_per_frame": us_per_frame, "k": k,
            "bytes_per_game_step": bytes_per_step, "achieved_GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS}


def load_traffic(workload_key, num_envs):
    """HBM bytes per launch from the committed PMC profile (profiles/traffic.json) of this workload at this batch
    size, or None."""
    p = REPO / "profiles" / "traffic.json"
    if p.exists():
        try:
            entry = json.loads(p.read_text()).get(workload_key, {})
            return entry.get("hbm_bytes_per_launch") if entry.get("num_envs") == num_envs else None
        except Exception:  # noqa: BLE001
            return None
    return None


def ensure_built():
    """Build libpikazoo_hip.so if the tree does not carry it (hipcc is on every box of this image)."""
    if not _native.LIB_PATH.exists():
        import importlib.util

        spec = importlib.util.spec_from_file_location("pz_build", REPO / "pika-zoo_amd" / "build.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()


def main():
    args = parse_args()
    if int(os.environ.get("RANK", "0")) == 0:
        ensure_built()
    rank, world, local_rank = dist.init_from_env(args.dist_backend)
    if os.environ.get("PZ_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    shard = dist.weak_shard(args.num_envs, rank, world)

    main_res = measure(args, shard, device, args.p2_computer, args.wrappers, args.launch,
                       with_cpu=(rank == 0 and world == 1 and not args.no_cpu))
    extra = {}
    if args.extra and world == 1:
        for key, (ai, wr) in {"cfg3_p2_computer": (True, False), "cfg5_fused_wrappers": (False, True)}.items():
            r = measure(args, shard, device, ai, wr, args.launch, with_cpu=False)
            extra[key] = {"value": r["value"], "launch_us": r["launch_us"]}
        extra["rollout_k32"] = measure_rollout(args, shard, device, k=32)
        extra["step_many_k32"] = measure_rollout(args, shard, device, k=32, tape=True)
        extra["rollout_k32_p2_computer"] = measure_rollout(args, shard, device, k=32, p2_computer=True)
        # the same kernel at larger batches (more waves per SIMD hide each other's latency)
        sweep = {}
        for n_big in (262144, 524288, 1048576):
            a2 = argparse.Namespace(**{**vars(args), "num_envs": n_big, "steps": 300, "warmup": 50})
            r = measure(a2, dist.weak_shard(n_big, rank, world), device, False, False, "cabi", with_cpu=False)
            gbps = BYTES_PER_ENV_STEP * n_big / (r["launch_us"] * 1e-6) / 1e9
            sweep[str(n_big)] = {"value": r["value"], "launch_us": r["launch_us"], "achieved_GBps": gbps,
                                 "frac_of_8TBps": gbps / HBM_PEAK_GBPS}
        extra["batch_sweep_random_random"] = sweep
        for mode in ("cabi", "api"):
            r = measure(args, shard, device, args.p2_computer, args.wrappers, mode, with_cpu=False)
            extra[f"launch_{mode}"] = {"value": r["value"], "launch_us": r["launch_us"]}

    if rank == 0:
        launch_s = main_res["launch_us"] * 1e-6
        alg_bytes = BYTES_PER_ENV_STEP * args.num_envs
        achieved = alg_bytes / launch_s / 1e9
        wl = "cfg3" if args.p2_computer else ("cfg5" if args.wrappers else "random_random")
        out = {
            "metric": "env-steps/sec (random policy, 65 536 envs per GPU)",
            "value": main_res["value"], "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main_res["wall_s"] * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.num_envs} games per GPU, both players uniform-random actions "
                            f"(Philox policy stream pre-generated in HBM), winning_score=15, serve=winner, "
                            f"auto-reset, p1_computer={args.p1_computer}, p2_computer={args.p2_computer}, "
                            f"fused_wrappers={args.wrappers}",
                "num_envs_per_gpu": args.num_envs, "num_envs_total": main_res["n_total"],
                "launch": args.launch,
                "kernel": ("pz::step_kernel<AI1,AI2,kActions,true> via pz_step" if args.num_envs >= 393216
                           else "pz::step_kernel<AI1,AI2,kActions,true,kScoutLoads> via pz_step" if args.p2_computer or args.p1_computer
                           else "pz::step_pair_kernel via pz_step"),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": load_traffic(wl, args.num_envs),
                "algorithmic_bytes_per_launch": alg_bytes, "launch_us": main_res["launch_us"],
            },
        }
        if "cpu" in main_res:
            out["cpu_baseline"] = main_res["cpu"]
        if extra:
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

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

Sequence of one measurement (every part replayed by the CPU oracle for the in-run parity check):
  reset -> BURN-IN (untimed; --burn-in frames of the on-device policy, so that the timed frames are
  steady-state play: rounds ending, auto-resets, collisions) -> W warm-up launches -> K x ceil(2 048 / K)
  launches captured in ONE hipGraph, every one on its OWN action slice (a cold tape: >= 2 048 distinct
  slices = 1 GB streamed from HBM whatever K is, so that the driver's `--steps 20`, the default run and
  every rocprofv3 trace measure one workload; `--action-tape hot` re-uses the K slices instead) -> one
  untimed replay (graph upload + calibration) -> R timed replays, R chosen so that the timed region
  lasts at least --min-time seconds whatever K is.  `steps` in the JSON line is K as given;
  `timed_steps` = the launches actually timed; value = games * timed_steps / wall.

Multi-GPU (one rank per GPU; under torch.distributed.run as the driver launches it, or plain
`python bench.py --gpus N`, which starts that launcher as a child process): weak scaling, rank r owns
global games [r*65536, (r+1)*65536); no collective on the step path, the counters are summed / gathered
after the timed region (RCCL; `rccl_ranks` = 0 and `dist_note` says why when they had to travel over
gloo).  Timing: barrier + synchronize on both sides, MAX over ranks; `per_rank` lists every rank's own
launch duration and rate, and the first and the last rank replay 1 024 of their games on the oracle.

The single JSON line is FLAT (scalars and short strings only, < 8 KB: the driver's record drops nested objects) and carries
  roofline     -- algorithmic HBM bytes per launch (649 B/game-step) / average launch duration measured with
                  HIP events on the launch stream over the timed region, against 8 TB/s (`frac`: a CONTRACT-bytes
                  figure, see `frac_basis`); the same from the wall clock `value` is computed from (`frac_wall`); and
                  from the PMC-measured bytes of profiles/traffic.json (`traffic`, `frac_traffic`, with the build the
                  counters were taken on: `traffic_build_id`, `traffic_stale`); `bound_regime`: what the launch runs
                  against at this batch size; and one scalar per figure of every other measurement of the run:
                  cold_tape_* / hot_tape_* (the same workload on both kinds of action tape), floor_* (what the launch is
                  made of: pz_probe_launch of the diagnostics library), two_chains_* (the batch as two sub-batch chains),
                  cfg2_* cfg3_* cfg3_compute_* cfg5_* n524288_* packed_* int16obs_* (the other single-GPU BASELINE configs,
                  the packed / int16 formats and the 524 288-game batch, each timed like the headline and oracle-checked),
                  rollout_k32_* step_many_k32_* ..._p2_computer_* rollout_k128_* (the k-frame launches, on their own
                  297 + 352 / k bytes), policy_fused_*, configs_parity_all_bit_exact, per_rank_* (N ranks);
  cpu_baseline -- the CPU oracle (oracle/pz_oracle.c, a C port of the reference's Python step, kind="port") timed on
                  this host's cores on a bounded sample of the same workload, and used to check the GPU trajectories
                  bit-for-bit on a lane subset; the reference's own Python step as measured in the survey container;
  config       -- workload, batch, tape, kernel, build id, ranks / rccl_ranks / dist_note.
The verbose blocks (every config's full entry, --extra, per-rank rows, what the regimes mean) go to --configs-out FILE,
or to stderr as one line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
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
# the same with the packed state format: groups A and B (32 bytes) read and written; the 4-byte tail only moves with a
# computer player (its boldness is read, the landing point written)
BYTES_PER_ENV_STEP_PACKED = 2 * 32 + 297
BYTES_PER_ENV_STEP_PACKED_AI = 2 * 36 + 297
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
INFINITY_CACHE_BYTES = 256 << 20
ACTION_SEED = 1
BURN_SEED = 2
GRAPH_MIN_LAUNCHES = 2048          # a replay must dwarf its own host-side launch cost (10-16 us)
WRAPPER_TABLE = (0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--num-envs", type=int, default=65536, help="games per GPU")
    ap.add_argument("--launch", choices=["graph", "cabi", "api"], default="graph",
                    help="graph: the K launches captured once in a hipGraph and replayed; cabi: K direct "
                         "C-ABI calls; api: K env.step() calls")
    ap.add_argument("--min-time", type=float, default=0.25, help="minimum duration of the timed region (s)")
    ap.add_argument("--burn-in", type=int, default=4096, help="untimed frames of random play before warm-up")
    ap.add_argument("--p2-computer", action="store_true", help="config 3: rule-based AI on player 2")
    ap.add_argument("--p1-computer", action="store_true", help="rule-based AI on player 1 (not a BASELINE config)")
    ap.add_argument("--wrappers", action="store_true", help="config 5: fused SimplifyAction+RewardByBallPosition")
    ap.add_argument("--no-flight-tables", action="store_true",
                    help="computer player: run the flight predictors in the kernel instead of the HBM look-up tables")
    ap.add_argument("--flight-tables", choices=["both", "power_hit", "none"], default=None,
                    help="computer players: which flight look-up tables the launches use (both: 927 + 82 MB per device, "
                         "power_hit: the 82 MB one, none: everything predicted in the kernel); default both")
    ap.add_argument("--action-dtype", choices=["int32", "int64", "uint8", "int16"], default="int32",
                    help="element type of the action tensors the launches read (pz_action_format)")
    ap.add_argument("--state-format", choices=["int32", "packed"], default="int32",
                    help="headline run: state as int32[44, N] columns (BASELINE's contract, default) or in the packed "
                         "36-byte format (SURVEY 8(f)-3)")
    ap.add_argument("--int16-obs", action="store_true",
                    help="headline run with int16 observations (same values, half the bytes; default int32 like the reference's Box)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall budget of the CPU baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=16, help="upper bound on CPU baseline threads")
    ap.add_argument("--check-lanes", type=int, default=2048, help="lanes replayed on the CPU oracle for parity")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline and every oracle replay")
    ap.add_argument("--no-configs", action="store_true", help="headline only (no configs 2/3/5 block)")
    ap.add_argument("--dist-backend", default=None,
                    help="torch.distributed backend (default nccl = RCCL). 'gloo' + PZ_BENCH_ONE_DEVICE=1 rehearses "
                         "the N>1 path with every rank on cuda:0 of a 1-GPU box")
    ap.add_argument("--extra", action="store_true", help="also time rollouts, launch modes and a batch sweep")
    ap.add_argument("--rollouts", action="store_true", help="only the k-frame kernels (for profiling runs)")
    ap.add_argument("--action-tape", choices=["cold", "hot"], default="cold",
                    help="headline run: every timed launch of a graph replay reads its own action slice (cold: >= 2 048 "
                         "distinct slices = 1 GB streamed from HBM, whatever --steps is -- the default and what every profile "
                         "under profiles/ ran) or the K slices are re-used out of the caches (hot: what a policy that has "
                         "just written its actions presents); the other one is measured beside it")
    ap.add_argument("--configs-out", default=None,
                    help="file for the verbose blocks (every config's full entry, --extra, per-rank rows, what the regimes "
                         "mean); without it they go to stderr as one line.  The stdout line stays below 8 KB")
    return ap.parse_args()


def make_env(shard, device, *, num_envs, p1_computer=False, p2_computer=False, wrappers=False, flight_tables=True,
             state_format="int32", obs16=False, validate_actions=False):
    from pikazoo_amd.wrappers import RewardByBallPosition, SimplifyAction

    env = pikazoo_v0.env(winning_score=15, serve="winner", is_player2_computer=p2_computer,
                         is_player1_computer=p1_computer,
                         num_envs=num_envs, device=device, seed=0, env_id_base=shard.env_id_base,
                         auto_reset=True, validate_actions=validate_actions, flight_tables=flight_tables,
                         state_format=state_format, observation_dtype=torch.int16 if obs16 else torch.int32)
    if wrappers:
        env = SimplifyAction(env)
        env = RewardByBallPosition(env, WRAPPER_TABLE, 216, 176)
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


def burn_in(raw, frames):
    """Untimed random play (k-frame launches of the on-device policy, stream BURN_SEED, t = 0..frames-1)."""
    done = 0
    while done < frames:
        k = min(512, frames - done)
        raw.step_random(BURN_SEED, t0=done, k=k)
        done += k
    raw.steps_done = 0


def graph_repeats(steps):
    """Repetitions of the K-step sequence inside one hipGraph: a replay holds >= GRAPH_MIN_LAUNCHES launches."""
    return max(1, math.ceil(GRAPH_MIN_LAUNCHES / steps))


def run_gpu(env, acts, warmup, steps, launch, min_time):
    """W untimed launches, then the timed unit -- `acts.shape[0] - warmup` launches, one per action slice of the tape
    behind the warm-up slices (a multiple of K: K itself on a hot tape, K x graph_repeats(K) distinct slices on a cold
    one) -- repeated until the timed region lasts >= min_time.

    Returns dict(wall, event_ms, timed_steps, passes): `passes` = how many times the unit's action sequence ran in
    total (untimed calibration pass included) -- what the oracle has to replay."""
    raw = env.unwrapped
    lib = _native.load()
    n = raw.num_envs
    st, cfg = raw._state_ptr, raw._cfg_ref
    o1, o2 = raw._obs[0].data_ptr(), raw._obs[1].data_ptr()
    r1, r2, tm = raw._rew_raw[0].data_ptr(), raw._rew_raw[1].data_ptr(), raw._term_u8.data_ptr()
    tables = raw._tables_ref
    a_ptr = acts.data_ptr()
    elem = acts.element_size()  # the launches read the tape's own element type (pz_config.action_format)
    a_stride = 2 * n * elem
    if acts.dtype != torch.int32:
        cfg_typed = _native.PzConfig.from_buffer_copy(raw._cfg)
        cfg_typed.action_format = _native.ACTION_FORMATS[str(acts.dtype).replace("torch.", "")]
        cfg = C.byref(cfg_typed)

    def launch_range(t_lo, t_hi, stream):
        s = stream.cuda_stream
        for t in range(t_lo, t_hi):
            rc = lib.pz_step(st, n, raw._stride, cfg, a_ptr + t * a_stride, a_ptr + t * a_stride + n * elem, o1, o2, r1, r2, tm,
                             None, tables, s)
            if rc:
                _native.check(rc, "pz_step")

    stream = torch.cuda.Stream(device=raw.device)
    unit_steps = acts.shape[0] - warmup  # launches per timed unit = action slices behind the warm-up ones
    assert unit_steps >= steps and unit_steps % steps == 0
    with torch.cuda.stream(stream):
        if launch == "api":
            names = raw.possible_agents
            # the per-step action dicts a policy would hand over (views built outside the timed loop)
            feed = [{names[0]: acts[t, 0], names[1]: acts[t, 1]} for t in range(warmup + unit_steps)]

            def unit():
                step = env.step
                for t in range(warmup, warmup + unit_steps):
                    step(feed[t])
            for t in range(warmup):
                env.step(feed[t])
        else:
            launch_range(0, warmup, stream)
            if launch == "graph":
                graph_launches = max(unit_steps, steps * graph_repeats(steps))
                stream.synchronize()
                graph = torch.cuda.CUDAGraph()
                # thread_local: the RCCL watchdog thread of a multi-rank run may poll events while we capture
                with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                    for _ in range(graph_launches // unit_steps):  # (hot tape: the K slices again and again)
                        launch_range(warmup, warmup + unit_steps, torch.cuda.current_stream(raw.device))
                # (capture records the launches without running them)
                unit_passes = graph_launches // unit_steps

                def unit():
                    graph.replay()
            else:
                unit_passes = 1

                def unit():
                    launch_range(warmup, warmup + unit_steps, stream)
        # untimed calibration pass (also the graph upload)
        stream.synchronize()
        t0 = time.perf_counter()
        unit()
        stream.synchronize()
        est = time.perf_counter() - t0
        reps = max(1, math.ceil(min_time / max(est, 1e-6)))
        reps = int(dist.all_reduce_max(reps, device=dist.collective_device(raw.device)))  # same count on every rank
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        stream.synchronize()
        dist.barrier()
        torch.cuda.synchronize(raw.device)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(reps):
            unit()
        ev1.record(stream)
        torch.cuda.synchronize(raw.device)
        dist.barrier()
        wall = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1)
    if launch == "api":
        unit_passes = 1
    passes = unit_passes * (reps + 1)
    raw.steps_done = warmup + unit_steps * passes
    return {"wall": wall, "event_ms": ev_ms, "timed_steps": unit_steps * unit_passes * reps, "passes": passes,
            "replays": reps, "launches_per_replay": unit_steps * unit_passes, "unit_steps": unit_steps}


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


def oracle_config(po, raw, p1_computer, p2_computer, wrappers, base):
    return po.make_config(winning_score=15, serve="winner", is_player2_computer=p2_computer,
                          is_player1_computer=p1_computer, simplify_action=wrappers,
                          additional_reward=WRAPPER_TABLE if wrappers else None, seed=0, env_id_base=base)


def oracle_parity(raw, seq, p1_computer, p2_computer, wrappers, lanes, cores):
    """Replay burn-in + warm-up + every pass of the K-step sequence on the first `lanes` games with the
    CPU oracle and compare the full state bit for bit."""
    from oracle import pz_oracle as po

    po.build()
    k = min(lanes, raw.num_envs)
    chk = po.OracleEnv(k, oracle_config(po, raw, p1_computer, p2_computer, wrappers, raw.env_id_base), nthreads=cores)
    chk.reset()
    if seq["burn_in"]:
        chk.rollout_random(BURN_SEED, 0, seq["burn_in"])
    if seq["warmup"]:
        chk.rollout_random(ACTION_SEED, 0, seq["warmup"])
    for _ in range(seq["passes"]):
        chk.rollout_random(ACTION_SEED, seq["warmup"], seq["unit_steps"])
    gpu_state = raw.read_state()[:, :k].cpu().numpy()
    total = seq["burn_in"] + seq["warmup"] + seq["passes"] * seq["unit_steps"]
    return {"parity_lanes_checked": k, "parity_steps_checked": total,
            "parity_bit_exact": bool((gpu_state == chk.state).all())}


def cpu_baseline(args, raw_gpu, p2_computer, wrappers):
    """Oracle timed on the host cores (bounded sample of the same workload)."""
    from oracle import pz_oracle as po

    po.build()
    cores = usable_cores(args.cpu_threads)
    n = raw_gpu.num_envs

    def cfg(base):
        return oracle_config(po, raw_gpu, args.p1_computer, p2_computer, wrappers, base)

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
    return {
        "value": value, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{n} games x {sample_steps} steps of the same workload, OpenMP static lane partition "
                  f"over {cores} threads ({t_spent:.1f} s)",
        "one_core_value": one_core,
        "config1_one_env_steps_per_s": one_env,  # 1 game stepped call by call from Python (ctypes overhead-bound)
    }


def measure(args, shard, device, *, num_envs=None, p2_computer=False, wrappers=False, launch=None, steps=None,
            warmup=None, burn=None, min_time=None, check_lanes=0, flight_tables=True, state_format="int32", obs16=False,
            validate_actions=False, tape="cold", action_dtype=torch.int32):
    """One timed measurement of the single-frame launch.  tape = "cold": every launch of the timed unit reads its own
    action slice (K x graph_repeats(K) distinct slices behind the warm-up ones: >= 1 GB at 65 536 games, streamed from
    HBM); "hot": the K slices are re-used."""
    num_envs = args.num_envs if num_envs is None else num_envs
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    burn = args.burn_in if burn is None else burn
    min_time = args.min_time if min_time is None else min_time
    launch = args.launch if launch is None else launch
    env = make_env(shard, device, num_envs=num_envs, p1_computer=args.p1_computer, p2_computer=p2_computer,
                   wrappers=wrappers, flight_tables=flight_tables, state_format=state_format, obs16=obs16,
                   validate_actions=validate_actions)
    raw = env.unwrapped
    env.reset()
    burn_in(raw, burn)
    unit_steps = steps * graph_repeats(steps) if (tape == "cold" and launch == "graph") else steps
    acts = pregenerate_actions(raw, warmup + unit_steps)
    if action_dtype != torch.int32:
        acts = acts.to(action_dtype)  # (outside every timed region: the tape as a policy of that dtype would leave it)
    torch.cuda.synchronize(device)
    run = run_gpu(env, acts, warmup, steps, launch, min_time)
    cdev = dist.collective_device(device)  # counters live on the GPU under nccl (RCCL), on the CPU under gloo
    wall = dist.all_reduce_max(run["wall"], device=cdev)
    n_total, = dist.all_reduce_sum([raw.num_envs], device=cdev)
    launch_us = run["event_ms"] * 1e3 / run["timed_steps"]
    per_step = BYTES_PER_ENV_STEP
    if state_format == "packed":
        per_step = BYTES_PER_ENV_STEP_PACKED_AI if (p2_computer or args.p1_computer) else BYTES_PER_ENV_STEP_PACKED
    if obs16:
        per_step -= 2 * 35 * 2  # int16 observation rows: 70 instead of 140 bytes per agent
    alg = per_step * num_envs
    res = {
        "wall_s": wall, "event_ms": run["event_ms"], "n_total": n_total, "timed_steps": run["timed_steps"],
        "replays": run["replays"], "launches_per_replay": run["launches_per_replay"],
        "value": n_total * run["timed_steps"] / wall,
        "launch_us": launch_us, "wall_us_per_step": wall * 1e6 / run["timed_steps"],
        "achieved_GBps": alg / (launch_us * 1e-6) / 1e9, "algorithmic_bytes_per_launch": alg,
        "raw": raw, "action_tape": tape, "action_tape_bytes": int(acts.numel() * acts.element_size()),
    }
    res["frac"] = res["achieved_GBps"] / HBM_PEAK_GBPS
    res["frac_wall"] = alg / (res["wall_us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
    if check_lanes and not args.no_cpu:  # (the caller says which ranks check)
        seq = {"burn_in": burn, "warmup": warmup, "unit_steps": run["unit_steps"], "passes": run["passes"]}
        res.update(oracle_parity(raw, seq, args.p1_computer, p2_computer, wrappers, check_lanes,
                                 usable_cores(args.cpu_threads)))
    del acts
    return res


def measure_launch_floor(device, num_envs, frame_steps=102, min_time=0.05):
    """What one launch of the headline's dependent chain is made of: pz_probe_launch -- the single-frame
    pair launch's geometry, LDS and buffers with none of its game logic -- replayed from a hipGraph like the headline
    (2 048 dependent launches per graph, HIP events on the launch stream, >= min_time per figure), on scratch buffers:
    an empty launch; with the launch's loads; with its loads and stores; with `frame_steps` steps of the frame's own
    idiom (cmp, cmp, s_and, cndmask, add) per wave in between -- 102 steps = the 408 VALU instructions a wave of the
    shipped human-vs-human frame issues."""
    sys.path.insert(0, str(REPO / "pika-zoo_amd"))
    import diag  # libpikazoo_diag.so (include/pikazoo_diag.h): not part of the product library

    lib = diag.load()
    n = int(num_envs)
    with torch.cuda.device(device):
        state = torch.zeros((44, n), dtype=torch.int32, device=device)
        acts = torch.zeros((2, n), dtype=torch.int32, device=device)
        obs = [torch.zeros((n, 35), dtype=torch.int32, device=device) for _ in range(2)]
        rew = [torch.zeros(n, dtype=torch.int32, device=device) for _ in range(2)]
        launches = 2048
        out = {"games": n, "launches_per_graph": launches, "frame_steps": frame_steps}
        stream = torch.cuda.Stream(device=device)
        def probe(what, raw_stream):
            return lib.pz_probe_launch(state.data_ptr(), n, n, acts[0].data_ptr(), acts[1].data_ptr(), obs[0].data_ptr(),
                                       obs[1].data_ptr(), rew[0].data_ptr(), rew[1].data_ptr(), what, frame_steps, raw_stream)

        for name, what in (("empty_us", 0), ("loads_us", 1), ("loads_stores_us", 2), ("loads_frame_stand_in_stores_us", 3)):
            with torch.cuda.stream(stream):
                # once eagerly, return code checked: a refused launch never aborts a capture in flight
                _native.check(probe(what, stream.cuda_stream), "pz_probe_launch")
                stream.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                    raw_stream = torch.cuda.current_stream(device).cuda_stream
                    for _ in range(launches):
                        probe(what, raw_stream)
                t0 = time.perf_counter()
                graph.replay()
                stream.synchronize()
                est = time.perf_counter() - t0
                reps = max(2, math.ceil(min_time / max(est, 1e-6)))
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(stream)
                for _ in range(reps):
                    graph.replay()
                ev1.record(stream)
                stream.synchronize()
            out[name] = ev0.elapsed_time(ev1) * 1e3 / (reps * launches)
            del graph
    return out


def measure_rollout(args, shard, device, k, tape=False, p2_computer=False, min_time=0.25, check_lanes=0, obs16=False,
                    num_envs=None):
    """pz_rollout_random (or, tape=True, pz_step_many on a pre-generated action tape): k frames per launch, every
    frame's outputs written to [k][n]... trajectory tensors (state in registers, read / written once per launch).
    Timed like the headline: the launches go through the C ABI into ONE hipGraph (>= 64 launches, each on its own
    step indices t0), one untimed replay, then R replays until the timed region lasts >= min_time; HIP events on the
    launch stream.  Algorithmic bytes per game-step of THESE kernels: 297 (8 of them the action words written resp.
    read) + 352 / k.  `check_lanes`: the first lanes' final state against the CPU oracle replaying the same launches."""
    env = make_env(shard, device, num_envs=args.num_envs if num_envs is None else num_envs, p1_computer=args.p1_computer,
                   p2_computer=p2_computer, obs16=obs16)
    raw = env.unwrapped
    lib = _native.load()
    env.reset()
    burn = min(args.burn_in, 1024)
    burn_in(raw, burn)
    n = raw.num_envs
    launches = max(64, 2048 // k)
    out = raw.rollout_random(ACTION_SEED, k, t0=0)  # allocates the trajectory tensors; an untimed launch (frames 0..k-1)
    placed = dict(raw.trajectory_placement)  # the two observation tensors in different ranks of the HBM?
    # what pure stores of the launch's pattern into THESE two tensors reach (pz_probe_write: nothing in front of the stores)
    from pikazoo_amd import placement

    pure_store_gbps = None  # (a tensor below one probe frame -- a small batch -- has no such figure)
    if min(t.numel() * t.element_size() for t in out["_obs"]) >= int(lib.pz_probe_frame_bytes()):
        with torch.cuda.device(device):
            pure_store_gbps = placement.pair_write_rate(out["_obs"][0], out["_obs"][1])
    tapes = pregenerate_actions(raw, launches * k).view(launches, k, 2, n) if tape else None
    ptrs = (out["_obs"][0].data_ptr(), out["_obs"][1].data_ptr(), out["_rew"][0].data_ptr(), out["_rew"][1].data_ptr(),
            out["_term"].data_ptr())
    eps = raw._episodes.data_ptr()

    def launch_all(stream):
        s = stream.cuda_stream
        for j in range(launches):
            if tape:
                rc = lib.pz_step_many(raw._state_ptr, n, raw._stride, raw._cfg_ref, tapes[j].data_ptr(), k, *ptrs, None, eps,
                                      raw._tables_ref, s)
            else:
                rc = lib.pz_rollout_random(raw._state_ptr, n, raw._stride, raw._cfg_ref, ACTION_SEED, j * k, k,
                                           out["actions"].data_ptr(), *ptrs, None, eps, raw._tables_ref, s)
            if rc:
                _native.check(rc, "k-frame launch")

    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            launch_all(torch.cuda.current_stream(device))
        t0 = time.perf_counter()
        graph.replay()
        stream.synchronize()
        est = time.perf_counter() - t0
        reps = max(1, math.ceil(min_time / max(est, 1e-6)))
        lead_in = 0
        if p2_computer or args.p1_computer:
            # a launch that gathers from the flight tables settles over its first ~100 ms of back-to-back launches (the
            # tables' hot lines find their way into the caches: tools/eager_vs_graph.py, 3.8 -> 3.2 us per frame): untimed
            lead_in = max(1, math.ceil(0.15 / max(est, 1e-6)))
            for _ in range(lead_in):
                graph.replay()
            stream.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(reps):
            graph.replay()
        ev1.record(stream)
        stream.synchronize()
        wall = time.perf_counter() - t0
    frames = reps * launches * k
    us_per_frame = ev0.elapsed_time(ev1) * 1e3 / frames
    bytes_per_step = (157 if obs16 else 297) + 352.0 / k  # (int16 rows: 2 x 70 instead of 2 x 140 bytes)
    gbps = bytes_per_step * n / (us_per_frame * 1e-6) / 1e9
    res = {"value": n * frames / wall, "us_per_frame": us_per_frame, "k": k, "launches_per_replay": launches,
           "replays": reps, "timed_frames": frames, "timed_seconds": wall, "bytes_per_game_step": bytes_per_step,
           "achieved_GBps": gbps, "frac": gbps / HBM_PEAK_GBPS, "regime": TRAJ_REGIME,
           "placement": placed, "pure_store_GBps": pure_store_gbps,
           "frac_of_pure_stores": None if pure_store_gbps is None else gbps / pure_store_gbps}
    if n < 16384:
        res["regime"] = "launch-latency"
    if check_lanes and not args.no_cpu:
        from oracle import pz_oracle as po

        po.build()
        lanes = min(check_lanes, n)
        chk = po.OracleEnv(lanes, oracle_config(po, raw, args.p1_computer, p2_computer, False, raw.env_id_base),
                           nthreads=usable_cores(args.cpu_threads))
        chk.reset()
        chk.rollout_random(BURN_SEED, 0, burn)
        chk.rollout_random(ACTION_SEED, 0, k)  # the allocating launch
        for _ in range(reps + 1 + lead_in):
            for j in range(launches):  # (the tape holds the policy stream's slices j * k ..: the same actions)
                chk.rollout_random(ACTION_SEED, j * k, k)
        res.update(parity_lanes_checked=lanes, parity_steps_checked=burn + k + (reps + 1 + lead_in) * launches * k,
                   parity_bit_exact=bool((raw.read_state()[:, :lanes].cpu().numpy() == chk.state).all()))
    return res


def measure_rollout_api(args, shard, device, k=32, tape=False, p2_computer=False, seconds=0.5):
    """The same k-frame launches issued eagerly through the env API (`rollout_random(out=...)` / `step_many(out=...)`):
    what a training loop that calls the Python API gets, host time included (HIP events around the calls)."""
    env = make_env(shard, device, num_envs=args.num_envs, p1_computer=args.p1_computer, p2_computer=p2_computer)
    raw = env.unwrapped
    env.reset()
    burn_in(raw, min(args.burn_in, 1024))
    tapes = pregenerate_actions(raw, 16 * k).view(16, k, 2, raw.num_envs) if tape else None

    def call(j, out):
        return raw.step_many(tapes[j % 16], out=out) if tape else raw.rollout_random(ACTION_SEED, k, t0=j * k, out=out)

    out = call(0, None)
    placed = dict(raw.trajectory_placement)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for j in range(8):
        out = call(j, out)
    torch.cuda.synchronize(device)
    calls = max(8, int(seconds / ((time.perf_counter() - t0) / 8)))
    # untimed lead-in: a launch that gathers from the flight tables needs ~50 ms of back-to-back launches before its rate
    # settles (the tables' hot lines find their way into the caches: tools/eager_vs_graph.py, 3.8 -> 3.2 us per frame)
    for j in range(calls // 3):
        out = call(j, out)
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    t0 = time.perf_counter()
    for j in range(calls):
        out = call(j, out)
    host = time.perf_counter() - t0
    ev1.record()
    torch.cuda.synchronize(device)
    us = ev0.elapsed_time(ev1) * 1e3 / (calls * k)
    return {"us_per_frame": us, "value": raw.num_envs / (us * 1e-6), "host_us_per_call": host / calls * 1e6,
            "calls": calls, "k": k, "placement": placed}


def measure_policy_in_the_loop(args, shard, device, launches=2048, fused=False):
    """The reference's own benchmark loop, literally: sample both agents' actions, then step -- here the policy
    kernel (pz_random_actions, uniform random like ``action_space.sample()``) and pz_step alternate inside one hipGraph,
    so every step's actions have just been written (cache-hot) and the policy launch is part of the timed region.
    fused=True: the same loop as ONE launch per step -- pz_step_random(k=1) draws the policy inside the step kernel."""
    env = make_env(shard, device, num_envs=args.num_envs, p1_computer=args.p1_computer)
    raw = env.unwrapped
    lib = _native.load()
    env.reset()
    burn_in(raw, min(args.burn_in, 1024))
    n = raw.num_envs
    a1 = torch.empty(n, dtype=torch.int32, device=device)
    a2 = torch.empty_like(a1)
    p = raw._ptrs

    def loop(stream, t0):
        s = stream.cuda_stream
        for t in range(t0, t0 + launches):
            if fused:
                rc = lib.pz_step_random(p[0], n, raw._stride, raw._cfg_ref, ACTION_SEED, t, 1, p[1], p[2], p[3], p[4], p[5],
                                        None, raw._episodes.data_ptr(), raw._tables_ref, s)
            else:
                rc = lib.pz_random_actions(a1.data_ptr(), a2.data_ptr(), n, raw.env_id_base, ACTION_SEED, t, raw.n_actions, s)
                rc = rc or lib.pz_step(p[0], n, raw._stride, raw._cfg_ref, a1.data_ptr(), a2.data_ptr(), p[1], p[2], p[3],
                                       p[4], p[5], None, raw._tables_ref, s)
            if rc:
                _native.check(rc, "policy + step")

    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        loop(stream, 0)
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            loop(torch.cuda.current_stream(device), launches)
        graph.replay()
        stream.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 16
        ev0.record(stream)
        for _ in range(reps):
            graph.replay()
        ev1.record(stream)
        stream.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / (reps * launches)
    return {"value": n / (us * 1e-6), "us_per_step": us, "launches_per_step": 1 if fused else 2,
            "note": ("pz_step_random(k=1): the random policy drawn inside the step launch" if fused else
                     "policy kernel + step kernel per step, both inside the timed region")}


def measure_two_chains(args, shard, device, steps=512, p2_computer=False, check_lanes=512):
    """The one structural lever a dependent-launch chain leaves: the batch as TWO independent sub-batch chains (games
    share nothing, pikazoo_env.py:96-98) whose launches could overlap -- (a) inside ONE hipGraph (fork once at its start,
    join once at its end, no per-step event), (b) as two hipGraphs replayed side by side on two streams.  Each step of
    sub-batch c is one pz_step launch on lanes [c n/2, (c+1) n/2) of the same tensors (pointer offsets, full column
    pitch, env_id_base moved along).  Timed like the headline (HIP events, >= 0.1 s); both halves' first lanes are
    replayed on the oracle.  Measured beside the one-launch headline, never instead of it (on MI355X / ROCm
    7.2 neither form gains -- a 32 768-game launch costs 5.3 us of which only 1.5 scale with its size)."""
    import ctypes as C

    env = make_env(shard, device, num_envs=args.num_envs, p1_computer=args.p1_computer, p2_computer=p2_computer)
    raw = env.unwrapped
    lib = _native.load()
    n, half = raw.num_envs, raw.num_envs // 2
    if half % 64 or half == 0:
        return None
    env.reset()
    burn = min(args.burn_in, 512)
    burn_in(raw, burn)
    acts = pregenerate_actions(raw, steps)
    cfgs = []
    for c in range(2):
        cfg = _native.PzConfig.from_buffer_copy(raw._cfg)
        cfg.env_id_base = raw.env_id_base + c * half
        cfgs.append(cfg)
    p = raw._ptrs  # state, obs1, obs2, rew1, rew2, terminated

    def launch(c, t, stream):
        lo = c * half
        rc = lib.pz_step(p[0] + 4 * lo, half, raw._stride, C.byref(cfgs[c]), acts[t, 0].data_ptr() + 4 * lo,
                         acts[t, 1].data_ptr() + 4 * lo, p[1] + 140 * lo, p[2] + 140 * lo, p[3] + 4 * lo, p[4] + 4 * lo,
                         p[5] + lo, None, raw._tables_ref, stream.cuda_stream)
        if rc:
            _native.check(rc, "pz_step (sub-batch)")

    passes = 0
    out = {}
    # (two priorities: two hardware queues -- streams of one priority may share a queue, and their graphs then run one
    # after the other: 10.7 us per step)
    main_s, side = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device, priority=-1)
    torch.cuda.synchronize(device)
    # (a) one graph, two branches
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main_s):
        with torch.cuda.graph(graph, stream=main_s, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream(device)
            fork = torch.cuda.Event()
            fork.record(cur)
            side.wait_event(fork)
            for t in range(steps):
                launch(0, t, cur)
                launch(1, t, side)
            join = torch.cuda.Event()
            join.record(side)
            cur.wait_event(join)
        graph.replay()
        main_s.synchronize()
        passes += 1
        t0 = time.perf_counter()
        graph.replay()
        main_s.synchronize()
        reps = max(2, math.ceil(0.1 / max(time.perf_counter() - t0, 1e-6)))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(main_s)
        for _ in range(reps):
            graph.replay()
        ev1.record(main_s)
        main_s.synchronize()
        passes += 1 + reps
        out["two_chains_one_graph_us"] = ev0.elapsed_time(ev1) * 1e3 / (reps * steps)
    del graph
    # (b) two graphs on two streams
    streams, graphs = (main_s, side), []
    for c in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(streams[c]):
            with torch.cuda.graph(g, stream=streams[c], capture_error_mode="thread_local"):
                cur = torch.cuda.current_stream(device)
                for t in range(steps):
                    launch(c, t, cur)
        graphs.append(g)
    torch.cuda.synchronize(device)
    e0 = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e0.record(streams[0])
    streams[1].wait_event(e0)
    for _ in range(reps):
        for c in range(2):
            with torch.cuda.stream(streams[c]):
                graphs[c].replay()
    for c in range(2):
        ends[c].record(streams[c])
    torch.cuda.synchronize(device)
    passes += reps
    out["two_graphs_two_streams_us"] = max(e0.elapsed_time(e) for e in ends) * 1e3 / (reps * steps)
    if check_lanes and not args.no_cpu:
        from oracle import pz_oracle as po

        po.build()
        ok = True
        for c in range(2):
            chk = po.OracleEnv(check_lanes, oracle_config(po, raw, args.p1_computer, p2_computer, False,
                                                          raw.env_id_base + c * half), nthreads=usable_cores(args.cpu_threads))
            chk.reset()
            chk.rollout_random(BURN_SEED, 0, burn)
            for _ in range(passes):
                chk.rollout_random(ACTION_SEED, 0, steps)
            ok = ok and bool((raw.read_state()[:, c * half:c * half + check_lanes].cpu().numpy() == chk.state).all())
        out["two_chains_parity_bit_exact"] = ok
    return out


_TRAFFIC = None
_TRAFFIC_STATUS = {}  # workload key -> {"build_id", "stale", "why"}: what load_traffic decided, for the line


def load_traffic(workload_key, num_envs):
    """Fabric-side bytes per launch from the committed PMC profile (profiles/traffic.json) of this workload at this
    batch size, or None: FETCH_SIZE x 2 + WRITE_SIZE, both calibrated on known-byte kernels of the same access widths
    (profiles/r0*_calibration.json).

    The counters were taken on ONE build (the entry's `build_id`).  They are used when the loaded library is that build,
    or when the instruction stream of the entry's kernel is unchanged (`kernel_digest`, tools/kernel_digest.py: a
    change elsewhere in the sources does not invalidate a kernel's counters); otherwise the figure is STALE: the line
    says so (`traffic_stale`) and carries no `traffic` / `frac_traffic` for it."""
    global _TRAFFIC
    if _TRAFFIC is None:
        try:
            _TRAFFIC = json.loads((REPO / "profiles" / "traffic.json").read_text())
        except Exception:  # noqa: BLE001
            _TRAFFIC = {}
    entry = _TRAFFIC.get(workload_key, {})
    if entry.get("num_envs") != num_envs or "hbm_bytes_per_launch" not in entry:
        return None
    status = _TRAFFIC_STATUS.get(workload_key)
    if status is None:
        status = {"build_id": entry.get("build_id"), "stale": False, "why": "same build"}
        if entry.get("build_id") != _native.build_id():
            status.update(stale=True, why="another build and no kernel digest to compare")
            try:
                sys.path.insert(0, str(REPO / "tools"))
                import kernel_digest

                if entry.get("kernel_digest") and kernel_digest.available():
                    same = kernel_digest.digest(entry["kernel"], _native.LIB_PATH) == entry["kernel_digest"]
                    status.update(stale=not same, why="another build, kernel instruction stream " +
                                  ("unchanged" if same else "CHANGED"))
            except Exception as exc:  # noqa: BLE001 - no disassembler: the figure stays stale
                status["why"] = f"another build; digest check failed: {type(exc).__name__}"
        _TRAFFIC_STATUS[workload_key] = status
    return None if status["stale"] else entry["hbm_bytes_per_launch"]


def traffic_key(num_envs, p2_computer=False, wrappers=False, flight_tables=True, state_format="int32", obs16=False,
                p1_computer=False):
    """The profiles/traffic.json entry of a single-frame workload (none was profiled with a computer player 1: another
    kernel, so no counted bytes for it)."""
    if p1_computer:
        return "p1_computer_unprofiled"
    mode = {True: "cfg3", "both": "cfg3", "power_hit": "cfg3_power_hit", False: "cfg3_compute", None: "cfg3_compute",
            "none": "cfg3_compute"}[flight_tables]
    wl = (mode if p2_computer else ("cfg5" if wrappers else "random_random"))
    if num_envs != 65536:
        wl += f"_{num_envs}"
    if state_format == "packed":
        wl = "packed_" + wl
    if obs16:
        wl += "_int16obs"
    return wl


def ensure_built(with_oracle=True):
    """(Re)build libpikazoo_hip.so when it is missing or was not built from the sources in this tree
    (build.py compares the source hash baked into the library; hipcc is on every box of this image) -- and the
    checker, so that no two ranks ever compile it at the same time."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("pz_build", REPO / "pika-zoo_amd" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    if with_oracle:
        from oracle import pz_oracle as po

        po.build()


def kernel_name(num_envs, ai, tables, packed=False):
    if packed:
        return ("pz::step_kernel<AI1,AI2,kActions,false,kNoScout,PACKED> via pz_step" if (ai and not tables)
                else "pz::step_pair_kernel<AI1,AI2,PACKED> via pz_step")
    if num_envs >= 393216:
        return "pz::step_kernel<AI1,AI2,kActions,true> via pz_step"
    if ai and not tables:
        return "pz::step_kernel<AI1,AI2,kActions,true,kScoutLoads> via pz_step"
    return "pz::step_pair_kernel<AI1,AI2> via pz_step"


def regime(num_envs):
    """What a single-frame launch of this batch size runs against.  One launch per step is a chain of dependent
    launches; with fewer workgroups than SIMDs it lasts as long as that chain's fixed latency.  From there on: does
    the per-step working set (state + both observation tensors + rewards, re-touched every launch) stay in the
    256 MiB Infinity Cache, or stream from / to HBM?"""
    if num_envs < 16384:  # < 256 workgroups: not one per CU
        return "launch-latency"
    ws = num_envs * (44 * 4 + 2 * 35 * 4 + 8 + 1 + 8)
    return "infinity-cache-resident" if ws < INFINITY_CACHE_BYTES // 2 else "hbm-streaming"


TRAJ_REGIME = "hbm-streaming, write-dominated"
BOUND_DETAIL = {
    TRAJ_REGIME: "k-frame launches: every frame's outputs stream to HBM (623 MB per 32-frame launch of int32 rows); "
                 "pure_store_GBps is what the launch's store pattern alone sustains on the entry's own two observation "
                 "tensors (pz_probe_write, >= 20 ms in this run; ~7 TB/s when they lie in different ranks of the HBM, "
                 "5.6 when they share one: `placement`, DESIGN 4.9) -- a REFERENCE rate, not a ceiling: the probe stores "
                 "`nt` from one wave per 64 games, the launches `sc0 sc1 nt`, and at k = 128 their own stores run 3-5 % "
                 "above it (frac_of_pure_stores 1.03-1.05); int16 rows are not write-bound",
    "launch-latency": "fewer workgroups than CUs: the launch lasts as long as the dependent-launch chain "
                      "(dispatch + load latency + frame + store acknowledge), not as long as its bytes",
    "infinity-cache-resident": "working set re-touched every launch out of the 256 MiB Infinity Cache: bound by the "
                               "time to the first stores plus each XCD's write drain (DESIGN 4.4), not by HBM",
    "hbm-streaming": "working set past the Infinity Cache: the launch streams HBM (6.3 TB/s of counted bytes is what "
                     "the memory system delivers, MI355X_MICROARCH.md)",
}


def fractions(r, num_envs, key):
    """The roofline figures of one single-frame measurement: on the algorithmic (contract) bytes and on the
    counter-measured bytes of profiles/traffic.json."""
    traffic = load_traffic(key, num_envs)
    launch_s = r["launch_us"] * 1e-6
    reg = regime(num_envs)
    return {"frac": r["frac"], "frac_wall": r["frac_wall"], "traffic": traffic,
            "frac_traffic": (traffic / launch_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
            "traffic_key": key, "regime": reg}  # (what a regime means: `regimes` of the verbose blocks)


def config_entry(r, workload, num_envs, key):
    e = {"workload": workload, "num_envs": num_envs, "value": r["value"], "launch_us": r["launch_us"],
         "wall_us_per_step": r["wall_us_per_step"], "timed_steps": r["timed_steps"],
         "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"], **fractions(r, num_envs, key)}
    for k in ("parity_bit_exact", "parity_lanes_checked", "parity_steps_checked"):
        if k in r:
            e[k] = r[k]
    return e


def relaunch_with_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (one rank per
    GPU), pass its output through and return its exit code.  Nothing in this process has touched the GPU yet, and it
    never replaces itself with another program."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if int(os.environ.get("RANK", "0")) == 0:
        ensure_built(with_oracle=not args.no_cpu)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_with_ranks(args))
    one_device = os.environ.get("PZ_BENCH_ONE_DEVICE") == "1"  # rehearsal: every rank on cuda:0 of a 1-GPU box
    rank, world, local_rank = dist.init_from_env(args.dist_backend, device_index=0 if one_device else None)
    local_rank = 0 if one_device else dist.local_device_index(local_rank)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist.barrier()  # rank 0 may have rebuilt the library: nobody loads it earlier
    shard = dist.weak_shard(args.num_envs, rank, world)
    tables = args.flight_tables if args.flight_tables is not None else ("none" if args.no_flight_tables else "both")
    action_dtype = getattr(torch, args.action_dtype)
    single = rank == 0 and world == 1
    cdev = dist.collective_device(device)

    if args.rollouts:
        out = {"rollout_k32": measure_rollout(args, shard, device, k=32),
               "step_many_k32": measure_rollout(args, shard, device, k=32, tape=True),
               "rollout_k32_p2_computer": measure_rollout(args, shard, device, k=32, p2_computer=True),
               "step_many_k32_p2_computer": measure_rollout(args, shard, device, k=32, tape=True, p2_computer=True),
               "rollout_k32_int16obs": measure_rollout(args, shard, device, k=32, obs16=True)}
        print(json.dumps(out), flush=True)
        return

    # the in-run oracle replay: all of it with one rank; with N ranks the first and the last rank check 1 024 lanes each
    check_lanes = args.check_lanes if single else (min(1024, args.check_lanes) if rank in (0, world - 1) else 0)
    main_res = measure(args, shard, device, p2_computer=args.p2_computer, wrappers=args.wrappers,
                       check_lanes=check_lanes, flight_tables=tables, state_format=args.state_format, obs16=args.int16_obs,
                       tape=args.action_tape, action_dtype=action_dtype)
    raw_main = main_res.pop("raw")
    # every rank's own figures (a straggler GPU is invisible in the MAX-over-ranks wall clock alone)
    parity = main_res.get("parity_bit_exact")
    rows = dist.all_gather_rows([rank, main_res["launch_us"], raw_main.num_envs * main_res["timed_steps"] /
                                 (main_res["event_ms"] * 1e-3), -1.0 if parity is None else float(parity),
                                 raw_main.env_id_base], device=cdev)
    per_rank = [{"rank": int(r[0]), "env_id_base": int(r[4]), "launch_us": r[1], "value": r[2],
                 "parity_bit_exact": None if r[3] < 0 else bool(r[3])} for r in rows]
    cpu = None
    if single and not args.no_cpu:
        cpu = cpu_baseline(args, raw_main, args.p2_computer, args.wrappers)
        for k in ("parity_bit_exact", "parity_lanes_checked", "parity_steps_checked"):
            cpu[k] = main_res.get(k)
    del raw_main

    # what the headline launch is made of, with the same batch size (the plain int32 contract only)
    launch_floor = launch_floor_error = None
    if single and args.state_format == "int32" and not args.int16_obs and args.launch == "graph":
        try:
            launch_floor = measure_launch_floor(device, args.num_envs)
        except Exception as exc:  # noqa: BLE001 - a diagnostic beside the line, never a reason to lose the line
            launch_floor_error = f"{type(exc).__name__}: {exc}"  # (and the line says that it was skipped, and why)
            print(f"[bench] launch_floor skipped: {launch_floor_error}", file=sys.stderr, flush=True)
    # the batch as two independent 32 768-game chains (beside the headline; a diagnostic, never a reason to lose the line)
    pipelined = None
    if single and not args.no_configs and args.state_format == "int32" and not args.int16_obs:
        try:
            pipelined = measure_two_chains(args, shard, device, p2_computer=args.p2_computer)
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] two-chain measurement skipped: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)

    # the same workload on the other kind of action tape (cold: every launch its own slice, >= 1 GB streamed from HBM;
    # hot: K slices re-used out of the caches, what a policy that has just written its actions presents)
    other_tape = None
    if single and not args.no_configs:
        other = "hot" if args.action_tape == "cold" else "cold"
        o_steps = min(args.steps, 20) if other == "hot" else args.steps
        r = measure(args, shard, device, p2_computer=args.p2_computer, wrappers=args.wrappers, flight_tables=tables,
                    state_format=args.state_format, obs16=args.int16_obs, steps=o_steps, warmup=5, burn=512,
                    min_time=0.1, tape=other, action_dtype=action_dtype)
        r.pop("raw")
        other_tape = {"action_tape": other, "steps": o_steps, "action_tape_bytes": r["action_tape_bytes"],
                      "value": r["value"], "launch_us": r["launch_us"], "frac": r["frac"]}

    configs = {}
    if single and not args.no_configs:
        # the other single-GPU BASELINE configs, each timed like the headline (shorter) and oracle-checked
        sub = dict(steps=1000, warmup=100, burn=2048, min_time=0.1, check_lanes=1024, launch="graph")
        specs = {
            "cfg2": ("4 096 games, random/random, winning_score=15, serve=winner", dict(num_envs=4096)),
            "cfg3": ("65 536 games, player 2 = rule-based computer (both flight look-up tables in HBM: 927 + 82 MB)",
                     dict(num_envs=65536, p2_computer=True)),
            "cfg3_power_hit": ("65 536 games, player 2 = rule-based computer, the 82 MB power-hit table only (the landing "
                               "point predicted in the kernel)",
                               dict(num_envs=65536, p2_computer=True, flight_tables="power_hit")),
            "cfg3_compute": ("65 536 games, player 2 = rule-based computer, flight predictors computed in the kernel",
                             dict(num_envs=65536, p2_computer=True, flight_tables="none")),
            "cfg5": ("65 536 games, fused SimplifyAction + RewardByBallPosition",
                     dict(num_envs=65536, wrappers=True)),
            # the same workloads on the packed state format (36 instead of 176 bytes of state per game;
            # `frac` of these entries is computed on their own 361 / 369 algorithmic bytes per game-step)
            "packed_headline": ("65 536 games, random/random, packed state format",
                                dict(num_envs=65536, state_format="packed")),
            "packed_cfg3": ("65 536 games, player 2 = rule-based computer (tables), packed state format",
                            dict(num_envs=65536, p2_computer=True, state_format="packed")),
            "packed_524288": ("524 288 games on one GPU, random/random, packed state format",
                              dict(num_envs=524288, state_format="packed", steps=300, warmup=50, burn=512)),
            "int32_524288": ("524 288 games on one GPU, random/random, int32 state (config 4's batch on one GPU)",
                             dict(num_envs=524288, steps=300, warmup=50, burn=512)),
            # opt-in int16 observations (the values of the int32 ones in half the bytes): 509 / 221 algorithmic bytes
            "int16obs_headline": ("65 536 games, random/random, int32 state, int16 observations",
                                  dict(num_envs=65536, obs16=True)),
            "packed_int16obs_headline": ("65 536 games, random/random, packed state, int16 observations",
                                         dict(num_envs=65536, state_format="packed", obs16=True)),
            "packed_int16obs_524288": ("524 288 games on one GPU, random/random, packed state, int16 observations",
                                       dict(num_envs=524288, state_format="packed", obs16=True, steps=300, warmup=50,
                                            burn=512)),
        }
        # the flight tables are built once per device, outside every timed region: say what that costs
        from pikazoo_amd import env as _env

        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        tables_struct, t_landing, t_power_hit = _env.flight_tables(device)
        torch.cuda.synchronize(device)
        table_info = {"build_ms_once_per_device": (time.perf_counter() - t0) * 1e3,
                      "bytes": int(t_landing.numel() + t_power_hit.numel()),
                      "bytes_power_hit_only": int(t_power_hit.numel())}
        for key, (wl, kw) in specs.items():
            r = measure(args, dist.weak_shard(kw["num_envs"], rank, world), device, **{**sub, **kw})
            r.pop("raw")
            tk = traffic_key(kw["num_envs"], kw.get("p2_computer", False), kw.get("wrappers", False),
                             kw.get("flight_tables", True), kw.get("state_format", "int32"), kw.get("obs16", False),
                             args.p1_computer)
            configs[key] = config_entry(r, wl, kw["num_envs"], tk)
        configs["cfg3"]["flight_tables"] = table_info
        # env.step() with the action tensors a policy hands over: torch's default integer dtype (int64: argmax, multinomial,
        # Categorical.sample, randint) goes into the launch as it is, like int32 (pz_action_format) -- eager, HIP events
        # around the timed loop, the env built with the constructor's defaults (validate_actions on)
        for name, dt in (("api_int32", torch.int32), ("api_int64", torch.int64)):
            r = measure(args, shard, device, launch="api", steps=500, warmup=50, burn=512, min_time=0.1,
                        validate_actions=True, action_dtype=dt)
            r.pop("raw")
            configs[name] = {"workload": f"env.step() eagerly, {dt} action tensors, validate_actions=True",
                             "num_envs": args.num_envs, "launch_us": r["launch_us"], "wall_us_per_step": r["wall_us_per_step"]}
        # the k-frame launches (state in registers, every frame's outputs to [k][N]... tensors), on
        # their own 297 + 352 / k algorithmic bytes per game-step; write-dominated, so the rate a pure fill reaches
        # on this box is printed beside them
        traj = {
            "rollout_k32": ("pz_rollout_random, k = 32: 65 536 games, random policy drawn in the kernel", dict()),
            "step_many_k32": ("pz_step_many, k = 32: 65 536 games, actions from a tape in HBM", dict(tape=True)),
            "rollout_k32_p2_computer": ("pz_rollout_random, k = 32, player 2 = rule-based computer (flight tables): two "
                                        "waves per 64 games", dict(p2_computer=True)),
            "step_many_k32_p2_computer": ("pz_step_many, k = 32, player 2 = rule-based computer (flight tables)",
                                          dict(p2_computer=True, tape=True)),
        }
        traj["rollout_k32_int16obs"] = ("pz_rollout_random, k = 32, int16 observation rows (165 B per game-step): two waves per "
                                        "64 games, each player's wave writing its agent's rows", dict(obs16=True))
        traj["rollout_k32_4096"] = ("pz_rollout_random, k = 32 at config 2's batch size (4 096 games = 64 workgroups: what one "
                                    "launch per frame leaves on the table there)", dict(num_envs=4096))
        traj["rollout_k128"] = ("pz_rollout_random, k = 128 (the launch's fixed costs -- state in and out, the first frame's "
                                "latency before the first store -- over four times as many frames)", dict(k=128))
        for key, (wl, kw) in traj.items():
            r = measure_rollout(args, shard, device, check_lanes=1024, **{"k": 32, **kw})
            # counted bytes of one k-frame launch (profiles/traffic.json: FETCH_SIZE x 2 + WRITE_SIZE of the same kernel)
            tr = None if args.p1_computer else load_traffic(key, kw.get("num_envs", args.num_envs))
            r.update(traffic=tr, frac_traffic=None if tr is None else
                     tr / (r["k"] * r["us_per_frame"] * 1e-6) / 1e9 / HBM_PEAK_GBPS)
            r.update(workload=wl, num_envs=kw.get("num_envs", args.num_envs))
            configs[key] = r
        # the reference's own loop, literally -- sample both agents' actions, then step -- as ONE launch per step
        r = measure_policy_in_the_loop(args, shard, device, fused=True)
        r.update(workload="pz_step_random(k = 1): the uniform random policy drawn inside the step launch",
                 num_envs=args.num_envs, launch_us=r["us_per_step"],
                 frac=BYTES_PER_ENV_STEP * args.num_envs / (r["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                 regime=regime(args.num_envs))
        configs["policy_fused_into_the_step"] = r

    extra = {}
    if args.extra and world == 1:
        extra["policy_in_the_loop"] = measure_policy_in_the_loop(args, shard, device)
        # the same kernel at larger batches (more waves per SIMD hide each other's latency)
        sweep = {}
        for n_big in (262144, 524288, 1048576):
            r = measure(args, dist.weak_shard(n_big, rank, world), device, num_envs=n_big, steps=300, warmup=50,
                        burn=512, min_time=0.1, launch="cabi")
            r.pop("raw")
            sweep[str(n_big)] = {"value": r["value"], "launch_us": r["launch_us"], "achieved_GBps": r["achieved_GBps"],
                                 "frac_of_8TBps": r["frac"], "regime": regime(n_big)}
        extra["batch_sweep_random_random"] = sweep
        # what the host costs per step: direct C-ABI calls and env.step() (eager and through the bound entry point),
        # for the three formats whose kernels differ most in duration
        for fmt, kw in (("int32", {}), ("packed", dict(state_format="packed")),
                        ("packed_int16obs", dict(state_format="packed", obs16=True))):
            for mode in ("cabi", "api"):
                r = measure(args, shard, device, p2_computer=args.p2_computer, wrappers=args.wrappers, launch=mode,
                            burn=512, min_time=0.1, **kw)
                r.pop("raw")
                extra[f"launch_{mode}_{fmt}"] = {"value": r["value"], "launch_us": r["launch_us"],
                                                 "wall_us_per_step": r["wall_us_per_step"]}
        # env.step() of an env built with the constructor's defaults (validate_actions=True: the range check runs inside
        # the launch and is polled without a sync) beside the validate_actions=False figure above
        r = measure(args, shard, device, p2_computer=args.p2_computer, wrappers=args.wrappers, launch="api", burn=512,
                    min_time=0.1, validate_actions=True)
        r.pop("raw")
        extra["launch_api_int32_default_constructor"] = {"value": r["value"], "launch_us": r["launch_us"],
                                                         "wall_us_per_step": r["wall_us_per_step"]}
        # the k-frame launches through the env API, eagerly (no hipGraph): host time per call stays below the launch
        extra["rollout_k32_api"] = measure_rollout_api(args, shard, device)
        extra["step_many_k32_api"] = measure_rollout_api(args, shard, device, tape=True)
        extra["rollout_k32_p2_computer_api"] = measure_rollout_api(args, shard, device, p2_computer=True)

    if rank == 0:
        alg_bytes = main_res["algorithmic_bytes_per_launch"]
        wl = traffic_key(args.num_envs, args.p2_computer, args.wrappers, tables, args.state_format, args.int16_obs,
                         args.p1_computer)
        fr = fractions(main_res, args.num_envs, wl)
        tstat = _TRAFFIC_STATUS.get(wl, {})
        checked = [p for p in per_rank if p["parity_bit_exact"] is not None]
        workload = (f"{args.num_envs} games/GPU, random policy both players (Philox stream in HBM), winning_score=15, "
                    f"serve=winner, auto-reset")  # (kept below the 128 characters the driver's record keeps of a string)
        if args.p1_computer or args.p2_computer or args.wrappers:
            workload += f", p1c={int(args.p1_computer)} p2c={int(args.p2_computer)} wrap={int(args.wrappers)}"
        out = {
            "metric": "env-steps/sec (random policy, 65 536 envs per GPU)",
            "value": main_res["value"], "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main_res["wall_us_per_step"] * 1e-3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "state_format": args.state_format, "observation_dtype": "int16" if args.int16_obs else "int32",
            "data": "synthetic",
            "timed_steps": main_res["timed_steps"], "replays": main_res["replays"],
            "launches_per_replay": main_res["launches_per_replay"], "timed_seconds": main_res["wall_s"],
            "burn_in_frames": args.burn_in,
            # ranks = processes in the job; rccl_ranks = ranks of the RCCL group the counters travelled over (0: they
            # went over gloo -- see dist_note -- which does not touch the measurement: no collective on the step path)
            "ranks": dist.world_size(), "rccl_ranks": dist.rccl_ranks(), "dist_backend": dist.backend_name(),
            "dist_note": dist.fallback_note(), "build_id": _native.build_id(),
            # Everything below is FLAT: scalars and short strings only (the driver's record keeps those and drops nested
            # objects); the full entries go to --configs-out / stderr.
            "config": {
                "workload": workload,
                "num_envs_per_gpu": args.num_envs, "num_envs_total": main_res["n_total"],
                # cold: every launch of a graph replay reads its own action slice (>= 2 048 distinct slices, streamed from
                # HBM whatever --steps is); hot: the K slices re-used out of the caches
                "action_tape": main_res["action_tape"], "action_tape_bytes": main_res["action_tape_bytes"],
                "launch": args.launch,
                "kernel": kernel_name(args.num_envs, args.p2_computer or args.p1_computer, tables != "none",
                                      args.state_format == "packed"),
                "flight_tables": tables, "action_dtype": args.action_dtype,
                "state_format": args.state_format, "observation_dtype": "int16" if args.int16_obs else "int32",
                "build_id": _native.build_id(), "ranks": dist.world_size(), "rccl_ranks": dist.rccl_ranks(),
                "dist_backend": dist.backend_name(), "dist_note": (dist.fallback_note() or "")[:120],
                # an RCCL start that hung was given up on in some rank: its communicator kernel may still hold CUs on that
                # GPU -- every figure of this line is then SUSPECT (pikazoo_amd/dist.py: left_behind_anywhere)
                "rccl_left_behind": dist.left_behind_anywhere(),
                "env_id_base_last_rank": per_rank[-1]["env_id_base"],
            },
            "roofline": {
                # the bound is HBM-side in every regime; which part of the memory system a launch of this batch size
                # actually runs against is the regime (the configs file says what each name means)
                "bound": "hbm", "bound_regime": fr["regime"], "achieved": main_res["achieved_GBps"],
                "rccl_left_behind": dist.left_behind_anywhere(),  # true: the figures below are suspect (see config)
                "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                # `frac` is a CONTRACT-bytes figure: the algorithmic 649 B/game-step over the HIP-event
                # launch duration over 8 TB/s -- not achieved HBM bandwidth: the changed-only write-back moves fewer
                # bytes (`traffic`, `frac_traffic`: PMC counters), and at this batch size they move through the
                # Infinity Cache (`bound_regime`)
                "frac": fr["frac"], "frac_basis": "algorithmic 649 B per game-step / HIP-event launch time / peak",
                "traffic": fr["traffic"], "traffic_key": fr["traffic_key"],
                # the build the PMC counters were taken on; stale = another build AND the kernel's instructions changed
                "traffic_build_id": tstat.get("build_id"), "traffic_stale": bool(tstat.get("stale", False)),
                "traffic_check": tstat.get("why"),
                "algorithmic_bytes_per_launch": alg_bytes, "launch_us": main_res["launch_us"],
                # the same fraction on the wall clock `value` is computed from (launch gaps included)
                "frac_wall": fr["frac_wall"],
                # ... and with the PMC-measured bytes instead of the 649 B/game-step contract figure
                "frac_traffic": fr["frac_traffic"],
                # every rank's own figures, condensed (rows: the configs file / `per_rank`)
                "per_rank_min_value": min(p["value"] for p in per_rank),
                "per_rank_max_launch_us": max(p["launch_us"] for p in per_rank),
                "parity_ranks_checked": len(checked),
                "parity_first_last_rank_bit_exact": (all(p["parity_bit_exact"] for p in checked) and
                                                     per_rank[0]["parity_bit_exact"] is not None and
                                                     per_rank[-1]["parity_bit_exact"] is not None) if checked else None,
            },
        }
        roof = out["roofline"]
        if world > 1:
            out["per_rank"] = per_rank  # (N rows: small; with one rank the flat figures above say it all)

        def put(prefix, entry, us_key="launch_us", what=("frac", "frac_traffic")):
            """One config's figures as flat scalars of `roofline`: <prefix>_launch_us (or _us_per_frame), _frac, ..."""
            if entry is None:
                return
            roof[f"{prefix}_{us_key}"] = round(entry[us_key], 4)
            for k in what:
                if entry.get(k) is not None:
                    roof[f"{prefix}_{k}"] = round(entry[k], 4)

        # the two kinds of action tape, whichever of them the headline ran on
        tapes = {main_res["action_tape"]: {"launch_us": main_res["launch_us"], "frac": fr["frac"]}}
        if other_tape is not None:
            tapes[other_tape["action_tape"]] = other_tape
        for kind in ("cold", "hot"):
            if kind in tapes:
                put(f"{kind}_tape", tapes[kind], what=("frac",))
        # what the headline launch is made of (pz_probe_launch of the diagnostics library)
        if launch_floor is not None:
            roof["floor_empty_us"] = round(launch_floor["empty_us"], 4)
            roof["floor_loads_us"] = round(launch_floor["loads_us"], 4)
            roof["floor_loads_stores_us"] = round(launch_floor["loads_stores_us"], 4)
            roof["floor_stand_in_us"] = round(launch_floor["loads_frame_stand_in_stores_us"], 4)
        elif launch_floor_error is not None:
            roof["floor_error"] = launch_floor_error[:120]
        if configs:
            # every BASELINE config and k-frame launch: the figures README / DESIGN 6 quote, one scalar each
            for prefix, key in (("cfg2", "cfg2"), ("cfg3", "cfg3"), ("cfg3_compute", "cfg3_compute"), ("cfg5", "cfg5"),
                                ("cfg3_power_hit", "cfg3_power_hit"),
                                ("n524288", "int32_524288"), ("packed", "packed_headline"), ("packed_cfg3", "packed_cfg3"),
                                ("packed_524288", "packed_524288"), ("int16obs", "int16obs_headline"),
                                ("packed_int16obs", "packed_int16obs_headline"),
                                ("packed_int16obs_524288", "packed_int16obs_524288"),
                                ("policy_fused", "policy_fused_into_the_step")):
                put(prefix, configs.get(key))
            for key in ("rollout_k32", "step_many_k32", "rollout_k32_p2_computer", "step_many_k32_p2_computer",
                        "rollout_k128", "rollout_k32_4096", "rollout_k32_int16obs"):
                put(key, configs.get(key), us_key="us_per_frame", what=("frac", "frac_traffic", "frac_of_pure_stores"))
            if "cfg3" in configs:
                roof["flight_tables_build_ms_once"] = round(configs["cfg3"]["flight_tables"]["build_ms_once_per_device"], 2)
                # config 3 in the three table modes (flight_tables= of the env) and what each costs in device memory
                for mode, key in (("both", "cfg3"), ("power_hit", "cfg3_power_hit"), ("none", "cfg3_compute")):
                    roof[f"cfg3_tables_{mode}_us"] = round(configs[key]["launch_us"], 4)
                roof["flight_tables_bytes_both"] = configs["cfg3"]["flight_tables"]["bytes"]
                roof["flight_tables_bytes_power_hit"] = configs["cfg3"]["flight_tables"]["bytes_power_hit_only"]
            for name in ("api_int32", "api_int64"):
                if name in configs:
                    roof[f"{name}_us"] = round(configs[name]["launch_us"], 4)
            verdicts = [v.get("parity_bit_exact") for v in configs.values() if "parity_bit_exact" in v]
            roof["configs_parity_checked"] = len(verdicts)
            roof["configs_parity_all_bit_exact"] = bool(verdicts) and all(verdicts)
            roof["configs_traffic_stale"] = sum(1 for st in _TRAFFIC_STATUS.values() if st.get("stale"))
        if pipelined is not None:
            roof.update({k: (round(v, 4) if isinstance(v, float) else v) for k, v in pipelined.items()})
        if cpu is not None:
            # the reference itself never travels to the GPU box; its own Python step, measured in the survey container
            # (Xeon 2.1 GHz, 1 of 8 vCPU)
            cpu["reference_python_steps_per_s_per_core"] = 55100
            out["cpu_baseline"] = cpu
        verbose = {"per_rank": per_rank, "regimes": BOUND_DETAIL}
        if launch_floor is not None:
            verbose["launch_floor"] = launch_floor
        if other_tape is not None:
            verbose["other_action_tape"] = other_tape
        if configs:
            verbose["configs"] = configs
        if extra:
            verbose["extra"] = extra
        verbose["traffic_status"] = _TRAFFIC_STATUS
        if args.configs_out:
            Path(args.configs_out).write_text(json.dumps(verbose, indent=1))
            out["configs_file"] = str(args.configs_out)
        else:
            print("[bench verbose] " + json.dumps(verbose), file=sys.stderr, flush=True)
        line = json.dumps(out)
        assert len(line) < 8000, f"the bench line grew to {len(line)} bytes: the driver's record keeps 8 KB"
        print(line, flush=True)
    # (destroys the process groups; leaves hard -- and non-zero: the line above is flagged, a retry belongs in a fresh
    # process -- when an RCCL start was left behind in some rank)
    dist.shutdown(exit_code=3 if dist.left_behind_anywhere() else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How the k-frame rollout's rate moves under SUSTAINED load (diagnostic): pz_rollout_random (k = 32, 65 536 games)
back to back for `--seconds`, one us-per-frame sample every `--sample` seconds, the device's clocks / power read
beside it (rocm-smi, every 2 s, from a thread), then an idle pause and a few more samples.

    python tools/sustain.py [--seconds 20] [--sample 0.25] [--pause 5] [--k 32] [--p2-computer]

The bench's k-frame entries are timed for 0.25 s each right after seconds of host-side oracle replay (an idle GPU);
this shows what the same kernel does once the box has been writing at 6-7 TB/s for a while.
"""
import argparse
import subprocess
import sys
import threading
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "pika-zoo_amd"))


def smi_loop(stop, rows):
    while not stop.is_set():
        t = time.perf_counter()
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True,
                                 text=True, timeout=10).stdout.strip().splitlines()
            rows.append((t, out))
        except Exception as exc:  # noqa: BLE001 - a diagnostic: say so and go on
            rows.append((t, [f"rocm-smi failed: {exc}"]))
        stop.wait(2.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--sample", type=float, default=0.25)
    ap.add_argument("--pause", type=float, default=5.0)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--num-envs", type=int, default=65536)
    ap.add_argument("--p2-computer", action="store_true")
    ap.add_argument("--eager", action="store_true", help="env.rollout_random() calls instead of C-ABI launches in a hipGraph")
    args = ap.parse_args()
    import torch
    from pikazoo_amd import pikazoo_v0

    env = pikazoo_v0.env(num_envs=args.num_envs, device="cuda:0", seed=0, is_player2_computer=args.p2_computer)
    raw = env.unwrapped
    env.reset()
    from pikazoo_amd import _native

    lib = _native.load()
    k = args.k
    out = raw.rollout_random(1, k, t0=0)
    torch.cuda.synchronize()
    t_launch = time.perf_counter()
    raw.rollout_random(1, k, t0=0)
    torch.cuda.synchronize()
    per_launch = time.perf_counter() - t_launch
    launches = max(1, int(args.sample / per_launch))
    print(subprocess.run(["rocm-smi", "--showclkfrq"], capture_output=True, text=True).stdout, flush=True)
    graph = None
    if not args.eager:
        ptrs = (out["_obs"][0].data_ptr(), out["_obs"][1].data_ptr(), out["_rew"][0].data_ptr(), out["_rew"][1].data_ptr(),
                out["_term"].data_ptr())
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                cs = torch.cuda.current_stream().cuda_stream
                for j in range(launches):
                    rc = lib.pz_rollout_random(raw._state_ptr, raw.num_envs, raw._stride, raw._cfg_ref, 1, j * k, k,
                                               out["actions"].data_ptr(), *ptrs, None, raw._episodes.data_ptr(),
                                               raw._tables_ref, cs)
                    assert rc == 0, rc
    stop, rows = threading.Event(), []
    th = threading.Thread(target=smi_loop, args=(stop, rows), daemon=True)
    th.start()
    t_start = time.perf_counter()

    def sample(tag, j0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if graph is not None:
            graph.replay()
        else:
            for j in range(launches):
                raw.rollout_random(1, k, t0=(j0 + j) * k)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (launches * k)
        print(f"{tag} t={time.perf_counter() - t_start:6.2f} s  {us:6.3f} us/frame "
              f"({(297 + 352 / k) * args.num_envs / us / 1e3:7.1f} GB/s)", flush=True)

    j0 = 0
    while time.perf_counter() - t_start < args.seconds:
        sample("load ", j0)
        j0 += launches
    print(f"idle for {args.pause} s", flush=True)
    time.sleep(args.pause)
    for _ in range(6):
        sample("after", j0)
        j0 += launches
    stop.set()
    th.join(timeout=15)
    for t, out in rows:
        print(f"smi t={t - t_start:6.2f} s")
        for line in out:
            print("   ", line)


if __name__ == "__main__":
    main()

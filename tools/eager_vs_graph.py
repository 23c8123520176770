#!/usr/bin/env python3
"""pz_rollout_random with a computer player: the same launches replayed from a hipGraph and issued eagerly, alternating
(diagnostic).  python tools/eager_vs_graph.py [--human]"""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "pika-zoo_amd"))
from pikazoo_amd import pikazoo_v0  # noqa: E402

human = "--human" in sys.argv
k, n = 32, 65536
env = pikazoo_v0.env(num_envs=n, device="cuda:0", seed=0, is_player2_computer=not human)
raw = env.unwrapped
env.reset()
out = raw.rollout_random(7, k, t0=0)
for j in range(20):
    out = raw.rollout_random(7, k, t0=j * k, out=out)
torch.cuda.synchronize()
print("placement", raw.trajectory_placement)
launches = 64
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        for j in range(launches):
            out = raw.rollout_random(7, k, t0=j * k, out=out)


def timed(fn, stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (launches * k)


def eager():
    global out
    for j in range(launches):
        out = raw.rollout_random(7, k, t0=j * k, out=out)


for rnd in range(5):
    tg = timed(g.replay, side)
    te = timed(eager, side)
    td = timed(eager, torch.cuda.current_stream())
    print(f"round {rnd}: graph {tg:.3f}  eager on the side stream {te:.3f}  eager on the default stream {td:.3f} us per frame", flush=True)

#!/usr/bin/env python3
"""Quick start: 65 536 games of Pikachu Volleyball on one MI355X through the PettingZoo-parallel API.

    python pika-zoo_amd/build.py        # once: hipcc --offload-arch=gfx950
    python examples/quickstart.py
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "pika-zoo_amd"))

from pikazoo_amd import pikazoo_v0  # noqa: E402
from pikazoo_amd.wrappers import NormalizeObservation, RecordEpisodeStatistics, SimplifyAction  # noqa: E402


def main():
    n = 65536
    # same kwargs as the reference's pikazoo_v0.env(...) + the batched ones
    env = pikazoo_v0.env(winning_score=15, serve="winner", is_player2_computer=True,
                         num_envs=n, device="cuda:0", seed=0, validate_actions=False)
    env = RecordEpisodeStatistics(NormalizeObservation(SimplifyAction(env)))  # fused into the step kernel
    obs, infos = env.reset()
    print("agents:", env.agents, "| obs", tuple(obs["player_1"].shape), obs["player_1"].dtype,
          "| actions: Discrete(%d)" % env.action_space("player_1").n)

    steps = 2000
    finished = torch.zeros((), dtype=torch.int64, device="cuda:0")  # keep the bookkeeping on the device:
    ret = torch.zeros((), dtype=torch.float32, device="cuda:0")     # no host sync inside the loop
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        # any policy producing int tensors [n] works; here: uniform random for player_1 (player_2 is the AI)
        actions = {a: torch.randint(0, 13, (n,), dtype=torch.int32, device="cuda:0") for a in env.agents}
        obs, rewards, terminations, truncations, infos = env.step(actions)
        done = terminations["player_1"]
        finished += done.sum()
        ret += torch.where(done, infos["player_1"]["episode"]["r"], 0.0).sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    finished, ret = int(finished), float(ret)
    print(f"{n * steps / dt / 1e9:.2f} G env-steps/s incl. the torch.randint policy and the bookkeeping; "
          f"{finished} episodes finished, mean return of player_1 {ret / max(finished, 1):+.2f}")

    # open-loop throughput: k frames of the on-device random policy per launch, all outputs kept
    raw = env.unwrapped
    out = raw.rollout_random(action_seed=1, k=32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        out = raw.rollout_random(action_seed=1, k=32, out=out)
    torch.cuda.synchronize()
    print(f"rollout_random: {n * 32 * 50 / (time.perf_counter() - t0) / 1e9:.2f} G env-steps/s; "
          f"trajectory obs {tuple(out['obs']['player_1'].shape)}")


if __name__ == "__main__":
    main()

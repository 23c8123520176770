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
        # any policy producing integer tensors [n] works -- int32, or what torch gives by default: int64 (argmax,
        # multinomial, Categorical.sample), uint8, int16: the launch reads them as they are --; here: the env's own seeded
        # device policy stream for player_1
        # (player_2 is the rule-based AI; its action entry is read but does not steer it)
        actions = env.unwrapped.random_actions(action_seed=7)
        obs, rewards, terminations, truncations, infos = env.step(actions)
        done = terminations["player_1"]
        finished += done.sum()
        ret += torch.where(done, infos["player_1"]["episode"]["r"], 0.0).sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    finished, ret = int(finished), float(ret)
    print(f"{n * steps / dt / 1e9:.2f} G env-steps/s through env.step() incl. the policy launch and the torch bookkeeping "
          f"(host-bound: five small torch kernels per step); "
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

    # a larger batch in the packed state format (36 instead of 176 bytes of state per game; same results) and with int16
    # observations (same values, half the bytes): this is where the step launch streams HBM, and fewer bytes are less time
    for fmt, odt in (("int32", torch.int32), ("packed", torch.int32), ("packed", torch.int16)):
        big = pikazoo_v0.env(num_envs=524288, device="cuda:0", seed=0, validate_actions=False, state_format=fmt,
                             observation_dtype=odt)
        big.reset()
        acts = big.random_actions(action_seed=7)
        for _ in range(20):
            big.step(acts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            big.step(acts)
        torch.cuda.synchronize()
        print(f"524 288 games, state_format={fmt}, observations {str(odt).split('.')[-1]}: "
              f"{524288 * 200 / (time.perf_counter() - t0) / 1e9:.1f} G env-steps/s")
        del big

    # rgb_array frames of a few games, drawn on the GPU from the state.  The sprites are the reference's PNG files:
    # point sprite_dir at <reference install>/pikazoo/env/img (found by itself when `pikazoo` is importable);
    # without them this demo falls back to a synthetic sprite set of the same geometry.
    from pikazoo_amd import render as pz_render

    img_dir = pz_render.default_image_dir()
    sprites = pz_render.load_sprites(img_dir, "cuda:0") if img_dir else pz_render.synthetic_sprites(0, "cuda:0")
    # (scenery=True adds the clouds and waves -- and, like the reference's render(), lets rendering advance the env RNG)
    viewer = pikazoo_v0.env(num_envs=1024, device="cuda:0", seed=0, render_mode="rgb_array", sprites=sprites,
                            is_player1_computer=True, is_player2_computer=True, scenery=True)
    viewer.reset()
    viewer.step_random(action_seed=3, k=200)
    frames = viewer.render(lanes=[0, 1, 2, 3])
    print("render:", tuple(frames.shape), frames.dtype, "(reference sprites)" if img_dir else "(synthetic sprites)")


if __name__ == "__main__":
    main()

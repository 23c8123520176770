#!/usr/bin/env python3
"""Long parity run on the GPU box (test infrastructure beyond the pytest suite; not collected by pytest): 65 536 games stepped through the single-frame
launch (`pz_step`: the pair kernel / scout kernel, actions from HBM) for tens of thousands of frames per
configuration, full state compared with the CPU oracle every `--every` frames on EVERY lane.

    python tests/soak.py [--frames 20000] [--every 2000] [--n 65536] [--packed] [--rollout K [--tape]] [--only TEXT] [--lib FILE]

--rollout K: the same frames through the k-frame launches instead (`pz_rollout_random`: the policy drawn in the
kernel, every frame's outputs to trajectory tensors; with --tape `pz_step_many` on the same actions as a tape).

--packed: the same on the packed state format (36 bytes per game); its sticky misfit flags are checked with every unpack.
--only TEXT: the configurations whose name contains TEXT.  --lib FILE: step through a DIAGNOSTIC build of the library
instead of the product (e.g. pika-zoo_amd/lib/ab_<name>.so of tools/ab.py --build: round 5 ran the one-computer
configurations on a build whose computer's wave is held back in front of its loads, so that every launch takes the
late-store path of pair_body's hand-shake); the product's build-id check is bypassed for it, nothing else changes.
Every run also tracks, frame by frame on the device, the extremes of the values the packed format stores in narrow
fields (ball y velocity: 13 bits signed; player y velocity: 6 bits signed) -- printed per configuration.
"""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "pika-zoo_amd"))
sys.path.insert(0, str(REPO))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import pz_oracle as po  # noqa: E402
from pikazoo_amd import pikazoo_v0  # noqa: E402
from pikazoo_amd.wrappers import (RecordEpisodeStatistics, RewardByBallPosition, RewardInNormalState,  # noqa: E402
                                  SimplifyAction)

TABLE = (0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01)


def main():
    args = sys.argv[1:]
    frames = int(args[args.index("--frames") + 1]) if "--frames" in args else 20000
    every = int(args[args.index("--every") + 1]) if "--every" in args else 2000
    n = int(args[args.index("--n") + 1]) if "--n" in args else 65536
    fmt = "packed" if "--packed" in args else "int32"
    rollout = int(args[args.index("--rollout") + 1]) if "--rollout" in args else 0
    tape = "--tape" in args
    if "--lib" in args:
        from pikazoo_amd import _native

        class _AnyBuild:  # (a diagnostic library carries another build id than the sources': that is its point)
            library_id = staticmethod(lambda path: "diagnostic")
            source_id = staticmethod(lambda: "diagnostic")

        _native.LIB_PATH = Path(args[args.index("--lib") + 1]).resolve()
        _native._pz_build = lambda: _AnyBuild
        print(f"stepping through the diagnostic library {_native.LIB_PATH.name}", flush=True)
    only = args[args.index("--only") + 1] if "--only" in args else ""
    configs = [
        ("human_vs_human", dict(), dict(), None),
        ("config 3: p2 computer, flight tables", dict(is_player2_computer=True), dict(is_player2_computer=True), None),
        ("p1 computer, flight tables", dict(is_player1_computer=True), dict(is_player1_computer=True), None),
        ("config 3: p2 computer, computed predictors", dict(is_player2_computer=True, flight_tables=False),
         dict(is_player2_computer=True), None),
        ("config 3: p2 computer, power-hit table only", dict(is_player2_computer=True, flight_tables="power_hit"),
         dict(is_player2_computer=True), None),
        ("both computer, power-hit table only", dict(is_player1_computer=True, is_player2_computer=True,
                                                     flight_tables="power_hit"),
         dict(is_player1_computer=True, is_player2_computer=True), None),
        ("both computer, random serve, tables", dict(is_player1_computer=True, is_player2_computer=True, serve="random"),
         dict(is_player1_computer=True, is_player2_computer=True, serve="random"), None),
        ("config 5 + RewardInNormalState + RecordEpisodeStatistics", dict(), dict(), "wrappers"),
    ]
    po.build()
    ok = True
    for name, kw, okw, wr in configs:
        if only not in name:
            continue
        env = pikazoo_v0.env(num_envs=n, device="cuda:0", seed=123, env_id_base=1 << 34, validate_actions=False,
                             state_format=fmt, **kw)
        ocfg_kw = dict(seed=123, env_id_base=1 << 34, **okw)
        if wr:
            env = RecordEpisodeStatistics(RewardInNormalState(RewardByBallPosition(SimplifyAction(env), TABLE), -0.001))
            ocfg_kw.update(simplify_action=True, additional_reward=TABLE, normal_state_reward=-0.001,
                           normal_state_outside=True, episode_stats=2)
        raw = env.unwrapped
        ref = po.OracleEnv(n, po.make_config(**ocfg_kw), nthreads=16)
        env.reset(), ref.reset()
        t0 = time.perf_counter()
        eps = 0
        # ball y velocity is observation word 33, the players' y velocities words 2 and 15 (raw observations only)
        track = not raw._cfg.normalize_obs
        ext = torch.zeros(2, dtype=torch.int32, device=raw.device)
        traj = None
        for start in range(0, frames, every):
            for t in range(start, start + every, rollout or 1):
                if rollout:
                    k = min(rollout, start + every - t)
                    if tape:
                        acts = torch.stack([torch.stack(tuple(raw.random_actions(77, t + j).values())) for j in range(k)])
                        traj = raw.step_many(acts, out=traj)
                    else:
                        traj = raw.rollout_random(77, k, t0=t, out=traj)
                    obs = traj["obs"]["player_1"].reshape(-1, 35)
                else:
                    obs = env.step(raw.random_actions(77, t))[0]["player_1"]
                if track:
                    ext = torch.maximum(ext, torch.stack((obs[:, 33].abs().max(), obs[:, [2, 15]].abs().max())))
            eps += ref.rollout_random(77, start, every)
            same = np.array_equal(raw.state.cpu().numpy(), ref.state)
            if wr:
                same = same and np.array_equal(raw.episode_returns.cpu().numpy(), ref.episode_returns) and \
                    np.array_equal(raw.episode_lengths.cpu().numpy(), ref.episode_lengths)
            if not same:
                ok = False
                print(f"MISMATCH {name}: after {start + every} frames", flush=True)
                break
        print(f"{name}: {n} games x {start + every} frames = {n * (start + every) / 1e9:.2f} G game-steps bit-exact vs oracle "
              f"on every lane: {same}; {eps} episodes finished; max |ball y velocity| {int(ext[0])}, max |player y velocity| "
              f"{int(ext[1])} over every frame; state format {fmt}; launches: "
              f"{('pz_step_many' if tape else 'pz_rollout_random') + f' k={rollout}' if rollout else 'pz_step'}; "
              f"{time.perf_counter() - t0:.0f} s", flush=True)
    print("SOAK", "PASSED" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

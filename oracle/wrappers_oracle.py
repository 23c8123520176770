"""CPU restatement of the reference's wrapper classes as operations on a step's OUTPUTS (test infrastructure).

The oracle proper (pz_oracle.c) restates the wrapper stacks the step kernel can fuse as branches of its step.  The
reference composes its wrappers in any order; for the orders that cannot be fused the product's wrapper classes apply
themselves to the outputs of a step with torch operations.  This module is what those are checked against on the CPU:
every wrapper of a stack, fusable or not, applied here in numpy -- float64 like the reference's Python arithmetic -- in
the stack's own order, on top of the oracle configured WITHOUT any fused wrapper.

Only tests/ may import this (like everything under oracle/).  Pinned by tests/golden/unfused_*.npz (captured from the
reference by oracle/ref_capture.py) and, on the fusable stacks, by the fused oracle itself.
"""
from __future__ import annotations

import numpy as np

from . import pz_oracle as po

# pikazoo_env.py:485-562 (player, opponent, ball)
_PLAYER_LOW = [32, 108, -15, -1, -2, 0, 0, 0, 0, 0, 0, 0, 0]
_PLAYER_HIGH = [400, 244, 16, 1, 3, 4, 4, 1, 1, 1, 1, 1, 1]
OBS_LOW = np.array(_PLAYER_LOW * 2 + [20, 0, 0, 0, 0, 0, -20, -124, 0], dtype=np.float64)
OBS_HIGH = np.array(_PLAYER_HIGH * 2 + [432, 252, 432, 252, 432, 252, 20, 124, 1], dtype=np.float64)
ACTION_MAP = ((0, 1, 2, 3, 4, 6, 7, 10, 11, 12, 13, 14, 16),   # wrappers/simplify_action.py:17
              (0, 1, 2, 4, 3, 7, 6, 10, 12, 11, 13, 15, 17))   # :18


class WrappedOracle:
    """``stack``: [(reference wrapper class name, kwargs), ...] innermost first (oracle.ref_capture.wrapper_stack).

    ``reset()`` -> (obs_p1, obs_p2); ``step(a1, a2)`` -> (obs [2][n, 35], rew [2][n] float64, term [n] uint8,
    episode) with ``episode`` = None or {"r": float64 [2, n], "l": int64 [n]} of the OUTERMOST RecordEpisodeStatistics
    (meaningful where ``term``), as the harness around the reference does it: a finished game is ``reset()`` right before
    its next step."""

    def __init__(self, num_envs: int, stack, **env_kwargs):
        self.stack = [(name, dict(kw)) for name, kw in stack]
        self.env = po.OracleEnv(num_envs, po.make_config(auto_reset=True, **env_kwargs))
        self.n = num_envs
        # per-wrapper state, in stack order
        self.state = []
        low, high = OBS_LOW.copy(), OBS_HIGH.copy()
        for name, kw in self.stack:
            st = {}
            if name == "NormalizeObservation":  # normalize_observation.py:13-16: the bounds of the env it wraps
                st["low"], st["high"] = low, high
                low, high = np.zeros(35), np.ones(35)  # :34-35, what the next one above it reads
            elif name == "RecordEpisodeStatistics":  # record_episode_statistics.py:15-16
                st["r"] = np.zeros((2, num_envs), np.float64)
                st["l"] = np.zeros(num_envs, np.int64)
            self.state.append(st)
        self._ended = np.zeros(num_envs, bool)

    @property
    def raw_state(self):
        return self.env.state

    def _obs_out(self, obs):
        o = [obs[0].astype(np.float64), obs[1].astype(np.float64)]
        integral = True
        for (name, _), st in zip(self.stack, self.state):
            if name == "NormalizeObservation":  # normalize_observation.py:22,30
                o = [(x - st["low"]) / (st["high"] - st["low"]) for x in o]
                integral = False
        return o, integral

    def reset(self):
        o1, o2 = self.env.reset()
        for (name, _), st in zip(self.stack, self.state):
            if name == "RecordEpisodeStatistics":  # record_episode_statistics.py:23-25
                st["r"][:] = 0.0
                st["l"][:] = 0
        self._ended[:] = False
        (o1, o2), _ = self._obs_out((o1, o2))
        return o1, o2

    def step(self, a1, a2):
        a = [np.asarray(a1, np.int64), np.asarray(a2, np.int64)]
        for name, _ in reversed(self.stack):  # actions travel from the outermost wrapper down
            if name == "SimplifyAction":      # simplify_action.py:22-23
                a = [np.array(ACTION_MAP[i], np.int64)[a[i]] for i in range(2)]
        # the harness resets a finished game right before its next step (the oracle: in place, auto_reset): every
        # RecordEpisodeStatistics zeroes its sums in that reset (:23-25)
        for (name, _), st in zip(self.stack, self.state):
            if name == "RecordEpisodeStatistics":
                st["r"][:, self._ended] = 0.0
                st["l"][self._ended] = 0
        obs, rew, term = self.env.step(a[0].astype(np.int32), a[1].astype(np.int32))
        term = term.copy()
        o = [obs[0].astype(np.float64), obs[1].astype(np.float64)]
        r = [rew[0].astype(np.float64), rew[1].astype(np.float64)]
        episode = None
        for (name, kw), st in zip(self.stack, self.state):  # outputs travel from the env up
            if name == "NormalizeObservation":
                o = [(x - st["low"]) / (st["high"] - st["low"]) for x in o]
            elif name == "RewardByBallPosition":  # reward_by_ball_position.py:22-29: player_1's observation, as it is HERE
                table = np.asarray(kw["additional_reward"], np.float64)
                zone = (o[0][:, 27] > kw.get("y_line", 176)).astype(np.int64) + 2 * (o[0][:, 26] >= kw.get("x_line", 216))
                r = [r[i] + table[i * 4 + zone] for i in range(2)]
            elif name == "RewardInNormalState":  # reward_in_normal_state.py:12-14
                r = [np.where(x == 0, float(kw["reward"]), x) for x in r]
            elif name == "RecordEpisodeStatistics":  # record_episode_statistics.py:30-39
                st["r"] += np.stack(r)
                st["l"] += 1
                episode = {"r": st["r"].copy(), "l": st["l"].copy()}
        self._ended = term.astype(bool)
        return o, r, term, episode

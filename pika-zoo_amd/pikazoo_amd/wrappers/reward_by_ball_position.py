"""RewardByBallPosition (reference: pikazoo/wrappers/reward_by_ball_position.py:6-31).

Every step (the terminal one included) each agent's reward gets ``additional_reward[i*4 + zone]``
added, ``zone = int(ball_y > y_line) + 2*int(ball_x >= x_line)`` from the post-step ball position,
``i`` = 0 for player_1 and 1 for player_2.

Normally **fused**: the add runs in the step kernel's epilogue in float32, so the rewards of a wrapped env are
``float32[num_envs]``.  Where the stack order cannot be a kernel branch -- a second RewardByBallPosition, one above
``NormalizeObservation`` (which, as in the reference, then compares the NORMALIZED coordinates ``obs[26], obs[27]`` with
the lines, :22-24), one above statistics of a wrapped reward or above another wrapper that runs outside the kernel -- the
same three lines run here on the step's outputs (``fused`` is False), float32 like the fused form.
"""
from __future__ import annotations

import torch

from .base import BaseParallelWrapper


class RewardByBallPosition(BaseParallelWrapper):
    def __init__(self, env, additional_reward, x_line: int = 216, y_line: int = 176):
        super().__init__(env)
        assert len(additional_reward) == 8  # reward_by_ball_position.py:15
        self.x_line = x_line
        self.y_line = y_line
        self.additional_reward = additional_reward
        raw = env.unwrapped
        self.fused = raw._fuse_ballpos_reward(additional_reward, x_line, y_line)
        if not self.fused:
            raw._note_unfused("RewardByBallPosition", reward=True)
            self._table = torch.tensor([float(v) for v in additional_reward], dtype=torch.float32, device=raw.device).view(2, 4)

    def step(self, actions):
        out = self.env.step(actions)
        if self.fused:
            return out
        obs, rews, terms, truncs, infos = out
        agents = self.possible_agents
        row = obs[agents[0]]  # reward_by_ball_position.py:22: the ball as player_1's observation shows it
        if self.unwrapped.scalar_api:
            zone = int(row[27] > self.y_line) + 2 * int(row[26] >= self.x_line)
            rews = {a: rews[a] + self.additional_reward[i * 4 + zone] for i, a in enumerate(agents)}
        else:
            zone = (row[:, 27] > self.y_line).to(torch.int64) + 2 * (row[:, 26] >= self.x_line).to(torch.int64)
            rews = {a: rews[a].to(torch.float32) + self._table[i][zone] for i, a in enumerate(agents)}
        return obs, rews, terms, truncs, infos

"""RewardByBallPosition (reference: pikazoo/wrappers/reward_by_ball_position.py:6-31), fused.

Every step (the terminal one included) each agent's reward gets ``additional_reward[i*4 + zone]``
added, ``zone = int(ball_y > y_line) + 2*int(ball_x >= x_line)`` from the post-step ball position,
``i`` = 0 for player_1 and 1 for player_2.  The add runs in the kernel epilogue in float32, so the
rewards of a wrapped env are ``float32[num_envs]``.
"""
from __future__ import annotations

from .base import BaseParallelWrapper


class RewardByBallPosition(BaseParallelWrapper):
    def __init__(self, env, additional_reward, x_line: int = 216, y_line: int = 176):
        super().__init__(env)
        assert len(additional_reward) == 8  # reward_by_ball_position.py:15
        self.x_line = x_line
        self.y_line = y_line
        self.additional_reward = additional_reward
        env.unwrapped._fuse_ballpos_reward(additional_reward, x_line, y_line)

"""RewardInNormalState (reference: pikazoo/wrappers/reward_in_normal_state.py:5-15), fused.

Every frame, an agent's reward of exactly 0 is replaced by the constant ``reward``.  The result of
the reference depends on where the wrapper sits relative to ``RewardByBallPosition`` (the zero test
runs before or after ``additional_reward`` is added); the fused kernel keeps that order
(``pz_config.normal_state_mode`` 1 = inside, 2 = outside).  Rewards of a wrapped env are float32.
"""
from __future__ import annotations

from .base import BaseParallelWrapper


class RewardInNormalState(BaseParallelWrapper):
    def __init__(self, env, reward):
        super().__init__(env)
        self.reward = reward
        env.unwrapped._fuse_normal_state_reward(reward)

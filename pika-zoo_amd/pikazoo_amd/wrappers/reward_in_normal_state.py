"""RewardInNormalState (reference: pikazoo/wrappers/reward_in_normal_state.py:5-15).

Every frame, an agent's reward of exactly 0 is replaced by the constant ``reward``.  The result of
the reference depends on where the wrapper sits relative to ``RewardByBallPosition`` (the zero test
runs before or after ``additional_reward`` is added).

Normally **fused**: the kernel keeps that order (``pz_config.normal_state_mode`` 1 = inside, 2 = outside) and the
rewards of a wrapped env are float32.  A second instance, or one above statistics of a wrapped reward / above a wrapper
that runs outside the kernel, applies the same test here on the step's outputs (``fused`` is False).
"""
from __future__ import annotations

import torch

from .base import BaseParallelWrapper


class RewardInNormalState(BaseParallelWrapper):
    def __init__(self, env, reward):
        super().__init__(env)
        self.reward = reward
        raw = env.unwrapped
        self.fused = raw._fuse_normal_state_reward(reward)
        if not self.fused:
            raw._note_unfused("RewardInNormalState", reward=True)

    def step(self, actions):
        out = self.env.step(actions)
        if self.fused:
            return out
        obs, rews, terms, truncs, infos = out
        if self.unwrapped.scalar_api:  # reward_in_normal_state.py:12-14
            rews = {a: (self.reward if rews[a] == 0 else rews[a]) for a in self.possible_agents}
        else:
            rews = {a: torch.where(rews[a] == 0, float(self.reward), rews[a].to(torch.float32)) for a in self.possible_agents}
        return obs, rews, terms, truncs, infos

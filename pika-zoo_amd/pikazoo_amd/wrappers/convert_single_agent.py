"""ConvertSingleAgent (reference: pikazoo/wrappers/convert_single_agent.py:5-28).

Exposes one side as a single-agent env: ``step(action)`` takes that side's action only and returns
that side's ``(obs, reward, terminated, truncated, info)``.  The reference draws the other side's
action with ``action_space(other).sample()`` (an unseeded third-party sampler); here it comes from the
env's seeded device policy stream (Philox, ``opponent_seed``), so runs are reproducible.
"""
from __future__ import annotations

from .base import BaseParallelWrapper


class ConvertSingleAgent(BaseParallelWrapper):
    def __init__(self, env, side: str, opponent_seed: int = 0):
        super().__init__(env)
        assert side in ("player_1", "player_2")  # convert_single_agent.py:8
        self.side = side
        self.other_side = "player_1" if side == "player_2" else "player_2"
        self.opponent_seed = int(opponent_seed)

    def reset(self, seed=None, options=None, **kw):
        obs, infos = self.env.reset(seed=seed, options=options, **kw)
        return obs[self.side], infos[self.side]

    def step(self, action):
        raw = self.env.unwrapped
        sampled = raw.random_actions(self.opponent_seed)[self.other_side]
        if raw.scalar_api:
            sampled = int(sampled[0].item())
        obs, rews, terms, truncs, infos = self.env.step({self.side: action, self.other_side: sampled})
        return obs[self.side], rews[self.side], terms[self.side], truncs[self.side], infos[self.side]

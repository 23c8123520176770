"""NormalizeObservation (reference: pikazoo/wrappers/normalize_observation.py:8-35), fused.

Observations become ``float32`` ``(obs - low) / (high - low)`` with the bounds of the env's
``observation_space`` (pikazoo_env.py:485-562), written directly by the step kernel (a policy network
can consume them without a cast).  The reference divides in float64 and declares a float32 Box; the
kernel's IEEE float32 quotient equals the float32 rounding of that double for every reachable value.
"""
from __future__ import annotations

from .base import BaseParallelWrapper


class NormalizeObservation(BaseParallelWrapper):
    def __init__(self, env):
        super().__init__(env)
        raw = env.unwrapped
        self.high = {a: raw.observation_space(a).high for a in raw.possible_agents}
        self.low = {a: raw.observation_space(a).low for a in raw.possible_agents}
        raw._fuse_normalize_obs()

    def observation_space(self, agent):
        return self.env.unwrapped.normalized_observation_space(agent)  # Box(0, 1, (35,), float32)

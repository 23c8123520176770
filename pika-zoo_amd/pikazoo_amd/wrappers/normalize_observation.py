"""NormalizeObservation (reference: pikazoo/wrappers/normalize_observation.py:8-35).

Observations become ``float32`` ``(obs - low) / (high - low)`` with the bounds of the wrapped env's
``observation_space`` (pikazoo_env.py:485-562).  Normally **fused**: written directly by the step kernel (a policy network
can consume them without a cast).  The reference divides in float64 and declares a float32 Box; the
kernel's IEEE float32 quotient equals the float32 rounding of that double for every reachable value.

Outside the kernel (``fused`` False) in two cases: the env was created with ``observation_dtype=torch.int16`` (the
kernel's float32 rows need the int32 buffers) -- the same quotient is then taken here; and a second instance, whose
bounds are the first one's Box(0, 1) (normalize_observation.py:15-16 reads the WRAPPED env's space): ``(obs - 0) / (1 - 0)``,
the identity.
"""
from __future__ import annotations

import torch

from .base import BaseParallelWrapper


class NormalizeObservation(BaseParallelWrapper):
    def __init__(self, env):
        super().__init__(env)
        raw = env.unwrapped
        self.high = {a: env.observation_space(a).high for a in raw.possible_agents}  # (of the wrapped env: :15-16)
        self.low = {a: env.observation_space(a).low for a in raw.possible_agents}
        self.fused = raw._fuse_normalize_obs()
        if not self.fused:
            raw._note_unfused("NormalizeObservation")
            self._lo = {a: torch.as_tensor(self.low[a], dtype=torch.float32, device=raw.device) for a in raw.possible_agents}
            self._range = {a: torch.as_tensor(self.high[a], dtype=torch.float32, device=raw.device) - self._lo[a]
                           for a in raw.possible_agents}

    def _normalize(self, obs):
        if self.unwrapped.scalar_api:
            return {a: (obs[a] - self.low[a]) / (self.high[a] - self.low[a]) for a in obs}
        return {a: (obs[a].to(torch.float32) - self._lo[a]) / self._range[a] for a in obs}

    def reset(self, seed=None, options=None, **kw):
        obs, infos = self.env.reset(seed=seed, options=options, **kw)
        return (obs if self.fused else self._normalize(obs)), infos

    def step(self, actions):
        out = self.env.step(actions)
        if self.fused:
            return out
        return (self._normalize(out[0]),) + tuple(out[1:])

    def observation_space(self, agent):
        return self.env.unwrapped.normalized_observation_space(agent)  # Box(0, 1, (35,), float32)

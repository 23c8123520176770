"""RecordEpisodeStatistics (reference: pikazoo/wrappers/record_episode_statistics.py:9-40).

Per game the running episode return of each agent (float64) and the episode length.  On the frame a game
terminates, ``infos[agent]["episode"] = {"r": return, "l": length}`` holds that episode's totals --
batched: ``[num_envs]`` tensors that are meaningful on the lanes where ``terminations[agent]`` is
True (``scalar_api`` envs get the reference's Python scalars, only on terminal steps).  The sums cover
the rewards as seen at the wrapper's position in the stack (inside or outside the reward wrappers).

Normally **fused**: the kernel keeps the three words per game (20 bytes next to the state, zeroed by ``reset`` and by
the in-place auto reset).  Above a reward wrapper that runs outside the kernel (or as a second instance) the same sums
are kept here with torch operations on the step's outputs (``fused`` False): same ``infos``, same reset rules.

Numerics: the returns are summed in float64 like the reference's Python floats (record_episode_statistics.py:31).
Directly on the env they are exact integers.  Above a float reward wrapper the addends are float32
rewards (each within 1e-8 of the reference's float64 reward), so the totals agree with the reference within 2e-6
over the reference fixtures' episodes.
"""
from __future__ import annotations

import torch

from .base import BaseParallelWrapper


class RecordEpisodeStatistics(BaseParallelWrapper):
    def __init__(self, env):
        super().__init__(env)
        raw = env.unwrapped
        self.fused = raw._fuse_episode_stats()
        if not self.fused:
            raw._note_unfused("RecordEpisodeStatistics")
            n = raw.num_envs
            self._returns = torch.zeros((2, n), dtype=torch.float64, device=raw.device)
            self._lengths = torch.zeros(n, dtype=torch.int32, device=raw.device)
            self._ended = torch.zeros(n, dtype=torch.bool, device=raw.device)  # terminated by the previous step

    @property
    def episode_rewards(self):
        raw = self.env.unwrapped
        return dict(zip(raw.possible_agents, raw.episode_returns if self.fused else self._returns))

    @property
    def episode_lengths(self):
        raw = self.env.unwrapped
        return {a: (raw.episode_lengths if self.fused else self._lengths) for a in raw.possible_agents}

    def reset(self, seed=None, options=None, mask=None, **kw):
        out = self.env.reset(seed=seed, options=options, **({} if mask is None else {"mask": mask}), **kw)
        if not self.fused:  # record_episode_statistics.py:23-25
            gone = (torch.ones_like(self._ended) if mask is None
                    else torch.as_tensor(mask, device=self._ended.device).to(torch.bool))
            self._returns.masked_fill_(gone.unsqueeze(0), 0.0)
            self._lengths.masked_fill_(gone, 0)
            self._ended &= ~gone
        return out

    def step(self, actions):
        out = self.env.step(actions)
        if self.fused:
            return out
        obs, rews, terms, truncs, infos = out
        raw = self.unwrapped
        agents = raw.possible_agents
        if raw.auto_reset:  # a game that ended on the previous step was reset in place before this frame
            self._returns.masked_fill_(self._ended.unsqueeze(0), 0.0)
            self._lengths.masked_fill_(self._ended, 0)
            counted = None
        else:               # ... or stays frozen: nothing is counted for it
            counted = ~self._ended
        if raw.scalar_api:
            add = torch.tensor([[float(rews[a])] for a in agents], dtype=torch.float64, device=self._returns.device)
            ended = torch.tensor([bool(terms[agents[0]])], device=self._ended.device)
        else:
            add = torch.stack([rews[a].to(torch.float64) for a in agents])
            ended = terms[agents[0]]
        if counted is None:
            self._returns += add
            self._lengths += 1
        else:
            self._returns += add * counted
            self._lengths += counted.to(torch.int32)
        self._ended = self._ended | ended if counted is not None else ended.clone()
        if raw.scalar_api:
            infos = {a: dict(infos[a]) for a in infos}
            if bool(ended[0]):  # record_episode_statistics.py:34-39
                for i, a in enumerate(agents):
                    if a in infos:
                        infos[a]["episode"] = {"r": self._returns[i, 0].item(), "l": int(self._lengths[0].item())}
        else:
            infos = {a: dict(infos[a], episode={"r": self._returns[i], "l": self._lengths}) for i, a in enumerate(agents)}
        return obs, rews, terms, truncs, infos

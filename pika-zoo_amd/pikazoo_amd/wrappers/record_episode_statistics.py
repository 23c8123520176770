"""RecordEpisodeStatistics (reference: pikazoo/wrappers/record_episode_statistics.py:9-40), fused.

Per game the kernel keeps the running episode return of each agent (float64) and the episode length (20 bytes
next to the state, zeroed by ``reset`` and by the in-place auto reset).  On the frame a game
terminates, ``infos[agent]["episode"] = {"r": return, "l": length}`` holds that episode's totals --
batched: ``[num_envs]`` tensors that are meaningful on the lanes where ``terminations[agent]`` is
True (``scalar_api`` envs get the reference's Python scalars, only on terminal steps).  The sums cover
the rewards as seen at the wrapper's position in the stack (inside or outside the reward wrappers).

Numerics: the returns are summed in float64 like the reference's Python floats (record_episode_statistics.py:31).
Directly on the env they are exact integers.  Above a float reward wrapper the addends are the kernel's float32
rewards (each within 1e-8 of the reference's float64 reward), so the totals agree with the reference within 2e-6
over the reference fixtures' episodes.
"""
from __future__ import annotations

from .base import BaseParallelWrapper


class RecordEpisodeStatistics(BaseParallelWrapper):
    def __init__(self, env):
        super().__init__(env)
        env.unwrapped._fuse_episode_stats()

    @property
    def episode_rewards(self):
        raw = self.env.unwrapped
        return dict(zip(raw.possible_agents, raw.episode_returns))

    @property
    def episode_lengths(self):
        raw = self.env.unwrapped
        return {a: raw.episode_lengths for a in raw.possible_agents}

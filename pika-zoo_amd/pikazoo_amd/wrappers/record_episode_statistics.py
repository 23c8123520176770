"""RecordEpisodeStatistics (reference: pikazoo/wrappers/record_episode_statistics.py:9-40), fused.

Per game the kernel keeps the running episode return of each agent and the episode length (three
words next to the state, zeroed by ``reset`` and by the in-place auto reset).  On the frame a game
terminates, ``infos[agent]["episode"] = {"r": return, "l": length}`` holds that episode's totals --
batched: ``[num_envs]`` tensors that are meaningful on the lanes where ``terminations[agent]`` is
True (``scalar_api`` envs get the reference's Python scalars, only on terminal steps).  The sums cover
the rewards as seen at the wrapper's position in the stack (inside or outside the reward wrappers).

Numerics: directly on the env the sums are exact int32.  Above a float reward wrapper the kernel adds float32
rewards in float32, where the reference adds Python floats (float64, record_episode_statistics.py:31): the two
agree within 2e-4 over the reference fixtures' episodes (tests/test_gpu_parity.py) and drift apart slowly with the
episode length.
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

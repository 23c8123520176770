"""SimplifyAction (reference: pikazoo/wrappers/simplify_action.py:7-28), fused into the kernel.

Actions become relative (FRONT/BACK) and drop the 5 moves that are meaningless in play: 18 -> 13
per side.  The two 13-entry remap tables (player_1: 0,1,2,3,4,6,7,10,11,12,13,14,16; player_2:
0,1,2,4,3,7,6,10,12,11,13,15,17) are composed with the key table at compile time inside
``pz_physics.hpp``; the wrapper only flips ``pz_config.simplify_action``.
"""
from __future__ import annotations

from .base import BaseParallelWrapper


class SimplifyAction(BaseParallelWrapper):
    def __init__(self, env):
        super().__init__(env)
        env.unwrapped._fuse_simplify_action()

    def action_space(self, agent):
        return self.env.unwrapped.action_spaces[agent]  # Discrete(13), simplify_action.py:20,27

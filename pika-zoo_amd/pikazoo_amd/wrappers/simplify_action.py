"""SimplifyAction (reference: pikazoo/wrappers/simplify_action.py:7-28).

Actions become relative (FRONT/BACK) and drop the 5 moves that are meaningless in play: 18 -> 13
per side.  Normally **fused**: the two 13-entry remap tables (player_1: 0,1,2,3,4,6,7,10,11,12,13,14,16; player_2:
0,1,2,4,3,7,6,10,12,11,13,15,17) are composed with the key table at compile time inside
``pz_physics.hpp``; the wrapper only flips ``pz_config.simplify_action``.

A second SimplifyAction on the same env maps the first one's 13 actions again (simplify_action.py:22-23: the reference
raises IndexError where the first map's result is 13 or more).  That runs here (``fused`` False): the actions go
through the table on the device, an entry the reference would fail on becomes an out-of-range action, which the env's
``validate_actions`` reports as that same IndexError.
"""
from __future__ import annotations

import torch

from .base import BaseParallelWrapper

ACTION_MAP = {"player_1": (0, 1, 2, 3, 4, 6, 7, 10, 11, 12, 13, 14, 16),
              "player_2": (0, 1, 2, 4, 3, 7, 6, 10, 12, 11, 13, 15, 17)}  # simplify_action.py:17-18


class SimplifyAction(BaseParallelWrapper):
    def __init__(self, env):
        super().__init__(env)
        raw = env.unwrapped
        self.fused = raw._fuse_simplify_action()
        if not self.fused:
            raw._note_unfused("SimplifyAction")
            if not raw.validate_actions:
                import warnings

                warnings.warn("an un-fused SimplifyAction maps a device action outside its 13-tuple to an out-of-range "
                              "action of the env below, and only that env's validate_actions reports it (as the "
                              "reference's IndexError): this env was built with validate_actions=False, so such an "
                              "action will step a game with an undefined input unnoticed", RuntimeWarning, stacklevel=2)
            # entry 13: what an action outside [0, 13) maps to -- invalid below as well
            self._maps = {a: torch.tensor(ACTION_MAP[a] + (255,), dtype=torch.int32, device=raw.device) for a in ACTION_MAP}

    def step(self, actions):
        if self.fused:
            return self.env.step(actions)
        raw = self.unwrapped
        mapped = {}
        for a in raw.possible_agents:
            v = actions[a]  # KeyError on a missing agent
            if isinstance(v, torch.Tensor) and v.device == raw.device:
                # An index outside [0, 13) becomes the out-of-range action 255, which the env's validate_actions reports
                # as the reference's IndexError (without it nothing reports it: an un-fused SimplifyAction on device
                # tensors relies on validate_actions, the constructor's default).  Negative indices too: the reference's
                # tuple indexing would count -13 .. -1 from the end -- a Python accident, not an interface, and the fused
                # form of this wrapper (and the env's own 18 actions) reject them; one meaning for both forms.
                idx = v.to(torch.int64)
                idx = torch.where((idx < 0) | (idx > 12), 13, idx)
                mapped[a] = self._maps[a][idx]
            else:  # host values: checked before anything is launched, like the env checks its own
                import numpy as np

                flat = np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v).reshape(-1)
                if flat.size and (int(flat.min()) < 0 or int(flat.max()) > 12):
                    raise IndexError("action out of range [0, 13)")
                out = np.array([ACTION_MAP[a][int(x)] for x in flat], dtype=np.int64)
                mapped[a] = int(out[0]) if np.ndim(v) == 0 else out
        return self.env.step(mapped)

    def action_space(self, agent):
        return self.env.unwrapped.action_spaces[agent]  # Discrete(13), simplify_action.py:20,27

"""``BaseParallelWrapper``: keeps ``env``, forwards the ParallelEnv API and any other attribute to it.

The reference's wrappers subclass ``pettingzoo.utils.BaseParallelWrapper`` (pikazoo/wrappers/simplify_action.py:3,7),
and downstream libraries test ``isinstance(env, BaseParallelWrapper)`` / ``isinstance(env, ParallelEnv)``: when
PettingZoo is importable its class is the base of this one (the same try-import as ``spaces.py`` and ``env.py``), so the
wrappers of this package pass those tests; the forwarding below stays this package's own either way (PettingZoo's
``reset`` takes no ``mask=``, and ``state`` is a tensor property here, not a method).  The build image has no
PettingZoo: there the base is a plain class.
"""
from __future__ import annotations

try:  # pragma: no cover - pettingzoo is not installed in the build image
    from pettingzoo.utils import BaseParallelWrapper as _PettingZooWrapper  # type: ignore
except Exception:  # noqa: BLE001
    try:  # pragma: no cover
        from pettingzoo.utils.wrappers import BaseParallelWrapper as _PettingZooWrapper  # type: ignore
    except Exception:  # noqa: BLE001
        _PettingZooWrapper = object


class BaseParallelWrapper(_PettingZooWrapper):
    def __init__(self, env):
        # (not PettingZoo's constructor: it copies `possible_agents` / `metadata` into the wrapper once -- here every
        # attribute the wrapper does not define itself is read through to the env, live)
        self.env = env

    def __getattr__(self, name):
        if name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    # attributes a PettingZoo base class defines itself (class attributes / properties shadow __getattr__): forwarded
    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def metadata(self):
        return self.env.metadata

    @property
    def possible_agents(self):
        return self.env.possible_agents

    @property
    def agents(self):
        return self.env.agents

    @property
    def num_agents(self):
        return self.env.num_agents

    @property
    def max_num_agents(self):
        return self.env.max_num_agents

    @property
    def state(self):
        return self.env.state

    def reset(self, seed=None, options=None, **kw):
        return self.env.reset(seed=seed, options=options, **kw)

    def step(self, actions):
        return self.env.step(actions)

    def step_random(self, *a, **kw):
        return self.env.step_random(*a, **kw)

    def observation_space(self, agent):
        return self.env.observation_space(agent)

    def action_space(self, agent):
        return self.env.action_space(agent)

    def render(self):
        return self.env.render()

    def close(self):
        return self.env.close()

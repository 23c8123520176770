"""Stand-in for ``pettingzoo.utils.BaseParallelWrapper`` (pettingzoo is not a dependency here):
keeps ``env``, forwards the ParallelEnv API and any other attribute to it."""
from __future__ import annotations


class BaseParallelWrapper:
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

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

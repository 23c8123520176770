"""Minimal `Discrete` / `Box` spaces.

gymnasium's classes are used when gymnasium is importable (so `isinstance` checks in user code
keep working); otherwise these stand-ins expose the attributes RL code reads (`n`, `low`, `high`,
`shape`, `dtype`, `contains`, `sample`).
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is not installed in the build image
    from gymnasium.spaces import Box, Discrete  # type: ignore
except Exception:  # noqa: BLE001

    class Discrete:
        def __init__(self, n: int, seed=None):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return int(self._rng.integers(0, self.n))

        def contains(self, x) -> bool:
            try:
                return 0 <= int(x) < self.n
            except (TypeError, ValueError):
                return False

        __contains__ = contains

        def __eq__(self, other):
            return isinstance(other, Discrete) and other.n == self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            low = np.asarray(low, dtype=self.dtype)
            high = np.asarray(high, dtype=self.dtype)
            if shape is not None:
                low = np.broadcast_to(low, shape).copy()
                high = np.broadcast_to(high, shape).copy()
            self.low, self.high = low, high
            self.shape = tuple(low.shape)
            self._rng = np.random.default_rng(seed)

        def sample(self):
            if np.issubdtype(self.dtype, np.integer):
                return self._rng.integers(self.low, self.high + 1).astype(self.dtype)
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape[-len(self.shape):] == self.shape and bool(
                np.all(x >= self.low) and np.all(x <= self.high))

        __contains__ = contains

        def __eq__(self, other):
            return (isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype
                    and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

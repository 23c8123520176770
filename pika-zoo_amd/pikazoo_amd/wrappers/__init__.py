"""Wrappers with the reference's names and constructor signatures (pikazoo/wrappers/__init__.py).

``SimplifyAction`` and ``RewardByBallPosition`` do not post-process on the host: they switch on the
corresponding fused branch of the HIP step kernel.
"""
from .base import BaseParallelWrapper
from .reward_by_ball_position import RewardByBallPosition
from .simplify_action import SimplifyAction

__all__ = ["BaseParallelWrapper", "SimplifyAction", "RewardByBallPosition"]

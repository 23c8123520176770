"""Wrappers with the reference's names and constructor signatures (pikazoo/wrappers/__init__.py).

All six classes of the reference are here.  Five of them do not post-process on the host: they switch
on the corresponding fused branch of the HIP step kernel (``ConvertSingleAgent`` is a thin view).
"""
from .base import BaseParallelWrapper
from .convert_single_agent import ConvertSingleAgent
from .normalize_observation import NormalizeObservation
from .record_episode_statistics import RecordEpisodeStatistics
from .reward_by_ball_position import RewardByBallPosition
from .reward_in_normal_state import RewardInNormalState
from .simplify_action import SimplifyAction

__all__ = ["BaseParallelWrapper", "SimplifyAction", "RewardByBallPosition", "RewardInNormalState",
           "NormalizeObservation", "RecordEpisodeStatistics", "ConvertSingleAgent"]

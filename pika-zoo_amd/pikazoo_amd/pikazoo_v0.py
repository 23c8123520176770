"""Entry point with the reference's name: ``pikazoo_v0.env(...)`` (pikazoo/pikazoo_v0.py:1-3)."""
from .env import env, raw_env

__all__ = ["env", "raw_env"]

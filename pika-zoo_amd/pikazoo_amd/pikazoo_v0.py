"""``pikazoo_v0`` -- the entry point users import, under the reference's name.

    from pikazoo_amd import pikazoo_v0
    env = pikazoo_v0.env(num_envs=65536, device="cuda:0", is_player2_computer=True)

``env(**kwargs)`` builds the batched MI355X environment (:class:`pikazoo_amd.env.raw_env`); the keyword
arguments of the reference (``winning_score``, ``serve``, ``is_player1_computer``, ``is_player2_computer``,
``render_mode``) keep their meaning, the batched ones (``num_envs``, ``device``, ``seed``, ``env_id_base``,
``auto_reset``, ``validate_actions``, ``scalar_api``, ``flight_tables``) are documented on the class.
"""
from . import env as _batched

raw_env = _batched.raw_env


def env(**kwargs) -> "_batched.raw_env":
    """Construct the environment; mirrors ``pikazoo_v0.env(**kwargs)`` of the reference."""
    return raw_env(**kwargs)


__all__ = ["env", "raw_env"]

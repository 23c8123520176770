"""pikazoo_amd -- MI355X-native batched Pikachu Volleyball (drop-in for pika-zoo's step path).

    from pikazoo_amd import pikazoo_v0
    env = pikazoo_v0.env(num_envs=65536, device="cuda:0", is_player2_computer=True)
"""
from ._version import VERSION, __version__  # noqa: F401

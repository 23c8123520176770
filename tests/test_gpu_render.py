"""GPU test of ``pz_render`` (render_mode="rgb_array"): frames drawn by the HIP kernel from the state tensor ==
the numpy frame oracle (oracle/render_oracle.py), bit for bit, on synthetic sprites of the reference's geometry
(the reference's PNG assets do not travel to the GPU box)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle_frames(raw, sprite_set, lanes):
    from oracle import render_oracle as ro

    st = raw.state.cpu().numpy()
    return np.stack([ro.frame(st[:, l], sprite_set.sprites_host, sprite_set.background_host) for l in lanes])


def test_frames_match_the_numpy_oracle():
    from pikazoo_amd import pikazoo_v0
    from pikazoo_amd.render import synthetic_sprites

    n = 192
    sprites = synthetic_sprites(7, "cuda:0")
    env = pikazoo_v0.env(num_envs=n, device="cuda:0", seed=4, render_mode="rgb_array", sprites=sprites,
                         is_player1_computer=True, is_player2_computer=True, winning_score=15)
    env.reset()
    seen_power = seen_flip = 0
    for chunk in range(6):
        env.step_random(3, k=37)
        frames = env.render()
        assert frames.shape == (n, 304, 432, 3) and frames.dtype == torch.uint8
        want = _oracle_frames(env, sprites, range(n))
        got = frames.cpu().numpy()
        if not np.array_equal(got, want):
            l, y, x, c = np.argwhere(got != want)[0]
            pytest.fail(f"chunk {chunk} lane {l} pixel ({x},{y}) channel {c}: hip {got[l, y, x, c]} != oracle {want[l, y, x, c]}")
        st = env.state
        seen_power += int((st[30] != 0).sum())
        seen_flip += int(((st[3] == 3) | (st[3] == 4) | (st[16] == 3) | (st[16] == 4)).sum())
    assert seen_power > 0 and seen_flip > 0  # hyper ball / trail and mirrored diving sprites were drawn
    # two-digit scores, a ball leaving the screen at the top / right edge, the hyper-ball rotation slot
    st = env.state.clone()
    st[38, :64] = torch.arange(64, device=st.device, dtype=torch.int32) % 16
    st[39, :64] = 15 - torch.arange(64, device=st.device, dtype=torch.int32) % 16
    st[26, :32], st[27, :32] = 432, 0
    st[35, 32:64] = 50
    env.state.copy_(st)
    lanes = torch.arange(0, 64, 3, device="cuda:0")
    frames = env.render(lanes=lanes)
    assert frames.shape[0] == lanes.numel()
    assert np.array_equal(frames.cpu().numpy(), _oracle_frames(env, sprites, lanes.tolist()))
    with pytest.raises(IndexError):
        env.render(lanes=[n])


def test_render_api_behaviour():
    from pikazoo_amd import pikazoo_v0
    from pikazoo_amd.render import synthetic_sprites

    sprites = synthetic_sprites(1, "cuda:0")
    one = pikazoo_v0.env(num_envs=1, scalar_api=True, render_mode="rgb_array", sprites=sprites, seed=2)
    one.reset()
    img = one.render()
    assert isinstance(img, np.ndarray) and img.shape == (304, 432, 3) and img.dtype == np.uint8  # pikazoo_env.py:378-384
    plain = pikazoo_v0.env(num_envs=2)
    with pytest.warns(UserWarning):
        assert plain.render() is None                                                          # :355-357
    with pytest.raises(NotImplementedError):
        pikazoo_v0.env(num_envs=2, render_mode="human")
    nosprites = pikazoo_v0.env(num_envs=2, render_mode="rgb_array")
    from pikazoo_amd import render as R
    if R.default_image_dir() is None:
        with pytest.raises(FileNotFoundError):
            nosprites.render()
    big = pikazoo_v0.env(num_envs=4096, render_mode="rgb_array", sprites=sprites)
    with pytest.raises(ValueError):
        big.render()
    assert big.render(lanes=[5, 4095]).shape == (2, 304, 432, 3)
    assert "rgb_array" in big.metadata["render_modes"]

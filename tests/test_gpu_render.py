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
    # a ball bounced off the net top above the screen (the reference's ball y goes negative), hyper ball and trail too
    st[27, 8:24], st[32, 8:24], st[34, 8:24], st[30, 8:24] = -12, -40, -110, 1
    env.set_state(st)
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
    # render(out=): the kernel writes m * 304 * 432 * 3 bytes through the pointer, so only exactly that buffer passes
    good = torch.empty((2, 304, 432, 3), dtype=torch.uint8, device="cuda:0")
    assert big.render(lanes=[5, 4095], out=good) is good
    assert torch.equal(good, big.render(lanes=[5, 4095]))
    for bad in (torch.empty((2, 304, 432), dtype=torch.uint8, device="cuda:0"),            # a channel short
                torch.empty((3, 304, 432, 3), dtype=torch.uint8, device="cuda:0"),         # wrong frame count
                torch.empty((2, 304, 432, 3), dtype=torch.int32, device="cuda:0"),         # wrong dtype
                torch.empty((2, 304, 432, 3), dtype=torch.uint8),                          # host memory
                torch.empty((2, 304, 432, 6), dtype=torch.uint8, device="cuda:0")[..., ::2]):  # not contiguous
        with pytest.raises(ValueError):
            big.render(lanes=[5, 4095], out=bad)


@pytest.mark.parametrize("name,fmt", [("render_human_human", "int32"), ("render_p2_computer", "packed")])
def test_rendering_with_scenery_follows_the_reference_render(name, fmt):
    """The reference's own render() (recorded by oracle/ref_capture.capture_render: constructor with a render_mode,
    render after reset and every few steps, per lane) replayed through the product: the 44 state words after every step
    and after every frame (rendering advances the env RNG), the clouds / waves and the punch effect's radius / y after
    every frame -- against the reference; the frames themselves against the numpy oracle (whose draw lists
    tests/test_render_cpu.py pins to the reference's)."""
    from conftest import load_golden
    from oracle import render_oracle as ro
    from pikazoo_amd import pikazoo_v0
    from pikazoo_amd.render import synthetic_sprites

    d = load_golden(name)
    meta = d["meta"]
    lanes, steps, periods = meta["lanes"], meta["steps"], meta["periods"]
    sprites = synthetic_sprites(11, "cuda:0")
    env = pikazoo_v0.env(num_envs=lanes, device="cuda:0", seed=meta["seed"], env_id_base=meta["env_id_base"],
                         render_mode="rgb_array", sprites=sprites, scenery=True, validate_actions=False,
                         state_format=fmt, **meta["env_kwargs"])
    assert np.array_equal(env.state.cpu().numpy(), d["state_ctor"])
    assert np.array_equal(env._scenery[:71, :lanes].cpu().numpy(), d["scenery_ctor"])
    env.reset()
    assert np.array_equal(env.state.cpu().numpy(), d["state0"])
    k = checked = 0

    def render(due, t):
        nonlocal k, checked
        frames = env.render(lanes=due).cpu().numpy()
        st = env.state.cpu().numpy()
        sc = env._scenery[:, :lanes].cpu().numpy()
        for j, i in enumerate(due):
            assert (int(d["frame_lane"][k]), int(d["frame_step"][k])) == (i, t)
            assert np.array_equal(st[:, i], d["frame_state"][k]), (k, i, t)
            assert np.array_equal(sc[:71, i], d["frame_scenery"][k]), (k, i, t)  # clouds, waves, punch radius / y
            if k % 5 == 0:
                want = ro.frame(st[:, i], sprites.sprites_host, sprites.background_host, sc[:, i])
                assert np.array_equal(frames[j], want), (k, i, t)
                checked += 1
            k += 1

    render(list(range(lanes)), -1)
    for t in range(steps):
        env.step(env.random_actions(meta["action_seed"], t))
        due = [i for i in range(lanes) if (t + 1) % periods[i] == 0]
        if due:
            assert np.array_equal(env.state.cpu().numpy(), d["states"][t]), t
            render(due, t)
    assert k == len(d["frame_lane"]) and checked > 300
    assert np.array_equal(env.state.cpu().numpy()[:43], d["states"][steps - 1][:43])


def test_scenery_api_behaviour():
    from pikazoo_amd import pikazoo_v0
    from pikazoo_amd.render import synthetic_sprites

    sprites = synthetic_sprites(2, "cuda:0")
    with pytest.raises(ValueError):
        pikazoo_v0.env(num_envs=4, scenery=True)  # needs a render_mode
    kw = dict(num_envs=64, seed=9, render_mode="rgb_array", sprites=sprites)
    plain, scen = pikazoo_v0.env(**kw), pikazoo_v0.env(scenery=True, **kw)
    assert bool((scen.state[43] == 42).all()) and bool((plain.state[43] == 2).all())  # 40 cloud draws at construction
    for e in (plain, scen):
        e.reset()
        e.step_random(5, k=30)
    before = plain.state.clone()
    plain.render()
    assert torch.equal(plain.state, before)                    # without scenery render() has no side effects
    before = scen.state.clone()
    f1 = scen.render(lanes=[3, 5])
    after = scen.state
    changed = (after != before).any(dim=0).nonzero().flatten().tolist()
    assert changed == [3, 5] and bool((after[43, [3, 5]] >= before[43, [3, 5]] + 27).all())  # 27 wave draws + respawns
    assert torch.equal(after[:43], before[:43])
    with pytest.raises(ValueError):
        scen.render(lanes=[1, 1])
    # checkpoints carry the clouds and waves
    sd = scen.state_dict()
    f2 = scen.render(lanes=[3, 5])
    other = pikazoo_v0.env(scenery=True, **kw)
    other.load_state_dict(sd)
    assert torch.equal(other.render(lanes=[3, 5]), f2) and not torch.equal(f1, f2)
    with pytest.raises(ValueError):
        plain.load_state_dict(sd)
    # a (masked) reset clears the punch effect and what the tracker remembers of the reset games only
    odd = pikazoo_v0.env(num_envs=70, seed=3, render_mode="rgb_array", sprites=sprites, scenery=True, winning_score=1,
                         auto_reset=False)
    odd.reset()
    for t in range(400):
        odd.step(odd.random_actions(4, t))
    sc = odd._scenery[:, :70].clone()
    assert int((sc[69] > 0).sum()) > 0 and int(sc[73].sum()) > 0          # live punch effects, finished games
    mask = torch.arange(70, device="cuda:0") % 2 == 0
    odd.reset(mask=mask)
    now = odd._scenery[:, :70]
    assert bool((now[[69, 71, 72, 73, 74]][:, mask] == 0).all())
    assert torch.equal(now[:, ~mask], sc[:, ~mask]) and torch.equal(now[:69], sc[:69])
    odd.reset()
    assert bool((odd._scenery[[69, 71, 72, 73, 74], :70] == 0).all())

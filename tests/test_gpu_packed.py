"""GPU tests of the packed state format (``cfg->packed_state`` / ``state_format="packed"``, 36 bytes per game):
what is specific to the format.  The trajectory and oracle parity tests of test_gpu_parity.py run in both formats
(their ``fmt`` parameter); here: pack / unpack round trips and misfit detection through the C ABI, the k-frame
launches, masked resets and frozen games, both sides of the kernel switch, the headline size on every lane, and
checkpoints moving between the two formats.  Bar: bit-exact, as everywhere.
"""
import numpy as np
import pytest
import torch

from test_gpu_parity import cpu, make_env

pytestmark = pytest.mark.gpu


def _lib():
    from pikazoo_amd import _native

    return _native.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pack(state, n, stride, pstride):
    lib = _lib()
    packed = torch.zeros(lib.pz_packed_state_bytes(pstride), dtype=torch.uint8, device=state.device)
    misfits = torch.zeros(1, dtype=torch.int64, device=state.device)
    assert lib.pz_pack_state(state.data_ptr(), n, stride, packed.data_ptr(), pstride, misfits.data_ptr(), _stream()) == 0
    return packed, int(misfits.item())


def _unpack(packed, n, pstride, stride):
    lib = _lib()
    out = torch.full((44, stride), -77, dtype=torch.int32, device=packed.device)
    flagged = torch.zeros(1, dtype=torch.int64, device=packed.device)
    assert lib.pz_unpack_state(packed.data_ptr(), n, pstride, out.data_ptr(), stride, flagged.data_ptr(), _stream()) == 0
    return out, int(flagged.item())


# field ranges of the packed format (include/pikazoo_hip.h), per state row: (low, high) inclusive
_PLAYER_RANGES = [(0, 511), (0, 255), (-32, 31), (0, 7), (0, 7), None, (0, 7), (-1, 1), (-2, 5), (0, 1), (0, 255), (0, 1),
                  (0, 1)]
_BALL_RANGES = [(0, 511), (-512, 511), (-32, 31), (-4096, 4095), (0, 1), (0, 511), (-512, 511), (0, 511), (-512, 511),
                (0, 63), (0, 65535), (0, 511)]
_ENV_RANGES = [(0, 65535), (0, 65535), (0, 1), (0, 1), (0, 1), None]


def _random_state(n, gen):
    rows = []
    for rng in _PLAYER_RANGES + _PLAYER_RANGES + _BALL_RANGES + _ENV_RANGES:
        if rng is None:
            rows.append(None)
        else:
            rows.append(torch.randint(rng[0], rng[1] + 1, (n,), generator=gen, dtype=torch.int64).to(torch.int32))
    for c in (5, 18):  # arm_swing_direction is +-1
        rows[c] = (torch.randint(0, 2, (n,), generator=gen, dtype=torch.int64) * 2 - 1).to(torch.int32)
    rows[43] = torch.randint(-2 ** 31, 2 ** 31, (n,), generator=gen, dtype=torch.int64).to(torch.int32)  # rng counter
    return torch.stack(rows)


def test_pack_unpack_round_trip_over_the_whole_field_ranges():
    """Every field at random over its whole packed range (far beyond what play produces), odd sizes, strides larger
    than n on both sides, and the extremes of every field on a few lanes."""
    gen = torch.Generator().manual_seed(5)
    n, stride, pstride = 10_000 + 37, 10_112, 10_240
    st = _random_state(n, gen)
    ranges = _PLAYER_RANGES + _PLAYER_RANGES + _BALL_RANGES + _ENV_RANGES
    for c, rng in enumerate(ranges):  # lanes 0 / 1: all minima / all maxima
        if rng is not None:
            st[c, 0], st[c, 1] = rng[0], rng[1]
    st[5, 0] = st[18, 0] = -1
    st[5, 1] = st[18, 1] = 1
    st[43, 0], st[43, 1] = -2 ** 31, 2 ** 31 - 1
    dev = torch.device("cuda:0")
    state = torch.zeros((44, stride), dtype=torch.int32, device=dev)
    state[:, :n] = st.to(dev)
    packed, misfits = _pack(state, n, stride, pstride)
    assert misfits == 0
    back, flagged = _unpack(packed, n, pstride, stride)
    assert flagged == 0
    assert torch.equal(back[:, :n], state[:, :n])
    assert bool((back[:, n:] == -77).all())  # lanes past n are not written


def test_pack_counts_states_that_do_not_fit_and_the_flag_survives():
    dev = torch.device("cuda:0")
    env = make_env(num_envs=256, seed=1)
    env.reset()
    env.step_random(3, k=40)
    state = env.unwrapped.state.contiguous().clone()
    packed, misfits = _pack(state, 256, 256, 256)
    assert misfits == 0
    bad = state.clone()
    bad[29, 7] = 5000      # ball y velocity beyond the 13-bit field
    bad[0, 9] = -3         # player 1 x negative
    bad[27, 200] = -600    # ball y below the 10-bit signed field
    bad[34, 201] = 700     # previous_previous_y above it
    bad[38, 100] = 70000   # score beyond 16 bits
    bad[5, 11] = 0         # arm_swing_direction is +-1
    packed, misfits = _pack(bad, 256, 256, 256)
    assert misfits == 6
    back, flagged = _unpack(packed, 256, 256, 256)
    assert flagged == 6
    good = torch.ones(256, dtype=torch.bool, device=dev)
    good[[7, 9, 100, 11, 200, 201]] = False
    assert torch.equal(back[:, good], state[:, good])
    # the Python host refuses both directions
    penv = make_env(num_envs=256, seed=1, state_format="packed")
    with pytest.raises(ValueError, match="outside the packed format"):
        penv.unwrapped.set_state(bad)
    assert penv.unwrapped.packed_misfits == 0
    penv.unwrapped._state_buf.copy_(packed)
    assert penv.unwrapped.packed_misfits == 6  # the poll a loop that never unpacks the state should make now and then
    with pytest.raises(RuntimeError, match="misfit flag"):
        penv.unwrapped.state
    lib = _lib()
    assert lib.pz_packed_state_bytes(1000) == 36000
    assert lib.pz_pack_state(state.data_ptr(), 256, 256, packed.data_ptr() + 4, 256, None, _stream()) == -4  # PZ_E_ALIGN
    assert lib.pz_pack_state(state.data_ptr(), 256, 128, packed.data_ptr(), 256, None, _stream()) == -2      # PZ_E_SIZE
    assert lib.pz_unpack_state(None, 256, 256, state.data_ptr(), 256, None, _stream()) == -1                 # PZ_E_NULL


def test_packed_config_limits():
    from pikazoo_amd import _native

    with pytest.raises(ValueError, match="32767"):
        make_env(num_envs=4, winning_score=40000, state_format="packed")
    env = make_env(num_envs=64, state_format="packed")
    raw = env.unwrapped
    raw._cfg.winning_score = 70000  # the C ABI's own limit: 16-bit score fields
    with pytest.raises(_native.PikazooNativeError):
        env.reset()
    raw._cfg.winning_score = 15
    raw._cfg.packed_state = 2
    with pytest.raises(_native.PikazooNativeError):
        env.reset()


@pytest.mark.parametrize("kw", [dict(), dict(is_player2_computer=True),
                                dict(is_player1_computer=True, is_player2_computer=True, flight_tables=False)])
def test_packed_equals_int32_in_every_launch_mode(kw, oracle):
    """pz_step, pz_step_random (k = 1 and k = 23), pz_rollout_random and pz_step_many in both formats from the same
    start: identical states, outputs and trajectories; the end state against the oracle."""
    n, aseed = 4096 + 64, 77
    base = dict(num_envs=n, seed=21, env_id_base=1 << 20, winning_score=2, serve="random")
    a = make_env(state_format="int32", **base, **kw)
    b = make_env(state_format="packed", **base, **kw)
    ra, rb = a.unwrapped, b.unwrapped
    assert torch.equal(ra.state, rb.state)
    oa, ob = a.reset()[0], b.reset()[0]
    assert torch.equal(oa["player_1"], ob["player_1"]) and torch.equal(ra.state, rb.state)
    t = 0
    for _ in range(60):
        acts = ra.random_actions(aseed, t)
        xa, xb = a.step(acts), b.step(acts)
        t += 1
    for e in (xa, xb):
        assert e[0]["player_1"].dtype == torch.int32
    for i in range(3):
        for ag in ("player_1", "player_2"):
            assert torch.equal(xa[i][ag], xb[i][ag])
    assert torch.equal(ra.state, rb.state)
    assert np.array_equal(cpu(xa[4]["player_1"]["score"]), cpu(xb[4]["player_1"]["score"]))
    for k in (1, 23):
        xa, xb = ra.step_random(aseed, t0=t, k=k), rb.step_random(aseed, t0=t, k=k)
        t += k
        assert torch.equal(ra.state, rb.state), k
        assert torch.equal(xa[0]["player_2"], xb[0]["player_2"]) and torch.equal(xa[1]["player_1"], xb[1]["player_1"])
    ta, tb = ra.rollout_random(aseed, 31, t0=t), rb.rollout_random(aseed, 31, t0=t)
    t += 31
    assert torch.equal(ta["actions"], tb["actions"]) and torch.equal(ta["terminations"], tb["terminations"])
    for ag in ("player_1", "player_2"):
        assert torch.equal(ta["obs"][ag], tb["obs"][ag]) and torch.equal(ta["rewards"][ag], tb["rewards"][ag])
    assert torch.equal(ra.state, rb.state)
    tape = torch.stack([torch.stack(list(ra.random_actions(aseed, t + j).values())) for j in range(19)])
    ta, tb = ra.step_many(tape), rb.step_many(tape)
    t += 19
    assert torch.equal(ta["obs"]["player_1"], tb["obs"]["player_1"]) and torch.equal(ra.state, rb.state)
    assert ra.episodes_done == rb.episodes_done
    okw = {k: v for k, v in kw.items() if k != "flight_tables"}
    ref = oracle.OracleEnv(n, oracle.make_config(winning_score=2, serve="random", seed=21, env_id_base=1 << 20, **okw),
                           nthreads=8)
    ref.reset()
    ref.rollout_random(aseed, 0, t)
    assert np.array_equal(cpu(rb.state), ref.state)


def test_packed_masked_reset_frozen_games_and_statistics(oracle):
    n = 320
    wr = dict(stack=[["RecordEpisodeStatistics", {}]])
    env = make_env(num_envs=n, seed=3, winning_score=1, auto_reset=False, state_format="packed", wrappers=wr,
                   is_player2_computer=True)
    raw = env.unwrapped
    ref = oracle.OracleEnv(n, oracle.make_config(winning_score=1, seed=3, auto_reset=False, is_player2_computer=True,
                                                 episode_stats=1))
    env.reset(), ref.reset()
    for t in range(4000):
        acts = raw.random_actions(8, t)
        obs, rew, term, _, infos = env.step(acts)
        ref.step(cpu(acts["player_1"]), cpu(acts["player_2"]))
        if t % 100 == 99 and bool(term["player_1"].all()):
            break
    assert bool(term["player_1"].all()) and np.array_equal(cpu(raw.state), ref.state)
    assert np.array_equal(cpu(raw.episode_returns), ref.episode_returns)
    assert np.array_equal(cpu(raw.episode_lengths), ref.episode_lengths)
    frozen = raw.state
    obs, rew, term, _, _ = env.step(acts)
    assert torch.equal(raw.state, frozen) and bool((rew["player_1"] == 0).all()) and bool(term["player_1"].all())
    mask = (torch.arange(n, device=raw.device) % 3 == 0)
    obs, _ = env.reset(mask=mask)
    r1, r2 = ref.reset(cpu(mask).astype(np.uint8))
    assert np.array_equal(cpu(raw.state), ref.state)
    assert np.array_equal(cpu(obs["player_1"]), r1) and np.array_equal(cpu(obs["player_2"]), r2)
    fresh = raw.observe()
    assert torch.equal(fresh["player_1"], obs["player_1"]) and torch.equal(fresh["player_2"], obs["player_2"])
    assert np.array_equal(cpu(raw.scores), ref.state[38:40].T) and raw.scores.dtype == torch.int16


@pytest.mark.parametrize("kw,steps", [(dict(), 60), (dict(is_player2_computer=True), 40)])
def test_packed_both_sides_of_the_kernel_switch(kw, steps, oracle):
    """The packed state is stepped by the pair kernel at every size (the int32 columns switch to one wave per 64 games
    at 393 216): both sides of that size."""
    switch = 393216
    for n in (switch - 64, switch):
        env = make_env(num_envs=n, seed=44, env_id_base=7, state_format="packed", **kw)
        env.reset()
        for t in range(steps):
            obs, rew, term, _, _ = env.step(env.unwrapped.random_actions(21, t))
        state = env.unwrapped.state
        for lo in (0, n - 1024):
            ref = oracle.OracleEnv(1024, oracle.make_config(seed=44, env_id_base=7 + lo, **kw), nthreads=4)
            ref.reset()
            for t in range(steps):
                a1, a2 = oracle.random_actions(1024, 7 + lo, 21, t)
                robs, rrew, rterm = ref.step(a1, a2)
            assert np.array_equal(cpu(state[:, lo:lo + 1024]), ref.state), (n, lo)
            assert np.array_equal(cpu(obs["player_1"][lo:lo + 1024]), robs[0]), (n, lo)
            assert np.array_equal(cpu(rew["player_2"][lo:lo + 1024]), rrew[1]), (n, lo)
            assert np.array_equal(cpu(term["player_1"][lo:lo + 1024]).astype(np.uint8), rterm), (n, lo)


@pytest.mark.parametrize("name,kw,wr", [
    ("random_random", dict(), {}),
    ("cfg3_p2_computer_tables", dict(is_player2_computer=True), {}),
    ("cfg5_fused_wrappers", dict(), dict(simplify_action=True,
                                         additional_reward=(0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01))),
])
def test_packed_headline_size_every_lane_vs_oracle(name, kw, wr, oracle):
    """65 536 games x 320 frames through pz_step on the packed state, all lanes against the oracle."""
    from oracle.ref_capture import fused_options

    n, steps, seed, base, aseed = 65536, 320, 11, 1 << 33, 99
    env = make_env(num_envs=n, seed=seed, env_id_base=base, winning_score=2, wrappers=wr, state_format="packed", **kw)
    raw = env.unwrapped
    ref = oracle.OracleEnv(n, oracle.make_config(winning_score=2, seed=seed, env_id_base=base, **kw,
                                                 **fused_options(wr)), nthreads=16)
    env.reset(), ref.reset()
    for t in range(steps):
        acts = raw.random_actions(aseed, t)
        obs, rew, term, _, _ = env.step(acts)
        if (t + 1) % 40 == 0:
            ref.rollout_random(aseed, t + 1 - 40, 40)
            hs = cpu(raw.state)
            if not np.array_equal(hs, ref.state):
                f, l = np.argwhere(hs != ref.state)[0]
                pytest.fail(f"{name}: frame {t} lane {l} word {oracle.FIELD_NAMES[f]}: hip {hs[f, l]} != "
                            f"oracle {ref.state[f, l]}")
    assert np.array_equal(cpu(obs["player_1"]), ref.obs[0]) and np.array_equal(cpu(obs["player_2"]), ref.obs[1])
    assert np.array_equal(cpu(rew["player_1"]), ref.rew[0]) and np.array_equal(cpu(rew["player_2"]), ref.rew[1])
    assert np.array_equal(cpu(term["player_1"]).astype(np.uint8), ref.term)


def test_checkpoints_move_between_the_formats():
    kw = dict(num_envs=1000, seed=8, is_player2_computer=True, winning_score=3)
    a = make_env(state_format="int32", **kw)
    a.reset()
    a.step_random(1, k=100)
    sd = a.unwrapped.state_dict()
    a.step_random(1, k=50)
    b = make_env(state_format="packed", **kw)
    b.unwrapped.load_state_dict(sd)
    b.step_random(1, k=50)
    assert torch.equal(b.unwrapped.state, a.unwrapped.state)
    sd2 = b.unwrapped.state_dict()
    assert sd2["state"].dtype == torch.int32 and tuple(sd2["state"].shape) == (44, 1000)
    c = make_env(state_format="int32", **kw)
    c.unwrapped.load_state_dict(sd2)
    b.step_random(1, k=30)
    c.step_random(1, k=30)
    assert torch.equal(b.unwrapped.state, c.unwrapped.state)


def test_packed_render_and_scalar_api():
    from pikazoo_amd import render as R

    sprites = R.synthetic_sprites(3, torch.device("cuda:0"))
    kw = dict(num_envs=8, seed=5, render_mode="rgb_array", sprites=sprites)
    a, b = make_env(state_format="int32", **kw), make_env(state_format="packed", **kw)
    for e in (a, b):
        e.reset()
        e.step_random(9, k=77)
    assert torch.equal(a.render(), b.render())
    env = make_env(num_envs=1, scalar_api=True, winning_score=1, state_format="packed", validate_actions=True)
    obs, infos = env.reset()
    assert isinstance(obs["player_1"], np.ndarray) and infos["player_1"]["score"] == [0, 0]
    steps = 0
    while env.agents:
        obs, rew, term, trunc, infos = env.step({a: env.action_space(a).sample() for a in env.agents})
        steps += 1
    assert term["player_1"] is True and sorted(infos["player_1"]["score"]) == [0, 1] and steps > 10


@pytest.mark.parametrize("fmt", ["int32", "packed"])
def test_planted_fast_balls_with_normalized_observations(fmt):
    """Negative ball y and out-of-bounds velocities through the fused NormalizeObservation: float32 (v - low) / range,
    equal to the oracle's (whose arithmetic tests/test_oracle_golden.py ties to the reference's float64 quotient)."""
    _replay_planted_through_hip("planted_fast_balls_both_computer", fmt, True, normalize=True)


@pytest.mark.parametrize("name,fmt,tables", [("planted_fast_balls_human", "int32", True),
                                             ("planted_fast_balls_human", "packed", True),
                                             ("planted_fast_balls_both_computer", "int32", True),
                                             ("planted_fast_balls_both_computer", "int32", False),
                                             ("planted_fast_balls_both_computer", "packed", True),
                                             ("planted_fast_balls_both_computer", "packed", False)])
def test_planted_fast_balls_follow_the_reference(name, fmt, tables):
    """The reference stepped from planted ball states random play never reaches (oracle/ref_capture.capture_planted):
    among them balls bounced off the net top to a negative y -- the ball's y and its trail are signed quantities, in
    the int32 columns and in the packed format's 10-bit fields alike."""
    _replay_planted_through_hip(name, fmt, tables)


@pytest.mark.parametrize("fmt,tables", [("int32", True), ("int32", False), ("packed", True), ("packed", False),
                                        ("int32", "power_hit"), ("packed", "power_hit")])
@pytest.mark.parametrize("name", ["planted_random_states_human", "planted_random_states_both_computer",
                                  "planted_random_states_p2_computer_random_serve"])
def test_random_planted_states_follow_the_reference(name, fmt, tables):
    """Every attribute of both players, the ball and the scores at random over its whole valid range, planted into the
    reference and stepped by it (oracle/ref_capture.capture_planted_random): the HIP path in both state formats, with
    and without the flight tables, reproduces every word of every frame."""
    if not tables and "human" in name:
        pytest.skip("no computer player: the tables are never consulted")
    _replay_planted_through_hip(name, fmt, tables)


def _replay_planted_through_hip(name, fmt, tables, normalize=False):
    """A planted-state fixture through the product: set_state(planted), then every frame's state against the reference's
    and every frame's observations (raw, or NormalizeObservation's float32) against _get_obs of that state."""
    from conftest import load_golden
    from oracle import pz_oracle as po
    from test_oracle_golden import replay_planted

    d = load_golden(name)

    def make(meta, planted):
        env = make_env(num_envs=meta["lanes"], seed=meta["seed"], env_id_base=meta["env_id_base"], state_format=fmt,
                       flight_tables=tables, wrappers=dict(stack=[["NormalizeObservation", {}]]) if normalize else {},
                       **meta["env_kwargs"])
        raw = env.unwrapped
        raw.set_state(torch.as_tensor(planted, device=raw.device))

        checker = po.OracleEnv(meta["lanes"], po.make_config(normalize_obs=normalize))

        def step(a1, a2):
            obs = env.step({"player_1": torch.as_tensor(a1, device=raw.device),
                            "player_2": torch.as_tensor(a2, device=raw.device)})[0]
            state = cpu(raw.state)
            checker.state[:] = state            # the observation is a function of the state: _get_obs of what we hold
            o1, o2 = checker.observe()
            assert np.array_equal(cpu(obs["player_1"]), o1) and np.array_equal(cpu(obs["player_2"]), o2)
            return state
        return step

    replay_planted(d, make)


def test_unreachable_player_state_raises_the_misfit_flag():
    """pz_pack_state accepts any state whose fields fit one by one; an UNREACHABLE combination (a player at the jump's
    apex still moving up at full speed) then leaves the court's y range a few frames later -- the step kernel raises
    the game's sticky misfit flag instead of corrupting the neighbouring fields silently, and the host refuses the state.
    The int32 columns step such a state like the reference does (see the random planted fixtures)."""
    from pikazoo_amd import _native

    env = make_env(num_envs=128, seed=2, state_format="packed", auto_reset=False)
    env.reset()
    st = env.unwrapped.state.clone()
    st[1, 5], st[2, 5], st[3, 5] = 110, -16, 1       # player 1 of game 5: y 110, y velocity -16, jumping
    st[14, 9], st[15, 9], st[16, 9] = 112, -16, 1    # player 2 of game 9
    env.unwrapped.set_state(st)                      # every field fits: accepted
    noop = torch.zeros(128, dtype=torch.int32, device="cuda:0")
    assert env.unwrapped.packed_misfits == 0
    for _ in range(12):
        env.step({"player_1": noop, "player_2": noop})
    assert env.unwrapped.packed_misfits == 2  # visible without unpacking anything (pz_count_packed_misfits)
    with pytest.raises(_native.PikazooNativeError, match="2 games carry"):
        env.unwrapped.state
    assert make_env(num_envs=8).unwrapped.packed_misfits == 0  # (int32 columns: nothing to flag)


@pytest.mark.parametrize("fmt", ["int32", "packed"])
@pytest.mark.parametrize("name", ["planted_random_states_both_computer", "planted_random_states_p2_computer_random_serve",
                                  "planted_fast_balls_both_computer"])
def test_k_frame_launches_from_planted_states(name, fmt):
    """The k-frame kernels (state in registers for the whole launch: pz_step_many on the fixture's own action tape,
    pz_step_random / pz_rollout_random against the oracle) started from the planted states: the reference's last frame,
    and the oracle's state after the random-policy frames."""
    from conftest import load_golden
    from oracle import pz_oracle as po

    d = load_golden(name)
    meta = d["meta"]
    lanes, frames = meta["lanes"], meta["frames"]
    lanes4 = lanes // 4 * 4                       # the trajectory kernels want a multiple of four games
    kw = dict(num_envs=lanes4, seed=meta["seed"], env_id_base=meta["env_id_base"], state_format=fmt, **meta["env_kwargs"])
    planted = torch.as_tensor(d["planted"][:, :lanes4].copy(), device="cuda:0")
    tape = np.stack([np.stack(po.random_actions(lanes, meta["env_id_base"], meta["action_seed"], meta["warm"] + t, 18))
                     for t in range(frames)])[:, :, :lanes4]
    env = make_env(**kw)
    env.unwrapped.set_state(planted)
    out = env.unwrapped.step_many(torch.as_tensor(tape, device="cuda:0"))
    assert np.array_equal(cpu(env.unwrapped.state), d["states"][frames - 1][:, :lanes4])
    okw = meta["env_kwargs"]
    cfg = po.make_config(winning_score=okw.get("winning_score", 15), serve=okw.get("serve", "winner"),
                         is_player1_computer=okw.get("is_player1_computer", False),
                         is_player2_computer=okw.get("is_player2_computer", False), seed=meta["seed"],
                         env_id_base=meta["env_id_base"])
    for mode in ("random", "rollout"):
        env = make_env(**kw)
        env.unwrapped.set_state(planted)
        ref = po.OracleEnv(lanes4, cfg, nthreads=4)
        ref.state[:] = d["planted"][:, :lanes4]
        if mode == "random":
            env.unwrapped.step_random(3, t0=0, k=17)
        else:
            traj = env.unwrapped.rollout_random(3, 17, t0=0)
        ref.rollout_random(3, 0, 17)
        assert np.array_equal(cpu(env.unwrapped.state), ref.state), mode
    assert np.array_equal(cpu(traj["obs"]["player_1"][-1]), ref.obs[0])


# ------------------------------------------------------------------------------------------------
# int16 observations (cfg.normalize_obs == 2 / observation_dtype=torch.int16)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt", ["int32", "packed"])
@pytest.mark.parametrize("n,kw", [(1, dict()), (63, dict(is_player2_computer=True)), (64, dict()),
                                  (1001, dict(is_player1_computer=True, is_player2_computer=True, flight_tables=False)),
                                  (4096 + 8, dict(is_player2_computer=True, winning_score=2))])
def test_int16_observations_hold_the_int32_values(n, kw, fmt, oracle):
    """The same run with int32 and with int16 observations: identical states / rewards / terminations, and every
    observation equal value for value -- after reset, after pz_step (pair kernel), pz_step_random (single wave),
    observe(), and against the oracle at the end.  Odd sizes included: an int16 tensor holds an even number of rows."""
    base = dict(num_envs=n, seed=6, env_id_base=11, state_format=fmt, **kw)
    a = make_env(**base)
    b = make_env(observation_dtype=torch.int16, **base)
    assert b.observation_space("player_1").dtype == np.int16
    oa, ob = a.reset()[0], b.reset()[0]
    for ag in ("player_1", "player_2"):
        assert ob[ag].dtype == torch.int16 and ob[ag].shape == (n, 35) and torch.equal(oa[ag], ob[ag].to(torch.int32))
    for t in range(80):
        acts = a.random_actions(5, t)
        xa, xb = (a.step(acts), b.step(acts)) if t % 3 else (a.step_random(5, t0=t), b.step_random(5, t0=t))
        if t % 10 == 0 or t == 79:
            for ag in ("player_1", "player_2"):
                assert torch.equal(xa[0][ag], xb[0][ag].to(torch.int32)), (t, ag)
                assert torch.equal(xa[1][ag], xb[1][ag])
            assert torch.equal(xa[2]["player_1"], xb[2]["player_1"]) and torch.equal(a.state, b.state)
    fa, fb = a.observe(), b.observe()
    assert fb["player_2"].dtype == torch.int16 and torch.equal(fa["player_2"], fb["player_2"].to(torch.int32))
    okw = {k: v for k, v in kw.items() if k != "flight_tables"}
    ref = oracle.OracleEnv(n, oracle.make_config(seed=6, env_id_base=11, **okw), nthreads=2)
    ref.reset()
    ref.rollout_random(5, 0, 80)
    assert np.array_equal(cpu(b.state), ref.state) and np.array_equal(cpu(xb[0]["player_1"]).astype(np.int32), ref.obs[0])


@pytest.mark.parametrize("fmt", ["int32", "packed"])
def test_int16_observations_in_the_trajectory_launches(fmt):
    n, k = 4096, 19
    base = dict(num_envs=n, seed=9, is_player2_computer=True, winning_score=3, state_format=fmt)
    a, b = make_env(**base), make_env(observation_dtype=torch.int16, **base)
    a.reset(), b.reset()
    ta, tb = a.rollout_random(4, k), b.rollout_random(4, k)
    assert tb["obs"]["player_1"].dtype == torch.int16 and tb["obs"]["player_1"].shape == (k, n, 35)
    for ag in ("player_1", "player_2"):
        assert torch.equal(ta["obs"][ag], tb["obs"][ag].to(torch.int32)) and torch.equal(ta["rewards"][ag], tb["rewards"][ag])
    tape = ta["actions"].clone()
    ta, tb = a.step_many(tape), b.step_many(tape)
    assert torch.equal(ta["obs"]["player_2"], tb["obs"]["player_2"].to(torch.int32)) and torch.equal(a.state, b.state)
    # the single-frame views follow the last frame
    assert torch.equal(b.observe()["player_1"], tb["obs"]["player_1"][-1])
    odd = make_env(num_envs=4100, observation_dtype=torch.int16, state_format=fmt)   # 4100 % 8 != 0
    odd.reset()
    with pytest.raises(ValueError, match="multiple of 8"):
        odd.rollout_random(1, 4)
    from pikazoo_amd.wrappers import NormalizeObservation
    norm = NormalizeObservation(odd)  # int16 rows: the quotient is taken on the step's outputs, outside the kernel
    assert norm.fused is False and norm.reset()[0]["player_1"].dtype == torch.float32


def _random_valid_states(n, rng):
    """int32[44, n]: every attribute at random over its valid range (players' (y, y_velocity) from the pairs a jump or
    a dive passes through), half of the balls next to a player -- the whole state space, reachable or not."""
    st = np.zeros((44, n), np.int32)
    for base, lo, hi in ((0, 32, 184), (13, 248, 400)):
        st[base + 0] = rng.integers(lo, hi + 1, n)
        st[base + 3] = rng.integers(0, 5, n)
        steps = rng.integers(0, 33, n)
        v0 = np.where(rng.random(n) < 0.7, -16, -5)
        y, v = np.full(n, 244), v0.copy()
        for k in range(33):
            move = (k < steps) & (y + v <= 244)
            y, v = np.where(move, y + v, y), np.where(move, v + 1, v)
        ground = np.isin(st[base + 3], (0, 4)) & (rng.random(n) < 0.7)
        st[base + 1], st[base + 2] = np.where(ground, 244, y), np.where(ground, 0, v)
        st[base + 4] = rng.integers(0, 5, n)
        st[base + 5] = rng.choice([-1, 1], n)
        st[base + 6] = rng.integers(0, 6, n)
        st[base + 7] = rng.integers(-1, 2, n)
        st[base + 8] = rng.integers(-1, 4, n)
        st[base + 9] = rng.integers(0, 2, n)
        st[base + 10] = rng.integers(0, 5, n)
        st[base + 11] = rng.integers(0, 2, n)
        st[base + 12] = rng.integers(0, 2, n)
    near = rng.random(n) < 0.5
    who = rng.random(n) < 0.5
    px, py = np.where(who, st[0], st[13]), np.where(who, st[1], st[14])
    st[26] = np.where(near, np.clip(px + rng.integers(-40, 41, n), 20, 432), rng.integers(20, 433, n))
    st[27] = np.where(near, np.clip(py + rng.integers(-40, 41, n), 0, 252), rng.integers(0, 253, n))
    st[28] = rng.integers(-20, 21, n)
    st[29] = np.where(rng.random(n) < 0.8, rng.integers(-120, 121, n), rng.integers(-300, 301, n))
    st[30] = rng.integers(0, 2, n)
    st[31], st[32] = rng.integers(20, 433, n), rng.integers(-100, 253, n)
    st[33], st[34] = rng.integers(20, 433, n), rng.integers(-100, 253, n)
    st[35] = rng.integers(0, 51, n)
    st[36] = rng.integers(20, 433, n)
    st[37] = rng.integers(20, 433, n)
    st[38], st[39] = rng.integers(0, 3, n), rng.integers(0, 3, n)
    st[40] = rng.integers(0, 2, n)
    st[43] = rng.integers(4, 1 << 20, n)
    return st


@pytest.mark.parametrize("fmt", ["int32", "packed"])
@pytest.mark.parametrize("kw", [dict(), dict(is_player1_computer=True, is_player2_computer=True),
                                dict(is_player2_computer=True, serve="random", flight_tables=False)])
def test_whole_state_space_at_scale_vs_oracle(kw, fmt, oracle):
    """262 144 random valid states (the fixtures hold 2 400 stepped by the reference itself; this is the same
    generator at scale against the oracle, which the live tests tie to the reference on such states): 12 frames of
    pz_step, every word of every game, both state formats."""
    n, frames = 262144, 12
    rng = np.random.default_rng(20241011)
    planted = _random_valid_states(n, rng)
    env = make_env(num_envs=n, seed=15, env_id_base=1 << 21, winning_score=3, state_format=fmt, **kw)
    raw = env.unwrapped
    raw.set_state(torch.as_tensor(planted, device=raw.device))
    okw = {k: v for k, v in kw.items() if k != "flight_tables"}
    ref = oracle.OracleEnv(n, oracle.make_config(winning_score=3, seed=15, env_id_base=1 << 21, **okw), nthreads=16)
    ref.state[:] = planted
    for t in range(frames):
        acts = raw.random_actions(8, t)
        obs = env.step(acts)[0]
        ref.step(cpu(acts["player_1"]), cpu(acts["player_2"]))
        if t % 4 == 3:
            hs = cpu(raw.state)
            if not np.array_equal(hs, ref.state):
                f, l = np.argwhere(hs != ref.state)[0]
                pytest.fail(f"frame {t} lane {l} word {oracle.FIELD_NAMES[f]}: hip {hs[f, l]} != oracle {ref.state[f, l]}; "
                            f"planted {planted[:, l].tolist()}")
    assert np.array_equal(cpu(obs["player_1"]), ref.obs[0]) and np.array_equal(cpu(obs["player_2"]), ref.obs[1])

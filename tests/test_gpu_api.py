"""GPU tests of the drop-in API surface: the reference's own two tests restated against the HIP
path (tests/test_parallel_api.py:5-7 -> API conformance; tests/env/test_env.py:7-21 -> observation
symmetry), plus the wrapper classes and the error behaviour of the reference's step()."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_env_observation_symmetry_scalar_api():
    """tests/env/test_env.py:7-21 verbatim in structure: both computer, action 0, until the game ends."""
    from pikazoo_amd import pikazoo_v0

    env = pikazoo_v0.env(winning_score=15, is_player1_computer=True, is_player2_computer=True, render_mode=None,
                         num_envs=1, scalar_api=True, auto_reset=False, seed=11)

    def divide_and_assert(observations):
        p1a, p2a = observations["player_1"][0:13], observations["player_1"][13:26]
        p2b, p1b = observations["player_2"][0:13], observations["player_2"][13:26]
        assert np.all(p1a == p1b) and np.all(p2a == p2b)

    observations, infos = env.reset()
    divide_and_assert(observations)
    steps = 0
    while env.agents:
        actions = {agent: 0 for agent in env.agents}
        observations, rewards, terminations, truncations, infos = env.step(actions)
        divide_and_assert(observations)
        steps += 1
        assert steps < 200000
    assert terminations == {"player_1": True, "player_2": True}
    assert max(infos["player_1"]["score"]) == 15
    with pytest.raises(RuntimeError):
        env.step({"player_1": 0, "player_2": 0})
    env.reset()
    assert env.agents == ["player_1", "player_2"]


def test_parallel_api_conformance_scalar():
    """What pettingzoo.test.parallel_api_test checks (tests/test_parallel_api.py:7), restated:
    dict keys == live agents, observations inside observation_space, agent list lifecycle,
    reset after termination, reward/termination types."""
    from pikazoo_amd import pikazoo_v0

    env = pikazoo_v0.env(num_envs=1, scalar_api=True, auto_reset=False, winning_score=2, seed=5)
    rng = np.random.default_rng(0)
    assert env.possible_agents == ["player_1", "player_2"] and env.metadata["name"] == "pikazoo_v0"
    for a in env.possible_agents:
        assert env.action_space(a).n == 18 and env.observation_space(a).shape == (35,)
        assert env.observation_space(a).dtype == np.int32
        assert env.observation_space(a) is env.observation_space(a)  # lru_cached like the reference
    episodes = 0
    obs, infos = env.reset(seed=42, options={"x": 1})
    for cycle in range(6000):
        assert set(obs) == set(env.agents) == set(infos)
        for a in env.agents:
            assert obs[a].shape == (35,) and env.observation_space(a).contains(obs[a])
        actions = {a: int(rng.integers(0, 18)) for a in env.agents}
        obs, rew, term, trunc, infos = env.step(actions)
        assert set(obs) == set(rew) == set(term) == set(trunc) == set(infos) == {"player_1", "player_2"}
        assert isinstance(rew["player_1"], int) and rew["player_1"] == -rew["player_2"]
        assert isinstance(term["player_1"], bool) and trunc["player_1"] is False
        assert len(infos["player_1"]["score"]) == 2
        if term["player_1"]:
            assert env.agents == [] and max(infos["player_1"]["score"]) == 2
            episodes += 1
            obs, infos = env.reset()
            assert env.agents == env.possible_agents
    assert episodes >= 3


def test_batched_api_shapes_and_types():
    from pikazoo_amd import pikazoo_v0

    n = 1024
    env = pikazoo_v0.env(num_envs=n, device="cuda:0", seed=1)
    assert env.num_envs == n and env.agents == env.possible_agents == ["player_1", "player_2"]
    obs, infos = env.reset()
    for a in env.agents:
        assert obs[a].shape == (n, 35) and obs[a].dtype == torch.int32 and obs[a].is_cuda
        assert infos[a]["score"].shape == (n, 2)
    a = {ag: torch.randint(0, 18, (n,), device="cuda:0") for ag in env.agents}  # int64 accepted
    obs, rew, term, trunc, infos = env.step(a)
    for ag in env.agents:
        assert rew[ag].shape == (n,) and rew[ag].dtype == torch.int32
        assert term[ag].shape == (n,) and term[ag].dtype == torch.bool
        assert trunc[ag].dtype == torch.bool and not bool(trunc[ag].any())
    # infos["score"] aliases live state like the reference's list (pikazoo_env.py:573-574)
    assert infos["player_1"]["score"].data_ptr() == env.scores.data_ptr()
    # numpy / list actions work for small batches
    env.step({"player_1": np.zeros(n, np.int64), "player_2": [0] * n})


def test_step_error_behaviour():
    from pikazoo_amd import pikazoo_v0

    env = pikazoo_v0.env(num_envs=8, seed=1)  # validate_actions is the default
    env.reset()
    ok = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    # the reference's table lookup raises IndexError (pikazoo_env.py:182).  Host values are checked before the launch ...
    with pytest.raises(IndexError):
        env.step({"player_1": [18] * 8, "player_2": [0] * 8})
    with pytest.raises(IndexError):
        env.step({"player_1": np.zeros(8, np.int64), "player_2": torch.full((8,), -1)})
    # ... device tensors inside it: the launch counts, check_actions() asks
    env.step({"player_1": ok + 18, "player_2": ok})
    with pytest.raises(IndexError):
        env.check_actions()
    env.check_actions()  # (raising cleared the counter)
    env.step({"player_1": ok, "player_2": ok - 1})
    with pytest.raises(IndexError):
        env.check_actions()
    env.step({"player_1": ok + 17, "player_2": ok})
    env.check_actions()
    with pytest.raises(KeyError):
        env.step({"player_1": ok})
    with pytest.raises(ValueError):
        env.step({"player_1": ok[:4], "player_2": ok})
    with pytest.raises(TypeError):
        env.step({"player_1": ok.float(), "player_2": ok})
    with pytest.raises(AssertionError):  # pikazoo_env.py:104
        pikazoo_v0.env(serve="loser")
    with pytest.raises(NotImplementedError):
        pikazoo_v0.env(render_mode="human")
    with pytest.raises(RuntimeError):
        pikazoo_v0.env(device="cpu")


def test_strict_action_validation_raises_from_the_call_that_was_handed_the_action():
    """validate_every=1 is the strict mode: step() and step_many() read the device counter before they return (one
    synchronisation per call), so the reference's IndexError (pikazoo_env.py:182) comes from the very call that was handed
    the out-of-range device action -- also from a caller's LAST step_many, which the asynchronous default would never
    report without a check_actions()."""
    from pikazoo_amd import pikazoo_v0

    n = 256
    env = pikazoo_v0.env(num_envs=n, seed=5, validate_every=1)
    raw = env.unwrapped
    env.reset()
    ok = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    bad = ok.clone()
    bad[77] = 18
    env.step({"player_1": ok + 17, "player_2": ok})
    with pytest.raises(IndexError):
        env.step({"player_1": ok, "player_2": bad})
    env.step({"player_1": ok, "player_2": ok})  # raised once: the counter was cleared
    tape = torch.zeros((40, 2, n), dtype=torch.int32, device="cuda:0")
    raw.step_many(tape)
    tape[39, 1, 200] = -3
    with pytest.raises(IndexError):
        raw.step_many(tape)  # the one and only call with the bad tape
    raw.check_actions()
    # the default stays asynchronous: the same last call returns, and check_actions() is how its caller asks
    lazy = pikazoo_v0.env(num_envs=n, seed=5).unwrapped
    lazy.reset()
    lazy.step_many(tape)
    with pytest.raises(IndexError):
        lazy.check_actions()


@pytest.mark.parametrize("kw", [dict(), dict(state_format="packed"), dict(is_player2_computer=True),
                                dict(is_player2_computer=True, flight_tables=False), dict(num_envs=393216 + 64)])
def test_out_of_range_actions_are_counted_in_the_launch_and_raised_without_a_sync(kw):
    """validate_actions (the default): the step kernels count actions outside [0, n_actions) -- every kernel family that
    reads actions: two waves per 64 games, one wave (the large batch, the scout-wave launch), the packed format, the
    tape kernels -- and env.step() raises the reference's IndexError (pikazoo_env.py:182) within 2 * validate_every
    steps without ever blocking on the device; with SimplifyAction the bound is 13."""
    from pikazoo_amd import pikazoo_v0
    from pikazoo_amd.wrappers import SimplifyAction

    kw = dict(dict(num_envs=200, seed=2, validate_every=4), **kw)
    n = kw["num_envs"]
    env = pikazoo_v0.env(**kw)
    raw = env.unwrapped
    env.reset()
    ok = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    for _ in range(20):
        env.step({"player_1": ok + 17, "player_2": ok})
    raw.check_actions()
    assert int(raw._faults.item()) == 0
    bad = ok.clone()
    bad[[3, 70, n - 1]] = torch.tensor([18, -1, 1 << 30], dtype=torch.int32, device="cuda:0")
    env.step({"player_1": ok, "player_2": bad})
    assert int(raw._faults.item()) == 3  # one count per offending game
    with pytest.raises(IndexError):  # nobody asks: it surfaces within 2 * validate_every steps
        for _ in range(2 * raw.validate_every):
            env.step({"player_1": ok, "player_2": ok})
    for _ in range(3 * raw.validate_every):  # ... once
        env.step({"player_1": ok, "player_2": ok})
    # the tape launches check what they park
    if n % 8 == 0 or n == 200:
        tape = torch.zeros((70, 2, n), dtype=torch.int32, device="cuda:0")
        raw.step_many(tape)
        raw.check_actions()
        tape[66, 0, 5] = 18   # in the second chunk of 64 frames
        tape[2, 1, 9] = 255   # a byte that would park as an action, 255, and 256 -> 0
        tape[3, 1, 10] = 256
        raw.step_many(tape)
        with pytest.raises(IndexError):
            raw.check_actions()
    env = SimplifyAction(pikazoo_v0.env(**kw))
    env.reset()
    env.step({"player_1": ok + 12, "player_2": ok})
    env.unwrapped.check_actions()
    env.step({"player_1": ok + 13, "player_2": ok})
    with pytest.raises(IndexError):
        env.unwrapped.check_actions()
    # switched off: nothing is counted, nothing raises (the range is then the caller's contract)
    off = pikazoo_v0.env(**dict(kw, validate_actions=False))
    off.reset()
    off.step({"player_1": bad, "player_2": ok})
    off.check_actions()
    assert off._faults is None


def test_wrappers_outside_the_kernel_second_instances_and_the_scalar_api():
    """What the reference fixtures do not reach of the wrappers' torch forms (pikazoo_amd/wrappers/*.py, `fused` False):
    a second RecordEpisodeStatistics and a second SimplifyAction (its map composed with the fused one; an entry the
    reference would raise IndexError on becomes an out-of-range action), NormalizeObservation on int16 observations and
    the scalar API's Python arithmetic -- each against the numpy restatement of the reference's classes
    (oracle/wrappers_oracle.py) on the same stack."""
    from oracle.wrappers_oracle import WrappedOracle
    from pikazoo_amd import pikazoo_v0
    import pikazoo_amd.wrappers as W

    table = [0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01]
    stack = [("RecordEpisodeStatistics", {}), ("RewardByBallPosition", dict(additional_reward=table, x_line=216, y_line=176)),
             ("RewardByBallPosition", dict(additional_reward=table, x_line=100, y_line=200)), ("RecordEpisodeStatistics", {}),
             ("NormalizeObservation", {})]
    n = 96
    env = pikazoo_v0.env(num_envs=n, seed=5, winning_score=1, observation_dtype=torch.int16)
    for name, kw in stack:
        env = getattr(W, name)(env, **{k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()})
    raw = env.unwrapped
    assert raw._unfused == ["RewardByBallPosition", "RecordEpisodeStatistics", "NormalizeObservation"]
    ref = WrappedOracle(n, stack, seed=5, winning_score=1)
    obs, _ = env.reset()
    o1, _ = ref.reset()
    assert obs["player_1"].dtype == torch.float32 and np.array_equal(obs["player_1"].cpu().numpy(), o1.astype(np.float32))
    for t in range(400):
        acts = raw.random_actions(3, t)
        obs, rew, term, trunc, infos = env.step(acts)
        o, r, tm, ep = ref.step(acts["player_1"].cpu().numpy(), acts["player_2"].cpu().numpy())
        assert np.array_equal(raw.state.cpu().numpy(), ref.raw_state)
        assert np.array_equal(obs["player_2"].cpu().numpy(), o[1].astype(np.float32))
        np.testing.assert_allclose(rew["player_1"].cpu().numpy(), r[0], rtol=0, atol=1e-6)
        done = tm.astype(bool)
        assert np.array_equal(term["player_1"].cpu().numpy(), done)
        if done.any():  # the OUTER statistics (unfused): sums of the doubly wrapped reward
            assert np.array_equal(infos["player_1"]["episode"]["l"].cpu().numpy()[done], ep["l"][done])
            np.testing.assert_allclose(infos["player_2"]["episode"]["r"].cpu().numpy()[done], ep["r"][1][done], rtol=0, atol=2e-6)
    assert int(term["player_1"].sum()) >= 0 and ref.raw_state[42].sum() >= 0
    # the inner (fused, raw-reward) statistics are still the kernel's
    assert raw.episode_lengths is not None and env.episode_lengths["player_1"] is not raw.episode_lengths

    # a second SimplifyAction: map(map(a)); a in {5, 6, 9, 11, 12} maps to 13 or more first -> IndexError in the reference
    env = W.SimplifyAction(W.SimplifyAction(pikazoo_v0.env(num_envs=8, seed=1)))
    plain = W.SimplifyAction(pikazoo_v0.env(num_envs=8, seed=1))
    env.reset(), plain.reset()
    m1 = torch.tensor([0, 1, 2, 3, 4, 6, 7, 10], dtype=torch.int32, device="cuda:0")   # = map_1 of 0..7 except 5, 6
    a = torch.tensor([0, 1, 2, 3, 4, 7, 8, 10], dtype=torch.int32, device="cuda:0")
    inner = torch.tensor([0, 1, 2, 3, 4, 10, 11, 13], dtype=torch.int32, device="cuda:0")  # map_1[a]: all < 13 but the last
    a, inner = a[:7], inner[:7]
    pad = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    env.step({"player_1": torch.cat([a, pad]), "player_2": torch.cat([a * 0, pad])})
    plain.step({"player_1": torch.cat([inner, pad]), "player_2": torch.cat([inner * 0, pad])})
    assert torch.equal(env.unwrapped.state, plain.unwrapped.state)
    env.unwrapped.check_actions()
    env.step({"player_1": torch.full((8,), 12, dtype=torch.int32, device="cuda:0"), "player_2": torch.cat([a * 0, pad])})
    with pytest.raises(IndexError):  # map_1[12] = 16: not an action of the inner SimplifyAction
        env.unwrapped.check_actions()
    with pytest.raises(IndexError):  # host values: on the step itself
        env.step({"player_1": [12] * 8, "player_2": [0] * 8})
    with pytest.raises(IndexError):  # ... and a host index past the tuple raises like the reference's `action_map[agent][13]`
        env.step({"player_1": [13] * 8, "player_2": [0] * 8})
    with pytest.warns(RuntimeWarning, match="validate_actions=False"):  # nobody would report a bad device action there
        W.SimplifyAction(W.SimplifyAction(pikazoo_v0.env(num_envs=8, seed=1, validate_actions=False)))
    # ONE meaning of a negative action for the fused and the un-fused form, for host and device values: out of range, like
    # the env's own 18 actions (the reference's tuple indexing would wrap -13 .. -1 around: a Python accident)
    for build in (lambda: W.SimplifyAction(pikazoo_v0.env(num_envs=8, seed=3, validate_every=1)),                      # fused
                  lambda: W.SimplifyAction(W.SimplifyAction(pikazoo_v0.env(num_envs=8, seed=3, validate_every=1)))):  # + un-fused
        e = build()
        e.reset()
        with pytest.raises(IndexError):
            e.step({"player_1": [-1] * 8, "player_2": [0] * 8})
        e = build()
        e.reset()
        with pytest.raises(IndexError):  # (validate_every=1: from the call that was handed the action)
            e.step({"player_1": torch.full((8,), -1, dtype=torch.int32, device="cuda:0"),
                    "player_2": torch.zeros(8, dtype=torch.int32, device="cuda:0")})

    # frozen games and the scalar API: one env, Python numbers, the reference's own loop shape
    stack = [("RewardByBallPosition", dict(additional_reward=table, x_line=216, y_line=176)), ("RecordEpisodeStatistics", {}),
             ("RewardInNormalState", dict(reward=0.5))]
    env = pikazoo_v0.env(num_envs=1, scalar_api=True, seed=11, winning_score=1)
    for name, kw in stack:
        env = getattr(W, name)(env, **{k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()})
    ref = WrappedOracle(1, stack, seed=11, winning_score=1)
    env.reset(), ref.reset()
    t, raw = 0, env.unwrapped
    while env.agents:
        acts = raw.random_actions(3, t)
        a1, a2 = int(acts["player_1"][0]), int(acts["player_2"][0])
        obs, rew, term, trunc, infos = env.step({"player_1": a1, "player_2": a2})
        o, r, tm, ep = ref.step([a1], [a2])
        assert isinstance(rew["player_1"], float) and abs(rew["player_1"] - r[0][0]) < 1e-6 and abs(rew["player_2"] - r[1][0]) < 1e-6
        assert term["player_1"] == bool(tm[0]) and ("episode" in infos["player_1"]) == bool(tm[0])
        t += 1
    assert infos["player_1"]["episode"]["l"] == t == int(ep["l"][0]) and abs(infos["player_2"]["episode"]["r"] - ep["r"][1][0]) < 2e-6


def test_wrappers_surface():
    from pikazoo_amd import pikazoo_v0
    from pikazoo_amd.wrappers import RewardByBallPosition, SimplifyAction

    table = (0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01)
    env = pikazoo_v0.env(num_envs=4096, seed=3)
    env = SimplifyAction(env)
    env = RewardByBallPosition(env, additional_reward=table, x_line=216, y_line=176)
    assert env.action_space("player_1").n == 13 and env.action_space("player_2").n == 13
    obs, _ = env.reset()
    for t in range(50):
        a = env.unwrapped.random_actions(9, t)
        assert int(a["player_1"].max()) <= 12
        obs, rew, term, trunc, infos = env.step(a)
        assert rew["player_1"].dtype == torch.float32
        # recompute the wrapper on the host from the observation (reward_by_ball_position.py:22-29)
        bx, by = obs["player_1"][:, 26], obs["player_1"][:, 27]
        zone = (by > 176).long() + 2 * (bx >= 216).long()
        tab = torch.tensor(table, device=bx.device)
        base = torch.where(env.unwrapped.state[41] != 0,
                           torch.where(env.unwrapped.state[40] != 0, -1.0, 1.0), 0.0)
        assert torch.equal(rew["player_1"], (base + tab[zone]).float())
        assert torch.equal(rew["player_2"], (-base + tab[4 + zone]).float())
    with pytest.raises(AssertionError):
        RewardByBallPosition(pikazoo_v0.env(num_envs=2), additional_reward=(1, 2, 3))
    # a second SimplifyAction is what it is in the reference: a second map, applied outside the kernel
    assert SimplifyAction(env).fused is False and env.unwrapped._unfused == ["SimplifyAction"]


def test_remaining_wrappers_surface():
    """RewardInNormalState / NormalizeObservation / RecordEpisodeStatistics / ConvertSingleAgent: the
    reference's class names and constructor signatures (pikazoo/wrappers/__init__.py:1-6)."""
    from pikazoo_amd import pikazoo_v0
    from pikazoo_amd.wrappers import (ConvertSingleAgent, NormalizeObservation, RecordEpisodeStatistics,
                                      RewardByBallPosition, RewardInNormalState)

    n = 2048
    env = pikazoo_v0.env(num_envs=n, seed=2, winning_score=1)
    env = RecordEpisodeStatistics(NormalizeObservation(RewardInNormalState(env, reward=-0.5)))
    assert env.observation_space("player_1").dtype == np.float32 and env.observation_space("player_1").shape == (35,)
    obs, infos = env.reset()
    assert obs["player_1"].dtype == torch.float32 and float(obs["player_1"].min()) >= 0.0
    raw = env.unwrapped
    seen_done = 0
    for t in range(300):
        obs, rew, term, trunc, infos = env.step(raw.random_actions(5, t))
        o = obs["player_1"]
        # every column except the ball's y velocity (reference bound is only "observed") stays in [0, 1]
        cols = [c for c in range(35) if c != 33]
        assert float(o[:, cols].min()) >= 0.0 and float(o[:, cols].max()) <= 1.0
        assert rew["player_1"].dtype == torch.float32
        assert bool(((rew["player_1"] == -0.5) | (rew["player_1"].abs() == 1.0)).all())
        ep = infos["player_1"]["episode"]
        assert ep["l"].shape == (n,) and ep["r"].dtype == torch.float64  # summed like Python floats
        done = term["player_1"]
        if bool(done.any()):
            seen_done += int(done.sum())
            # a winning_score=1 episode: l frames, all but the last paid -0.5, the last +-1
            assert torch.allclose(ep["r"][done], (ep["l"][done] - 1).double() * -0.5 + rew["player_1"][done].double())
    assert seen_done > 0
    assert env.episode_lengths["player_1"].shape == (n,)

    # stack orders the kernel cannot fuse are neither refused nor reinterpreted: the wrapper applies itself to the step's
    # outputs (trajectories against the reference: tests/test_gpu_parity.py, tests/golden/unfused_*.npz)
    assert RewardByBallPosition(NormalizeObservation(pikazoo_v0.env(num_envs=4)), additional_reward=(0,) * 8).fused is False
    e = RecordEpisodeStatistics(RewardInNormalState(pikazoo_v0.env(num_envs=4), 0.1))
    assert e.fused and RewardByBallPosition(e, additional_reward=(0,) * 8).fused is False
    twice = NormalizeObservation(NormalizeObservation(pikazoo_v0.env(num_envs=4)))
    assert twice.fused is False and twice.env.fused and twice.unwrapped._unfused == ["NormalizeObservation"]

    # single-agent view: the other side plays the seeded device policy
    single = ConvertSingleAgent(pikazoo_v0.env(num_envs=64, seed=4, winning_score=2), "player_2", opponent_seed=9)
    o, info = single.reset()
    assert o.shape == (64, 35) and "score" in info
    twin = pikazoo_v0.env(num_envs=64, seed=4, winning_score=2)
    twin.reset()
    for t in range(100):
        act = torch.full((64,), t % 18, dtype=torch.int32, device="cuda:0")
        o, r, term, trunc, info = single.step(act)
        other = twin.random_actions(9, t)["player_1"]
        o2, r2, term2, _, _ = twin.step({"player_1": other, "player_2": act})
        assert torch.equal(o, o2["player_2"]) and torch.equal(r, r2["player_2"]) and torch.equal(term, term2["player_2"])
    with pytest.raises(AssertionError):
        ConvertSingleAgent(twin, "player_3")


def test_record_episode_statistics_scalar_api():
    """record_episode_statistics.py:31-39 through the scalar API: the "episode" entry exists only on the
    terminal step and carries the sums of that episode."""
    from pikazoo_amd import pikazoo_v0
    from pikazoo_amd.wrappers import RecordEpisodeStatistics

    env = RecordEpisodeStatistics(pikazoo_v0.env(num_envs=1, scalar_api=True, auto_reset=False, winning_score=3,
                                                 seed=21))
    rng = np.random.default_rng(3)
    for episode in range(3):
        env.reset()
        total, length = {"player_1": 0, "player_2": 0}, 0
        while env.agents:
            acts = {a: int(rng.integers(0, 18)) for a in env.agents}
            obs, rew, term, trunc, infos = env.step(acts)
            length += 1
            for a in rew:
                total[a] += rew[a]
            assert ("episode" in infos["player_1"]) == term["player_1"]
        assert infos["player_1"]["episode"] == {"r": total["player_1"], "l": length}
        assert infos["player_2"]["episode"] == {"r": total["player_2"], "l": length}
        assert abs(total["player_1"]) <= 3 and total["player_1"] == -total["player_2"]


def _fresh(n, seed, **kw):
    from pikazoo_amd import pikazoo_v0

    kw.setdefault("validate_actions", False)
    env = pikazoo_v0.env(num_envs=n, seed=seed, **kw)
    env.reset()
    return env


def test_calls_are_ordered_on_the_callers_stream_and_envs_are_independent():
    """include/pikazoo_hip.h: every entry point is asynchronous on the stream it is given and the
    library keeps no global state -- two environments driven from two side streams at the same time
    must produce what each produces alone on the default stream."""
    n, steps = 8192, 200
    acts = torch.randint(0, 18, (steps, 2, n), dtype=torch.int32, device="cuda")
    alone = []
    for seed, kw in ((3, {}), (4, {"is_player2_computer": True})):
        env = _fresh(n, seed, **kw)
        for t in range(steps):
            env.step({"player_1": acts[t, 0], "player_2": acts[t, 1]})
        alone.append(env.state.clone())
    torch.cuda.synchronize()

    envs = [_fresh(n, 3), _fresh(n, 4, is_player2_computer=True)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for t in range(steps):
        for env, s in zip(envs, streams):
            with torch.cuda.stream(s):
                env.step({"player_1": acts[t, 0], "player_2": acts[t, 1]})
    for s in streams:
        s.synchronize()
    for env, want in zip(envs, alone):
        assert torch.equal(env.state, want)


@pytest.mark.parametrize("validate", [False, True])
def test_graph_replay_equals_eager_launches(validate):
    """bench.py replays K captured pz_step launches (hipGraph); the replayed trajectory must be the
    eager one, including the outputs left in the environment's buffers.  With validate_actions (the default) the
    captured launches count out-of-range actions like eager ones; the env's polling stays out of the capture."""
    n, k = 65536, 64
    acts = torch.randint(0, 18, (k, 2, n), dtype=torch.int32, device="cuda")
    eager = _fresh(n, 9)
    for t in range(k):
        obs, rew, term, _, _ = eager.step({"player_1": acts[t, 0], "player_2": acts[t, 1]})
    want_state = eager.state.clone()
    want_obs, want_rew, want_term = obs["player_1"].clone(), rew["player_1"].clone(), term["player_1"].clone()

    env = _fresh(n, 9, validate_actions=validate, validate_every=16)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            for t in range(k):
                obs, rew, term, _, _ = env.step({"player_1": acts[t, 0], "player_2": acts[t, 1]})
        graph.replay()
    side.synchronize()
    assert torch.equal(env.state, want_state)
    assert torch.equal(obs["player_1"], want_obs) and torch.equal(rew["player_1"], want_rew)
    assert torch.equal(term["player_1"], want_term)
    if validate:
        env.check_actions()
        acts[5, 1, 77] = 18  # the graph reads the tape at replay time: a bad action now is counted by the replayed launch
        with torch.cuda.stream(side):
            graph.replay()
        side.synchronize()
        with pytest.raises(IndexError):
            env.check_actions()


def test_output_ring_keeps_the_previous_results_intact():
    """The reference returns fresh arrays from every step (pikazoo_env.py:215-235); this env returns views of its own
    buffers -- by default one set, overwritten by the next step; with output_ring=k the last k results stay intact."""
    from pikazoo_amd import pikazoo_v0

    one = pikazoo_v0.env(num_envs=300, seed=5, validate_actions=False)
    two = pikazoo_v0.env(num_envs=300, seed=5, validate_actions=False, output_ring=2)
    with pytest.raises(ValueError):
        pikazoo_v0.env(num_envs=4, output_ring=0)
    o1, _ = one.reset()
    o2, _ = two.reset()
    prev1 = prev2 = kept = None
    for t in range(8):
        a = one.random_actions(3, t)
        r1, r2 = one.step(a), two.step(a)
        for x, y in zip(r1[:3], r2[:3]):  # observations, rewards, terminations: the same trajectory either way
            for agent in one.possible_agents:
                assert torch.equal(x[agent], y[agent])
        if prev2 is not None:
            # ring of two: the previous step's tensors are other tensors and still hold the previous step's values
            assert r2[0]["player_1"].data_ptr() != prev2[0]["player_1"].data_ptr()
            for got, want in zip((prev2[0]["player_1"], prev2[0]["player_2"], prev2[1]["player_1"], prev2[2]["player_1"]), kept):
                assert torch.equal(got, want)
            # default: one set of buffers, the same tensors every step
            assert r1[0]["player_1"].data_ptr() == prev1[0]["player_1"].data_ptr()
        prev1, prev2 = r1, r2
        kept = [r2[0]["player_1"].clone(), r2[0]["player_2"].clone(), r2[1]["player_1"].clone(), r2[2]["player_1"].clone()]
    # reset() rotates too: the first observations of a ring env survive the first step
    three = pikazoo_v0.env(num_envs=64, seed=1, validate_actions=False, output_ring=3)
    first, _ = three.reset()
    snap = first["player_2"].clone()
    three.step(three.random_actions(1, 0))
    three.step(three.random_actions(1, 1))
    assert torch.equal(first["player_2"], snap)


@pytest.mark.parametrize("ring", [1, 2])
def test_single_frame_views_follow_a_k_frame_launch_lazily(ring):
    """After rollout_random / step_many the env's own single-frame buffers owe the last frame; they are brought up
    to date when looked at (no copy launches behind every k-frame launch), and a step in between simply overwrites
    them: the trajectory and every returned tuple are what k single steps give."""
    from pikazoo_amd import pikazoo_v0

    kw = dict(num_envs=256, seed=11, validate_actions=False, winning_score=2)
    a = pikazoo_v0.env(output_ring=ring, **kw)
    b = pikazoo_v0.env(**kw)
    a.reset(), b.reset()
    k = 8
    traj = a.rollout_random(5, k)
    for t in range(k):
        last = b.step(b.random_actions(5, t))
    assert a._last_traj is traj  # nothing copied yet
    obs = a._pack_obs()  # looked at: settled
    assert a._last_traj is None
    for ag in a.possible_agents:
        assert torch.equal(obs[ag], last[0][ag]) and torch.equal(obs[ag], traj["obs"][ag][-1])
    assert torch.equal(a._rewards()[0], last[1]["player_1"]) and torch.equal(a._term, last[2]["player_1"])
    # a step right behind a k-frame launch: its results are the step's, not the rollout's last frame
    traj = a.rollout_random(5, k)
    for t in range(k, 2 * k):
        b.step(b.random_actions(5, t))
    ra, rb = a.step(a.random_actions(5, 2 * k)), b.step(b.random_actions(5, 2 * k))
    assert a._last_traj is None
    for x, y in zip(ra[:3], rb[:3]):
        for ag in a.possible_agents:
            assert torch.equal(x[ag], y[ag])
    # a masked reset re-derives every game's observation row from the state: the unmasked games still show the k-frame launch's last frame
    tape = torch.stack([torch.stack(list(b.random_actions(5, 2 * k + 1 + t).values())) for t in range(k)])
    traj = a.step_many(tape)
    for t in range(k):
        last = b.step({"player_1": tape[t, 0], "player_2": tape[t, 1]})
    mask = torch.zeros(256, dtype=torch.bool, device="cuda")
    mask[::3] = True
    oa, _ = a.reset(mask=mask)
    ob, _ = b.reset(mask=mask)
    for ag in a.possible_agents:
        assert torch.equal(oa[ag], ob[ag])
        assert torch.equal(oa[ag][~mask], traj["obs"][ag][-1][~mask])
    assert torch.equal(a.state, b.state)


def test_trajectory_tensors_are_placed_by_measurement_and_results_do_not_depend_on_it():
    """rollout_random / step_many allocate their two observation tensors through placement.alloc_pair (the pair is
    probed with pz_probe_write and, when it shares a rank of the device memory, re-allocated elsewhere); small
    tensors are left alone; the trajectory is the same with and without."""
    from pikazoo_amd import pikazoo_v0, placement

    kw = dict(num_envs=65536, seed=3, validate_actions=False)
    a = pikazoo_v0.env(**kw)
    b = pikazoo_v0.env(place_trajectories=False, **kw)
    a.reset(), b.reset()
    small = a.rollout_random(2, 4)  # 37 MB per tensor: nothing to place
    assert a.trajectory_placement == {"probed": False, "bytes": 4 * 65536 * 35 * 4}
    b.rollout_random(2, 4)
    assert b.trajectory_placement == {}
    ta, tb = a.rollout_random(2, 32, t0=4), b.rollout_random(2, 32, t0=4)
    info = a.trajectory_placement
    assert info["probed"] and info["bytes"] == 32 * 65536 * 35 * 4 and 0.5 < info["ratio"] < 1.3
    assert info["distinct"] == (info["ratio"] < placement.DISTINCT_BELOW)
    for ag in a.possible_agents:
        assert torch.equal(ta["obs"][ag], tb["obs"][ag]) and torch.equal(ta["rewards"][ag], tb["rewards"][ag])
    assert torch.equal(ta["terminations"], tb["terminations"]) and torch.equal(a.state, b.state)
    x, y = ta["_obs"]
    if info["distinct"]:
        assert placement.pair_ratio(x, y) < placement.DISTINCT_BELOW  # (measured again; overwrites both)
    del small
    # the same blocks handed out again are not probed again
    ptrs = {x.data_ptr(), y.data_ptr()}
    del ta, x, y
    again = a.rollout_random(2, 32, t0=36)
    if {t.data_ptr() for t in again["_obs"]} == ptrs:
        assert a.trajectory_placement.get("cached") is True


def test_placement_gives_up_promptly_and_leaves_the_process_memory_alone(monkeypatch):
    """The allocator walk of placement.alloc_pair is bounded and optional: with no budget, with the free memory 'below
    two spacers', with another tenant on the device, with ranks sharing it, or switched off by the environment it
    returns the first pair at once -- no spacer allocated, torch's cached memory untouched, no OOM -- and says why;
    a walk that does run keeps its spacers in a private pool (the process's reserved memory is what it was, plus at
    most the one candidate block that is kept)."""
    import time

    from pikazoo_amd import placement

    dev = torch.device("cuda:0")
    shape, dt = (32, 65536, 35), torch.int32
    placement.reset()
    keep = torch.empty(1 << 28, dtype=torch.uint8, device=dev)  # something of ours in torch's cache
    del keep
    cached = torch.cuda.memory_reserved(dev)
    assert cached >= 1 << 28

    def run(**kw):
        placement.reset()
        t0 = time.perf_counter()
        a, b = placement.alloc_pair(shape, dt, dev, **kw)
        took, info = time.perf_counter() - t0, dict(placement.last_info)
        assert a.shape == b.shape == shape and a.data_ptr() != b.data_ptr()
        return info, took

    info, took = run(max_spacer_bytes=0)
    assert info["probed"] and info["candidates"] == 1 and info["spacer_gib"] == 0.0 and took < 2.0
    if not info["distinct"]:
        assert "budget" in info["walk"]
    real = torch.cuda.mem_get_info
    # free memory below two spacers + two candidate blocks: the walk stops before its first allocation
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (9 << 30, real(d)[1]))
    info, took = run()
    assert info["candidates"] == 1 and info["spacer_gib"] == 0.0 and took < 2.0
    # somebody else holds memory on the device
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (real(d)[0] - (8 << 30), real(d)[1]))
    info, took = run()
    assert info["candidates"] == 1 and (info["distinct"] or "somebody else" in info["walk"])
    monkeypatch.setattr(torch.cuda, "mem_get_info", real)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")  # eight ranks on this one GPU
    info, took = run()
    assert info["candidates"] == 1 and (info["distinct"] or "shared" in info["walk"])
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    monkeypatch.setenv("PIKAZOO_PLACE_TRAJECTORIES", "0")
    info, took = run()
    assert info == {"probed": False, "bytes": 32 * 65536 * 35 * 4}
    monkeypatch.delenv("PIKAZOO_PLACE_TRAJECTORIES")
    # an allocator / torch that does not behave the way the probing assumes (here: the probe itself raises): two plainly
    # allocated tensors and the reason, never an exception -- placement moves time, not results
    def broken(*a, **kw):
        raise RuntimeError("this allocator has no private pools")
    monkeypatch.setattr(placement, "pair_ratio", broken)
    info, took = run()
    assert info["probed"] is False and "no private pools" in info["error"]
    from pikazoo_amd import pikazoo_v0

    env = pikazoo_v0.env(num_envs=65536, seed=2)
    env.reset()
    traj = env.rollout_random(5, 32)
    assert "error" in env.trajectory_placement and traj["obs"]["player_1"].shape == (32, 65536, 35)
    monkeypatch.undo()
    # a real walk: whatever it held is back with the driver afterwards, torch's own cache was never emptied
    torch.cuda.synchronize()
    before = torch.cuda.memory_reserved(dev)
    placement.reset()
    a, b = placement.alloc_pair(shape, dt, dev)
    info = dict(placement.last_info)
    torch.cuda.synchronize()
    grown = torch.cuda.memory_reserved(dev) - before
    assert grown <= 2 * (32 * 65536 * 35 * 4) + (2 << 30), (grown, info)   # the pair (+ one 1 GiB candidate block), no spacer
    assert torch.cuda.memory_reserved(dev) >= cached                       # (nobody called empty_cache on the process)
    assert info["spacer_gib"] <= placement.DEFAULT_SPACER_BUDGET / (1 << 30)
    # the block a walk found stays in a private pool of its own: the next pair of this size takes it without walking
    if info["spacer_gib"] > 0 and info["distinct"]:
        del a, b
        a, b = placement.alloc_pair(shape, dt, dev)
        again = dict(placement.last_info)
        assert again["distinct"] and again["spacer_gib"] == 0.0, (info, again)


def test_placement_keeps_a_bounded_amount_and_reset_returns_it_to_the_driver(monkeypatch):
    """What placement keeps for the next pair (the block a walk found in another rank, in a private pool that
    torch.cuda.empty_cache() cannot empty) is bounded -- KEEP_FAR_BYTES / PIKAZOO_PLACE_KEEP_GIB in all, oldest unused
    block dropped first -- is reported (`retained_gib`, retained_bytes()), and goes back to the driver with reset():
    the device's free memory returns to where it was."""
    from pikazoo_amd import placement

    dev = torch.device("cuda:0")
    placement.reset()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info(dev)[0]
    walked = 0
    for k in (32, 36, 40):  # three sizes: three kept blocks if every walk finds one
        a, b = placement.alloc_pair((k, 65536, 35), torch.int32, dev)
        info = dict(placement.last_info)
        walked += info.get("spacer_gib", 0) > 0 and info["distinct"]
        assert placement.retained_bytes() <= placement.KEEP_FAR_BYTES
        if "retained_gib" in info:
            assert abs(info["retained_gib"] * (1 << 30) - placement.retained_bytes()) < 1
        del a, b
    assert placement.retained_bytes() >= (walked > 0) * (1 << 30)
    # a limit of 1.5 GiB: a second kept block pushes the first (unused) one out
    monkeypatch.setenv("PIKAZOO_PLACE_KEEP_GIB", "1.5")
    for k in (44, 48):
        a, b = placement.alloc_pair((k, 65536, 35), torch.int32, dev)
        del a, b
        assert placement.retained_bytes() <= int(1.5 * (1 << 30)) + 48 * 65536 * 35 * 4
    monkeypatch.delenv("PIKAZOO_PLACE_KEEP_GIB")
    placement.reset()
    assert placement.retained_bytes() == 0
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info(dev)[0]
    assert free1 >= free0 - (64 << 20), (free0, free1)  # everything the walks and the kept blocks held is back


def test_the_binding_stub_of_integration_md_runs_as_written(oracle):
    """INTEGRATION.md section 2 shows the ~40-line ctypes stub a maintainer of the reference would add
    (`pikazoo/env/_hip_backend.py`).  This runs that very text -- the first python block of the file, only the library's
    path made absolute -- against the oracle: documentation that is executed cannot drift from the ABI it describes."""
    import re
    from pathlib import Path

    from pikazoo_amd import _native

    repo = Path(__file__).resolve().parent.parent
    text = (repo / "INTEGRATION.md").read_text()
    block = re.search(r"```python\n(.*?)```", text, flags=re.S).group(1)
    assert 'C.CDLL("libpikazoo_hip.so")' in block and "class HipBackend" in block
    ns = {}
    exec(block.replace('C.CDLL("libpikazoo_hip.so")', f'C.CDLL("{_native.LIB_PATH}")'), ns)  # noqa: S102
    n = 300  # (ragged: not a multiple of 64)
    cfg = ns["PzConfig"](winning_score=2, serve_mode=0, auto_reset=1, seed=11, env_id_base=5)
    backend = ns["HipBackend"](n, cfg, device="cuda:0")
    ref = oracle.OracleEnv(n, oracle.make_config(winning_score=2, seed=11, env_id_base=5))
    backend.reset()
    o1, _ = ref.reset()
    assert np.array_equal(backend.obs[0].cpu().numpy(), o1)
    g = torch.Generator(device="cpu").manual_seed(2)
    for _ in range(400):
        a = torch.randint(0, 18, (2, n), generator=g, dtype=torch.int32)
        dev = a.to("cuda:0")
        backend.step(dev[0], dev[1])
        robs, rrew, rterm = ref.step(a[0].numpy(), a[1].numpy())
    torch.cuda.synchronize()
    assert np.array_equal(backend.state.cpu().numpy(), ref.state)
    assert np.array_equal(backend.obs[1].cpu().numpy(), robs[1]) and np.array_equal(backend.rew[0].cpu().numpy(), rrew[0])
    assert np.array_equal(backend.term.cpu().numpy(), rterm)


def test_launch_floor_probe_runs_on_scratch_buffers_and_touches_nothing_else():
    """pz_probe_launch (the headline launch's geometry without its game): every `what` launches for ragged and
    full batches, writes only inside the buffers it was handed (guard words behind each stay intact), and from `what` = 2 on
    really stores (the scratch state changes).  Lives in libpikazoo_diag.so (include/pikazoo_diag.h), not in the product."""
    import diag

    lib = diag.load()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    guard = 0x5A5A5A5A
    for n in (1, 63, 64, 100, 4096):
        sizes = {"state": 44 * n, "a1": n, "a2": n, "o1": 35 * n, "o2": 35 * n, "r1": n, "r2": n}
        bufs = {k: torch.full((v + 64,), guard, dtype=torch.int32, device=dev) for k, v in sizes.items()}
        # (16-byte alignment of the observation buffers: torch allocations are 512-byte aligned)
        for k, v in sizes.items():
            bufs[k][:v] = 0
        for what in (0, 1, 2, 3):
            rc = lib.pz_probe_launch(bufs["state"].data_ptr(), n, n, bufs["a1"].data_ptr(), bufs["a2"].data_ptr(),
                                     bufs["o1"].data_ptr(), bufs["o2"].data_ptr(), bufs["r1"].data_ptr(),
                                     bufs["r2"].data_ptr(), what, 102, stream)
            assert rc == 0, (n, what, rc)
            if what == 3:  # (another step count: the looped form of the stand-in)
                assert lib.pz_probe_launch(bufs["state"].data_ptr(), n, n, bufs["a1"].data_ptr(), bufs["a2"].data_ptr(),
                                           bufs["o1"].data_ptr(), bufs["o2"].data_ptr(), bufs["r1"].data_ptr(),
                                           bufs["r2"].data_ptr(), 3, 46, stream) == 0
            torch.cuda.synchronize()
            for k, v in sizes.items():
                assert bool((bufs[k][v:] == guard).all()), (n, what, k)
            if what < 2:
                assert all(bool((bufs[k][:sizes[k]] == 0).all()) for k in sizes), (n, what)
        assert bool((bufs["o1"][:35 * n] != 0).any()) and bool((bufs["state"][:44 * n] != 0).any())
        assert bool((bufs["a1"][:n] == 0).all()) and bool((bufs["a2"][:n] == 0).all())  # the actions are only read


def test_step_through_the_bound_entry_point_equals_pz_step():
    """raw_env.step goes through pz_step_bind / pz_step_bound (the arguments prepared once per output set and
    configuration); the trajectory must be the one direct pz_step calls produce, wrappers fused later re-bind."""
    from pikazoo_amd import _native, pikazoo_v0
    from pikazoo_amd.wrappers import RewardByBallPosition, SimplifyAction

    lib = _native.load()
    n = 1000
    env = pikazoo_v0.env(num_envs=n, seed=9, is_player2_computer=True, validate_actions=False)
    ref = pikazoo_v0.env(num_envs=n, seed=9, is_player2_computer=True, validate_actions=False)
    env.reset()
    ref.reset()
    raw = ref.unwrapped
    stream = torch.cuda.current_stream().cuda_stream

    def direct(a):  # the twelve-argument call on ref's own buffers
        p = raw._ptrs
        assert lib.pz_step(p[0], n, raw._stride, raw._cfg_ref, a["player_1"].data_ptr(), a["player_2"].data_ptr(), p[1], p[2],
                           p[3], p[4], p[5], raw._stats_ptr(), raw._tables_ref, stream) == 0

    for t in range(40):
        a = env.random_actions(2, t)
        obs, rew, term, _, _ = env.step(a)
        direct(a)
    assert torch.equal(env.state, ref.state) and torch.equal(obs["player_1"], raw._obs[0])
    # a wrapper fused after the first steps changes the configuration: the next step must not use the stale block
    env = RewardByBallPosition(SimplifyAction(env), (0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01), 216, 176)
    raw._fuse_simplify_action()
    raw._fuse_ballpos_reward((0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01), 216, 176)
    for t in range(40, 80):
        a = env.unwrapped.random_actions(2, t)
        obs, rew, term, _, _ = env.step(a)
        direct(a)
    assert rew["player_1"].dtype == torch.float32
    assert torch.equal(env.unwrapped.state, ref.state) and torch.equal(rew["player_1"].view(torch.int32), raw._rew_raw[0])
    # the block is plain data the library validates: an unbound one is refused, bad arguments fail at bind time
    import ctypes as C

    blank = C.create_string_buffer(int(lib.pz_step_bound_bytes()))
    a = env.unwrapped.random_actions(2, 0)
    assert lib.pz_step_bound(blank, a["player_1"].data_ptr(), a["player_2"].data_ptr(), stream) == -3  # PZ_E_CONFIG
    p = raw._ptrs
    assert lib.pz_step_bind(blank, p[0], n, raw._stride, raw._cfg_ref, p[1] + 4, p[2], p[3], p[4], p[5], None, None) == -4
    assert lib.pz_step_bind(blank, p[0], n, raw._stride, raw._cfg_ref, p[1], p[2], p[3], p[4], p[5], None, None) == 0
    assert lib.pz_step_bound(blank, None, a["player_2"].data_ptr(), stream) == -1


def test_scalar_api_default_loop_terminates_like_the_reference():
    """`while env.agents:` around a scalar_api env must end at the game's end without any extra kwarg
    (auto_reset defaults to False there; the reference empties `agents`, pikazoo_env.py:237-238)."""
    from pikazoo_amd import pikazoo_v0

    env = pikazoo_v0.env(num_envs=1, scalar_api=True, winning_score=1, seed=3)
    assert env.auto_reset is False
    env.reset()
    rng = np.random.default_rng(0)
    frames = 0
    while env.agents:
        obs, rew, term, trunc, infos = env.step({a: int(rng.integers(0, 18)) for a in env.agents})
        frames += 1
        assert frames < 5000
    assert term == {"player_1": True, "player_2": True} and sorted(rew.values()) == [-1, 1]
    with pytest.raises(RuntimeError):
        env.step({"player_1": 0, "player_2": 0})
    env.reset()
    assert env.agents == ["player_1", "player_2"]
    # the batched default stays auto-reset
    assert pikazoo_v0.env(num_envs=4).auto_reset is True


def test_checkpoint_round_trip_with_fused_statistics():
    """state_dict() carries the RecordEpisodeStatistics words, the counters and the configuration: a restore in
    the middle of episodes continues bit for bit (incl. infos["episode"]), and a checkpoint of another seed /
    wrapper stack is refused."""
    from pikazoo_amd import pikazoo_v0
    from pikazoo_amd.wrappers import RecordEpisodeStatistics, RewardInNormalState

    def make(seed=5, reward=0.25):
        e = pikazoo_v0.env(num_envs=512, seed=seed, env_id_base=40, winning_score=1, is_player2_computer=True,
                           validate_actions=False)
        return RecordEpisodeStatistics(RewardInNormalState(e, reward))

    a = make()
    a.reset()
    a.unwrapped.step_random(3, k=130)
    sd = a.unwrapped.state_dict()
    assert sd["episode_stats"] is not None and int(a.unwrapped.episode_lengths.max()) > 0
    a.unwrapped.step_random(3, k=170)
    b = make()
    b.reset()
    b.unwrapped.load_state_dict(sd)
    assert b.unwrapped.steps_done == 130
    b.unwrapped.step_random(3, k=170)
    assert torch.equal(a.unwrapped.state, b.unwrapped.state)
    assert torch.equal(a.unwrapped._stats, b.unwrapped._stats)
    assert a.unwrapped.episodes_done == b.unwrapped.episodes_done > 0
    with pytest.raises(ValueError, match="configuration"):
        make(seed=6).unwrapped.load_state_dict(sd)
    with pytest.raises(ValueError, match="configuration"):
        make(reward=0.5).unwrapped.load_state_dict(sd)
    plain = pikazoo_v0.env(num_envs=512, seed=5, env_id_base=40, winning_score=1, is_player2_computer=True)
    with pytest.raises(ValueError):
        plain.load_state_dict(sd)


@pytest.mark.parametrize("name", ["single_agent_player_2", "single_agent_player_1_vs_computer"])
def test_convert_single_agent_matches_the_reference_wrapper(name):
    """ConvertSingleAgent against fixtures captured from the reference's own wrapper class
    (wrappers/convert_single_agent.py:16-28; its `action_space(other).sample()` fed the policy stream the product's
    wrapper draws on device): what step() returns for the controlled side, frame by frame."""
    from conftest import load_golden
    from pikazoo_amd import pikazoo_v0
    from pikazoo_amd.wrappers import ConvertSingleAgent

    d = load_golden(name)
    meta = d["meta"]
    env = pikazoo_v0.env(num_envs=meta["lanes"], seed=meta["seed"], env_id_base=meta["env_id_base"],
                         validate_actions=False, **meta["env_kwargs"])
    single = ConvertSingleAgent(env, meta["side"], opponent_seed=meta["opponent_seed"])
    obs, info = single.reset()
    assert np.array_equal(obs.cpu().numpy(), d["obs_reset"]) and "score" in info
    acts = torch.as_tensor(d["actions"].astype(np.int32), device="cuda:0")
    T, L = meta["steps"], meta["lanes"]
    h_obs = torch.empty((T, L, 35), dtype=torch.int32, device="cuda:0")
    h_rew = torch.empty((T, L), dtype=torch.int32, device="cuda:0")
    h_term = torch.empty((T, L), dtype=torch.bool, device="cuda:0")
    h_score = torch.empty((T, L, 2), dtype=torch.int32, device="cuda:0")
    for t in range(T):
        o, r, term, trunc, info = single.step(acts[t])
        h_obs[t].copy_(o), h_rew[t].copy_(r), h_term[t].copy_(term), h_score[t].copy_(info["score"])
    assert not bool(trunc.any())
    assert np.array_equal(h_obs.cpu().numpy(), d["obs"].astype(np.int32))
    assert np.array_equal(h_rew.cpu().numpy(), d["rew"].astype(np.int32))
    assert np.array_equal(h_term.cpu().numpy().astype(np.uint8), d["term"])
    assert np.array_equal(h_score.cpu().numpy(), d["score"].astype(np.int32))
    st = d["states"][-1].astype(np.int32)
    st[43] = d["rng_counter"][-1]
    assert np.array_equal(env.state.cpu().numpy(), st)


def test_pettingzoos_own_parallel_api_test_when_it_is_installed():
    """The reference's one API test is `parallel_api_test(env, 1_000_000)` (tests/test_parallel_api.py:5-7).  With
    PettingZoo installed `raw_env` IS a ParallelEnv (pikazoo_amd/env.py) and the harness itself runs against the scalar
    API (num_envs = 1: the reference's exact return types); this image has no PettingZoo, where the test skips and
    `test_parallel_api_conformance_scalar` above stays the restatement of what the harness checks."""
    pz_test = pytest.importorskip("pettingzoo.test", reason="pettingzoo is not installed in this image (no network): the "
                                  "harness is restated in test_parallel_api_conformance_scalar")
    from pettingzoo import ParallelEnv

    from pikazoo_amd import pikazoo_v0

    env = pikazoo_v0.env(num_envs=1, scalar_api=True)
    assert isinstance(env, ParallelEnv)
    pz_test.parallel_api_test(env, num_cycles=10_000)

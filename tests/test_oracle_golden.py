"""CPU tests: the oracle (oracle/pz_oracle.c) against golden vectors captured from the reference.

The fixtures under tests/golden/ were produced by oracle/ref_capture.py from the UNMODIFIED
reference (pikazoo/env/pikazoo_env.py:149-240 driving pikazoo/env/physics.py), with the env RNG
stream injected.  Bit-exact bar: every one of the 44 state words, both 35-dim observations, rewards
and terminations, at every step.
"""
import numpy as np
import pytest

from conftest import DIGEST_FIXTURES, FULL_FIXTURES, UNFUSED_FIXTURES, golden_state, load_golden, oracle_config_from_meta


def test_philox_known_answers(oracle):
    # Random123 kat_vectors for philox4x32-10
    assert oracle.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert oracle.philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == (
        0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert oracle.philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344),
                                (0xA4093822, 0x299F31D0)) == (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def test_philox_numpy_cross_check(oracle):
    rng = np.random.default_rng(0)
    ctr = rng.integers(0, 2**32, size=(4, 257), dtype=np.uint64)
    key = (0xDEADBEEF, 0x12345678)
    got = oracle.philox4x32_10_numpy(ctr, key)
    for j in range(0, 257, 16):
        assert tuple(int(g[j]) for g in got) == oracle.philox4x32_10(ctr[:, j], key)


def test_random_actions_stream(oracle):
    a1, a2 = oracle.random_actions(1000, 5, 42, 3, 18)
    assert a1.min() >= 0 and a1.max() <= 17 and a2.min() >= 0 and a2.max() <= 17
    # lane i of a batch == lane 0 of a batch based at i (global env ids)
    b1, b2 = oracle.random_actions(1, 5 + 123, 42, 3, 18)
    assert (a1[123], a2[123]) == (b1[0], b2[0])
    c1, _ = oracle.random_actions(1000, 5, 42, 4, 18)
    assert not np.array_equal(a1, c1)
    s1, s2 = oracle.random_actions(5000, 0, 1, 0, 13)
    assert s1.max() == 12 and s2.max() == 12


def test_fresh_env_first_observation(oracle):
    # SURVEY appendix A.3 (probe-confirmed on the reference)
    env = oracle.OracleEnv(3, oracle.make_config(seed=1))
    o1, o2 = env.reset()
    exp1 = [36, 244, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, 396, 244, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0,
            56, 0, 0, 0, 0, 0, 0, 1, 0]
    assert o1[0].tolist() == exp1
    assert o2[0].tolist() == exp1[13:26] + exp1[0:13] + exp1[26:]
    assert (env.state[43] == 4).all()  # ctor 2 draws + reset 2 draws


@pytest.mark.parametrize("name", FULL_FIXTURES)
def test_oracle_matches_reference_trajectory(oracle, name):
    d = load_golden(name)
    meta = d["meta"]
    cfg = oracle_config_from_meta(meta)
    env = oracle.OracleEnv(meta["lanes"], cfg)
    assert np.array_equal(env.state, d["state_ctor"])
    o1, o2 = env.reset()
    assert np.array_equal(env.state, d["state0"])
    if not cfg.normalize_obs:
        assert np.array_equal(o1, d["obs_reset"][:, 0]) and np.array_equal(o2, d["obs_reset"][:, 1])
    fused = bool(cfg.ballpos_reward or cfg.normal_state_mode)
    float_obs = bool(cfg.normalize_obs)
    if float_obs:  # the reference divides in float64; the build emits the float32 rounding of it
        assert o1.dtype == np.float32 and np.array_equal(o1, d["obs_reset"][:, 0].astype(np.float32))
        assert np.array_equal(o2, d["obs_reset"][:, 1].astype(np.float32))
    for t in range(meta["steps"]):
        a = d["actions"][t].astype(np.int32)
        # the fixture's actions are the build's own Philox policy stream
        r1, r2 = oracle.random_actions(meta["lanes"], meta["env_id_base"], meta["action_seed"], t,
                                       meta["n_actions"])
        assert np.array_equal(a[0], r1) and np.array_equal(a[1], r2)
        obs, rew, term = env.step(a[0], a[1])
        st = golden_state(d, t)
        if not np.array_equal(env.state, st):
            f, l = np.argwhere(env.state != st)[0]
            pytest.fail(f"{name}: step {t} lane {l} field {oracle.FIELD_NAMES[f]}: "
                        f"oracle {env.state[f, l]} != reference {st[f, l]}")
        exp_obs = d["obs"][t].astype(np.float32) if float_obs else d["obs"][t]
        assert np.array_equal(obs[0], exp_obs[0]), (name, t)
        assert np.array_equal(obs[1], exp_obs[1]), (name, t)
        assert np.array_equal(term, d["term"][t]), (name, t)
        if cfg.episode_stats_mode:
            # infos[agent]["episode"] = {"r", "l"} appears exactly on terminal steps
            # (record_episode_statistics.py:34-39); the counters hold those values then
            done = d["ep_l"][t] >= 0
            assert np.array_equal(done, term.astype(bool))
            if done.any():
                assert np.array_equal(env.episode_lengths[done], d["ep_l"][t][done])
                np.testing.assert_allclose(env.episode_returns[:, done], d["ep_r"][t][:, done], rtol=0, atol=2e-6)
        if fused:
            # reference adds a Python float to an int in float64 (reward_by_ball_position.py:29);
            # the build emits float32: tolerance = 1 ulp of fp32 at |r|<=~10 (1e-6 abs)
            for i in range(2):
                assert rew[i].dtype == np.float32
                np.testing.assert_allclose(rew[i], d["rew"][t, i], rtol=0, atol=1e-6)
        else:
            assert np.array_equal(rew[0], d["rew"][t, 0]) and np.array_equal(rew[1], d["rew"][t, 1])
    assert d["term"].sum() == meta["episodes"]


WRAPPED_FUSABLE = ["cfg5_wrappers_float", "wrappers_int_table", "normal_state_inside_ballpos", "normal_state_outside_ballpos",
                   "normalize_observation", "record_stats_raw", "full_wrapper_stack"]


@pytest.mark.parametrize("name", UNFUSED_FIXTURES + WRAPPED_FUSABLE)
def test_wrapper_restatement_matches_reference_on_any_stack_order(oracle, name):
    """oracle/wrappers_oracle.py -- every wrapper of a stack applied in numpy to the outputs of the UNWRAPPED oracle, in
    the stack's own order -- against the reference: on the stacks the kernel cannot fuse (RewardByBallPosition above
    NormalizeObservation, statistics between two reward wrappers, doubled wrappers: tests/golden/unfused_*.npz) and,
    to tie it to the fused restatement in pz_oracle.c, on the fusable ones too.  float64 like the reference: observations
    and rewards must agree to the last bit or two of a double."""
    from oracle.ref_capture import wrapper_stack
    from oracle.wrappers_oracle import WrappedOracle

    d = load_golden(name)
    meta = d["meta"]
    kw = meta["env_kwargs"]
    env = WrappedOracle(meta["lanes"], wrapper_stack(meta["wrappers"]), winning_score=kw.get("winning_score", 15),
                        serve=kw.get("serve", "winner"), is_player1_computer=kw.get("is_player1_computer", False),
                        is_player2_computer=kw.get("is_player2_computer", False), seed=meta["seed"],
                        env_id_base=meta["env_id_base"])
    assert np.array_equal(env.raw_state, d["state_ctor"])
    o1, o2 = env.reset()
    assert np.array_equal(env.raw_state, d["state0"])
    np.testing.assert_allclose(o1, d["obs_reset"][:, 0], rtol=0, atol=1e-15)
    np.testing.assert_allclose(o2, d["obs_reset"][:, 1], rtol=0, atol=1e-15)
    has_stats = "ep_l" in d
    for t in range(meta["steps"]):
        a = d["actions"][t].astype(np.int32)
        obs, rew, term, episode = env.step(a[0], a[1])
        assert np.array_equal(env.raw_state, golden_state(d, t)), (name, t)
        assert np.array_equal(term, d["term"][t]), (name, t)
        for i in range(2):
            np.testing.assert_allclose(obs[i], d["obs"][t, i], rtol=0, atol=1e-15, err_msg=f"{name} step {t}")
            np.testing.assert_allclose(rew[i], d["rew"][t, i], rtol=0, atol=1e-12, err_msg=f"{name} step {t}")
        if has_stats:
            done = d["ep_l"][t] >= 0
            assert np.array_equal(done, term.astype(bool))
            if done.any():
                assert np.array_equal(episode["l"][done], d["ep_l"][t][done])
                np.testing.assert_allclose(episode["r"][:, done], d["ep_r"][t][:, done], rtol=0, atol=1e-9)
    assert d["term"].sum() == meta["episodes"]


@pytest.mark.parametrize("name", DIGEST_FIXTURES)
def test_oracle_matches_reference_long_run_digests(oracle, name):
    d = load_golden(name)
    meta = d["meta"]
    cfg = oracle_config_from_meta(meta)
    env = oracle.OracleEnv(meta["lanes"], cfg, nthreads=4)
    env.reset()
    assert np.array_equal(env.state, d["state0"])
    every = meta["digest_every"]
    episodes = 0
    for k, dg in enumerate(d["digests"]):
        episodes += env.rollout_random(meta["action_seed"], k * every, every)
        assert env.digest() == int(dg), f"{name}: digest mismatch after {(k + 1) * every} steps"
    assert np.array_equal(env.state, d["final_state"])
    assert episodes == meta["episodes"]


def test_observation_symmetry_property(oracle):
    """The reference's own property test (tests/env/test_env.py:7-21): both computer, action 0."""
    n = 16
    env = oracle.OracleEnv(n, oracle.make_config(winning_score=15, is_player1_computer=True,
                                                 is_player2_computer=True, seed=3, auto_reset=False))
    o1, o2 = env.reset()
    zeros = np.zeros(n, np.int32)
    steps = 0
    while True:
        assert np.array_equal(o1[:, 0:13], o2[:, 13:26]) and np.array_equal(o1[:, 13:26], o2[:, 0:13])
        assert np.array_equal(o1[:, 26:], o2[:, 26:])
        if env.term.all() or steps > 60000:
            break
        (o1, o2), _, _ = env.step(zeros, zeros)
        steps += 1
    assert env.term.all(), "all games should finish (computer vs computer ~15k steps)"


def test_no_auto_reset_freezes_terminated_lanes(oracle):
    cfg = oracle.make_config(winning_score=1, seed=9, auto_reset=False)
    env = oracle.OracleEnv(8, cfg)
    env.reset()
    for t in range(400):
        a1, a2 = oracle.random_actions(8, 0, 5, t)
        env.step(a1, a2)
    assert env.term.all()
    frozen = env.state.copy()
    obs, rew, term = env.step(a1, a2)
    assert np.array_equal(env.state, frozen) and term.all()
    assert (rew[0] == 0).all() and (rew[1] == 0).all()
    # masked reset revives only the chosen lanes
    mask = np.array([1, 0, 1, 0, 0, 0, 0, 1], np.uint8)
    env.reset(mask)
    assert (env.state[42] == (1 - mask)).all()
    assert (env.state[38:40, mask.astype(bool)] == 0).all()


def test_multithreaded_step_equals_scalar(oracle):
    cfg = oracle.make_config(is_player2_computer=True, seed=11)
    e1, e4 = oracle.OracleEnv(1000, cfg, nthreads=1), oracle.OracleEnv(1000, cfg, nthreads=4)
    e1.reset(), e4.reset()
    for t in range(200):
        a1, a2 = oracle.random_actions(1000, 0, 8, t)
        e1.step(a1, a2), e4.step(a1, a2)
    assert np.array_equal(e1.state, e4.state) and e1.digest() == e4.digest()
    e1.rollout_random(8, 200, 300)
    for t in range(200, 500):
        a1, a2 = oracle.random_actions(1000, 0, 8, t)
        e4.step(a1, a2)
    assert np.array_equal(e1.state, e4.state)
    assert np.array_equal(e1.obs[0], e4.obs[0]) and np.array_equal(e1.term, e4.term)


def test_flight_simulation_edge_cases(oracle):
    # loop cap (physics.py:33,681): a ball that can never come down returns x at the cap
    # ceiling rule forces yv=1 so every flight lands; check determinism on the extremes instead
    assert oracle.expected_landing_x(56, 0, 0, 1) == 56
    # wall bounce asymmetry: left wall at x<20, right wall at x>432 (physics.py:403)
    xl = oracle.expected_landing_x(30, 200, -20, -10)
    xr = oracle.expected_landing_x(420, 200, 20, -10)
    assert 20 <= xl <= 432 + 20 and 0 <= xr <= 432 + 20
    # net top: predictor uses y < 192 (strict) where the real ball uses <= (physics.py:670 vs :412)
    a = oracle.expected_landing_x(216, 191, 0, 5)
    b = oracle.expected_landing_x(216, 192, 0, 5)
    assert a == 216 and b == 216
    # power-hit predictor ignores the current x velocity (physics.py:841-845)
    assert oracle.expected_landing_x_power_hit(1, 1, 100, 100, -7, 10) == \
        oracle.expected_landing_x_power_hit(1, 1, 100, 100, 13, 10)


SINGLE_AGENT_FIXTURES = ["single_agent_player_2", "single_agent_player_1_vs_computer"]


@pytest.mark.parametrize("name", SINGLE_AGENT_FIXTURES)
def test_oracle_matches_reference_single_agent_view(oracle, name):
    """Fixtures captured from the reference's ConvertSingleAgent (wrappers/convert_single_agent.py:16-28): the
    controlled side's actions and the opponent draws its `action_space(other).sample()` was fed.  The oracle stepped
    with both reproduces the env's state and what the wrapper returned for the controlled side."""
    d = load_golden(name)
    meta = d["meta"]
    me = 0 if meta["side"] == "player_1" else 1
    kw = meta["env_kwargs"]
    cfg = oracle.make_config(winning_score=kw.get("winning_score", 15), serve=kw.get("serve", "winner"),
                             is_player1_computer=kw.get("is_player1_computer", False),
                             is_player2_computer=kw.get("is_player2_computer", False),
                             seed=meta["seed"], env_id_base=meta["env_id_base"])
    env = oracle.OracleEnv(meta["lanes"], cfg)
    obs0 = env.reset()
    assert np.array_equal(obs0[me], d["obs_reset"])
    for t in range(meta["steps"]):
        own, opp = d["actions"][t].astype(np.int32), d["sampled"][t].astype(np.int32)
        # both streams are the build's Philox policy stream (word `me` / `1 - me` of the draw)
        assert np.array_equal(own, oracle.random_actions(meta["lanes"], meta["env_id_base"], meta["action_seed"], t)[me])
        assert np.array_equal(opp, oracle.random_actions(meta["lanes"], meta["env_id_base"], meta["opponent_seed"], t)[1 - me])
        a = [None, None]
        a[me], a[1 - me] = own, opp
        obs, rew, term = env.step(a[0], a[1])
        st = d["states"][t].astype(np.int32)
        st[43] = d["rng_counter"][t]
        assert np.array_equal(env.state, st), (name, t)
        assert np.array_equal(obs[me], d["obs"][t]) and np.array_equal(rew[me], d["rew"][t]), (name, t)
        assert np.array_equal(term, d["term"][t]) and np.array_equal(env.state[38:40].T, d["score"][t]), (name, t)
    assert d["term"].sum() > 10


_SANITIZED_REPLAY = r'''
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
from conftest import golden_state, load_golden, oracle_config_from_meta
from oracle import pz_oracle as po
assert po._SO.name == "libpz_oracle_asan.so"
for name in ("both_computer", "full_wrapper_stack", "serve_random"):
    d = load_golden(name)
    meta = d["meta"]
    env = po.OracleEnv(meta["lanes"], oracle_config_from_meta(meta), nthreads=2)
    env.reset()
    for t in range(meta["steps"]):
        env.step(d["actions"][t, 0], d["actions"][t, 1])
    assert np.array_equal(env.state, golden_state(d, meta["steps"] - 1)), name
big = po.OracleEnv(1000 + 37, po.make_config(is_player1_computer=True, is_player2_computer=True, winning_score=2,
                                              serve="random", seed=5, env_id_base=(1 << 40) + 3, episode_stats=1), nthreads=4)
big.reset(np.arange(1037) % 2)
big.rollout_random(9, 0, 400)
big.observe()
print("sanitized replay ok", int(big.state[43].sum()))
'''


def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """sanitizers on the CPU build (GPU ASan is not available on the pool).  The oracle compiled with
    -fsanitize=address,undefined replays fixtures of every branch family (both computer players, all wrappers,
    random serve) and a ragged multi-threaded batch; any report fails the run."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    repo = Path(__file__).resolve().parent.parent
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not Path(asan).exists():
        pytest.skip("libasan not available")
    script = tmp_path / "replay.py"
    script.write_text(_SANITIZED_REPLAY)
    env = dict(os.environ, PZ_ORACLE_SANITIZED="1", LD_PRELOAD=asan, OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, str(script), str(repo)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "sanitized replay ok" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


PLANTED_FIXTURES = ["planted_fast_balls_human", "planted_fast_balls_both_computer"]
# every attribute of players / ball / scores at random over its whole valid range (capture_planted_random)
PLANTED_RANDOM_FIXTURES = ["planted_random_states_human", "planted_random_states_both_computer",
                           "planted_random_states_p2_computer_random_serve"]


def replay_planted(d, make_stepper):
    """Planted-state fixtures (oracle/ref_capture.capture_planted): `make_stepper(meta, planted_state)` returns
    step(a1, a2) -> int32[44, lanes] state; every frame is compared with the reference's."""
    from oracle import pz_oracle as po

    meta = d["meta"]
    step = make_stepper(meta, d["planted"])
    for t in range(meta["frames"]):
        a1, a2 = po.random_actions(meta["lanes"], meta["env_id_base"], meta["action_seed"], meta["warm"] + t, 18)
        got = step(a1, a2)
        want = d["states"][t]
        if not np.array_equal(got, want):
            f, l = np.argwhere(got != want)[0]
            pytest.fail(f"frame {t} lane {l} (planted ball {meta['cases'][l]}) word {po.FIELD_NAMES[f]}: "
                        f"{got[f, l]} != reference {want[f, l]}")


@pytest.mark.parametrize("name", PLANTED_FIXTURES + PLANTED_RANDOM_FIXTURES)
def test_oracle_matches_reference_on_planted_fast_balls(name, oracle):
    """Ball states random play practically never reaches, written into the reference's own ball and stepped by it: balls
    faster than the court is high over the net top (bounced to a NEGATIVE y: the ceiling is tested before the net,
    physics.py:406-419), at the walls, at the ceiling, with and without computer players (their flight predictions
    start from those states)."""
    d = load_golden(name)
    if name in PLANTED_FIXTURES:
        assert int(d["states"][:, 27].min()) < 0  # the corner is in the fixture

    def make(meta, planted):
        kw = meta["env_kwargs"]
        env = oracle.OracleEnv(meta["lanes"], oracle.make_config(
            winning_score=kw.get("winning_score", 15), serve=kw.get("serve", "winner"),
            is_player1_computer=kw.get("is_player1_computer", False),
            is_player2_computer=kw.get("is_player2_computer", False), seed=meta["seed"], env_id_base=meta["env_id_base"]))
        env.state[:] = planted

        def step(a1, a2):
            env.step(a1, a2)
            return env.state
        return step

    replay_planted(d, make)


def test_normalized_observation_equals_the_float64_quotient_over_extended_ranges(oracle):
    """NormalizeObservation (normalize_observation.py:22,30) divides int64 arrays in float64; the build emits float32.
    For these small integers the float32 quotient IS the float32 rounding of the float64 one (a double rounding could
    only differ within 2^-54 of a float32 midpoint, and |v * 2^k - m * r| >= 1 keeps v / r away from every midpoint).
    Checked on the oracle's own arithmetic over every observation column and a value range far wider than its bounds
    (the ball's y reaches negative values, velocities exceed the observed bounds)."""
    from pikazoo_amd.env import OBS_HIGH, OBS_LOW

    values = np.arange(-700, 701, dtype=np.int32)
    n = len(values)
    env = oracle.OracleEnv(n, oracle.make_config(normalize_obs=True))
    # state rows behind the observation columns of player 1's row: player (x, y, yv, dive, lying, frame, delay, state
    # one-hot x5, key), the same for player 2, ball (x, y, px, py, ppx, ppy, xv, yv, power)
    player_rows = [0, 1, 2, 7, 8, 4, 6]
    ball_rows = [26, 27, 31, 32, 33, 34, 28, 29, 30]
    for col, row in ([(c, r) for c, r in enumerate(player_rows)] + [(13 + c, 13 + r) for c, r in enumerate(player_rows)] +
                     [(26 + c, r) for c, r in enumerate(ball_rows)]):
        env.state[:] = 0
        env.state[row] = values
        o1, _ = env.observe()
        want = ((values.astype(np.int64) - int(OBS_LOW[col])) / (np.int64(OBS_HIGH[col]) - np.int64(OBS_LOW[col]))).astype(np.float32)
        assert o1.dtype == np.float32 and np.array_equal(o1[:, col], want), (col, row)


def test_a_ball_keeps_its_landing_point_along_a_free_flight(tmp_path):
    """The statement behind the k-frame pair kernel's `known` landing points (pz_physics.hpp: pair_frame_head,
    flight_keeps_landing_point): for every ball B of the landing table's domain that the world step
    (physics.py:359-431) moves to W(B) without touching the ground, the predictor (:643-686) gives P(W(B)) == P(B) --
    unless B is over the net at y == 192 or without x velocity.  tests/flight_rule.c walks all 4.6e8 balls against the
    oracle's predictor (a few seconds on eight threads)."""
    import subprocess
    from pathlib import Path

    repo = Path(__file__).resolve().parent.parent
    exe = tmp_path / "flight_rule"
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", str(exe), str(repo / "tests" / "flight_rule.c"),
                           str(repo / "oracle" / "pz_oracle.c"), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "violations 0;" in out.stdout and "states 463826671" in out.stdout, out.stdout

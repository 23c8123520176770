"""GPU tests (``-m gpu``) of the two things ABI 10 added to the step path:

* the action vectors go into the launch in the caller's element type (``pz_config.action_format``: int32, int64 --
  torch's default integer dtype --, uint8, int16), range-checked on the FULL value: the reference raises ``IndexError``
  from its table lookup (pikazoo_env.py:182), a cast to int32 would wrap 2**32 + 3 to 3 before any check saw it;
* the landing point of a computer player's ball is predicted only on the frames that interrupt a flight
  (``pz_config.landing_fresh`` / ``landing_reuse=``), where the reference predicts on every frame (physics.py:314-315):
  the 44 state words must stay the reference's bit for bit, in every flight-table mode (both / power_hit / none), in both
  state formats, and whatever else writes the state in between.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden, golden_state

pytestmark = pytest.mark.gpu

DTYPES = [torch.int64, torch.int16, torch.uint8, torch.int32]
TABLE_MODES = [True, "power_hit", False]


def cpu(t):
    return t.detach().cpu().numpy()


def make(n, **kw):
    from pikazoo_amd import pikazoo_v0

    kw.setdefault("device", "cuda:0")
    return pikazoo_v0.env(num_envs=n, **kw)


def oracle_env(oracle, n, seed, base=0, **kw):
    return oracle.OracleEnv(n, oracle.make_config(seed=seed, env_id_base=base, **kw), nthreads=8)


# ------------------------------------------------------------------------------------------------
# 1. action tensors of every integer dtype
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,kw", [(4096, dict()),                                                  # config 2
                                  (4096 + 37, dict(is_player2_computer=True)),                      # config 3's kernel, ragged
                                  (2048, dict(is_player2_computer=True, flight_tables=False)),      # one wave + scout
                                  (2048, dict(is_player1_computer=True, state_format="packed"))])
def test_action_tensors_of_every_integer_dtype_match_the_oracle(dtype, n, kw, oracle):
    """int64 / int16 / uint8 / int32 tensors straight into the launch (no cast kernel): bit-exact against the oracle."""
    seed, aseed, steps = 5, 77, 300
    env = make(n, seed=seed, **kw)
    okw = {k: v for k, v in kw.items() if k.startswith("is_player")}
    ref = oracle_env(oracle, n, seed, **okw)
    env.reset(), ref.reset()
    fmt_seen = set()
    for t in range(steps):
        a1, a2 = oracle.random_actions(n, 0, aseed, t, 18)
        d1 = torch.as_tensor(a1, device=env.device).to(dtype)
        d2 = torch.as_tensor(a2, device=env.device).to(dtype)
        obs, rew, term, _, _ = env.step({"player_1": d1, "player_2": d2})
        robs, rrew, rterm = ref.step(a1, a2)
        fmt_seen.add(env._a1_seen[3])
        if t % 25 == 0 or t == steps - 1:
            assert np.array_equal(cpu(env.read_state()), ref.state), (dtype, t)
            assert np.array_equal(cpu(obs["player_2"]), robs[1]) and np.array_equal(cpu(rew["player_1"]), rrew[0])
    from pikazoo_amd import _native

    assert fmt_seen == {_native.ACTION_FORMATS[str(dtype).replace("torch.", "")]}, "the tensors were cast on the way in"
    env.check_actions()  # nothing out of range was counted


def test_other_integer_dtypes_are_widened_and_two_dtypes_meet_in_int64(oracle):
    n, seed = 512, 3
    env, ref = make(n, seed=seed), oracle_env(oracle, n, seed)
    env.reset(), ref.reset()
    for t, (dt1, dt2) in enumerate([(torch.int8, torch.int8), (torch.int64, torch.uint8), (torch.int16, torch.int32),
                                    (torch.int32, torch.int64)] * 10):
        a1, a2 = oracle.random_actions(n, 0, 9, t, 18)
        env.step({"player_1": torch.as_tensor(a1, device=env.device).to(dt1),
                  "player_2": torch.as_tensor(a2, device=env.device).to(dt2)})
        ref.step(a1, a2)
    assert np.array_equal(cpu(env.read_state()), ref.state)
    env.check_actions()
    with pytest.raises(TypeError):
        env.step({"player_1": torch.zeros(n, device=env.device), "player_2": torch.zeros(n, dtype=torch.int64, device=env.device)})


@pytest.mark.parametrize("dtype,bad", [(torch.int64, 2 ** 32 + 3), (torch.int64, -1), (torch.int64, 18),
                                       (torch.int64, -(2 ** 40)), (torch.int64, 2 ** 31), (torch.int16, -1),
                                       (torch.int16, 18 + 256), (torch.uint8, 200), (torch.int32, -1), (torch.int32, 18)])
@pytest.mark.parametrize("kw", [dict(), dict(is_player2_computer=True), dict(is_player2_computer=True, flight_tables=False),
                                dict(num_envs=393216 + 64)])
def test_out_of_range_actions_of_every_dtype_raise_index_error(dtype, bad, kw):
    """The reference raises IndexError from `action_key_map[actions[agent]]` (pikazoo_env.py:182).  The launch checks the
    FULL value: an int64 2**32 + 3 is a fault, not action 3.  Strict mode (validate_every=1): from the same call."""
    kw = dict(kw)
    n = kw.pop("num_envs", 1000)
    for agent in ("player_1", "player_2"):
        env = make(n, seed=1, validate_every=1, **kw)
        env.reset()
        good = torch.full((n,), 3, dtype=dtype, device=env.device)
        env.step({"player_1": good, "player_2": good})  # in range: nothing raised
        one_bad = good.clone()
        one_bad[n - 1 if agent == "player_1" else 0] = bad  # (the batch's last game: its range check must reach it)
        acts = {"player_1": good, "player_2": good}
        acts[agent] = one_bad
        with pytest.raises(IndexError):
            env.step(acts)
        env.step({"player_1": good, "player_2": good})  # the counter was reset with the error
    # the default mode polls: the error comes from check_actions() at the latest
    env = make(n, seed=1, **kw)
    env.reset()
    env.step({"player_1": one_bad, "player_2": good})
    with pytest.raises(IndexError):
        env.check_actions()


def test_simplified_actions_are_checked_against_thirteen_in_every_dtype():
    from pikazoo_amd.wrappers import SimplifyAction

    for dtype in DTYPES:
        env = SimplifyAction(make(256, seed=2, validate_every=1))
        env.reset()
        ok = torch.full((256,), 12, dtype=dtype, device="cuda:0")
        env.step({"player_1": ok, "player_2": ok})
        with pytest.raises(IndexError):
            env.step({"player_1": ok + 1, "player_2": ok})


def test_step_many_takes_an_int64_tape_without_wrapping(oracle):
    """pz_step_many parks its tape from int32 rows: an int64 tape is narrowed with saturation first (a plain cast would
    wrap 2**32 + 3 to 3 in front of the launch's range check), the smaller integer types are widened."""
    n, k, seed = 1024, 48, 6
    tape = np.stack([np.stack(oracle.random_actions(n, 0, 21, t, 18)) for t in range(k)])  # [k, 2, n]
    states = []
    for dtype in (torch.int32, torch.int64, torch.int16, torch.uint8):
        env = make(n, seed=seed, validate_every=1)
        env.reset()
        env.step_many(torch.as_tensor(tape, device=env.device).to(dtype))
        states.append(env.read_state())
    assert all(torch.equal(states[0], s) for s in states[1:])
    ref = oracle_env(oracle, n, seed)
    ref.reset()
    for t in range(k):
        ref.step(tape[t, 0], tape[t, 1])
    assert np.array_equal(cpu(states[0]), ref.state)
    for bad in (2 ** 32 + 3, -1, 18):
        env = make(n, seed=seed, validate_every=1)
        env.reset()
        t64 = torch.as_tensor(tape, device=env.device).to(torch.int64)
        t64[k - 1, 1, n - 1] = bad
        with pytest.raises(IndexError):
            env.step_many(t64)


def test_c_abi_counts_on_the_full_value_and_refuses_unknown_formats():
    """Straight through the C ABI: pz_step with cfg.action_format, a caller-owned fault counter."""
    from pikazoo_amd import _native

    lib = _native.load()
    n, dev = 640, torch.device("cuda:0")
    cfg = _native.PzConfig()
    cfg.winning_score, cfg.auto_reset, cfg.seed = 15, 1, 4
    faults = torch.zeros(1, dtype=torch.int64, device=dev)
    cfg.action_faults = faults.data_ptr()
    state = torch.zeros((44, n), dtype=torch.int32, device=dev)
    obs = [torch.zeros((n, 35), dtype=torch.int32, device=dev) for _ in range(2)]
    rew = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
    term = torch.zeros(n, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.pz_init(state.data_ptr(), n, n, C.byref(cfg), s) == 0

    def step(a1, a2):
        return lib.pz_step(state.data_ptr(), n, n, C.byref(cfg), a1.data_ptr(), a2.data_ptr(), obs[0].data_ptr(),
                           obs[1].data_ptr(), rew[0].data_ptr(), rew[1].data_ptr(), term.data_ptr(), None, None, s)

    for name, dtype, bads in (("int64", torch.int64, [2 ** 32 + 3, -1, 2 ** 63 - 1, 18]), ("int16", torch.int16, [-1, 300]),
                              ("uint8", torch.uint8, [18, 255]), ("int32", torch.int32, [-1, 18, 2 ** 31 - 1])):
        cfg.action_format = _native.ACTION_FORMATS[name]
        good = torch.full((n,), 17, dtype=dtype, device=dev)
        assert step(good, good) == 0
        assert int(faults.item()) == 0
        for k, bad in enumerate(bads):
            a = good.clone()
            a[5 + k] = bad
            assert step(good, a) == 0
        assert int(faults.item()) == len(bads), (name, int(faults.item()))
        faults.zero_()
    cfg.action_format = 4
    assert step(good, good) == -3  # PZ_E_CONFIG
    cfg.action_format = _native.ACTION_FORMATS["int64"]
    tape = torch.zeros((4, 2, n), dtype=torch.int64, device=dev)
    k_obs = [torch.zeros((4, n, 35), dtype=torch.int32, device=dev) for _ in range(2)]
    k_rew = [torch.zeros((4, n), dtype=torch.int32, device=dev) for _ in range(2)]
    rc = lib.pz_step_many(state.data_ptr(), n, n, C.byref(cfg), tape.data_ptr(), 4, k_obs[0].data_ptr(), k_obs[1].data_ptr(),
                          k_rew[0].data_ptr(), k_rew[1].data_ptr(), torch.zeros((4, n), dtype=torch.uint8, device=dev).data_ptr(),
                          None, None, None, s)
    assert rc == -3  # the tape is int32 (include/pikazoo_hip.h: how a caller brings another element type)


# ------------------------------------------------------------------------------------------------
# 2. the landing point along a free flight (pz_config.landing_fresh)
# ------------------------------------------------------------------------------------------------
AI_FIXTURES = ["cfg3_p2_computer", "p1_computer", "both_computer", "full_wrapper_stack"]


@pytest.mark.parametrize("reuse", [True, False])
@pytest.mark.parametrize("fmt", ["int32", "packed"])
@pytest.mark.parametrize("tables", TABLE_MODES)
@pytest.mark.parametrize("name", AI_FIXTURES)
def test_reference_trajectories_with_and_without_landing_reuse(name, tables, fmt, reuse):
    """The reference's own trajectories (tests/golden) in every flight-table mode, with the landing point predicted on
    every frame like the reference and only where a flight is interrupted.  The state is READ (a copy) on every frame and
    never written: with `reuse` the notes stay up from frame to frame, which is the path under test."""
    from conftest import apply_product_wrappers
    from pikazoo_amd import pikazoo_v0

    d = load_golden(name)
    meta = d["meta"]
    env = apply_product_wrappers(pikazoo_v0.env(**meta["env_kwargs"], num_envs=meta["lanes"], seed=meta["seed"],
                                                env_id_base=meta["env_id_base"], device="cuda:0", flight_tables=tables,
                                                state_format=fmt, landing_reuse=reuse), meta["wrappers"] or {})
    raw = env.unwrapped
    assert raw.landing_reuse is reuse and (raw._fresh is not None) == reuse
    assert raw.flight_tables == {True: "both", "power_hit": "power_hit", False: "none"}[tables]
    env.reset()
    assert np.array_equal(cpu(raw.read_state()), d["state0"])
    acts = torch.as_tensor(d["actions"].astype(np.int32), device=raw.device)
    fresh_seen = 0
    for t in range(meta["steps"]):
        obs, rew, term, _, _ = env.step({"player_1": acts[t, 0], "player_2": acts[t, 1]})
        st = cpu(raw.read_state())
        want = golden_state(d, t)
        if not np.array_equal(st, want):
            f, l = np.argwhere(st != want)[0]
            pytest.fail(f"{name} tables={tables} {fmt} reuse={reuse}: step {t} lane {l} word {f}: {st[f, l]} != {want[f, l]}")
        if reuse:
            fresh_seen += int(raw._fresh[:raw.num_envs].sum().item())
    if raw.obs_dtype == torch.int32:
        assert np.array_equal(cpu(obs["player_1"]), d["obs"][-1, 0].astype(np.int32))
    if reuse:  # the notes were up on most frames: the path under test did run
        assert fresh_seen > 0.8 * meta["steps"] * meta["lanes"], fresh_seen


@pytest.mark.parametrize("tables", TABLE_MODES)
@pytest.mark.parametrize("kw", [dict(is_player2_computer=True), dict(is_player1_computer=True, is_player2_computer=True),
                                dict(is_player1_computer=True, winning_score=2, auto_reset=False)])
def test_a_fresh_note_means_the_stored_point_is_the_stored_balls_prediction(kw, tables, oracle):
    """The invariant itself, against the oracle's predictor: wherever a game's note is up, expected_landing_point_x of
    the stored state equals calculate_expected_landing_point_x_for (physics.py:643-686) of the stored ball."""
    n, seed = 2048, 8
    env = make(n, seed=seed, flight_tables=tables, landing_reuse=True, **kw)
    env.reset()
    checked = 0
    for t in range(400):
        env.step(env.random_actions(31, t))
        if t % 40 == 39:
            st, fresh = cpu(env.read_state()), cpu(env._fresh[:n]).astype(bool)
            for l in np.flatnonzero(fresh)[::7]:
                assert st[36, l] == oracle.expected_landing_x(int(st[26, l]), int(st[27, l]), int(st[28, l]), int(st[29, l])), (t, l)
                checked += 1
    assert checked > 500


@pytest.mark.parametrize("fmt", ["int32", "packed"])
@pytest.mark.parametrize("tables", TABLE_MODES)
def test_writes_to_the_state_between_steps_drop_the_notes(tables, fmt, oracle):
    """Anything but a step launch that writes the state makes the stored landing points stale: set_state, a masked
    reset, a torch operation on `env.state` or on a view kept from earlier (the env watches the tensor's version
    counter).  Each time the next frames must equal the oracle's, which predicts on every frame."""
    n, seed, aseed = 1024, 12, 99
    env = make(n, seed=seed, is_player2_computer=True, flight_tables=tables, state_format=fmt, winning_score=3,
               landing_reuse=True)
    ref = oracle_env(oracle, n, seed, is_player2_computer=True, winning_score=3)
    env.reset(), ref.reset()
    kept_view = env.state if fmt == "int32" else None
    t = 0

    def run(frames):
        nonlocal t
        for _ in range(frames):
            a1, a2 = oracle.random_actions(n, 0, aseed, t, 18)
            env.step({"player_1": torch.as_tensor(a1, device=env.device), "player_2": torch.as_tensor(a2, device=env.device)})
            ref.step(a1, a2)
            t += 1
        assert np.array_equal(cpu(env.read_state()), ref.state), t

    run(60)
    assert int(env._fresh.sum().item()) > n // 2
    rng = np.random.default_rng(0)
    # (1) the ball moved by hand, through set_state
    st = ref.state.copy()
    st[26] = rng.integers(20, 433, n)
    st[27] = rng.integers(0, 200, n)
    st[28] = rng.integers(-10, 11, n)
    st[29] = rng.integers(-30, 31, n)
    ref.state[:] = st
    env.set_state(torch.as_tensor(st, device=env.device))
    assert int(env._fresh.sum().item()) == 0
    run(40)
    # (2) a masked reset: only the reset games lose their note
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    before = cpu(env._fresh[:n]).copy()
    env.reset(mask=torch.as_tensor(mask, device=env.device)), ref.reset(mask)
    after = cpu(env._fresh[:n])
    assert not after[mask == 1].any() and np.array_equal(after[mask == 0], before[mask == 0])
    run(40)
    if fmt == "int32":
        # (3) a torch write through the live tensor, and (4) through a view kept from before the steps
        st = ref.state.copy()
        st[26] = rng.integers(20, 433, n)
        st[29] = rng.integers(-20, 21, n)
        ref.state[:] = st
        env.state[26] = torch.as_tensor(st[26], device=env.device)
        env.state[29].copy_(torch.as_tensor(st[29], device=env.device))
        run(40)
        st = ref.state.copy()
        st[27] = rng.integers(0, 150, n)
        ref.state[:] = st
        kept_view[27] = torch.as_tensor(st[27], device=env.device)
        run(40)
        # (5) only the stored landing point scribbled over: the reference never reads it (it predicts afresh)
        kept_view[36] = 7
        ref.state[36] = 7
        run(40)
    # (6) a writer the env cannot see says so itself
    env.invalidate_landing()
    assert int(env._fresh.sum().item()) == 0
    run(20)


@pytest.mark.parametrize("tables", [True, "power_hit"])
def test_k_frame_launches_keep_and_use_the_notes(tables, oracle):
    """pz_step_random / pz_rollout_random / pz_step_many read the notes for their first frame and leave them for the next
    launch, on both sides of the kernel switch at 393 216 games and mixed with single frames."""
    for n in (4096, 393216 + 128):
        seed = 17
        env = make(n, seed=seed, is_player2_computer=True, flight_tables=tables, landing_reuse=True)
        lanes = min(n, 2048)
        ref = oracle_env(oracle, lanes, seed, is_player2_computer=True)
        env.reset(), ref.reset()
        t = 0
        for launch in range(6):
            if launch % 3 == 0:
                env.rollout_random(55, 16, t0=t)
            elif launch % 3 == 1:
                env.step_random(55, t0=t, k=16)
            else:
                tape = torch.stack([torch.stack([env.random_actions(55, t + j)[a] for a in env.possible_agents])
                                    for j in range(16)])
                env.step_many(tape)
            ref.rollout_random(55, t, 16)
            t += 16
            assert int(env._fresh[:n].sum().item()) > n // 2
            env.step(env.random_actions(55, t))
            ref.rollout_random(55, t, 1)
            t += 1
            assert np.array_equal(cpu(env.read_state()[:, :lanes]), ref.state), (n, launch)


def test_landing_reuse_is_off_without_a_computer_player_and_on_request():
    from conftest import EVERY_FRAME

    if EVERY_FRAME:
        pytest.skip("PZ_TEST_LANDING_REUSE=0 changes the default this test is about")
    hh = make(64, seed=1)
    assert hh.landing_reuse is False and hh._fresh is None and hh._cfg.landing_fresh is None
    hh.invalidate_landing()  # a no-op
    off = make(64, seed=1, is_player2_computer=True, landing_reuse=False)
    assert off.landing_reuse is False and off._cfg.landing_fresh is None
    on = make(64, seed=1, is_player2_computer=True)
    assert on.landing_reuse is True and on._cfg.landing_fresh == on._fresh.data_ptr()
    with pytest.raises(ValueError):
        make(64, flight_tables="landing")


def test_flight_table_modes_cost_what_the_readme_says():
    from pikazoo_amd import env as E

    # landing uint16[193][23][253][413] (+ 2 bytes of padding), power_hit uint16[65][192][413][8]
    assert E.flight_table_bytes(True) == E.flight_table_bytes("both") == 927_653_344 + 82_467_840
    assert E.flight_table_bytes("power_hit") == 82_467_840 and E.flight_table_bytes(False) == E.flight_table_bytes("none") == 0
    e = make(64, seed=1, is_player2_computer=True, flight_tables="power_hit")
    assert e._tables[0].landing is None and e._tables[0].power_hit is not None

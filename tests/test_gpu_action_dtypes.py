"""GPU tests (``-m gpu``) of what ABI 10 added to the step path: the action vectors go into the launch in the caller's
element type (``pz_config.action_format``: int32, int64 -- torch's default integer dtype --, uint8, int16), range-checked
on the FULL value: the reference raises ``IndexError`` from its table lookup (pikazoo_env.py:182), a cast to int32 would
wrap 2**32 + 3 to 3 before any check saw it.  (The flight-table mode ``"power_hit"`` of the same ABI runs through the
parity parametrisations of test_gpu_parity.py / test_gpu_packed.py.)
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.int64, torch.int16, torch.uint8, torch.int32]


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
# 2. the flight-table modes
# ------------------------------------------------------------------------------------------------
def test_flight_table_modes_cost_what_the_readme_says():
    from pikazoo_amd import env as E

    # landing uint16[193][23][253][413] (+ 2 bytes of padding), power_hit uint16[65][192][413][8]
    assert E.flight_table_bytes(True) == E.flight_table_bytes("both") == 927_653_344 + 82_467_840
    assert E.flight_table_bytes("power_hit") == 82_467_840 and E.flight_table_bytes(False) == E.flight_table_bytes("none") == 0
    e = make(64, seed=1, is_player2_computer=True, flight_tables="power_hit")
    assert e._tables[0].landing is None and e._tables[0].power_hit is not None and e.flight_tables == "power_hit"
    assert make(64, seed=1, is_player2_computer=True).flight_tables == "both" and make(64, seed=1).flight_tables == "none"
    with pytest.raises(ValueError):
        make(64, flight_tables="landing")

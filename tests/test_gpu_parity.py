"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the C ABI
(libpikazoo_hip.so via pikazoo_amd), against

* the golden trajectories captured from the unmodified reference (tests/golden/*.npz),
* the CPU oracle (oracle/pz_oracle.c) on seeded random batches at sizes it finishes in seconds,
* size-independent properties at BASELINE.json's full sizes (65 536 and 524 288 lanes).

Bar: bit-exact for every integer (44 state words, 2x35 observations, int rewards, terminations);
float32 rewards of the fused RewardByBallPosition are bit-exact against the oracle's fp32 add and
within 1e-6 of the reference's float64 sum.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import DIGEST_FIXTURES, FULL_FIXTURES, UNFUSED_FIXTURES, golden_state, load_golden, oracle_config_from_meta

pytestmark = pytest.mark.gpu


def make_env(meta=None, **over):
    from conftest import apply_product_wrappers
    from pikazoo_amd import pikazoo_v0

    kw = dict(meta["env_kwargs"]) if meta else {}
    wr = (meta["wrappers"] or {}) if meta else {}
    if meta:
        kw.update(num_envs=meta["lanes"], seed=meta["seed"], env_id_base=meta["env_id_base"])
    wr = dict(wr)
    wr.update(over.pop("wrappers", {}))
    kw.update(over)
    kw.setdefault("device", "cuda:0")
    kw.setdefault("validate_actions", False)
    return apply_product_wrappers(pikazoo_v0.env(**kw), wr)


def cpu(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------
# 1. golden trajectories of the reference
# ------------------------------------------------------------------------------------------------
AI_FIXTURES = ["cfg3_p2_computer", "p1_computer", "both_computer", "full_wrapper_stack"]


@pytest.mark.parametrize("name,tables,fmt", [(n, True, "int32") for n in FULL_FIXTURES] +
                         [(n, False, "int32") for n in AI_FIXTURES] + [(n, True, "packed") for n in FULL_FIXTURES] +
                         [(n, False, "packed") for n in AI_FIXTURES] +
                         [(n, "power_hit", f) for n in AI_FIXTURES for f in ("int32", "packed")])
def test_hip_matches_reference_trajectory(name, tables, fmt):
    """`tables`: the computer player's flight predictions come from the HBM look-up tables (default), from the 82 MB
    power-hit table alone ("power_hit": the landing point predicted in the kernel), or are all
    iterated in the kernel (the scout-wave launch); fixtures without a computer player never use them.
    `fmt`: the state lives in HBM as int32 columns or in the packed format (36 bytes per game); `raw.state` is the
    same int32[44, lanes] either way."""
    d = load_golden(name)
    meta = d["meta"]
    env = make_env(meta, flight_tables=tables, state_format=fmt)
    raw = env.unwrapped
    T, L = meta["steps"], meta["lanes"]
    assert np.array_equal(cpu(raw.state), d["state_ctor"])
    obs, infos = env.reset(seed=123)  # seed ignored like the reference
    assert np.array_equal(cpu(raw.state), d["state0"])
    odt = np.float32 if raw.obs_dtype == torch.float32 else np.int32
    assert np.array_equal(cpu(obs["player_1"]), d["obs_reset"][:, 0].astype(odt))
    assert np.array_equal(cpu(obs["player_2"]), d["obs_reset"][:, 1].astype(odt))

    dev = raw.device
    acts = torch.as_tensor(d["actions"].astype(np.int32), device=dev)  # [T, 2, L]
    h_state = torch.empty((T, 44, L), dtype=torch.int32, device=dev)
    h_obs = torch.empty((T, 2, L, 35), dtype=raw.obs_dtype, device=dev)
    has_stats = raw.episode_lengths is not None
    h_epr = torch.zeros((T, 2, L), dtype=torch.float64, device=dev)
    h_epl = torch.zeros((T, L), dtype=torch.int32, device=dev)
    h_rew = torch.empty((T, 2, L), dtype=raw.reward_dtype, device=dev)
    h_term = torch.empty((T, L), dtype=torch.bool, device=dev)
    for t in range(T):
        obs, rew, term, trunc, infos = env.step({"player_1": acts[t, 0], "player_2": acts[t, 1]})
        h_state[t].copy_(raw.state)
        h_obs[t, 0].copy_(obs["player_1"])
        h_obs[t, 1].copy_(obs["player_2"])
        h_rew[t, 0].copy_(rew["player_1"])
        h_rew[t, 1].copy_(rew["player_2"])
        h_term[t].copy_(term["player_1"])
        assert term["player_1"] is term["player_2"]
        if has_stats:
            h_epr[t, 0].copy_(infos["player_1"]["episode"]["r"])
            h_epr[t, 1].copy_(infos["player_2"]["episode"]["r"])
            h_epl[t].copy_(infos["player_1"]["episode"]["l"])
    h_state, h_obs, h_rew, h_term = cpu(h_state), cpu(h_obs), cpu(h_rew), cpu(h_term)
    assert not cpu(trunc["player_1"]).any()
    for t in range(T):
        st = golden_state(d, t)
        if not np.array_equal(h_state[t], st):
            f, l = np.argwhere(h_state[t] != st)[0]
            pytest.fail(f"{name}: step {t} lane {l} word {f}: hip {h_state[t][f, l]} != reference {st[f, l]}")
    if raw.obs_dtype == torch.float32:  # NormalizeObservation: float32 rounding of the reference's double
        assert np.array_equal(h_obs, d["obs"].astype(np.float32))
    else:
        assert np.array_equal(h_obs, d["obs"].astype(np.int32))
    if has_stats:  # infos[agent]["episode"] = {"r", "l"} on terminal frames (record_episode_statistics.py:34-39)
        done = d["ep_l"] >= 0
        assert np.array_equal(done, d["term"].astype(bool))
        assert np.array_equal(cpu(h_epl)[done], d["ep_l"][done])
        for i in range(2):
            # float64 sums of float32 rewards vs the reference's float64 sums of float64 rewards
            np.testing.assert_allclose(cpu(h_epr)[:, i][done], d["ep_r"][:, i][done], rtol=0, atol=2e-6)
    assert np.array_equal(h_term.astype(np.uint8), d["term"])
    if raw.reward_dtype == torch.float32:
        np.testing.assert_allclose(h_rew, d["rew"], rtol=0, atol=1e-6)
    else:
        assert np.array_equal(h_rew, d["rew"].astype(np.int32))


@pytest.mark.parametrize("fmt", ["int32", "packed"])
@pytest.mark.parametrize("name", UNFUSED_FIXTURES)
def test_wrapper_stacks_the_kernel_cannot_fuse_match_the_reference(name, fmt):
    """The reference composes its wrappers in any order.  Where an order cannot be a kernel branch -- RewardByBallPosition
    above NormalizeObservation (it then reads the normalized coordinates), RecordEpisodeStatistics between two reward
    wrappers, a second RewardByBallPosition / RewardInNormalState / NormalizeObservation -- the wrapper class applies
    itself to the step's outputs with torch operations, outside everything that is fused: same trajectories as the
    reference's own stack (tests/golden/unfused_*.npz), nothing raises NotImplementedError."""
    d = load_golden(name)
    meta = d["meta"]
    env = make_env(meta, state_format=fmt)
    raw = env.unwrapped
    assert raw._unfused, "this stack is expected to leave something outside the kernel"
    T, L = meta["steps"], meta["lanes"]
    assert np.array_equal(cpu(raw.state), d["state_ctor"])
    obs, infos = env.reset()
    assert np.array_equal(cpu(raw.state), d["state0"])
    for i, ag in enumerate(raw.possible_agents):
        np.testing.assert_allclose(cpu(obs[ag]).astype(np.float64), d["obs_reset"][:, i], rtol=0, atol=1e-6)
    acts = torch.as_tensor(d["actions"].astype(np.int32), device=raw.device)
    has_stats = "ep_l" in d
    for t in range(T):
        obs, rew, term, trunc, infos = env.step({"player_1": acts[t, 0], "player_2": acts[t, 1]})
        assert np.array_equal(cpu(raw.state), golden_state(d, t)), (name, t)
        assert np.array_equal(cpu(term["player_1"]).astype(np.uint8), d["term"][t])
        for i, ag in enumerate(raw.possible_agents):
            # float32 here, float64 in the reference: a normalized observation is the float32 rounding of its quotient,
            # a reward within 1e-6
            assert np.array_equal(cpu(obs[ag]).astype(np.float32), d["obs"][t, i].astype(np.float32)), (name, t)
            np.testing.assert_allclose(cpu(rew[ag]).astype(np.float64), d["rew"][t, i], rtol=0, atol=1e-6)
        if has_stats:
            done = d["ep_l"][t] >= 0
            if done.any():
                assert np.array_equal(cpu(infos["player_1"]["episode"]["l"])[done], d["ep_l"][t][done])
                for i, ag in enumerate(raw.possible_agents):
                    # float64 sums of float32 rewards (each within half a float32 ulp of the reference's float64 reward:
                    # 6e-8 relative -- the integer table of the doubled stack pays up to 8 per step, its returns reach 160)
                    np.testing.assert_allclose(cpu(infos[ag]["episode"]["r"])[done], d["ep_r"][t][i][done], rtol=1e-7, atol=2e-6)
    # the k-frame launches return the kernel's own outputs: refused while part of the stack runs outside it
    with pytest.raises(RuntimeError):
        raw.rollout_random(1, 4)
    with pytest.raises(RuntimeError):
        env.step_random(1)


@pytest.mark.parametrize("name", DIGEST_FIXTURES)
def test_hip_matches_reference_long_run_digests(name, oracle):
    """20 000-step runs of the reference, one digest per 500 steps, replayed with the on-device
    random policy (k frames per launch)."""
    d = load_golden(name)
    meta = d["meta"]
    env = make_env(meta)
    raw = env.unwrapped
    env.reset()
    assert np.array_equal(cpu(raw.state), d["state0"])
    every = meta["digest_every"]
    for k, dg in enumerate(d["digests"]):
        env.step_random(meta["action_seed"], t0=k * every, k=every)
        assert oracle.digest(cpu(raw.state)) == int(dg), f"{name}: after {(k + 1) * every} steps"
    assert np.array_equal(cpu(raw.state), d["final_state"])
    assert raw.episodes_done == meta["episodes"]


# ------------------------------------------------------------------------------------------------
# 2. HIP vs CPU oracle on seeded random batches
# ------------------------------------------------------------------------------------------------
CASES = {
    "cfg2_4096_random_random": dict(n=4096, steps=1500, kw=dict(winning_score=15, serve="winner")),
    "cfg3_p2_computer": dict(n=16384, steps=600, kw=dict(is_player2_computer=True)),
    "p1_computer_alternate": dict(n=4096, steps=600, kw=dict(is_player1_computer=True, serve="alternate")),
    "both_computer_random_serve_ws2": dict(n=4096, steps=800, kw=dict(is_player1_computer=True,
                                                                     is_player2_computer=True, serve="random",
                                                                     winning_score=2)),
    "cfg5_wrappers": dict(n=16384, steps=600, kw=dict(winning_score=15),
                          wr=dict(simplify_action=True,
                                  additional_reward=(0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01))),
    "all_wrappers_fused": dict(n=8192, steps=500, kw=dict(winning_score=2, is_player2_computer=True),
                               wr=dict(stack=[["SimplifyAction", {}], ["RewardInNormalState", dict(reward=-0.002)],
                                              ["RewardByBallPosition",
                                               dict(additional_reward=[0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01])],
                                              ["NormalizeObservation", {}], ["RecordEpisodeStatistics", {}]])),
    "stats_inside_normal_outside": dict(n=4096, steps=400, kw=dict(winning_score=1),
                                        wr=dict(stack=[["RecordEpisodeStatistics", {}],
                                                       ["RewardByBallPosition",
                                                        dict(additional_reward=[1, -2, 3, -4, 5, -6, 7, -8])],
                                                       ["RewardInNormalState", dict(reward=0.125)]])),
    "ws1_no_auto_reset": dict(n=2048, steps=500, kw=dict(winning_score=1, auto_reset=False)),
    "ragged_batch": dict(n=1000 + 37, steps=300, kw=dict(is_player2_computer=True, winning_score=3)),
}


def _has_computer(kw):
    return bool(kw.get("is_player1_computer") or kw.get("is_player2_computer"))


@pytest.mark.parametrize("case,tables,fmt", [(c, True, "int32") for c in CASES] +
                         [(c, False, "int32") for c in CASES if _has_computer(CASES[c]["kw"])] +
                         [(c, True, "packed") for c in CASES] +
                         [(c, False, "packed") for c in ("cfg3_p2_computer", "both_computer_random_serve_ws2")] +
                         [(c, "power_hit", f) for c in CASES if _has_computer(CASES[c]["kw"]) for f in ("int32", "packed")])
def test_hip_matches_oracle_random_batches(case, tables, fmt, oracle):
    c = CASES[case]
    n, steps, kw, wr = c["n"], c["steps"], dict(c["kw"]), c.get("wr", {})
    seed, base, aseed = 99, 12345, 4242
    env = make_env(num_envs=n, seed=seed, env_id_base=base, wrappers=wr, flight_tables=tables, state_format=fmt, **kw)
    assert (env.unwrapped._tables_ref is not None) == (bool(tables) and _has_computer(kw))
    raw = env.unwrapped
    from oracle.ref_capture import fused_options
    ocfg = oracle.make_config(
        winning_score=kw.get("winning_score", 15), serve=kw.get("serve", "winner"),
        is_player1_computer=kw.get("is_player1_computer", False),
        is_player2_computer=kw.get("is_player2_computer", False),
        auto_reset=kw.get("auto_reset", True), seed=seed, env_id_base=base, **fused_options(wr))
    ref = oracle.OracleEnv(n, ocfg, nthreads=8)
    assert np.array_equal(cpu(raw.state), ref.state)
    obs, _ = env.reset()
    r1, r2 = ref.reset()
    assert np.array_equal(cpu(obs["player_1"]), r1) and np.array_equal(cpu(obs["player_2"]), r2)
    n_act = raw.n_actions
    for t in range(steps):
        acts = env.random_actions(aseed, t)
        a1, a2 = oracle.random_actions(n, base, aseed, t, n_act)
        if t % 50 == 0:
            assert np.array_equal(cpu(acts["player_1"]), a1) and np.array_equal(cpu(acts["player_2"]), a2)
        obs, rew, term, trunc, infos = env.step(acts)
        robs, rrew, rterm = ref.step(a1, a2)
        if t % 10 == 0 or t == steps - 1:
            hs = cpu(raw.state)
            if not np.array_equal(hs, ref.state):
                f, l = np.argwhere(hs != ref.state)[0]
                pytest.fail(f"{case}: step {t} lane {l} word {oracle.FIELD_NAMES[f]}: "
                            f"hip {hs[f, l]} != oracle {ref.state[f, l]}")
            assert np.array_equal(cpu(obs["player_1"]), robs[0]), (case, t)
            assert np.array_equal(cpu(obs["player_2"]), robs[1]), (case, t)
            assert np.array_equal(cpu(rew["player_1"]), rrew[0]), (case, t)  # bit-exact, fp32 included
            assert np.array_equal(cpu(rew["player_2"]), rrew[1]), (case, t)
            assert np.array_equal(cpu(term["player_1"]).astype(np.uint8), rterm), (case, t)
            assert np.array_equal(cpu(infos["player_1"]["score"]), ref.state[38:40].T)
            if ref.stats is not None:  # bit-exact, float64 sums included (same order of the same adds)
                assert np.array_equal(cpu(raw.episode_returns), ref.episode_returns), (case, t)
                assert raw.episode_returns.dtype == torch.float64
                assert np.array_equal(cpu(infos["player_2"]["episode"]["l"]), ref.episode_lengths)
    assert ref.state[43].min() >= 4  # draws happened


def test_step_random_equals_step_with_policy_stream(oracle):
    n, aseed = 8192, 31337
    kw = dict(num_envs=n, seed=5, env_id_base=777, is_player2_computer=True, winning_score=4)
    e1, e2, e3 = make_env(**kw), make_env(**kw), make_env(**kw)
    for e in (e1, e2, e3):
        e.reset()
    total_term = 0
    for t in range(240):
        o1 = e1.step(e1.random_actions(aseed, t))
        o2 = e2.step_random(aseed)  # t0 defaults to steps_done
        total_term += int(o1[2]["player_1"].sum().item())
        if t % 40 == 0 or t == 239:
            assert torch.equal(e1.state, e2.state)
            for a in ("player_1", "player_2"):
                assert torch.equal(o1[0][a], o2[0][a]) and torch.equal(o1[1][a], o2[1][a])
            assert torch.equal(o1[2]["player_1"], o2[2]["player_1"])
    # k frames in one launch == k launches; outputs are the last frame's
    for k0 in range(0, 240, 60):
        o3 = e3.step_random(aseed, t0=k0, k=60)
    assert torch.equal(e1.state, e3.state)
    assert torch.equal(o1[0]["player_1"], o3[0]["player_1"]) and torch.equal(o1[1]["player_2"], o3[1]["player_2"])
    assert e2.episodes_done == total_term == e3.episodes_done and total_term > 0
    ocfg = oracle.make_config(winning_score=4, is_player2_computer=True, seed=5, env_id_base=777)
    ref = oracle.OracleEnv(n, ocfg, nthreads=8)
    ref.reset()
    assert ref.rollout_random(aseed, 0, 240) == total_term
    assert np.array_equal(cpu(e3.state), ref.state)


# ------------------------------------------------------------------------------------------------
# 3. full-size properties (no oracle): 65 536 and 524 288 lanes
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,kw", [
    (65536, dict(is_player2_computer=True)),
    (524288, dict()),
])
def test_full_size_properties(n, kw, oracle):
    from pikazoo_amd.env import OBS_HIGH, OBS_LOW

    env = make_env(num_envs=n, seed=2024, env_id_base=0, winning_score=3, **kw)
    raw = env.unwrapped
    env.reset()
    lo = torch.as_tensor(OBS_LOW, device=raw.device)
    hi = torch.as_tensor(OBS_HIGH, device=raw.device)
    terms = 0
    for t in range(400):
        obs, rew, term, trunc, infos = env.step_random(7)
        if t % 25 == 0 or t == 399:
            o1, o2 = obs["player_1"], obs["player_2"]
            # the reference's own property test (tests/env/test_env.py:18-21) on every lane
            assert torch.equal(o1[:, 0:13], o2[:, 13:26]) and torch.equal(o1[:, 13:26], o2[:, 0:13])
            assert torch.equal(o1[:, 26:], o2[:, 26:])
            # observation_space bounds (pikazoo_env.py:485-562).  Column 33 (ball y velocity) is exempt:
            # the reference's +-124 is "the minimum and maximum values I observed" (README.md:88-89) and
            # the reference itself exceeds it (+-145 seen in 2e8 oracle steps of this very run).
            inb = (o1 >= lo) & (o1 <= hi)
            inb[:, 33] = True
            assert bool(inb.all())
            # observation == _get_obs(state); terminated == game_ended; rewards antisymmetric
            fresh = raw.observe()
            assert torch.equal(fresh["player_1"], o1) and torch.equal(fresh["player_2"], o2)
            assert torch.equal(term["player_1"], raw.state[42] != 0)
            assert torch.equal(rew["player_1"], -rew["player_2"])
            assert bool((rew["player_1"].abs() <= 1).all())
            assert bool((raw.scores <= 3).all()) and bool((raw.scores >= 0).all())
            # a reward is paid exactly on round-ending frames
            assert torch.equal(rew["player_1"] != 0, raw.state[41] != 0)
        terms += int(term["player_1"].sum().item())
    assert terms > 0 and raw.episodes_done == terms
    # determinism + shard invariance: the first 4 096 lanes of a shard starting at lane 1 000
    sub = make_env(num_envs=4096, seed=2024, env_id_base=1000, winning_score=3, **kw)
    sub.reset()
    sub.step_random(7, t0=0, k=400)
    assert torch.equal(sub.unwrapped.state, raw.state[:, 1000:1000 + 4096])
    # cross-check a slice against the oracle at full length
    ocfg = oracle.make_config(winning_score=3, is_player2_computer=kw.get("is_player2_computer", False), seed=2024,
                              env_id_base=n - 512)
    ref = oracle.OracleEnv(512, ocfg, nthreads=8)
    ref.reset()
    ref.rollout_random(7, 0, 400)
    assert np.array_equal(cpu(raw.state[:, n - 512:]), ref.state)


# ------------------------------------------------------------------------------------------------
# 4. C ABI edge cases
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tables", [True, "power_hit", False])
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 129])
def test_small_and_ragged_sizes(n, tables, oracle):
    env = make_env(num_envs=n, seed=1, env_id_base=5, is_player1_computer=True, is_player2_computer=True,
                   winning_score=1, flight_tables=tables)
    env.reset()
    ref = oracle.OracleEnv(n, oracle.make_config(winning_score=1, is_player1_computer=True, is_player2_computer=True,
                                                 seed=1, env_id_base=5))
    ref.reset()
    for t in range(300):
        acts = env.random_actions(3, t)
        obs, rew, term, _, _ = env.step(acts)
        robs, rrew, rterm = ref.step(cpu(acts["player_1"]), cpu(acts["player_2"]))
    assert np.array_equal(cpu(env.unwrapped.state), ref.state)
    assert np.array_equal(cpu(obs["player_1"]), robs[0]) and np.array_equal(cpu(obs["player_2"]), robs[1])
    assert np.array_equal(cpu(rew["player_1"]), rrew[0])


def test_stride_larger_than_n_and_error_codes(oracle):
    from pikazoo_amd import _native

    lib = _native.load()
    n, stride = 100, 256
    cfg = _native.PzConfig()
    cfg.winning_score, cfg.serve_mode, cfg.p2_computer, cfg.auto_reset, cfg.seed, cfg.env_id_base = 2, 0, 1, 1, 77, 9
    dev = torch.device("cuda:0")
    state = torch.full((44, stride), -7, dtype=torch.int32, device=dev)
    obs1 = torch.zeros((n, 35), dtype=torch.int32, device=dev)
    obs2 = torch.zeros_like(obs1)
    rew1 = torch.zeros(n, dtype=torch.int32, device=dev)
    rew2 = torch.zeros_like(rew1)
    term = torch.zeros(n, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.pz_init(state.data_ptr(), n, stride, C.byref(cfg), s) == 0
    assert lib.pz_reset(state.data_ptr(), n, stride, C.byref(cfg), None, obs1.data_ptr(), obs2.data_ptr(), None, s) == 0
    for t in range(100):
        assert lib.pz_step_random(state.data_ptr(), n, stride, C.byref(cfg), 11, t, 1, obs1.data_ptr(),
                                  obs2.data_ptr(), rew1.data_ptr(), rew2.data_ptr(), term.data_ptr(), None, None,
                                  None, s) == 0
    torch.cuda.synchronize()
    assert bool((state[:, n:] == -7).all()), "columns beyond n must not be touched"
    ref = oracle.OracleEnv(n, oracle.make_config(winning_score=2, is_player2_computer=True, seed=77, env_id_base=9))
    ref.reset()
    ref.rollout_random(11, 0, 100)
    assert np.array_equal(cpu(state[:, :n]), ref.state)
    assert np.array_equal(cpu(obs1), ref.obs[0])
    # argument errors are reported, not launched
    assert lib.pz_init(None, n, stride, C.byref(cfg), s) == -1
    assert lib.pz_init(state.data_ptr(), n, n - 1, C.byref(cfg), s) == -2
    cfg.winning_score = 0
    assert lib.pz_init(state.data_ptr(), n, stride, C.byref(cfg), s) == -3
    cfg.winning_score = 2
    assert lib.pz_step_random(state.data_ptr(), n, stride, C.byref(cfg), 11, 0, 0, obs1.data_ptr(), obs2.data_ptr(),
                              rew1.data_ptr(), rew2.data_ptr(), term.data_ptr(), None, None, None, s) == -2
    assert lib.pz_observe(state.data_ptr(), n, stride, 0, 0, obs1.data_ptr() + 4, obs2.data_ptr(), s) == -4
    cfg.normal_state_mode = 3
    assert lib.pz_init(state.data_ptr(), n, stride, C.byref(cfg), s) == -3
    cfg.normal_state_mode = 0
    assert lib.pz_init(state.data_ptr(), 0, stride, C.byref(cfg), s) == 0  # empty batch is a no-op
    assert b"aligned" in lib.pz_error_string(-4)


def test_masked_reset_and_frozen_lanes(oracle):
    n = 256
    env = make_env(num_envs=n, seed=3, winning_score=1, auto_reset=False)
    raw = env.unwrapped
    ref = oracle.OracleEnv(n, oracle.make_config(winning_score=1, seed=3, auto_reset=False))
    env.reset(), ref.reset()
    for t in range(500):
        acts = env.random_actions(8, t)
        obs, rew, term, _, _ = env.step(acts)
        ref.step(cpu(acts["player_1"]), cpu(acts["player_2"]))
    assert bool(term["player_1"].all()) and np.array_equal(cpu(raw.state), ref.state)
    frozen = raw.state.clone()
    obs, rew, term, _, _ = env.step(acts)
    assert torch.equal(raw.state, frozen) and bool((rew["player_1"] == 0).all()) and bool(term["player_1"].all())
    mask = (torch.arange(n, device=raw.device) % 3 == 0)
    obs, _ = env.reset(mask=mask)
    r1, r2 = ref.reset(cpu(mask).astype(np.uint8))
    assert np.array_equal(cpu(raw.state), ref.state)
    assert np.array_equal(cpu(obs["player_1"]), r1) and np.array_equal(cpu(obs["player_2"]), r2)
    assert torch.equal(raw.state[42] != 0, ~mask)


@pytest.mark.parametrize("kw", [dict(is_player2_computer=True), dict(is_player1_computer=True, is_player2_computer=True)])
def test_k_frame_launches_with_freezing_computer_games(kw, oracle):
    """Between the frames of one launch the landing point after a ball-player collision is only evaluated
    where it stays observable: on the last frame and for a game that freezes on that frame (auto_reset off).
    Games end at different frames of the k-frame launches here; the whole state -- expected_landing_point_x
    included -- must equal the oracle's frame-by-frame run after every launch."""
    n, k = 4096, 37
    env = make_env(num_envs=n, seed=12, env_id_base=40, winning_score=1, auto_reset=False, **kw)
    ref = oracle.OracleEnv(n, oracle.make_config(winning_score=1, seed=12, env_id_base=40, auto_reset=False, **kw),
                           nthreads=4)
    env.reset(), ref.reset()
    frozen_seen = 0
    for r in range(12):
        env.unwrapped.step_random(5, t0=r * k, k=k)
        ref.rollout_random(5, r * k, k)
        st = cpu(env.unwrapped.state)
        assert np.array_equal(st, ref.state), r
        frozen_seen = int((st[42] != 0).sum())  # game_ended
    assert 0 < frozen_seen <= n
    out = env.unwrapped.rollout_random(5, k, t0=12 * k)   # the trajectory kernel takes the same path
    ref.rollout_random(5, 12 * k, k)
    assert np.array_equal(cpu(env.unwrapped.state), ref.state)


def test_checkpoint_roundtrip():
    env = make_env(num_envs=512, seed=8, is_player2_computer=True)
    env.reset()
    env.step_random(1, k=100)
    sd = env.unwrapped.state_dict()
    env.step_random(1, k=50)
    after = env.unwrapped.state.clone()
    env2 = make_env(num_envs=512, seed=8, is_player2_computer=True)
    env2.unwrapped.load_state_dict(sd)
    env2.step_random(1, k=50)
    assert torch.equal(env2.unwrapped.state, after)


# ------------------------------------------------------------------------------------------------
# 5. the closed-form fast-forward of the flight predictors == the reference's iteration
# ------------------------------------------------------------------------------------------------
def _selftest(x, y, xv, yv, full_net):
    import diag  # libpikazoo_diag.so (include/pikazoo_diag.h): the product's predictors, compiled from its own headers

    lib = diag.load()
    n = x.numel()
    fast, it = torch.empty_like(x), torch.empty_like(x)
    rc = lib.pz_selftest_predictor(x.data_ptr(), y.data_ptr(), xv.data_ptr(), yv.data_ptr(), n, int(full_net),
                                   fast.data_ptr(), it.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return fast, it


@pytest.mark.parametrize("full_net", [True, False])
def test_predictor_fast_forward_equals_iteration_exhaustive(full_net, oracle):
    """Every ball state of the reachable domain and far beyond it: x in [20,432], y in [0,252], x velocity in
    [-20,20] (physics.py:607-626 bounds it), EVERY integer y velocity in [-300,300] (play reaches +-145);
    2.6e9 states per predictor form."""
    dev = torch.device("cuda:0")
    xs = torch.arange(20, 433, dtype=torch.int32, device=dev)
    ys = torch.arange(0, 253, dtype=torch.int32, device=dev)
    xvs = torch.arange(-20, 21, dtype=torch.int32, device=dev)
    gx, gy, gxv = (t.contiguous().reshape(-1) for t in torch.meshgrid(xs, ys, xvs, indexing="ij"))
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for yv in range(-300, 301):
        gyv = torch.full_like(gx, yv)
        fast, it = _selftest(gx, gy, gxv, gyv, full_net)
        bad += (fast != it).sum()
    assert int(bad.item()) == 0
    # spot-check the device iteration itself against the CPU oracle
    g = torch.Generator(device="cpu").manual_seed(5)
    m = 3000
    X = torch.randint(20, 433, (m,), generator=g, dtype=torch.int32)
    Y = torch.randint(0, 253, (m,), generator=g, dtype=torch.int32)
    XV = torch.randint(-20, 21, (m,), generator=g, dtype=torch.int32)
    YV = torch.randint(-150, 151, (m,), generator=g, dtype=torch.int32)
    if full_net:
        exp = [oracle.expected_landing_x(int(X[j]), int(Y[j]), int(XV[j]), int(YV[j])) for j in range(m)]
        sxv, syv = XV, YV
    else:
        # expected_landing_point_x_when_power_hit substitutes the velocities (physics.py:841-845)
        xd = torch.randint(0, 2, (m,), generator=g, dtype=torch.int32)
        yd = torch.randint(-1, 2, (m,), generator=g, dtype=torch.int32)
        exp = [oracle.expected_landing_x_power_hit(int(xd[j]), int(yd[j]), int(X[j]), int(Y[j]), int(XV[j]),
                                                   int(YV[j])) for j in range(m)]
        sxv = torch.where(X < 216, (xd + 1) * 10, -(xd + 1) * 10).to(torch.int32)
        syv = (YV.abs() * yd * 2).to(torch.int32)
    fast, it = _selftest(X.to(dev), Y.to(dev), sxv.to(dev).contiguous(), syv.to(dev).contiguous(), full_net)
    assert cpu(fast).tolist() == exp and cpu(it).tolist() == exp


def test_predictor_extreme_inputs():
    """Out-of-domain speeds, the iteration cap and the net-top bounce loop (x velocity 0 inside the net
    box never lands: both forms must stop at the cap with the same x)."""
    dev = torch.device("cuda:0")
    cases = [(216, 180, 0, 5), (216, 100, 0, 1), (200, 0, 0, 1), (30, 252, -20, -3), (432, 0, 20, 1000),
             (20, 0, -20, -1000), (216, 176, 1, 0), (216, 177, -1, 0), (240, 191, 0, 16), (192, 192, 0, 1),
             (300, 10, 7, -2000), (56, 0, 0, 1), (376, 0, 0, 1)]
    t = torch.tensor(cases, dtype=torch.int32, device=dev)
    for full_net in (True, False):
        fast, it = _selftest(*(t[:, k].contiguous() for k in range(4)), full_net)
        assert torch.equal(fast, it), (full_net, fast.tolist(), it.tolist())


# ------------------------------------------------------------------------------------------------
# 6. k-frame rollout with every frame's outputs kept == k single steps
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kw,wr", [
    (dict(winning_score=2), {}),
    (dict(is_player2_computer=True, winning_score=3, serve="random"), {}),
    (dict(is_player1_computer=True, is_player2_computer=True, winning_score=1),
     dict(simplify_action=True, additional_reward=(0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01))),
])
def test_rollout_random_equals_stepwise(kw, wr, oracle):
    n, k, rounds, aseed = 4096 + 64, 40, 5, 2718
    a = make_env(num_envs=n, seed=6, env_id_base=300, wrappers=wr, **kw)
    b = make_env(num_envs=n, seed=6, env_id_base=300, wrappers=wr, **kw)
    a.reset(), b.reset()
    out = None
    for r in range(rounds):
        out = a.unwrapped.rollout_random(aseed, k, out=out)
        assert out["actions"].shape == (k, 2, n) and out["obs"]["player_1"].shape == (k, n, 35)
        for t in range(k):
            acts = b.unwrapped.random_actions(aseed)  # t = steps_done
            assert torch.equal(out["actions"][t, 0], acts["player_1"]) and torch.equal(out["actions"][t, 1],
                                                                                       acts["player_2"])
            obs, rew, term, _, _ = b.step(acts)
            for ag in ("player_1", "player_2"):
                assert torch.equal(out["obs"][ag][t], obs[ag]), (r, t, ag)
                assert torch.equal(out["rewards"][ag][t], rew[ag]), (r, t, ag)
            assert torch.equal(out["terminations"][t], term["player_1"]), (r, t)
        assert torch.equal(a.unwrapped.state, b.unwrapped.state)
    assert a.unwrapped.steps_done == b.unwrapped.steps_done == k * rounds
    assert a.unwrapped.episodes_done == int(sum(0 for _ in ())) + a.unwrapped.episodes_done  # counter readable
    # and against the oracle at the end
    from oracle.ref_capture import fused_options
    ocfg = oracle.make_config(
        winning_score=kw.get("winning_score", 15), serve=kw.get("serve", "winner"),
        is_player1_computer=kw.get("is_player1_computer", False),
        is_player2_computer=kw.get("is_player2_computer", False), seed=6, env_id_base=300, **fused_options(wr))
    ref = oracle.OracleEnv(n, ocfg, nthreads=8)
    ref.reset()
    eps = ref.rollout_random(aseed, 0, k * rounds)
    assert np.array_equal(cpu(a.unwrapped.state), ref.state)
    assert a.unwrapped.episodes_done == eps
    # the single-frame views follow the last frame
    o = a.unwrapped._pack_obs()
    assert torch.equal(o["player_1"], out["obs"]["player_1"][-1])


def test_rollout_argument_checks():
    env = make_env(num_envs=6)  # not a multiple of 4
    env.reset()
    with pytest.raises(ValueError):
        env.unwrapped.rollout_random(1, 2)
    out = env.unwrapped.rollout_random(1, 1)  # a single frame has no alignment constraint
    assert out["obs"]["player_1"].shape == (1, 6, 35)


# ------------------------------------------------------------------------------------------------
# 7. long runs against the oracle (rare branches: double collisions, net-top bounces, long flights)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kw", [
    dict(winning_score=15),
    dict(winning_score=15, is_player2_computer=True),
    dict(winning_score=4, is_player1_computer=True, is_player2_computer=True, serve="alternate"),
])
def test_long_run_digests_vs_oracle(kw, oracle):
    """4 096 games x 20 000 frames (8.2e7 game-steps per config), state digest compared every 2 000 frames."""
    n, chunk, chunks = 4096, 2000, 10
    env = make_env(num_envs=n, seed=77, env_id_base=1 << 20, **kw)
    env.reset()
    ref = oracle.OracleEnv(n, oracle.make_config(seed=77, env_id_base=1 << 20, **kw), nthreads=8)
    ref.reset()
    eps = 0
    for c in range(chunks):
        env.unwrapped.step_random(5, t0=c * chunk, k=chunk)
        eps += ref.rollout_random(5, c * chunk, chunk)
        assert oracle.digest(cpu(env.unwrapped.state)) == ref.digest(), f"after {(c + 1) * chunk} frames"
    assert np.array_equal(cpu(env.unwrapped.state), ref.state)
    assert env.unwrapped.episodes_done == eps and eps > 0


def test_wide_seeds_and_step_index_wraparound(oracle):
    """64-bit seeds, game ids above 2^32 and a policy step index that crosses 2^32 (counter word 3)."""
    n = 1024
    seed, aseed, base = 0xFEDCBA9876543210, 0x0123456789ABCDEF, (1 << 40) + 12345
    t0 = (1 << 32) - 5
    env = make_env(num_envs=n, seed=seed, env_id_base=base, is_player2_computer=True, winning_score=1)
    ref = oracle.OracleEnv(n, oracle.make_config(seed=seed, env_id_base=base, is_player2_computer=True,
                                                 winning_score=1), nthreads=4)
    env.reset(), ref.reset()
    env.unwrapped.step_random(aseed, t0=t0, k=300)
    ref.rollout_random(aseed, t0, 300)
    assert np.array_equal(cpu(env.unwrapped.state), ref.state)
    a = env.unwrapped.random_actions(aseed, t0 + 4)
    b = env.unwrapped.random_actions(aseed, t0 + 5)  # = 2^32: high word of t becomes 1
    r4 = oracle.random_actions(n, base, aseed, t0 + 4)
    r5 = oracle.random_actions(n, base, aseed, t0 + 5)
    assert np.array_equal(cpu(a["player_1"]), r4[0]) and np.array_equal(cpu(b["player_2"]), r5[1])
    assert not torch.equal(a["player_1"], b["player_1"])


def test_step_many_equals_single_steps(oracle):
    """pz_step_many: k frames of GIVEN actions per launch == k calls of pz_step on the slices."""
    n, k = 2048, 64
    kw = dict(num_envs=n, seed=31, env_id_base=9, is_player1_computer=True, winning_score=2)
    wr = dict(stack=[["RewardInNormalState", dict(reward=0.5)], ["RecordEpisodeStatistics", {}]])
    a, b = make_env(wrappers=wr, **kw), make_env(wrappers=wr, **kw)
    a.reset(), b.reset()
    g = torch.Generator(device="cpu").manual_seed(4)
    out = None
    for r in range(3):
        tape = torch.randint(0, 18, (k, 2, n), generator=g, dtype=torch.int32).to("cuda:0")
        out = a.unwrapped.step_many(tape, out=out)
        for t in range(k):
            obs, rew, term, _, infos = b.step({"player_1": tape[t, 0], "player_2": tape[t, 1]})
            for ag in ("player_1", "player_2"):
                assert torch.equal(out["obs"][ag][t], obs[ag]) and torch.equal(out["rewards"][ag][t], rew[ag])
            assert torch.equal(out["terminations"][t], term["player_1"])
        assert torch.equal(a.unwrapped.state, b.unwrapped.state)
        assert torch.equal(a.unwrapped._stats, b.unwrapped._stats)
    with pytest.raises(ValueError):
        a.unwrapped.step_many(torch.zeros((2, 2, n + 1), dtype=torch.int32, device="cuda:0"))
    va = make_env(num_envs=8, validate_actions=True)
    va.reset()
    va.unwrapped.step_many(torch.full((2, 2, 8), 18, dtype=torch.int32, device="cuda:0"))
    with pytest.raises(IndexError):  # (counted by the launch as it parks the tape: tests/test_gpu_api.py)
        va.unwrapped.check_actions()


@pytest.mark.parametrize("fmt", ["int32", "packed"])
@pytest.mark.parametrize("variant", ["human", "p2_computer", "both_computer", "int16_rows", "p2_computer_computed"])
def test_step_many_refills_its_parked_tape(variant, fmt, oracle):
    """pz_step_many parks its tape in LDS 64 frames at a time, one byte per action: launches of k = 150 (two refills, the
    last chunk 22 frames = one whole batch of rows and a partial one), k = 64 (exactly one chunk) and k = 65 (a one-frame
    refill), every action 0..17 on the tape, a ragged batch -- every frame's outputs and the state after every launch
    against the oracle; in each of the kernels that fetch a tape (one wave per 64 games; two waves with a computer
    player, with two, and with int16 rows; the scout-wave launch without flight tables)."""
    n = 64 * 5 + 24  # (int16 rows want a multiple of eight games)
    kw = dict(num_envs=n, seed=77, env_id_base=5, state_format=fmt, winning_score=2)
    okw = dict(seed=77, env_id_base=5, winning_score=2)
    if variant in ("p2_computer", "p2_computer_computed", "both_computer"):
        kw.update(is_player2_computer=True), okw.update(is_player2_computer=True)
    if variant == "both_computer":
        kw.update(is_player1_computer=True), okw.update(is_player1_computer=True)
    if variant == "p2_computer_computed":
        kw.update(flight_tables=False)
    if variant == "int16_rows":
        kw.update(observation_dtype=torch.int16)
    env = make_env(**kw)
    raw = env.unwrapped
    env.reset()
    ref = oracle.OracleEnv(n, oracle.make_config(**okw), nthreads=4)
    ref.reset()
    rng = np.random.default_rng(12)
    for launch, k in enumerate((150, 64, 65)):
        tape = rng.integers(0, 18, size=(k, 2, n), dtype=np.int32)
        tape[:18, 0, 0] = np.arange(18)  # every action value on the tape for sure
        tape[:18, 1, 1] = np.arange(18)[::-1]
        out = raw.step_many(torch.from_numpy(tape).to(raw.device))
        for f in range(k):
            robs, rrew, rterm = ref.step(tape[f, 0], tape[f, 1])
            ctx = (variant, fmt, launch, f)
            assert np.array_equal(cpu(out["obs"]["player_1"][f]).astype(np.int32), robs[0]), ctx
            assert np.array_equal(cpu(out["obs"]["player_2"][f]).astype(np.int32), robs[1]), ctx
            assert np.array_equal(cpu(out["rewards"]["player_1"][f]), rrew[0]), ctx
            assert np.array_equal(cpu(out["rewards"]["player_2"][f]), rrew[1]), ctx
            assert np.array_equal(cpu(out["terminations"][f]).astype(np.uint8), rterm), ctx
        assert np.array_equal(cpu(raw.state), ref.state), (variant, fmt, launch)


@pytest.mark.parametrize("fmt", ["int32", "packed"])
def test_large_batch_offsets(fmt, oracle):
    """2^24 games in one launch (3 GB of state -- 0.6 GB packed --, 2 x 2.3 GB of observations): the 32-bit buffer
    offsets near their upper range, through pz_step_random and pz_step (single-wave kernel / packed pair kernel),
    checked against the oracle on slices at the start, middle and end of the batch."""
    n, seed, base, aseed = 1 << 24, 5, 3, 17
    env = make_env(num_envs=n, seed=seed, env_id_base=base, winning_score=1, state_format=fmt)
    raw = env.unwrapped
    env.reset()
    for t in range(3):
        obs, rew, term, _, _ = env.step_random(aseed)
    for t in range(3, 6):
        obs, rew, term, _, _ = env.step(raw.random_actions(aseed, t))
    state = raw.state  # (packed: one unpacked copy)
    for lo in (0, n // 2 - 256, n - 512):
        ref = oracle.OracleEnv(512, oracle.make_config(winning_score=1, seed=seed, env_id_base=base + lo))
        ref.reset()
        for t in range(6):
            a1, a2 = oracle.random_actions(512, base + lo, aseed, t)
            robs, rrew, rterm = ref.step(a1, a2)
        assert np.array_equal(cpu(state[:, lo:lo + 512]), ref.state), lo
        assert np.array_equal(cpu(obs["player_1"][lo:lo + 512]), robs[0]), lo
        assert np.array_equal(cpu(obs["player_2"][lo:lo + 512]), robs[1]), lo
        assert np.array_equal(cpu(rew["player_2"][lo:lo + 512]), rrew[1]), lo
    del env, raw, obs, rew, term, state
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
# 8. randomized configuration sweep straight through the C ABI (odd sizes, stride > n, every flag mix)
# ------------------------------------------------------------------------------------------------
def test_randomized_config_sweep_vs_oracle(oracle):
    import random

    from pikazoo_amd import _native

    from pikazoo_amd.env import flight_tables

    lib = _native.load()
    rnd = random.Random(int(os.environ.get("PZ_SWEEP_SEED", "20241008")))  # (another seed: another set of configurations)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    tables_ref = C.byref(flight_tables(dev)[0])
    hit_only = flight_tables(dev, landing=False)[0]  # (ABI 10: the 82 MB power-hit table alone)
    hit_only_ref = C.byref(hit_only)
    trials = int(os.environ.get("PZ_SWEEP_TRIALS", "60"))  # a one-off long run is kept under profiles/
    for trial in range(trials):
        if trial and trial % 5000 == 0:  # (a long run says that it is alive: the GPU box kills a silent command as hung)
            print(f"[sweep] {trial} of {trials} configurations bit-exact so far", flush=True)
        # both flight look-up tables, the power-hit table alone, or computed predictors
        tb = rnd.choice([tables_ref, tables_ref, hit_only_ref, None, None])
        # (ABI 10) the element type of pz_step's action vectors: int32, int64, uint8, int16
        act_name = rnd.choice(["int32", "int32", "int64", "int64", "uint8", "int16"])
        act_dtype = getattr(torch, act_name)
        n = rnd.choice([1, 2, 3, 31, 63, 64, 65, 100, 127, 128, 129, 255, 300, 511, 640, 700])
        stride = n + rnd.choice([0, 0, 1, 7, 64, 130])
        k = dict(winning_score=rnd.choice([1, 1, 2, 3]), serve=rnd.choice(["winner", "alternate", "random"]),
                 is_player1_computer=rnd.random() < 0.4, is_player2_computer=rnd.random() < 0.4,
                 simplify_action=rnd.random() < 0.5,
                 additional_reward=[rnd.choice([0.0, 0.5, -0.25, 1.0, -1.0]) for _ in range(8)] if rnd.random() < 0.5
                 else None,
                 x_line=rnd.choice([216, 100, 300]), y_line=rnd.choice([176, 60, 240]),
                 normal_state_reward=rnd.choice([None, None, 0.125, -0.5]), normal_state_outside=rnd.random() < 0.5,
                 normalize_obs=rnd.random() < 0.4, episode_stats=rnd.choice([0, 0, 1, 2]),
                 auto_reset=rnd.random() < 0.8, seed=rnd.getrandbits(64), env_id_base=rnd.getrandbits(40))
        if k["normal_state_reward"] is None or k["additional_reward"] is None:
            k["normal_state_outside"] = False  # "outside" only means something relative to RewardByBallPosition
        ocfg = oracle.make_config(**k)
        cfg = _native.PzConfig.from_buffer_copy(ocfg)  # identical layout (checked by the CPU tests)
        packed = rnd.random() < 0.5                     # the state in the packed format (36 bytes per game)
        cfg.packed_state = int(packed)
        ref = oracle.OracleEnv(n, ocfg, nthreads=2)
        if packed:
            state = torch.full((36 * stride,), 0xA5, dtype=torch.uint8, device=dev)
        else:
            state = torch.full((44, stride), -99, dtype=torch.int32, device=dev)

        def columns():
            """(int32[44, n] state, nothing past lane n was written)"""
            if not packed:
                return cpu(state[:, :n]), bool((state[:, n:] == -99).all())
            out = torch.full((44, stride), -99, dtype=torch.int32, device=dev)
            flagged = torch.zeros(1, dtype=torch.int64, device=dev)
            assert lib.pz_unpack_state(state.data_ptr(), n, stride, out.data_ptr(), stride, flagged.data_ptr(), stream) == 0
            assert int(flagged.item()) == 0
            clean = all(bool((part == 0xA5).all()) for part in (
                state[16 * n:16 * stride], state[16 * stride + 16 * n:32 * stride], state[32 * stride + 4 * n:]))
            return cpu(out[:, :n]), clean and bool((out[:, n:] == -99).all())
        # episode statistics: double[2][stride] returns + int32[stride] lengths
        stats = torch.zeros(20 * stride, dtype=torch.uint8, device=dev) if k["episode_stats"] else None
        obs = [torch.zeros((n, 35), dtype=torch.int32, device=dev) for _ in range(2)]
        rew = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
        term = torch.zeros(n, dtype=torch.uint8, device=dev)
        sp = None if stats is None else stats.data_ptr()
        assert lib.pz_init(state.data_ptr(), n, stride, C.byref(cfg), stream) == 0
        assert lib.pz_reset(state.data_ptr(), n, stride, C.byref(cfg), None, obs[0].data_ptr(), obs[1].data_ptr(), sp,
                            stream) == 0
        ref.reset()
        n_act = 13 if k["simplify_action"] else 18
        aseed = rnd.getrandbits(64)
        t = 0
        for phase in range(4):
            # (the trajectory launches want a multiple of four games)
            mode = rnd.choice(["step", "random", "random_k"] + (["tape", "rollout"] if n % 4 == 0 else []))
            if mode in ("tape", "rollout"):
                # pz_step_many / pz_rollout_random: every frame's outputs against the oracle's frame; k on both sides of
                # the tape's 64-frame chunk
                kk = rnd.choice([2, 7, 33, 64, 65, 130])
                t_obs = [torch.full((kk, n, 35), -7, dtype=torch.int32, device=dev) for _ in range(2)]
                t_rew = [torch.full((kk, n), -7, dtype=torch.int32, device=dev) for _ in range(2)]
                t_term = torch.full((kk, n), 9, dtype=torch.uint8, device=dev)
                acts = [oracle.random_actions(n, k["env_id_base"], aseed, t + f, n_act) for f in range(kk)]
                tape = torch.from_numpy(np.stack([np.stack(a) for a in acts]).astype(np.int32)).to(dev)
                if mode == "tape":
                    assert lib.pz_step_many(state.data_ptr(), n, stride, C.byref(cfg), tape.data_ptr(), kk,
                                            t_obs[0].data_ptr(), t_obs[1].data_ptr(), t_rew[0].data_ptr(),
                                            t_rew[1].data_ptr(), t_term.data_ptr(), sp, None, tb, stream) == 0
                else:
                    t_act = torch.full((kk, 2, n), -7, dtype=torch.int32, device=dev)
                    assert lib.pz_rollout_random(state.data_ptr(), n, stride, C.byref(cfg), aseed, t, kk, t_act.data_ptr(),
                                                 t_obs[0].data_ptr(), t_obs[1].data_ptr(), t_rew[0].data_ptr(),
                                                 t_rew[1].data_ptr(), t_term.data_ptr(), sp, None, tb, stream) == 0
                    assert torch.equal(t_act, tape), (trial, phase, mode)
                h_obs, h_rew, h_term = [cpu(x) for x in t_obs], [cpu(x) for x in t_rew], cpu(t_term)
                for f in range(kk):
                    robs, rrew, rterm = ref.step(*acts[f])
                    ctx = (trial, phase, mode, kk, f, n, stride, tb is not None, packed, k)
                    assert np.array_equal(h_obs[0][f], robs[0].view(np.int32)), ctx
                    assert np.array_equal(h_obs[1][f], robs[1].view(np.int32)), ctx
                    assert np.array_equal(h_rew[0][f], rrew[0].view(np.int32)), ctx
                    assert np.array_equal(h_rew[1][f], rrew[1].view(np.int32)), ctx
                    assert np.array_equal(h_term[f], rterm), ctx
                for dst, src in ((obs[0], t_obs[0]), (obs[1], t_obs[1]), (rew[0], t_rew[0]), (rew[1], t_rew[1]), (term, t_term)):
                    dst.copy_(src[-1])  # the single-frame buffers of this test follow the last frame
                t += kk
            elif mode == "step":
                cfg.action_format = _native.ACTION_FORMATS[act_name]
                for _ in range(15):
                    a1, a2 = oracle.random_actions(n, k["env_id_base"], aseed, t, n_act)
                    d1, d2 = torch.as_tensor(a1, device=dev).to(act_dtype), torch.as_tensor(a2, device=dev).to(act_dtype)
                    assert lib.pz_step(state.data_ptr(), n, stride, C.byref(cfg), d1.data_ptr(), d2.data_ptr(),
                                       obs[0].data_ptr(), obs[1].data_ptr(), rew[0].data_ptr(), rew[1].data_ptr(),
                                       term.data_ptr(), sp, tb, stream) == 0
                    ref.step(a1, a2)
                    t += 1
                cfg.action_format = 0  # (pz_step_many parks int32 rows)
            else:
                kk = 1 if mode == "random" else rnd.choice([2, 5, 17])
                reps = 12 if mode == "random" else 2
                for _ in range(reps):
                    assert lib.pz_step_random(state.data_ptr(), n, stride, C.byref(cfg), aseed, t, kk,
                                              obs[0].data_ptr(), obs[1].data_ptr(), rew[0].data_ptr(),
                                              rew[1].data_ptr(), term.data_ptr(), sp, None, tb, stream) == 0
                    ref.rollout_random(aseed, t, kk)
                    t += kk
            torch.cuda.synchronize()
            ctx = (trial, phase, mode, n, stride, "none" if tb is None else ("both" if tb is tables_ref else "power_hit"),
                   act_name, packed, k)
            got, untouched = columns()
            assert np.array_equal(got, ref.state), ctx
            assert untouched, ctx
            assert np.array_equal(cpu(obs[0]), ref.obs[0].view(np.int32)) and np.array_equal(cpu(obs[1]),
                                                                                           ref.obs[1].view(np.int32)), ctx
            assert np.array_equal(cpu(rew[0]), ref.rew[0].view(np.int32)) and np.array_equal(cpu(rew[1]),
                                                                                           ref.rew[1].view(np.int32)), ctx
            assert np.array_equal(cpu(term), ref.term), ctx
            if stats is not None:
                ret = stats[:16 * stride].view(torch.float64).view(2, stride)
                assert np.array_equal(cpu(ret[:, :n]), ref.episode_returns), ctx
                assert np.array_equal(cpu(stats[16 * stride:].view(torch.int32)[:n]), ref.episode_lengths), ctx
                assert bool((ret[:, n:] == 0).all()), ctx


@pytest.mark.parametrize("kw,steps,switch", [
    (dict(winning_score=1), 12, 393216),
    (dict(winning_score=1, is_player2_computer=True), 60, 393216),
    (dict(winning_score=1, is_player1_computer=True, is_player2_computer=True, serve="random"), 60, 393216),
])
@pytest.mark.parametrize("tables", [True, "power_hit", False])
def test_two_wave_and_single_wave_kernels_agree_across_the_size_switch(kw, steps, switch, tables, oracle):
    """Below 393 216 games pz_step runs two waves per 64 games (split by player; with a computer player and no
    flight tables: frame wave + scout wave for the flight predictions), from there on one: both sides of the
    switch against the oracle."""
    if not tables and not _has_computer(kw):
        pytest.skip("no computer player: the tables are never consulted")
    for n in (switch - 64, switch):
        env = make_env(num_envs=n, seed=44, env_id_base=7, flight_tables=tables, **kw)
        env.reset()
        for t in range(steps):
            obs, rew, term, _, _ = env.step(env.unwrapped.random_actions(21, t))
        for lo in (0, n - 1024):
            ref = oracle.OracleEnv(1024, oracle.make_config(seed=44, env_id_base=7 + lo, **kw), nthreads=4)
            ref.reset()
            for t in range(steps):
                a1, a2 = oracle.random_actions(1024, 7 + lo, 21, t)
                robs, rrew, rterm = ref.step(a1, a2)
            assert np.array_equal(cpu(env.unwrapped.state[:, lo:lo + 1024]), ref.state), (n, lo)
            assert np.array_equal(cpu(obs["player_2"][lo:lo + 1024]), robs[1]), (n, lo)
            assert np.array_equal(cpu(rew["player_1"][lo:lo + 1024]), rrew[0]), (n, lo)
            assert np.array_equal(cpu(term["player_1"][lo:lo + 1024]).astype(np.uint8), rterm), (n, lo)


@pytest.mark.parametrize("fmt,obs_dtype", [("int32", torch.int32), ("packed", torch.int32), ("int32", torch.int16)])
@pytest.mark.parametrize("kw", [dict(winning_score=1, is_player2_computer=True),
                                dict(winning_score=1, is_player1_computer=True),
                                dict(winning_score=2, is_player1_computer=True, is_player2_computer=True, serve="random"),
                                dict(winning_score=1)])  # (human vs human: two waves per 64 games on int16 rows only)
def test_rollout_pair_and_single_wave_kernels_agree_across_the_size_switch(kw, fmt, obs_dtype, oracle):
    """pz_rollout_random with a computer player on the flight tables runs two waves per 64 games below 393 216 games
    (rollout_pair_kernel: the frame split by player, looped) and one from there on: both sides of the switch, every
    frame's outputs and the final state against the oracle on the first and the last 512 games; two launches, so the
    second starts from what the first wrote back."""
    k, m = 20, 512
    for n in (393216 - 64, 393216):
        env = make_env(num_envs=n, seed=45, env_id_base=11, state_format=fmt, observation_dtype=obs_dtype, **kw)
        raw = env.unwrapped
        env.reset()
        refs = {}
        for lo in (0, n - m):
            refs[lo] = oracle.OracleEnv(m, oracle.make_config(seed=45, env_id_base=11 + lo, **kw), nthreads=4)
            refs[lo].reset()
        out = None
        for launch in range(2):
            out = raw.rollout_random(23, k, t0=launch * k, out=out)
            for lo, ref in refs.items():
                for f in range(k):
                    a1, a2 = oracle.random_actions(m, 11 + lo, 23, launch * k + f)
                    robs, rrew, rterm = ref.step(a1, a2)
                    ctx = (n, lo, launch, f)
                    assert np.array_equal(cpu(out["actions"][f, 0, lo:lo + m]), a1), ctx
                    assert np.array_equal(cpu(out["obs"]["player_1"][f, lo:lo + m]).astype(np.int32), robs[0]), ctx
                    assert np.array_equal(cpu(out["obs"]["player_2"][f, lo:lo + m]).astype(np.int32), robs[1]), ctx
                    assert np.array_equal(cpu(out["rewards"]["player_2"][f, lo:lo + m]), rrew[1]), ctx
                    assert np.array_equal(cpu(out["terminations"][f, lo:lo + m]).astype(np.uint8), rterm), ctx
                assert np.array_equal(cpu(raw.state[:, lo:lo + m]), ref.state), (n, lo, launch)


@pytest.mark.parametrize("fmt", ["int32", "packed"])
def test_human_vs_human_k_frame_launches_on_int16_rows_vs_oracle(fmt, oracle):
    """Human vs human with int16 observation rows, pz_rollout_random and pz_step_many run on two waves per 64 games
    (each player's wave writes its agent's rows): a rollout, then the same number of frames from a tape, then a
    rollout again -- every frame's outputs and the state after every launch against the oracle, a ragged batch."""
    n, k = 64 * 37 + 8, 24
    env = make_env(num_envs=n, seed=46, env_id_base=3, state_format=fmt, observation_dtype=torch.int16, winning_score=1)
    raw = env.unwrapped
    env.reset()
    ref = oracle.OracleEnv(n, oracle.make_config(seed=46, env_id_base=3, winning_score=1), nthreads=4)
    ref.reset()
    t = 0
    for launch in range(3):
        acts = [oracle.random_actions(n, 3, 29, t + f) for f in range(k)]
        if launch == 1:
            tape = torch.from_numpy(np.stack([np.stack(a) for a in acts]).astype(np.int32)).to(raw.device)
            out = raw.step_many(tape)
        else:
            out = raw.rollout_random(29, k, t0=t)
        assert out["obs"]["player_1"].dtype == torch.int16
        for f in range(k):
            robs, rrew, rterm = ref.step(*acts[f])
            ctx = (fmt, launch, f)
            assert np.array_equal(cpu(out["actions"][f, 1]), acts[f][1]), ctx
            assert np.array_equal(cpu(out["obs"]["player_1"][f]).astype(np.int32), robs[0]), ctx
            assert np.array_equal(cpu(out["obs"]["player_2"][f]).astype(np.int32), robs[1]), ctx
            assert np.array_equal(cpu(out["rewards"]["player_1"][f]), rrew[0]), ctx
            assert np.array_equal(cpu(out["rewards"]["player_2"][f]), rrew[1]), ctx
            assert np.array_equal(cpu(out["terminations"][f]).astype(np.uint8), rterm), ctx
        assert np.array_equal(cpu(raw.state), ref.state), (fmt, launch)
        t += k


# ------------------------------------------------------------------------------------------------
# 9. the headline size against the oracle on EVERY lane (BASELINE configs 3, 4-shard and 5 at 65 536 games)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,kw,wr", [
    ("random_random_pair_kernel", dict(), {}),
    ("cfg3_p2_computer_tables", dict(is_player2_computer=True), {}),
    ("cfg3_p2_computer_scout", dict(is_player2_computer=True, flight_tables=False), {}),
    ("cfg3_p2_computer_power_hit_table", dict(is_player2_computer=True, flight_tables="power_hit"), {}),
    ("cfg5_fused_wrappers", dict(), dict(simplify_action=True,
                                         additional_reward=(0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01))),
])
def test_headline_size_every_lane_vs_oracle(name, kw, wr, oracle):
    """65 536 games x 320 frames through pz_step (one launch per frame, actions from HBM), all lanes compared
    with the oracle: full state every 40 frames, observations / rewards / terminations on the last frame."""
    from oracle.ref_capture import fused_options

    n, steps, seed, base, aseed = 65536, 320, 11, 1 << 33, 99
    okw = {k: v for k, v in kw.items() if k != "flight_tables"}
    env = make_env(num_envs=n, seed=seed, env_id_base=base, winning_score=2, wrappers=wr, **kw)
    raw = env.unwrapped
    ref = oracle.OracleEnv(n, oracle.make_config(winning_score=2, seed=seed, env_id_base=base, **okw,
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
    assert int(ref.state[42].sum()) >= 0 and ref.state[43].min() >= 4


# ------------------------------------------------------------------------------------------------
# 10. the flight look-up tables themselves
# ------------------------------------------------------------------------------------------------
def test_flight_tables_hold_the_iterative_predictors(oracle):
    """Every entry of both tables == the frame-by-frame iteration (pz_selftest_predictor's out_iter) on the ball
    state the entry stands for, and a sample of them == the CPU oracle.  Index arithmetic is restated here."""
    from pikazoo_amd import _native
    from pikazoo_amd.env import flight_tables

    dev = torch.device("cuda:0")
    _, landing, power_hit = flight_tables(dev)
    lib = _native.load()
    YV, HYV = 96, 64
    entries = (2 * YV + 1) * 23 * 253 * 413
    assert landing.numel() == lib.pz_flight_table_bytes(0) == 2 * entries + 2  # (+ padding: read dword-wise)
    assert power_hit.numel() == lib.pz_flight_table_bytes(1) == 16 * (HYV + 1) * 192 * 413
    land = landing.view(torch.int16)[:entries].view(2 * YV + 1, 23, 253, 413)
    xs = torch.arange(20, 433, dtype=torch.int32, device=dev)
    ys = torch.arange(0, 253, dtype=torch.int32, device=dev)
    xv_values = [-20] + list(range(-10, 11)) + [20]
    gy, gx = (t.contiguous().reshape(-1) for t in torch.meshgrid(ys, xs, indexing="ij"))
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for yi in range(2 * YV + 1):
        for xi, xv in enumerate(xv_values):
            _, it = _selftest(gx, gy, torch.full_like(gx, xv), torch.full_like(gx, yi - YV), True)
            bad += (land[yi, xi].reshape(-1).to(torch.int32) != it).sum()
    assert int(bad.item()) == 0
    hit = power_hit.view(torch.int16).view(HYV + 1, 192, 413, 8)
    hy = torch.arange(61, 253, dtype=torch.int32, device=dev)
    gy, gx = (t.contiguous().reshape(-1) for t in torch.meshgrid(hy, xs, indexing="ij"))
    for ayv in range(HYV + 1):
        for c in range(6):
            xdir, ydir = (1 if c < 3 else 0), (c % 3) - 1
            sxv = torch.where(gx < 216, (xdir + 1) * 10, -(xdir + 1) * 10).to(torch.int32)
            _, it = _selftest(gx, gy, sxv, torch.full_like(gx, ayv * ydir * 2), False)
            bad += (hit[ayv, :, :, c].reshape(-1).to(torch.int32) != it).sum()
    assert int(bad.item()) == 0
    # a sample straight against the oracle's scalar predictors
    g = torch.Generator(device="cpu").manual_seed(9)
    m = 2000
    X = torch.randint(20, 433, (m,), generator=g)
    Y = torch.randint(0, 253, (m,), generator=g)
    XI = torch.randint(0, 23, (m,), generator=g)
    YVs = torch.randint(-YV, YV + 1, (m,), generator=g)
    got = land.cpu()[YVs + YV, XI, Y, X - 20]
    for j in range(m):
        assert int(got[j]) == oracle.expected_landing_x(int(X[j]), int(Y[j]), xv_values[int(XI[j])], int(YVs[j]))
    Yh = torch.randint(61, 253, (m,), generator=g)
    A = torch.randint(0, HYV + 1, (m,), generator=g)
    Cc = torch.randint(0, 6, (m,), generator=g)
    goth = hit.cpu()[A, Yh - 61, X - 20, Cc]
    for j in range(m):
        c = int(Cc[j])
        # the oracle substitutes the velocities itself (physics.py:841-845): pass the ball's |y velocity|
        assert int(goth[j]) == oracle.expected_landing_x_power_hit(1 if c < 3 else 0, (c % 3) - 1, int(X[j]),
                                                                   int(Yh[j]), 0, int(A[j]))


@pytest.mark.parametrize("fmt", ["int32", "packed"])
def test_ball_states_outside_the_tables_take_the_computed_path(fmt, oracle):
    """A ball faster than the tables' domain (|y velocity| > 96 resp. 64) or with an x velocity they do not list is
    predicted in the kernel: plant such states and compare the next frames with the oracle."""
    n = 4096
    kw = dict(is_player1_computer=True, is_player2_computer=True, winning_score=3)
    env = make_env(num_envs=n, seed=3, env_id_base=50, state_format=fmt, **kw)
    raw = env.unwrapped
    ref = oracle.OracleEnv(n, oracle.make_config(seed=3, env_id_base=50, **kw), nthreads=8)
    env.reset(), ref.reset()
    env.step_random(5, t0=0, k=40), ref.rollout_random(5, 0, 40)
    g = torch.Generator(device="cpu").manual_seed(1)
    st = raw.state.cpu()
    st[29] = torch.randint(-200, 201, (n,), generator=g, dtype=torch.int32)          # ball y velocity
    st[28] = torch.randint(-20, 21, (n,), generator=g, dtype=torch.int32)            # x velocities 11..19 included
    # balls next to a jumping computer player, so that the power-hit scan runs on fast balls too
    st[3, ::2] = 1
    st[1, ::2] = 200
    st[26, ::2] = st[0, ::2] + 10
    st[27, ::2] = 190
    raw.set_state(st.to(raw.device))
    ref.state[:] = st.numpy()
    negative_y = 0
    for t in range(40, 70):
        acts = raw.random_actions(5, t)
        obs = env.step(acts)[0]
        robs = ref.step(cpu(acts["player_1"]), cpu(acts["player_2"]))[0]
        assert np.array_equal(cpu(raw.state), ref.state), t
        assert np.array_equal(cpu(obs["player_1"]), robs[0]), t
        negative_y += int((ref.state[27] < 0).sum())
    # the planted fast balls over the net top are bounced to y - y_velocity < 0 (physics.py:406-419: the ceiling is
    # tested before the net): the reference's ball y is a signed quantity, and so are its copies in the trail
    assert negative_y > 0

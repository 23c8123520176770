"""Live check of the oracle against the UNMODIFIED reference, in the build container only.

The reference checkout (/root/reference) does not exist on the GPU box; there this module is skipped and the
committed fixtures (tests/golden/, produced by the same harness) carry the parity pin.  Here it re-runs the
capture harness on fresh seeds, so the pin does not rest on the committed files alone.
"""
import json

import numpy as np
import pytest

from oracle import ref_capture as rc

pytestmark = pytest.mark.skipif(not rc.reference_available(), reason="reference checkout not mounted")

TABLE = [0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01]
LIVE = [
    ("human_human", dict(winning_score=5, serve="winner"), None),
    ("p1_computer_random_serve", dict(winning_score=3, serve="random", is_player1_computer=True), None),
    ("both_computer_alternate", dict(winning_score=2, serve="alternate", is_player1_computer=True,
                                     is_player2_computer=True), None),
    ("wrapper_stack", dict(winning_score=2, is_player2_computer=True),
     dict(stack=[["SimplifyAction", {}], ["RecordEpisodeStatistics", {}],
                 ["RewardByBallPosition", dict(additional_reward=TABLE)],
                 ["RewardInNormalState", dict(reward=0.03125)], ["NormalizeObservation", {}]])),
]


@pytest.mark.parametrize("name,kw,wr", LIVE)
def test_oracle_tracks_live_reference(oracle, name, kw, wr):
    lanes, steps, seed, aseed, base = 6, 2500, 424242, 31, 99
    d = rc.capture(name, lanes, steps, seed=seed, action_seed=aseed, env_id_base=base, env_kwargs=kw,
                   wrappers=wr, full=True)
    meta = json.loads(bytes(d["meta"]).decode())
    opt = rc.fused_options(wr)
    cfg = oracle.make_config(winning_score=kw.get("winning_score", 15), serve=kw.get("serve", "winner"),
                             is_player1_computer=kw.get("is_player1_computer", False),
                             is_player2_computer=kw.get("is_player2_computer", False), seed=seed,
                             env_id_base=base, **opt)
    env = oracle.OracleEnv(lanes, cfg)
    assert np.array_equal(env.state, d["state_ctor"])
    env.reset()
    assert np.array_equal(env.state, d["state0"])
    float_obs, float_rew = bool(cfg.normalize_obs), env.float_rewards
    for t in range(steps):
        a = d["actions"][t].astype(np.int32)
        obs, rew, term = env.step(a[0], a[1])
        st = d["states"][t].astype(np.int32)
        st[43] = d["rng_counter"][t]
        assert np.array_equal(env.state, st), (name, t)
        exp = d["obs"][t].astype(np.float32) if float_obs else d["obs"][t]
        assert np.array_equal(obs[0], exp[0]) and np.array_equal(obs[1], exp[1]), (name, t)
        assert np.array_equal(term, d["term"][t]), (name, t)
        if float_rew:
            np.testing.assert_allclose(rew[0], d["rew"][t, 0], rtol=0, atol=1e-6)
        else:
            assert np.array_equal(rew[0], d["rew"][t, 0]) and np.array_equal(rew[1], d["rew"][t, 1])
        if cfg.episode_stats_mode:
            done = d["ep_l"][t] >= 0
            if done.any():
                assert np.array_equal(env.episode_lengths[done], d["ep_l"][t][done])
                np.testing.assert_allclose(env.episode_returns[:, done], d["ep_r"][t][:, done], rtol=0, atol=2e-6)
    assert meta["episodes"] == int(d["term"].sum()) and meta["episodes"] > 0


@pytest.mark.parametrize("kw", [dict(winning_score=4), dict(winning_score=2, is_player1_computer=True, is_player2_computer=True),
                                dict(winning_score=3, is_player2_computer=True, serve="random"),
                                dict(winning_score=3, is_player1_computer=True, serve="alternate")])
def test_oracle_tracks_live_reference_from_planted_states(oracle, kw):
    """Fresh planted states on every configuration (the committed fixtures hold one draw of them): every attribute of
    players / ball / scores at random over its valid range, plus the fast-ball corners, stepped by the live reference."""
    for d in (rc.capture_planted_random("live", 1500, seed=31337, action_seed=5, env_id_base=7700, env_kwargs=kw,
                                        frames=10, plant_seed=2024),
              rc.capture_planted("live", seed=31338, action_seed=6, env_id_base=8800, env_kwargs=kw, frames=12)):
        meta = json.loads(bytes(d["meta"]).decode())
        env = oracle.OracleEnv(meta["lanes"], oracle.make_config(
            winning_score=kw.get("winning_score", 15), serve=kw.get("serve", "winner"),
            is_player1_computer=kw.get("is_player1_computer", False),
            is_player2_computer=kw.get("is_player2_computer", False), seed=meta["seed"], env_id_base=meta["env_id_base"]))
        env.state[:] = d["planted"]
        for t in range(meta["frames"]):
            a1, a2 = oracle.random_actions(meta["lanes"], meta["env_id_base"], meta["action_seed"], meta["warm"] + t, 18)
            env.step(a1, a2)
            assert np.array_equal(env.state, d["states"][t]), (kw, t)


def test_draw_list_tracks_live_reference_on_planted_states():
    """render() of the live reference on random planted states (every pose / mirroring / score digit / hyper ball, balls
    partly or wholly off screen): the oracle's draw list equals the recorded blits (punch effect aside)."""
    from oracle import render_oracle as ro

    files = ro.sprite_files()
    sizes = None
    rng = np.random.default_rng(77)
    checked = 0
    for i in range(400):
        env, raw, shim = rc.make_reference_env(5, 300 + i, None, render_mode="rgb_array", winning_score=15)
        env.reset()
        for pl in (raw.physics.player1, raw.physics.player2):
            pl.state = int(rng.integers(0, 5))
            pl.frame_number = int(rng.integers(0, 5 if pl.state < 3 else (2 if pl.state == 3 else 1)))
            pl.diving_direction = int(rng.integers(-1, 2))
            pl.y = int(rng.integers(108, 245))
        b = raw.physics.ball
        b.x, b.y = int(rng.integers(20, 433)), int(rng.integers(-120, 253))
        b.previous_x, b.previous_y = int(rng.integers(20, 433)), int(rng.integers(-120, 253))
        b.previous_previous_x, b.previous_previous_y = int(rng.integers(20, 433)), int(rng.integers(-120, 253))
        b.is_power_hit = bool(rng.integers(0, 2))
        b.fine_rotation = int(rng.integers(0, 51))
        b.rotation = b.fine_rotation // 10
        raw.scores = [int(rng.integers(0, 16)), int(rng.integers(0, 16))]
        raw.render()
        blits = raw.screen.blits[rc.BACKGROUND_BLITS:]
        if sizes is None:
            sizes = [(0, 0)] * ro.SPRITE_COUNT
            import struct
            for k, f in enumerate(files):
                with open(rc.REFERENCE_ROOT / "pikazoo" / "env" / "img" / f, "rb") as fh:
                    head = fh.read(24)
                sizes[k] = struct.unpack(">II", head[16:24])
        want = [(files.index(f), int(flip), x, y, w, h) for (f, flip, scaled), x, y, w, h in blits if f != "ball_punch.png"]
        sc = rc.extract_scenery(raw)
        got = ro.draw_list(rc.extract_state(raw, shim), sizes, sc)
        assert got == want, i
        checked += len(got)
    assert checked > 400 * 45

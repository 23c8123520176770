"""Golden-vector capture from the UNMODIFIED reference (build container only).

Runs ``/root/reference``'s ``pikazoo_v0`` in-process, with

* in-memory stand-ins for the three third-party imports that are not installed here
  (``gymnasium``, ``pettingzoo``, ``pygame`` -- none of them is on the step path), and
* the env RNG injected through the stubbed ``gymnasium.utils.seeding.np_random``: a
  duck-typed object exposing ``integers(lo, hi)`` that yields the build's Philox stream
  (draw index = number of ``integers`` calls so far, i.e. the reference's own call order),

and writes small ``.npz`` fixtures under ``tests/golden/``.  Nothing of the reference is
copied: the fixtures hold inputs (actions, seeds, kwargs) and outputs (state words read
from the reference's attributes, observations, rewards, terminations) only.

Usage (here, not on the GPU box -- /root/reference does not exist there):
    python oracle/ref_capture.py            # regenerate every fixture
TEST INFRASTRUCTURE: not imported by the product.
"""
from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True  # the reference mount is read-only
_HERE = Path(__file__).resolve().parent
_REPO = _HERE.parent
if str(_REPO) not in sys.path:
    sys.path.insert(0, str(_REPO))

from oracle import pz_oracle as po  # noqa: E402

REFERENCE_ROOT = Path("/root/reference")
GOLDEN = _REPO / "tests" / "golden"


# --------------------------------------------------------------------------------------
# RNG shim + third-party stand-ins
# --------------------------------------------------------------------------------------
class PhiloxShim:
    """Duck-typed stand-in for ``numpy.random.Generator``: only ``integers`` is ever called
    by the reference (physics.py:218,613,728,729,795; pikazoo_env.py:246)."""

    def __init__(self, seed: int, env_id: int):
        self.seed = seed
        self.env_id = env_id
        self.counter = 0

    def integers(self, low, high=None):
        if high is None:
            low, high = 0, low
        assert low == 0
        v = po.env_draw(self.seed, self.env_id, self.counter, int(high))
        self.counter += 1
        return np.int64(v)  # numpy returns np.int64 for scalar draws


_NEXT_SHIM: list = []
_NEXT_SAMPLES: list = []  # what the stubbed Discrete.sample() returns next (ConvertSingleAgent's opponent draws)


def _np_random_stub(seed=None):
    shim = _NEXT_SHIM.pop() if _NEXT_SHIM else PhiloxShim(0, 0)
    return shim, seed


def install_stubs():
    if "gymnasium" in sys.modules and getattr(sys.modules["gymnasium"], "_pz_stub", False):
        return
    gym = types.ModuleType("gymnasium")
    gym._pz_stub = True
    spaces = types.ModuleType("gymnasium.spaces")

    class Space:
        pass

    class Discrete(Space):
        def __init__(self, n):
            self.n = n

        def sample(self):
            # gymnasium's sampler is an unseeded third-party RNG; the harness feeds it the stream it wants recorded
            v = _NEXT_SAMPLES.pop(0)
            assert 0 <= v < self.n
            return v

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    spaces.Space, spaces.Discrete, spaces.Box = Space, Discrete, Box
    utils = types.ModuleType("gymnasium.utils")
    seeding = types.ModuleType("gymnasium.utils.seeding")
    seeding.np_random = _np_random_stub
    utils.seeding = seeding
    logger = types.ModuleType("gymnasium.logger")
    logger.warn = lambda *a, **k: None
    gym.spaces, gym.utils, gym.logger = spaces, utils, logger

    pz = types.ModuleType("pettingzoo")

    class ParallelEnv:
        pass

    pz.ParallelEnv = ParallelEnv
    pz_utils = types.ModuleType("pettingzoo.utils")
    pz_utils_env = types.ModuleType("pettingzoo.utils.env")
    pz_utils_env.ParallelEnv = ParallelEnv

    class BaseParallelWrapper(ParallelEnv):
        """Behavioural stand-in: stores env, forwards reset/step and attribute access."""

        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            if name == "env":
                raise AttributeError(name)
            return getattr(self.env, name)

        def reset(self, seed=None, options=None):
            return self.env.reset(seed=seed, options=options)

        def step(self, actions):
            return self.env.step(actions)

        def observation_space(self, agent):
            return self.env.observation_space(agent)

        def action_space(self, agent):
            return self.env.action_space(agent)

    pz_utils.BaseParallelWrapper = BaseParallelWrapper
    pz_utils.env = pz_utils_env
    pz.utils = pz_utils

    pygame = _recording_pygame()

    sys.modules.update({
        "gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.utils": utils,
        "gymnasium.utils.seeding": seeding, "gymnasium.logger": logger,
        "pettingzoo": pz, "pettingzoo.utils": pz_utils, "pettingzoo.utils.env": pz_utils_env,
        "pygame": pygame,
    })
    if str(REFERENCE_ROOT) not in sys.path:
        sys.path.insert(0, str(REFERENCE_ROOT))


def _png_size(path):
    """(width, height) from a PNG's IHDR chunk."""
    with open(path, "rb") as f:
        head = f.read(24)
    assert head[:8] == b"\x89PNG\r\n\x1a\n" and head[12:16] == b"IHDR", path
    return int.from_bytes(head[16:20], "big"), int.from_bytes(head[20:24], "big")


def _recording_pygame():
    """Stand-in for the uninstalled third-party ``pygame``: surfaces carry a description instead of pixels and a
    surface that is blitted onto records (what, where).  With it the reference's own ``render()``
    (pikazoo_env.py:250-384) runs unmodified and leaves its draw list -- which sprite, flipped / scaled how, at
    which position -- on the screen surface, and its clouds / waves / punch effect advance exactly as they do under
    the real library (they only depend on the env RNG).  No pixels are produced."""
    pg = types.ModuleType("pygame")
    pg.SRCALPHA = 0x00010000

    class Surface:
        def __init__(self, size, flags=0, desc=None):
            self.size = (int(size[0]), int(size[1]))
            self.desc = desc        # (file name, flip_x, scaled size or None) once it holds an image
            self.blits = []         # [(desc, x, y, w, h)] in call order

        def get_size(self):
            return self.size

        def get_width(self):
            return self.size[0]

        def get_height(self):
            return self.size[1]

        def blit(self, source, dest):
            if self.desc is None and not self.blits and dest == (0, 0) and source.size == self.size and \
                    source.desc is not None and source.desc[0].endswith(".png") and getattr(source, "is_file", False):
                self.desc = source.desc   # get_image: a fresh SRCALPHA surface receives the loaded file
                return
            self.blits.append((source.desc, int(dest[0]), int(dest[1]), source.size[0], source.size[1]))

    image = types.ModuleType("pygame.image")

    def load(path):
        sfc = Surface(_png_size(path), desc=(Path(path).name, False, None))
        sfc.is_file = True
        return sfc

    image.load = load
    transform = types.ModuleType("pygame.transform")

    def flip(surface, flip_x, flip_y):
        assert not flip_y
        name, fx, scaled = surface.desc
        return Surface(surface.size, desc=(name, fx != bool(flip_x), scaled))

    def scale(surface, size):
        name, fx, _ = surface.desc
        return Surface(size, desc=(name, fx, (int(size[0]), int(size[1]))))

    transform.flip, transform.scale = flip, scale
    surfarray = types.ModuleType("pygame.surfarray")
    surfarray.pixels3d = lambda surface: np.zeros((surface.size[0], surface.size[1], 3), np.uint8)
    pg.Surface, pg.image, pg.transform, pg.surfarray = Surface, image, transform, surfarray
    pg.init = lambda: None
    pg.quit = lambda: None
    return pg


def wrapper_stack(wrappers):
    """Ordered (innermost first) list of (reference wrapper class name, kwargs).

    Accepts the short dict form of the first fixtures ({"simplify_action": True,
    "additional_reward": [...], "x_line": .., "y_line": ..}) or {"stack": [[name, kwargs], ...]}."""
    wrappers = wrappers or {}
    if "stack" in wrappers:
        return [(n, dict(k)) for n, k in wrappers["stack"]]
    stack = []
    if wrappers.get("simplify_action"):
        stack.append(("SimplifyAction", {}))
    if wrappers.get("additional_reward") is not None:
        stack.append(("RewardByBallPosition", dict(additional_reward=list(wrappers["additional_reward"]),
                                                   x_line=wrappers.get("x_line", 216),
                                                   y_line=wrappers.get("y_line", 176))))
    return stack


def fused_options(wrappers) -> dict:
    """What the fused kernel / the oracle must be configured with to equal that wrapper stack."""
    opt = dict(simplify_action=False, additional_reward=None, x_line=216, y_line=176,
               normal_state_reward=None, normal_state_outside=False, normalize_obs=False, episode_stats=0)
    seen_reward_wrapper = False
    stack = wrapper_stack(wrappers)
    for idx, (name, kw) in enumerate(stack):
        if name == "SimplifyAction":
            opt["simplify_action"] = True
        elif name == "RewardByBallPosition":
            assert not opt["normalize_obs"], "RewardByBallPosition must sit below NormalizeObservation"
            assert opt["additional_reward"] is None, "one RewardByBallPosition"
            assert opt["episode_stats"] != 2, "statistics between two reward wrappers"
            opt.update(additional_reward=list(kw["additional_reward"]), x_line=kw.get("x_line", 216),
                       y_line=kw.get("y_line", 176))
            seen_reward_wrapper = True
        elif name == "RewardInNormalState":
            assert opt["normal_state_reward"] is None, "one RewardInNormalState"
            assert opt["episode_stats"] != 2, "statistics between two reward wrappers"
            opt["normal_state_reward"] = kw["reward"]
            opt["normal_state_outside"] = opt["additional_reward"] is not None
            seen_reward_wrapper = True
        elif name == "NormalizeObservation":
            assert not opt["normalize_obs"], "one NormalizeObservation"
            opt["normalize_obs"] = True
        elif name == "RecordEpisodeStatistics":
            later_reward = any(n in ("RewardByBallPosition", "RewardInNormalState") for n, _ in stack[idx + 1:])
            assert not (seen_reward_wrapper and later_reward), "statistics between two reward wrappers"
            opt["episode_stats"] = 2 if seen_reward_wrapper else 1
        else:
            raise ValueError(name)
    return opt


def fusable(wrappers) -> bool:
    """Can the step kernel express this stack as its fused branches (fused_options), or do some of its wrappers have to
    run on the step's outputs (the product's wrapper classes then do that themselves; oracle/wrappers_oracle.py is the
    CPU restatement the tests check them and these fixtures with)?"""
    try:
        fused_options(wrappers)
        return True
    except AssertionError:
        return False
    except ValueError:
        raise


def stack_traits(wrappers) -> dict:
    """What a capture has to know about a stack, fusable or not: the size of the action space, whether observations /
    rewards come out as floats, whether infos carry episode statistics."""
    names = [n for n, _ in wrapper_stack(wrappers)]
    if names.count("SimplifyAction") > 1 or names.count("RecordEpisodeStatistics") > 1:
        raise ValueError("captures take at most one SimplifyAction / RecordEpisodeStatistics")
    return dict(n_actions=13 if "SimplifyAction" in names else 18,
                float_obs="NormalizeObservation" in names,
                float_reward=any(n in ("RewardByBallPosition", "RewardInNormalState") for n in names),
                has_stats="RecordEpisodeStatistics" in names)


def reference_available() -> bool:
    return (REFERENCE_ROOT / "pikazoo" / "env" / "physics.py").exists()


def make_reference_env(seed: int, env_id: int, wrappers: dict | None = None, **kwargs):
    """Construct the unmodified reference env with the Philox stream injected.

    Draw indices 0,1 are consumed by the constructor (physics.py:120-121 -> :218)."""
    install_stubs()
    from pikazoo import pikazoo_v0  # the reference package

    shim = PhiloxShim(seed, env_id)
    _NEXT_SHIM.append(shim)
    raw = pikazoo_v0.env(**kwargs)
    assert raw.np_random is shim and raw.physics.np_random is shim
    assert raw.physics.player1.np_random is shim and raw.physics.player2.np_random is shim
    env = raw
    import pikazoo.wrappers as ref_wrappers

    for name, kw in wrapper_stack(wrappers):
        kw = dict(kw)
        if "additional_reward" in kw:
            kw["additional_reward"] = tuple(kw["additional_reward"])
        env = getattr(ref_wrappers, name)(env, **kw)
    return env, raw, shim


def extract_state(raw, shim) -> np.ndarray:
    """Read the W=44 persistent words from the reference objects' attributes."""
    s = np.zeros(po.W, np.int64)
    for base, pl, kb in ((0, raw.physics.player1, raw.keyboard_array[0]),
                         (po.P_WORDS, raw.physics.player2, raw.keyboard_array[1])):
        s[base + po.P_X] = pl.x
        s[base + po.P_Y] = pl.y
        s[base + po.P_YVEL] = pl.y_velocity
        s[base + po.P_STATE] = pl.state
        s[base + po.P_FRAME] = pl.frame_number
        s[base + po.P_ARM_SWING] = pl.normal_status_arm_swing_direction
        s[base + po.P_DELAY] = pl.delay_before_next_frame
        s[base + po.P_DIVING_DIR] = pl.diving_direction
        s[base + po.P_LYING_DOWN] = pl.lying_down_duration_left
        s[base + po.P_COLLISION] = int(pl.is_collision_with_ball_happened)
        s[base + po.P_BOLDNESS] = pl.computer_boldness
        s[base + po.P_STAND_BY] = pl.computer_where_to_stand_by
        s[base + po.P_HIT_KEY_PREV] = int(kb.power_hit_key_is_down_previous)
    b = raw.physics.ball
    s[po.B_X], s[po.B_Y], s[po.B_XVEL], s[po.B_YVEL] = b.x, b.y, b.x_velocity, b.y_velocity
    s[po.B_POWER_HIT] = int(b.is_power_hit)
    s[po.B_PREV_X], s[po.B_PREV_Y] = b.previous_x, b.previous_y
    s[po.B_PPREV_X], s[po.B_PPREV_Y] = b.previous_previous_x, b.previous_previous_y
    s[po.B_FINE_ROT] = b.fine_rotation
    s[po.B_EXPECTED_X] = b.expected_landing_point_x
    s[po.B_PUNCH_X] = b.punch_effect_x
    s[po.E_SCORE1], s[po.E_SCORE2] = raw.scores
    s[po.E_P2_SERVE] = int(raw.is_player2_serve)
    s[po.E_ROUND_ENDED] = int(raw.round_ended)
    s[po.E_GAME_ENDED] = int(raw.game_ended)
    s[po.E_RNG_COUNTER] = shim.counter
    return s


# --------------------------------------------------------------------------------------
# capture
# --------------------------------------------------------------------------------------
def capture(name: str, lanes: int, steps: int, seed: int, action_seed: int, env_id_base: int,
            env_kwargs: dict, wrappers: dict | None = None, full: bool = True,
            digest_every: int = 0) -> dict:
    """Run `lanes` reference envs for `steps` steps under the random policy.

    full=True stores every state/obs/reward; digest_every>0 stores one 64-bit digest of the
    [W, lanes] state matrix every that many steps instead (long runs for rare branches)."""
    wrappers = wrappers or {}
    traits = stack_traits(wrappers)
    n_actions = traits["n_actions"]
    fused_reward = traits["float_reward"]
    float_obs = traits["float_obs"]
    has_stats = traits["has_stats"]
    envs = [make_reference_env(seed, env_id_base + i, wrappers, **env_kwargs) for i in range(lanes)]
    state_ctor = np.stack([extract_state(raw, shim) for _, raw, shim in envs], axis=1)
    obs_reset = np.zeros((lanes, 2, po.OBS), np.float64 if float_obs else np.int64)
    for i, (env, raw, shim) in enumerate(envs):
        obs, infos = env.reset(seed=1234 + i)  # the reference ignores seed (pikazoo_env.py:149)
        obs_reset[i, 0], obs_reset[i, 1] = obs["player_1"], obs["player_2"]
        assert set(infos) == {"player_1", "player_2"}
    state0 = np.stack([extract_state(raw, shim) for _, raw, shim in envs], axis=1)

    out = dict(state_ctor=state_ctor.astype(np.int32), state0=state0.astype(np.int32),
               obs_reset=obs_reset if float_obs else obs_reset.astype(np.int32))
    if full:
        actions = np.zeros((steps, 2, lanes), np.int32)
        states = np.zeros((steps, po.W, lanes), np.int32)
        obs_all = np.zeros((steps, 2, lanes, po.OBS), np.float64 if float_obs else np.int32)
        ep_r = np.full((steps, 2, lanes), np.nan)   # infos[agent]["episode"]["r"] where present
        ep_l = np.full((steps, lanes), -1, np.int64)  # infos[agent]["episode"]["l"] where present
        rew = np.zeros((steps, 2, lanes), np.float64)
        term = np.zeros((steps, lanes), np.uint8)
    digests = []
    cur = np.zeros((po.W, lanes), np.int32)
    episodes = 0
    for t in range(steps):
        a1, a2 = po.random_actions(lanes, env_id_base, action_seed, t, n_actions)
        for i, (env, raw, shim) in enumerate(envs):
            if not raw.agents:  # harness auto-reset: reset() right before the next step
                env.reset()
            obs, rews, terms, truncs, infos = env.step({"player_1": int(a1[i]), "player_2": int(a2[i])})
            assert not any(truncs.values())
            assert terms["player_1"] == terms["player_2"]
            cur[:, i] = extract_state(raw, shim)
            episodes += int(terms["player_1"])
            if full:
                obs_all[t, 0, i], obs_all[t, 1, i] = obs["player_1"], obs["player_2"]
                rew[t, 0, i], rew[t, 1, i] = rews["player_1"], rews["player_2"]
                term[t, i] = int(terms["player_1"])
                if not fused_reward:
                    assert isinstance(rews["player_1"], int)
                if has_stats:
                    assert ("episode" in infos["player_1"]) == bool(terms["player_1"])
                    if "episode" in infos["player_1"]:
                        ep_r[t, 0, i] = infos["player_1"]["episode"]["r"]
                        ep_r[t, 1, i] = infos["player_2"]["episode"]["r"]
                        ep_l[t, i] = infos["player_1"]["episode"]["l"]
                        assert infos["player_2"]["episode"]["l"] == ep_l[t, i]
        if full:
            actions[t, 0], actions[t, 1] = a1, a2
            states[t] = cur
        if digest_every and (t + 1) % digest_every == 0:
            digests.append(po.digest(cur))
    meta = dict(name=name, lanes=lanes, steps=steps, seed=seed, action_seed=action_seed,
                env_id_base=env_id_base, env_kwargs=env_kwargs, wrappers=wrappers,
                n_actions=n_actions, episodes=episodes, digest_every=digest_every,
                fields=po.FIELD_NAMES)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    if full:
        out.update(actions=actions.astype(np.uint8), states=states.astype(np.int16),
                   rng_counter=states[:, po.E_RNG_COUNTER, :].astype(np.int32),
                   obs=(obs_all if float_obs else obs_all.astype(np.int16)), term=term,
                   rew=(rew if fused_reward else rew.astype(np.int8)))
        if has_stats:
            out.update(ep_r=ep_r, ep_l=ep_l.astype(np.int32))
        assert np.array_equal(out["states"].astype(np.int32)[:, :po.E_RNG_COUNTER],
                              states[:, :po.E_RNG_COUNTER])  # int16 is lossless here
    if digest_every:
        out["digests"] = np.array(digests, dtype=np.uint64)
        out["final_state"] = cur.copy()
    return out


def capture_single_agent(name: str, lanes: int, steps: int, seed: int, action_seed: int, opponent_seed: int,
                         env_id_base: int, side: str, env_kwargs: dict) -> dict:
    """The reference's ``ConvertSingleAgent`` (wrappers/convert_single_agent.py) around the unmodified env: the
    controlled side plays policy stream `action_seed`, the other side's ``action_space(other).sample()`` is fed policy
    stream `opponent_seed` (word 0 / 1 of the draw for player_1 / player_2 -- what the product's wrapper draws on
    device).  Stores what the wrapper returns for the controlled side at every step."""
    install_stubs()
    import pikazoo.wrappers as ref_wrappers

    me = 0 if side == "player_1" else 1
    envs = []
    for i in range(lanes):
        env, raw, shim = make_reference_env(seed, env_id_base + i, None, **env_kwargs)
        envs.append((ref_wrappers.ConvertSingleAgent(env, side), raw, shim))
    obs_reset = np.zeros((lanes, po.OBS), np.int32)
    for i, (env, raw, shim) in enumerate(envs):
        o, info = env.reset()
        obs_reset[i] = o
        assert list(info) == ["score"]
    actions = np.zeros((steps, lanes), np.int32)
    sampled = np.zeros((steps, lanes), np.int32)
    obs = np.zeros((steps, lanes, po.OBS), np.int32)
    rew = np.zeros((steps, lanes), np.int32)
    term = np.zeros((steps, lanes), np.uint8)
    score = np.zeros((steps, lanes, 2), np.int32)
    states = np.zeros((steps, po.W, lanes), np.int32)
    for t in range(steps):
        own = po.random_actions(lanes, env_id_base, action_seed, t, 18)[me]
        opp = po.random_actions(lanes, env_id_base, opponent_seed, t, 18)[1 - me]
        for i, (env, raw, shim) in enumerate(envs):
            if not raw.agents:
                env.reset()
            _NEXT_SAMPLES.append(int(opp[i]))
            o, r, te, tr, info = env.step(int(own[i]))
            assert not _NEXT_SAMPLES and tr is False
            obs[t, i], rew[t, i], term[t, i], score[t, i] = o, r, int(te), info["score"]
            states[t, :, i] = extract_state(raw, shim)
        actions[t], sampled[t] = own, opp
    meta = dict(name=name, lanes=lanes, steps=steps, seed=seed, action_seed=action_seed, opponent_seed=opponent_seed,
                env_id_base=env_id_base, side=side, env_kwargs=env_kwargs)
    return dict(meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), obs_reset=obs_reset,
                actions=actions.astype(np.uint8), sampled=sampled.astype(np.uint8), obs=obs.astype(np.int16),
                rew=rew.astype(np.int8), term=term, score=score.astype(np.int8), states=states.astype(np.int16),
                rng_counter=states[:, po.E_RNG_COUNTER, :].astype(np.int32))


TEST_TABLE = (0.0, -0.01, 0.0, 0.01, 0.0, 0.01, 0.0, -0.01)
INT_TABLE = (1, -2, 3, -4, 5, -6, 7, -8)

# --------------------------------------------------------------------------------------
# render(): the reference's draw list and what it does to the env RNG (render_mode="rgb_array")
# --------------------------------------------------------------------------------------
# 10 clouds x (x, y, x velocity, size_diff_turn_number), wave vertical_coord, its velocity, 27 y_coords,
# ball.punch_effect_radius, ball.punch_effect_y
SCENERY_WORDS = 71
# static part of a frame: draw_background (pikazoo_env.py:296-325), identical on every frame
BACKGROUND_BLITS = 12 * 27 + 1 + 27 + 25 + 2 + 2 * 27 + 1 + 12


def extract_scenery(raw) -> np.ndarray:
    """cloud_array / wave_ of the reference (cloud_and_wave.py:12-50) as SCENERY_WORDS ints."""
    out = np.zeros(SCENERY_WORDS, np.int32)
    for i, c in enumerate(raw.cloud_array):
        out[4 * i:4 * i + 4] = (c.top_left_point_x, c.top_left_point_y, c.top_left_point_x_velocity,
                                c.size_diff_turn_number)
    out[40], out[41] = raw.wave_.vertical_coord, raw.wave_.vertical_coord_velocity
    out[42:69] = raw.wave_.y_coords
    out[69], out[70] = raw.physics.ball.punch_effect_radius, raw.physics.ball.punch_effect_y
    return out


def capture_render(name: str, steps: int, periods, seed: int, action_seed: int, env_id_base: int, env_kwargs: dict) -> dict:
    """One reference env per entry of `periods`, constructed with render_mode="rgb_array" (the constructor then draws
    the ten clouds from the env RNG, pikazoo_env.py:475-477), stepped under the random policy; lane i calls the
    reference's own ``render()`` after reset and after every periods[i]-th step.  Recorded per frame: the draw list the
    reference issued (file, mirrored, x, y, width, height of every blit behind the static background), the clouds /
    wave after it, and the 44 state words after it (render() advances the env RNG, so later steps differ from an
    un-rendered run).  ``render()`` itself returns a blank array here: the stand-in pygame composes nothing."""
    lanes = len(periods)
    kw = dict(env_kwargs, render_mode="rgb_array")
    envs = [make_reference_env(seed, env_id_base + i, None, **kw) for i in range(lanes)]
    names: list = []
    frames = []   # (lane, step (-1: after reset), entries)

    def render(i, t):
        env, raw, shim = envs[i]
        if raw.screen is not None:
            raw.screen.blits.clear()
        out = raw.render()
        assert out.shape == (304, 432, 3)
        blits = raw.screen.blits
        static, dynamic = blits[:BACKGROUND_BLITS], blits[BACKGROUND_BLITS:]
        if not frames:
            frames_static.extend(static)
        assert static == frames_static, "draw_background changed between frames"
        entries = []
        for (file, flip, scaled), x, y, w, h in dynamic:
            if file not in names:
                names.append(file)
            entries.append((names.index(file), int(flip), x, y, w, h))
        frames.append((i, t, entries, extract_scenery(raw), extract_state(raw, shim)))

    frames_static: list = []
    state_ctor = np.stack([extract_state(raw, shim) for _, raw, shim in envs], axis=1)
    scenery_ctor = np.stack([extract_scenery(raw) for _, raw, _ in envs], axis=1)
    for i, (env, raw, shim) in enumerate(envs):
        env.reset()
    state0 = np.stack([extract_state(raw, shim) for _, raw, shim in envs], axis=1)
    for i in range(lanes):
        render(i, -1)
    states = np.zeros((steps, po.W, lanes), np.int32)   # after step t (and before a render that follows it)
    for t in range(steps):
        a1, a2 = po.random_actions(lanes, env_id_base, action_seed, t, 18)
        for i, (env, raw, shim) in enumerate(envs):
            if not raw.agents:
                env.reset()
            env.step({"player_1": int(a1[i]), "player_2": int(a2[i])})
            states[t, :, i] = extract_state(raw, shim)
            if (t + 1) % periods[i] == 0:
                render(i, t)
    most = max(len(f[2]) for f in frames)
    draw = np.full((len(frames), most, 6), -1, np.int16)
    for k, f in enumerate(frames):
        draw[k, :len(f[2])] = np.asarray(f[2], np.int16).reshape(-1, 6)
    static = []
    for (file, flip, scaled), x, y, w, h in frames_static:
        assert not flip and scaled is None
        if file not in names:
            names.append(file)
        static.append((names.index(file), x, y, w, h))
    meta = dict(name=name, lanes=lanes, steps=steps, periods=list(periods), seed=seed, action_seed=action_seed,
                env_id_base=env_id_base, env_kwargs=env_kwargs, files=names, fields=po.FIELD_NAMES)
    return dict(
        state_ctor=state_ctor.astype(np.int32), scenery_ctor=scenery_ctor, state0=state0.astype(np.int32), states=states,
        frame_lane=np.asarray([f[0] for f in frames], np.int32), frame_step=np.asarray([f[1] for f in frames], np.int32),
        frame_count=np.asarray([len(f[2]) for f in frames], np.int32), frame_draw=draw,
        frame_scenery=np.stack([f[3] for f in frames]), frame_state=np.stack([f[4] for f in frames]).astype(np.int32),
        background=np.asarray(static, np.int16), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))


def capture_planted(name: str, seed: int, action_seed: int, env_id_base: int, env_kwargs: dict, frames: int = 16) -> dict:
    """Corners random play practically never reaches: one reference env per planted ball state (very fast balls over
    the net top / at the walls / at the ceiling, in both directions), stepped `frames` frames under the random policy.
    The planted values are written into the reference's own ball attributes after 30 frames of normal play; recorded
    like the trajectory fixtures (state words after every step)."""
    cases = []
    for x in (20, 194, 216, 238, 432):
        for y in (0, 100, 177, 190, 192, 193, 252):
            for xv, yv in ((13, 200), (-20, 177), (7, -250), (0, 292), (20, 60), (-10, -96), (1, 1)):
                cases.append((x, y, xv, yv))
    lanes = len(cases)
    envs = [make_reference_env(seed, env_id_base + i, None, **env_kwargs) for i in range(lanes)]
    for env, raw, shim in envs:
        env.reset()
    warm = 30
    for t in range(warm):
        a1, a2 = po.random_actions(lanes, env_id_base, action_seed, t, 18)
        for i, (env, raw, shim) in enumerate(envs):
            env.step({"player_1": int(a1[i]), "player_2": int(a2[i])})
    for (x, y, xv, yv), (env, raw, shim) in zip(cases, envs):
        b = raw.physics.ball
        b.x, b.y, b.x_velocity, b.y_velocity = x, y, xv, yv
    planted = np.stack([extract_state(raw, shim) for _, raw, shim in envs], axis=1).astype(np.int32)
    states = np.zeros((frames, po.W, lanes), np.int32)
    for t in range(frames):
        a1, a2 = po.random_actions(lanes, env_id_base, action_seed, warm + t, 18)
        for i, (env, raw, shim) in enumerate(envs):
            if not raw.agents:
                env.reset()
            env.step({"player_1": int(a1[i]), "player_2": int(a2[i])})
            states[t, :, i] = extract_state(raw, shim)
    meta = dict(name=name, lanes=lanes, frames=frames, warm=warm, seed=seed, action_seed=action_seed,
                env_id_base=env_id_base, env_kwargs=env_kwargs, cases=cases, fields=po.FIELD_NAMES)
    return dict(planted=planted, states=states, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))


def capture_planted_random(name: str, lanes: int, seed: int, action_seed: int, env_id_base: int, env_kwargs: dict,
                           frames: int = 12, plant_seed: int = 123) -> dict:
    """The same with every attribute of both players, the ball and the scores drawn at random over its whole valid
    range (reachable or not: the reference's code is defined for all of them), half of the balls next to a player."""
    rng = np.random.default_rng(plant_seed)
    envs = [make_reference_env(seed, env_id_base + i, None, **env_kwargs) for i in range(lanes)]
    for env, raw, shim in envs:
        env.reset()
        for pl, lo, hi in ((raw.physics.player1, 32, 184), (raw.physics.player2, 248, 400)):
            pl.x = int(rng.integers(lo, hi + 1))
            pl.state = int(rng.integers(0, 5))
            # (y, y_velocity) from the pairs a jump or a dive passes through: a player's y is the one attribute a frame
            # moves without clamping, so an unreachable pair could leave the court
            if pl.state in (0, 4) and rng.random() < 0.7:
                pl.y, pl.y_velocity = 244, 0
            else:
                y, v = 244, (-16 if rng.random() < 0.7 else -5)
                for _ in range(int(rng.integers(0, 33))):
                    if y + v > 244:
                        break
                    y, v = y + v, v + 1
                pl.y, pl.y_velocity = y, v
            pl.frame_number = int(rng.integers(0, 5))
            pl.normal_status_arm_swing_direction = int(rng.choice([-1, 1]))
            pl.delay_before_next_frame = int(rng.integers(0, 6))
            pl.diving_direction = int(rng.integers(-1, 2))
            pl.lying_down_duration_left = int(rng.integers(-1, 4))
            pl.is_collision_with_ball_happened = bool(rng.integers(0, 2))
            pl.computer_boldness = int(rng.integers(0, 5))
            pl.computer_where_to_stand_by = int(rng.integers(0, 2))
        b = raw.physics.ball
        near = rng.random() < 0.5
        p = raw.physics.player1 if rng.random() < 0.5 else raw.physics.player2
        b.x = int(np.clip(p.x + rng.integers(-40, 41), 20, 432)) if near else int(rng.integers(20, 433))
        b.y = int(np.clip(p.y + rng.integers(-40, 41), 0, 252)) if near else int(rng.integers(0, 253))
        b.x_velocity = int(rng.integers(-20, 21))
        b.y_velocity = int(rng.integers(-120, 121) if rng.random() < 0.8 else rng.integers(-300, 301))
        b.is_power_hit = bool(rng.integers(0, 2))
        b.fine_rotation = int(rng.integers(0, 51))
        b.rotation = b.fine_rotation // 10
        for kb in raw.keyboard_array:
            kb.power_hit_key_is_down_previous = bool(rng.integers(0, 2))
        top = env_kwargs.get("winning_score", 15)
        raw.scores = [int(rng.integers(0, top)), int(rng.integers(0, top))]
    planted = np.stack([extract_state(raw, shim) for _, raw, shim in envs], axis=1).astype(np.int32)
    states = np.zeros((frames, po.W, lanes), np.int32)
    for t in range(frames):
        a1, a2 = po.random_actions(lanes, env_id_base, action_seed, t, 18)
        for i, (env, raw, shim) in enumerate(envs):
            if not raw.agents:
                env.reset()
            env.step({"player_1": int(a1[i]), "player_2": int(a2[i])})
            states[t, :, i] = extract_state(raw, shim)
    meta = dict(name=name, lanes=lanes, frames=frames, warm=0, seed=seed, action_seed=action_seed,
                env_id_base=env_id_base, env_kwargs=env_kwargs, cases=[[] for _ in range(lanes)], fields=po.FIELD_NAMES)
    return dict(planted=planted, states=states, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))


PLANTED_RANDOM_RUNS = [
    ("planted_random_states_human", dict(winning_score=3)),
    ("planted_random_states_both_computer", dict(winning_score=3, is_player1_computer=True, is_player2_computer=True)),
    ("planted_random_states_p2_computer_random_serve", dict(winning_score=2, is_player2_computer=True, serve="random")),
]


PLANTED_RUNS = [
    ("planted_fast_balls_human", dict(winning_score=15)),
    ("planted_fast_balls_both_computer", dict(winning_score=15, is_player1_computer=True, is_player2_computer=True)),
]


RENDER_RUNS = [
    # name, steps, render periods per lane, env kwargs
    ("render_human_human", 1500, (1, 2, 3, 7, 25, 40), dict(winning_score=15)),
    ("render_p2_computer", 1200, (1, 4, 30, 30), dict(winning_score=15, is_player2_computer=True, serve="alternate")),
]


FIXTURES = [
    # name, lanes, steps, env_kwargs, wrappers
    ("cfg2_human_human", 6, 3000, dict(winning_score=15, serve="winner"), None),
    ("cfg3_p2_computer", 6, 3000, dict(winning_score=15, serve="winner", is_player2_computer=True), None),
    ("p1_computer", 4, 2000, dict(winning_score=15, serve="winner", is_player1_computer=True), None),
    ("both_computer", 4, 3000, dict(winning_score=15, is_player1_computer=True, is_player2_computer=True), None),
    ("serve_alternate", 4, 2000, dict(winning_score=15, serve="alternate"), None),
    ("serve_random", 4, 2000, dict(winning_score=15, serve="random", is_player2_computer=True), None),
    ("winning_score_1", 4, 1000, dict(winning_score=1, serve="winner"), None),
    ("winning_score_3", 4, 1500, dict(winning_score=3, serve="alternate", is_player2_computer=True), None),
    ("cfg5_wrappers_float", 6, 3000, dict(winning_score=15, serve="winner"),
     dict(simplify_action=True, additional_reward=TEST_TABLE, x_line=216, y_line=176)),
    ("wrappers_int_table", 4, 1500, dict(winning_score=5, serve="winner", is_player2_computer=True),
     dict(simplify_action=True, additional_reward=INT_TABLE, x_line=200, y_line=150)),
    ("simplify_only", 4, 1500, dict(winning_score=15, serve="winner"), dict(simplify_action=True)),
]

FIXTURES += [
    ("normal_state_only", 4, 1500, dict(winning_score=3, serve="winner"),
     dict(stack=[["RewardInNormalState", dict(reward=-0.001)]])),
    ("normal_state_inside_ballpos", 4, 1500, dict(winning_score=3, is_player2_computer=True),
     dict(stack=[["RewardInNormalState", dict(reward=0.25)],
                 ["RewardByBallPosition", dict(additional_reward=list(TEST_TABLE), x_line=216, y_line=176)]])),
    ("normal_state_outside_ballpos", 4, 1500, dict(winning_score=3, serve="alternate"),
     dict(stack=[["RewardByBallPosition", dict(additional_reward=list(TEST_TABLE), x_line=216, y_line=176)],
                 ["RewardInNormalState", dict(reward=0.5)]])),
    ("normalize_observation", 4, 800, dict(winning_score=2, is_player2_computer=True),
     dict(stack=[["NormalizeObservation", {}]])),
    ("record_stats_raw", 4, 2500, dict(winning_score=2, serve="winner"),
     dict(stack=[["RecordEpisodeStatistics", {}]])),
    ("full_wrapper_stack", 4, 2500, dict(winning_score=2, serve="random", is_player2_computer=True),
     dict(stack=[["SimplifyAction", {}],
                 ["RewardInNormalState", dict(reward=-0.002)],
                 ["RewardByBallPosition", dict(additional_reward=list(TEST_TABLE), x_line=216, y_line=176)],
                 ["NormalizeObservation", {}],
                 ["RecordEpisodeStatistics", {}]])),
]

# stacks the step kernel cannot fuse (the reference composes its wrappers in any order): the product's wrapper classes
# apply what cannot be a kernel branch on the step's outputs; the CPU side is oracle/wrappers_oracle.py
UNFUSED_FIXTURES = [
    # RewardByBallPosition ABOVE NormalizeObservation: it compares the normalized obs[26], obs[27] with the lines
    # (reward_by_ball_position.py:22-24), i.e. every step lands in zone 0
    ("unfused_ballpos_above_normalize", 4, 1200, dict(winning_score=2, serve="winner"),
     dict(stack=[["NormalizeObservation", {}],
                 ["RewardByBallPosition", dict(additional_reward=list(TEST_TABLE), x_line=216, y_line=176)]])),
    # RecordEpisodeStatistics BETWEEN two reward wrappers: its sums see the first one only
    ("unfused_stats_between_reward_wrappers", 4, 1500, dict(winning_score=2, is_player2_computer=True),
     dict(stack=[["RewardByBallPosition", dict(additional_reward=list(TEST_TABLE), x_line=216, y_line=176)],
                 ["RecordEpisodeStatistics", {}],
                 ["RewardInNormalState", dict(reward=0.5)]])),
    # two RewardByBallPosition, two RewardInNormalState, NormalizeObservation twice (the second one's bounds are the
    # first one's Box(0, 1): the identity), SimplifyAction below it all
    ("unfused_doubled_wrappers", 4, 1200, dict(winning_score=2, serve="alternate"),
     dict(stack=[["SimplifyAction", {}],
                 ["RewardByBallPosition", dict(additional_reward=list(TEST_TABLE), x_line=216, y_line=176)],
                 ["RewardByBallPosition", dict(additional_reward=list(INT_TABLE), x_line=200, y_line=150)],
                 ["RewardInNormalState", dict(reward=-0.002)],
                 ["NormalizeObservation", {}],
                 ["NormalizeObservation", {}],
                 ["RewardInNormalState", dict(reward=0.25)],
                 ["RecordEpisodeStatistics", {}]])),
]

FIXTURES += [
    # BASELINE.json configs[0]: one env, pikazoo_v0.env() defaults, random actions, 10 000 steps
    ("cfg1_one_env_10k", 1, 10000, dict(), None),
]

SINGLE_AGENT_RUNS = [
    # name, lanes, steps, side, env_kwargs
    ("single_agent_player_2", 4, 1200, "player_2", dict(winning_score=2, serve="winner")),
    ("single_agent_player_1_vs_computer", 4, 1200, "player_1", dict(winning_score=2, is_player2_computer=True)),
]

DIGEST_RUNS = [
    ("digest_human_human", 48, 20000, dict(winning_score=15, serve="winner"), None),
    ("digest_p2_computer", 48, 20000, dict(winning_score=15, serve="winner", is_player2_computer=True), None),
    ("digest_both_computer_random_serve", 32, 20000,
     dict(winning_score=5, serve="random", is_player1_computer=True, is_player2_computer=True), None),
]


def main(argv=None):
    import argparse
    import time

    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--skip-digests", action="store_true")
    args = ap.parse_args(argv)
    assert reference_available(), "the reference is only mounted in the build container"
    GOLDEN.mkdir(parents=True, exist_ok=True)
    for k, (name, lanes, steps, kw, wr) in enumerate(FIXTURES):
        if args.only and args.only != name:
            continue
        t0 = time.time()
        data = capture(name, lanes, steps, seed=20241008 + k, action_seed=77 + k,
                       env_id_base=1000 * k, env_kwargs=kw, wrappers=wr, full=True)
        np.savez_compressed(GOLDEN / f"{name}.npz", **data)
        meta = json.loads(bytes(data["meta"]).decode())
        print(f"{name}: {lanes}x{steps} episodes={meta['episodes']} "
              f"{(GOLDEN / (name + '.npz')).stat().st_size / 1e3:.0f} kB {time.time() - t0:.1f}s")
    for k, (name, lanes, steps, kw, wr) in enumerate(UNFUSED_FIXTURES):
        if args.only and args.only != name:
            continue
        assert not fusable(wr), name
        data = capture(name, lanes, steps, seed=31337 + k, action_seed=140 + k, env_id_base=30000 + 1000 * k, env_kwargs=kw,
                       wrappers=wr, full=True)
        np.savez_compressed(GOLDEN / f"{name}.npz", **data)
        meta = json.loads(bytes(data["meta"]).decode())
        print(f"{name}: {lanes}x{steps} episodes={meta['episodes']} "
              f"{(GOLDEN / (name + '.npz')).stat().st_size / 1e3:.0f} kB")
    for k, (name, lanes, steps, side, kw) in enumerate(SINGLE_AGENT_RUNS):
        if args.only and args.only != name:
            continue
        data = capture_single_agent(name, lanes, steps, seed=4242 + k, action_seed=31 + k, opponent_seed=900 + k,
                                    env_id_base=7000 + 10 * k, side=side, env_kwargs=kw)
        np.savez_compressed(GOLDEN / f"{name}.npz", **data)
        print(f"{name}: {lanes}x{steps} terminations={int(data['term'].sum())} "
              f"{(GOLDEN / (name + '.npz')).stat().st_size / 1e3:.0f} kB")
    for k, (name, kw) in enumerate(PLANTED_RUNS):
        if args.only and args.only != name:
            continue
        data = capture_planted(name, seed=8080 + k, action_seed=51 + k, env_id_base=12000 + 1000 * k, env_kwargs=kw)
        np.savez_compressed(GOLDEN / f"{name}.npz", **data)
        print(f"{name}: {data['planted'].shape[1]} planted states x {data['states'].shape[0]} frames, min ball y "
              f"{int(data['states'][:, 27].min())} {(GOLDEN / (name + '.npz')).stat().st_size / 1e3:.0f} kB")
    for k, (name, kw) in enumerate(PLANTED_RANDOM_RUNS):
        if args.only and args.only != name:
            continue
        data = capture_planted_random(name, 800, seed=9090 + k, action_seed=61 + k, env_id_base=20000 + 1000 * k,
                                      env_kwargs=kw, plant_seed=321 + k)
        np.savez_compressed(GOLDEN / f"{name}.npz", **data)
        print(f"{name}: {data['planted'].shape[1]} planted states x {data['states'].shape[0]} frames "
              f"{(GOLDEN / (name + '.npz')).stat().st_size / 1e3:.0f} kB")
    for k, (name, steps, periods, kw) in enumerate(RENDER_RUNS):
        if args.only and args.only != name:
            continue
        data = capture_render(name, steps, periods, seed=606 + k, action_seed=17 + k, env_id_base=9000 + 100 * k,
                              env_kwargs=kw)
        np.savez_compressed(GOLDEN / f"{name}.npz", **data)
        print(f"{name}: {len(periods)} lanes x {steps} steps, {len(data['frame_lane'])} frames "
              f"{(GOLDEN / (name + '.npz')).stat().st_size / 1e3:.0f} kB")
    if not args.skip_digests:
        for k, (name, lanes, steps, kw, wr) in enumerate(DIGEST_RUNS):
            if args.only and args.only != name:
                continue
            t0 = time.time()
            data = capture(name, lanes, steps, seed=555 + k, action_seed=999 + k,
                           env_id_base=(1 << 33) + 64 * k if k == 2 else 50000 * (k + 1),
                           env_kwargs=kw, wrappers=wr, full=False, digest_every=500)
            np.savez_compressed(GOLDEN / f"{name}.npz", **data)
            meta = json.loads(bytes(data["meta"]).decode())
            print(f"{name}: {lanes}x{steps} episodes={meta['episodes']} {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()

"""ctypes/numpy front end of the CPU parity oracle (TEST INFRASTRUCTURE, not product).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  It loads ``oracle/_build/libpz_oracle.so`` (built from ``pz_oracle.c`` by
``oracle/Makefile``) and exposes the batched entry points on numpy arrays.

State layout: ``int32[W=44, n]`` field-major; see ``pz_oracle.h``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libpz_oracle.so"
if os.environ.get("PZ_ORACLE_SANITIZED") == "1":  # tests: the -fsanitize=address,undefined build (oracle/Makefile: asan)
    _SO = _HERE / "_build" / "libpz_oracle_asan.so"

W = 44
OBS = 35
AGENTS = ("player_1", "player_2")
SERVE_MODES = {"winner": 0, "alternate": 1, "random": 2}

# field indices (pz_oracle.h)
P_X, P_Y, P_YVEL, P_STATE, P_FRAME, P_ARM_SWING, P_DELAY, P_DIVING_DIR, P_LYING_DOWN, \
    P_COLLISION, P_BOLDNESS, P_STAND_BY, P_HIT_KEY_PREV = range(13)
P_WORDS = 13
B_X, B_Y, B_XVEL, B_YVEL, B_POWER_HIT, B_PREV_X, B_PREV_Y, B_PPREV_X, B_PPREV_Y, \
    B_FINE_ROT, B_EXPECTED_X, B_PUNCH_X = range(26, 38)
E_SCORE1, E_SCORE2, E_P2_SERVE, E_ROUND_ENDED, E_GAME_ENDED, E_RNG_COUNTER = range(38, 44)

FIELD_NAMES = (
    [f"p1.{n}" for n in ("x", "y", "y_velocity", "state", "frame_number", "arm_swing", "delay",
                         "diving_direction", "lying_down", "collision", "boldness", "stand_by",
                         "hit_key_prev")]
    + [f"p2.{n}" for n in ("x", "y", "y_velocity", "state", "frame_number", "arm_swing", "delay",
                           "diving_direction", "lying_down", "collision", "boldness", "stand_by",
                           "hit_key_prev")]
    + [f"ball.{n}" for n in ("x", "y", "x_velocity", "y_velocity", "is_power_hit", "previous_x",
                             "previous_y", "previous_previous_x", "previous_previous_y",
                             "fine_rotation", "expected_landing_point_x", "punch_effect_x")]
    + ["score1", "score2", "is_player2_serve", "round_ended", "game_ended", "rng_counter"]
)
assert len(FIELD_NAMES) == W


class Config(C.Structure):
    """Mirror of ``pzo_config`` (pz_oracle.h); same byte layout as the product's pz_config."""

    _fields_ = [
        ("winning_score", C.c_int32),
        ("serve_mode", C.c_int32),
        ("p1_computer", C.c_int32),
        ("p2_computer", C.c_int32),
        ("simplify_action", C.c_int32),
        ("ballpos_reward", C.c_int32),
        ("x_line", C.c_int32),
        ("y_line", C.c_int32),
        ("additional_reward", C.c_float * 8),
        ("auto_reset", C.c_int32),
        ("packed_state", C.c_int32),  # storage format of the product; ignored here
        ("normal_state_mode", C.c_int32),
        ("normal_state_reward", C.c_float),
        ("normalize_obs", C.c_int32),
        ("episode_stats_mode", C.c_int32),
        ("seed", C.c_uint64),
        ("env_id_base", C.c_int64),
        ("action_faults", C.c_void_p),  # product-side diagnostic pointer (layout only; never read by the oracle)
        ("action_format", C.c_int32),   # product-side (ABI 10; layout only: the oracle takes int32 actions)
        ("reserved0", C.c_int32),
    ]


def make_config(winning_score=15, serve="winner", is_player1_computer=False,
                is_player2_computer=False, simplify_action=False, additional_reward=None,
                x_line=216, y_line=176, auto_reset=True, seed=0, env_id_base=0,
                normal_state_reward=None, normal_state_outside=False, normalize_obs=False,
                episode_stats=0) -> Config:
    """normal_state_reward: RewardInNormalState's constant (None = off); normal_state_outside: that
    wrapper sits outside RewardByBallPosition; episode_stats: 0 off, 1 RecordEpisodeStatistics
    directly on the env's rewards, 2 on the fully wrapped rewards."""
    cfg = Config()
    cfg.winning_score = int(winning_score)
    cfg.serve_mode = SERVE_MODES[serve]
    cfg.p1_computer = int(bool(is_player1_computer))
    cfg.p2_computer = int(bool(is_player2_computer))
    cfg.simplify_action = int(bool(simplify_action))
    cfg.ballpos_reward = int(additional_reward is not None)
    cfg.x_line = int(x_line)
    cfg.y_line = int(y_line)
    if additional_reward is not None:
        assert len(additional_reward) == 8
        for i, v in enumerate(additional_reward):
            cfg.additional_reward[i] = float(v)
    cfg.auto_reset = int(bool(auto_reset))
    cfg.normal_state_mode = 0 if normal_state_reward is None else (2 if normal_state_outside else 1)
    cfg.normal_state_reward = 0.0 if normal_state_reward is None else float(normal_state_reward)
    cfg.normalize_obs = int(bool(normalize_obs))
    cfg.episode_stats_mode = int(episode_stats)
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    cfg.env_id_base = int(env_id_base)
    return cfg


def build(force: bool = False) -> Path:
    """Compile the C oracle with gcc (no-op when the .so is newer than its sources)."""
    srcs = [_HERE / "pz_oracle.c", _HERE / "pz_oracle.h"]
    if force or not _SO.exists() or any(s.stat().st_mtime > _SO.stat().st_mtime for s in srcs):
        target = "asan" if _SO.name.endswith("_asan.so") else "all"
        subprocess.check_call(["make", "-s", "-C", str(_HERE)] + (["-B"] if force else []) + [target],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_SO))
        i32p, u8p, vp = C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_void_p
        L.pzo_philox4x32_10.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
        L.pzo_philox4x32_10.restype = None
        L.pzo_env_draw.argtypes = [C.c_uint64, C.c_int64, C.c_uint32, C.c_uint32]
        L.pzo_env_draw.restype = C.c_int32
        L.pzo_random_actions.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int32]
        L.pzo_random_actions.restype = None
        L.pzo_init.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(Config)]
        L.pzo_init.restype = None
        L.pzo_reset.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(Config), vp, vp, vp, vp]
        L.pzo_reset.restype = None
        L.pzo_step.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(Config), vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
        L.pzo_step.restype = None
        L.pzo_rollout_random.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(Config), C.c_uint64,
                                         C.c_uint64, C.c_int32, vp, vp, vp, vp, vp, vp,
                                         C.POINTER(C.c_int64), C.c_int]
        L.pzo_rollout_random.restype = None
        L.pzo_observe.argtypes = [vp, C.c_int64, C.c_int64, C.c_int32, vp, vp]
        L.pzo_observe.restype = None
        L.pzo_digest.argtypes = [vp, C.c_int64, C.c_int64]
        L.pzo_digest.restype = C.c_uint64
        L.pzo_expected_landing_x.argtypes = [C.c_int32] * 4
        L.pzo_expected_landing_x.restype = C.c_int32
        L.pzo_expected_landing_x_power_hit.argtypes = [C.c_int32] * 6
        L.pzo_expected_landing_x_power_hit.restype = C.c_int32
        del i32p, u8p
        _lib = L
    return _lib


def _p(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def philox4x32_10(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().pzo_philox4x32_10(*[int(c) & 0xFFFFFFFF for c in ctr], *[int(k) & 0xFFFFFFFF for k in key], out)
    return tuple(int(v) for v in out)


def philox4x32_10_numpy(ctr, key):
    """Independent vectorised numpy Philox4x32-10 (cross-check of the C one)."""
    c = [np.asarray(x, dtype=np.uint64) & np.uint64(0xFFFFFFFF) for x in ctr]
    k0, k1 = (int(key[0]) & 0xFFFFFFFF), (int(key[1]) & 0xFFFFFFFF)
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & m32, p1 & m32,
             ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & m32, p0 & m32]
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in c]


def env_draw(seed: int, env_id: int, idx: int, n: int) -> int:
    return int(lib().pzo_env_draw(int(seed) & 0xFFFFFFFFFFFFFFFF, int(env_id), int(idx) & 0xFFFFFFFF, int(n)))


def random_actions(n: int, env_id_base: int, action_seed: int, t: int, n_actions: int = 18):
    a1 = np.empty(n, np.int32)
    a2 = np.empty(n, np.int32)
    lib().pzo_random_actions(_p(a1), _p(a2), n, env_id_base, action_seed, t, n_actions)
    return a1, a2


class OracleEnv:
    """Batched CPU env with the oracle's semantics; numpy in/out."""

    def __init__(self, num_envs: int, cfg: Config, nthreads: int = 1):
        self.n = int(num_envs)
        self.cfg = cfg
        self.nthreads = nthreads
        self.state = np.zeros((W, self.n), np.int32)
        odt = np.float32 if cfg.normalize_obs else np.int32
        self.obs = [np.zeros((self.n, OBS), odt) for _ in range(2)]
        self.float_rewards = bool(cfg.ballpos_reward or cfg.normal_state_mode)
        rdt = np.float32 if self.float_rewards else np.int32
        self.rew = [np.zeros(self.n, rdt) for _ in range(2)]
        self.term = np.zeros(self.n, np.uint8)
        # RecordEpisodeStatistics buffer: float64[2][n] running returns, then int32[n] lengths (20 n bytes)
        self.stats = np.zeros(20 * self.n, np.uint8) if cfg.episode_stats_mode else None
        lib().pzo_init(_p(self.state), self.n, self.n, C.byref(cfg))

    def _stats_ptr(self):
        return None if self.stats is None else _p(self.stats)

    @property
    def episode_returns(self):
        """float64 [2, n]: running returns of player 1 / player 2 (view of the statistics buffer)."""
        return self.stats[:16 * self.n].view(np.float64).reshape(2, self.n)

    @property
    def episode_lengths(self):
        return self.stats[16 * self.n:].view(np.int32)

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().pzo_reset(_p(self.state), self.n, self.n, C.byref(self.cfg),
                        None if m is None else _p(m), _p(self.obs[0]), _p(self.obs[1]), self._stats_ptr())
        return self.obs[0], self.obs[1]

    def step(self, a1, a2):
        a1 = np.ascontiguousarray(a1, np.int32)
        a2 = np.ascontiguousarray(a2, np.int32)
        lib().pzo_step(_p(self.state), self.n, self.n, C.byref(self.cfg), _p(a1), _p(a2),
                       _p(self.obs[0]), _p(self.obs[1]), _p(self.rew[0]), _p(self.rew[1]),
                       _p(self.term), self._stats_ptr(), self.nthreads)
        return self.obs, self.rew, self.term

    def rollout_random(self, action_seed: int, t0: int, k: int) -> int:
        fin = C.c_int64(0)
        lib().pzo_rollout_random(_p(self.state), self.n, self.n, C.byref(self.cfg), action_seed, t0, k,
                                 _p(self.obs[0]), _p(self.obs[1]), _p(self.rew[0]), _p(self.rew[1]),
                                 _p(self.term), self._stats_ptr(), C.byref(fin), self.nthreads)
        return int(fin.value)

    def observe(self):
        odt = np.float32 if self.cfg.normalize_obs else np.int32
        o1 = np.zeros((self.n, OBS), odt)
        o2 = np.zeros((self.n, OBS), odt)
        lib().pzo_observe(_p(self.state), self.n, self.n, int(self.cfg.normalize_obs), _p(o1), _p(o2))
        return o1, o2

    def digest(self) -> int:
        return int(lib().pzo_digest(_p(self.state), self.n, self.n))


def digest(state: np.ndarray) -> int:
    state = np.ascontiguousarray(state, np.int32)
    assert state.shape[0] == W
    return int(lib().pzo_digest(_p(state), state.shape[1], state.shape[1]))


def expected_landing_x(x, y, xv, yv) -> int:
    return int(lib().pzo_expected_landing_x(x, y, xv, yv))


def expected_landing_x_power_hit(xdir, ydir, x, y, xv, yv) -> int:
    return int(lib().pzo_expected_landing_x_power_hit(xdir, ydir, x, y, xv, yv))


if __name__ == "__main__":
    print(build(force=bool(os.environ.get("FORCE"))))

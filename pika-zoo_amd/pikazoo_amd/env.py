"""Batched Pikachu-Volleyball environment on MI355X: the host side of the fused HIP step kernel.

Mirrors the PettingZoo *parallel* API of the reference's ``raw_env``
(pikazoo/env/pikazoo_env.py:72-240): same constructor kwargs, ``possible_agents`` / ``agents``,
``reset(seed, options) -> (obs, infos)``, ``step(actions) -> (obs, rewards, terminations,
truncations, infos)``, ``observation_space(agent)``, ``action_space(agent)`` -- but every dict
value is a device tensor with a leading ``num_envs`` axis (one GPU lane per independent game).

All game state lives in one ``int32[44, num_envs]`` tensor in HBM (field-major); a step is ONE
kernel launch through the C ABI in ``include/pikazoo_hip.h``.  PyTorch is used only for device
memory and streams.  There is no CPU fallback: constructing the env without the HIP library or
without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, Optional

import numpy as np
import torch

from . import _native
from .spaces import Box, Discrete

# The reference is `class raw_env(ParallelEnv)` (pikazoo_env.py:72) and downstream libraries test `isinstance(env,
# ParallelEnv)`: when PettingZoo is importable its class is the base (like spaces.py does with gymnasium's spaces); the
# build image has neither, so the stand-in only carries the name.
try:  # pragma: no cover - pettingzoo is not installed in the build image
    from pettingzoo import ParallelEnv  # type: ignore
except Exception:  # noqa: BLE001

    class ParallelEnv:  # type: ignore[no-redef]
        """Stand-in base when PettingZoo is not installed (``raw_env`` defines the whole parallel API itself)."""

AGENTS = ["player_1", "player_2"]

# action vectors a step launch reads as they are (pz_action_format); any other integer dtype is widened on the host
_ACTION_FORMAT = {torch.int32: 0, torch.int64: 1, torch.uint8: 2, torch.int16: 3}
# flight_tables= of the env -> (landing table, power-hit table)
_TABLE_MODES = {True: (True, True), "both": (True, True), "power_hit": (False, True),
                False: (False, False), None: (False, False), "none": (False, False)}

# observation bounds: pikazoo/env/pikazoo_env.py:485-562 (player, opponent, ball)
_PLAYER_LOW = [32, 108, -15, -1, -2, 0, 0, 0, 0, 0, 0, 0, 0]
_PLAYER_HIGH = [400, 244, 16, 1, 3, 4, 4, 1, 1, 1, 1, 1, 1]
_BALL_LOW = [20, 0, 0, 0, 0, 0, -20, -124, 0]
_BALL_HIGH = [432, 252, 432, 252, 432, 252, 20, 124, 1]
OBS_LOW = np.array(_PLAYER_LOW * 2 + _BALL_LOW, dtype=np.int32)
OBS_HIGH = np.array(_PLAYER_HIGH * 2 + _BALL_HIGH, dtype=np.int32)

# state columns used on the host (include/pikazoo_hip.h)
_E_SCORE_P1 = 38
_E_GAME_ENDED = 42


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()



# torch.cuda.current_stream(...).cuda_stream builds a Stream object per call (2.8 us); the raw getter behind it
# costs 0.08 us, which keeps env.step()'s host side (about 5.5 us) below the duration of the launch it issues
_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device  # (the raw getter: 0.1 us instead of 0.3)


def _raw_stream(device_index: int) -> int:
    if _get_raw_stream is not None:
        return _get_raw_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


# The computer player's flight look-up tables (pz_flight_tables in include/pikazoo_hip.h): caller-owned
# device memory as far as the C ABI is concerned; the Python host keeps ONE pair per device, built by the
# first env with a computer player (about 1 GB of the 288 GB, a few milliseconds of kernel time).
_FLIGHT_TABLES = {}


def flight_tables(device: torch.device, landing: bool = True, power_hit: bool = True):
    """``(PzFlightTables, landing, power_hit)`` of `device`; each table is built on first use and kept per device
    (``landing``: 927 MB, ``power_hit``: 82 MB; a table that was not asked for is NULL in the struct)."""
    key = device.index
    have = _FLIGHT_TABLES.setdefault(key, {})
    lib = _native.load()
    want = [w for w, on in (("landing", landing), ("power_hit", power_hit)) if on and w not in have]
    if want:
        with torch.cuda.device(device):
            for which in want:
                have[which] = torch.empty(lib.pz_flight_table_bytes(0 if which == "landing" else 1), dtype=torch.uint8,
                                          device=device)
            _native.check(lib.pz_build_flight_tables(have["landing"].data_ptr() if "landing" in want else None,
                                                     have["power_hit"].data_ptr() if "power_hit" in want else None,
                                                     _raw_stream(key)), "pz_build_flight_tables")
            torch.cuda.current_stream(device).synchronize()  # envs on other streams may use them right away
    t_landing = have["landing"] if landing else None
    t_hit = have["power_hit"] if power_hit else None
    return _native.PzFlightTables(_ptr(t_landing), _ptr(t_hit)), t_landing, t_hit


def flight_table_bytes(mode=True) -> int:
    """Device memory the flight tables of ``flight_tables=mode`` take (per process and device)."""
    landing, power_hit = _TABLE_MODES[mode]
    lib = _native.load()
    return int(landing) * int(lib.pz_flight_table_bytes(0)) + int(power_hit) * int(lib.pz_flight_table_bytes(1))


class _OutputSet:
    """One set of the buffers a step writes (observations, rewards, terminations) with what is cached per set: the raw
    pointers, the `pz_step_bind` block and the result tuple ``step()`` returns (its dicts hold views of these buffers)."""

    __slots__ = ("obs", "rew", "term_u8", "term", "ptrs", "bound", "bound_key", "result", "result_key")

    def __init__(self, obs, rew, term_u8, state_ptr):
        self.obs, self.rew, self.term_u8 = obs, rew, term_u8
        self.term = term_u8.view(torch.bool)
        self.ptrs = (state_ptr, obs[0].data_ptr(), obs[1].data_ptr(), rew[0].data_ptr(), rew[1].data_ptr(),
                     term_u8.data_ptr())
        # per action format (pz_action_format 0..3): the ctypes block filled by pz_step_bind and the configuration it was
        # bound for (_result_key) -- the block of a format is bound when a tensor of that dtype first comes in
        self.bound = [None] * 4
        self.bound_key = [None] * 4
        self.result = None
        self.result_key = None


class raw_env(ParallelEnv):
    """``pikazoo_v0.raw_env`` for ``num_envs`` games at once.

    Reference kwargs (pikazoo_env.py:79-86): ``winning_score``, ``serve`` in {"winner",
    "alternate", "random"}, ``is_player1_computer``, ``is_player2_computer``, ``render_mode``
    (None or "rgb_array": :meth:`render` draws frames on the GPU from the state; the sprites come from
    ``sprite_dir`` -- the reference's ``pikazoo/env/img`` directory, found by itself when the reference
    package is installed -- or from a ready ``sprites`` set, see :mod:`pikazoo_amd.render`; there is no
    "human" window).

    Batched-env kwargs: ``num_envs``; ``device`` (a CUDA/HIP device); ``seed`` (Philox key of
    the env RNG stream -- the reference seeds PCG64 from OS entropy and ignores ``reset(seed)``,
    pikazoo_env.py:96,149); ``env_id_base`` (global id of lane 0, so shards of one job draw
    disjoint streams); ``auto_reset`` (a finished game is ``reset()`` in place right before its
    next frame, exactly what ``if not env.agents: env.reset()`` does around the reference);
    ``validate_actions`` (default on: an action outside ``[0, action_space(agent).n)`` raises the
    reference's ``IndexError``, pikazoo_env.py:182.  Actions handed over as Python / numpy values are checked on the
    host before the launch, like the reference checks them before it steps.  For device tensors the range check runs
    INSIDE the step kernel -- a launch cannot raise, so it counts into a device counter the env polls every
    ``validate_every`` steps through an asynchronous copy it only reads one poll later (no device synchronisation on the
    step path), and the ``IndexError`` then comes up to ``2 * validate_every`` steps late (:meth:`step_many`: one call
    late); :meth:`check_actions` asks now (one sync), and ``validate_every=1`` is the STRICT mode: every :meth:`step` /
    :meth:`step_many` call reads the counter before it returns (one device synchronisation per call), so the error comes
    from the call that was handed the action, as in the reference -- the frame has run by then, which the reference's
    has not.  The offending game's input for that frame is undefined; no other game and no memory is affected); ``scalar_api`` (``num_envs == 1`` only: return numpy rows / Python
    scalars and empty ``agents`` on termination, i.e. the reference's exact return types;
    ``auto_reset`` then defaults to False, so that ``while env.agents:`` loops end like they do around
    the reference); ``flight_tables`` (computer players only: ``True`` / ``"both"``: look both flight predictions up in
    per-device HBM tables -- 927 MB for the landing point + 82 MB for the six power-hit candidates, per process and device;
    ``"power_hit"``: only the 82 MB table -- the candidates are what diverges most, the landing point is one closed-form
    flight predicted in the kernel; ``False``: everything computed in the kernel; results are identical in all three,
    config 3 runs at 8.7 / 13.2 / 14.5 us per 65 536-game step);
    ``state_format`` ("int32": the state lives in HBM as the ``int32[44, num_envs]`` tensor :attr:`state`,
    live and writable; "packed": as 36 bytes per game instead of 176 -- the bit-packed format of
    ``include/pikazoo_hip.h`` -- which makes large batches about a third faster; every result is identical,
    :attr:`state` then returns an unpacked copy, and checkpoints are interchangeable between the two);
    ``scenery`` (with ``render_mode="rgb_array"``: draw the reference's clouds and waves too.  They are renderer-owned
    state driven by the env RNG, so -- exactly as in the reference, pikazoo_env.py:475-477, cloud_and_wave.py:53-78 --
    the constructor then draws 40 values behind the two boldness draws and every ``render()`` advances the RNG of the
    games it draws; ``observation_dtype`` (``torch.int32``: the reference's Box dtype; ``torch.int16``: the same
    values in half the bytes -- observations are the largest stream a step writes, so large batches run up to a third
    faster again; the fused ``NormalizeObservation`` emits float32 and cannot be combined with it).  With ``scenery``
    the punch effect is drawn too: its two ball attributes are tracked after every ``step()``, a k-frame
    launch clears it; off by default, which keeps ``render()`` free of side effects and ``step()`` a single launch).

    ``output_ring`` (k >= 1 rotating sets of the observation / reward / termination buffers: the reference returns
    FRESH arrays from every ``step`` (pikazoo_env.py:215-235), this env returns views of env-owned buffers -- with
    the default ring of 1 the next ``step`` overwrites them, with ``output_ring=k`` the results of the last k
    ``step`` / ``reset`` calls stay intact, e.g. 2 for a loop that still reads the previous observation).

    ``place_trajectories`` (default on): the two large observation tensors :meth:`rollout_random` / :meth:`step_many`
    allocate are placed in different ranks of the device memory when that can be arranged (pikazoo_amd/placement.py:
    the k-frame launches then run a quarter faster; same results either way).

    Returned tensors are views of env-owned buffers that the ``output_ring``-th next ``step`` overwrites;
    ``clone()`` what must outlive that.
    """

    metadata = {"render_modes": ["rgb_array"], "name": "pikazoo_v0", "render_fps": 20, "is_parallelizable": True}

    def __init__(self, winning_score: int = 15, serve: str = "winner", is_player1_computer: bool = False,
                 is_player2_computer: bool = False, render_mode=None, *, num_envs: int = 1,
                 device="cuda", seed: int = 0, env_id_base: int = 0, auto_reset: Optional[bool] = None,
                 validate_actions: bool = True, scalar_api: bool = False, flight_tables=True,
                 sprite_dir=None, sprites=None, state_format: str = "int32", scenery: bool = False,
                 observation_dtype=torch.int32, output_ring: int = 1, place_trajectories: bool = True,
                 validate_every: int = 64):
        assert serve in ("winner", "alternate", "random")  # pikazoo_env.py:104
        if render_mode not in (None, "rgb_array"):
            raise NotImplementedError('render_mode must be None or "rgb_array" (no "human" window on a GPU batch)')
        if int(winning_score) < 1:
            raise ValueError("winning_score must be >= 1")
        if int(num_envs) < 1:
            raise ValueError("num_envs must be >= 1")
        if state_format not in ("int32", "packed"):
            raise ValueError('state_format must be "int32" or "packed"')
        if observation_dtype in ("int16", "int32"):
            observation_dtype = getattr(torch, observation_dtype)
        if observation_dtype not in (torch.int32, torch.int16):
            raise ValueError("observation_dtype must be torch.int32 or torch.int16")
        if scenery and render_mode is None:
            raise ValueError('scenery=True needs render_mode="rgb_array"')
        if state_format == "packed" and int(winning_score) > 32767:
            raise ValueError("the packed state format holds scores up to 32767")
        if int(output_ring) < 1:
            raise ValueError("output_ring must be >= 1")
        if not isinstance(flight_tables, (bool, str, type(None))) or flight_tables not in _TABLE_MODES:
            raise ValueError('flight_tables must be True / "both", "power_hit" or False / "none"')
        self._lib = _native.load()  # raises when the HIP library has not been built
        self._step_bound = self._lib.pz_step_bound
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"pikazoo_amd runs on MI355X only (got device {self.device}); no CPU fallback")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: pikazoo_amd has no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._dev_index = self.device.index
        if scalar_api and num_envs != 1:
            raise ValueError("scalar_api needs num_envs == 1")
        if auto_reset is None:
            auto_reset = not scalar_api

        self.possible_agents = AGENTS[:]
        self.agents = self.possible_agents[:]
        self.num_envs = int(num_envs)
        self.winning_score = int(winning_score)
        self.serve = serve
        self.render_mode = render_mode
        self._sprites = sprites
        self._sprite_dir = sprite_dir
        self.auto_reset = bool(auto_reset)
        self.validate_actions = bool(validate_actions)
        self.scalar_api = bool(scalar_api)
        self.seed = int(seed)
        self.env_id_base = int(env_id_base)
        self.state_format = state_format

        cfg = _native.PzConfig()
        cfg.winning_score = self.winning_score
        cfg.serve_mode = _native.SERVE_MODES[serve]
        cfg.p1_computer = int(bool(is_player1_computer))
        cfg.p2_computer = int(bool(is_player2_computer))
        cfg.simplify_action = 0
        cfg.ballpos_reward = 0
        cfg.x_line, cfg.y_line = 216, 176
        cfg.auto_reset = int(self.auto_reset)
        cfg.packed_state = int(state_format == "packed")
        cfg.normalize_obs = 2 if observation_dtype == torch.int16 else 0  # 0 int32, 1 float32 normalized, 2 int16
        cfg.seed = self.seed & 0xFFFFFFFFFFFFFFFF
        cfg.env_id_base = self.env_id_base
        # validate_actions: the step kernels count out-of-range actions into this device word (pz_config.action_faults);
        # it is copied to pinned host memory now and then, asynchronously, and looked at when the copy has landed
        self._faults = self._faults_host = self._faults_event = None
        self._faults_pending = False
        self._since_poll = 0
        self.validate_every = max(1, int(validate_every))
        if self.validate_actions:
            self._faults = torch.zeros(1, dtype=torch.int64, device=self.device)
            self._faults_host = torch.zeros(1, dtype=torch.int64).pin_memory()
            self._faults_event = torch.cuda.Event()
            cfg.action_faults = self._faults.data_ptr()
        self._cfg = cfg
        self._cfg_ref = C.byref(cfg)
        self._cfg_version = 0  # bumped by every _fuse_* method
        self._unfused = []           # wrapper classes of this stack that apply themselves outside the kernel (_fuse_*)
        self._unfused_reward = False
        # the action tensors step() saw last: (weak reference, data pointer, dtype, pz_action_format) of a tensor that
        # passed the format checks
        self._a1_seen = self._a2_seen = (lambda: None, 0, None, 0)
        self._tables = None
        self._tables_ref = None  # `const pz_flight_tables*` of every step call (None: compute in the kernel)
        computer = bool(cfg.p1_computer or cfg.p2_computer)
        self.flight_tables = "none"
        if computer and any(_TABLE_MODES[flight_tables]):
            self.flight_tables = "both" if all(_TABLE_MODES[flight_tables]) else "power_hit"
            self._tables = globals()["flight_tables"](self.device, *_TABLE_MODES[flight_tables])
            self._tables_ref = C.byref(self._tables[0])

        n, dev = self.num_envs, self.device
        # columns are padded to a multiple of 64 games so that every workgroup's 256-byte segment of a column is
        # aligned whatever num_envs is (a ragged pitch costs ~10 % per launch); `state` is the [44, n] view
        self._stride = (n + 63) // 64 * 64

        if cfg.packed_state:
            # group A [stride][4] dwords, group B [stride][4] dwords, tail [stride] dwords (include/pikazoo_hip.h)
            self._state_buf = torch.zeros(_native.PACKED_BYTES_PER_GAME * self._stride, dtype=torch.uint8, device=dev)
            self._state_view = None
            # the two scores are the halves of group A's third dword: a live int16 [n, 2] view
            self._scores = self._state_buf[:16 * self._stride].view(torch.int16).view(self._stride, 8)[:n, 4:6]
        else:
            self._state_buf = torch.zeros((_native.STATE_WORDS, self._stride), dtype=torch.int32, device=dev)
            self._state_view = self._state_buf[:, :n]
            self._scores = self._state_view[_E_SCORE_P1:_E_SCORE_P1 + 2].t()  # live [n, 2] view of the state
        self._state_ptr = self._state_buf.data_ptr()
        # the output buffers: `output_ring` sets, used in rotation (one set: every step overwrites the last results)
        self._ring = []
        for _ in range(int(output_ring)):
            obs = [torch.zeros((n, _native.OBS_DIM), dtype=torch.int32, device=dev) for _ in range(2)]
            # one 4-byte word per lane and agent; viewed as int32 or float32 (RewardByBallPosition)
            rew = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
            term = torch.zeros(n, dtype=torch.uint8, device=dev)
            self._ring.append(_OutputSet(obs, rew, term, self._state_ptr))
        self._place_trajectories = bool(place_trajectories)
        self.trajectory_placement = {}  # what placement.alloc_pair did for the latest trajectory tensors (diagnostic)
        self._ring_pos = 0
        self._last_traj = None  # the latest k-frame result whose last frame the single-frame buffers still owe
        self._use_outputs(self._ring[0])
        self._trunc = torch.zeros(n, dtype=torch.bool, device=dev)  # always False (pikazoo_env.py:234)
        self._episodes = torch.zeros(1, dtype=torch.int64, device=dev)
        self._stats = None  # RecordEpisodeStatistics buffer (20 bytes per game), allocated when the wrapper is fused
        self._ep_returns = self._ep_lengths = None
        self.steps_done = 0  # frames stepped by this env (per lane)

        self.action_spaces = {a: Discrete(18) for a in self.possible_agents}
        self._spaces = {}
        with torch.cuda.device(self.device):
            _native.check(self._lib.pz_init(self._state_ptr, n, self._stride, self._cfg_ref, self._stream()),
                          "pz_init")
        self._scenery = None
        if scenery:  # get_all_image's ten clouds: 40 draws of the env stream per game, right behind the constructor's two
            self._scenery = torch.zeros((_native.SCENERY_WORDS, self._stride), dtype=torch.int32, device=dev)
            self._on_int32_state(lambda ptr: _native.check(
                self._lib.pz_scenery_init(self._scenery.data_ptr(), ptr, n, self._stride, self._cfg_ref, self._stream()),
                "pz_scenery_init"))

    # ------------------------------------------------------------------------------------------
    def _use_outputs(self, out):
        """Make `out` (an _OutputSet of the ring) the set the next launch writes."""
        self._out = out
        self._obs, self._rew_raw, self._term_u8, self._term, self._ptrs = out.obs, out.rew, out.term_u8, out.term, out.ptrs

    def _next_outputs(self):
        """Rotate to the ring's next output set (a no-op with the default ring of one)."""
        if len(self._ring) > 1:
            self._ring_pos = (self._ring_pos + 1) % len(self._ring)
            self._use_outputs(self._ring[self._ring_pos])

    def _bound_step(self, fmt: int = 0):
        """The current output set's `pz_step_bind` block for the current configuration and the action format `fmt`
        (bound on first use; a wrapper fused later, or statistics switched on, re-binds)."""
        out, key = self._out, self._result_key()
        if out.bound[fmt] is None or out.bound_key[fmt] != key:
            if out.bound[fmt] is None:
                out.bound[fmt] = C.create_string_buffer(int(self._lib.pz_step_bound_bytes()))
            p = out.ptrs
            self._cfg.action_format = fmt  # (the block copies the configuration; every other call takes int32)
            try:
                _native.check(self._lib.pz_step_bind(out.bound[fmt], p[0], self.num_envs, self._stride, self._cfg_ref, p[1],
                                                     p[2], p[3], p[4], p[5], self._stats_ptr(), self._tables_ref),
                              "pz_step_bind")
            finally:
                self._cfg.action_format = 0
            out.bound_key[fmt] = key
        return out.bound[fmt]

    def _stream(self):
        """Raw handle of the caller's current stream on this env's device (every launch goes there)."""
        return _raw_stream(self._dev_index)

    @property
    def unwrapped(self):
        return self

    @property
    def num_agents(self):
        return len(self.agents)

    @property
    def max_num_agents(self):
        return len(self.possible_agents)

    @property
    def scores(self) -> torch.Tensor:
        """``[num_envs, 2]`` live view (the reference's ``scores`` list, pikazoo_env.py:100): int32, or int16 with
        the packed state format."""
        return self._scores

    @property
    def state(self) -> torch.Tensor:
        """The game state as ``int32[44, num_envs]`` (rows: ``pz_player_field`` x 2, ``pz_ball_field``,
        ``pz_env_field`` of include/pikazoo_hip.h).  With ``state_format="int32"`` this is the live tensor the
        kernels step (writable); with "packed" it is an unpacked copy -- write through :meth:`set_state`."""
        if self._state_view is not None:
            return self._state_view
        return self._unpacked()[0][:, :self.num_envs]

    def read_state(self) -> torch.Tensor:
        """A copy of the game state as ``int32[44, num_envs]`` (either format)."""
        if self._state_view is not None:
            return self._state_view.clone()
        return self._unpacked()[0][:, :self.num_envs]

    def _unpacked(self):
        """(int32[44, stride] copy of the packed state, its stride)"""
        out = torch.empty((_native.STATE_WORDS, self._stride), dtype=torch.int32, device=self.device)
        flagged = torch.zeros(1, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._lib.pz_unpack_state(self._state_ptr, self.num_envs, self._stride, out.data_ptr(),
                                                    self._stride, flagged.data_ptr(), self._stream()), "pz_unpack_state")
        if int(flagged.item()):
            raise _native.PikazooNativeError(
                f"{int(flagged.item())} games carry the packed format's misfit flag: a value left its field")
        return out, self._stride

    def _on_int32_state(self, call):
        """Run ``call(pointer to int32[44, stride] columns)`` for the entry points that take (and may write) int32
        columns only; a packed state is unpacked for the call and packed again."""
        with torch.cuda.device(self.device):
            if self._state_view is not None:
                return call(self._state_ptr)
            cols, _ = self._unpacked()
            out = call(cols.data_ptr())
            _native.check(self._lib.pz_pack_state(cols.data_ptr(), self.num_envs, self._stride, self._state_ptr,
                                                  self._stride, None, self._stream()), "pz_pack_state")
            return out

    def _track_scenery(self, resync: bool = False):
        """The punch effect's radius / y are ball attributes outside the 44 words: re-derive what this frame did to them
        (``pz_scenery_track``).  After a k-frame launch (`resync`) the inner frames are gone: the effect is cleared."""
        with torch.cuda.device(self.device):
            cols = self._state_buf if self._state_view is not None else self._unpacked()[0]  # read-only for the call
            _native.check(self._lib.pz_scenery_track(self._scenery.data_ptr(), cols.data_ptr(), self.num_envs, self._stride,
                                                     self._cfg_ref, int(resync), self._stream()), "pz_scenery_track")

    def set_state(self, state: torch.Tensor):
        """Overwrite the state of every game with ``int32[44, num_envs]`` columns (either format)."""
        state = torch.as_tensor(state, device=self.device)
        if state.shape != (_native.STATE_WORDS, self.num_envs) or state.dtype != torch.int32:
            raise ValueError(f"state must be int32[{_native.STATE_WORDS}, {self.num_envs}]")
        if self._state_view is not None:
            self._state_view.copy_(state)
            return
        src = state.contiguous()
        misfits = torch.zeros(1, dtype=torch.int64, device=self.device)
        staged = torch.empty_like(self._state_buf)
        with torch.cuda.device(self.device):
            _native.check(self._lib.pz_pack_state(src.data_ptr(), self.num_envs, self.num_envs, staged.data_ptr(),
                                                  self._stride, misfits.data_ptr(), self._stream()), "pz_pack_state")
        if int(misfits.item()):
            raise ValueError(f"{int(misfits.item())} games hold values outside the packed format's fields")
        self._state_buf.copy_(staged)

    @property
    def packed_misfits(self) -> int:
        """Games of a ``state_format="packed"`` env whose sticky misfit flag is set: a value left its field (never
        seen in 1.3e11 game-steps of play; a planted, unreachable state can do it).  The step kernels raise the flag
        and keep stepping -- nothing else reports it until the state is unpacked (:attr:`state`, :meth:`state_dict`,
        which raise) -- so a loop that never reads the state should poll this now and then (one small launch reading
        8 bytes per game, and a sync).  Always 0 for the int32 format."""
        if self._state_view is not None:
            return 0
        flagged = torch.zeros(1, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._lib.pz_count_packed_misfits(self._state_ptr, self.num_envs, self._stride,
                                                            flagged.data_ptr(), self._stream()), "pz_count_packed_misfits")
        return int(flagged.item())

    @property
    def episodes_done(self) -> int:
        """Games finished inside ``step_random`` launches (device counter; syncs)."""
        return int(self._episodes.item())

    # ---- fusable wrappers (set by pikazoo_amd.wrappers) -----------------------------------------
    # Every _fuse_* method answers whether the kernel took the wrapper over.  False: this stack order cannot be expressed
    # as a kernel branch (the reference composes its wrappers in any order) and the wrapper class applies itself to the
    # step's outputs with torch operations instead -- which puts it OUTSIDE everything fused, exactly where a wrapper
    # constructed later sits in the reference's stack.  Once one reward wrapper works that way, every later wrapper that
    # reads rewards must too (`_unfused_reward`).
    def _note_unfused(self, name: str, reward: bool = False):
        self._unfused.append(name)
        self._unfused_reward = self._unfused_reward or reward

    def _fuse_simplify_action(self) -> bool:
        """wrappers/simplify_action.py:16-25 inside the kernel: actions become Discrete(13)."""
        if self._cfg.simplify_action:
            return False  # a second SimplifyAction maps the first one's 13 actions again: composed on the host
        self._cfg.simplify_action = 1
        self._cfg_version += 1
        self.action_spaces = {a: Discrete(13) for a in self.possible_agents}
        return True

    def _fuse_ballpos_reward(self, additional_reward, x_line: int, y_line: int) -> bool:
        """wrappers/reward_by_ball_position.py:20-31 inside the kernel: rewards become float32."""
        assert len(additional_reward) == 8  # reward_by_ball_position.py:15
        # not as a kernel branch: a second RewardByBallPosition; above NormalizeObservation (it then reads the NORMALIZED
        # ball coordinates, reward_by_ball_position.py:22 -- what the reference does, and what the wrapper class then
        # does); above statistics that already sum a wrapped reward; above a reward wrapper that runs outside the kernel
        if self._cfg.ballpos_reward or self._cfg.normalize_obs == 1 or self._cfg.episode_stats_mode == 2 or \
                self._unfused_reward or "NormalizeObservation" in self._unfused:
            return False
        self._cfg.ballpos_reward = 1
        self._cfg.x_line, self._cfg.y_line = int(x_line), int(y_line)
        for i, v in enumerate(additional_reward):
            self._cfg.additional_reward[i] = float(v)
        self._cfg_version += 1
        return True

    def _fuse_normal_state_reward(self, reward) -> bool:
        """wrappers/reward_in_normal_state.py:10-15 inside the kernel; remembers whether it was applied
        before or after RewardByBallPosition (the reference's result depends on the wrapper order)."""
        if self._cfg.normal_state_mode or self._cfg.episode_stats_mode == 2 or self._unfused_reward:
            return False  # a second one / above statistics of a wrapped reward / above an unfused reward wrapper
        self._cfg.normal_state_mode = 2 if self._cfg.ballpos_reward else 1
        self._cfg.normal_state_reward = float(reward)
        self._cfg_version += 1
        return True

    def _fuse_normalize_obs(self) -> bool:
        """wrappers/normalize_observation.py:18-35 inside the kernel: observations become float32."""
        if self._cfg.normalize_obs != 0 or "NormalizeObservation" in self._unfused or "RewardByBallPosition" in self._unfused:
            # already normalized; int16 observations (the kernel's float32 rows need the int32 buffers); or a
            # RewardByBallPosition BELOW this wrapper runs outside the kernel and reads the raw coordinates from the
            # observations the kernel hands it (fused, the normalization would move below it)
            return False
        self._cfg.normalize_obs = 1
        self._cfg_version += 1
        return True

    def _fuse_episode_stats(self) -> bool:
        """wrappers/record_episode_statistics.py:27-40 inside the kernel (three words per game)."""
        if self._cfg.episode_stats_mode or self._unfused_reward:
            return False  # a second one / above a reward wrapper that runs outside the kernel
        wrapped = bool(self._cfg.ballpos_reward or self._cfg.normal_state_mode)
        self._cfg.episode_stats_mode = 2 if wrapped else 1
        # include/pikazoo_hip.h: double[2][stride] running returns, then int32[stride] episode lengths
        self._stats = torch.zeros(20 * self._stride, dtype=torch.uint8, device=self.device)
        self._ep_returns = self._stats[:16 * self._stride].view(torch.float64).view(2, self._stride)[:, :self.num_envs]
        self._ep_lengths = self._stats[16 * self._stride:].view(torch.int32)[:self.num_envs]
        self._cfg_version += 1
        return True

    def _no_unfused_wrappers(self, what: str):
        if self._unfused:
            raise RuntimeError(f"{what} returns the kernel's own outputs; {', '.join(self._unfused)} of this stack run outside "
                               "the kernel (an order it cannot fuse) and would be missing from them: use step()")

    @property
    def episode_returns(self) -> Optional[torch.Tensor]:
        """``float64[2, num_envs]`` running episode returns (player_1, player_2): summed in float64 like the
        reference's Python floats (record_episode_statistics.py:31); live view of the kernel's buffer."""
        return self._ep_returns

    @property
    def episode_lengths(self) -> Optional[torch.Tensor]:
        return self._ep_lengths

    @property
    def n_actions(self) -> int:
        return 13 if self._cfg.simplify_action else 18

    @property
    def reward_dtype(self):
        return torch.float32 if (self._cfg.ballpos_reward or self._cfg.normal_state_mode) else torch.int32

    @property
    def obs_dtype(self):
        return (torch.int32, torch.float32, torch.int16)[self._cfg.normalize_obs]

    def _obs_view(self, i, buf=None):
        """The env-owned observation buffer of agent i (or `buf`, an int32 [n, 35] buffer) in the current format: the
        int16 rows occupy the front of the same storage (an even number of them: include/pikazoo_hip.h)."""
        buf = self._obs[i] if buf is None else buf
        fmt = self._cfg.normalize_obs
        if fmt == 0:
            return buf
        if fmt == 1:
            return buf.view(torch.float32)
        n = self.num_envs
        return buf.view(torch.int16).view(-1)[:(n + 1) // 2 * 2 * _native.OBS_DIM].view(-1, _native.OBS_DIM)[:n]

    # ---- spaces (pikazoo_env.py:481-568) ----------------------------------------------------------
    def observation_space(self, agent=None):
        # the reference lru_caches this method (pikazoo_env.py:481): same object on every call; cached per
        # instance here (an lru_cache on the method would pin the env and its device tensors forever)
        sp = self._spaces.get("obs")
        if sp is None:
            dt = np.int16 if self._cfg.normalize_obs == 2 else np.int32
            sp = self._spaces["obs"] = Box(low=OBS_LOW.astype(dt), high=OBS_HIGH.astype(dt), shape=(35,), dtype=dt)
        return sp

    def normalized_observation_space(self, agent=None):
        sp = self._spaces.get("norm")
        if sp is None:  # normalize_observation.py:35
            sp = self._spaces["norm"] = Box(low=0.0, high=1.0, shape=(35,), dtype=np.float32)
        return sp

    def action_space(self, agent):
        return self.action_spaces[agent]

    # ---- results packing ----------------------------------------------------------------------------
    def _rewards(self):
        if self._last_traj is not None:
            self._settle_last_frame()
        dt = self.reward_dtype
        return [r if dt == torch.int32 else r.view(torch.float32) for r in self._rew_raw]

    def _infos(self):
        infos = {a: {"score": self._scores} for a in self.agents}  # aliased like pikazoo_env.py:573-574
        if self._stats is not None:
            # record_episode_statistics.py:34-39: {"r", "l"} of the episode that just ended; here per
            # lane, meaningful where terminations[agent] is True (running sums elsewhere)
            ret = self.episode_returns
            for i, a in enumerate(self.agents):
                infos[a]["episode"] = {"r": ret[i], "l": self._ep_lengths}
        return infos

    def _pack_obs(self):
        if self._last_traj is not None:
            self._settle_last_frame()
        if self.scalar_api:
            return {a: self._obs_view(i)[0].cpu().numpy() for i, a in enumerate(self.possible_agents)}
        return {a: self._obs_view(i) for i, a in enumerate(self.possible_agents)}

    def _stats_ptr(self):
        return None if self._stats is None else self._stats.data_ptr()

    def _result_key(self):
        """Changes whenever the configuration does (a wrapper fused, statistics switched on): what a `pz_step_bind`
        block and a cached result tuple were made for."""
        return self._cfg_version

    def _pack_step(self):
        rew = self._rewards()  # (settles the last frame of a k-frame launch first)
        if not self.scalar_api:
            obs = self._pack_obs()
            out = (obs, dict(zip(self.possible_agents, rew)), {a: self._term for a in self.possible_agents},
                   {a: self._trunc for a in self.possible_agents}, self._infos())
            return out
        # the reference's exact return types (numpy rows, Python scalars, list of scores)
        ended = bool(self._term_u8.item())
        score = self._scores[0].tolist()
        agents = self.agents
        infos = {a: {"score": score} for a in agents}
        if self._stats is not None and ended:
            ret = self.episode_returns
            for i, a in enumerate(agents):
                infos[a]["episode"] = {"r": ret[i][0].item(), "l": int(self._ep_lengths[0].item())}
        out = (self._pack_obs(), {a: rew[i][0].item() for i, a in enumerate(agents)},
               {a: ended for a in agents}, {a: False for a in agents}, infos)
        if ended and not self.auto_reset:
            self.agents = []  # pikazoo_env.py:237-238
        return out

    # ---- reset (pikazoo_env.py:149-173) ---------------------------------------------------------------
    def reset(self, seed=None, options=None, mask: Optional[torch.Tensor] = None):
        """``reset()`` of every game (or of the lanes where ``mask`` is non-zero).

        ``seed`` and ``options`` are accepted and ignored, like the reference (it never re-seeds,
        pikazoo_env.py:149-173); the env stream continues from each lane's draw counter."""
        self.agents = self.possible_agents[:]
        self._next_outputs()
        if mask is not None and self._last_traj is not None:
            # (pz_reset writes the observation rows of EVERY game from the state -- the unmasked ones unchanged -- but
            # leaves rewards and flags alone: those must hold the k-frame launch's last frame first)
            self._settle_last_frame()
        self._last_traj = None
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            if m.shape != (self.num_envs,):
                raise ValueError(f"mask must have shape ({self.num_envs},)")
        with torch.cuda.device(self.device):
            _native.check(self._lib.pz_reset(self._state_ptr, self.num_envs, self._stride, self._cfg_ref,
                                             _ptr(m), self._obs[0].data_ptr(), self._obs[1].data_ptr(),
                                             self._stats_ptr(), self._stream()), "pz_reset")
        if self._scenery is not None:  # a new round clears the punch effect (physics.py:274-275); nothing to remember
            words = [69, 71, 72, 73, 74]
            keep = self._scenery[words, :self.num_envs] * (m == 0).to(torch.int32) if m is not None else 0
            self._scenery[words, :self.num_envs] = keep
        if self.scalar_api:
            return self._pack_obs(), {a: {"score": self._scores[0].tolist()} for a in self.agents}
        return self._pack_obs(), self._infos()

    # ---- step (pikazoo_env.py:175-240) ------------------------------------------------------------------
    def _action_tensor(self, a) -> torch.Tensor:
        if not isinstance(a, torch.Tensor) or a.device.type == "cpu":
            # host values: range-checked here, before anything is launched (pikazoo_env.py:182 raises before it steps)
            host = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
            host = host.reshape(-1)
            if self.validate_actions and host.size and host.dtype.kind in "iu" and \
                    (int(host.min()) < 0 or int(host.max()) >= self.n_actions):
                raise IndexError(f"action out of range [0, {self.n_actions})")
            a = torch.as_tensor(host, device=self.device)
        elif a.device != self.device:
            a = a.to(self.device)
        a = a.reshape(-1)
        if a.dtype not in _ACTION_FORMAT:
            # int32 / int64 / uint8 / int16 go into the launch as they are (pz_action_format: no cast kernel, and the
            # range check sees the full value); the remaining integer dtypes are widened, which cannot wrap
            if a.dtype.is_floating_point or a.dtype == torch.bool or a.dtype.is_complex:
                raise TypeError(f"actions must be integer tensors, got {a.dtype}")
            a = a.to(torch.int64 if a.dtype in (getattr(torch, "uint32", None), getattr(torch, "uint64", None)) else torch.int32)
        a = a.contiguous()
        if a.numel() != self.num_envs:
            raise ValueError(f"expected {self.num_envs} actions per agent, got {a.numel()}")
        return a

    def step(self, actions: Dict[str, torch.Tensor]):
        """One frame of every game.  ``actions[agent]``: int tensor ``[num_envs]`` in
        ``[0, action_space(agent).n)`` (ints / arrays are accepted for small batches).  As in the
        reference, both agents' key states are read even for a computer-controlled side."""
        if not self.agents:
            raise RuntimeError("step() after termination: call reset() first (agents is empty)")
        a1 = actions["player_1"]  # KeyError on a missing agent
        a2 = actions["player_2"]
        n = self.num_envs
        # fast path: an int32 / int64 / uint8 / int16 device tensor of the right size needs no conversion: the launch
        # reads the caller's element type (pz_action_format; torch's default integer dtype is int64).  A tensor OBJECT
        # that passed once (a policy writing its actions into the same buffer every step) is recognised by identity + data
        # pointer and only re-checked for what an in-place operation can change without moving it (dtype, size, strides:
        # `set_`, `as_strided_`, `.data = ...`) -- the host side of a step has to stay below the duration of the launch
        # it issues.  The env keeps a weak reference only.
        seen, seen_ptr, seen_dtype, f1 = self._a1_seen
        p1 = a1.data_ptr() if type(a1) is torch.Tensor else 0
        if not (seen() is a1 and p1 == seen_ptr and a1.dtype is seen_dtype and a1.numel() == n and a1.is_contiguous()):
            if not (type(a1) is torch.Tensor and a1.dtype in _ACTION_FORMAT and a1.get_device() == self._dev_index
                    and a1.numel() == n and a1.is_contiguous()):
                a1 = self._action_tensor(a1)
                p1 = a1.data_ptr()
                f1 = _ACTION_FORMAT[a1.dtype]
            else:
                f1 = _ACTION_FORMAT[a1.dtype]
                self._a1_seen = (weakref.ref(a1), p1, a1.dtype, f1)
        seen, seen_ptr, seen_dtype, f2 = self._a2_seen
        p2 = a2.data_ptr() if type(a2) is torch.Tensor else 0
        if not (seen() is a2 and p2 == seen_ptr and a2.dtype is seen_dtype and a2.numel() == n and a2.is_contiguous()):
            if not (type(a2) is torch.Tensor and a2.dtype in _ACTION_FORMAT and a2.get_device() == self._dev_index
                    and a2.numel() == n and a2.is_contiguous()):
                a2 = self._action_tensor(a2)
                p2 = a2.data_ptr()
                f2 = _ACTION_FORMAT[a2.dtype]
            else:
                f2 = _ACTION_FORMAT[a2.dtype]
                self._a2_seen = (weakref.ref(a2), p2, a2.dtype, f2)
        if f1 != f2:  # one launch reads one element type: two dtypes meet in int64, which holds all four
            a1, a2 = self._action_tensor(a1).to(torch.int64), self._action_tensor(a2).to(torch.int64)
            p1, p2, f1 = a1.data_ptr(), a2.data_ptr(), 1
        if len(self._ring) > 1:
            self._next_outputs()
        self._last_traj = None  # (this launch overwrites the single-frame buffers)
        out = self._out
        # pz_step through its prepared-argument form: one FFI call with four scalars (the twelve buffers and the
        # configuration were bound once) -- the host side of a step stays below the duration of the launch it issues
        bound = out.bound[f1] if out.bound_key[f1] == self._cfg_version else self._bound_step(f1)
        if _get_device() == self._dev_index:
            rc = self._step_bound(bound, p1, p2, _raw_stream(self._dev_index))
        else:
            with torch.cuda.device(self.device):
                rc = self._lib.pz_step_bound(bound, p1, p2, self._stream())
        if rc:
            _native.check(rc, "pz_step")
        if self._scenery is not None:
            self._track_scenery()
        self.steps_done += 1
        if self._faults is not None:  # validate_actions: the launch counted out-of-range actions (pikazoo_env.py:182)
            self._since_poll += 1
            if self.validate_every == 1 and not torch.cuda.is_current_stream_capturing():
                self.check_actions()  # strict mode: the error comes from the call that was handed the action
            elif self._since_poll >= self.validate_every:
                self._poll_action_faults()
        if self.scalar_api:
            self.check_actions()  # (this API synchronises on every step anyway: the reference's error on the step itself)
            return self._pack_step()
        if out.result is None or out.result_key != self._cfg_version:
            out.result, out.result_key = self._pack_step(), self._cfg_version
        return out.result

    # ---- out-of-range actions (validate_actions) -------------------------------------------------------------------
    def _raise_action_fault(self):
        self._faults.zero_()
        self._faults_pending = False
        raise IndexError(f"action out of range [0, {self.n_actions}) in a recent step: that game's input for the frame "
                         "was undefined (the reference raises on the step itself, pikazoo_env.py:182)")

    def _poll_action_faults(self):
        """Look at the counter copy the PREVIOUS poll requested, then request a new one.  The copy was queued
        ``validate_every`` steps ago: waiting for it is no device synchronisation -- it only keeps the host from
        running more than two poll intervals ahead of the GPU -- and it is what bounds how late the error comes."""
        self._since_poll = 0
        if torch.cuda.is_current_stream_capturing():
            return  # (a hipGraph capture records launches only: the counter is looked at by the steps outside it)
        if self._faults_pending:
            self._faults_event.synchronize()
            self._faults_pending = False
            if int(self._faults_host[0]) != 0:
                self.check_actions()  # (the rare path: confirmed on the device word itself before anything is raised)
        if not self._faults_pending:
            with torch.cuda.device(self.device):
                self._faults_host.copy_(self._faults, non_blocking=True)
                self._faults_event.record()
            self._faults_pending = True

    def check_actions(self):
        """Raise ``IndexError`` now if any step since the last check was handed an action outside
        ``[0, action_space(agent).n)`` (one device synchronisation; a no-op with ``validate_actions=False``)."""
        if self._faults is not None and int(self._faults.item()) != 0:
            self._raise_action_fault()

    def step_random(self, action_seed: int, t0: Optional[int] = None, k: int = 1):
        """``k`` frames under the uniform random policy drawn on device (Philox stream
        ``action_seed``, step indices ``t0 .. t0+k-1``; ``t0`` defaults to ``steps_done``) in ONE
        launch.  Returns the last frame's step tuple."""
        self._no_unfused_wrappers("step_random")
        if t0 is None:
            t0 = self.steps_done
        self._next_outputs()
        self._last_traj = None
        with torch.cuda.device(self.device):
            _native.check(self._lib.pz_step_random(self._state_ptr, self.num_envs, self._stride,
                                                   self._cfg_ref, int(action_seed) & 0xFFFFFFFFFFFFFFFF, int(t0),
                                                   int(k), self._obs[0].data_ptr(), self._obs[1].data_ptr(),
                                                   self._rew_raw[0].data_ptr(), self._rew_raw[1].data_ptr(),
                                                   self._term_u8.data_ptr(), self._stats_ptr(),
                                                   self._episodes.data_ptr(), self._tables_ref, self._stream()),
                          "pz_step_random")
        if self._scenery is not None:
            self._track_scenery(resync=int(k) != 1)
        self.steps_done += int(k)
        return self._pack_step()

    def rollout_random(self, action_seed: int, k: int, t0: Optional[int] = None, out: Optional[dict] = None):
        """``k`` frames under the random policy in ONE launch, keeping every frame's outputs.

        Returns a dict of trajectory tensors: ``actions`` ``int32[k, 2, N]``, ``obs``
        ``{agent: int32[k, N, 35]}``, ``rewards`` ``{agent: [k, N]}``, ``terminations``
        ``bool[k, N]``.  Bit-identical to ``k`` calls of ``step(random_actions(...))``; the state
        tensor is read and written once.  Pass the previous result as ``out`` to reuse its buffers."""
        self._no_unfused_wrappers("rollout_random")
        if t0 is None:
            t0 = self.steps_done
        k, n, dev = int(k), self.num_envs, self.device
        if k < 1:
            raise ValueError("k must be >= 1")
        if k > 1 and n % self._traj_multiple() != 0:
            raise ValueError(f"rollout_random needs num_envs to be a multiple of {self._traj_multiple()}")
        if out is None or out["_k"] != k:
            out = self._alloc_trajectory(k)
            out["actions"] = torch.empty((k, 2, n), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _native.check(self._lib.pz_rollout_random(
                self._state_ptr, n, self._stride, self._cfg_ref, int(action_seed) & 0xFFFFFFFFFFFFFFFF, int(t0), k,
                out["actions"].data_ptr(), out["_obs"][0].data_ptr(), out["_obs"][1].data_ptr(),
                out["_rew"][0].data_ptr(), out["_rew"][1].data_ptr(), out["_term"].data_ptr(),
                self._stats_ptr(), self._episodes.data_ptr(), self._tables_ref, self._stream()), "pz_rollout_random")
        if self._scenery is not None:
            self._track_scenery(resync=True)
        self.steps_done += k
        return self._finish_trajectory(out)

    def step_many(self, actions: torch.Tensor, out: Optional[dict] = None):
        """``k`` frames of GIVEN actions (``int32[k, 2, N]``: frame, agent, game) in ONE launch, keeping
        every frame's outputs; same result dict as :meth:`rollout_random`.  Bit-identical to ``k`` calls
        of :meth:`step` on the ``k`` slices (a recorded action tape, an open-loop plan ...).

        ``validate_actions``: the launch range-checks the tape while it parks it and counts into the env's device
        counter.  By default that counter is read asynchronously, so an out-of-range action on the tape raises
        ``IndexError`` from the NEXT call that polls (the next ``step_many`` / a later ``step``) -- a caller whose last
        call this is asks with :meth:`check_actions` before it trusts the trajectory; with ``validate_every=1`` (strict)
        this call synchronises and raises itself."""
        self._no_unfused_wrappers("step_many")
        n, dev = self.num_envs, self.device
        if actions.dim() != 3 or actions.shape[1] != 2 or actions.shape[2] != n:
            raise ValueError(f"actions must have shape [k, 2, {n}]")
        if actions.dtype.is_floating_point or actions.dtype == torch.bool or actions.dtype.is_complex:
            raise TypeError(f"actions must be an integer tensor, got {actions.dtype}")
        if actions.dtype != torch.int32 and actions.dtype.itemsize >= 4:
            # (int64, and the unsigned 32- / 64-bit types.)  The launch parks the tape from int32 rows; a plain cast would
            # WRAP (2**32 + 3 -> 3) in front of its range check, so the narrowing saturates first: every out-of-range value
            # stays out of range (-1 or 127), and the launch counts it like any other (two small torch launches per k
            # frames, no synchronisation)
            actions = actions.to(device=dev, dtype=torch.int64).clamp(-1, 127).to(torch.int32)
        if actions.dtype != torch.int32 or actions.device != dev or not actions.is_contiguous():
            actions = actions.to(device=dev, dtype=torch.int32).contiguous()  # (the smaller integer types: widened, exact)
        k = int(actions.shape[0])
        if k > 1 and n % self._traj_multiple() != 0:
            raise ValueError(f"step_many needs num_envs to be a multiple of {self._traj_multiple()}")
        if out is None or out["_k"] != k:
            out = self._alloc_trajectory(k)
        out["actions"] = actions
        with torch.cuda.device(dev):
            _native.check(self._lib.pz_step_many(
                self._state_ptr, n, self._stride, self._cfg_ref, actions.data_ptr(), k, out["_obs"][0].data_ptr(),
                out["_obs"][1].data_ptr(), out["_rew"][0].data_ptr(), out["_rew"][1].data_ptr(),
                out["_term"].data_ptr(), self._stats_ptr(), self._episodes.data_ptr(), self._tables_ref,
                self._stream()), "pz_step_many")
        if self._scenery is not None:
            self._track_scenery(resync=True)
        self.steps_done += k
        if self._faults is not None:  # (the launch range-checked the tape as it parked it)
            if self.validate_every == 1 and not torch.cuda.is_current_stream_capturing():
                self.check_actions()  # strict: this call raises
            else:
                self._poll_action_faults()  # read one call later (no synchronisation here)
        return self._finish_trajectory(out)

    def _alloc_trajectory(self, k):
        """The output tensors of a k-frame launch; the two observation tensors in different ranks of the device
        memory when they are large enough for that to matter (placement.alloc_pair)."""
        n, dev = self.num_envs, self.device
        shape = (k, n, _native.OBS_DIM)
        if self._place_trajectories:
            from . import placement

            with torch.cuda.device(dev):
                obs = list(placement.alloc_pair(shape, self._traj_obs_dtype(), dev))
            self.trajectory_placement = dict(placement.last_info)
        else:
            obs = [torch.empty(shape, dtype=self._traj_obs_dtype(), device=dev) for _ in range(2)]
        return {"_k": k, "_obs": obs,
                "_rew": [torch.empty((k, n), dtype=torch.int32, device=dev) for _ in range(2)],
                "_term": torch.empty((k, n), dtype=torch.uint8, device=dev)}

    def _traj_multiple(self):
        """every frame's observation slab of a trajectory launch must stay 16-byte aligned: 140- resp. 70-byte rows"""
        return 8 if self._cfg.normalize_obs == 2 else 4

    def _traj_obs_dtype(self):
        return torch.int16 if self._cfg.normalize_obs == 2 else torch.int32

    def _finish_trajectory(self, out):
        self._next_outputs()
        dt, odt = self.reward_dtype, self.obs_dtype
        if out.get("_views") != (dt, odt):
            rew = [r if dt == torch.int32 else r.view(torch.float32) for r in out["_rew"]]
            out["obs"] = dict(zip(self.possible_agents, [o if o.dtype == odt else o.view(odt) for o in out["_obs"]]))
            out["rewards"] = dict(zip(self.possible_agents, rew))
            out["terminations"] = out["_term"].view(torch.bool)
            out["_views"] = (dt, odt)
        # the single-frame views follow the last frame -- when somebody looks at them (_settle_last_frame): five
        # small copy launches after every k-frame launch cost the GPU 43 us, half a 32-frame rollout
        self._last_traj = out
        return out

    def _settle_last_frame(self):
        """Bring the env-owned single-frame buffers up to the last frame of the latest k-frame launch."""
        out, self._last_traj = self._last_traj, None
        if out is not None:
            self._obs_view(0).copy_(out["obs"][self.possible_agents[0]][-1])
            self._obs_view(1).copy_(out["obs"][self.possible_agents[1]][-1])
            self._rew_raw[0].copy_(out["_rew"][0][-1])
            self._rew_raw[1].copy_(out["_rew"][1][-1])
            self._term_u8.copy_(out["_term"][-1])

    def random_actions(self, action_seed: int, t: Optional[int] = None):
        """The policy stream of :meth:`step_random` as two ``int32[num_envs]`` device tensors."""
        if t is None:
            t = self.steps_done
        a1 = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        a2 = torch.empty_like(a1)
        with torch.cuda.device(self.device):
            _native.check(self._lib.pz_random_actions(a1.data_ptr(), a2.data_ptr(), self.num_envs, self.env_id_base,
                                                      int(action_seed) & 0xFFFFFFFFFFFFFFFF, int(t), self.n_actions,
                                                      self._stream()), "pz_random_actions")
        return {self.possible_agents[0]: a1, self.possible_agents[1]: a2}

    def observe(self):
        """``_get_obs`` (pikazoo_env.py:576-624) of the current state into fresh tensors."""
        o1 = torch.empty_like(self._obs[0])
        o2 = torch.empty_like(self._obs[1])
        with torch.cuda.device(self.device):
            _native.check(self._lib.pz_observe(self._state_ptr, self.num_envs, self._stride,
                                               int(self._cfg.normalize_obs), int(self._cfg.packed_state),
                                               o1.data_ptr(), o2.data_ptr(), self._stream()), "pz_observe")
        return {self.possible_agents[0]: self._obs_view(0, o1), self.possible_agents[1]: self._obs_view(1, o2)}

    # ---- checkpoint: the state tensor + the fused-wrapper words + what the trajectory is keyed by -------------
    _CFG_KEYS = ("winning_score", "serve_mode", "p1_computer", "p2_computer", "simplify_action", "ballpos_reward",
                 "x_line", "y_line", "auto_reset", "normal_state_mode", "normal_state_reward", "normalize_obs",
                 "episode_stats_mode", "seed", "env_id_base")

    def _cfg_dict(self):
        d = {k: getattr(self._cfg, k) for k in self._CFG_KEYS}
        d["normalize_obs"] = int(self._cfg.normalize_obs == 1)  # (int32 vs int16 observations: same trajectory)
        d["additional_reward"] = [float(v) for v in self._cfg.additional_reward]
        return d

    def state_dict(self):
        """Everything a bit-exact continuation needs: the state tensor, the RecordEpisodeStatistics words, the
        counters, and the configuration (Philox key, game ids, fused wrappers) the trajectory depends on."""
        return {"state": self.state.clone(), "steps_done": self.steps_done, "seed": self.seed,
                "env_id_base": self.env_id_base, "config": self._cfg_dict(),
                "episode_stats": None if self._stats is None else self._stats.clone(),
                "episodes_done": self._episodes.clone(),
                "scenery": None if self._scenery is None else self._scenery[:, :self.num_envs].clone()}

    def load_state_dict(self, sd):
        """Restore a :meth:`state_dict`.  Raises when it was taken from an env whose Philox key, game ids or
        (fused-wrapper) configuration differ from this one: the continuation would silently be another
        trajectory."""
        if tuple(sd["state"].shape) != (_native.STATE_WORDS, self.num_envs):
            raise ValueError("state shape mismatch")
        cfg = sd.get("config")
        if cfg is not None:
            mine = self._cfg_dict()
            diff = {k: (cfg.get(k), mine[k]) for k in mine if cfg.get(k) != mine[k]}
            if diff:
                raise ValueError(f"checkpoint was taken with another configuration (saved, this env): {diff}")
        elif int(sd.get("seed", self.seed)) != self.seed or int(sd.get("env_id_base", self.env_id_base)) != self.env_id_base:
            raise ValueError("checkpoint was taken with another seed / env_id_base")
        stats = sd.get("episode_stats")
        if (stats is None) != (self._stats is None):
            raise ValueError("checkpoint and env disagree on RecordEpisodeStatistics")
        scenery = sd.get("scenery")
        if (scenery is None) != (self._scenery is None):
            raise ValueError("checkpoint and env disagree on scenery= (the clouds and waves of render())")
        self.set_state(sd["state"])
        if scenery is not None:
            self._scenery[:, :self.num_envs].copy_(scenery)
        if stats is not None:
            self._stats.copy_(stats)
        if sd.get("episodes_done") is not None:
            self._episodes.copy_(sd["episodes_done"])
        self.steps_done = int(sd["steps_done"])
        self.agents = self.possible_agents[:]

    def render(self, lanes=None, out: Optional[torch.Tensor] = None):
        """``rgb_array`` frames of the current state (pikazoo_env.py:250-384, drawn by ``pz_render``):
        ``uint8[m, 304, 432, 3]`` for the games `lanes` (an int sequence / tensor; default: every game, as long
        as that stays below 1 GiB); ``scalar_api`` envs get the reference's ``[304, 432, 3]`` numpy array.
        An env created with ``scenery=True`` also draws the clouds and waves -- and, like the reference's ``render()``,
        then advances the env RNG of the games it draws (`lanes` must be distinct); it also draws the punch effect."""
        if self.render_mode is None:  # the reference warns and returns None (pikazoo_env.py:355-357)
            import warnings

            warnings.warn("You are calling render method without specifying any render mode.")
            return None
        from . import render as _render

        if self._sprites is None:
            d = self._sprite_dir or _render.default_image_dir()
            if d is None:
                raise FileNotFoundError(
                    "render needs the reference's sprites: pass sprite_dir=<.../pikazoo/env/img> (or sprites=) to the "
                    "env; they are not redistributed with this package")
            self._sprites = _render.load_sprites(d, self.device)
        lane_t = None
        if lanes is not None:
            lane_t = torch.as_tensor(lanes, device=self.device).reshape(-1).to(torch.int32).contiguous()
            if lane_t.numel() and (int(lane_t.min()) < 0 or int(lane_t.max()) >= self.num_envs):
                raise IndexError("lane out of range")
        elif self.num_envs * _render.HEIGHT * _render.WIDTH * 3 > (1 << 30):
            raise ValueError(f"rendering all {self.num_envs} games needs more than 1 GiB: pass lanes=")
        if self._scenery is not None:
            if lane_t is not None and int(torch.unique(lane_t).numel()) != int(lane_t.numel()):
                raise ValueError("lanes must be distinct when the scenery is drawn (every frame ticks its game's clouds)")
            # pz_render then advances the rng counters of the drawn games in the int32 columns
            frames = self._on_int32_state(lambda ptr: _render.render(
                self._lib, ptr, self.device, self.num_envs, self._stride, self._sprites, lane_t, self._stream(), out,
                scenery=self._scenery, cfg_ref=self._cfg_ref))
        else:
            with torch.cuda.device(self.device):
                cols = self._state_buf if self._state_view is not None else self._unpacked()[0]  # (kept alive over the launch)
                frames = _render.render(self._lib, cols.data_ptr(), self.device, self.num_envs, self._stride,
                                        self._sprites, lane_t, self._stream(), out)
        return frames[0].cpu().numpy() if self.scalar_api else frames

    def close(self):
        pass


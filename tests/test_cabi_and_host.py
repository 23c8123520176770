"""CPU tests (no GPU needed): the C-ABI library loads and exports every symbol include/*.h
declares, the binding's struct matches the library, host-side logic (spaces, sharding, the N>1
counter all-reduce over gloo) behaves, and the product has no CPU fallback."""
import ctypes as C
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def built_lib():
    sys.path.insert(0, str(REPO / "pika-zoo_amd"))
    import build as pz_build

    return pz_build.build()


def free_port() -> str:
    """A TCP port nobody holds right now (the rendezvous tests would otherwise collide with a concurrent session, or with
    their own previous run's socket in TIME_WAIT, and fail after their full timeout)."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return str(sock.getsockname()[1])


def header_functions(header="pikazoo_hip.h"):
    text = (REPO / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pz_[a-z_]+)\s*\(", text)))


def dynamic_pz_symbols(lib_path):
    """The pz_* functions a shared library really exports (`nm -D`), not what dlsym happens to find."""
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib_path)], check=True, capture_output=True, text=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.split()[-2:-1] == ["T"] and ln.split()[-1].startswith("pz_")})


def test_library_exports_every_declared_symbol(built_lib):
    from pikazoo_amd import _native

    lib = C.CDLL(str(built_lib))
    names = header_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pikazoo_hip.h but not exported"
    assert sorted(_native.exported_names()) == names, "binding and header disagree"
    # ... and nothing else: the product library exports exactly the entry points a maintainer of the reference would
    # bind; diagnostics live in libpikazoo_diag.so behind include/pikazoo_diag.h
    assert dynamic_pz_symbols(built_lib) == names
    assert not {"pz_probe_launch", "pz_selftest_predictor"} & set(names)


def test_diagnostics_library_is_separate_and_matches_its_header(built_lib):
    import build as pz_build
    import diag

    names = header_functions("pikazoo_diag.h")
    assert names == sorted(diag.SIGNATURES) == dynamic_pz_symbols(pz_build.DIAG_LIB)
    assert pz_build.library_id(pz_build.DIAG_LIB) == pz_build.source_id()  # one build id for both libraries
    # nothing in the package binds or loads it
    for p in (REPO / "pika-zoo_amd" / "pikazoo_amd").rglob("*.py"):
        assert "pikazoo_diag" not in p.read_text() and "import diag" not in p.read_text(), p
    lib = diag.load()
    fake = C.c_void_p(4096)
    # the launch-floor probe: pz_step's seven buffers, `what` in 0..3, at least one game
    probe = lib.pz_probe_launch
    assert probe(None, 8, 8, fake, fake, fake, fake, fake, fake, 0, 0, None) == -1
    assert probe(fake, 0, 0, fake, fake, fake, fake, fake, fake, 0, 0, None) == -2
    assert probe(fake, 8, 4, fake, fake, fake, fake, fake, fake, 0, 0, None) == -2
    assert probe(fake, 8, 8, fake, fake, fake, fake, fake, fake, 4, 0, None) == -3
    assert probe(fake, 8, 8, fake, fake, fake, fake, fake, fake, 3, 5000, None) == -3
    assert probe(fake, 8, 8, fake, fake, C.c_void_p(4100), fake, fake, fake, 2, 0, None) == -4
    assert lib.pz_selftest_predictor(None, fake, fake, fake, 8, 1, fake, fake, None) == -1
    assert lib.pz_selftest_predictor(fake, fake, fake, fake, 0, 1, fake, fake, None) == 0


def test_binding_matches_library_layout(built_lib):
    from pikazoo_amd import _native

    lib = _native.load()
    assert lib.pz_abi_version() == 10 and lib.pz_state_words() == 44 and lib.pz_obs_dim() == 35
    assert lib.pz_config_bytes() == C.sizeof(_native.PzConfig) == 120
    assert lib.pz_error_string(0) == b"ok" and b"NULL" in lib.pz_error_string(-1)
    assert C.sizeof(_native.PzFlightTables) == 16


def test_library_carries_the_digest_of_the_sources_it_was_built_from(built_lib, tmp_path):
    """The .so is git-ignored but travels to the GPU box with the working tree: the binding must refuse one
    that was built from other sources (build.py bakes a digest of csrc + header + flags into it)."""
    import build as pz_build
    from pikazoo_amd import _native

    assert pz_build.library_id(built_lib) == pz_build.source_id() == _native.build_id()
    assert re.fullmatch(r"[0-9a-f]{16}", _native.build_id())
    # a library stamped with another digest is reported as stale (checked on a copy: nothing is loaded)
    stale = tmp_path / "libpikazoo_hip.so"
    stale.write_bytes(built_lib.read_bytes().replace(b"pz_build_id:" + pz_build.source_id().encode(),
                                                     b"pz_build_id:" + b"0" * 16))
    assert pz_build.library_id(stale) == "0" * 16 != pz_build.source_id()
    # extra compiler flags are part of the digest: a diagnostic variant written to the product path is not the product
    assert pz_build.source_id(("-DPZ_DIAGNOSTIC_BUILD=1",)) != pz_build.source_id() == pz_build.source_id(())
    # ... and build.py never writes one there: the product path holds the product
    with pytest.raises(ValueError, match="diagnostic"):
        pz_build.build(extra_flags=("-DPZ_DIAGNOSTIC_BUILD=1",))
    saved, _native._lib, _native.LIB_PATH = (_native._lib, _native.LIB_PATH), None, stale
    try:
        with pytest.raises(_native.PikazooNativeError, match="stale"):
            _native.load()
    finally:
        _native._lib, _native.LIB_PATH = saved


_PETTINGZOO_PROBE = r'''
import sys, types
# a stand-in PettingZoo: just the two classes the reference's env and wrappers subclass
pz, utils = types.ModuleType("pettingzoo"), types.ModuleType("pettingzoo.utils")
class ParallelEnv:  # pettingzoo.ParallelEnv (pikazoo/env/pikazoo_env.py:21,72)
    def state(self):
        raise NotImplementedError
    @property
    def num_agents(self):
        return len(self.agents)
class BaseParallelWrapper(ParallelEnv):  # pettingzoo.utils.BaseParallelWrapper (pikazoo/wrappers/simplify_action.py:3,7)
    def __init__(self, env):
        self.env = env
pz.ParallelEnv, utils.BaseParallelWrapper, pz.utils = ParallelEnv, BaseParallelWrapper, utils
if sys.argv[2] == "with":
    sys.modules["pettingzoo"], sys.modules["pettingzoo.utils"] = pz, utils
sys.path.insert(0, sys.argv[1])
from pikazoo_amd import env, pikazoo_v0, wrappers
from pikazoo_amd.wrappers import base
real = sys.argv[2] == "with"
assert issubclass(pikazoo_v0.raw_env, ParallelEnv) == real, pikazoo_v0.raw_env.__mro__
assert (env.ParallelEnv is ParallelEnv) == real
for name in wrappers.__all__:
    cls = getattr(wrappers, name)
    assert issubclass(cls, base.BaseParallelWrapper)
    assert issubclass(cls, BaseParallelWrapper) == real and issubclass(cls, ParallelEnv) == real, (name, cls.__mro__)
# the forwarding stays this package's own either way: attributes the PettingZoo base defines itself are read through
class Inner:
    possible_agents = agents = ["player_1", "player_2"]
    metadata = {"name": "pikazoo_v0"}
    state = "a tensor here, not PettingZoo's method"
    num_agents = max_num_agents = 2
    unwrapped = "inner"
    extra = 7
w = base.BaseParallelWrapper(Inner())
assert w.state == Inner.state and w.num_agents == 2 and w.unwrapped == "inner" and w.extra == 7
assert w.possible_agents is Inner.possible_agents and w.metadata is Inner.metadata and w.agents is Inner.agents
print("ok", real)
'''


@pytest.mark.parametrize("pettingzoo", ["with", "without"])
def test_env_and_wrappers_subclass_pettingzoos_classes_when_it_is_importable(tmp_path, pettingzoo):
    """The reference is `class raw_env(ParallelEnv)` and its wrappers are `BaseParallelWrapper`s (pikazoo_env.py:72,
    wrappers/simplify_action.py:7); downstream libraries test `isinstance`.  With PettingZoo importable this package's
    classes take its classes as bases (the try-import of spaces.py), without it (this image) plain stand-ins.  Checked in a
    child process with a stand-in `pettingzoo` in sys.modules: the MRO switches, the wrappers' forwarding does not."""
    script = tmp_path / "probe.py"
    script.write_text(_PETTINGZOO_PROBE)
    r = subprocess.run([sys.executable, str(script), str(REPO / "pika-zoo_amd"), pettingzoo], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == f"ok {pettingzoo == 'with'}", r.stdout + r.stderr


def test_a_diagnostic_build_is_never_taken_for_the_product(built_lib, tmp_path):
    """csrc/pz_diagnostic.hpp is the one compile-time switch of the kernel sources.  A library built with it (tools/ab.py,
    tools/stamps.py: into tools/bin/) says so itself -- pz_build_id() == "diagnostic" --, carries no product id in its
    file, and is refused by the binding when somebody puts it on the product's path; build.py refuses the flag (checked
    in test_library_carries_the_digest_of_the_sources_it_was_built_from).  Built here as a subset variant: seconds."""
    import build as pz_build
    from pikazoo_amd import _native

    sys.path.insert(0, str(REPO / "tools"))
    import ab

    variant = tmp_path / "libpikazoo_hip.so"
    bits = ab.diag_bits(bits=32, subset=705)  # no early stores; single-frame pair kernels, human and player 2 = computer
    subprocess.check_call([pz_build.hipcc_path(), *pz_build.FLAGS, "-shared", "-fPIC", f"-DPZ_DIAGNOSTIC_BUILD={bits}u",
                           f"-I{REPO / 'include'}", f"-I{REPO / 'pika-zoo_amd' / 'csrc'}", "-o", str(variant),
                           str(REPO / "pika-zoo_amd" / "csrc" / "pz_kernels.hip")])
    lib = C.CDLL(str(variant))
    lib.pz_build_id.restype = C.c_char_p
    assert lib.pz_build_id() == b"diagnostic" and lib.pz_abi_version() == _native.ABI_VERSION
    assert pz_build.library_id(variant) != pz_build.source_id()
    assert {"pz_debug_read_stamps", "pz_debug_read_frame_stamps"} <= set(dynamic_pz_symbols(variant))  # (the product: not)
    saved, _native._lib, _native.LIB_PATH = (_native._lib, _native.LIB_PATH), None, variant
    try:
        with pytest.raises(_native.PikazooNativeError, match="stale"):
            _native.load()
    finally:
        _native._lib, _native.LIB_PATH = saved


def test_header_layout_matches_oracle_layout(oracle):
    """The product header and the oracle declare the state layout independently; keep them equal."""
    from pikazoo_amd import _native

    text = (REPO / "include" / "pikazoo_hip.h").read_text()
    assert "#define PZ_STATE_WORDS 44" in text and "PZ_B_X = 26" in text and "PZ_E_SCORE_P1 = 38" in text
    assert f"#define PZ_PACKED_BYTES_PER_GAME {_native.PACKED_BYTES_PER_GAME}" in text
    assert f"#define PZ_SCENERY_WORDS {_native.SCENERY_WORDS}" in text and f"#define PZ_ABI_VERSION {_native.ABI_VERSION}" in text
    assert C.sizeof(oracle.Config) == C.sizeof(_native.PzConfig)
    for (n1, t1), (n2, t2) in zip(oracle.Config._fields_, _native.PzConfig._fields_):
        assert n1 == n2 and C.sizeof(t1) == C.sizeof(t2)


def test_no_cpu_fallback_and_no_oracle_in_product():
    from pikazoo_amd import pikazoo_v0

    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            pikazoo_v0.env(num_envs=4)
    with pytest.raises(RuntimeError):
        pikazoo_v0.env(num_envs=4, device="cpu")
    # nothing under the product tree may reference the oracle
    for p in (REPO / "pika-zoo_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".hpp", ".h", ".cpp"):
            assert "oracle" not in p.read_text().lower().replace("oracle/", "oracle/") or p.name == "never", p
    # ... and neither may the tuning tools, the examples or the public header: oracle/ is for tests/, smoke() and
    # bench.py's cpu_baseline / in-run parity leg only
    for d in ("tools", "examples", "include"):
        for p in (REPO / d).rglob("*"):
            if p.suffix in (".py", ".sh", ".c", ".h"):
                text = p.read_text()
                assert "pz_oracle" not in text and "from oracle" not in text and "import oracle" not in text, p


def test_spaces():
    from pikazoo_amd.env import OBS_HIGH, OBS_LOW
    from pikazoo_amd.spaces import Box, Discrete

    d = Discrete(18)
    assert d.n == 18 and d.contains(17) and not d.contains(18) and 0 <= d.sample() < 18
    b = Box(OBS_LOW, OBS_HIGH, shape=(35,), dtype=np.int32)
    assert b.shape == (35,) and b.contains(OBS_LOW) and not b.contains(OBS_LOW - 1)
    # pikazoo_env.py:485-562
    assert OBS_LOW.tolist()[:13] == [32, 108, -15, -1, -2, 0, 0, 0, 0, 0, 0, 0, 0]
    assert OBS_HIGH.tolist()[26:] == [432, 252, 432, 252, 432, 252, 20, 124, 1]
    assert OBS_LOW.tolist()[26:] == [20, 0, 0, 0, 0, 0, -20, -124, 0]


def test_sharding_covers_all_lanes_once():
    from pikazoo_amd.dist import shard_for_rank, weak_shard

    for n, w in [(65536, 8), (10, 3), (7, 8), (524288, 8)]:
        shards = [shard_for_rank(n, r, w) for r in range(w)]
        assert sum(s.n_local for s in shards) == n
        pos = 0
        for s in shards:
            assert s.env_id_base == pos
            pos += s.n_local
    s = weak_shard(65536, 3, 8)
    assert (s.n_local, s.env_id_base, s.n_global) == (65536, 196608, 524288)


def test_a_rank_finds_its_gpu_whatever_the_launcher_left_visible(monkeypatch):
    """LOCAL_RANK is the device index when every rank sees the node's GPUs (torchrun); a launcher that narrows each rank
    to one visible device leaves index 0 everywhere; without a device the rank number stands (and RCCL is not tried)."""
    import torch
    from pikazoo_amd import dist

    for count, expect in ((8, [0, 1, 2, 3, 4, 5, 6, 7]), (1, [0] * 8), (4, [0, 1, 2, 3, 0, 1, 2, 3]), (0, list(range(8)))):
        monkeypatch.setattr(torch.cuda, "device_count", lambda count=count: count)
        assert [dist.local_device_index(r) for r in range(8)] == expect


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "pika-zoo_amd")); sys.path.insert(0, sys.argv[1])
from pikazoo_amd import dist
from oracle import pz_oracle as po
rank, world, _ = dist.init_from_env("gloo")
N, STEPS = 96, 300
sh = dist.shard_for_rank(N, rank, world)
# each rank steps its own shard with the CPU oracle standing in for the GPU (test only)
env = po.OracleEnv(sh.n_local, po.make_config(is_player2_computer=True, winning_score=2, seed=7, env_id_base=sh.env_id_base))
env.reset()
eps = env.rollout_random(3, 0, STEPS)
steps, episodes = dist.all_reduce_sum([sh.n_local * STEPS, eps])
tmax = dist.all_reduce_max(float(rank + 1))
dist.barrier()
np.save(os.path.join(sys.argv[2], f"state_{rank}.npy"), env.state)
if rank == 0:
    open(os.path.join(sys.argv[2], "agg.txt"), "w").write(f"{steps} {episodes} {tmax}")
'''


_FALLBACK_WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "pika-zoo_amd"))
from pikazoo_amd import dist
rank, world, _ = dist.init_from_env("nccl")   # no GPU in this container: RCCL cannot come up
total, = dist.all_reduce_sum([rank + 1])
if rank == 0:
    open(os.path.join(sys.argv[2], "fallback.txt"), "w").write(
        f"{dist.backend_name()} {world} {total} {dist.rccl_ranks()} {dist.fallback_note()}")
'''


def test_counters_fall_back_to_gloo_when_rccl_cannot_start(tmp_path):
    """RCCL only carries the aggregate counters; if it cannot start, the ranks agree -- collectively, over the gloo
    control group -- on leaving them there and say so: `rccl_ranks` is then 0, not the world size."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (RCCL starts fine with one)")
    script = tmp_path / "worker.py"
    script.write_text(_FALLBACK_WORKER)
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(REPO),
                           str(tmp_path)], env=env, timeout=600)
    backend, world, total, rccl, note = (tmp_path / "fallback.txt").read_text().split(" ", 4)
    assert (backend, world, total, rccl) == ("gloo", "2", "3", "0") and "unavailable" in note


_HANG_WORKER = r'''
import os, sys, threading, time
sys.path.insert(0, os.path.join(sys.argv[1], "pika-zoo_amd"))
from pikazoo_amd import dist
rank = int(os.environ["RANK"])
# stand-ins for a node whose RCCL rendezvous never answers on rank 1 (no GPU here): the preconditions hold, rank 0's
# communicator "comes up", rank 1's start blocks for ever
dist._device_ready = lambda index: (True, None)
dist._shares_a_gpu = lambda rank, world, index: False
dist._start_rccl = (lambda r, w, i: threading.Event().wait()) if rank == 1 else (lambda r, w, i: (None, "stand-in: no RCCL on this box"))
t0 = time.time()
rank, world, _ = dist.init_from_env("nccl")
total, = dist.all_reduce_sum([rank + 1])             # the counters still travel: over gloo
open(os.path.join(sys.argv[2], f"hang_{rank}.txt"), "w").write(
    f"{dist.backend_name()} {world} {total} {dist.rccl_ranks()} {int(dist.left_behind())}{int(dist.left_behind_anywhere())} {time.time() - t0:.1f} {dist.fallback_note()}")
print(f"rank {rank} done", flush=True)
dist.shutdown()                                      # must not block on the start that was left behind
raise SystemExit("shutdown() returned although an RCCL start was left behind in the job")
'''


def test_an_rccl_start_that_hangs_is_given_up_on_and_the_job_still_ends(tmp_path):
    """RCCL's rendezvous / first all-reduce can HANG instead of failing (a fabric link that never answers).  The start
    runs in a thread the rank can give up on (dist._start_rccl_bounded): after PIKAZOO_RCCL_START_TIMEOUT_S the rank
    reports failure, every rank agrees over gloo to leave the counters there, the line's `dist_note` says what happened,
    and `dist.shutdown()` ends every rank without touching the half-made communicator.  Rehearsed here with a start that
    blocks for ever on rank 1 of 2."""
    script = tmp_path / "worker.py"
    script.write_text(_HANG_WORKER)
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1", PIKAZOO_RCCL_START_TIMEOUT_S="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(REPO), str(tmp_path)],
                       env=env, timeout=300, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for rank in (0, 1):
        backend, world, total, rccl, left, took, note = (tmp_path / f"hang_{rank}.txt").read_text().split(" ", 6)
        assert (backend, world, total, rccl) == ("gloo", "2", "3", "0") and float(took) < 60
        assert left == f"{rank}1"  # only rank 1 left a start behind, and EVERY rank knows that somebody did ...
        assert "unavailable" in note and ("did not come up within 2 s" in note) == (rank == 1)  # ... and says so


def test_two_rank_sharding_equals_single_process(tmp_path, oracle):
    """world_size-2 gloo run: shards stepped independently + one counter all-reduce give the same
    trajectories and totals as the single-process batch (env ids are global)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(REPO),
                           str(tmp_path)], env=env, timeout=600)
    whole = oracle.OracleEnv(96, oracle.make_config(is_player2_computer=True, winning_score=2, seed=7))
    whole.reset()
    eps = whole.rollout_random(3, 0, 300)
    got = np.concatenate([np.load(tmp_path / "state_0.npy"), np.load(tmp_path / "state_1.npy")], axis=1)
    assert np.array_equal(got, whole.state)
    steps, episodes, tmax = (tmp_path / "agg.txt").read_text().split()
    assert int(steps) == 96 * 300 and int(episodes) == eps and float(tmax) == 2.0


_WORKER8 = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "pika-zoo_amd")); sys.path.insert(0, sys.argv[1])
from pikazoo_amd import dist
from oracle import pz_oracle as po
rank, world, _ = dist.init_from_env("gloo")
PER_GPU, SAMPLE, STEPS = 65536, 48, 200
sh = dist.weak_shard(PER_GPU, rank, world)                 # BASELINE config 4: 8 x 65 536 games
# the first SAMPLE games of the rank's range on the CPU oracle (it stands in for the GPU here; test only)
env = po.OracleEnv(SAMPLE, po.make_config(winning_score=2, seed=5, env_id_base=sh.env_id_base))
env.reset()
eps = env.rollout_random(9, 0, STEPS)
n_total, episodes = dist.all_reduce_sum([sh.n_local, eps], device=dist.collective_device(None))
rows = dist.all_gather_rows([rank, sh.env_id_base, eps])
dist.barrier()
np.save(os.path.join(sys.argv[2], f"state8_{rank}.npy"), env.state)
if rank == world - 1:
    open(os.path.join(sys.argv[2], "agg8.txt"), "w").write(repr((n_total, episodes, rows, sh.n_global, dist.world_size(),
                                                                 dist.backend_name(), dist.rccl_ranks())))
'''


def test_config_4_at_its_real_rank_count_eight_ranks_over_gloo(tmp_path, oracle):
    """BASELINE config 4 -- 524 288 games over 8 ranks -- at its real rank count.  (Eight processes may not share the
    one GPU of the builder's box -- its process guard allows six on the card, tests/test_gpu_multirank.py runs bench.py with four ranks
    -- so the eight ranks meet here, over gloo, with the oracle stepping a sample of every rank's game range:) rank r owns
    global games [r * 65 536, (r + 1) * 65 536), the counters sum to the job's totals on every rank, the gathered
    per-rank rows arrive in rank order, and every rank's games are the games a single process steps under those ids."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8)
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                           "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(REPO),
                           str(tmp_path)], env=env, timeout=900)
    n_total, episodes, rows, n_global, world, backend, rccl = eval((tmp_path / "agg8.txt").read_text())  # noqa: S307
    assert (n_total, n_global, world, backend, rccl) == (524288, 524288, 8, "gloo", 0)
    assert [int(r[0]) for r in rows] == list(range(8)) and [int(r[1]) for r in rows] == [r * 65536 for r in range(8)]
    total = 0
    for r in range(8):
        one = oracle.OracleEnv(48, oracle.make_config(winning_score=2, seed=5, env_id_base=r * 65536))
        one.reset()
        total += one.rollout_random(9, 0, 200)
        assert np.array_equal(np.load(tmp_path / f"state8_{r}.npy"), one.state), r
        assert int(rows[r][2]) >= 0
    assert episodes == total == sum(int(r[2]) for r in rows) and total > 0


def test_argument_validation_without_a_gpu(built_lib):
    """Argument errors are reported before anything touches the device (no launch on these paths)."""
    from pikazoo_amd import _native

    lib = _native.load()
    cfg = _native.PzConfig()
    cfg.winning_score, cfg.serve_mode = 15, 0
    fake = C.c_void_p(4096)  # never dereferenced: every call below returns before a launch
    assert lib.pz_init(None, 0, 0, C.byref(cfg), None) == -1            # PZ_E_NULL
    assert lib.pz_init(fake, -1, 0, C.byref(cfg), None) == -2           # PZ_E_SIZE
    assert lib.pz_init(fake, 8, 4, C.byref(cfg), None) == -2            # stride < n
    assert lib.pz_init(fake, 0, 24_403_224, C.byref(cfg), None) == -2   # more games than one launch addresses
    assert lib.pz_init(fake, 0, 24_403_223, C.byref(cfg), None) == 0    # empty batch: no-op
    cfg.serve_mode = 3
    assert lib.pz_init(fake, 0, 0, C.byref(cfg), None) == -3            # PZ_E_CONFIG
    cfg.serve_mode, cfg.winning_score = 0, 0
    assert lib.pz_init(fake, 0, 0, C.byref(cfg), None) == -3
    cfg.winning_score = 1
    assert lib.pz_step(fake, 0, 0, C.byref(cfg), None, fake, fake, fake, fake, fake, fake, None, None, None) == -1
    assert lib.pz_step_random(fake, 0, 0, C.byref(cfg), 1, 0, 0, fake, fake, fake, fake, fake, None, None, None,
                              None) == -2
    assert lib.pz_observe(fake, 0, 0, 0, 0, C.c_void_p(4100), fake, None) == -4   # PZ_E_ALIGN
    assert lib.pz_step(fake, 0, 0, C.byref(cfg), fake, fake, fake, fake, fake, fake, fake, None, None, None) == 0
    for bad in (_native.PzFlightTables(None, 4104),   # power-hit table not 16-byte aligned
                _native.PzFlightTables(4098, 4096)):  # landing table not 4-byte aligned (it is read dword-wise)
        assert lib.pz_step(fake, 8, 8, C.byref(cfg), fake, fake, fake, fake, fake, fake, fake, None, C.byref(bad),
                           None) == -4
        assert lib.pz_step_random(fake, 8, 8, C.byref(cfg), 1, 0, 1, fake, fake, fake, fake, fake, None, None,
                                  C.byref(bad), None) == -4
    assert lib.pz_build_flight_tables(None, None, None) == -1
    # the prepared-argument form of pz_step: a host block the caller owns, validated at bind time
    block = C.create_string_buffer(int(lib.pz_step_bound_bytes()))
    assert lib.pz_step_bound(block, fake, fake, None) == -3                                        # not bound yet
    assert lib.pz_step_bind(None, fake, 8, 8, C.byref(cfg), fake, fake, fake, fake, fake, None, None) == -1
    assert lib.pz_step_bind(block, fake, 8, 4, C.byref(cfg), fake, fake, fake, fake, fake, None, None) == -2
    assert lib.pz_step_bind(block, fake, 8, 8, C.byref(cfg), C.c_void_p(4100), fake, fake, fake, fake, None, None) == -4
    assert lib.pz_step_bind(block, fake, 0, 0, C.byref(cfg), fake, fake, fake, fake, fake, None, None) == 0
    assert lib.pz_step_bound(block, fake, fake, None) == 0                                         # n == 0: no launch
    assert lib.pz_step_bound(block, None, fake, None) == -1
    assert lib.pz_count_packed_misfits(None, 8, 8, fake, None) == -1 and lib.pz_count_packed_misfits(fake, 8, 4, fake, None) == -2
    # the placement probe: at least one [1024 x 8960]-byte frame per buffer, 16-byte aligned, one of the two may be NULL
    assert lib.pz_probe_frame_bytes() == 1024 * 8960
    assert lib.pz_probe_write(None, None, 1 << 30, None) == -1 and lib.pz_probe_write(fake, None, 1024 * 8960 - 1, None) == -2
    assert lib.pz_probe_write(C.c_void_p(4100), fake, 1 << 30, None) == -4
    # the landing table: its 2-byte entries (an odd number) + 2 bytes of padding -- the look-up loads whole dwords
    assert lib.pz_flight_table_bytes(0) == 2 * 193 * 23 * 253 * 413 + 2 and lib.pz_flight_table_bytes(1) == 16 * 65 * 192 * 413


def test_no_vmem_store_data_hazard_in_device_code(built_lib, tmp_path):
    """A buffer store of more than 64 bits must not be followed directly by a VALU write of its data
    registers.  The compiler only guards that pattern for stores without an SGPR offset (LLVM
    GCNHazardRecognizer::createsVALUHazard); on gfx950 a store WITH one was seen picking up the new
    value in a quarter of the wave (observation words holding address bits).  flush_rows therefore
    uses per-wave descriptors with no SGPR offset; this scans the shipped code object so that a future
    kernel change cannot bring the pattern back unnoticed."""
    import shutil

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    lib = shutil.copy(built_lib, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, cwd=tmp_path, capture_output=True)
    objs = [p for p in tmp_path.iterdir() if "gfx950" in p.name]
    assert len(objs) == 1, [p.name for p in tmp_path.iterdir()]
    asm = subprocess.run([objdump, "-d", str(objs[0])], check=True, capture_output=True, text=True).stdout
    ins = [m.group(1) for m in (re.match(r"^\s+([a-z_0-9]+ .*?)\s*//", line) for line in asm.splitlines()) if m]
    stores = hazards = 0
    for k, text in enumerate(ins):
        m = re.match(r"(?:buffer|global|flat|scratch)_store_dwordx[34] (?:v\d+, |off, )?v\[(\d+):(\d+)\]", text)
        m = m or re.match(r"buffer_store_dwordx[34] v\[(\d+):(\d+)\]", text)
        if not m:
            continue
        stores += 1
        lo, hi = int(m.group(1)), int(m.group(2))
        states = 0  # wait states since the store: the hazard needs one (gfx90a: two); `s_nop N` supplies N + 1
        for nxt in ins[k + 1:k + 3]:
            if states >= 2 or nxt.startswith("s_endpgm"):  # (the next kernel's code follows an s_endpgm)
                break
            nop = re.match(r"s_nop (\d+)", nxt)
            states += int(nop.group(1)) + 1 if nop else 1
            w = re.match(r"v_(?!cmp|nop|readfirstlane|readlane)\w+ (?:v\[(\d+):(\d+)\]|v(\d+))", nxt)
            if w:
                a, b = (int(w.group(1)), int(w.group(2))) if w.group(1) else (int(w.group(3)),) * 2
                hazards += not (b < lo or a > hi)
    assert stores > 100, "the scan did not see the observation stores"
    assert hazards == 0


def test_counted_bytes_carry_their_build_and_go_stale_with_the_kernel(built_lib, monkeypatch):
    """profiles/traffic.json (the PMC-counted bytes bench.py prints as `roofline.traffic`) names the build its counters
    were taken on and the digest of each entry's kernel (tools/kernel_digest.py).  bench.load_traffic uses an entry on
    that build, or on another build whose kernel is instruction-for-instruction the same; after a kernel change without
    a new profile it returns nothing and the line says `traffic_stale` -- never a stale figure without a warning."""
    import json

    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(REPO / "tools"))
    import bench
    import kernel_digest
    from pikazoo_amd import _native

    committed = json.loads((REPO / "profiles" / "traffic.json").read_text())
    entries = {k: v for k, v in committed.items() if isinstance(v, dict)}
    assert len(entries) >= 15
    for key, e in entries.items():
        assert re.fullmatch(r"[0-9a-f]{16}", e.get("build_id") or ""), key
        assert re.fullmatch(r"[0-9a-f]{16}", e.get("kernel_digest") or ""), key
    if not kernel_digest.available():
        pytest.skip("llvm-objdump not available")
    e = dict(entries["random_random"])
    now = kernel_digest.digest(e["kernel"], built_lib)
    assert now is not None and kernel_digest.digest("pz::no_such_kernel", built_lib) is None

    def load(entry):
        monkeypatch.setattr(bench, "_TRAFFIC", {"random_random": entry})
        monkeypatch.setattr(bench, "_TRAFFIC_STATUS", {})
        return bench.load_traffic("random_random", entry["num_envs"]), dict(bench._TRAFFIC_STATUS.get("random_random", {}))

    # the same build: used as it is
    value, st = load(dict(e, build_id=_native.build_id()))
    assert value == e["hbm_bytes_per_launch"] and st["stale"] is False
    # another build, the kernel's instruction stream unchanged: still valid, and the line names the counters' build
    value, st = load(dict(e, build_id="0" * 16, kernel_digest=now))
    assert value == e["hbm_bytes_per_launch"] and st == {"build_id": "0" * 16, "stale": False,
                                                          "why": "another build, kernel instruction stream unchanged"}
    # another build and another kernel: stale, no figure
    value, st = load(dict(e, build_id="0" * 16, kernel_digest="f" * 16))
    assert value is None and st["stale"] is True and "CHANGED" in st["why"]
    # an entry without a digest (round 4's file) on another build: stale too
    old = {k: v for k, v in e.items() if k != "kernel_digest"}
    value, st = load(dict(old, build_id="0" * 16))
    assert value is None and st["stale"] is True
    # another batch size than the entry's: no figure, no verdict
    monkeypatch.setattr(bench, "_TRAFFIC", {"random_random": e})
    assert bench.load_traffic("random_random", 4096) is None


def _kernel_instructions(built_lib, tmp_path, needle):
    """The instruction texts of the first gfx950 kernel whose demangled name contains `needle`."""
    import shutil

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    lib = shutil.copy(built_lib, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, cwd=tmp_path, capture_output=True)
    objs = [p for p in tmp_path.iterdir() if "gfx950" in p.name]
    asm = subprocess.run([objdump, "-d", "--demangle", str(objs[0])], check=True, capture_output=True, text=True).stdout
    body, inside = [], False
    for line in asm.splitlines():
        head = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if head:
            if inside:
                break
            inside = needle in head.group(1)
        elif inside:
            m = re.match(r"^\s+([a-z_0-9]+(?: .*?)?)\s*//", line)
            if m:
                body.append(m.group(1))
    assert body, f"no kernel named *{needle}* in the code object"
    return body


@pytest.mark.parametrize("kernel", ["step_pair_kernel<false, true, false, false>", "step_pair_kernel<true, false, false, false>",
                                    "step_pair_kernel<false, true, false, true>"])
def test_early_stores_of_the_one_computer_launch_are_ordered_behind_the_partners_loads(built_lib, tmp_path, kernel):
    """pair_body, one computer player (config 3): the human player's wave stores columns IN FRONT of the exchange barrier
    that the computer's wave loads at the top of the launch.  The only thing that orders those stores behind those loads
    is an LDS hand-shake: the computer's wave sets a word after `s_waitcnt vmcnt(0)` on its loads, the human
    player's wave clears it first thing and stores early only if it reads it set.  This scans the shipped code object for
    exactly that shape, so that a later kernel change cannot quietly take the edge away:
      * the word is set (the magic value) with an s_waitcnt vmcnt(0) and no vector-memory load between it and the ds_write;
      * in the human player's wave every global store in front of the barrier comes after the ds_read of the word and the
        scalar compare with the magic value."""
    ins = _kernel_instructions(built_lib, tmp_path, kernel)
    magic = "0x10add0e5"
    flag = "offset:8192" if kernel.startswith("step_pair_kernel<false") else "offset:17152"  # word 2048 of the human wave's region
    sets = [k for k, t in enumerate(ins) if t.startswith("v_mov_b32") and magic in t]
    assert len(sets) == 1, sets
    at = sets[0]
    write = next(k for k in range(at, at + 6) if ins[k].startswith("ds_write_b32") and flag in ins[k])
    waits = [k for k in range(max(0, at - 12), at) if ins[k].startswith("s_waitcnt vmcnt(0)")]
    assert waits, "the flag is set without waiting for the wave's loads"
    assert not any(t.startswith(("buffer_load", "global_load", "flat_load")) for t in ins[waits[-1]:write]), "a load behind the wait"
    barriers = [k for k, t in enumerate(ins) if t.startswith("s_barrier")]
    assert len(barriers) == 2  # one per wave: the role branches are laid out one after the other
    clears = [k for k, t in enumerate(ins) if t.startswith("ds_write_b32") and flag in t and k != write]
    assert len(clears) == 1, clears
    clear = clears[0]
    human_barrier = next(b for b in barriers if b > clear)
    computer_barrier = next(b for b in barriers if b != human_barrier)
    assert write < computer_barrier, "the word is set behind the computer's barrier"
    stores = ("buffer_store", "global_store", "flat_store")
    reads = [k for k in range(clear, human_barrier) if ins[k].startswith("ds_read_b32") and flag in ins[k]]
    checks = [k for k in range(clear, human_barrier) if ins[k].startswith("s_cmp") and magic in ins[k]]
    assert reads and checks and reads[0] < checks[0]
    early = [k for k in range(clear, human_barrier) if ins[k].startswith(stores)]
    assert early, "no early stores left: drop the hand-shake with them"
    assert min(early) > checks[0], "a global store in front of the barrier is not behind the hand-shake"


def test_no_scratch_and_bounded_scalar_spills_in_the_step_kernels(built_lib, tmp_path):
    """Code-object notes of the shipped library: no step / rollout kernel touches scratch memory or spills a VGPR, and
    the k-frame kernels of a plain configuration (no fused wrapper, no statistics: the PLAIN instantiations the host
    picks for them) stay within the scalar register file -- round 3's generic forms carried 61-81 SGPR spills around
    their frame loop.  "Touches": the single-frame kernels and the PLAIN forms reserve no scratch at all.  A GENERIC
    k-frame form may carry a few dead spill slots in its descriptor: the scalar register allocator leaves them behind in
    one or another of those kernels (`-mllvm -print-after=prologepilog`: a 16- and a 4-byte frame object, every SGPR slot
    dead) -- which one changes with unrelated edits, round 6 saw it move three times while the spill counts fell -- so
    the invariant held here is the one that matters at run time: its code holds no scratch instruction."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf not available")
    sys.path.insert(0, str(REPO / "tools"))
    import kernel_notes

    rows = kernel_notes.notes(Path(built_lib))
    kernels = {name.split("(")[0].replace("void pz::", ""): r for name, r in rows}
    step = {k: r for k, r in kernels.items() if k.startswith(("step_kernel<", "step_pair_kernel<", "rollout_pair_kernel<"))}
    assert len(step) > 100, "the scan did not see the step kernels"

    def generic_k_frame_form(name):  # <..., MODE 2 / 3, ..., PLAIN false>
        args = [x.strip() for x in name[name.index("<") + 1:name.rindex(">")].split(",")]
        return args[2] in ("2", "3") and args[-1] == "false"

    for name, r in step.items():
        assert r[".vgpr_spill_count"] == 0, f"{name} spills VGPRs"
        reserved = r[".private_segment_fixed_size"]
        if reserved:
            assert generic_k_frame_form(name) and reserved <= 64, f"{name} reserves {reserved} B of scratch"
            ins = _kernel_instructions(built_lib, tmp_path, name)
            assert not any(t.startswith("scratch_") or "s[0:3]" in t for t in ins), f"{name} uses scratch memory"
    # <AI1, AI2, MODE (2 rollout, 3 tape), SPARSE, SCOUT, PACKED, OBS16, PLAIN> / <AI1, AI2, MODE, PACKED, OBS16, PLAIN>
    # (the human-vs-human rollout keeps its generic form -- the PLAIN one is not even instantiated: it measured slower at
    # that launch's write ceiling, pz_kernels.hip launch_step_players -- with 53 spills where round 3 had 74)
    bounds = {"step_kernel<false, false, 2, false, 0, false, false, false>": 60,
              "rollout_pair_kernel<false, true, 2, false, false, true>": 25,
              "step_kernel<false, false, 3, false, 0, false, false, true>": 25,
              "rollout_pair_kernel<false, true, 3, false, false, true>": 25,
              "step_pair_kernel<false, false, false, false>": 0,
              "step_pair_kernel<false, true, false, false>": 8}
    for name, most in bounds.items():
        assert name in step, f"{name} is not in the library"
        assert step[name][".sgpr_spill_count"] <= most, f"{name}: {step[name]['.sgpr_spill_count']} SGPR spills (> {most})"

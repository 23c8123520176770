"""Multi-GPU sharding of the batch: one process per GPU, no collective on the step path.

Games are independent (each reference env owns its physics and RNG, pikazoo_env.py:96-98), so
rank r simply owns the contiguous global lane range ``[r*n_local, (r+1)*n_local)``;
``env_id_base`` feeds the Philox counter, which makes trajectories independent of the GPU count.
The only collective is a SUM all-reduce of a few int64 counters per reporting window
(``torch.distributed``: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass
from datetime import timedelta

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Shard:
    rank: int
    world_size: int
    n_local: int
    env_id_base: int
    n_global: int


def shard_for_rank(n_global: int, rank: int, world_size: int) -> Shard:
    """Contiguous split of ``n_global`` lanes; the first ``n_global % world_size`` ranks get one more."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    q, r = divmod(int(n_global), int(world_size))
    n_local = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return Shard(rank, world_size, n_local, base, int(n_global))


def weak_shard(n_per_gpu: int, rank: int, world_size: int) -> Shard:
    """Weak scaling: every rank owns ``n_per_gpu`` lanes of a ``world_size * n_per_gpu`` job."""
    return Shard(rank, world_size, int(n_per_gpu), rank * int(n_per_gpu), world_size * int(n_per_gpu))


_COUNTERS = None       # the RCCL group the counters travel over (None: the gloo control group / no group at all)
_FALLBACK_NOTE = None
_LEFT_BEHIND = False   # an RCCL start that did not return in time is still running in a daemon thread of this process
_LEFT_BEHIND_ANYWHERE = False  # ... in some rank of the job (agreed on over gloo right after the start)
RCCL_START_TIMEOUT_S = 90.0   # (PIKAZOO_RCCL_START_TIMEOUT_S) how long a rank waits for its communicator + probe all-reduce


def _shares_a_gpu(rank: int, world: int, index: int):
    """Collective over the control group: do two ranks of one host use the same GPU?  (RCCL refuses that --
    "Duplicate GPU detected" -- so every rank learns it here, before any of them has touched RCCL.)"""
    import socket

    mine = (socket.gethostname(), os.environ.get("ROCR_VISIBLE_DEVICES"), os.environ.get("HIP_VISIBLE_DEVICES"),
            os.environ.get("CUDA_VISIBLE_DEVICES"), int(index))
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    return len(set(everyone)) < world


def local_device_index(local_rank: int) -> int:
    """The GPU of this rank: LOCAL_RANK when every rank sees all of the node's GPUs (torchrun's default), folded into
    the visible ones when a launcher narrowed each rank's *_VISIBLE_DEVICES (then usually to one: index 0 everywhere)."""
    count = torch.cuda.device_count()  # (does not initialise the GPU)
    return int(local_rank) % count if count > 0 else int(local_rank)


def _device_ready(index: int):
    """This rank's own precondition for RCCL: (ok, why not)."""
    try:
        if not torch.cuda.is_available():
            return False, "no HIP device visible"
        torch.cuda.set_device(index)
        return True, None
    except Exception as exc:  # noqa: BLE001
        return False, f"{type(exc).__name__}: {exc}"[:300]


def _start_rccl(rank: int, world: int, index: int):
    """This rank's attempt at an RCCL group over all ranks: (group or None, why not).  `dist.new_group` is a collective:
    every rank calls this or none does (init_from_env agrees on that over gloo first).  A call that HANGS inside RCCL is
    not caught here but by `_start_rccl_bounded`, which runs this in a thread it can give up on.

    A communicator that is given up on keeps its own threads: ProcessGroupNCCL's watchdog would throw (or SIGABRT the
    whole process) when the stuck collective passes the group's timeout, and its heartbeat monitor kills a process whose
    watchdog stopped answering after 8-10 minutes -- a job that went on over gloo would be killed mid-run by a group it
    no longer uses.  Both are switched off for this process BEFORE the group is created (they are read at construction;
    a value the user exported wins): TORCH_NCCL_ASYNC_ERROR_HANDLING=0 (a timed-out collective aborts nothing) and
    TORCH_NCCL_ENABLE_MONITORING=0 (no heartbeat kill).  The group's timeout stays at ten minutes as the second line: with
    either variable overridden, that is how long a job that left a start behind may still run (`left_behind_anywhere()`
    tells it so; bench.py prints it and ends through `shutdown()`)."""
    try:
        for var in ("TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCH_NCCL_ENABLE_MONITORING"):
            os.environ.setdefault(var, "0")
        dev = torch.device("cuda", index)
        torch.cuda.set_device(dev)  # (the current device is per thread)
        group = dist.new_group(backend="nccl", timeout=timedelta(seconds=600))
        probe = torch.ones(1, dtype=torch.int64, device=dev)  # the communicator works before anything is timed
        dist.all_reduce(probe, group=group)
        if int(probe.item()) != world:
            raise RuntimeError(f"RCCL all-reduce of ones gave {int(probe.item())}, expected {world}")
        return group, None
    except Exception as exc:  # noqa: BLE001 - RCCL only carries the aggregate counters: keep the job alive
        return None, f"{type(exc).__name__}: {exc}"[:300]


def _start_rccl_bounded(rank: int, world: int, index: int, seconds: float | None = None):
    """`_start_rccl` with a bound on how long it may take.  RCCL's rendezvous and first all-reduce FAIL in most broken
    set-ups, and then `_start_rccl` returns the reason; in some (a fabric link that never answers, a peer that died in
    its own start) they HANG.  The attempt therefore runs in a daemon thread: when it has not returned after `seconds`,
    this rank reports failure like any other -- the ranks then agree over gloo to leave the counters there -- and the
    thread is left behind (`left_behind()`: the process must then end through `shutdown()`, which does not try to tear
    the half-made communicator down).  Nothing on the step path ever touches RCCL -- but a probe all-reduce that was
    already enqueued when the rank gave up keeps an RCCL kernel spinning on the GPU, holding CUs and a queue slot beside
    whatever is timed afterwards: figures taken in such a job are FLAGGED (`left_behind_anywhere()`; bench.py:
    `rccl_left_behind`), not trusted."""
    global _LEFT_BEHIND
    import threading

    if seconds is None:
        try:
            seconds = float(os.environ["PIKAZOO_RCCL_START_TIMEOUT_S"])
        except (KeyError, ValueError):
            seconds = RCCL_START_TIMEOUT_S
    box = {}
    worker = threading.Thread(target=lambda: box.setdefault("result", _start_rccl(rank, world, index)), daemon=True,
                              name="pikazoo-rccl-start")
    worker.start()
    worker.join(seconds)
    if worker.is_alive():
        _LEFT_BEHIND = True
        return None, f"RCCL did not come up within {seconds:.0f} s on rank {rank} (its start was left behind)"
    return box.get("result", (None, "the RCCL start thread ended without a result"))


def left_behind() -> bool:
    """True when an RCCL start of this process never returned (see `_start_rccl_bounded`)."""
    return _LEFT_BEHIND


def left_behind_anywhere() -> bool:
    """True when an RCCL start never returned in SOME rank of the job (agreed on over gloo by `init_from_env`): a
    communicator kernel may still be spinning on that rank's GPU, so the job's timings are suspect and the job should end
    soon (through `shutdown()`; a retry belongs in a fresh child process)."""
    return _LEFT_BEHIND_ANYWHERE


def shutdown(exit_code: int = 0):
    """End of a job: destroy the process groups -- unless an RCCL start was left behind in this process: tearing its
    half-made communicator down may block for ever, so the ranks meet at a last gloo barrier, flush their output and
    leave through `os._exit` (the daemon thread and RCCL's own threads go with the process)."""
    if not dist.is_initialized():
        return
    try:
        anyone = torch.tensor([1 if _LEFT_BEHIND else 0], dtype=torch.int64)
        dist.all_reduce(anyone, op=dist.ReduceOp.MAX)  # (over gloo: the default group)
        hung = int(anyone.item()) != 0
    except Exception:  # noqa: BLE001
        hung = _LEFT_BEHIND
    if not hung:
        dist.destroy_process_group()
        return
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(int(exit_code))


def init_from_env(backend: str | None = None, device_index: int | None = None) -> tuple[int, int, int]:
    """Join the process group described by RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun).

    Returns (rank, world_size, local_rank).  Single-process runs (no WORLD_SIZE) skip the group.
    `device_index`: the GPU this rank uses when it is not LOCAL_RANK (rehearsals with several ranks on one GPU).

    The default group is ALWAYS gloo (barriers, and every decision the ranks have to take together); with
    `backend` "nccl" (the default on a GPU box) an RCCL group over all ranks is added for the counters.  Whether
    RCCL is used is decided COLLECTIVELY: every rank reports over gloo whether its communicator came up and its
    probe all-reduce gave the world size, and if any rank failed, all of them leave the counters on gloo and say
    so (`fallback_note`, `rccl_ranks() == 0`).  The preconditions (a usable device on every rank, no two ranks on one GPU)
    are agreed on over gloo first, so either every rank enters RCCL's rendezvous or none does; an RCCL start that HANGS
    instead of failing is given up on after `RCCL_START_TIMEOUT_S` (`_start_rccl_bounded`) and reads as a failure of that
    rank -- the job goes on over gloo and ends through `shutdown()`.  (`_shares_a_gpu` compares
    host name, the *_VISIBLE_DEVICES strings and the device index: two hosts of the same name would read as one.)"""
    global _COUNTERS, _FALLBACK_NOTE, _LEFT_BEHIND_ANYWHERE
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under a launcher (WORLD_SIZE set, even to 1) the group is created, so a 1-rank torchrun run
    # exercises exactly the code path of the N-rank runs
    if "WORLD_SIZE" in os.environ and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=180))
        if backend == "nccl":
            index = local_device_index(local_rank) if device_index is None else int(device_index)
            # every precondition is agreed on over gloo BEFORE anybody enters RCCL's own rendezvous (dist.new_group is a
            # collective: a rank that skipped it would leave the others waiting in its store barrier for two minutes)
            ready, why = _device_ready(index)
            everyone_ready = torch.tensor([1 if ready else 0], dtype=torch.int64)
            dist.all_reduce(everyone_ready, op=dist.ReduceOp.MIN)
            if int(everyone_ready.item()) == 0:
                group, why = None, why or "another rank has no usable HIP device"
            elif _shares_a_gpu(rank, world, index):
                group, why = None, "two ranks share one GPU (RCCL refuses duplicate GPUs)"
            else:
                group, why = _start_rccl_bounded(rank, world, index)
            # [came up here, no start left behind here]: MIN over gloo -- one failure anywhere moves everybody, and everybody
            # learns whether some rank still has a start (and maybe its probe all-reduce's kernel) hanging
            verdict = torch.tensor([1 if group is not None else 0, 0 if _LEFT_BEHIND else 1], dtype=torch.int64)
            dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
            _LEFT_BEHIND_ANYWHERE = int(verdict[1].item()) == 0
            if int(verdict[0].item()) == 1:
                _COUNTERS = group
            else:
                _COUNTERS = None  # (a communicator that did come up on this rank is left alone: tearing it down may block)
                _FALLBACK_NOTE = ("nccl (RCCL) unavailable: " + (why or "another rank could not start it"))[:300]
                if rank == 0:
                    print(f"[pikazoo_amd.dist] {_FALLBACK_NOTE}; the counters stay on gloo", file=sys.stderr, flush=True)
        elif backend != "gloo":
            raise ValueError(f"backend must be 'nccl' or 'gloo', got {backend!r}")
    return rank, world, local_rank


def fallback_note():
    """Why the counters are not on the requested nccl backend (None when they are, or when it was never asked for)."""
    return _FALLBACK_NOTE


def _on_counters_device(t: torch.Tensor, device):
    return t.to(device) if (_COUNTERS is not None and device is not None) else t


def all_reduce_sum(values, device=None) -> list[int]:
    """SUM over ranks of a short list of integer counters (one tiny all-reduce: RCCL when it is up, else gloo)."""
    t = _on_counters_device(torch.tensor([int(v) for v in values], dtype=torch.int64), device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_COUNTERS)
    return [int(v) for v in t.tolist()]


def all_reduce_max(value: float, device=None) -> float:
    t = _on_counters_device(torch.tensor([float(value)], dtype=torch.float64), device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_COUNTERS)
    return float(t.item())


def all_gather_rows(row, device=None) -> list[list[float]]:
    """Every rank's short list of floats, in rank order (one all-gather)."""
    t = _on_counters_device(torch.tensor([float(v) for v in row], dtype=torch.float64), device)
    if not dist.is_initialized():
        return [t.tolist()]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t, group=_COUNTERS)
    return [o.tolist() for o in out]


def get_rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def world_size() -> int:
    """Ranks in the process group as the backend sees them (1 without a group)."""
    return dist.get_world_size() if dist.is_initialized() else 1


def backend_name() -> str:
    """What carries the counters: "nccl" (= RCCL on ROCm), "gloo", or "none" for a single process without a group."""
    if not dist.is_initialized():
        return "none"
    return "nccl" if _COUNTERS is not None else "gloo"


def rccl_ranks() -> int:
    """Ranks in the RCCL group the counters travel over; 0 when they travel over gloo (or there is no group)."""
    return dist.get_world_size(group=_COUNTERS) if _COUNTERS is not None else 0


def barrier():
    if dist.is_initialized():
        dist.barrier()


def collective_device(gpu_device):
    """Device the counters of all_reduce_* must live on: the GPU under nccl (RCCL), the CPU under gloo."""
    return gpu_device if _COUNTERS is not None else None

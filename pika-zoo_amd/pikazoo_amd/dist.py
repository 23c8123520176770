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


def init_from_env(backend: str | None = None, device_index: int | None = None) -> tuple[int, int, int]:
    """Join the process group described by RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun).

    Returns (rank, world_size, local_rank).  Single-process runs (no WORLD_SIZE) skip the group.
    `device_index`: the GPU this rank uses when it is not LOCAL_RANK (rehearsals with several ranks on one GPU)."""
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
        if backend == "nccl":
            try:
                index = local_rank if device_index is None else int(device_index)
                torch.cuda.set_device(index)
                dev = torch.device("cuda", index)
                dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
                probe = torch.ones(1, dtype=torch.int64, device=dev)  # the communicator works before anything is timed
                dist.all_reduce(probe)
                if int(probe.item()) != world:
                    raise RuntimeError(f"RCCL all-reduce of ones gave {int(probe.item())}, expected {world}")
            except Exception as exc:  # noqa: BLE001 - RCCL only carries the aggregate counters: keep the job alive
                global _FALLBACK_NOTE
                _FALLBACK_NOTE = f"nccl (RCCL) unavailable: {type(exc).__name__}: {exc}"[:300]
                print(f"[pikazoo_amd.dist] {_FALLBACK_NOTE}; the counters fall back to gloo", file=sys.stderr, flush=True)
                if dist.is_initialized():
                    try:
                        dist.destroy_process_group()
                    except Exception:  # noqa: BLE001
                        pass
                # same rendezvous (under torchrun the store is served by the launcher's agent, not by rank 0)
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=180))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


_FALLBACK_NOTE = None


def fallback_note():
    """Why the process group is not on the requested nccl backend (None when it is, or was never asked for)."""
    return _FALLBACK_NOTE


def all_reduce_sum(values, device=None) -> list[int]:
    """SUM over ranks of a short list of integer counters (one tiny all-reduce)."""
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def all_reduce_max(value: float, device=None) -> float:
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def get_rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def world_size() -> int:
    """Ranks in the process group as the backend sees them (1 without a group)."""
    return dist.get_world_size() if dist.is_initialized() else 1


def backend_name() -> str:
    """"nccl" (= RCCL on ROCm), "gloo", or "none" for a single process without a group."""
    return str(dist.get_backend()) if dist.is_initialized() else "none"


def barrier():
    if dist.is_initialized():
        dist.barrier()


def collective_device(gpu_device):
    """Device the counters of all_reduce_* must live on: the GPU under nccl (RCCL), the CPU under gloo."""
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return gpu_device
    return None

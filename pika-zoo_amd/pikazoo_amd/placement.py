"""Rank-aware placement of the two observation tensors a k-frame launch writes.

MI355X's device memory answers two concurrent write streams a quarter faster when they go to different thirds of
it (the three ranks of the 12-high HBM3E stacks, laid out as coarse regions of the physical address space) than
when both go to the same third: ``pz_rollout_random`` / ``pz_step_many`` run at 2.8 instead of 3.6 us per frame
(65 536 games) when their two ``[k, N, 35]`` observation tensors do not share a rank.  Which rank an allocation
lands in is the driver's choice -- consecutive allocations of a fresh process usually share one -- but it can be
MEASURED: ``pz_probe_write`` issues the launch's store pattern into one buffer, the other, and both, and two buffers
of one rank take as long together as one after the other, two of different ranks 0.8 of that.

:func:`alloc_pair` allocates the pair, probes it, and when the two share a rank walks the allocator on -- spacer
blocks, a new candidate, probe again -- until a candidate in another rank turns up (or a budget is spent; then the
last pair is used as it is).  Everything it held on the way sits in a private memory pool and goes back to the driver
with it; the process's own cached memory is never touched.  The walk is bounded (32 GiB by default, a quarter of the
free memory at most), skipped when the device is shared, and ``PIKAZOO_PLACE_TRAJECTORIES=0`` switches the whole
thing off.  No result depends on any of this.
"""
from __future__ import annotations

import os
import time

import torch

from . import _native

MIN_BYTES = 256 << 20          # smaller tensors are left alone: the probe of a smaller pair would sit in the 256 MiB
                               # Infinity Cache, and the launches that write them are not bound by the write rate (int16
                               # rows at k = 32: 2.88 us per frame in one rank and in two)
SPACER_BYTES = 4 << 30         # the physical runs of one rank seen on MI355X boxes are 2 .. 63 GiB long
CANDIDATE_BLOCK = 1 << 30      # candidates are carved from blocks of at least this size: a small allocation is put into
                               # whatever hole fits it (usually next to the first tensor), a large one into fresh memory
MAX_SPACER_BYTES = 96 << 30    # one rank is 96 GB: further than that the allocator has left the first candidate's rank
                               # (what a caller may ask for with max_spacer_bytes=)
DEFAULT_SPACER_BUDGET = 32 << 30   # what alloc_pair walks by itself (the driver's bench run needed 8 GiB / 3 candidates)
OTHER_TENANT_BYTES = 4 << 30   # device memory held by somebody else (another process / rank): above this, no walk
DISTINCT_BELOW = 0.86          # t(a, b) / (t(a) + t(b)): ~1.0 in one rank, 0.77-0.83 in two (in between: a tensor that
                               # straddles two ranks)
KEEP_FAR_BYTES = 8 << 30       # blocks a walk found in another rank are kept (reserved, invisible to
                               # torch.cuda.empty_cache()) for the next pair of their size -- up to this much in all
                               # (``PIKAZOO_PLACE_KEEP_GIB``); beyond it the oldest kept block that is not in use goes
                               # back to the driver first

_verdicts: dict = {}           # (device index, low ptr, high ptr, bytes) -> ratio measured for that pair
_far_pools: dict = {}          # (device index, bytes) -> private pools holding one block each that a walk found in another rank
_gave_up: dict = {}            # (device index, bytes) -> time of a walk that spent its budget: pairs of that size are not
                               # walked again for RETRY_AFTER_S (a loop that allocates per iteration must not stall per iteration)
RETRY_AFTER_S = 30.0
last_info: dict = {}           # what the latest alloc_pair did (diagnostic; bench.py prints it)


WARM_LAUNCHES = 24             # ~2.5 ms of the probe before anything is timed: after an idle spell (a host-side pause of
                               # a few seconds is enough) the first launches run at low clocks, and a slow t(a) in front
                               # of a normal t(a, b) reads as "two ranks"


def _time_probe(lib, a_ptr, b_ptr, nbytes, reps=3):
    """Best of `reps` launches of pz_probe_write on the current stream, in microseconds."""
    stream = torch.cuda.current_stream()
    raw = stream.cuda_stream
    _native.check(lib.pz_probe_write(a_ptr, b_ptr, nbytes, raw), "pz_probe_write")  # warm-up
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        _native.check(lib.pz_probe_write(a_ptr, b_ptr, nbytes, raw), "pz_probe_write")
        e1.record(stream)
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def _span_bytes(t: torch.Tensor) -> int:
    """Bytes from the tensor's first element to the end of its storage (a view of a larger block: the block's)."""
    return t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()


def pair_ratio(a: torch.Tensor, b: torch.Tensor) -> float:
    """t(a, b) / (t(a) + t(b)) of the k-frame store pattern (OVERWRITES both tensors, and what lies behind them in
    their storage blocks): ~1.0 when the two allocations share a rank of the device memory, ~0.8 when they do not.
    The three cases are timed in turn, twice, behind a warm-up that brings the clocks up; the best of each counts."""
    lib = _native.load()
    nbytes = min(_span_bytes(a), _span_bytes(b))
    with torch.cuda.device(a.device):
        raw = torch.cuda.current_stream().cuda_stream
        for _ in range(WARM_LAUNCHES):
            _native.check(lib.pz_probe_write(a.data_ptr(), b.data_ptr(), nbytes, raw), "pz_probe_write")
        t_a = t_b = t_ab = float("inf")
        for _ in range(2):
            t_ab = min(t_ab, _time_probe(lib, a.data_ptr(), b.data_ptr(), nbytes))
            t_a = min(t_a, _time_probe(lib, a.data_ptr(), None, nbytes))
            t_b = min(t_b, _time_probe(lib, None, b.data_ptr(), nbytes))
    return t_ab / (t_a + t_b)


def pair_write_rate(a: torch.Tensor, b: torch.Tensor, seconds: float = 0.02) -> float:
    """GB/s the k-frame store pattern SUSTAINS into both tensors at once (OVERWRITES them): back-to-back probe launches
    for at least `seconds`, timed as one region with events on the launch stream -- ~7 000 in two ranks, ~5 500 in one.
    (A single best-of-3 launch of ~80 us, as in round 3, reads a few per cent low: a kernel measured against it over
    0.25 s then seems to beat its own ceiling.)"""
    lib = _native.load()
    nbytes = min(_span_bytes(a), _span_bytes(b))
    frame = int(lib.pz_probe_frame_bytes())
    if nbytes < frame:
        raise ValueError(f"pair_write_rate needs tensors of at least one probe frame ({frame} bytes)")
    with torch.cuda.device(a.device):
        stream = torch.cuda.current_stream()
        raw = stream.cuda_stream
        for _ in range(WARM_LAUNCHES):
            _native.check(lib.pz_probe_write(a.data_ptr(), b.data_ptr(), nbytes, raw), "pz_probe_write")
        one = _time_probe(lib, a.data_ptr(), b.data_ptr(), nbytes, reps=1)
        launches = max(8, int(seconds * 1e6 / max(one, 1.0)) + 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(launches):
            _native.check(lib.pz_probe_write(a.data_ptr(), b.data_ptr(), nbytes, raw), "pz_probe_write")
        e1.record(stream)
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / launches
    return 2 * (nbytes // frame) * frame / (us * 1e-6) / 1e9


def _key(a, b):
    lo, hi = sorted((a.data_ptr(), b.data_ptr()))
    return (a.device.index, lo, hi, a.numel() * a.element_size())


def enabled() -> bool:
    """``PIKAZOO_PLACE_TRAJECTORIES=0`` in the environment switches the placement off for the whole process."""
    return os.environ.get("PIKAZOO_PLACE_TRAJECTORIES", "1").strip().lower() not in ("0", "false", "off", "no")


def reset():
    """Forget every cached verdict and give the kept blocks back to the driver (those no tensor uses any more: a pool
    dies with its last block)."""
    _verdicts.clear()
    _gave_up.clear()
    _far_pools.clear()


def _keep_limit() -> int:
    try:
        return int(max(0.0, float(os.environ["PIKAZOO_PLACE_KEEP_GIB"])) * (1 << 30))
    except (KeyError, ValueError):
        return KEEP_FAR_BYTES


def retained_bytes() -> int:
    """Device memory the kept far blocks reserve (in use by a live tensor or waiting for the next pair of their size)."""
    return sum(size for pools in _far_pools.values() for _pool, size in pools)


def _trim_far_pools(extra: int = 0):
    """Keep the kept blocks within ``KEEP_FAR_BYTES`` (minus `extra` bytes about to be added): the oldest blocks that no
    tensor uses are dropped first -- their pools die, the memory goes back to the driver.  A block that is in use cannot
    be dropped, so the total may stay above the limit until its tensor has died."""
    limit = _keep_limit()
    entries = [(k, p, size) for k, pools in _far_pools.items() for p, size in pools]  # oldest first
    total = sum(size for _k, _p, size in entries)
    if total + extra <= limit:
        return
    _far_pools.clear()
    for k, p, size in entries:
        if total + extra > limit and _has_free_block(p, size):
            total -= size  # dropped: the last reference to its pool goes with `entries`
        else:
            _far_pools.setdefault(k, []).append((p, size))
    del entries


def _keep_far_pool(key, pool, block_bytes: int):
    """Remember the pool of a block a walk found in another rank, within the limit (`_trim_far_pools`)."""
    if block_bytes > _keep_limit():
        return  # (never kept: the pool dies with the tensor that uses its block)
    _trim_far_pools(extra=block_bytes)
    _far_pools.setdefault(key, []).append((pool, block_bytes))


def _has_free_block(pool, nbytes: int) -> bool:
    """Does the private pool hold an unused block of at least `nbytes`?"""
    try:
        return any(blk.get("state") == "inactive" and blk.get("size", 0) >= nbytes
                   for seg in pool.snapshot() for blk in seg.get("blocks", ()))
    except Exception:  # noqa: BLE001 - an allocator without pool snapshots: no reuse, the walk still works
        return False


def _default_budget() -> int:
    """``DEFAULT_SPACER_BUDGET``, or ``PIKAZOO_PLACE_MAX_GIB`` from the environment (up to one rank, 96 GiB: the runs of
    one rank seen on MI355X boxes are 2 - 63 GiB long, so 32 GiB finds the other rank in most allocations, not in all)."""
    try:
        gib = float(os.environ["PIKAZOO_PLACE_MAX_GIB"])
    except (KeyError, ValueError):
        return DEFAULT_SPACER_BUDGET
    return int(min(max(gib, 0.0), MAX_SPACER_BYTES / (1 << 30)) * (1 << 30))


def _why_not_walk(device, budget: int):
    """A reason not to walk the allocator (the pair is then used as allocated), or None.  The walk holds gigabytes for
    a few milliseconds: it is only done where this process has the device to itself and the memory is plainly there."""
    if budget < SPACER_BYTES:
        return "no budget (max_spacer_bytes)"
    world = os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE")
    if world and world.isdigit() and int(world) > max(1, torch.cuda.device_count()):
        return f"{world} ranks on {torch.cuda.device_count()} visible GPU(s): the device is shared"
    free, total = torch.cuda.mem_get_info(device)
    others = (total - free) - torch.cuda.memory_reserved(device)
    if others > OTHER_TENANT_BYTES:
        return f"{others / (1 << 30):.1f} GiB of the device are held by somebody else"
    if not hasattr(torch.cuda, "MemPool") or not hasattr(torch.cuda, "use_mem_pool"):
        return "this torch has no private memory pools (torch.cuda.MemPool)"
    return None


def alloc_pair(shape, dtype, device, *, max_spacer_bytes: int = None):
    """`_alloc_pair_probed` behind a safety net.  Placement moves time, never results, and it leans on things that differ
    between torch versions and allocators (private memory pools, the caching allocator's block reuse, pool snapshots): if
    anything in it raises, the caller gets two plainly allocated tensors and ``last_info["error"]`` says why -- the k-frame
    launches then run at the one-rank rate (a quarter slower), they do not fail."""
    global last_info
    try:
        return _alloc_pair_probed(shape, dtype, device, max_spacer_bytes=max_spacer_bytes)
    except torch.cuda.OutOfMemoryError:
        raise  # (the tensors themselves do not fit: not placement's to hide)
    except Exception as exc:  # noqa: BLE001
        last_info = {"probed": False, "error": f"{type(exc).__name__}: {exc}"[:200]}
        device = torch.device(device)
        return torch.empty(shape, dtype=dtype, device=device), torch.empty(shape, dtype=dtype, device=device)


def _alloc_pair_probed(shape, dtype, device, *, max_spacer_bytes: int = None):
    """Two uninitialised tensors of `shape` that do not share a rank of the device memory, when that can be arranged.

    Returns (a, b).  `last_info` says what happened: ``probed`` (False for small tensors, while a hipGraph is being
    captured, or with ``PIKAZOO_PLACE_TRAJECTORIES=0``), ``ratio`` of the pair returned, ``distinct``, ``candidates``
    tried, ``spacer_gib`` walked, ``walk`` (why the allocator was not walked, if it was not).

    What the walk may hold, for the few milliseconds it lasts: at most `max_spacer_bytes` (default
    ``DEFAULT_SPACER_BUDGET`` = 32 GiB or ``PIKAZOO_PLACE_MAX_GIB``, and never more than a quarter of the free memory) -- in a PRIVATE memory pool
    that is released to the driver afterwards, so nothing of the process's own cached memory is touched.  It is not
    walked at all when somebody else holds memory on the device or ranks share it (``_why_not_walk``)."""
    global last_info
    device = torch.device(device)
    probe = torch.empty((), dtype=dtype).element_size()
    numel = 1
    for d in shape:
        numel *= int(d)
    nbytes = numel * probe
    info = {"probed": False, "bytes": nbytes}
    last_info = info
    if nbytes < MIN_BYTES or not enabled() or torch.cuda.is_current_stream_capturing():
        return torch.empty(shape, dtype=dtype, device=device), torch.empty(shape, dtype=dtype, device=device)

    def fresh(elems):  # (the front of a block of `elems` elements; the view keeps the block alive)
        block = torch.empty(elems, dtype=dtype, device=device)
        return block[:numel].view(shape)

    _trim_far_pools()  # (the limit may have been lowered, and blocks kept for other sizes may have become free)
    a = fresh(numel)
    # a block an earlier walk found for this size lives in a private pool of its own (`_far_pools`): when the tensor that
    # used it has died, the block is free in that pool and is handed out again here -- no second walk
    b = None
    for pool, _size in _far_pools.get((device.index, nbytes), ()):
        if not _has_free_block(pool, max(numel, CANDIDATE_BLOCK // probe) * probe):
            continue  # (its block is in use: allocating in this pool now would only grow it)
        with torch.cuda.use_mem_pool(pool, device=device):
            cand = fresh(max(numel, CANDIDATE_BLOCK // probe))
        if _verdicts.get(_key(a, cand), 1.0) < DISTINCT_BELOW or pair_ratio(a, cand) < DISTINCT_BELOW:
            b = cand
            break
        del cand
    if b is None:
        b = fresh(numel)
    known = _verdicts.get(_key(a, b))
    if known is not None:  # (a pair the allocator hands out again: its blocks came back from torch's cache)
        info.update(probed=True, cached=True, ratio=known, distinct=known < DISTINCT_BELOW, candidates=0, spacer_gib=0.0)
        return a, b
    ratio = pair_ratio(a, b)
    _verdicts[_key(a, b)] = ratio
    info.update(probed=True, ratio=ratio, distinct=ratio < DISTINCT_BELOW, candidates=1, spacer_gib=0.0)
    if ratio < DISTINCT_BELOW:
        return a, b
    if time.monotonic() - _gave_up.get((device.index, nbytes), -1e9) < RETRY_AFTER_S:
        info["walk"] = "skipped: a walk for this size spent its budget a moment ago"
        return a, b
    free, _total = torch.cuda.mem_get_info(device)
    budget = int(max_spacer_bytes) if max_spacer_bytes is not None else _default_budget()
    budget = min(budget, free // 4)
    why = _why_not_walk(device, budget)
    if why is not None:
        info["walk"] = why
        return a, b
    # walk the allocator on: everything tried stays allocated meanwhile, so that the next candidate is other memory --
    # all of it in a private pool, which goes back to the driver as a whole
    block_elems = max(numel, CANDIDATE_BLOCK // probe)
    held, held_pools, walked, tried, found = [], [], 0, 1, None
    pool = torch.cuda.MemPool()  # the spacers
    try:
        while walked + SPACER_BYTES <= budget:
            free, _total = torch.cuda.mem_get_info(device)
            if free < 2 * SPACER_BYTES + 2 * block_elems * probe:
                info["walk"] = "stopped: free memory below two spacers"
                break
            with torch.cuda.use_mem_pool(pool, device=device):
                held.append(torch.empty(SPACER_BYTES, dtype=torch.uint8, device=device))
            cand_pool = torch.cuda.MemPool()  # every candidate in a pool of its own: the one that is kept keeps its pool
            with torch.cuda.use_mem_pool(cand_pool, device=device):
                cand = fresh(block_elems)  # (a view: keeps its block alive)
            walked += SPACER_BYTES
            tried += 1
            r = pair_ratio(a, cand)
            if r < DISTINCT_BELOW:
                found, ratio = cand, r
                _keep_far_pool((device.index, nbytes), cand_pool, block_elems * probe)
                break
            held.append(cand)
            held_pools.append(cand_pool)
            del cand, cand_pool
    except torch.cuda.OutOfMemoryError:
        info["walk"] = "stopped: out of memory"
    if found is not None:
        b = found
        _verdicts[_key(a, b)] = ratio
    else:
        _gave_up[(device.index, nbytes)] = time.monotonic()
    # tensors first, pools second: a pool releases the blocks that are unused when it dies (spacers, rejected candidates:
    # back to the driver here); the block of the candidate that is kept stays in its own pool (`_far_pools`) for the
    # next pair of this size
    held.clear()
    held_pools.clear()
    del held, held_pools, pool
    info.update(ratio=ratio, distinct=ratio < DISTINCT_BELOW, candidates=tried, spacer_gib=walked / (1 << 30),
                retained_gib=retained_bytes() / (1 << 30))
    return a, b

"""Data-parallel pieces of the train step (SURVEY §8e / DESIGN.md §7).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI; "gloo" on CPU for tests).
The ONLY exchange of a step is one all-reduce(sum) over the flat fp32 mapper-gradient bucket; the
mean is folded into the optimizer (`grad_div = world_size`).  The reference would instead wrap the text
encoder in DDP, which misses the dict-held object mappers and all-reduces the 152 MB embedding table
(training/coach.py:97-99, models/net_clip_text_embedding.py:25-32) — deliberately not reproduced.
"""
from __future__ import annotations

import os

import torch


def world_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


COLLECTIVE_CALLS = 0  # collectives issued by this process on the data path (tests assert ONE per optimisation step)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None


def all_agree(ok: bool) -> bool:
    """True on every rank iff `ok` is True on EVERY rank (one tiny MIN all-reduce over the process group; the identity
    without one).  Every decision that changes which collectives a rank issues — the library communicator or torch's, the
    exchange inside the step's graph or between two graphs — goes through this, so that ranks never disagree on the path
    (a rank that falls back alone mis-pairs its collectives with the others' and the job deadlocks)."""
    dist = _dist()
    if dist is None:
        return bool(ok)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


class RcclComm:
    """An RCCL communicator behind the C ABI (include/vneti.h: vneti_comm_unique_id / vneti_comm_init /
    vneti_allreduce_flat / vneti_comm_destroy; csrc/comm.hip): the step's one exchange as a stream-ordered library call —
    what a binder of the C ABI that does not run torch.distributed uses, and capturable in the step's hipGraph.
    `exchange(obj)` must return rank 0's `obj` on every rank (any side channel: here torch.distributed's object broadcast
    over the already initialised process group; a 128-byte id is all that crosses it).
    Construction is collective and its outcome is AGREED: if the id cannot be made on rank 0 or the initialisation fails on
    any rank, every rank raises RuntimeError (and the ranks that did succeed release their handle)."""

    def __init__(self, rank: int, world: int, exchange=None, agree=None):
        import ctypes as C
        from . import lib
        exchange = exchange or share_from_rank0
        agree = agree or (all_agree if exchange is share_from_rank0 else (lambda ok: ok))
        self._h = None
        self.rank, self.world = rank, world
        msg = None
        if rank == 0:  # a failure here must reach the other ranks, who are about to wait for the id
            try:
                buf = C.create_string_buffer(128)
                lib.call("comm_unique_id", buf)
                msg = bytes(buf.raw)
            except RuntimeError as err:
                msg = err
        msg = exchange(msg)
        if isinstance(msg, Exception):
            raise RuntimeError(f"rank 0 could not create the RCCL id: {msg}")
        h = C.c_void_p()
        err = None
        try:
            lib.call("comm_init", msg, rank, world, C.byref(h))
        except RuntimeError as e:
            err = e
        if not agree(err is None):
            if err is None:
                lib.call("comm_destroy", h)
            raise RuntimeError(f"RCCL communicator initialisation failed on {'this' if err else 'another'} rank"
                               + (f": {err}" if err else ""))
        self._h = h

    @property
    def closed(self) -> bool:
        return self._h is None

    def all_reduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        from . import lib
        if self._h is None:
            raise RuntimeError("this RCCL communicator was closed (process group re-initialised or interpreter exit): an "
                               "engine holding it — and any graph captured over it — must be rebuilt")
        assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.is_cuda
        lib.call("allreduce_flat", self._h, flat.data_ptr(), flat.numel(), torch.cuda.current_stream().cuda_stream)
        return flat

    def close(self):
        from . import lib
        if self._h:
            h, self._h = self._h, None
            lib.call("comm_destroy", h)


_direct: "RcclComm" = None  # set by enable_direct_rccl(): the exchange goes through vneti_allreduce_flat instead of torch


def enable_direct_rccl():
    """route the step's all-reduce through the library's own RCCL communicator (the default on the nccl backend when a
    TrainStepEngine with world_size > 1 is built; VNETI_RCCL_DIRECT=0 keeps torch's).  Needs an initialised process group
    for the id.  The communicator belongs to ONE (rank, world): a process group re-initialised with another shape gets a
    new one, and it is destroyed at interpreter exit (before torch tears the device context down)."""
    global _direct
    dist = _dist()
    if _direct is not None and (dist is None or (_direct.rank, _direct.world) != (dist.get_rank(), dist.get_world_size())):
        disable_direct_rccl()
    if _direct is None and dist is not None:
        _direct = RcclComm(dist.get_rank(), dist.get_world_size())
        global _atexit_set
        if not _atexit_set:
            import atexit
            atexit.register(disable_direct_rccl)
            _atexit_set = True
    return _direct


_atexit_set = False


def disable_direct_rccl():
    """destroy the library communicator (no-op when none exists); the exchange falls back to torch.distributed"""
    global _direct
    if _direct is not None:
        try:
            _direct.close()
        finally:
            _direct = None


def all_reduce_sum_(flat: torch.Tensor, comm=None) -> torch.Tensor:
    """in-place sum over ranks of the flat gradient bucket (no-op without a process group).  `comm`: an explicit
    communicator object (RcclComm) — the call is then stream-ordered and capturable."""
    global COLLECTIVE_CALLS
    if comm is not None:
        comm.all_reduce_sum_(flat)
        if not torch.cuda.is_current_stream_capturing():  # a captured node is counted per replay (TrainStepEngine.step)
            COLLECTIVE_CALLS += 1
        return flat
    dist = _dist()
    if dist is not None:
        if _direct is not None and (_direct.rank, _direct.world) == (dist.get_rank(), dist.get_world_size()):
            _direct.all_reduce_sum_(flat)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        COLLECTIVE_CALLS += 1
    return flat


def reduce_plan(n_obj: int, n_objects: int, active: int, total: int):
    """Which slices of the flat mapper-gradient bucket [object mapper 0 .. K-1 | view mapper] one optimisation step has to
    exchange: everything for a single object mapper; with several (learnable_mode 3, one mapper per scene,
    training/coach.py:505-552) only the scene every rank trained this step and the shared view mapper — 1.13 MB instead
    of the 50.4 MB bucket at BASELINE config 4 (88 scenes, SD-2.1 widths)."""
    if n_objects == 1:
        return [(0, total)]
    plan = [(active * n_obj, (active + 1) * n_obj)]
    if total > n_objects * n_obj:
        plan.append((n_objects * n_obj, total))
    return plan


def all_reduce_plan_(flat: torch.Tensor, plan, stage: torch.Tensor = None, comm=None) -> int:
    """sum the planned slices over ranks in place with ONE collective (north_star: one exchange step per optimisation
    step); returns the payload bytes handed to it.  A single slice is reduced where it lies; several (learnable_mode 3: the
    active scene's segment + the view mapper, which are not adjacent in the bucket) are packed into the contiguous `stage`
    buffer (allocated on first use when not given), reduced there and copied back — two small device copies each way
    instead of a second collective's launch + ring latency."""
    moved = sum(b - a for a, b in plan) * flat.element_size()
    if len(plan) == 1:
        a, b = plan[0]
        all_reduce_sum_(flat[a:b], comm)
        return moved
    n = sum(b - a for a, b in plan)
    if stage is None or stage.numel() < n:
        stage = torch.empty(n, dtype=flat.dtype, device=flat.device)
    off = 0
    for a, b in plan:
        stage[off:off + b - a].copy_(flat[a:b])
        off += b - a
    all_reduce_sum_(stage[:n], comm)
    off = 0
    for a, b in plan:
        flat[a:b].copy_(stage[off:off + b - a])
        off += b - a
    return moved


_share_picks = 0  # depth of shared_picks() contexts


class shared_picks:
    """`with shared_picks():` — engines built inside share rank 0's autotuner picks through a broadcast.  Only code that
    EVERY rank runs in the same order may open it (TrainStepEngine.__init__); an engine a single rank builds on its own
    (rank 0's ValidationHandler -> InferenceEngine, training/coach.py:119-125 analogue) must stay outside, or its broadcasts
    would pair with the other ranks' gradient all-reduce."""

    def __enter__(self):
        global _share_picks
        _share_picks += 1
        return self

    def __exit__(self, *exc):
        global _share_picks
        _share_picks -= 1
        return False


def sharing_picks() -> bool:
    return _share_picks > 0


def share_from_rank0(obj):
    """rank 0's `obj` (any picklable value) on every rank; the identity without a process group.  Control plane only (the
    autotuner's tile picks, once per engine build): every rank must replay ONE schedule, or the weak-scaling value is the
    slowest rank's private pick and results differ across world sizes."""
    dist = _dist()
    if dist is None:
        return obj
    box = [obj if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def data_seed(base_seed: int, rank: int) -> int:
    """rank r draws its own images / noise / timesteps; mapper initialisation uses the SAME seed on
    every rank (the reference re-seeds with torch.manual_seed(0) inside every mapper constructor)."""
    return base_seed + rank


def scaled_lr(lr: float, grad_accum: int, batch_size: int, world: int, scale_lr: bool = True) -> float:
    """training/coach.py:728-733: lr * gradient_accumulation_steps * train_batch_size * num_processes"""
    return lr * grad_accum * batch_size * world if scale_lr else lr

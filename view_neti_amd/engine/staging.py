"""Per-step host -> device uploads that do not stall the host.

`Coach.train` hands the engine a new batch every step (training/coach.py:154-163: the dataloader's tensors).  A `copy_` from
ordinary (pageable) host memory into a device buffer blocks the host until the stream has reached it — i.e. until the PREVIOUS
step's graph has finished — so the host could never enqueue step i + 1 while step i runs, and the GPU idled for the upload and
launch latency of every step (bench_coach with the device input pipeline, same box: 38.17 steps/s with blocking uploads, 38.44 -
38.52 with this stager, 39.22 for the resident-batch replay).  Here every upload goes
through a pinned staging slot: the host writes the slot, enqueues an asynchronous copy and moves on; a slot is reused only after
the event recorded behind its copies has passed (`depth` steps of run-ahead, then back-pressure)."""
from __future__ import annotations

from typing import Dict, List, Optional

import os

import torch

MAX_STAGED_BYTES = 1 << 20
_BLOCKING = os.environ.get("VNETI_NO_STAGER", "0") == "1"  # lab switch: plain blocking copy_ (what rounds 1-5 did)


class HostStager:
    def __init__(self, depth: int = 4):
        self.depth = depth
        self.slots: List[Dict] = [dict(ev=None, bufs={}) for _ in range(depth)]
        self.k = 0
        self.cur: Optional[Dict] = None

    def begin(self):
        slot = self.slots[self.k % self.depth]
        self.k += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()  # the copies that last read this slot are done (normally long ago)
        self.cur = slot

    def upload(self, name: str, dev: torch.Tensor, host) -> None:
        """dev.copy_(host) without blocking the host; `host` is any CPU tensor (or something torch.as_tensor takes) of dev's
        shape; converted to dev's dtype on the host side"""
        assert self.cur is not None, "upload() outside begin() / end()"
        host = torch.as_tensor(host)
        # already on the device: an ordinary stream-ordered copy (also the A/B switch).  LARGE payloads (the host input
        # pipeline's 12.6 MB pixel batch) stay on the blocking path as well: staged through a pinned slot they were slower, not
        # faster (Coach with the host pipeline 14.8 - 16.9 vs 26.0 steps/s, profiles/r06_coach_stager_ab.txt) — that variant is
        # bound by the dataloader workers' CPU time, and the extra 12.6 MB host copy competes with them
        if host.is_cuda or _BLOCKING or dev.numel() * dev.element_size() > MAX_STAGED_BYTES:
            dev.copy_(host.reshape(dev.shape))
            return
        buf = self.cur["bufs"].get(name)
        if buf is None or buf.shape != dev.shape or buf.dtype != dev.dtype:
            buf = self.cur["bufs"][name] = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
        buf.copy_(host.reshape(dev.shape))
        dev.copy_(buf, non_blocking=True)

    def end(self):
        ev = torch.cuda.Event()
        ev.record()
        self.cur["ev"] = ev
        self.cur = None

"""CU-partitioned HIP streams (include/vneti.h: vneti_stream_create_cu_mask) — a MEASUREMENT aid, not on the step's path.

Round 6 asked whether the NEXT batch's VAE encode — which depends on no trainable state (training/coach.py:165-169: frozen
VAE, `.detach()`) — hides beside the current step when it runs on a stream whose kernels may only occupy a fixed subset of
the 256 compute units, so that a chip-filling encoder convolution can never take every CU away from the 10-20 us launches of
the UNet's small levels (the failure of every earlier overlap experiment, profiles/LAB_NOTES.md round 2).  Measured
(tools/lab/cu_mask_probe.py, profiles/r06_cu_mask_probe.txt): the mask binds — a LINEAR hipGraph launched on a masked stream
runs on that stream's queue, the VAE graph takes 15.5 / 10.9 / 8.2 ms on 64 / 96 / 128 CUs against 5.25 on the chip — and the
pair [rest of the step || masked VAE] is never shorter than the sequential step (24.83 - 25.99 ms against 24.94): the work
is conserved, the chip is power- and fill-bound as a whole, there is no idle capacity to harvest.  The step stays sequential.

Mask convention (measured on gfx950): bit i of the mask is logical CU i and the driver deals logical CUs round-robin over the
8 XCDs — bits 0..7 are CU 0 of XCD 0..7, bits 8..15 CU 1 of every XCD, and so on.  `per_xcd_mask(k)` therefore keeps the XCDs
balanced (k CUs on each), which the GEMM kernels' `block -> XCD = block % 8` tile order relies on.  `whole_xcd_mask(n)` (bits
i with i % 8 < n) does NOT confine work to n XCDs: a mask that leaves an XCD without any CU is not applied at all (the VAE
graph ran at full speed under every such mask) — in SPX mode the dispatcher deals workgroups to all XCDs, an XCD cannot be
masked out.  Kept because the probe documents exactly that.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import lib

N_CU = 256
N_XCD = 8
CU_PER_XCD = N_CU // N_XCD


def per_xcd_mask(k: int, first: int = 0) -> List[int]:
    """k compute units on EVERY XCD (logical CUs first .. first+k-1 of each): bits [8*first, 8*(first+k))"""
    if not 0 < k <= CU_PER_XCD - first:
        raise ValueError(f"per_xcd_mask({k}, first={first}): an XCD has {CU_PER_XCD} CUs")
    return _words(range(N_XCD * first, N_XCD * (first + k)))


def whole_xcd_mask(n: int, first: int = 0) -> List[int]:
    """bits i with first <= i % 8 < first + n, i.e. every CU of n XCDs and none of the others — measured NOT to bind (see the
    module docstring): the runtime ignores a mask that empties an XCD"""
    if not 0 < n <= N_XCD - first:
        raise ValueError(f"whole_xcd_mask({n}, first={first}): the chip has {N_XCD} XCDs")
    return _words(i for i in range(N_CU) if first <= i % N_XCD < first + n)


def _words(bits) -> List[int]:
    w = [0] * (N_CU // 32)
    for b in bits:
        w[b // 32] |= 1 << (b % 32)
    return w


class CUMaskStream:
    """a HIP stream restricted to the CUs of `mask` (list of 32-bit words), usable as a torch stream: `.stream` is a
    torch.cuda.ExternalStream over the handle.  The handle is released with the object."""

    def __init__(self, mask: List[int]):
        arr = (C.c_uint * len(mask))(*mask)
        h = C.c_void_p()
        lib.call("stream_create_cu_mask", arr, len(mask), C.byref(h))
        self._h = h
        self.mask = list(mask)
        self.n_cus = sum(bin(w).count("1") for w in mask)
        self.stream = torch.cuda.ExternalStream(h.value)

    def runtime_mask(self) -> List[int]:
        """the mask the HIP runtime reports for the stream (hipExtStreamGetCUMask)"""
        arr = (C.c_uint * len(self.mask))()
        lib.call("stream_get_cu_mask", self._h, arr, len(self.mask))
        return list(arr)

    def close(self):
        if self._h is not None and self._h.value:
            torch.cuda.synchronize()
            lib.call("stream_destroy", self._h)
        self._h = None
        self.stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 (interpreter shutdown)
            pass

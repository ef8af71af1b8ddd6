"""Shared machinery of the static HIP launch schedules (UNet, VAE encoder, text encoder).

A schedule is built once for fixed shapes: `fwd` / `bwd` are plain lists of bound kernel launches
(functools.partial over view_neti_amd.ops), replayed in order on the current stream — which makes
the whole train step capturable in one hipGraph.  Gradient accumulation where two paths meet is
resolved while the backward list is built (in execution order): the first contribution writes,
later ones accumulate in place through the kernels' fused residual/accumulate operands.
"""
from __future__ import annotations

from functools import partial
from typing import Dict, List

import torch

from .. import lib, ops, packing


class T:
    """Activation handle: forward view `v`, gradient view `g`, and whether `g` already holds a
    contribution (tracked in backward *execution* order while the schedule is built)."""

    __slots__ = ("v", "g", "gw", "rows", "cols", "children", "need_grad", "producer")

    def __init__(self, v, g=None, need_grad=True):
        self.v = v
        self.g = g
        self.gw = False
        self.rows, self.cols = v.shape
        self.children: List["T"] = []
        self.need_grad = need_grad
        self.producer = None  # index in Schedule.fwd of the single GEMM that writes all of `v` (see _produced)


def rup(x, m):
    return (x + m - 1) // m * m


class Schedule:
    def __init__(self, batch: int, groups: int, eps: float, device: str = "cuda", need_backward: bool = True):
        self.B = batch
        self.groups = groups
        self.eps = eps
        self.dev = device
        self.need_backward = need_backward
        self.fwd_pre: List = []  # launches that do not depend on the schedule's main input (may run early)
        self.fwd: List = []
        self.bwd: List = []
        self.tape: List = []
        self.bytes = 0
        self._scratch: Dict[str, torch.Tensor] = {}
        self.temb_off: Dict[str, tuple] = {}
        self.temb_all = None
        # vneti_groupnorm_ws_floats upper bound: <=256 slabs x 2G partials + 2*B*G finals
        self.gn_ws = self._buf((batch * 256 * 2 * groups + 2 * batch * groups,), torch.float32)
        # f32 scratch for split-K GEMM / q-split attention partials: one per schedule, so two schedules
        # may run concurrently on different streams (bind_workspace() pins it into every launch)
        self.ws = self._buf((16 * 2 ** 20,), torch.float32)
        self.ws_side = None
        self._side_stream = None
        self._gn_fusable: List[dict] = []
        self.gn_sums = None
        if ops._default_ws is None or ops._default_ws.device != torch.device(device, torch.cuda.current_device()):
            ops.set_default_gemm_workspace(torch.empty(16 * 2 ** 20, dtype=torch.float32, device=device))

    # ------------------------------------------------------------------ memory helpers
    def _buf(self, shape, dtype=None, zero=False):
        dtype = dtype or lib.act_dtype()  # the library's 16-bit format (fp16, or bf16 in the -DVN_BF16 build)
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.dev)
        self.bytes += t.numel() * t.element_size()
        return t

    def _tmp(self, name, rows, cols, dtype=None):
        dtype = dtype or lib.act_dtype()
        """reusable scratch for backward temporaries (single stream => sequential lifetimes)."""
        key = f"{name}:{dtype}"
        n = rows * cols
        cur = self._scratch.get(key)
        if cur is None or cur.numel() < n:
            self._scratch[key] = self._buf((n,), dtype)
            cur = self._scratch[key]
        return cur[:n].view(rows, cols)

    def _w16(self, t):
        w = t.to(device=self.dev, dtype=lib.act_dtype()).contiguous()
        self.bytes += w.numel() * 2
        return w

    def _w32(self, t):
        w = t.to(device=self.dev, dtype=torch.float32).contiguous()
        self.bytes += w.numel() * 4
        return w

    # ------------------------------------------------------------------ gradient bookkeeping
    def _grad(self, t: T):
        if t.g is None:
            t.g = self._buf((t.rows, t.cols))
        return t.g

    def _contrib(self, t: T, fn, extra=None):
        """fn(out, accum) must launch a kernel computing out = result (+ accum)."""
        g = self._grad(t)
        if t.gw:
            if extra is not None:
                self.bwd.append(partial(ops.add, g, extra, g))
            self.bwd.append(self._tagged(partial(fn, g, g), fn))
        else:
            self.bwd.append(self._tagged(partial(fn, g, extra), fn))
            t.gw = True
            for c in t.children:
                c.gw = True

    @staticmethod
    def _tagged(p, fn):
        if hasattr(fn, "vn_cost"):
            p.vn_cost = fn.vn_cost
        return p

    def _contrib_gemm(self, t: T, A, Bm, **kw):
        """gradient contribution computed by a GEMM (accumulates through the fused residual)."""
        g = self._grad(t)
        resid = g if t.gw else None
        self.bwd.append(partial(ops.gemm, A, Bm, g, resid=resid, **kw))
        if not t.gw:
            t.gw = True
            for c in t.children:
                c.gw = True

    # ------------------------------------------------------------------ GEMM tile autotuning
    _tile_cache: Dict[tuple, int] = {}
    _TILE_DIMS = {1: (128, 128), 2: (128, 64), 3: (64, 64), 4: (256, 128), 5: (256, 256), 6: (256, 128), 7: (256, 128),
                  8: (256, 128), 9: (128, 128), 10: (128, 128), 11: (128, 64), 12: (64, 64), 13: (128, 128), 14: (128, 64),
                  15: (64, 64), 16: (256, 256), 17: (256, 128), 18: (256, 128)}

    @staticmethod
    def _gemm_key(f):
        kw = f.keywords
        A, Bm, out = f.args[:3]
        conv = kw.get("conv")
        M = kw.get("M") or A.shape[-2]
        N = kw.get("N") or Bm.shape[-2]
        K = kw.get("K") or Bm.shape[-1]
        ck = (conv["mode"], conv["stride"], conv["ups"], conv["Hi"], conv["Wi"]) if conv else None
        return (M, N, K, kw.get("batch") or 1, ck, out.dtype == torch.float32, kw.get("geglu") or 0)

    def autotune(self, candidates=(1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18), reps=8):
        """Measure, don't guess: time every distinct GEMM/conv problem of this schedule under each
        tile configuration, with the split-K heuristic and with split-K forced off (the f32 partials
        and the reduce launch are not always worth the extra blocks), and pin the fastest pair.  Launches are
        timed COLD (a 640 MB fill between them evicts L2 and the MALL, as the step's own traffic does): ranking
        them on cache-hot repeats picked configurations that were 1-2 % slower in the step.
        ~0.5 s per engine; results are cached per problem signature across engines."""
        if not torch.cuda.is_available():
            return
        import os
        cold = None if os.environ.get("VNETI_AUTOTUNE_HOT") else torch.empty(160 * 2 ** 20, dtype=torch.float32, device=self.dev)
        if os.environ.get("VNETI_AUTOTUNE_CANDS"):
            candidates = tuple(int(x) for x in os.environ["VNETI_AUTOTUNE_CANDS"].split(","))
        cold_reps = int(os.environ.get("VNETI_AUTOTUNE_REPS", "9"))
        warm_a = bool(int(os.environ.get("VNETI_AUTOTUNE_WARM_A", "1")))
        cache = Schedule._tile_cache
        # optional on-disk cache of the picks (profiling runs reuse a previous run's picks so that the rocprofv3
        # per-kernel averages are those of the step, not of the autotuner's probes)
        cache_path = os.environ.get("VNETI_AUTOTUNE_CACHE")
        if cache_path and os.path.exists(cache_path) and not cache:
            import ast
            import json
            for k, v in json.load(open(cache_path)).items():
                cache[ast.literal_eval(k)] = tuple(v)  # keys: repr() of tuples of ints / None / bools written below
        n_before = len(cache)
        # data parallel: rank 0 measures, every rank replays ITS picks (ranks that tuned on their own pinned different
        # tiles / split-K factors: the weak-scaling value was then the slowest rank's private schedule).  The other ranks
        # wait here for rank 0's cache and find every problem of the (identical) schedule in it.  This is a COLLECTIVE, so
        # it happens only inside `parallel.shared_picks()` — which TrainStepEngine.__init__, run by every rank, opens —
        # and never for an engine that one rank builds on its own (rank 0's validation / inference engines).
        from .. import parallel
        share = parallel.sharing_picks() and parallel._dist() is not None
        rank0 = (not share) or parallel._dist().get_rank() == 0
        if share and not rank0:
            cache.update(parallel.share_from_rank0(None))
        for lst in (self.fwd_pre, self.fwd, self.bwd):
            for idx, f in enumerate(lst):
                if getattr(f, "func", None) is not ops.gemm or f.keywords.get("tile_hint"):
                    continue
                key = self._gemm_key(f)
                conv = f.keywords.get("conv")
                # the 8-phase tiles (16 / 17) take either K order of an implicit conv at the same cost (their gather offsets
                # are linear in the tap); chunk-major — the nine taps of a 64-channel chunk in consecutive K-tiles — keeps the
                # re-reads of the input rows in L2 and wins on the wide layers (256 channels at 256^2: 326 -> 293 us cold)
                try_cm = bool(conv) and not conv.get("ups") and not conv.get("korder") and \
                    not (conv["mode"] == 2 and conv["stride"] == 2) and conv["Ci"] % 64 == 0 and conv["Ci"] >= 128
                B_cm = None
                if key not in cache:
                    best, best_t = (0, 0, 0), float("inf")
                    M_, N_, K_ = key[:3]
                    # tile 18 = the halo-patch form of 17 (a block owns 16 x 16 pixels, the input patch stays in LDS for all
                    # nine taps): stride-1 pad-1 3x3 convolutions (forward or transposed gather) on a 16-pixel grid, chunk-major K
                    # only; split-K in whole channel chunks
                    halo_ok = try_cm and conv["mode"] in (1, 2) and conv["stride"] == 1 and conv["pad_t"] == 1 and conv["pad_l"] == 1 \
                        and conv["Ho"] % 16 == 0 and conv["Wo"] % 16 == 0 and conv["Hi"] == conv["Ho"] and conv["Wi"] == conv["Wo"]
                    variants = [(h, 0) for h in candidates if h != 18] + \
                        ([(h, 1) for h in candidates if h in (16, 17) or (h == 18 and halo_ok)] if try_cm else [])
                    if try_cm:
                        B_cm = packing._chunk_major(f.args[1], f.args[1].shape[0], conv["Ci"])
                    for h, ko in variants:
                        bm, bn = self._TILE_DIMS[h % 100]
                        tiles = -(-M_ // bm) * -(-N_ // bn) * key[3]
                        # 0 = library heuristic, 1 = no split, explicit factors where the grid leaves CUs idle and K is deep
                        sks = (0, 1) + (tuple(x for x in (2, 3, 4, 6, 8, 12)
                                              if x * 8 <= K_ // 64 and tiles * x <= 1024 and x * key[3] * M_ * N_ <= 16 * 2 ** 20)
                                        if tiles < 256 else ())
                        if f.keywords.get("geglu"):
                            sks = (1,)  # the GEGLU epilogues do not exist in the split-K reduce kernel
                        for sk in sks:
                            kw = dict(f.keywords)
                            kw["tile_hint"], kw["split_k"] = h, sk
                            args = f.args
                            if ko:
                                kw["conv"] = dict(conv, korder=1)
                                args = (f.args[0], B_cm) + tuple(f.args[2:])
                            ops.gemm(*args, **kw)
                            ops.gemm(*args, **kw)
                            if cold is None:
                                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                                s.record()
                                for _ in range(reps):
                                    ops.gemm(*args, **kw)
                                e.record()
                                e.synchronize()
                                t = s.elapsed_time(e)
                            else:
                                # cold timing: inside the step a GEMM's weights (and everything older) were evicted by
                                # its predecessors — a fill of a buffer larger than L2 + MALL between the timed launches
                                # restores that — while its activation operand was written by the launch just before it:
                                # a no-op in-place add re-touches it after the fill (picks move by +0.4 % on the step)
                                ts = []
                                for _ in range(cold_reps):
                                    cold.fill_(0)
                                    if warm_a:  # the activation operand as its producer just left it (L2 / MALL), weights cold
                                        args[0].add_(0)  # (replaying the 2-4 preceding launches instead measured no better)
                                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                                    s.record()
                                    ops.gemm(*args, **kw)
                                    e.record()
                                    e.synchronize()
                                    ts.append(s.elapsed_time(e))
                                t = sorted(ts)[len(ts) // 2]  # median: one slow outlier must not veto a candidate
                            if t < best_t:
                                best, best_t = (h, sk, ko), t
                    cache[key] = best
                pick = tuple(cache[key]) + (0,) * (3 - len(cache[key]))
                kw = dict(f.keywords)
                kw["tile_hint"], kw["split_k"] = pick[0], pick[1]
                if pick[2] and try_cm:  # this launch runs chunk-major: its own re-ordered copy of the packed weight
                    if B_cm is None:
                        B_cm = packing._chunk_major(f.args[1], f.args[1].shape[0], conv["Ci"])
                    kw["conv"] = dict(conv, korder=1)
                    self.bytes += B_cm.numel() * B_cm.element_size()
                    f_cm = partial(ops.gemm, f.args[0], B_cm, *f.args[2:], **f.keywords)
                    if getattr(f, "side", False):
                        f_cm.side = True
                    f = f_cm
                lst[idx] = self._rebound(f, ops.gemm, kw)
        if share and rank0:
            parallel.share_from_rank0(dict(cache))
        if cache_path and len(cache) != n_before and parallel.world_info()[0] == 0:
            import json  # rank 0 only, and atomically: a concurrent or torn write would poison the next run's load
            tmp = f"{cache_path}.{os.getpid()}.tmp"
            with open(tmp, "w") as fh:
                json.dump({repr(k): list(v) for k, v in cache.items()}, fh)
            os.replace(tmp, cache_path)

    def bind_workspace(self):
        """pin this schedule's own split-K / q-split scratch into every launch that may use one."""
        for lst in (self.fwd_pre, self.fwd, self.bwd):
            for idx, f in enumerate(lst):
                fn = getattr(f, "func", None)
                if fn is ops.gemm or fn is ops.attn_bwd_dkv:
                    kw = dict(f.keywords)
                    # launches that run on the side stream get their own scratch: they may overlap main-stream GEMMs
                    kw["workspace"] = self.ws_side if getattr(f, "side", False) else self.ws
                    lst[idx] = self._rebound(f, fn, kw)

    @staticmethod
    def _rebound(f, fn, kw):
        p = partial(fn, *f.args, **kw)
        if getattr(f, "side", False):
            p.side = True
        return p

    def _side(self, launch):
        """mark a launch as independent of the main chain until the end of the list: it is forked onto a second
        stream (in a captured graph: a parallel branch) and joined when the list finishes."""
        launch.side = True
        if self.ws_side is None:
            self.ws_side = self._buf((4 * 2 ** 20,), torch.float32)
        return launch

    # ------------------------------------------------------------------ GroupNorm statistics in the producer
    GN_SLOTS = 8  # (16 and 32 measured: no gain — the slot atomics are not what the statistics launches wait for)

    def _produced(self, t: T):
        """the launch just appended to `fwd` is the one GEMM that writes every element of t.v"""
        t.producer = len(self.fwd) - 1
        return t

    # ------------------------------------------------------------------ two-launch GroupNorm (no finalize launch)
    GN2_MAX = 512  # (forward, backward) slot-sum slices of a schedule: 2 per GroupNorm

    def _gn2_slice(self):
        """[B, GN_SLOTS, G, 4] 64-bit words (fixed-point sum, sum of squares: csrc/common.h vn_fx_*) of the schedule's slot-sum arena (zeroed by one launch at the head of `fwd`): the
        statistics pass of a big GroupNorm adds its slab sums there and the apply kernel finishes them
        (vneti_groupnorm_fwd_2l / _bwd_2l); small GroupNorms never touch it"""
        if not hasattr(self, "_gn2_arena"):
            self._gn2_arena = self._buf((self.GN2_MAX, self.B, self.GN_SLOTS, self.groups, 4), torch.int64, zero=True)
            self._gn2_used = 0
        assert self._gn2_used < self.GN2_MAX
        self._gn2_used += 1
        return self._gn2_arena[self._gn2_used - 1]

    def fuse_gn_stats(self, fuse: bool = True):
        """finalises the schedule's GroupNorms; must run once per schedule even with fuse=False: the two-launch
        GroupNorms add into the slot-sum arena, which has to be cleared at the head of every forward"""
        n = self._fuse_gn_stats() if fuse else 0
        if getattr(self, "_gn2_used", 0):
            self.fwd.insert(0, self._gn2_arena[:self._gn2_used].zero_)
        return n

    def _fuse_gn_stats(self):
        """After autotuning: wherever a GroupNorm input comes out of one GEMM that runs without split-K, that GEMM's
        epilogue accumulates the per-(sample, group) sums (vneti_gemm_desc.gn_sums) and the GroupNorm becomes ONE
        launch (vneti_groupnorm_fwd_sums) instead of statistics + finalize + apply: one read of the tensor and two
        launches less per layer.  One memset at the head of the list clears all the sums of the schedule."""
        import os
        if os.environ.get("VNETI_NO_GN_FUSE"):
            return 0
        S, G, B = self.GN_SLOTS, self.groups, self.B
        todo = []
        for rec in self._gn_fusable:
            f = self.fwd[rec["prod"]]
            kw = f.keywords
            hw, Cc = rec["hw"], rec["C"]
            cpg = Cc // G
            # split-K launches keep the standalone statistics pass (their epilogue runs in the reduce kernel);
            # an un-tuned launch (split_k not pinned) is pinned to 1 here
            if getattr(f, "func", None) is ops.conv3x3_in:  # the direct conv_in: no split-K, same gn_sums contract
                if hw % 256 == 0 and cpg % 4 == 0 and 128 % cpg == 0 and kw.get("gn_sums") is None:
                    todo.append(rec)
                continue
            if getattr(f, "func", None) is not ops.gemm or kw.get("batch"):
                continue
            if kw.get("split_k") != 1:  # 0 / unset = the library heuristic: ask what it resolves to
                M_, N_, K_ = self._gemm_key(f)[:3]
                ws = kw.get("workspace")
                wsb = ws.numel() * ws.element_size() if ws is not None else 0
                if ops.gemm_select_split(M_, N_, K_, 1, kw.get("tile_hint") or 0, wsb) != 1:
                    continue
            if hw < 64 or not (cpg >= 8 or cpg == 4) or kw.get("gn_sums") is not None:
                continue
            todo.append(rec)
        if not todo:
            return 0
        self.gn_sums = self._buf((len(todo), B, S, G, 4), torch.int64, zero=True)
        for i, rec in enumerate(todo):
            sums = self.gn_sums[i]
            f = self.fwd[rec["prod"]]
            kw = dict(f.keywords)
            kw.update(gn_sums=sums, gn_hw=rec["hw"], gn_groups=G, gn_slots=S)
            if f.func is ops.gemm:
                kw.update(split_k=1)
            self.fwd[rec["prod"]] = self._rebound(f, f.func, kw)
            g = rec["gn"]
            self.fwd[rec["idx"]] = partial(ops.groupnorm_fwd_sums, rec["x"], rec["y"], g["gamma"], g["beta"], sums, S,
                                           g["mean"], g["rstd"], B, rec["hw"], rec["C"], G, rec["eps"], rec["silu"])
        arena = self.gn_sums
        self.fwd.insert(0, arena.zero_)
        self._gn_fusable = []
        return len(todo)

    # ------------------------------------------------------------------ layer builders
    def _gn(self, x: T, name, w, eps, silu):
        Cc = x.cols
        hw = x.rows // self.B
        rec = dict(kind="gn", x=x, gamma=self._w32(w[name + ".weight"]), beta=self._w32(w[name + ".bias"]),
                   mean=self._buf((self.B * self.groups,), torch.float32),
                   rstd=self._buf((self.B * self.groups,), torch.float32), silu=silu, hw=hw)
        y = self._buf((x.rows, Cc))
        import os
        if os.environ.get("VNETI_GN_3L"):  # A/B aid: the statistics / finalize / apply form
            self.fwd.append(partial(ops.groupnorm_fwd, x.v, y, rec["gamma"], rec["beta"], rec["mean"], rec["rstd"],
                                    self.gn_ws, self.B, hw, Cc, self.groups, eps, silu))
        else:
            self.fwd.append(partial(ops.groupnorm_fwd_2l, x.v, y, rec["gamma"], rec["beta"], self._gn2_slice(),
                                    self.GN_SLOTS, rec["mean"], rec["rstd"], self.B, hw, Cc, self.groups, eps, silu))
            if self.need_backward:
                rec["bsums"] = self._gn2_slice()
        if x.producer is not None:
            self._gn_fusable.append(dict(idx=len(self.fwd) - 1, prod=x.producer, gn=rec, x=x.v, y=y, hw=hw, C=Cc,
                                         eps=eps, silu=silu))
        return y, rec

    def _gn_bwd_fn(self, rec, dy):
        x = rec["x"]
        if "bsums" not in rec:
            fn = lambda out, accum: ops.groupnorm_bwd(dy, x.v, rec["gamma"], rec["beta"], rec["mean"], rec["rstd"], out,
                                                      self.gn_ws, self.B, rec["hw"], x.cols, self.groups,
                                                      rec["silu"], accum=accum)
        else:
            fn = lambda out, accum: ops.groupnorm_bwd_2l(dy, x.v, rec["gamma"], rec["beta"], rec["mean"], rec["rstd"], out,
                                                         rec["bsums"], self.GN_SLOTS, self.gn_ws, self.B, rec["hw"], x.cols,
                                                         self.groups, rec["silu"], accum=accum)
        # (tools/kernel_roofline.py: launch class and algorithmic bytes — dy and x read in the statistics pass and again
        #  in the apply pass, dx written once)
        fn.vn_cost = ("GroupNorm(+SiLU) bwd", 5.0 * self.B * rec["hw"] * x.cols * 2)
        return fn

    def _conv_desc(self, Hi, Wi, Ci, Ho, Wo, stride, pad, ups, ldx, mode=1):
        return dict(mode=mode, Hi=Hi, Wi=Wi, Ci=Ci, Ho=Ho, Wo=Wo, stride=stride, pad_t=pad, pad_l=pad, ups=ups,
                    ldx=ldx, korder=1 if (packing.KORDER_CM and Ci % 64 == 0) else 0)  # the packing's own predicate

    def _resnet(self, x: T, cin, cout, name, w, out_view, h, wd, need_dx=True):
        """ResnetBlock2D: GN+SiLU -> conv1 (+ time-embedding row add) -> GN+SiLU -> conv2 + shortcut."""
        M = x.rows
        n1, gn1 = self._gn(x, name + "norm1", w, self.eps, True)
        w1 = self._w16(packing.conv3x3_fwd(w[name + "conv1.weight"]))
        b1 = self._w32(w[name + "conv1.bias"])
        radd = None
        if name in self.temb_off:
            off, n = self.temb_off[name]
            radd = self.temb_all[:, off:off + n]
        h1 = T(self._buf((M, cout)))
        self.fwd.append(partial(ops.gemm, n1, w1, h1.v, bias=b1, rowadd=radd, rows_per_group=h * wd, M=M,
                                conv=self._conv_desc(h, wd, cin, h, wd, 1, 1, 0, cin)))
        self._produced(h1)
        n2, gn2 = self._gn(h1, name + "norm2", w, self.eps, True)
        w2 = self._w16(packing.conv3x3_fwd(w[name + "conv2.weight"]))
        b2 = self._w32(w[name + "conv2.bias"])
        out = T(out_view if out_view is not None else self._buf((M, cout)))
        wsc = None
        if cin != cout:
            wsc = self._w16(w[name + "conv_shortcut.weight"].reshape(cout, cin))
            bsc = self._w32(w[name + "conv_shortcut.bias"])
            sc_buf = self._buf((M, cout))
            self.fwd.append(partial(ops.gemm, x.v, wsc, sc_buf, bias=bsc))
            resid = sc_buf
        else:
            resid = x.v
        self.fwd.append(partial(ops.gemm, n2, w2, out.v, bias=b2, resid=resid, M=M,
                                conv=self._conv_desc(h, wd, cout, h, wd, 1, 1, 0, cout)))
        self._produced(out)
        rec = dict(kind="resnet", x=x, out=out, h1=h1, gn1=gn1, gn2=gn2, cin=cin, cout=cout, h=h, wd=wd,
                   need_dx=need_dx, name=name)
        if self.need_backward and need_dx:
            rec["w2d"] = self._w16(packing.conv3x3_dgrad(w[name + "conv2.weight"]))
            rec["w1d"] = self._w16(packing.conv3x3_dgrad(w[name + "conv1.weight"]))
            if wsc is not None:
                rec["wscd"] = self._w16(w[name + "conv_shortcut.weight"].reshape(cout, cin).t())
        self.tape.append(rec)
        return out

    def _resnet_bwd(self, r):
        if not r["need_dx"]:
            return
        x, out, h, wd, cin, cout = r["x"], r["out"], r["h"], r["wd"], r["cin"], r["cout"]
        M = x.rows
        dout = out.g
        assert out.gw, f"resnet {r['name']}: output gradient was never produced"
        dn2 = self._tmp("dA", M, cout)
        self.bwd.append(partial(ops.gemm, dout, r["w2d"], dn2, M=M,
                                conv=self._conv_desc(h, wd, cout, h, wd, 1, 1, 0, dout.stride(0), mode=2)))
        dh1 = self._tmp("dB", M, cout)
        self.bwd.append(partial(self._gn_bwd_fn(r["gn2"], dn2), dh1, None))
        dn1 = self._tmp("dC", M, cin)
        self.bwd.append(partial(ops.gemm, dh1, r["w1d"], dn1, M=M,
                                conv=self._conv_desc(h, wd, cout, h, wd, 1, 1, 0, cout, mode=2)))
        if "wscd" in r:
            self._contrib_gemm(x, dout, r["wscd"])
            self._contrib(x, self._gn_bwd_fn(r["gn1"], dn1))
        else:
            self._contrib(x, self._gn_bwd_fn(r["gn1"], dn1), extra=dout)

    def _ln(self, xv, name, w):
        rows, Cc = xv.shape
        rec = dict(x=xv, gamma=self._w32(w[name + ".weight"]), beta=self._w32(w[name + ".bias"]),
                   mean=self._buf((rows,), torch.float32), rstd=self._buf((rows,), torch.float32))
        y = self._buf((rows, Cc))
        self.fwd.append(partial(ops.layernorm_fwd, xv, y, rec["gamma"], rec["beta"], rec["mean"], rec["rstd"], 1e-5))
        return y, rec

    def _ln_bwd(self, rec, dy, dx, accum, f16_copy=None):
        ops.layernorm_bwd(dy, rec["x"], rec["gamma"], rec["mean"], rec["rstd"], dx, accum=accum, f16_copy=f16_copy)

    def _downsample(self, x: T, Cc, name, w, out_view, h, wd, pad=1):
        """Downsample2D: 3x3 stride-2 conv; pad=1 (UNet) or pad=0 with bottom/right zero padding (VAE)."""
        M = x.rows // 4
        wf = self._w16(packing.conv3x3_fwd(w[name + "weight"]))
        b = self._w32(w[name + "bias"])
        out = T(out_view if out_view is not None else self._buf((M, Cc)))
        self.fwd.append(partial(ops.gemm, x.v, wf, out.v, bias=b, M=M,
                                conv=self._conv_desc(h, wd, Cc, h // 2, wd // 2, 2, pad, 0, x.v.stride(0))))
        self._produced(out)
        rec = dict(kind="down", x=x, out=out, C=Cc, h=h, wd=wd, pad=pad)
        if self.need_backward:
            rec["wd_"] = self._w16(packing.conv3x3_dgrad(w[name + "weight"]))
        self.tape.append(rec)
        return out

    def _downsample_bwd(self, r):
        x, out, Cc, h, wd = r["x"], r["out"], r["C"], r["h"], r["wd"]
        assert out.gw
        desc = self._conv_desc(h // 2, wd // 2, Cc, h, wd, 2, r["pad"], 0, out.g.stride(0), mode=2)
        self._contrib_gemm(x, out.g, r["wd_"], M=x.rows, conv=desc)

    # ------------------------------------------------------------------ execution
    def forward_pre(self):
        for f in self.fwd_pre:
            f()

    def forward_main(self):
        for f in self.fwd:
            f()

    def forward(self):
        self.forward_pre()
        self.forward_main()

    def _run(self, lst):
        side = None
        main = None
        for f in lst:
            if getattr(f, "side", False):
                if side is None:
                    if self._side_stream is None:
                        self._side_stream = torch.cuda.Stream()
                    side, main = self._side_stream, torch.cuda.current_stream()
                side.wait_stream(main)  # fork point: everything issued so far
                with torch.cuda.stream(side):
                    f()
            else:
                f()
        if side is not None:
            main.wait_stream(side)  # join

    def backward(self):
        self._run(self.bwd)

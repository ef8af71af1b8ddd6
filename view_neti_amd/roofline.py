"""Per-launch cost model and per-class roofline timing of the train step (SURVEY 8d "per-kernel bar"): every launch of the
step's schedule gets a class, a bound (MFMA 2.5 PF dense f16 / HBM 8 TB/s, MI355X_MICROARCH.md) and its ALGORITHMIC cost;
classes are timed with HIP events on the launch stream while the whole list replays in schedule order, so each launch starts
with its operands evicted by its predecessors, as inside the step.  Used by bench.py (`roofline.classes` of the JSON line)
and tools/kernel_roofline.py (profiles/rNN_kernel_roofline.json).  Measurement code: nothing on the step's path imports it.
"""
from __future__ import annotations

import torch

from . import ops

import os as _os

_CSRC = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "csrc")
MFMA_PEAK_TFLOPS = 2500.0  # fp16/bf16 dense MFMA peak
HBM_PEAK_TBS = 8.0         # HBM3E spec peak
PF, TB = MFMA_PEAK_TFLOPS, HBM_PEAK_TBS


def kernel_tree_sha() -> str:
    """sha256 (16 hex digits) over the kernel sources (csrc/*.hip, *.h, include/vneti.h): counter passes are stamped with it
    (tools/pmc_summary.py) and bench.py quotes a committed `traffic` figure only when the stamp matches the tree it runs"""
    import hashlib
    import os
    here = _CSRC
    files = sorted(os.path.join(here, f) for f in os.listdir(here) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(os.path.dirname(here)), "include", "vneti.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def gemm_cost(f):
    """(M, N, K, batch, FLOPs, algorithmic HBM bytes, MFMA floor [s], HBM floor [s]) of one bound ops.gemm launch.
    Bytes: every operand once — A (the plain matrix, or the NHWC image an implicit conv gathers from: NOT its 9x im2col
    expansion), the weights, the output, plus the fused epilogue operands."""
    kw = f.keywords
    A, Bm = f.args[0], f.args[1]
    M = kw.get("M") or A.shape[-2]
    N, K = kw.get("N") or Bm.shape[-2], kw.get("K") or Bm.shape[-1]
    batch = kw.get("batch") or 1
    conv = kw.get("conv")
    a_bytes = (M // (conv["Ho"] * conv["Wo"])) * conv["Hi"] * conv["Wi"] * conv["Ci"] * 2 if conv else M * K * 2 * batch
    out = f.args[2]
    c_cols = N * (2 if kw.get("geglu") == 2 else 1)
    extra = sum(M * c * 2 for c, key in ((N, "resid"), (N, "out2"), (c_cols, "gate")) if kw.get(key) is not None)
    if kw.get("geglu") == 1:
        extra -= M * N  # out2 of the GEGLU projection is [M, N/2]
    nbytes = a_bytes + N * K * 2 * batch + M * c_cols * out.element_size() * batch + extra
    flops = 2.0 * M * N * K * batch
    return M, N, K, batch, flops, nbytes, flops / (MFMA_PEAK_TFLOPS * 1e12), nbytes / (HBM_PEAK_TBS * 1e12)



def cost(f):
    """(class, flops, bytes) of one launch; None = not classified (closures of the backward builders etc.)"""
    fn, a, kw = getattr(f, "func", None), getattr(f, "args", ()), getattr(f, "keywords", {}) or {}
    if fn is ops.gemm:
        # rated against its TRUE bound: below the ridge (algorithmic bytes / 8 TB/s > FLOPs / 2.5 PF) a launch is a
        # bandwidth kernel — the short-K linears (N = K = 320 / 640 at M = 16384 / 4096 ...) — and counts by bytes
        M, N, K, batch, flops, nbytes, t_mfma, t_hbm = gemm_cost(f)
        if t_hbm > t_mfma:
            return ("gemm, below the ridge (short-K linears: HBM-bound)", 0, nbytes)
        return ("gemm / implicit-GEMM conv (MFMA-bound)", flops, 0)
    if fn in (ops.attn_fwd, ops.attn_bwd_dq, ops.attn_bwd_dkv):
        i = {ops.attn_fwd: 5, ops.attn_bwd_dq: 7, ops.attn_bwd_dkv: 8}[fn]
        Bn, H, Nq, Nk, D = a[i:i + 5]
        causal = a[i + 6]
        mm = {ops.attn_fwd: 2, ops.attn_bwd_dq: 3, ops.attn_bwd_dkv: 4}[fn]  # matmuls of Nq x Nk x D executed
        name = {ops.attn_fwd: "attention fwd", ops.attn_bwd_dq: "attention bwd dQ", ops.attn_bwd_dkv: "attention bwd dK/dV"}[fn]
        scores = Bn * H * Nq * Nk * (0.5 if causal else 1.0)
        # what the head dim allows: per score and SIMD, the MFMA cycles EXECUTED (32x32x16 = 32 cycles per
        # 1024 scores and k-step; contractions over d pad to 16, output rows over d to 32) plus the VALU issue (2 cycles per
        # wave64 instruction, v_exp_f32 at 5/3 of one: MI355X_MICROARCH.md) — the two pipes are observed to ADD on this part
        ks, db = -(-D // 16), -(-D // 32)
        mfma_c = {ops.attn_fwd: ks + 2 * db, ops.attn_bwd_dq: 2 * ks + 2 * db, ops.attn_bwd_dkv: 2 * ks + 4 * db}[fn] * 32 / 1024.0
        valu_c = {ops.attn_fwd: 2.8, ops.attn_bwd_dq: 4.0, ops.attn_bwd_dkv: 4.0}[fn] * 2 / 64.0 + (2 * 5 / 3) / 64.0
        cob = scores * (mfma_c + valu_c) / (1024 * 2.4e9)  # seconds: 256 CUs x 4 SIMDs at the 2.4 GHz peak clock
        return (name, mm * 2.0 * scores * D, 0, cob)
    if fn is ops.attn_bwd_small:  # (Q, K, V, dO, O, lse, dQ, dK, dV, Bn, H, N, D, scale, causal): S and dP in both roles + dQ, dK, dV
        Bn, H, N, D = a[9:13]
        return ("attention bwd (short sequences, one launch)", 7 * 2.0 * Bn * H * N * N * D * (0.5 if a[14] else 1.0), 0)
    if fn in (ops.groupnorm_fwd, ops.groupnorm_fwd_sums, ops.groupnorm_fwd_2l):
        Bn, HW, C = (a[7], a[8], a[9]) if fn is ops.groupnorm_fwd else (a[8], a[9], a[10])  # _sums and _2l share a layout
        return ("GroupNorm(+SiLU) fwd" + (" (stats in producer)" if fn is ops.groupnorm_fwd_sums else ""), 0, 2.0 * Bn * HW * C * 2)
    if fn is ops.layernorm_fwd:
        x, y = a[0], a[1]
        return ("LayerNorm fwd", 0, x.numel() * x.element_size() + y.numel() * y.element_size())
    if getattr(f, "vn_cost", None) is not None:  # closures tagged where they are built (engine/schedule.py)
        return (f.vn_cost[0], 0, f.vn_cost[1])
    if getattr(fn, "__name__", "") == "_ln_bwd":  # partial(self._ln_bwd, rec, dy, dx, accum[, f16_copy])
        x = a[0]["x"]
        n = x.numel()
        extra = (a[3].numel() * a[3].element_size() if a[3] is not None else 0) + (n * 2 if len(a) > 4 and a[4] is not None else 0)
        return ("LayerNorm bwd", 0, n * x.element_size() + a[1].numel() * a[1].element_size() + a[2].numel() * a[2].element_size() + extra)
    if fn in (ops.groupnorm_bwd, ops.groupnorm_bwd_2l):
        x = a[1]
        return ("GroupNorm(+SiLU) bwd", 0, 5.0 * x.numel() * 2)
    if fn is ops.add:
        return ("elementwise glue (add, 2x2 sum, transposes, noise, loss)", 0, sum(t.numel() * t.element_size() for t in a[:3]))
    if fn is ops.sum2x2:
        return ("elementwise glue (add, 2x2 sum, transposes, noise, loss)", 0, a[0].numel() * 2 + a[1].numel() * 2)
    if fn in (ops.sample_add_noise, ops.mse_loss_grad, getattr(ops, "transpose", None), getattr(ops, "timestep_embedding", None)):
        return ("elementwise glue (add, 2x2 sum, transposes, noise, loss)", 0,
                sum(t.numel() * t.element_size() for t in a if isinstance(t, torch.Tensor)))
    if fn is ops.layernorm_bwd:
        return ("LayerNorm bwd", 0, sum(t.numel() * t.element_size() for t in (a[0], a[1], a[5])))
    if fn is ops.conv3x3_in:
        return ("VAE conv_in (direct)", 0, a[0].numel() * 4 + a[3].numel() * 2)
    if fn is ops.softmax_rows:
        return ("softmax rows (VAE mid attention)", 0, 2.0 * a[0].numel() * 2)
    if fn is ops.geglu_fwd:
        return ("GEGLU fwd", 0, a[0].numel() * 2 + a[1].numel() * 2)
    if fn is ops.geglu_bwd:
        return ("GEGLU bwd", 0, a[0].numel() * 2 + 2 * a[1].numel() * 2)
    return None




def classify(launches):
    """{class name: {fs: [launch...], flops, bytes, cob}} over a launch list (TrainStepEngine.launches())"""
    classes = {}
    for f in launches:
        c = cost(f)
        if c is None:
            # the small launches of the text path (mapper, embeddings, bypass), the device RNG and the glue: latency-bound;
            # their byte count is every tensor argument once (an upper bound of what they move)
            args = list(getattr(f, "args", ())) + list((getattr(f, "keywords", None) or {}).values())
            c = ("small launches (text path, RNG, layout glue)", 0,
                 float(sum(t.numel() * t.element_size() for t in args if isinstance(t, torch.Tensor))) or 1.0)
        d = classes.setdefault(c[0], dict(fs=[], flops=0.0, bytes=0.0, cob=0.0))
        d["fs"].append(f)
        d["flops"] += c[1]
        d["bytes"] += c[2]
        if len(c) > 3:
            d["cob"] += c[3]
    return classes


def time_classes(launches, classes=None, reps: int = 2):
    """ONE pass over the schedule per repetition with an event pair around every launch (events cost ~1 us of stream time
    each, the same for every class); returns {class: {launches, ms_per_step, bound, achieved, peak, unit, frac[, ...]}}"""
    classes = classes or classify(launches)
    owner = {id(f): name for name, d in classes.items() for f in d["fs"]}
    tot = {name: 0.0 for name in classes}
    for rep in range(reps + 1):
        evs = []
        for f in launches:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            f()
            e.record()
            evs.append((owner[id(f)], s, e))
        torch.cuda.synchronize()
        if rep:  # the first pass warms up
            for name, s, e in evs:
                tot[name] += s.elapsed_time(e)
    out = {}
    for name, d in classes.items():
        ms = tot[name] / reps
        r = dict(launches=len(d["fs"]), ms_per_step=ms)
        if d["flops"]:
            r.update(bound="mfma", algorithmic_gflop=d["flops"] / 1e9, achieved=d["flops"] / (ms * 1e-3) / 1e12, peak=PF,
                     unit="TFLOP/s")
            r["frac"] = r["achieved"] / PF
            if d.get("cob"):
                # the co-bound: the fraction of the MFMA peak this head-dim mix allows when the (padded) MFMA cycles and
                # the softmax's VALU / exp issue serialise, and how much of THAT the kernels reach
                r["cobound_frac_of_peak"] = d["flops"] / d["cob"] / 1e12 / PF
                r["frac_of_cobound"] = d["cob"] / (ms * 1e-3)
        elif d["bytes"]:
            r.update(bound="hbm", algorithmic_mb=d["bytes"] / 1e6, achieved=d["bytes"] / (ms * 1e-3) / 1e12, peak=TB,
                     unit="TB/s")
            r["frac"] = r["achieved"] / TB
        out[name] = r
    return out

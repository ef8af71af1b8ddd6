"""Dev tool: every launch class of the train step against ITS roofline (SURVEY §8d "per-kernel bar"): launches are
replayed in schedule order (so each starts with its operands evicted by its predecessors, as inside the step), timed
with HIP events per class, and divided into the class's algorithmic FLOPs (MFMA kernels, peak 2.5 PF dense f16) or
bytes (HBM kernels, peak 8 TB/s).  Writes gpurun_out/<tag>_kernel_roofline.json and prints a markdown table.
   python tools/kernel_roofline.py r01f"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench
from view_neti_amd import ops

tag = sys.argv[1] if len(sys.argv) > 1 else "r01x"
args = argparse.Namespace(model="sd15", batch=4, resolution=512)
_, eng = bench.build_engine(args, 0, 1)
eng.step_eager(); torch.cuda.synchronize()
PF, TB = 2500.0, 8.0


def cost(f):
    """(class, flops, bytes) of one launch; None = not classified (closures of the backward builders etc.)"""
    fn, a, kw = getattr(f, "func", None), getattr(f, "args", ()), getattr(f, "keywords", {}) or {}
    if fn is ops.gemm:
        # rated against its TRUE bound: below the ridge (algorithmic bytes / 8 TB/s > FLOPs / 2.5 PF) a launch is a
        # bandwidth kernel — the short-K linears (N = K = 320 / 640 at M = 16384 / 4096 ...) — and counts by bytes
        M, N, K, batch, flops, nbytes, t_mfma, t_hbm = bench.gemm_cost(f)
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


launches = eng.launches()
classes = {}
for f in launches:
    c = cost(f)
    if c is None:
        # the small launches of the text path (mapper, embeddings, bypass), the device RNG and the glue: latency-bound; their
        # byte count is every tensor argument once (an upper bound of what they move)
        args = list(getattr(f, "args", ())) + list((getattr(f, "keywords", None) or {}).values())
        c = ("small launches (text path, RNG, layout glue)", 0,
             float(sum(t.numel() * t.element_size() for t in args if isinstance(t, torch.Tensor))) or 1.0)
    name = c[0]
    d = classes.setdefault(name, dict(fs=[], flops=0.0, bytes=0.0, cob=0.0))
    d["fs"].append(f)
    if c:
        d["flops"] += c[1]; d["bytes"] += c[2]
        if len(c) > 3:
            d["cob"] += c[3]
if os.environ.get("PER_LAUNCH"):
    # every HBM-bound launch by itself (events around each one, in schedule order), grouped by (class, bytes)
    import collections
    per = collections.OrderedDict()
    hbm = {id(f): (name, cost(f)) for name, d in classes.items() if not d["flops"] for f in d["fs"]}
    for rep in range(3):
        evs = []
        for f in launches:
            if id(f) in hbm:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); f(); e.record(); evs.append((f, s, e))
            else:
                f()
        torch.cuda.synchronize()
        if rep:
            for f, s, e in evs:
                name, c = hbm[id(f)]
                nb = c[2] if c else 0.0
                fn = getattr(getattr(f, "func", f), "__name__", "?")
                k = (name, fn, int(nb))
                a = per.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e) * 1e3
    print("class | fn | MB | launches/step | us each | TB/s | ms/step")
    for (name, fn, nb), (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        n2 = n / 2; us = t / n
        print(f"{name[:38]:38s} {fn[:22]:22s} {nb/1e6:8.2f} MB x{n2:5.1f} {us:7.1f} us {nb/us/1e6 if us else 0:6.2f} TB/s {t/2/1e3:6.3f} ms")
    sys.exit(0)
# time: replay the whole list in order, recording events only around the launches of one class at a time
out = {}
for name, d in classes.items():
    mine = set(id(f) for f in d["fs"])
    tot = 0.0
    for rep in range(3):
        evs = []
        for f in launches:
            if id(f) in mine:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); f(); e.record(); evs.append((s, e))
            else:
                f()
        torch.cuda.synchronize()
        if rep:
            tot += sum(s.elapsed_time(e) for s, e in evs)
    ms = tot / 2
    r = dict(launches=len(d["fs"]), ms_per_step=ms)
    if d["flops"]:
        r.update(bound="mfma", algorithmic_gflop=d["flops"] / 1e9, achieved=d["flops"] / (ms * 1e-3) / 1e12, peak=PF, unit="TFLOP/s")
        r["frac"] = r["achieved"] / PF
        if d.get("cob"):
            # the co-bound: the fraction of the MFMA peak this head-dim mix allows when the (padded) MFMA cycles and the
            # softmax's VALU / exp issue serialise, and how much of THAT the kernels reach
            r["cobound_frac_of_peak"] = d["flops"] / d["cob"] / 1e12 / PF
            r["frac_of_cobound"] = d["cob"] / (ms * 1e-3)
    elif d["bytes"]:
        r.update(bound="hbm", algorithmic_mb=d["bytes"] / 1e6, achieved=d["bytes"] / (ms * 1e-3) / 1e12, peak=TB, unit="TB/s")
        r["frac"] = r["achieved"] / TB
    out[name] = r
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/{tag}_kernel_roofline.json", "w"), indent=1)
print("| launch class | launches/step | ms/step | algorithmic cost | achieved | of peak |")
print("|---|---|---|---|---|---|")
for name, r in sorted(out.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if "frac" in r:
        costs = f"{r['algorithmic_gflop']/1e3:.2f} TFLOP" if r["bound"] == "mfma" else f"{r['algorithmic_mb']/1e3:.2f} GB"
        extra = f" (co-bound {r['cobound_frac_of_peak']:.2f} of peak: at {r['frac_of_cobound']:.2f} of it)" if "frac_of_cobound" in r else ""
        print(f"| {name} | {r['launches']} | {r['ms_per_step']:.2f} | {costs} | {r['achieved']:.2f} {r['unit']} | {r['frac']:.2f} of {r['peak']:g}{extra} |")
    else:
        print(f"| {name} | {r['launches']} | {r['ms_per_step']:.2f} | — | — | — |")

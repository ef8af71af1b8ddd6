"""UNet2DConditionModel forward + input-gradient backward as a static schedule of HIP kernel
launches (no autograd, no tracing compiler).

What the reference does at this point: `model_pred = self.unet(noisy_latents, timesteps, _hs).sample`
followed by `accelerator.backward(loss)` (training/coach.py:197-198,214) through diffusers'
UNet with the XTI attention processor installed on every attention module
(training/coach.py:679-680, models/xti_attention_processor.py:9-57).  All UNet weights are frozen
(training/coach.py:642-653): the backward here is dgrad-only and its sole products are the
gradients of the 16 key contexts and 16 value contexts.

Layout: every activation is channels-last — an image is a [B*H*W, C] f16 matrix — so the
conv -> transformer -> conv transitions need no permutes, and skip-connection concats are free:
each up-block resnet owns one [M, C_h + C_skip] buffer; the producer of `h` and the down-path
producer of the skip write straight into its column slices (row stride = full width).

The schedule is built once (shapes are static); `forward()`/`backward()` just replay the launch
lists, which makes the step capturable in a hipGraph.
"""
from __future__ import annotations

from functools import partial
from typing import Dict

import torch

from .. import ops, packing
from .. import sd_config as sc
from .schedule import Schedule, T, rup as _rup


class UNetEngine(Schedule):
    def __init__(self, cfg: sc.UNetConfig, weights: Dict[str, torch.Tensor], batch: int, height: int, width: int,
                 ctx_len: int = 77, device: str = "cuda", need_backward: bool = True, autotune: bool = True,
                 fuse_gn: bool = True):
        super().__init__(batch, cfg.norm_num_groups, cfg.norm_eps, device, need_backward)
        self.cfg = cfg
        self.H, self.W = height, width
        self.L = ctx_len
        self.nl = cfg.n_cross_layers
        B, L, Dc = batch, ctx_len, cfg.cross_attention_dim
        # inputs / outputs (static buffers)
        self.x_in = self._buf((B, cfg.in_channels, height, width), torch.float32)
        self.timesteps = torch.zeros(B, dtype=torch.int64, device=device)
        self.ctx_k = self._buf((self.nl, B * L, Dc))
        self.ctx_v = self._buf((self.nl, B * L, Dc))
        self.dctx_k = self._buf((self.nl, B * L, Dc))
        self.dctx_v = self._buf((self.nl, B * L, Dc))
        self.pred = self._buf((B * height * width, 8))[:, : cfg.out_channels]
        self.dpred = self._buf((B * height * width, 8), zero=True)[:, : cfg.out_channels]
        self._pack_time_weights(weights)
        self._plan_kv()
        self._build(weights)
        self._kv_fwd_ops()
        if need_backward:
            self._build_backward()
            self._kv_bwd_ops()
        if autotune:
            self.autotune()
        self.bind_workspace()
        self.fuse_gn_stats(fuse=fuse_gn)

    # ------------------------------------------------------------------ time embedding
    def _pack_time_weights(self, w):
        cfg = self.cfg
        names = [k[: -len("time_emb_proj.weight")] for k in sc.unet_shapes(cfg) if k.endswith("time_emb_proj.weight")]
        self.temb_off = {}
        ws, bs, off = [], [], 0
        for n in names:
            wt = w[n + "time_emb_proj.weight"]
            self.temb_off[n] = (off, wt.shape[0])
            ws.append(wt)
            bs.append(w[n + "time_emb_proj.bias"])
            off += wt.shape[0]
        self.temb_total = off
        self.w_temb_all = self._w16(torch.cat(ws, 0))
        self.b_temb_all = self._w32(torch.cat(bs, 0))
        self.w_t1 = self._w16(w["time_embedding.linear_1.weight"])
        self.b_t1 = self._w32(w["time_embedding.linear_1.bias"])
        self.w_t2 = self._w16(w["time_embedding.linear_2.weight"])
        self.b_t2 = self._w32(w["time_embedding.linear_2.bias"])
        c0, td = cfg.block_out_channels[0], cfg.temb_dim
        self.t_sin = self._buf((self.B, c0))
        self.t_h = self._buf((self.B, td))
        self.t_emb = self._buf((self.B, td))
        self.temb_all = self._buf((self.B, self.temb_total))

    def _time_ops(self):
        f = self.fwd_pre  # depends on the timesteps only
        f.append(partial(ops.timestep_embedding, self.timesteps, self.t_sin))
        f.append(partial(ops.gemm, self.t_sin, self.w_t1, self.t_h, bias=self.b_t1, act=ops.ACT_SILU, tile_hint=3))
        # every consumer applies SiLU to temb first (ResnetBlock2D), so store SiLU(temb) directly
        f.append(partial(ops.gemm, self.t_h, self.w_t2, self.t_emb, bias=self.b_t2, act=ops.ACT_SILU, tile_hint=3))
        f.append(partial(ops.gemm, self.t_emb, self.w_temb_all, self.temb_all, bias=self.b_temb_all, tile_hint=3))

    # ------------------------------------------------------------------ XTI key/value projections, batched
    def _cross_widths(self):
        """channel width of every cross-attention layer in XTI order (down blocks, mid, up blocks)"""
        cfg = self.cfg
        boc, lpb = cfg.block_out_channels, cfg.layers_per_block
        ws = []
        for i, a in enumerate(cfg.down_has_attn):
            ws += [boc[i]] * lpb if a else []
        ws.append(boc[-1])
        for i, a in enumerate(reversed(cfg.down_has_attn)):
            ws += [tuple(reversed(boc))[i]] * (lpb + 1) if a else []
        assert len(ws) == self.nl
        return ws

    def _plan_kv(self):
        """The 2 x 16 to_k / to_v projections of the XTI contexts (and their 32 dgrads) are 77-row GEMMs: launch
        latency, not work.  Consecutive layers of one width share stacked weight / K / V buffers so that each run is
        ONE batched launch per operand (SD-1.5: 5 runs => 10 launches forward, 10 backward, instead of 32 + 32)."""
        B, L, Dc = self.B, self.L, self.cfg.cross_attention_dim
        ws = self._cross_widths()
        self.kv_runs, self.kv_slot = [], {}
        l0 = 0
        while l0 < len(ws):
            n = 1
            while l0 + n < len(ws) and ws[l0 + n] == ws[l0]:
                n += 1
            Cc = ws[l0]
            run = dict(l0=l0, n=n, C=Cc, wk=self._buf((n, Cc, Dc)), wv=self._buf((n, Cc, Dc)),
                       k=self._buf((n, B * L, Cc)), v=self._buf((n, B * L, Cc)))
            if self.need_backward:
                run.update(wkd=self._buf((n, Dc, Cc)), wvd=self._buf((n, Dc, Cc)),
                           dk=self._buf((n, B * L, Cc)), dv=self._buf((n, B * L, Cc)))
            for j in range(n):
                self.kv_slot[l0 + j] = (run, j)
            self.kv_runs.append(run)
            l0 += n

    def _kv_fwd_ops(self):
        B, L, Dc = self.B, self.L, self.cfg.cross_attention_dim
        for run in self.kv_runs:
            l0, n, Cc = run["l0"], run["n"], run["C"]
            for ctx, wgt, out in ((self.ctx_k, run["wk"], run["k"]), (self.ctx_v, run["wv"], run["v"])):
                # depends on the text side only: prologue launches
                self.fwd_pre.append(partial(ops.gemm, ctx[l0], wgt[0], out[0], batch=n, strideA=B * L * Dc,
                                            strideB=Cc * Dc, strideC=B * L * Cc))

    def _kv_bwd_ops(self):
        """dctx_k[l] = dK_l . Wk_l, dctx_v[l] = dV_l . Wv_l for every layer, after the last transformer backward
        (nothing in the UNet backward reads them)"""
        B, L, Dc = self.B, self.L, self.cfg.cross_attention_dim
        for run in self.kv_runs:
            l0, n, Cc = run["l0"], run["n"], run["C"]
            for dsrc, wgt, dctx in ((run["dk"], run["wkd"], self.dctx_k), (run["dv"], run["wvd"], self.dctx_v)):
                self.bwd.append(partial(ops.gemm, dsrc[0], wgt[0], dctx[l0], batch=n, strideA=B * L * Cc,
                                        strideB=Dc * Cc, strideC=B * L * Dc))

    def _transformer(self, x: T, Cc, heads, name, w, layer_idx, out_view, h, wd, need_dx=True):
        cfg = self.cfg
        B, L, Dc = self.B, self.L, cfg.cross_attention_dim
        M, N = x.rows, h * wd
        D = Cc // heads
        scale = D ** -0.5
        t = name + "transformer_blocks.0."
        r = dict(kind="transformer", x=x, C=Cc, heads=heads, D=D, N=N, name=name, layer=layer_idx, need_dx=need_dx,
                 scale=scale)
        g, r["gn"] = self._gn(x, name + "norm", w, 1e-6, False)
        w_in = w[name + "proj_in.weight"].reshape(Cc, Cc)
        r["w_in"], b_in = self._w16(w_in), self._w32(w[name + "proj_in.bias"])
        h0 = self._buf((M, Cc))
        self.fwd.append(partial(ops.gemm, g, r["w_in"], h0, bias=b_in))
        # ---- attn1 (self) ----
        n1, r["ln1"] = self._ln(h0, t + "norm1", w)
        wqkv = torch.cat([w[t + "attn1.to_q.weight"], w[t + "attn1.to_k.weight"], w[t + "attn1.to_v.weight"]], 0)
        r["wqkv"] = self._w16(wqkv)
        qkv = self._buf((M, 3 * Cc))
        self.fwd.append(partial(ops.gemm, n1, r["wqkv"], qkv))
        q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]
        o1 = self._buf((M, Cc))
        lse1 = self._buf((B, heads, N), torch.float32)
        self.fwd.append(partial(ops.attn_fwd, q, k, v, o1, lse1, B, heads, N, N, D, scale, False))
        r["wo1"], bo1 = self._w16(w[t + "attn1.to_out.0.weight"]), self._w32(w[t + "attn1.to_out.0.bias"])
        h1 = self._buf((M, Cc))
        self.fwd.append(partial(ops.gemm, o1, r["wo1"], h1, bias=bo1, resid=h0))
        # ---- attn2 (XTI cross attention: K from ctx_k[layer], V from ctx_v[layer]) ----
        n2, r["ln2"] = self._ln(h1, t + "norm2", w)
        r["wq2"] = self._w16(w[t + "attn2.to_q.weight"])
        run, slot = self.kv_slot[layer_idx]
        assert run["C"] == Cc, (name, layer_idx, Cc, run["C"])
        run["wk"][slot].copy_(w[t + "attn2.to_k.weight"])
        run["wv"][slot].copy_(w[t + "attn2.to_v.weight"])
        q2 = self._buf((M, Cc))
        k2, v2 = run["k"][slot], run["v"][slot]   # written by the batched prologue launches (_kv_fwd_ops)
        self.fwd.append(partial(ops.gemm, n2, r["wq2"], q2))
        o2 = self._buf((M, Cc))
        lse2 = self._buf((B, heads, N), torch.float32)
        self.fwd.append(partial(ops.attn_fwd, q2, k2, v2, o2, lse2, B, heads, N, L, D, scale, False))
        r["wo2"], bo2 = self._w16(w[t + "attn2.to_out.0.weight"]), self._w32(w[t + "attn2.to_out.0.bias"])
        h2 = self._buf((M, Cc))
        self.fwd.append(partial(ops.gemm, o2, r["wo2"], h2, bias=bo2, resid=h1))
        # ---- feed-forward (GEGLU) ----
        n3, r["ln3"] = self._ln(h2, t + "norm3", w)
        # GEGLU in the projection's epilogue: the rows of ff.net.0.proj are interleaved [h0..3 g0..3 h4..7 ...] at pack
        # time so a 16-byte chunk of the output tile holds matching halves; `p` (kept for the backward) stays in that
        # layout, `gg` = h * gelu(g) comes out of the same launch (no standalone geglu pass over the [M, 8C] tensor)
        wff1_il = packing.geglu_interleave(w[t + "ff.net.0.proj.weight"])
        r["wff1"], bff1 = self._w16(wff1_il), self._w32(packing.geglu_interleave(w[t + "ff.net.0.proj.bias"]))
        r["wff2"], bff2 = self._w16(w[t + "ff.net.2.weight"]), self._w32(w[t + "ff.net.2.bias"])
        p = self._buf((M, 8 * Cc))
        gg = self._buf((M, 4 * Cc))
        self.fwd.append(partial(ops.gemm, n3, r["wff1"], p, bias=bff1, out2=gg, geglu=1, split_k=1))
        h3 = self._buf((M, Cc))
        self.fwd.append(partial(ops.gemm, gg, r["wff2"], h3, bias=bff2, resid=h2))
        r["w_out"], b_out = self._w16(w[name + "proj_out.weight"].reshape(Cc, Cc)), self._w32(w[name + "proj_out.bias"])
        out = T(out_view if out_view is not None else self._buf((M, Cc)))
        self.fwd.append(partial(ops.gemm, h3, r["w_out"], out.v, bias=b_out, resid=x.v))
        self._produced(out)
        r.update(out=out, h0=h0, h1=h1, h2=h2, qkv=qkv, o1=o1, lse1=lse1, q2=q2, k2=k2, v2=v2, o2=o2, lse2=lse2,
                 p=p)
        if self.need_backward:
            tr = lambda a: self._w16(a.t())
            run["wkd"][slot].copy_(w[t + "attn2.to_k.weight"].t())
            run["wvd"][slot].copy_(w[t + "attn2.to_v.weight"].t())
            r["w_outd"] = tr(w[name + "proj_out.weight"].reshape(Cc, Cc))
            r["wff2d"], r["wff1d"] = tr(w[t + "ff.net.2.weight"]), tr(wff1_il)
            r["wo2d"] = tr(w[t + "attn2.to_out.0.weight"])
            if need_dx:
                r["wq2d"] = tr(w[t + "attn2.to_q.weight"])
                r["wo1d"] = tr(w[t + "attn1.to_out.0.weight"])
                r["wqkvd"] = tr(wqkv)
                r["w_ind"] = tr(w_in)
        self.tape.append(r)
        return out

    def _transformer_bwd(self, r):
        B, L, Cc, heads, D, N = self.B, self.L, r["C"], r["heads"], r["D"], r["N"]
        x, out = r["x"], r["out"]
        M = x.rows
        dout = out.g
        assert out.gw, f"transformer {r['name']}: output gradient was never produced"
        li = r["layer"]
        bw = self.bwd
        dh3 = self._tmp("tA", M, Cc)
        bw.append(partial(ops.gemm, dout, r["w_outd"], dh3))
        # ff.net.2 dgrad with the GEGLU backward in its epilogue: the [M, 4C] result d(h*gelu(g)) is never stored, the
        # launch writes dp = [d*gelu(g) | d*h*gelu'(g)] in p's interleaved layout
        dp = self._tmp("tC", M, 8 * Cc)
        bw.append(partial(ops.gemm, dh3, r["wff2d"], dp, gate=r["p"], gate_act=ops.ACT_GELU, geglu=2, split_k=1))
        dn3 = self._tmp("tD", M, Cc)
        bw.append(partial(ops.gemm, dp, r["wff1d"], dn3))
        dh2 = self._tmp("tE", M, Cc)
        bw.append(partial(self._ln_bwd, r["ln3"], dn3, dh2, dh3))
        # ---- attn2 backward ----
        do2 = self._tmp("tA", M, Cc)  # dh3 is dead
        bw.append(partial(ops.gemm, dh2, r["wo2d"], do2))
        delta = self._tmp("tdelta", B * heads, N, torch.float32)
        dq2 = self._tmp("tD", M, Cc)  # dn3 is dead
        # delta = rowsum(dO o O) comes out of the dQ kernel, which therefore runs before dK/dV
        if r["need_dx"]:
            bw.append(partial(ops.attn_bwd_dq, r["q2"], r["k2"], r["v2"], do2, r["lse2"], delta, dq2, B, heads, N, L, D,
                              r["scale"], False, O=r["o2"]))
        else:
            bw.append(partial(ops.attn_bwd_delta, do2, r["o2"], delta, B, heads, N, D))
        # per-layer dK/dV land in the run's stacked buffers; their projections back to the context gradients are
        # batched after the last layer (_kv_bwd_ops)
        run, slot = self.kv_slot[li]
        dk2, dv2 = run["dk"][slot], run["dv"][slot]
        bw.append(partial(ops.attn_bwd_dkv, r["q2"], r["k2"], r["v2"], do2, r["lse2"], delta, dk2, dv2, B, heads, N, L,
                          D, r["scale"], False))
        if not r["need_dx"]:
            return
        dn2 = self._tmp("tA", M, Cc)  # do2 is dead after dq
        bw.append(partial(ops.gemm, dq2, r["wq2d"], dn2))
        dh1 = self._tmp("tF", M, Cc)
        bw.append(partial(self._ln_bwd, r["ln2"], dn2, dh1, dh2))
        # ---- attn1 backward ----
        do1 = self._tmp("tA", M, Cc)
        bw.append(partial(ops.gemm, dh1, r["wo1d"], do1))
        qkv = r["qkv"]
        q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]
        dqkv = self._tmp("tC", M, 3 * Cc)  # dp is dead
        dq, dk, dv = dqkv[:, :Cc], dqkv[:, Cc:2 * Cc], dqkv[:, 2 * Cc:]
        bw.append(partial(ops.attn_bwd_dq, q, k, v, do1, r["lse1"], delta, dq, B, heads, N, N, D, r["scale"], False,
                          O=r["o1"]))
        bw.append(partial(ops.attn_bwd_dkv, q, k, v, do1, r["lse1"], delta, dk, dv, B, heads, N, N, D, r["scale"],
                          False))
        dn1 = self._tmp("tD", M, Cc)
        bw.append(partial(ops.gemm, dqkv, r["wqkvd"], dn1))
        dh0 = self._tmp("tE", M, Cc)  # dh2 is dead
        bw.append(partial(self._ln_bwd, r["ln1"], dn1, dh0, dh1))
        dg = self._tmp("tA", M, Cc)
        bw.append(partial(ops.gemm, dh0, r["w_ind"], dg))
        self._contrib(x, self._gn_bwd_fn(r["gn"], dg), extra=dout)

    def _upsample(self, x: T, Cc, name, w, out_view, h, wd):
        M = x.rows * 4
        wf = self._w16(packing.conv3x3_fwd(w[name + "weight"]))
        b = self._w32(w[name + "bias"])
        out = T(out_view if out_view is not None else self._buf((M, Cc)))
        self.fwd.append(partial(ops.gemm, x.v, wf, out.v, bias=b, M=M,
                                conv=self._conv_desc(h, wd, Cc, 2 * h, 2 * wd, 1, 1, 1, x.v.stride(0))))
        self._produced(out)
        rec = dict(kind="up", x=x, out=out, C=Cc, h=h, wd=wd)
        if self.need_backward:
            rec["wd_"] = self._w16(packing.conv3x3_dgrad(w[name + "weight"]))
        self.tape.append(rec)
        return out

    def _upsample_bwd(self, r):
        x, out, Cc, h, wd = r["x"], r["out"], r["C"], r["h"], r["wd"]
        assert out.gw and not x.gw
        M = out.rows
        dup = self._tmp("dA", M, Cc)
        self.bwd.append(partial(ops.gemm, out.g, r["wd_"], dup, M=M,
                                conv=self._conv_desc(2 * h, 2 * wd, Cc, 2 * h, 2 * wd, 1, 1, 0, out.g.stride(0), mode=2)))
        g = self._grad(x)
        self.bwd.append(partial(ops.sum2x2, dup, g, self.B, h, wd, Cc))
        x.gw = True

    # ------------------------------------------------------------------ whole network
    def _build(self, w):
        cfg = self.cfg
        B, H, W = self.B, self.H, self.W
        boc = cfg.block_out_channels
        nlev = len(boc)
        nres = cfg.layers_per_block + 1
        self._time_ops()
        # concat buffers of the up-block resnets, created first so that producers can target
        # their column slices: cat[(i,j)] = [ h (rin) | skip ]
        catT = {}
        for i in range(nlev):
            lvl = nlev - 1 - i
            m = B * (H >> lvl) * (W >> lvl)
            for j in range(nres):
                rin, skip, _ = sc.up_block_channels(cfg, i, j)
                catT[(i, j)] = T(self._buf((m, rin + skip)),
                                 self._buf((m, rin + skip)) if self.need_backward else None)

        def h_slot(key):
            rin = sc.up_block_channels(cfg, *key)[0]
            return catT[key].v[:, :rin]

        def adopt(t: T, key, part):
            """t lives in a column slice of cat[key]; its gradient is the same slice of dcat[key]
            and becomes 'written' when the consuming up-resnet has back-propagated."""
            rin = sc.up_block_channels(cfg, *key)[0]
            cat = catT[key]
            if cat.g is not None:
                t.g = cat.g[:, :rin] if part == "h" else cat.g[:, rin:]
            cat.children.append(t)
            return t

        # the skip stack is popped in reverse push order
        skip_targets = list(reversed([(i, j) for i in range(nlev) for j in range(nres)]))
        nskip = [0]

        def next_skip():
            key = skip_targets[nskip[0]]
            nskip[0] += 1
            rin = sc.up_block_channels(cfg, *key)[0]
            return key, catT[key].v[:, rin:]

        # conv_in (im2col on the NCHW f32 latents, K padded to 64)
        M0 = B * H * W
        col = self._buf((M0, 64))
        w_in = self._w16(packing.pad_rows(packing.conv3x3_fwd(w["conv_in.weight"]), 64))
        b_in = self._w32(w["conv_in.bias"])
        key, sv = next_skip()
        self.fwd.append(partial(ops.im2col3x3_small, self.x_in, col, B, cfg.in_channels, H, W, H, W, 1, 1, 1,
                                self.x_in.stride()))
        self.fwd.append(partial(ops.gemm, col, w_in, sv, bias=b_in))
        hcur = adopt(self._produced(T(sv, need_grad=False)), key, "skip")
        layer = 0
        first = True  # nothing upstream of the first cross-attention needs a gradient
        cin = boc[0]
        for i, cout in enumerate(boc):
            h, wd = H >> i, W >> i
            for j in range(cfg.layers_per_block):
                name = f"down_blocks.{i}.resnets.{j}."
                key, sv = next_skip()
                if cfg.down_has_attn[i]:
                    r_out = self._resnet(hcur, cin if j == 0 else cout, cout, name, w, None, h, wd, need_dx=not first)
                    hcur = self._transformer(r_out, cout, cfg.num_heads[i], f"down_blocks.{i}.attentions.{j}.", w,
                                             layer, sv, h, wd, need_dx=not first)
                    layer += 1
                else:
                    hcur = self._resnet(hcur, cin if j == 0 else cout, cout, name, w, sv, h, wd, need_dx=not first)
                adopt(hcur, key, "skip")
                first = False
            if i < nlev - 1:
                key, sv = next_skip()
                hcur = self._downsample(hcur, cout, f"down_blocks.{i}.downsamplers.0.conv.", w, sv, h, wd)
                adopt(hcur, key, "skip")
            cin = cout
        assert nskip[0] == len(skip_targets)
        # mid block
        lvl = nlev - 1
        h, wd = H >> lvl, W >> lvl
        cm = boc[-1]
        hcur = self._resnet(hcur, cm, cm, "mid_block.resnets.0.", w, None, h, wd)
        hcur = self._transformer(hcur, cm, cfg.num_heads[-1], "mid_block.attentions.0.", w, layer, None, h, wd)
        layer += 1
        hcur = self._resnet(hcur, cm, cm, "mid_block.resnets.1.", w, h_slot((0, 0)), h, wd)
        adopt(hcur, (0, 0), "h")
        # up blocks
        up_has_attn = tuple(reversed(cfg.down_has_attn))
        up_heads = tuple(reversed(cfg.num_heads))
        for i in range(nlev):
            lvl = nlev - 1 - i
            h, wd = H >> lvl, W >> lvl
            for j in range(nres):
                rin, skip, out_c = sc.up_block_channels(cfg, i, j)
                nxt = (i, j + 1) if j < nres - 1 else None  # stage output feeds the next resnet of this block?
                nxt_view = h_slot(nxt) if nxt is not None else None
                name = f"up_blocks.{i}.resnets.{j}."
                if up_has_attn[i]:
                    r_out = self._resnet(catT[(i, j)], rin + skip, out_c, name, w, None, h, wd)
                    hcur = self._transformer(r_out, out_c, up_heads[i], f"up_blocks.{i}.attentions.{j}.", w, layer,
                                             nxt_view, h, wd)
                    layer += 1
                else:
                    hcur = self._resnet(catT[(i, j)], rin + skip, out_c, name, w, nxt_view, h, wd)
                if nxt is not None:
                    adopt(hcur, nxt, "h")
            if i < nlev - 1:
                out_c = tuple(reversed(boc))[i]
                hcur = self._upsample(hcur, out_c, f"up_blocks.{i}.upsamplers.0.conv.", w, h_slot((i + 1, 0)), h, wd)
                adopt(hcur, (i + 1, 0), "h")
        assert layer == self.nl
        # output head
        c0 = boc[0]
        n, gn = self._gn(hcur, "conv_norm_out", w, cfg.norm_eps, True)
        w_o = self._w16(packing.conv3x3_fwd(w["conv_out.weight"]))
        b_o = self._w32(w["conv_out.bias"])
        self.fwd.append(partial(ops.gemm, n, w_o, self.pred, bias=b_o, M=M0,
                                conv=self._conv_desc(H, W, c0, H, W, 1, 1, 0, c0)))
        rec = dict(kind="head", x=hcur, gn=gn)
        if self.need_backward:
            wt = w["conv_out.weight"].flip(2, 3).transpose(0, 1).contiguous()  # [Ci][Co][3][3], taps flipped
            rec["wd_"] = self._w16(packing.pad_rows(packing.conv3x3_fwd(wt), 64))
        self.tape.append(rec)

    def _head_bwd(self, r):
        x = r["x"]
        M0 = self.B * self.H * self.W
        col = self._tmp("hcol", M0, 64)
        dp = self.dpred
        ld = dp.stride(0)
        self.bwd.append(partial(ops.im2col3x3_small, dp, col, self.B, self.cfg.out_channels, self.H, self.W, self.H,
                                self.W, 1, 1, 1, (self.H * self.W * ld, 1, self.W * ld, ld)))
        dn = self._tmp("dA", M0, x.cols)
        self.bwd.append(partial(ops.gemm, col, r["wd_"], dn))
        self._contrib(x, self._gn_bwd_fn(r["gn"], dn))

    def _build_backward(self):
        # skip tensors: their gradient slices live inside the dcat buffers, which the up-path
        # resnet backward fills first (marking them written through the children lists).
        for r in reversed(self.tape):
            k = r["kind"]
            if k == "head":
                self._head_bwd(r)
            elif k == "resnet":
                self._resnet_bwd(r)
            elif k == "transformer":
                self._transformer_bwd(r)
            elif k == "down":
                self._downsample_bwd(r)
            elif k == "up":
                self._upsample_bwd(r)

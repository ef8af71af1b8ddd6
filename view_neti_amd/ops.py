"""Thin torch-tensor front-ends over the C ABI (include/vneti.h).

torch is used only for device memory and the current HIP stream; every function below ends in
exactly one (or, for GroupNorm, three) hand-written HIP kernel launches from libvneti_hip.so.
All matrices are 2-D views `[rows, cols]` whose last stride is 1; the row stride may exceed
`cols` (views into wider buffers).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _l

ACT_NONE, ACT_SILU, ACT_QUICK_GELU, ACT_GELU = 0, 1, 2, 3


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    if t is None:
        return 0
    assert t.stride(-1) == 1, "last dim must be contiguous"
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


_default_ws = None


def set_default_gemm_workspace(t):
    """f32 scratch used by split-K GEMMs when the caller passes none (single stream only)."""
    global _default_ws
    _default_ws = t


def gemm(A, B, out, *, bias=None, rowadd=None, rows_per_group=1, resid=None, alpha=1.0, act=0,
         tile_hint=0, batch=0, strideA=0, strideB=0, strideC=0, M=None, N=None, K=None, conv=None,
         lda=None, ldc=None, workspace=None, split_k=0, gate=None, gate_act=0, out2=None, act2=0,
         gn_sums=None, gn_hw=0, gn_groups=0, gn_slots=0, geglu=0):
    """out[M,N] = epi(alpha * A[M,K] @ B[N,K]^T).  `conv` = dict(mode, Hi, Wi, Ci, Ho, Wo, stride,
    pad_t, pad_l, ups, ldx) turns A into an implicit im2col view of an NHWC image."""
    d = _l.GemmDesc()
    d.A, d.B, d.C = _p(A), _p(B), _p(out)
    d.ldb = _ld(B)
    d.ldc = ldc if ldc is not None else _ld(out)
    d.N = N if N is not None else B.shape[-2]
    d.K = K if K is not None else B.shape[-1]
    if conv is None:
        d.M = M if M is not None else A.shape[-2]
        d.lda = lda if lda is not None else _ld(A)
        d.conv_mode = 0
    else:
        d.M = M
        d.lda = 0
        d.conv_mode = conv["mode"]
        for k in ("Hi", "Wi", "Ci", "Ho", "Wo", "stride", "pad_t", "pad_l", "ups", "ldx"):
            setattr(d, k, int(conv.get(k, 0)))
        d.conv_korder = int(conv.get("korder", 0))
    d.batch = batch
    d.strideA, d.strideB, d.strideC = strideA, strideB, strideC
    d.bias = _p(bias)
    d.rowadd = _p(rowadd)
    d.ld_rowadd = _ld(rowadd) if rowadd is not None else 0
    d.rows_per_group = rows_per_group
    d.resid = _p(resid)
    d.ldr = _ld(resid) if resid is not None else 0
    d.alpha = alpha
    d.act = act
    d.out_f32 = 1 if out.dtype == torch.float32 else 0
    d.tile_hint = tile_hint
    if gate is not None:  # out *= act'(gate): activation backward fused into the dgrad GEMM
        d.gate_src, d.ld_gate, d.gate_act = _p(gate), _ld(gate), gate_act
    if out2 is not None:  # out2 = act2(out)
        d.C2, d.ldc2, d.act2 = _p(out2), _ld(out2), act2
    if gn_sums is not None:  # GroupNorm (sum, sumsq) of the output accumulated by the epilogue
        d.gn_sums, d.gn_hw, d.gn_groups, d.gn_slots = _p(gn_sums), gn_hw, gn_groups, gn_slots
        d.gn_cpg = d.N // gn_groups
    d.geglu = geglu  # 1: out2 = h * gelu(g) of the interleaved out; 2: out[M, 2N] = GEGLU backward against `gate`
    ws = workspace if workspace is not None else _default_ws
    if ws is not None:
        d.workspace = ws.data_ptr()
        d.workspace_bytes = ws.numel() * ws.element_size()
    d.split_k = split_k if ws is not None else 1
    rc = _l.load().vneti_gemm_f16(C.byref(d), stream())
    _l.check(rc, "gemm_f16")


def im2col3x3_small(x, out, Bn, Cc, Hi, Wi, Ho, Wo, stride, pad_t, pad_l, strides):
    sb, sc, sy, sx = strides
    _l.call("im2col3x3_small", _p(x), 1 if x.dtype == torch.float32 else 0, sb, sc, sy, sx, _p(out),
            Bn, Cc, Hi, Wi, Ho, Wo, stride, pad_t, pad_l, stream())


def conv3x3_in(x, w_packed, bias, out, Bn, Cc, H, W, strides, gn_sums=None, gn_hw=0, gn_groups=0, gn_slots=0):
    """3x3 / stride 1 / pad 1 conv of a <= 3-channel image straight from the pixels (csrc/conv_in.hip);
    `w_packed` from packing.conv_in_direct; gn_* as in `gemm` (gn_hw must be H * W)"""
    sb, sc, sy, sx = strides
    assert gn_sums is None or gn_hw == H * W
    _l.call("conv3x3_in", _p(x), 1 if x.dtype == torch.float32 else 0, sb, sc, sy, sx, _p(w_packed), _p(bias), _p(out),
            _ld(out), Bn, Cc, H, W, w_packed.shape[0], _p(gn_sums), gn_groups, gn_slots, stream())


def transpose(inp, out, rows, cols, batch, ld_in, stride_in, ld_out, stride_out):
    _l.call("transpose_f16", _p(inp), ld_in, stride_in, _p(out), ld_out, stride_out, rows, cols, batch, stream())


def transpose_multi(items):
    """items: up to 4 tuples with the arguments of `transpose`; one launch for all of them."""
    import ctypes
    arr = (_l.TransposeDesc * len(items))()
    for d, (inp, out, rows, cols, batch, ld_in, stride_in, ld_out, stride_out) in zip(arr, items):
        d.inp, d.ld_in, d.stride_in = _p(inp), ld_in, stride_in
        d.out, d.ld_out, d.stride_out = _p(out), ld_out, stride_out
        d.rows, d.cols, d.batch = rows, cols, batch
    _l.call("transpose_f16_multi", ctypes.addressof(arr), len(items), stream())


def groupnorm_ws_floats(Bn, HW, Cc, G):
    n = _l.load().vneti_groupnorm_ws_floats(Bn, HW, Cc, G)
    if n < 0:
        raise RuntimeError(f"groupnorm: unsupported shape {(Bn, HW, Cc, G)}")
    return int(n)


def groupnorm_fwd(x, y, gamma, beta, mean, rstd, ws, Bn, HW, Cc, G, eps, silu):
    _l.call("groupnorm_fwd", _p(x), _ld(x), _p(y), _ld(y), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(ws),
            Bn, HW, Cc, G, eps, 1 if silu else 0, stream())


def groupnorm_fwd_sums(x, y, gamma, beta, sums, slots, mean, rstd, Bn, HW, Cc, G, eps, silu):
    _l.call("groupnorm_fwd_sums", _p(x), _ld(x), _p(y), _ld(y), _p(gamma), _p(beta), _p(sums), slots, _p(mean), _p(rstd),
            Bn, HW, Cc, G, eps, 1 if silu else 0, stream())


def groupnorm_fwd_2l(x, y, gamma, beta, sums, slots, mean, rstd, Bn, HW, Cc, G, eps, silu):
    """GroupNorm forward in two launches (statistics into the caller-zeroed `sums` [Bn, slots, G, 2], apply finishes
    them); small tensors take the one-launch kernel"""
    _l.call("groupnorm_fwd_2l", _p(x), _ld(x), _p(y), _ld(y), _p(gamma), _p(beta), _p(sums), slots, _p(mean), _p(rstd),
            Bn, HW, Cc, G, eps, 1 if silu else 0, stream())


def groupnorm_bwd_2l(dy, x, gamma, beta, mean, rstd, dx, sums, slots, ws, Bn, HW, Cc, G, silu, accum=None):
    _l.call("groupnorm_bwd_2l", _p(dy), _ld(dy), _p(x), _ld(x), _p(gamma), _p(beta), _p(mean), _p(rstd),
            _p(dx), _ld(dx), _p(accum), _ld(accum) if accum is not None else 0, _p(sums), slots, _p(ws),
            Bn, HW, Cc, G, 1 if silu else 0, stream())


def groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dx, ws, Bn, HW, Cc, G, silu, accum=None):
    _l.call("groupnorm_bwd", _p(dy), _ld(dy), _p(x), _ld(x), _p(gamma), _p(beta), _p(mean), _p(rstd),
            _p(dx), _ld(dx), _p(accum), _ld(accum) if accum is not None else 0, _p(ws),
            Bn, HW, Cc, G, 1 if silu else 0, stream())


def layernorm_fwd(x, y, gamma, beta, mean, rstd, eps):
    rows, Cc = x.shape
    _l.call("layernorm_fwd", _p(x), 1 if x.dtype == torch.float32 else 0, _ld(x), _p(y), _ld(y), _p(gamma),
            _p(beta), _p(mean), _p(rstd), rows, Cc, eps, stream())


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, accum=None, f16_copy=None):
    rows, Cc = x.shape
    _l.call("layernorm_bwd", _p(dy), 1 if dy.dtype == torch.float32 else 0, _ld(dy), _p(x),
            1 if x.dtype == torch.float32 else 0, _ld(x), _p(gamma), _p(mean), _p(rstd), _p(dx),
            1 if dx.dtype == torch.float32 else 0, _ld(dx), _p(accum),
            _ld(accum) if accum is not None else 0, _p(f16_copy), _ld(f16_copy) if f16_copy is not None else 0,
            rows, Cc, stream())


def attn_fwd(Q, K, V, O, lse, Bn, H, Nq, Nk, D, scale, causal):
    _l.call("attn_fwd", _p(Q), _ld(Q), _p(K), _ld(K), _p(V), _ld(V), _p(O), _ld(O), _p(lse), Bn, H, Nq, Nk, D,
            scale, 1 if causal else 0, stream())


def attn_bwd_delta(dO, O, delta, Bn, H, Nq, D):
    _l.call("attn_bwd_delta", _p(dO), _ld(dO), _p(O), _ld(O), _p(delta), Bn, H, Nq, D, stream())


def attn_bwd_dq(Q, K, V, dO, lse, delta, dQ, Bn, H, Nq, Nk, D, scale, causal, O=None):
    """O given: delta is computed inside the kernel and written to `delta` (no attn_bwd_delta launch)."""
    _l.call("attn_bwd_dq", _p(Q), _ld(Q), _p(K), _ld(K), _p(V), _ld(V), _p(dO), _ld(dO), _p(lse), _p(delta),
            _p(O), _ld(O) if O is not None else 0, _p(dQ), _ld(dQ), Bn, H, Nq, Nk, D, scale, 1 if causal else 0,
            stream())


def attn_bwd_dkv(Q, K, V, dO, lse, delta, dK, dV, Bn, H, Nq, Nk, D, scale, causal, workspace=None):
    ws = workspace if workspace is not None else _default_ws
    _l.call("attn_bwd_dkv", _p(Q), _ld(Q), _p(K), _ld(K), _p(V), _ld(V), _p(dO), _ld(dO), _p(lse), _p(delta),
            _p(dK), _ld(dK), _p(dV), _ld(dV), Bn, H, Nq, Nk, D, scale, 1 if causal else 0, _p(ws),
            ws.numel() if ws is not None else 0, stream())


def attn_bwd_small_ok(N, D):
    """sizes vneti_attn_bwd_small carries (the caller picks between it and attn_bwd_dq + attn_bwd_dkv)"""
    return N <= 96 and D == 64


def attn_bwd_small(Q, K, V, dO, O, lse, dQ, dK, dV, Bn, H, N, D, scale, causal):
    """dQ, dK, dV of a short self-attention (N <= 96, D = 64) in one launch"""
    _l.call("attn_bwd_small", _p(Q), _ld(Q), _p(K), _ld(K), _p(V), _ld(V), _p(dO), _ld(dO), _p(O), _ld(O), _p(lse),
            _p(dQ), _ld(dQ), _p(dK), _ld(dK), _p(dV), _ld(dV), Bn, H, N, D, scale, 1 if causal else 0, stream())


def softmax_rows(x, rows, cols):
    _l.call("softmax_rows_f16", _p(x), _ld(x), rows, cols, stream())


def add(a, b, out):
    rows, cols = out.shape
    _l.call("add_f16", _p(a), _ld(a), _p(b), _ld(b), _p(out), _ld(out), rows, cols, stream())


def geglu_fwd(p, out):
    rows, c4 = out.shape
    _l.call("geglu_fwd", _p(p), _ld(p), _p(out), _ld(out), rows, c4, stream())


def geglu_bwd(dy, p, dp):
    rows, c4 = dy.shape
    _l.call("geglu_bwd", _p(dy), _ld(dy), _p(p), _ld(p), _p(dp), _ld(dp), rows, c4, stream())


def act_fwd(x, y, act):
    _l.call("act_fwd_f16", _p(x), _p(y), x.numel(), act, stream())


def act_bwd(dy, x, dx, act):
    _l.call("act_bwd_f16", _p(dy), _p(x), _p(dx), x.numel(), act, stream())


def timestep_embedding(t, out):
    Bn, dim = out.shape
    _l.call("timestep_embedding", _p(t), _p(out), Bn, dim, stream())


def sum2x2(inp, out, Bn, H, W, Cc):
    _l.call("sum2x2_f16", _p(inp), _ld(inp), _p(out), _ld(out), Bn, H, W, Cc, stream())


def rng_fill_normal(out, state, stream_id):
    _l.call("rng_fill_normal", _p(out), out.numel(), _p(state), stream_id, stream())


def rng_fill_randint(out, high, state, stream_id):
    _l.call("rng_fill_randint", _p(out), out.numel(), high, _p(state), stream_id, stream())


def rng_advance(state):
    _l.call("rng_advance", _p(state), stream())


def sample_add_noise(moments, eps, noise, t, ac, scaling, vpred, latents, noisy, target, Bn, Lc, HW):
    _l.call("sample_add_noise", _p(moments), _ld(moments), _p(eps), _p(noise), _p(t), _p(ac), scaling,
            1 if vpred else 0, _p(latents), _p(noisy), _p(target), Bn, Lc, HW, stream())


def gn_sums_decode(sums: torch.Tensor) -> torch.Tensor:
    """[..., 4] int64 slot sums (S1.hi, S1.lo, S2.hi, S2.lo: the fixed-point accumulators of csrc/common.h) -> [..., 2]
    float64 (sum, sum of squares); slots are summed by the caller as integers first (`.sum(dim)` on the int64 tensor)."""
    import numpy as np
    a = sums.detach().cpu().numpy().astype(np.int64)
    out = np.empty(a.shape[:-1] + (2,), dtype=np.float64)
    for q in range(2):
        hi, lo = a[..., 2 * q].astype(object), a[..., 2 * q + 1].astype(object)
        # total * 2^40 = lo + k * 2^64 with k the integer that brings it next to hi * 2^36 (exact in Python integers)
        k = np.vectorize(lambda h, l: (h * 2 ** 36 - l + 2 ** 63) // 2 ** 64, otypes=[object])(hi, lo)
        out[..., q] = np.vectorize(lambda l, kk: float(l + kk * 2 ** 64) / 2.0 ** 40, otypes=[np.float64])(lo, k)
    return torch.from_numpy(out)


def latent_sample(moments, eps, scaling, latents, Bn, Lc, HW):
    _l.call("latent_sample", _p(moments), _ld(moments), _p(eps), scaling, _p(latents), Bn, Lc, HW, stream())


def add_noise(latents, noise, t, ac, vpred, noisy, target, Bn, Lc, HW):
    _l.call("add_noise", _p(latents), _p(noise), _p(t), _p(ac), 1 if vpred else 0, _p(noisy), _p(target), Bn, Lc, HW,
            stream())


def cfg_sampler_step(pred, x, m_prev, x_in, Bn, Lc, HW, guidance, alpha_t, sigma_t, cx, c0, c1, v_prediction):
    _l.call("cfg_sampler_step", _p(pred), _ld(pred), _p(x), _p(m_prev), _p(x_in), Bn, Lc, HW, guidance, alpha_t,
            sigma_t, cx, c0, c1, 1 if v_prediction else 0, stream())


def cfg_sampler_step_table(pred, x, m_prev, x_in, Bn, Lc, HW, guidance, coef_table, step, v_prediction):
    _l.call("cfg_sampler_step_table", _p(pred), _ld(pred), _p(x), _p(m_prev), _p(x_in), Bn, Lc, HW, guidance,
            _p(coef_table), _p(step), 1 if v_prediction else 0, stream())


def table_fill_i64(dst, table, step):
    _l.call("table_fill_i64", _p(dst), dst.numel(), _p(table), _p(step), stream())


def counter_advance(counter):
    _l.call("counter_advance", _p(counter), stream())


def conv1x1_nchw(x, W, bias, out, Bn, Ci, Co, HW, in_scale=1.0):
    _l.call("conv1x1_nchw_f32", _p(x), _p(W), _p(bias), _p(out), Bn, Ci, Co, HW, in_scale, stream())


def image_postprocess(img, out, n_pix, channels):
    _l.call("image_postprocess", _p(img), _ld(img), _p(out), n_pix, channels, stream())


def mse_loss_grad(pred, target, dpred, loss_sum, loss_scale, Bn, Lc, HW):
    _l.call("mse_loss_grad", _p(pred), _ld(pred), _p(target), _p(dpred), _ld(dpred), _p(loss_sum), _p(loss_scale),
            Bn, Lc, HW, stream())


OPT_CHECK, OPT_APPLY, OPT_FINISH, OPT_ALL = 1, 2, 4, 7


def adamw_flat(p, g, m, v, hyper, scaler, step, growth_interval=2000, phases=OPT_ALL):
    _l.call("adamw_flat", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _p(scaler), _p(step), growth_interval,
            phases, stream())


def adamw_segments(p, g, m, v, seg_len, n_seg, seg_step, active, hyper, scaler, step, growth_interval=2000,
                   phases=OPT_ALL):
    _l.call("adamw_segments", _p(p), _p(g), _p(m), _p(v), seg_len, n_seg, _p(seg_step), _p(active), active.numel(),
            _p(hyper), _p(scaler), _p(step), growth_interval, phases, stream())


def nested_dropout_mask(mask, nl, Bn, hidden, prob, state, stream_id):
    _l.call("nested_dropout_mask", _p(mask), nl, Bn, hidden, prob, _p(state), stream_id, stream())


def mapper_num_params(enc_dim, hidden, D, has_bypass=True):
    return _l.call_ll("mapper_num_params", enc_dim, hidden, D, 1 if has_bypass else 0)


def mapper_save_floats(R, enc_dim, hidden):
    return _l.call_ll("mapper_save_floats", R, enc_dim, hidden)


def mapper_rowgrad_floats(R, hidden, D, has_bypass=True):
    return _l.call_ll("mapper_rowgrad_floats", R, hidden, D, 1 if has_bypass else 0)


def mapper_fwd(params, data, w_enc, hidden_mask, norm_scale, word, bypass, save, R, enc_dim, hidden, D, has_bypass,
               slot=None, slot_stride=0, enc_in=None):
    _l.call("mapper_fwd", _p(params), _p(slot), slot_stride, _p(data), data.shape[1] if data is not None else 0,
            _p(w_enc), _p(hidden_mask),
            norm_scale if norm_scale is not None else -1.0, _p(word), _p(bypass), _p(save), R, enc_dim, hidden, D,
            1 if has_bypass else 0, _p(enc_in), stream())


def mapper_bwd(params, hidden_mask, norm_scale, word, dword_src, dword_rows, ld_src, dbypass, save, rowgrads, grads,
               accumulate, R, enc_dim, hidden, D, has_bypass, slot=None, slot_stride=0, denc=None):
    _l.call("mapper_bwd", _p(params), _p(slot), slot_stride, _p(hidden_mask), norm_scale if norm_scale is not None else -1.0, _p(word),
            _p(dword_src), _p(dword_rows), ld_src, _p(dbypass), _p(save), _p(rowgrads), _p(grads),
            1 if accumulate else 0, R, enc_dim, hidden, D, 1 if has_bypass else 0, _p(denc), stream())


def mapper_legacy_input_params(enc_dim, pe_dim):
    return _l.call_ll("mapper_legacy_input_params", enc_dim, pe_dim)


def mapper_legacy_input_fwd(params_in, timesteps, w_pe, enc_out, nl, Bn, enc_dim, pe_dim, slot=None, slot_stride=0):
    _l.call("mapper_legacy_input_fwd", _p(params_in), _p(slot), slot_stride, _p(timesteps), _p(w_pe), _p(enc_out), nl, Bn,
            enc_dim, pe_dim, stream())


def mapper_legacy_input_bwd(timesteps, w_pe, denc, grads_in, accumulate, nl, Bn, enc_dim, pe_dim, slot=None, slot_stride=0):
    _l.call("mapper_legacy_input_bwd", _p(timesteps), _p(w_pe), _p(denc), _p(grads_in), _p(slot), slot_stride,
            1 if accumulate else 0, nl, Bn, enc_dim, pe_dim, stream())


def text_embed(tok_emb, pos_emb, ids, pos_obj, word_obj, pos_view, word_view, X, nl, Bn, L, D):
    _l.call("text_embed", _p(tok_emb), _p(pos_emb), _p(ids), _p(pos_obj), _p(word_obj), _p(pos_view), _p(word_view),
            _p(X), nl, Bn, L, D, stream())


def text_final_fwd(last, gamma, beta, eps, pos_obj, byp_obj, alpha_obj, pos_view, byp_view, alpha_view, ctx_k, ctx_v,
                   nl, Bn, L, D, unconstrained_obj=False, unconstrained_view=False, norm_terms=None):
    _l.call("text_final_fwd", _p(last), _p(gamma), _p(beta), eps, _p(pos_obj), _p(byp_obj), alpha_obj,
            1 if unconstrained_obj else 0, _p(pos_view), _p(byp_view), alpha_view, 1 if unconstrained_view else 0,
            _p(norm_terms), _p(ctx_k), _p(ctx_v), nl, Bn, L, D, stream())


def text_final_bwd(last, gamma, eps, pos_obj, byp_obj, alpha_obj, dbyp_obj, pos_view, byp_view, alpha_view, dbyp_view,
                   dctx_k, dctx_v, dX, nl, Bn, L, D, unconstrained_obj=False, unconstrained_view=False,
                   norm_terms=None):
    _l.call("text_final_bwd", _p(last), _p(gamma), eps, _p(pos_obj), _p(byp_obj), alpha_obj,
            1 if unconstrained_obj else 0, _p(dbyp_obj), _p(pos_view), _p(byp_view), alpha_view,
            1 if unconstrained_view else 0, _p(dbyp_view), _p(norm_terms), _p(dctx_k), _p(dctx_v), _p(dX), nl, Bn, L,
            D, stream())


def cast_f32_f16(x, y):
    _l.call("cast_f32_f16", _p(x), _p(y), x.numel(), stream())


def mapper_inputs(timesteps, view_params, data, nl, Bn):
    nv = 0 if view_params is None else view_params.shape[1]
    _l.call("mapper_inputs", _p(timesteps), _p(view_params), nv, _p(data), nl, Bn, stream())


def gemm_select_split(M, N, K, batch, tile_hint, workspace_bytes):
    fn = _l.load().vneti_gemm_select_split
    fn.argtypes = _l.INT_FUNCS["gemm_select_split"]
    return int(fn(M, N, K, batch, tile_hint, workspace_bytes))


def gemm_select_tile(M, N, batch=1):
    fn = _l.load().vneti_gemm_select_tile
    fn.argtypes = _l.INT_FUNCS["gemm_select_tile"]
    return int(fn(M, N, batch))


# ------------------------------------------------------------------ device-side input pipeline (csrc/image.hip)
def img_resample_ksize(in_size, out_size, filt):
    fn = _l.load().vneti_img_resample_ksize
    fn.argtypes = _l.INT_FUNCS["img_resample_ksize"]
    return int(fn(in_size, out_size, filt))


def img_resample_coeffs(in_size, out_size, filt, bounds, kk):
    _l.call("img_resample_coeffs", in_size, out_size, filt, _p(bounds), _p(kk), stream())


def img_resample_pass(inp, in_w, out, out_h, out_w, bounds, kk, ksize, horizontal):
    _l.call("img_resample_pass", _p(inp), in_w, _p(out), out_h, out_w, _p(bounds), _p(kk), ksize, 1 if horizontal else 0,
            stream())


def img_crop(inp, in_w, top, left, out, h, w, flip=False):
    _l.call("img_crop", _p(inp), in_w, top, left, _p(out), h, w, 1 if flip else 0, stream())


def img_enhance(img, h, w, mode, alpha, scratch8):
    _l.call("img_enhance", _p(img), h, w, mode, float(alpha), _p(scratch8), stream())


def img_hue(img, h, w, shift):
    _l.call("img_hue", _p(img), h, w, int(shift), stream())


def img_blur5(inp, out, tmp, h, w, k5):
    import ctypes
    arr = (ctypes.c_float * 5)(*[float(x) for x in k5])
    _l.call("img_blur5", _p(inp), _p(out), _p(tmp), h, w, ctypes.addressof(arr), stream())


def img_affine_nearest(inp, out, h, w, a6, fill):
    import ctypes
    arr = (ctypes.c_int * 6)(*[int(x) for x in a6])
    _l.call("img_affine_nearest", _p(inp), _p(out), h, w, ctypes.addressof(arr), int(fill), stream())


def img_to_f32_chw(img, out, h, w):
    _l.call("img_to_f32_chw", _p(img), _p(out), h, w, stream())

"""Seam A of INTEGRATION.md as real code: a drop-in object for the diffusers attention-processor protocol.

The reference installs `XTIAttenProc()` on every attention module of the UNet (`unet.set_attn_processor`,
training/coach.py:679-680; models/xti_attention_processor.py:9-57).  `HipXTIAttenProc` has the same
`__call__(attn, hidden_states, encoder_hidden_states=None, attention_mask=None)` signature, the same dict dispatch
(K from `CONTEXT_TENSOR_i`, V from `CONTEXT_TENSOR_BYPASS_i`) and the same `this_idx` counter, and computes
to_q / to_k / to_v, the fused softmax attention and to_out with the HIP kernels of libvneti_hip.so through the C ABI
(`vneti_gemm_f16`, `vneti_attn_fwd`, `vneti_attn_bwd_dq`, `vneti_attn_bwd_dkv`).  Gradients flow to the hidden
states and to the two context tensors through a `torch.autograd.Function` (the attention weights are frozen in the
reference, coach.py:642-653, so no weight gradients are produced).

`attn` is duck-typed exactly as the reference uses it: `.heads`, `.to_q/.to_k/.to_v` (bias-free Linears),
`.to_out[0]` (Linear with bias), `.to_out[1]` (Dropout), `.cross_attention_norm`, `.prepare_attention_mask`.
Constraints of the kernels: f16 CUDA tensors, head dim in {40, 64, 80, 160}, channel counts multiples of 64 (8 for
the output widths).  There is no fallback: without the extension the import of `view_neti_amd.lib` raises.
"""
from __future__ import annotations

from typing import Dict, Optional, Union

import torch

from .. import lib, ops

_HEAD_DIMS = (40, 64, 80, 160)


def _w16(lin, transpose=False):
    w = lin.weight.detach()
    w = w.t() if transpose else w
    return w.to(lib.act_dtype()).contiguous()


class _Packed:
    """f16 forward / dgrad (pre-transposed) copies of one attention module's frozen weights"""

    def __init__(self, attn):
        self.key = tuple(int(l.weight._version) for l in (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0]))
        self.wq, self.wk, self.wv, self.wo = (_w16(l) for l in (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0]))
        self.wq_t, self.wk_t, self.wv_t, self.wo_t = (_w16(l, True) for l in
                                                      (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0]))
        b = attn.to_out[0].bias
        self.bo = None if b is None else b.detach().float().contiguous()
        for lin in (attn.to_q, attn.to_k, attn.to_v):
            if getattr(lin, "bias", None) is not None:
                raise NotImplementedError("to_q/to_k/to_v carry a bias: not the SD CrossAttention layout")


def _packed(attn) -> _Packed:
    p = getattr(attn, "_vneti_packed", None)
    key = tuple(int(l.weight._version) for l in (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0]))
    if p is None or p.key != key or p.wq.device != attn.to_q.weight.device:
        p = _Packed(attn)
        try:
            attn._vneti_packed = p
        except Exception:  # objects that refuse new attributes: repack per call
            pass
    return p


def _mm(A, W, bias=None):
    out = torch.empty((A.shape[0], W.shape[0]), dtype=lib.act_dtype(), device=A.device)
    ops.gemm(A, W, out, bias=bias)
    return out


class _XTIAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, ctx_k, ctx_v, pk: _Packed, heads: int):
        B, N, C = hidden.shape
        Nk = ctx_k.shape[1]
        D = C // heads
        h2 = hidden.reshape(B * N, C)
        k_src = ctx_k.reshape(B * Nk, ctx_k.shape[-1])
        v_src = ctx_v.reshape(B * Nk, ctx_v.shape[-1])
        q, k, v = _mm(h2, pk.wq), _mm(k_src, pk.wk), _mm(v_src, pk.wv)
        o = torch.empty_like(q)
        lse = torch.empty((B, heads, N), dtype=torch.float32, device=q.device)
        scale = D ** -0.5
        ops.attn_fwd(q, k, v, o, lse, B, heads, N, Nk, D, scale, False)
        out = _mm(o, pk.wo, pk.bo)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.pk, ctx.dims = pk, (B, N, Nk, C, heads, D, scale, ctx_k.shape[-1], ctx_v.shape[-1])
        ctx.same_kv = ctx_k.data_ptr() == ctx_v.data_ptr()
        ctx.self_attn = ctx_k.data_ptr() == hidden.data_ptr()
        return out.view(B, N, C)

    @staticmethod
    def backward(ctx, dout):
        q, k, v, o, lse = ctx.saved_tensors
        pk = ctx.pk
        B, N, Nk, C, heads, D, scale, Dk, Dv = ctx.dims
        dout2 = dout.reshape(B * N, C).to(lib.act_dtype()).contiguous()
        do = _mm(dout2, pk.wo_t)
        delta = torch.empty((B * heads, N), dtype=torch.float32, device=q.device)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops.attn_bwd_dq(q, k, v, do, lse, delta, dq, B, heads, N, Nk, D, scale, False, O=o)  # also publishes delta
        ops.attn_bwd_dkv(q, k, v, do, lse, delta, dk, dv, B, heads, N, Nk, D, scale, False)
        dh = _mm(dq, pk.wq_t).view(B, N, C)
        dck = _mm(dk, pk.wk_t).view(B, Nk, Dk)
        dcv = _mm(dv, pk.wv_t).view(B, Nk, Dv)
        return dh, dck, dcv, None, None


class HipXTIAttenProc:
    """`unet.set_attn_processor(HipXTIAttenProc())` — the MI355X replacement of XTIAttenProc."""

    def __call__(self, attn, hidden_states: torch.Tensor,
                 encoder_hidden_states: Optional[Union[torch.Tensor, Dict[str, torch.Tensor]]] = None,
                 attention_mask: Optional[torch.Tensor] = None):
        _ehs, _ehs_bypass = None, None
        if encoder_hidden_states is not None:
            if isinstance(encoder_hidden_states, dict):                       # xti_attention_processor.py:16-22
                this_idx = encoder_hidden_states["this_idx"]
                _ehs = encoder_hidden_states[f"CONTEXT_TENSOR_{this_idx}"]
                if f"CONTEXT_TENSOR_BYPASS_{this_idx}" in encoder_hidden_states:
                    _ehs_bypass = encoder_hidden_states[f"CONTEXT_TENSOR_BYPASS_{this_idx}"]
                encoder_hidden_states["this_idx"] += 1
                encoder_hidden_states["this_idx"] %= 16
            else:
                _ehs = encoder_hidden_states
        if attention_mask is not None:
            raise NotImplementedError("attention_mask: Stable Diffusion never passes one to these modules")
        if _ehs is not None and getattr(attn, "cross_attention_norm", False):
            raise NotImplementedError("cross_attention_norm (xti_attention_processor.py:34-36) is off for every SD model")
        if hidden_states.dtype != lib.act_dtype() or not hidden_states.is_cuda:
            raise TypeError("HipXTIAttenProc runs the fp16 path on the GPU (coach.py:792-794 hard-casts the UNet)")
        B, N, C = hidden_states.shape
        heads = attn.heads
        if C % heads or C // heads not in _HEAD_DIMS:
            raise NotImplementedError(f"head dim {C // heads if C % heads == 0 else C / heads} not in {_HEAD_DIMS}")
        hidden_states = hidden_states.contiguous()
        k_src = hidden_states if _ehs is None else _ehs.to(lib.act_dtype()).contiguous()
        v_src = k_src if _ehs_bypass is None else _ehs_bypass.to(lib.act_dtype()).contiguous()
        out = _XTIAttentionFn.apply(hidden_states, k_src, v_src, _packed(attn), heads)
        return attn.to_out[1](out)                                            # dropout(p=0), :55

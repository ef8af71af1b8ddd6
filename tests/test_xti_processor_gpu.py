"""Seam A: `HipXTIAttenProc` — the drop-in for the reference's attention-processor object
(models/xti_attention_processor.py:9-57, installed at training/coach.py:679-680) — against G5b, the outputs and
autograd gradients of the REAL XTIAttenProc at head dims the HIP kernels implement (oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


class _Attn(torch.nn.Module):
    """what the processor sees of diffusers' CrossAttention (duck-typed exactly like the fixture generator)"""

    def __init__(self, wq, wk, wv, wo, bo, heads):
        super().__init__()
        C, Dk = wq.shape[0], wk.shape[1]
        self.heads = heads
        self.to_q = torch.nn.Linear(C, C, bias=False)
        self.to_k = torch.nn.Linear(Dk, C, bias=False)
        self.to_v = torch.nn.Linear(Dk, C, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Dropout(0.0)])
        self.cross_attention_norm = False
        for lin, w in ((self.to_q, wq), (self.to_k, wk), (self.to_v, wv), (self.to_out[0], wo)):
            lin.weight.data = w.clone()
        self.to_out[0].bias.data = bo.clone()
        self.requires_grad_(False)  # frozen, coach.py:642-653

    def prepare_attention_mask(self, mask, n, b):
        return mask


def _rel(a, b):
    a, b = a.float().cpu(), torch.as_tensor(b).float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item(), (a - b).abs().max().item(), b.abs().max().item()


def _check(a, b, rel=3e-3, what=""):
    r, mx, ref = _rel(a, b)
    assert r < rel and mx < 8e-3 * max(1.0, ref), f"{what}: rel {r:.2e} max-abs {mx:.2e} (ref max {ref:.2e})"


@pytest.mark.parametrize("tag", ["d40", "d64"])
def test_hip_xti_processor_matches_reference_processor(tag):
    from view_neti_amd.compat.xti_attention_processor import HipXTIAttenProc
    f = np.load(os.path.join(G, f"g5b_xti_attention_{tag}.npz"))
    dev = "cuda"
    T = lambda k: torch.from_numpy(f[k]).to(dev)
    heads = int(f["heads"])
    cross = _Attn(T("wq"), T("wk"), T("wv"), T("wo"), T("bo"), heads).to(dev).half()
    self_ = _Attn(T("sq"), T("sk"), T("sv"), T("so"), T("sbo"), heads).to(dev).half()
    proc = HipXTIAttenProc()
    gy = T("gy")

    def leaf(k):
        return T(k).clone().requires_grad_(True)

    # encoder_hidden_states=None: self-attention, counter untouched
    hs = leaf("hs")
    y = proc(self_, hs, None)
    assert y.shape == hs.shape and y.dtype == torch.float16
    _check(y, f["y_self"], what="self out")
    (g,) = torch.autograd.grad((y.float() * gy.float()).sum(), hs)
    _check(g, f["g_self"], what="self d hidden")
    # a plain tensor context: K and V from the same tensor
    hs, c0 = leaf("hs"), leaf("ctx0")
    y = proc(cross, hs, c0)
    _check(y, f["y_tensor"], what="tensor ctx out")
    g_hs, g_c = torch.autograd.grad((y.float() * gy.float()).sum(), [hs, c0])
    _check(g_hs, f["g_tensor_hs"], what="tensor ctx d hidden")
    _check(g_c, f["g_tensor_ctx"], what="tensor ctx d ctx")
    # the XTI dict: K from CONTEXT_TENSOR_i, V from CONTEXT_TENSOR_BYPASS_i, counter 15 -> 0 -> 1
    ctx = {"this_idx": 15}
    for i in (15, 0):
        ctx[f"CONTEXT_TENSOR_{i}"] = leaf(f"ctx{i}")
        ctx[f"CONTEXT_TENSOR_BYPASS_{i}"] = leaf(f"ctxb{i}")
    seq = []
    for call in range(2):
        i = ctx["this_idx"]
        hs = leaf("hs")
        y = proc(cross, hs, ctx)
        seq.append(ctx["this_idx"])
        _check(y, f[f"y{call}"], what=f"dict call {call} out")
        g_hs, g_k, g_v = torch.autograd.grad((y.float() * gy.float()).sum(),
                                             [hs, ctx[f"CONTEXT_TENSOR_{i}"], ctx[f"CONTEXT_TENSOR_BYPASS_{i}"]])
        _check(g_hs, f[f"g{call}_hs"], what=f"dict call {call} d hidden")
        _check(g_k, f[f"g{call}_ctx"], what=f"dict call {call} d CONTEXT_TENSOR (keys)")
        _check(g_v, f[f"g{call}_ctxb"], what=f"dict call {call} d CONTEXT_TENSOR_BYPASS (values)")
    assert seq == list(f["seq"]) == [0, 1]


def test_hip_xti_processor_refuses_what_it_does_not_implement():
    from view_neti_amd.compat.xti_attention_processor import HipXTIAttenProc
    f = np.load(os.path.join(G, "g5b_xti_attention_d64.npz"))
    T = lambda k: torch.from_numpy(f[k]).cuda()
    attn = _Attn(T("wq"), T("wk"), T("wv"), T("wo"), T("bo"), int(f["heads"])).cuda().half()
    proc = HipXTIAttenProc()
    with pytest.raises(NotImplementedError):
        proc(attn, T("hs"), T("ctx0"), attention_mask=torch.ones(1, device="cuda"))
    with pytest.raises(TypeError):
        proc(attn, T("hs").float(), T("ctx0"))
    attn.heads = 16  # head dim 8
    with pytest.raises(NotImplementedError):
        proc(attn, T("hs"), T("ctx0"))

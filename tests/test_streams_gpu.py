"""CU-partitioned streams behind the C ABI (include/vneti.h: vneti_stream_create_cu_mask / _get_cu_mask / _destroy) — the
measurement aid of round 6's pipelined-VAE probe (tools/lab/cu_mask_probe.py): the runtime holds the mask it was given, work
launched on the stream (directly and as a captured graph) computes what it computes anywhere else, and a mask confined to a
quarter of the chip really slows a chip-filling launch down."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cu_mask_stream_round_trip_and_results():
    from view_neti_amd import ops, streams as S
    m = S.per_xcd_mask(8)
    assert sum(bin(w).count("1") for w in m) == 64 and m == [0xFFFFFFFF, 0xFFFFFFFF, 0, 0, 0, 0, 0, 0]
    assert sum(bin(w).count("1") for w in S.whole_xcd_mask(3)) == 96
    with pytest.raises(ValueError):
        S.per_xcd_mask(33)
    s = S.CUMaskStream(m)
    assert s.n_cus == 64 and s.runtime_mask() == m
    a = torch.randn(512, 256, device="cuda").half()
    b = torch.randn(512, 256, device="cuda").half()
    ref = torch.zeros_like(a)
    ops.add(a, b, ref)
    out = torch.zeros_like(a)
    s.stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s.stream):
        ops.add(a, b, out)
    torch.cuda.current_stream().wait_stream(s.stream)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    s.close()
    s.close()  # idempotent


def test_cu_mask_binds_a_graph_launch():
    """a linear hipGraph replayed on a masked stream runs on that stream's queue: a 64-CU mask makes a chip-filling GEMM
    measurably slower than the unmasked replay of the same graph (profiles/r06_cu_mask_probe.txt: 3x on the VAE encoder)"""
    from view_neti_amd import ops, streams as S
    M = N = K = 4096
    A = torch.randn(M, K, device="cuda").half()
    B = torch.randn(N, K, device="cuda").half()
    Cm = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    ops.set_default_gemm_workspace(torch.empty(16 * 2 ** 20, dtype=torch.float32, device="cuda"))
    run = lambda: [ops.gemm(A, B, Cm) for _ in range(8)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        run()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            run()
    torch.cuda.synchronize()
    ref = Cm.clone()

    def timed(stream):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            g.replay()
            s.record()
            for _ in range(3):
                g.replay()
            e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e)

    masked = S.CUMaskStream(S.per_xcd_mask(8))
    t_full, t_masked = timed(torch.cuda.Stream()), timed(masked.stream)
    assert torch.equal(Cm, ref)
    print(f"[cu mask] 8 x 4096^3 GEMM graph: unmasked {t_full:.2f} ms, 64-CU mask {t_masked:.2f} ms ({t_masked / t_full:.2f}x)")
    assert t_masked > 1.5 * t_full, "the CU mask did not bind the graph launch"
    masked.close()

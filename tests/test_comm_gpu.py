"""The data-parallel exchange behind the C ABI (include/vneti.h: vneti_comm_unique_id / vneti_comm_init /
vneti_allreduce_flat / vneti_comm_destroy, csrc/comm.hip) on the one GPU a test box has: a world-size-1 RCCL communicator
is a real communicator (id generation, ncclCommInitRank, ncclAllReduce on the caller's stream) whose sum over ranks is the
identity — so the whole call path, its stream ordering and its hipGraph capture are exercised; the N > 1 arithmetic of the
step is covered by tests/test_dp_gpu.py (gloo) and tests/test_dp_gloo.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rccl_comm_world1_allreduce_and_graph_capture():
    from view_neti_amd import lib, parallel
    comm = parallel.RcclComm(0, 1, exchange=lambda b: b)
    x = torch.randn(108416, device="cuda")  # the mapper bucket of SD-1.5 (D = 768)
    ref = x.clone()
    comm.all_reduce_sum_(x)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    # stream-ordered, no host sync: producer kernel -> all-reduce -> consumer kernel, captured and replayed
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        y = torch.zeros_like(x)
        comm.all_reduce_sum_(x)  # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            x.mul_(2.0)
            comm.all_reduce_sum_(x)
            y.copy_(x).add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    x.copy_(ref)
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(x, ref * 4.0) and torch.equal(y, ref * 4.0 + 1.0)
    comm.close()
    comm.close()  # idempotent
    with pytest.raises(RuntimeError):
        lib.call("allreduce_flat", None, x.data_ptr(), x.numel(), None)

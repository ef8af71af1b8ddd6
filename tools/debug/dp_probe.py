"""debug: where do two gloo ranks on one GPU diverge?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, torch.multiprocessing as mp

def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", VNETI_NO_GN_FUSE="1")
    import torch.distributed as dist
    import test_dp_gpu as T
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, eng = T._build(world, 1, False)
    T._feed(cfg, eng, 0, rank, False)
    def gather(t):
        t = t.detach().float().cpu().contiguous()
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return out
    def report(name, t):
        a, b = gather(t)
        if rank == 0:
            print(f"{name}: equal={torch.equal(a, b)} maxdiff={(a-b).abs().max().item():.3e} |a|={a.abs().max().item():.3e}", flush=True)
    report("params at init", eng.params)
    mode = os.environ.get("PROBE_MODE", "eager")
    if mode == "graph":
        eng.capture()
        report("params after capture", eng.params)
        report("scaler after capture", eng.scaler)
        report("opt_step after capture", eng.opt_step.float())
    for step in range(3):
        T._feed(cfg, eng, step, rank, False)
        if mode == "graph":
            eng.graph_a.replay()
        else:
            eng.forward_backward(False)
        torch.cuda.synchronize()
        report(f"step {step} local grads (expected to differ)", eng.grads)
        eng.all_reduce()
        torch.cuda.synchronize()
        report(f"step {step} reduced grads", eng.grads)
        if mode == "graph":
            eng.graph_b.replay()
        else:
            eng.optimizer_step()
        torch.cuda.synchronize()
        report(f"step {step} params", eng.params)
        report(f"step {step} exp_avg", eng.exp_avg)
        report(f"step {step} scaler", eng.scaler)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29811), nprocs=2, join=True)

import sys, os, math, torch
sys.path.insert(0, os.getcwd())
from view_neti_amd import ops
DEV="cuda"
cold = torch.empty(160 * 2 ** 20, dtype=torch.float32, device=DEV)
def t(fn, hot, touch=None, reps=9):
    ts=[]
    for _ in range(reps):
        if not hot:
            cold.fill_(0)
            if touch is not None: touch.add_(0)
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize(); ts.append(s.elapsed_time(e)*1e3)
    return sorted(ts)[len(ts)//2]
for (M,N,K) in [(16384,320,320),(4096,640,640),(16384,2560,320),(4928,3072,768),(16384,960,320)]:
    A=(torch.randn(M,K,device=DEV)).half(); B=(torch.randn(N,K,device=DEV)/math.sqrt(K)).half()
    out=torch.zeros(M,N,dtype=torch.float16,device=DEV)
    for h in (13, 9, 5, 19):
        f=lambda: ops.gemm(A,B,out,tile_hint=h,split_k=1)
        f(); torch.cuda.synchronize()
        print(M,N,K,'tile',h,'hot %.1f us'%t(f,True),'cold(B cold, A warm) %.1f us'%t(f,False,A), flush=True)

import os, sys, torch
sys.path.insert(0, os.getcwd())
from view_neti_amd import ops
B,H,N,D=1,8,4096,40
C=H*D
q=torch.randn(B*N,C,device="cuda").half(); k=torch.randn(B*N,C,device="cuda").half(); v=torch.randn(B*N,C,device="cuda").half()
o=torch.zeros_like(q); lse=torch.zeros(B,H,N,device="cuda")
os.environ["VNETI_ATTN_WIDE"]="1"
for _ in range(3): ops.attn_fwd(q,k,v,o,lse,B,H,N,N,D,D**-0.5,False)
s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): ops.attn_fwd(q,k,v,o,lse,B,H,N,N,D,D**-0.5,False)
e.record(); torch.cuda.synchronize()
print("wide fwd B1 N4096: %.1f us per launch = %.0f ns per 64-key tile" % (s.elapsed_time(e)*100, s.elapsed_time(e)*100*1000/64))

import os, sys, argparse, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("VNETI_ALLOW_SYNTHETIC_WEIGHTS", "1")
import torch, bench
from view_neti_amd import ops
args = argparse.Namespace(model="sd15", batch=4, resolution=512)
_, eng = bench.build_engine(args, 0, 1)
c = collections.Counter()
for f in eng.launches():
    fn = getattr(f, "func", f)
    name = getattr(fn, "__name__", None) or type(fn).__name__
    if getattr(fn, "__self__", None) is not None and isinstance(fn.__self__, torch.Tensor):
        name = "torch." + name
    c[name] += 1
print(sum(c.values()), "launch callables")
for k, v in c.most_common(): print(f"{v:5d} {k}")

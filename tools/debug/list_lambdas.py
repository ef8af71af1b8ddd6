"""Dev tool: which closures of the step's launch list are not `functools.partial`s of an op, where they were built, how long
each takes (events, eager)."""
import argparse, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
args = argparse.Namespace(model="sd15", batch=4, resolution=512)
_, eng = bench.build_engine(args, 0, 1)
eng.step_eager(); torch.cuda.synchronize()
launches = eng.launches()
agg = collections.OrderedDict()
for rep in range(3):
    evs = []
    for f in launches:
        if getattr(f, "func", None) is None:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); f(); e.record(); evs.append((f, s, e))
        else:
            f()
    torch.cuda.synchronize()
    if rep:
        for f, s, e in evs:
            c = getattr(f, "__code__", None)
            k = (os.path.basename(c.co_filename), c.co_firstlineno, getattr(f, "__name__", "?")) if c else ("?", 0, repr(f)[:40])
            a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e) * 1e3
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]}:{k[1]} {k[2]:20s} x{n/2:5.1f} {t/n:7.1f} us each  {t/2/1e3:6.3f} ms/step")

"""Dev tool: every launch class of the train step against ITS roofline (SURVEY §8d "per-kernel bar"): launches are
replayed in schedule order (so each starts with its operands evicted by its predecessors, as inside the step), timed
with HIP events per class, and divided into the class's algorithmic FLOPs (MFMA kernels, peak 2.5 PF dense f16) or
bytes (HBM kernels, peak 8 TB/s).  Writes gpurun_out/<tag>_kernel_roofline.json and prints a markdown table.
   python tools/kernel_roofline.py r01f"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench
from view_neti_amd import ops
from view_neti_amd.roofline import PF, TB, classify, cost, time_classes

tag = sys.argv[1] if len(sys.argv) > 1 else "r01x"
args = argparse.Namespace(model="sd15", batch=4, resolution=512)
_, eng = bench.build_engine(args, 0, 1)
eng.step_eager(); torch.cuda.synchronize()
launches = eng.launches()
classes = classify(launches)
if os.environ.get("PER_LAUNCH"):
    # every HBM-bound launch by itself (events around each one, in schedule order), grouped by (class, bytes)
    import collections
    per = collections.OrderedDict()
    hbm = {id(f): (name, cost(f)) for name, d in classes.items() if not d["flops"] for f in d["fs"]}
    for rep in range(3):
        evs = []
        for f in launches:
            if id(f) in hbm:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); f(); e.record(); evs.append((f, s, e))
            else:
                f()
        torch.cuda.synchronize()
        if rep:
            for f, s, e in evs:
                name, c = hbm[id(f)]
                nb = c[2] if c else 0.0
                fn = getattr(getattr(f, "func", f), "__name__", "?")
                k = (name, fn, int(nb))
                a = per.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e) * 1e3
    print("class | fn | MB | launches/step | us each | TB/s | ms/step")
    for (name, fn, nb), (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        n2 = n / 2; us = t / n
        print(f"{name[:38]:38s} {fn[:22]:22s} {nb/1e6:8.2f} MB x{n2:5.1f} {us:7.1f} us {nb/us/1e6 if us else 0:6.2f} TB/s {t/2/1e3:6.3f} ms")
    sys.exit(0)
out = time_classes(launches, classes)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/{tag}_kernel_roofline.json", "w"), indent=1)
print("| launch class | launches/step | ms/step | algorithmic cost | achieved | of peak |")
print("|---|---|---|---|---|---|")
for name, r in sorted(out.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if "frac" in r:
        costs = f"{r['algorithmic_gflop']/1e3:.2f} TFLOP" if r["bound"] == "mfma" else f"{r['algorithmic_mb']/1e3:.2f} GB"
        extra = f" (co-bound {r['cobound_frac_of_peak']:.2f} of peak: at {r['frac_of_cobound']:.2f} of it)" if "frac_of_cobound" in r else ""
        print(f"| {name} | {r['launches']} | {r['ms_per_step']:.2f} | {costs} | {r['achieved']:.2f} {r['unit']} | {r['frac']:.2f} of {r['peak']:g}{extra} |")
    else:
        print(f"| {name} | {r['launches']} | {r['ms_per_step']:.2f} | — | — | — |")

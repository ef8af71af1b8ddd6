"""Dev tool: fold the rocprofv3 outputs of tools/profile_round.sh into the small files kept under profiles/.

  <tag>_kernel_stats.csv   per-kernel totals/averages of the default bench command (rocprofv3 --stats)
  <tag>_pmc.json           per-kernel mean FETCH_SIZE / WRITE_SIZE per launch, corrected as
                           /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the counters are in
                           KiB-like units of 1024 B... see `units` below; FETCH_SIZE is doubled on gfx950.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                    "gpurun_out")


def find(d, pat):
    hits = glob.glob(os.path.join(root, d, "**", pat), recursive=True)
    return hits[0] if hits else None


stats = find(f"{tag}_stats", "*kernel_stats.csv")
if stats:
    rows = list(csv.reader(open(stats)))
    with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows[:80])  # the long tail is noise
    print("kernel stats ->", f"{tag}_kernel_stats.csv", f"({len(rows) - 1} kernels)")

out = {"units": "bytes per launch; FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KB (x1024 here); FETCH_SIZE is "
                "additionally doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section); "
                "WRITE_SIZE is uncalibrated", "kernels": {}}
acc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = find(f"{tag}_pmc_{c}", "*counter_collection.csv")
    if not p:
        print("missing counter file for", c)
        continue
    for r in csv.DictReader(open(p)):
        name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        if r.get("Counter_Name") != c:
            continue
        acc[name][c].append(float(r["Counter_Value"]))
for name, d in acc.items():
    e = {"launches": max(len(v) for v in d.values())}
    if d.get("FETCH_SIZE"):
        e["fetch_bytes"] = 2.0 * 1024.0 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
    if d.get("WRITE_SIZE"):
        e["write_bytes"] = 1024.0 * sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    e["hbm_bytes"] = e.get("fetch_bytes", 0.0) + e.get("write_bytes", 0.0)
    out["kernels"][name] = e
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:  # stamp: which kernel sources these counters belong to (bench.py quotes `traffic` only for a matching tree)
    from view_neti_amd.roofline import kernel_tree_sha
    out["kernel_tree_sha"] = kernel_tree_sha()
except Exception as err:  # noqa: BLE001
    out["kernel_tree_sha"] = None
    print("no kernel-tree stamp:", err)
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes"] * kv[1]["launches"])[:40]
out["kernels"] = dict(top)
json.dump(out, open(os.path.join(root, f"{tag}_pmc.json"), "w"), indent=1)
for k, v in top[:12]:
    print(f"{v['launches']:5d} x {v['hbm_bytes'] / 1e6:9.2f} MB/launch  {k[:110]}")

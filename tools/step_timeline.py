"""Dev tool (GPU box): where one captured train step's wall time goes — kernel-busy time, gaps between kernels, and the
per-kernel sums of exactly one step (not of the autotuner's probes or the warm-up, which `--stats` averages in).

    tools/step_timeline.py <tag>            (run through gpurun; writes gpurun_out/<tag>_step_timeline.json)

Runs `bench.py --steps 6 --warmup 3 --no-roofline --no-cpu-baseline` under `rocprofv3 --kernel-trace`, so the trace ENDS
with six graph replays; the step's launch sequence is found as the period of the kernel-name sequence at the end of the
trace.  Reports for the mean of the last five periods: wall (first start -> next step's first start), the union of the
kernel intervals (two streams overlap in places), the sum of the durations, the idle gaps, and the top kernels.
"""
import csv
import glob
import json
import os
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04x"
repo = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(repo, "gpurun_out")
os.makedirs(out, exist_ok=True)
d = os.path.join("/tmp", f"{tag}_trace")
env = dict(os.environ, TMPDIR="/tmp")
cache = os.path.join(out, f"{tag}_autotune.json")
if os.path.exists(cache):
    env["VNETI_AUTOTUNE_CACHE"] = cache
if not glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                    os.path.join(repo, "bench.py"), "--steps", "6", "--warmup", "3", "--no-roofline", "--no-cpu-baseline", "--no-extras"],
                   cwd="/tmp", env=env, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
n = len(names)
# a step ends with the GradScaler update (one launch per optimisation step, the last node of the captured graph); the
# order of the nodes of parallel graph branches may differ between replays, so the steps are cut there, not by periodicity
ends = [i for i, nm in enumerate(names) if "scaler_update" in nm]
if len(ends) < 7:
    raise SystemExit(f"only {len(ends)} optimisation steps in the trace")
cuts = ends[-6:]  # the last six steps' final launches
L = cuts[-1] - cuts[-2]
steps = []
for k in range(5):
    seg = rows[cuts[k] + 1:cuts[k + 1] + 1]
    nxt = rows[cuts[k + 1] + 1][0] if cuts[k + 1] + 1 < n else None
    wall = (nxt if nxt is not None else max(e for _, e, _ in seg)) - seg[0][0]
    ivs = sorted((s, e) for s, e, _ in seg)
    busy, cur_s, cur_e = 0, ivs[0][0], ivs[0][1]
    for s, e in ivs[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    steps.append(dict(wall=wall, busy=busy, sum=sum(e - s for s, e, _ in seg)))
per = {}
for s, e, nm in rows[cuts[0] + 1:cuts[5] + 1]:
    a = per.setdefault(nm, [0, 0])
    a[0] += 1
    a[1] += e - s
m = lambda key: sum(s[key] for s in steps[:-1]) / (len(steps) - 1) / 1e6  # (the last period has no successor start)
res = {"launches_per_step": L, "wall_ms": m("wall"), "kernel_busy_union_ms": m("busy"), "kernel_sum_ms": m("sum"),
       "idle_gap_ms": m("wall") - m("busy"),
       "top_kernels": [{"kernel": k[:150], "launches_per_step": v[0] / 5, "ms_per_step": v[1] / 5e6, "avg_us": v[1] / v[0] / 1e3}
                       for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:60]]}
json.dump(res, open(os.path.join(out, f"{tag}_step_timeline.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "top_kernels"}))
for t in res["top_kernels"][:45]:
    print(f"{t['ms_per_step']:7.3f} ms {t['launches_per_step']:6.1f} x {t['avg_us']:7.1f} us  {t['kernel'][:110]}")

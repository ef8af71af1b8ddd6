#!/bin/bash
# Dev tool: idle gaps between kernels inside one replayed train-step graph (rocprofv3 --kernel-trace timestamps).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last ~3 graph replays: find the last occurrences of adamw_kernel (end of a step)
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
a, b = ends[-2] + 1, ends[-1]   # one full replayed step (scaler_update may follow adamw; fine)
step = rows[a:b + 1]
t0, t1 = step[0][0], max(r[1] for r in step)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in step:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps = []
pe = step[0][1]
for s, e, n in step[1:]:
    if s > pe: gaps.append((s - pe, n))
    pe = max(pe, e)
print(f"kernels in step: {len(step)}; wall {(t1 - t0) / 1e6:.3f} ms; union busy {busy / 1e6:.3f} ms; idle {(t1 - t0 - busy) / 1e6:.3f} ms")
import collections
hist = collections.Counter()
for g, _ in gaps: hist[min(g // 1000, 10)] += 1
print("gap histogram (us bucket -> count):", sorted(hist.items()))
print("sum of gaps %.3f ms, mean %.2f us over %d gaps" % (sum(g for g, _ in gaps) / 1e6, sum(g for g, _ in gaps) / max(len(gaps), 1) / 1e3, len(gaps)))
durs = sorted((e - s for s, e, _ in step))
print("kernel duration percentiles (us): p10 %.1f p50 %.1f p90 %.1f; <5us: %d kernels" % (durs[len(durs)//10]/1e3, durs[len(durs)//2]/1e3, durs[9*len(durs)//10]/1e3, sum(d < 5000 for d in durs)))
PY
rm -rf $OUT/t

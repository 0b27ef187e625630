#!/bin/bash
# tools/finalize_trace.sh -- per-launch durations of partials_finalize_kernel inside the bench step (rocprofv3 kernel trace), with the
# kernel that ran right before each on the same queue: which of the 37 launches per step are slow, and next to what.
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/fin
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/bench.py --steps 3 --warmup 3 > $out/bench.json 2> $out/err.txt
cd $root
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/fin/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
# the last full step: the launches between the last two adamw_ema_kernel launches
idx = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
last = rows[idx[-2] + 1: idx[-1] + 1]
print("step: %d launches, %.2f ms" % (len(last), (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6))
prev_by_q = {}
out = []
for i, r in enumerate(last):
    q = r["Queue_Id"]
    if "partials_finalize" in r["Kernel_Name"]:
        p = prev_by_q.get(q)
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap = (int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3 if p else -1
        # kernels overlapping in time on other queues
        ov = [o["Kernel_Name"].replace("void (anonymous namespace)::", "")[:44] for o in last if o["Queue_Id"] != q and int(o["Start_Timestamp"]) < int(r["End_Timestamp"]) and int(o["End_Timestamp"]) > int(r["Start_Timestamp"])]
        out.append(((int(r["Start_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6, dur, gap, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), (p["Kernel_Name"].split("<")[0][-30:] if p else "-"), ov[:2]))
    prev_by_q[q] = r
for o in out:
    if o[1] > 30: print("t=%7.2f ms  %8.1f us  gap %7.1f  grid %s  after %-30s  with %s" % o)
print("sum %.1f us over %d launches" % (sum(o[1] for o in out), len(out)))
# main-queue idle time: gaps between consecutive launches of the busiest queue
from collections import Counter
mq = Counter(r["Queue_Id"] for r in last).most_common(1)[0][0]
m = [r for r in last if r["Queue_Id"] == mq]
gaps = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, a["Kernel_Name"].replace("void (anonymous namespace)::", "")[:40], b["Kernel_Name"].replace("void (anonymous namespace)::", "")[:40]) for a, b in zip(m, m[1:]))
print("main queue: %d launches, busy %.2f ms, gaps %.2f ms; largest gaps:" % (len(m), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in m) / 1e6, sum(g[0] for g in gaps) / 1e3))
for g in gaps[-8:]: print("   %8.1f us  %s -> %s" % g)
# slowest main-queue launches relative to their typical (min over the step of same kernel+grid) duration
best = {}
for r in m:
    k = (r["Kernel_Name"], r["Grid_Size_X"]); d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    best[k] = min(best.get(k, d), d)
infl = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) - best[(r["Kernel_Name"], r["Grid_Size_X"])]) / 1e3, r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:50], r["Grid_Size_X"]) for r in m)
print("inflation of main-queue launches over their fastest same-shape launch in the step: total %.2f ms" % (sum(i[0] for i in infl) / 1e3))
for i in infl[-10:]: print("   +%8.1f us  %s grid %s" % i)
PY
rm -rf $out/trace

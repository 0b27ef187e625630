"""Idle time of the GPU inside one training step from a rocprofv3 --kernel-trace CSV: per queue the busy time, and for the whole device the time
during which NO kernel was running (union of all kernel intervals), over the steady-state part of the trace.
python tools/gap_analysis.py <kernel_trace.csv> [steps_to_use]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
lo, hi = marks[-nsteps - 1] + 1, marks[-1] + 1
sel = rows[lo:hi]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
wall = (t1 - t0) / 1e6 / nsteps
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per_q = defaultdict(float)
for r in sel:
    per_q[r.get("Queue_Id", "?")] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 / nsteps
print(f"{nsteps} steps: wall {wall:.2f} ms/step, some kernel running {busy / 1e6 / nsteps:.2f} ms/step, device idle {wall - busy / 1e6 / nsteps:.2f} ms/step in {len(gaps) / nsteps:.0f} gaps/step")
print("kernel time per queue (ms/step):", {k: round(v, 2) for k, v in per_q.items()})
big = sorted(gaps, reverse=True)[:8]
print("largest gaps (us):", [round(g / 1e3, 1) for g, _ in big])
hist = defaultdict(int)
for g, _ in gaps:
    hist[min(int(g / 1e3) // 5 * 5, 50)] += 1
print("gap histogram (us bucket: count/step):", {k: round(v / nsteps, 1) for k, v in sorted(hist.items())})

"""Print the per-kernel timeline of the LAST repetition recorded in a rocprofv3 results .db (kernel-trace):
usage: python tools/trace_tail.py <results.db> [n_repetitions]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, workgroup_x, lds_size from kernels order by start").fetchall()
n = len(rows) // (int(sys.argv[2]) if len(sys.argv) > 2 else 13)
last = rows[-n:]
t0, tot, agg = last[0][1], 0.0, {}
for name, s, e, gx, wx, lds in last:
    d = (e - s) / 1000
    tot += d
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    agg[short] = agg.get(short, 0) + d
    print(f"{(s - t0) / 1000:8.1f} {d:7.1f}us wgs {gx // max(wx, 1):7d} lds {lds:6d} {short}")
print(f"sum of kernel time {tot:.1f} us, span {(last[-1][2] - t0) / 1000:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v:8.1f} us  {k}")

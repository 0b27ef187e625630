"""Kernel statistics (the columns of `rocprofv3 --stats` kernel_stats.csv) from a rocprofv3 results .db:
usage: python tools/db_stats.py <results.db> [out.csv]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for n, k, s, a, mn, mx in rows:
    lines.append(f'"{n}",{k},{s},{a:.3f},{100.0 * s / tot:.2f},{mn},{mx}')
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
for ln in lines[:40]:
    print(ln[:230])

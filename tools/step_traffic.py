"""HBM traffic of one training step per kernel family from rocprofv3 PMC passes over tools/step_run.py (FETCH_SIZE and WRITE_SIZE in
SEPARATE runs, kernel trace only beside them; both counters in KB, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md).
usage: python tools/step_traffic.py <fetch dir> <write dir> <steps in the run, warm-up included>"""
import collections
import csv
import glob
import re
import sys


def load(d, ctr):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr:
            continue
        n = r["Kernel_Name"]
        m = re.search(r"(\w+_kernel)", n)
        k = m.group(1) if m else n[:40]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return acc


fd, wd, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
keys = sorted(set(F) | set(W), key=lambda k: -(2 * F[k][1] + W[k][1]))
tf = tw = 0.0
print("| kernel family | launches / step | fetched MB / step (2 x FETCH_SIZE) | written MB / step (WRITE_SIZE) |\n|---|---|---|---|")
for k in keys:
    f, w = 2 * 1024 * F[k][1] / steps / 1e6, 1024 * W[k][1] / steps / 1e6      # counters in KB = 1024 B (WRITE_SIZE of a known store pins the unit), FETCH_SIZE doubled
    tf += f; tw += w
    if f + w >= 20:
        print(f"| `{k}` | {F[k][0] / steps:.1f} | {f:.0f} | {w:.0f} |")
print(f"| **all kernels** | | **{tf:.0f}** | **{tw:.0f}** |")
print(f"\n{(tf + tw) / 1e3:.1f} GB per step.")

"""Per-family kernel time per step of two rocprofv3 kernel-stats CSVs (tools/step_run.py B steps side): python tools/fam_cmp.py old.csv new.csv [steps+warmup]"""
import csv
import re
import sys


def fam(path):
    d = {}
    for r in csv.DictReader(open(path)):
        m = re.match(r'(?:void )?(?:\(anonymous namespace\)::)?(\w+)', r['Name'])
        k = m.group(1) if m else r['Name'][:40]
        d.setdefault(k, [0, 0.0])
        d[k][0] += int(r['Calls']); d[k][1] += float(r['TotalDurationNs']) / 1e6
    return d


old, new = fam(sys.argv[1]), fam(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 21
to = tn = 0.0
for k in sorted(set(new) | set(old), key=lambda k: -old.get(k, [0, 0])[1]):
    o, n = old.get(k, [0, 0]), new.get(k, [0, 0])
    to += o[1] / steps; tn += n[1] / steps
    if max(o[1], n[1]) / steps > 0.15:
        print(f"{k:36s} old {o[1] / steps:7.2f} ms ({o[0] // steps if o[0] else 0:3d}) new {n[1] / steps:7.2f} ms ({n[0] // steps if n[0] else 0:3d})")
print(f"{'sum':36s} old {to:7.2f} ms       new {tn:7.2f} ms")

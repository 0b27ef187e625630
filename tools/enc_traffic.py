"""HBM traffic of the student sparse-encoder forward from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs of
tools/encoder_profile.py, kernel trace only beside them; MI355X_MICROARCH.md "HBM": both counters are in KB, FETCH_SIZE reports half of
the bytes of wide coalesced reads on gfx950 and is doubled here).
usage: python tools/enc_traffic.py <fetch dir> <write dir> <batch> [size B|L|H] [patch] [mask ratio] [json out]
(json out: {"batch", "fetch_bytes", "write_bytes", "counted_bytes", "algorithmic_bytes", "launches"} per forward -- what bench.py reads for
`encoder_fwd_hbm.counted_bytes`)"""
import collections
import csv
import glob
import re
import sys


def load(d, ctr):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def short(n):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
    s = (m.group(1) + (m.group(2) or "")) if m else n[:40]
    return s.replace("unsigned short", "bf16").replace("(anonymous namespace)::", "")


def forwards(rows):
    """split the dispatch list into encoder forwards: each starts with the stem kernel"""
    starts = [i for i, r in enumerate(rows) if "stem_conv" in r["Kernel_Name"]]
    return [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]


fd, wd, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
F, W = forwards(load(fd, "FETCH_SIZE")), forwards(load(wd, "WRITE_SIZE"))
F, W = F[-8:], W[-8:]                                   # steady state
n = len(F[0])
assert all(len(f) == n for f in F + W), "the forwards do not have the same launch sequence"
fam = collections.OrderedDict()
tot_f = tot_w = 0.0
for i in range(n):
    k = short(F[0][i]["Kernel_Name"])
    f = sum(2.0 * float(x[i]["Counter_Value"]) for x in F) / len(F) * 1024     # KB (1024 B) -> bytes, x2 (gfx950 FETCH_SIZE)
    w = sum(float(x[i]["Counter_Value"]) for x in W) / len(W) * 1024
    e = fam.setdefault(k, [0, 0.0, 0.0]); e[0] += 1; e[1] += f; e[2] += w
    tot_f += f; tot_w += w
SIZE = sys.argv[4] if len(sys.argv) > 4 else "B"
PATCH, MR = (sys.argv[5] if len(sys.argv) > 5 else "128"), (sys.argv[6] if len(sys.argv) > 6 else "0.6")
per_vol = {"S": 36.1e6, "B": 1105.3e6, "L": 6135e6, "H": 30622e6}[SIZE]
algo = per_vol * B
print(f"# HBM traffic of the student sparse-encoder forward, STUNet-{SIZE} {PATCH}^3 bf16 mask {MR}, B={B} (rocprofv3 --pmc, mean of {len(F)} forwards, {n} launches each)\n")
print("| kernel | launches | fetched (2 x FETCH_SIZE) MB | written (WRITE_SIZE) MB |\n|---|---|---|---|")
for k, (c, f, w) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"| `{k}` | {c} | {f / 1e6:.1f} | {w / 1e6:.1f} |")
print(f"| **total** | {n} | **{tot_f / 1e6:.1f}** | **{tot_w / 1e6:.1f}** |\n")
print(f"fetched + written = {(tot_f + tot_w) / 1e9:.3f} GB per forward = {(tot_f + tot_w) / algo:.2f} x the algorithmic {algo / 1e9:.3f} GB "
      f"(SURVEY.md 8d: {per_vol / 1e6:.1f} MB per volume).")

if len(sys.argv) > 7:
    import json
    json.dump({"size": SIZE, "patch": int(PATCH), "mask_ratio": float(MR), "batch": B, "launches": n, "fetch_bytes": tot_f, "write_bytes": tot_w,
               "counted_bytes": tot_f + tot_w, "algorithmic_bytes": algo,
               "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/encoder_profile.py, FETCH_SIZE doubled (gfx950), KB = 1024 B; mean of the last 8 forwards"},
              open(sys.argv[7], "w"), indent=1)

"""One stream group's chain of kernels in a production run: mean duration of every (kernel, launch grid) of the busiest queue of a
rocprofv3 kernel trace, in the order of their first appearance inside an iteration.  usage: chain_table.py kernel_trace.csv"""
import csv
import re
import sys
from collections import defaultdict

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
byq = defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
q = max(byq, key=lambda k: len(byq[k]))
rs = sorted(byq[q], key=lambda r: int(r["Start_Timestamp"]))
dur, gap, first = defaultdict(list), defaultdict(list), {}
prev_end = None
for i, r in enumerate(rs):
    name = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0])
    name = re.sub(r"gs::", "", name)
    key = (name[:58], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[key].append(e - s)
    if prev_end is not None:
        gap[key].append(max(s - prev_end, 0))
    first.setdefault(key, i)
    prev_end = e
tot = 0.0
print(f"queue {q}: {len(rs)} launches")
print(f"{'kernel':58s} {'grid':>14s} {'n':>6s} {'dur us':>8s} {'gap us':>7s}")
n_iter = max(len(v) for v in dur.values())
for key in sorted(dur, key=lambda k: first[k]):
    if len(dur[key]) < 0.5 * n_iter:
        continue
    d, g = np.mean(dur[key]) / 1e3, np.mean(gap[key]) / 1e3 if gap[key] else 0.0
    tot += (d + g) * len(dur[key]) / n_iter
    print(f"{key[0]:58s} {str(key[1:]):>14s} {len(dur[key]):6d} {d:8.2f} {g:7.2f}")
print(f"sum of (duration + gap) per iteration: {tot:.1f} us")

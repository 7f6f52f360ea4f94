"""How the two stream groups of the production run share the chip: every kernel of a rocprofv3 kernel trace is classed as
F (a full-chip level-0 launch: k_jsum, level-0 tile smoothers, k_blend_tile) or S (the latency-bound rest of a group's chain:
tile launches of the levels below 0, coarse V-cycle, scalar tails); prints the share of wall time in each (queue A, queue B) state,
the mean duration of every F kernel split by what the OTHER queue was doing while it ran, and the period of an iteration.
usage: phase_overlap.py kernel_trace.csv [n_fine]     (n_fine = 513: which launches are level 0)"""
import csv
import re
import sys
from collections import defaultdict

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
n_fine = int(sys.argv[2]) if len(sys.argv) > 2 else 513
byq = defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
qs = sorted(byq, key=lambda k: -len(byq[k]))[:2]
tiles_y0 = {-(-n_fine // w) for w in (32, 34, 36, 14)}  # level-0 grid.y of the 48-row pre (window 32) / post (34 / 36) smoothers and the tail (14)


def klass(r):
    name = r["Kernel_Name"]
    gy = int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1)
    gx = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    if "k_jsum<" in name or "k_blend_tile" in name:
        return "F"
    if "k_smooth_tile" in name:
        return "F" if gx >= -(-n_fine // 124) and gy >= n_fine // 40 else "S"
    return "S"


ev = {}
for q in qs:
    rs = sorted(byq[q], key=lambda r: int(r["Start_Timestamp"]))
    ev[q] = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), klass(r), re.sub(r"^void |gs::", "", r["Kernel_Name"].split("(")[0])[:40],
              int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * int(r["Grid_Size_Y"])) for r in rs]
t0 = max(ev[q][len(ev[q]) // 10][0] for q in qs)   # steady state: skip the first / last tenth
t1 = min(ev[q][-len(ev[q]) // 10][1] for q in qs)


def state_at(q, ts):
    """state of queue q at times ts (sorted): 'F', 'S' or '-' (idle)"""
    out = np.full(len(ts), "-", dtype="<U1")
    starts = np.array([e[0] for e in ev[q]])
    ends = np.array([e[1] for e in ev[q]])
    idx = np.searchsorted(starts, ts, side="right") - 1
    ok = (idx >= 0) & (ts < ends[np.clip(idx, 0, len(ends) - 1)])
    kinds = np.array([e[2] for e in ev[q]])
    out[ok] = kinds[idx[ok]]
    return out


ts = np.arange(t0, t1, 200)  # 0.2 us grid
sa, sb = state_at(qs[0], ts), state_at(qs[1], ts)
print(f"queues {qs}, steady window {1e-6 * (t1 - t0):.1f} ms")
print("share of wall time by (queue A, queue B) state  [F = full-chip level-0 kernel, S = latency-bound chain kernel, - = idle]")
tot = len(ts)
for a in "FS-":
    print("   " + "  ".join(f"{a}{b}: {100.0 * np.count_nonzero((sa == a) & (sb == b)) / tot:5.1f}%" for b in "FS-"))
# duration of each kernel of queue A by the other queue's state at its midpoint
for q, o in ((qs[0], qs[1]),):
    mids = np.array([(e[0] + e[1]) // 2 for e in ev[q]])
    other = state_at(o, mids)
    dur = defaultdict(lambda: defaultdict(list))
    for e, st in zip(ev[q], other):
        if t0 <= e[0] and e[1] <= t1:
            dur[(e[3], e[4])][st].append((e[1] - e[0]) / 1e3)
    print(f"mean duration (us) of queue {q}'s kernels by the other queue's state at their midpoint [count]")
    for key in sorted(dur, key=lambda k: -sum(len(v) for v in dur[k].values()))[:14]:
        cells = "  ".join(f"{st}: {np.mean(v):6.2f} [{len(v):4d}]" for st, v in sorted(dur[key].items()))
        print(f"   {key[0]:40s} grid {key[1]:5d}  {cells}")
# period: start-to-start of k_blend_tile on queue A
bl = [e[0] for e in ev[qs[0]] if "k_blend_tile" in e[3] and t0 <= e[0] <= t1]
if len(bl) > 2:
    print(f"iteration period (start to start of k_blend_tile): {np.mean(np.diff(bl)) / 1e3:.1f} us")

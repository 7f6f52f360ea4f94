"""Per-queue timeline statistics from a rocprofv3 kernel trace: busy time, gaps between consecutive launches of a queue,
cross-queue overlap, and per-kernel durations.  usage: timeline_stats.py kernel_trace.csv"""
import csv
import sys
from collections import defaultdict

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
q = np.array([int(r["Queue_Id"]) for r in rows])
s = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
e = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
name = np.array([r["Kernel_Name"].split("(")[0][:60] for r in rows])
t0 = s.min()
print("launches", len(rows), "span ms", (e.max() - t0) / 1e6)
for qi in np.unique(q):
    m = q == qi
    o = np.argsort(s[m])
    ss, ee, nn = s[m][o], e[m][o], name[m][o]
    busy = (ee - ss).sum()
    gaps = ss[1:] - ee[:-1]
    span = ee.max() - ss.min()
    print(f"queue {qi}: launches {m.sum()} span {span/1e6:.1f} ms busy {busy/1e6:.1f} ms ({100*busy/span:.1f}%) "
          f"gap median {np.median(gaps)/1e3:.2f} us mean {gaps.mean()/1e3:.2f} us p90 {np.percentile(gaps,90)/1e3:.2f} us")
    # gap by the kernel that FOLLOWS the gap
    g = defaultdict(list)
    for n_, gp in zip(nn[1:], gaps):
        g[n_].append(gp)
    d = defaultdict(list)
    for n_, a, b in zip(nn, ss, ee):
        d[n_].append(b - a)
    if m.sum() > 1000:
        print("   kernel                                                        n    dur_us  gap_before_us")
        for n_ in sorted(d, key=lambda k: -np.sum(d[k])):
            if len(d[n_]) < 50:
                continue
            print(f"   {n_:60s} {len(d[n_]):6d} {np.mean(d[n_])/1e3:8.2f} {np.mean(g.get(n_, [0]))/1e3:8.2f}")
# cross-queue overlap: fraction of time with 0 / 1 / 2+ kernels running
ev = np.concatenate([np.stack([s, np.ones_like(s)], 1), np.stack([e, -np.ones_like(e)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
depth = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0])
tot = dt.sum()
for k in range(0, 5):
    print(f"time with {k} kernels in flight: {100 * dt[depth[:-1] == k].sum() / tot:.1f}%")
print(f"time with >=5: {100 * dt[depth[:-1] >= 5].sum() / tot:.1f}%")

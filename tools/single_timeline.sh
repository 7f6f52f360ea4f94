#!/bin/bash
# kernel trace of ONE 513x513 free-boundary equilibrium (tools/single_bench.py): per-kernel durations and the gaps in front of them
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_single
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 tools/single_bench.py 513 1 > $out/bench.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, numpy as np
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
s = np.array([int(r["Start_Timestamp"]) for r in rows]); e = np.array([int(r["End_Timestamp"]) for r in rows])
name = [r["Kernel_Name"].split("(")[0].replace("void ", "")[:48] for r in rows]
grid = [int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) for r in rows]
d, g = defaultdict(list), defaultdict(list)
for i in range(1, len(rows)):
    k = (name[i], grid[i]); d[k].append(e[i] - s[i]); g[k].append(s[i] - e[i - 1])
print(f"{'kernel':50s} {'grid':>9s} {'n':>6s} {'dur_us':>8s} {'gap_before_us(median)':>22s}")
for k in sorted(d, key=lambda k: -np.sum(d[k]))[:16]:
    print(f"{k[0]:50s} {k[1]:9d} {len(d[k]):6d} {np.mean(d[k])/1e3:8.2f} {np.median(g[k])/1e3:22.2f}")
PY
cat $out/bench.log
rm -rf $out

#!/bin/bash
# True (rocprofv3 kernel-trace) durations of every kernel / launch shape with ONE stream group, i.e. each kernel alone on the chip
# and 32 members per launch -- the cross-check of bench.py's HIP-event timings (which include the event pair + dispatch latency).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/iso_trace
rm -rf $out; mkdir -p $out
export SCPN_HIP_GROUPS=${GROUPS_N:-1}
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-newton --no-configs > $out/bench.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, statistics, sys
from collections import defaultdict
d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
    d[(n, g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
print("kernel,grid_threads,launches,median_us,mean_us,total_ms")
for (n, g), v in rows[:24]:
    print(f"\"{n}\",{g},{len(v)},{statistics.median(v) / 1e3:.2f},{sum(v) / len(v) / 1e3:.2f},{sum(v) / 1e6:.2f}")
PY
rm -rf $out

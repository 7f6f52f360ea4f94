#!/bin/bash
# kernel trace (start / end / queue of every launch) of a short bench run, for the per-stream timeline analysis of
# tools/timeline_stats.py.  Run on the GPU box from the repo root.
set -e
tag=${1:-tl}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-newton --no-configs > $out/bench.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_stats.py $f > $out/timeline.txt
# keep only a compact binary of the trace (the csv is large)
python3 - "$f" "$out/trace.npz" <<'PY'
import csv, sys, numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
names = sorted({r["Kernel_Name"] for r in rows})
idx = {n: i for i, n in enumerate(names)}
np.savez_compressed(sys.argv[2], names=np.array(names), k=np.array([idx[r["Kernel_Name"]] for r in rows], dtype=np.int16),
                    q=np.array([int(r["Queue_Id"]) for r in rows], dtype=np.int32),
                    s=np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64),
                    e=np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64))
PY
rm -f "$f"
cat $out/timeline.txt

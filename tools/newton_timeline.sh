#!/bin/bash
# kernel trace of the Newton lane on the bench ensemble (tools/newton_only_bench.py): per-queue busy time, gaps, kernels in flight
set -e
tag=${1:-nt}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 tools/newton_only_bench.py 513 32 2 > $out/bench.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_stats.py $f > $out/timeline.txt
rm -f "$f"
cat $out/bench.log $out/timeline.txt

#!/bin/bash
# r05 session 21: kernel-trace timeline of a FOUR-group run: which hardware queues the four streams land on, kernels in flight
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s21; mkdir -p $O
for q in 8 16; do
  export GPU_MAX_HW_QUEUES=$q SCPN_HIP_GROUPS=4
  rocprofv3 --kernel-trace --output-format csv -d $O/tr$q -o t -- python3 tools/fixed_iters_bench.py 100 1 > $O/run$q.log 2>&1 || exit 1
  f=$(find $O/tr$q -name "*kernel_trace.csv" | head -1)
  python3 tools/timeline_stats.py $f > $O/timeline_groups4_q$q.txt; rm -rf $O/tr$q
  grep iteration $O/run$q.log; grep -E "^queue|in flight" $O/timeline_groups4_q$q.txt
done

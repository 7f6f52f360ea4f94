#!/bin/bash
# r05 session 11: chain table of one stream group with the paired launches
set -o pipefail
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s11; mkdir -p $O
for P in 3 0; do
  export SCPN_HIP_PAIR=$P
  rocprofv3 --kernel-trace --output-format csv -d $O/tr$P -o t -- python3 tools/fixed_iters_bench.py 200 1 > $O/run$P.log 2>&1 || exit 1
  f=$(find $O/tr$P -name "*kernel_trace.csv" | head -1)
  python3 tools/chain_table.py $f > $O/chain_pair$P.txt && rm -rf $O/tr$P
  grep iteration $O/run$P.log; cat $O/chain_pair$P.txt
done

#!/bin/bash
# r05 session 2: production chain of one stream group with and without the CU reserve (kernel trace of the fixed-iteration run)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s02; mkdir -p $O
for R in 0 16; do
  export SCPN_HIP_CURESERVE=$R
  rocprofv3 --kernel-trace --output-format csv -d $O/tr$R -o t -- python3 tools/fixed_iters_bench.py 200 1 > $O/run$R.log 2>&1 || exit 1
  f=$(find $O/tr$R -name "*kernel_trace.csv" | head -1)
  python3 tools/chain_table.py $f > $O/chain_cureserve$R.txt && rm -rf $O/tr$R
  cat $O/run$R.log $O/chain_cureserve$R.txt
done

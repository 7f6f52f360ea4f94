#!/bin/bash
# r06 session 3: exact vs contracted arithmetic -- (a) production chain of one stream group (kernel trace of the fixed-iteration
# run, two groups), (b) every kernel alone on the chip at full fill (one stream group, 32 members per launch)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s03; mkdir -p $O
for M in exact contracted; do
  export SCPN_HIP_ARITH=$M
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$M -o t -- python3 tools/fixed_iters_bench.py 200 1 > $O/run_$M.log 2>&1 || exit 1
  f=$(find $O/tr_$M -name "*kernel_trace.csv" | head -1)
  python3 tools/chain_table.py $f > $O/chain_$M.txt && rm -rf $O/tr_$M
  cat $O/run_$M.log $O/chain_$M.txt
  SCPN_HIP_GROUPS=1 rocprofv3 --kernel-trace --output-format csv -d $O/iso_$M -o t -- python3 tools/fixed_iters_bench.py 100 1 > $O/iso_run_$M.log 2>&1 || exit 1
  f=$(find $O/iso_$M -name "*kernel_trace.csv" | head -1)
  python3 tools/chain_table.py $f > $O/iso_chain_$M.txt && rm -rf $O/iso_$M
  cat $O/iso_run_$M.log $O/iso_chain_$M.txt
done

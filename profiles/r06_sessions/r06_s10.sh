#!/bin/bash
# NOTE: SCPN_HIP_TAILFUSE exists only with profiles/r06_negative/fused_picard_tail.diff.txt applied (archived experiment, not in the tree)
# r06 session 10: chain of one stream group (two groups, production) and kernels alone at full fill (one group), fused tail on / off
set -o pipefail
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s10; mkdir -p $O
export SCPN_HIP_ARITH=contracted
for T in 1 0; do
  export SCPN_HIP_TAILFUSE=$T
  SCPN_HIP_GROUPS=1 rocprofv3 --kernel-trace --output-format csv -d $O/iso_$T -o t -- python3 tools/fixed_iters_bench.py 100 1 > $O/iso_run_$T.log 2>&1 || exit 1
  f=$(find $O/iso_$T -name "*kernel_trace.csv" | head -1)
  python3 tools/chain_table.py $f > $O/iso_chain_tail$T.txt && rm -rf $O/iso_$T
  grep us/iteration $O/iso_run_$T.log; cat $O/iso_chain_tail$T.txt
done

#!/bin/bash
# r05 session 6: how the two stream groups share the chip (phase overlap of full-chip and latency-bound kernels)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s06; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 tools/fixed_iters_bench.py 200 1 > $O/run.log 2>&1 || exit 1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 tools/phase_overlap.py $f > $O/phase_overlap.txt; python3 tools/chain_table.py $f > $O/chain.txt
rm -rf $O/tr
cat $O/run.log | grep iteration; cat $O/phase_overlap.txt

#!/bin/bash
# round-4 session 21: one group's chain in the production trace after the coarse-kernel / Gauss-Seidel / zero-input changes
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04v; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --output-format csv -d $out/t -o t -- python3 tools/fixed_iters_bench.py 200 1 > $out/run.log 2>&1
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 tools/chain_table.py $f | tee $out/chain.txt
rm -rf $out/t

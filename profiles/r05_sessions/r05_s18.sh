#!/bin/bash
# r05 session 18: final tree -- whole GPU suite (timed), isolated kernel durations, production chain of one group
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s18; mkdir -p $O
SECONDS=0; timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; echo "suite wall: ${SECONDS}s"; [ $rc -eq 0 ] || exit $rc
bash tools/isolated_kernel_times.sh > $O/r05a_isolated_kernel_times.txt 2>&1; head -16 $O/r05a_isolated_kernel_times.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 tools/fixed_iters_bench.py 200 1 > $O/run.log 2>&1 || exit 1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 tools/chain_table.py $f > $O/r05a_chain_one_group.txt; python3 tools/phase_overlap.py $f > $O/r05a_phase_overlap.txt; rm -rf $O/tr
cat $O/r05a_chain_one_group.txt
python -c "import __graft_entry__ as g; g.smoke()"

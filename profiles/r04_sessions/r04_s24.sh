#!/bin/bash
# round-4 session 24: final profile set r04e (kernel stats, PMC passes, isolated kernel times, bench line, chain table, coarse stamps)
set -e
export GPU_MAX_HW_QUEUES=8
bash tools/collect_profiles.sh r04e
bash tools/isolated_kernel_times.sh > gpurun_out/summary_r04e/r04e_isolated_kernel_times.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/summary_r04e/r04e_bench.json 2> gpurun_out/summary_r04e/bench.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04e_t -o t -- python3 tools/fixed_iters_bench.py 200 1 > gpurun_out/summary_r04e/chain_run.log 2>&1
f=$(find gpurun_out/r04e_t -name "*kernel_trace.csv" | head -1)
python3 tools/chain_table.py $f > gpurun_out/summary_r04e/r04e_chain_one_group.txt
rm -rf gpurun_out/r04e_t
timeout -k 10 200 python tools/cv_stamps.py > gpurun_out/summary_r04e/r04e_coarse_vcycle_stamps.txt 2>&1
ls gpurun_out/summary_r04e
python -c "
import json; d=json.load(open('gpurun_out/summary_r04e/r04e_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d['roofline']['traffic'])"

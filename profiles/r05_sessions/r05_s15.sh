#!/bin/bash
# r05 session 15: the round's committed evidence -- full GPU test suite, the bench line, rocprofv3 kernel stats + PMC passes (tools/collect_profiles.sh)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s15; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -4 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 2 > $O/r05a_bench.json 2> $O/bench.err; rc=$?; tail -3 $O/bench.err; [ $rc -eq 0 ] || exit $rc
python -c "
import json; d=json.load(open('$O/r05a_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['psi_rel_linf_vs_gpu'], d['parity']['pass'], d['strong_scaling_n1']['value'])"
bash tools/collect_profiles.sh r05a > $O/collect.log 2>&1; rc=$?; tail -5 $O/collect.log; exit $rc

#!/bin/bash
# r05 session 3: new parity tests (native lane, kappa scan) + the bench line of the new workload
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_freegs_gate.py tests/test_gpu_parity.py -x -q -m gpu -k "freegs or native or five or config5 or strict" > $O/pytest_new.log 2>&1; rc=$?
tail -5 $O/pytest_new.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; rc=$?
tail -c 1500 $O/bench.json; tail -5 $O/bench.err
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --total-members 256 --no-cpu-baseline --no-roofline --no-newton --no-configs > $O/bench_strong256.json 2> $O/bench_strong256.err
head -c 1200 $O/bench_strong256.json

#!/bin/bash
# r05 session 5: run-ahead Picard loop (SCPN_HIP_RUNAHEAD = iterations the host may enqueue beyond the device's progress word;
# 0 = the blocking poll every check_every iterations of rounds 1-4) on the bench workload and on ONE equilibrium
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s05; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "picard or free_boundary or config2 or non_finite" > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for i in 1 2 3; do
 for val in 0 1 2 3 4; do
  SCPN_HIP_RUNAHEAD=$val timeout -k 10 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --no-newton --no-configs 2>/dev/null | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('RUNAHEAD=$val', d['value'], d['ms_per_step'], d['config']['picard_iterations_per_member'])"
 done
done | tee $O/ab_runahead_bench.txt
for i in 1 2; do
 for val in 0 2; do
  echo -n "single RUNAHEAD=$val: "; SCPN_HIP_RUNAHEAD=$val timeout -k 10 200 python tools/single_bench.py 513 5 1
 done
done | tee $O/ab_runahead_single.txt

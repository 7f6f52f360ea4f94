#!/bin/bash
# round-4 session 11: register-tile Jacobi seed relaxation vs the LDS-tile form
set -e
out=gpurun_out/r04l; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_contracts.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
for i in 1 2 3; do
  for j in lds reg; do echo -n "jacobi=$j (20 iterations per call: set-up dominated): "; SCPN_HIP_JACOBI=$j timeout -k 10 200 python tools/fixed_iters_bench.py 20 5; done
done 2>&1 | tee $out/ab.txt
for j in lds reg; do echo "bench jacobi=$j: "; SCPN_HIP_JACOBI=$j python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-newton --no-configs | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a $out/ab.txt

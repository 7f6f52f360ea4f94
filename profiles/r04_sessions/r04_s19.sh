#!/bin/bash
# round-4 session 19: coarse levels start from zero without the zero field in memory (SCPN_HIP_ZEROIN=0: written and read back)
set -e
out=gpurun_out/r04t; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_contracts.py tests/test_gpu_api.py tests/test_gpu_variants.py tests/test_gpu_newton.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
for i in 1 2 3; do
  for g in 1 0; do echo -n "ZEROIN=$g: "; SCPN_HIP_ZEROIN=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done
done 2>&1 | tee $out/ab.txt
for g in 1 0; do echo -n "single ZEROIN=$g: "; SCPN_HIP_ZEROIN=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee -a $out/ab.txt

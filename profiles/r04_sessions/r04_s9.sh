#!/bin/bash
# round-4 session 9: the Picard tail rebuilds its source from the current iterate (24 instead of 32 B per point)
set -e
out=gpurun_out/r04i; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_contracts.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
for i in 1 2 3; do
  for g in 0 1; do
    echo -n "TAILGEN=$g: "; SCPN_HIP_TAILGEN=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  done
done 2>&1 | tee $out/ab.txt
for g in 0 1; do echo -n "single TAILGEN=$g: "; SCPN_HIP_TAILGEN=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee -a $out/ab.txt

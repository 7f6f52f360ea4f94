#!/bin/bash
# round-4 session 8: k_picard_final with its partials in one round trip; stream priorities (experiment)
set -e
out=gpurun_out/r04h; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=scpn_fusion_core_amd
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_contracts.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
for i in 1 2 3; do
  echo -n "glds0 (old final): "; SCPN_HIP_LIB=$PWD/$P/libscpn_hip.glds0.so timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  for pr in none hi alt; do
    echo -n "prod prio=$pr: "; SCPN_HIP_PRIO=$pr timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  done
done 2>&1 | tee $out/ab.txt
echo -n "single prod: "; timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1 | tee -a $out/ab.txt

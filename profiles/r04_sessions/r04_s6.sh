#!/bin/bash
# round-4 session 6: rewritten Picard tail (per-column accumulators, cold-path sanitising, power-of-two scaled squares) vs round 3's
set -e
out=gpurun_out/r04f; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=scpn_fusion_core_amd
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_contracts.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -15 | tee $out/tests.txt
for i in 1 2 3; do
  for v in bl3 prod; do
    lib=$PWD/$P/libscpn_hip.$v.so; [ $v = prod ] && lib=$PWD/$P/libscpn_hip.so
    echo -n "$v: "; SCPN_HIP_LIB=$lib timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  done
done 2>&1 | tee $out/ab.txt
for v in bl3 prod; do
  lib=$PWD/$P/libscpn_hip.$v.so; [ $v = prod ] && lib=$PWD/$P/libscpn_hip.so
  echo -n "single $v: "; SCPN_HIP_LIB=$lib timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1
done 2>&1 | tee -a $out/ab.txt

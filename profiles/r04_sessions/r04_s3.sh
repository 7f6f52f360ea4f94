#!/bin/bash
# round-4 session 3: residual coefficients of the six-row DEFECT tile shared through LDS; full GPU suite on the product build
set -e
out=gpurun_out/r04c; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=scpn_fusion_core_amd
for i in 1 2 3; do
  for v in p0 sd0 prod; do
    lib=$PWD/$P/libscpn_hip.$v.so; [ $v = prod ] && lib=$PWD/$P/libscpn_hip.so
    echo -n "$v: "; SCPN_HIP_LIB=$lib timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  done
done 2>&1 | tee $out/ab.txt
for v in p0 prod; do
  lib=$PWD/$P/libscpn_hip.$v.so; [ $v = prod ] && lib=$PWD/$P/libscpn_hip.so
  echo -n "single $v: "; SCPN_HIP_LIB=$lib timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1
done 2>&1 | tee -a $out/ab.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $out/gpu_tests.txt

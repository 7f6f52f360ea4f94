#!/bin/bash
# round-4 session 2: prolongation load phase of the post-smoother (rows in flight, distinct coarse rows up front)
set -e
out=gpurun_out/r04b; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=scpn_fusion_core_amd
for i in 1 2; do
  for v in p0 lg2 pw2 pw3 pw6; do
    echo -n "$v: "; SCPN_HIP_LIB=$PWD/$P/libscpn_hip.$v.so timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  done
done 2>&1 | tee $out/ab.txt
for v in p0 pw2 pw6; do
  echo "== single equilibrium $v"; SCPN_HIP_LIB=$PWD/$P/libscpn_hip.$v.so timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1
done 2>&1 | tee -a $out/ab.txt
SCPN_HIP_LIB=$PWD/$P/libscpn_hip.pw2.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 | tee $out/parity_pw2.txt

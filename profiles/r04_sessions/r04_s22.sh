#!/bin/bash
# round-4 session 22: coarse kernel -- restriction by the block lanes, 9-column level with fused prolongation / defect
set -e
out=gpurun_out/r04w; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=$PWD/scpn_fusion_core_amd
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_contracts.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
timeout -k 10 200 python tools/cv_stamps.py 2>&1 | tee $out/cv_stamps.txt
for i in 1 2 3; do
  for v in so prev.so; do echo -n "$v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done
done 2>&1 | tee $out/ab.txt
for v in so prev.so; do echo -n "single $v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee -a $out/ab.txt

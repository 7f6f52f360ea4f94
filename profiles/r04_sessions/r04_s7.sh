#!/bin/bash
# round-4 session 7: LDS-DMA staging of the post-smoother's inner-row sources (all rows of a wave in flight); bit-identity of the switches
set -e
out=gpurun_out/r04g; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=scpn_fusion_core_amd
timeout -k 10 900 python -m pytest tests/test_gpu_variants.py -x -q -m gpu 2>&1 | tail -15 | tee $out/tests.txt
for i in 1 2 3; do
  for v in glds0 prod; do
    lib=$PWD/$P/libscpn_hip.$v.so; [ $v = prod ] && lib=$PWD/$P/libscpn_hip.so
    echo -n "$v: "; SCPN_HIP_LIB=$lib timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  done
done 2>&1 | tee $out/ab.txt

#!/bin/bash
# round-4 session 1: baseline iteration cost, poll-interval sensitivity, tile-geometry variants (fixed 200 iterations, 32 x 513^2)
set -e
out=gpurun_out/r04a; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=scpn_fusion_core_amd
for i in 1 2; do
  for ce in 8 50 200; do echo -n "base CHECK_EVERY=$ce: "; CHECK_EVERY=$ce timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done
  for v in g72x12 g60x10 g64x8w5 g96x16; do
    echo -n "$v: "; SCPN_HIP_LIB=$PWD/$P/libscpn_hip.$v.so timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  done
done 2>&1 | tee $out/ab.txt

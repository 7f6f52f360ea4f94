#!/bin/bash
# round-4 session 20: Picard tail fused into the level-0 post-smoother (SCPN_HIP_BLENDFUSE=0: separate k_blend_tile launch)
set -e
out=gpurun_out/r04u; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 | tee $out/tests.txt
for i in 1 2 3; do
  for g in 1 0; do echo -n "BLENDFUSE=$g: "; SCPN_HIP_BLENDFUSE=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done
done 2>&1 | tee $out/ab.txt
for g in 1 0; do echo -n "single BLENDFUSE=$g: "; SCPN_HIP_BLENDFUSE=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee -a $out/ab.txt

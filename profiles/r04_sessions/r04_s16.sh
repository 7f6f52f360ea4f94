#!/bin/bash
# round-4 session 16: coarse V-cycle kernel, 17^2..65^2 levels as 2x2 register blocks per lane (CV_BLOCKS)
set -e
out=gpurun_out/r04q; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=$PWD/scpn_fusion_core_amd
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_contracts.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
timeout -k 10 200 python tools/cv_stamps.py > $out/cv_stamps_blocks.txt 2>&1
# (round 5: the stamped variant is passed to tools/cv_stamps.py through its environment instead of rewriting the tracked tool with sed)
timeout -k 10 200 env STAMPS_VARIANT=stamps_cvb0 STAMPS_DEFINES="CV_BLOCKS=0" python tools/cv_stamps.py > $out/cv_stamps_pairs.txt 2>&1
paste $out/cv_stamps_blocks.txt $out/cv_stamps_pairs.txt | tee $out/cv_stamps.txt
for i in 1 2 3; do
  for v in so cvb0.so; do echo -n "$v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done
done 2>&1 | tee $out/ab.txt
for v in so cvb0.so; do echo -n "single $v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee -a $out/ab.txt

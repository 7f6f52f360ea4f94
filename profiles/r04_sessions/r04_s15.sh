#!/bin/bash
# round-4 session 15: Gauss-Seidel forms (omega == 1.0) of the tile smoother AND the coarse V-cycle kernel; SCPN_HIP_GS1=0 = general forms
set -e
out=gpurun_out/r04p; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_contracts.py tests/test_gpu_api.py tests/test_gpu_variants.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
for i in 1 2 3; do
  for g in 1 0; do echo -n "GS1=$g: "; SCPN_HIP_GS1=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done
done 2>&1 | tee $out/ab.txt
for g in 1 0; do echo -n "single GS1=$g: "; SCPN_HIP_GS1=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee -a $out/ab.txt
for g in 1 0; do echo -n "bench GS1=$g: "; SCPN_HIP_GS1=$g python bench.py --steps 3 --warmup 1 --no-roofline --no-newton --no-configs | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cpu_baseline']['psi_rel_linf_vs_gpu'], d['cpu_baseline']['picard_iterations_gpu_member0'])"; done 2>&1 | tee -a $out/ab.txt

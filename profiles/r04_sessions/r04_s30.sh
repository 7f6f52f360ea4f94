#!/bin/bash
# round-4 session 30: speculative k_jsum behind the fused tail + one scalar launch for two iterations (SCPN_HIP_SPECJSUM=0: plain order)
set -e
out=gpurun_out/r04ad; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
SCPN_HIP_SUMFUSE=0 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_reference_contracts.py -x -q -m gpu 2>&1 | tail -8 | tee $out/tests.txt
for i in 1 2 3; do
  for g in 1 0; do echo -n "SPECJSUM=$g: "; SCPN_HIP_SPECJSUM=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done
done 2>&1 | tee $out/ab.txt
for g in 1 0; do echo -n "bench SPECJSUM=$g: "; SCPN_HIP_SPECJSUM=$g python bench.py --steps 5 --warmup 1 --no-roofline --no-newton --no-configs | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cpu_baseline']['psi_rel_linf_vs_gpu'], d['cpu_baseline']['picard_iterations_gpu_member0'], d['config']['picard_iterations_per_member'])"; done 2>&1 | tee -a $out/ab.txt

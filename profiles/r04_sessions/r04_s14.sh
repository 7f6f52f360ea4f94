#!/bin/bash
# round-4 session 14: omega == 1.0 instantiations of the tile smoother (relaxation step as one fma)
set -e
out=gpurun_out/r04o; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=$PWD/scpn_fusion_core_amd
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_contracts.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -5 | tee $out/tests.txt
for i in 1 2 3; do
  for v in so gs0.so; do echo -n "$v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done
done 2>&1 | tee $out/ab.txt
for v in so gs0.so; do echo -n "single $v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee -a $out/ab.txt
for v in so gs0.so; do echo -n "bench $v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v python bench.py --steps 3 --warmup 1 --no-roofline --no-newton --no-configs | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cpu_baseline']['psi_rel_linf_vs_gpu'], d['cpu_baseline']['picard_iterations_gpu_member0'])"; done 2>&1 | tee -a $out/ab.txt

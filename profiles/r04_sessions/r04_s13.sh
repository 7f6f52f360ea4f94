#!/bin/bash
# round-4 session 13: store policy of the produced fields (plain / non-temporal / sc1 write-through / sc0 sc1)
set -e
out=gpurun_out/r04n; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=$PWD/scpn_fusion_core_amd
for i in 1 2 3; do
  for v in so st1.so st2.so st3.so; do echo -n "$v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done
done 2>&1 | tee $out/ab.txt
for v in so st1.so st2.so st3.so; do echo -n "single $v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee -a $out/ab.txt
for v in so st1.so st2.so; do echo -n "bench $v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v python bench.py --steps 3 --warmup 1 --no-roofline --no-newton --no-configs | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cpu_baseline']['psi_rel_linf_vs_gpu'], d['cpu_baseline']['picard_iterations_gpu_member0'])"; done 2>&1 | tee -a $out/ab.txt

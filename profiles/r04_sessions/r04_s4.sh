#!/bin/bash
# round-4 session 4: stream-group count with enough hardware queues (GPU_MAX_HW_QUEUES) for every group
set -e
out=gpurun_out/r04d; mkdir -p $out
for q in 8 16 unset; do
  for g in 1 2 3 4 6 8; do
    if [ $q = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    echo -n "queues=$q groups=$g: "; SCPN_HIP_GROUPS=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  done
done 2>&1 | tee $out/groups.txt

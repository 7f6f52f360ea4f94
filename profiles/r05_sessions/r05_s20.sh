#!/bin/bash
# r05 session 20: four stream groups once more (round 3 / 4: the queues ran about two-wide), with more hardware queues
set -o pipefail
O=gpurun_out/r05_s20; mkdir -p $O
for q in 8 16; do for g in 2 4; do echo -n "GPU_MAX_HW_QUEUES=$q GROUPS=$g: "; GPU_MAX_HW_QUEUES=$q SCPN_HIP_GROUPS=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 2; done; done 2>&1 | tee $O/ab_groups4.log
echo -n "64 members, 2 groups: "; GPU_MAX_HW_QUEUES=8 SCPN_HIP_GROUPS=2 timeout -k 10 200 python tools/fixed_iters_bench.py 100 2 513 64
echo -n "64 members, 4 groups: "; GPU_MAX_HW_QUEUES=8 SCPN_HIP_GROUPS=4 timeout -k 10 200 python tools/fixed_iters_bench.py 100 2 513 64

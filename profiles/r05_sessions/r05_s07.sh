#!/bin/bash
# r05 session 7: s_setprio 3 for the latency-bound links of the chain (SCPN_HIP_SETPRIO bit mask: 1 tiles of levels >= 1, 2 coarse kernel, 4 scalar tails)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s07; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { cat $O/smoke.log; exit 1; }
tools/ab_fixed_env.sh SCPN_HIP_SETPRIO 0 8 10 0 8 > $O/ab_setprio_level0.log 2>&1
cat $O/ab_setprio_level0.log
for v in 0 8; do echo -n "single SETPRIO=$v: "; SCPN_HIP_SETPRIO=$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee $O/ab_setprio_level0_single.log

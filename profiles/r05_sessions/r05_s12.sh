#!/bin/bash
# r05 session 12: paired launches, quick A/B of a variant (fixed-iteration runs only)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s12; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { cat $O/smoke.log; exit 1; }
for i in 1 2; do for v in 0 1 2 3; do echo -n "SCPN_HIP_PAIR=$v: "; SCPN_HIP_PAIR=$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 2; done; done 2>&1 | tee $O/ab_pair_quick.log

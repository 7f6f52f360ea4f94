#!/bin/bash
# r05 session 1: CU reservation for the coarse V-cycle kernel (SCPN_HIP_CURESERVE = CUs per group the other group's stream may not use)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s01; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || exit 1
tools/ab_fixed_env.sh SCPN_HIP_CURESERVE 0 8 16 32 0 16 > $O/ab_cureserve.log 2>&1
cat $O/ab_cureserve.log
# single equilibrium is unaffected (one group) -- sanity only
python bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err && tail -c 600 $O/bench_default.json

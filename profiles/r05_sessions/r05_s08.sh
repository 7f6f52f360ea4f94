#!/bin/bash
# r05 session 8: runtime switches that take links off the chain, re-measured with this round's kernels: J_raw sum finished by the
# pre-smoother (SCPN_HIP_SUMFUSE=1; default: only for groups of <= 14 members), three stream groups
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s08; mkdir -p $O
tools/ab_fixed_env.sh SCPN_HIP_SUMFUSE -1 1 > $O/ab_sumfuse.log 2>&1; cat $O/ab_sumfuse.log
tools/ab_fixed_env.sh SCPN_HIP_GROUPS 2 3 > $O/ab_groups.log 2>&1; cat $O/ab_groups.log

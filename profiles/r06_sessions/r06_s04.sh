#!/bin/bash
# r06 session 4: (a) J_raw sum finished by its first consumer (SCPN_HIP_SUMFUSE) under contracted arithmetic; (b) in-kernel stamps of the
# level-0 post-smoother (PROLONG) tile, exact vs contracted, 32 members per launch
set -o pipefail
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s04; mkdir -p $O
export SCPN_HIP_ARITH=contracted
bash tools/ab_fixed_env.sh SCPN_HIP_SUMFUSE 0 1 > $O/ab_sumfuse.txt 2>&1; cat $O/ab_sumfuse.txt
for M in exact contracted; do
  SCPN_HIP_ARITH=$M STAMPS_MODE=post STAMPS_DEFINES="STAMP_PROLONG=true" python3 tools/tile_stamps.py > $O/stamps_post_$M.txt 2>&1; cat $O/stamps_post_$M.txt
  SCPN_HIP_ARITH=$M STAMPS_MODE=defect python3 tools/tile_stamps.py > $O/stamps_pre_$M.txt 2>&1; cat $O/stamps_pre_$M.txt
done

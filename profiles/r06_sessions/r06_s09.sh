#!/bin/bash
# NOTE: SCPN_HIP_TAILFUSE exists only with profiles/r06_negative/fused_picard_tail.diff.txt applied (archived experiment, not in the tree)
# r06 session 9: fused Picard tail (level-0 post-smoother + blend / norms / topology) -- contracted tests against the oracle, then
# per-iteration time with and without it
set -o pipefail
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s09; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_fast_arith.py -x -q -m gpu > $O/fast.txt 2>&1 || { tail -60 $O/fast.txt; exit 1; }
tail -3 $O/fast.txt
export SCPN_HIP_ARITH=contracted
bash tools/ab_fixed_env.sh SCPN_HIP_TAILFUSE 0 1 > $O/ab_tailfuse.txt 2>&1; cat $O/ab_tailfuse.txt

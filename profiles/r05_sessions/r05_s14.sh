#!/bin/bash
# NOTE (round 6): this script drives switches / tools that are NOT in the tree at HEAD (SCPN_HIP_SPECJSUM ): they exist only with the archived diffs of
# profiles/r05_negative/ applied (cu_reservation_for_the_coarse_kernel.md, issue_priority_s_setprio.md, paired_tile_launches.diff.txt) or, for the
# timing emulation, profiles/r05_emu_fuse_timing_experiment.diff.txt.  Without them the environment variables are ignored (a no-op A/B) and
# tools/pair_stamps.py / libscpn_hip.base.so are absent.  Kept as the record of how the round-5 tables were produced.
if [ -z "$SCPN_R05_DIFFS_APPLIED" ]; then echo "$0: archived experiment -- apply the round-5 diffs first (see the header), then export SCPN_R05_DIFFS_APPLIED=1" >&2; exit 2; fi
# r05 session 14: the archived speculative-k_jsum + merged-scalar-tail order (r04_negative) re-applied on this round's tree:
# ONE equilibrium and the 32-member ensemble
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s14; mkdir -p $O
SCPN_HIP_SUMFUSE=0 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "picard or free_boundary or non_finite" > $O/pytest_spec.log 2>&1 || { tail -30 $O/pytest_spec.log; exit 1; }
tail -2 $O/pytest_spec.log
for i in 1 2 3; do
  echo -n "B=1 sum in first consumer (default): "; timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1
  echo -n "B=1 SUMFUSE=0 SPECJSUM=1: "; SCPN_HIP_SUMFUSE=0 SCPN_HIP_SPECJSUM=1 timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1
  echo -n "B=1 SUMFUSE=0 SPECJSUM=0: "; SCPN_HIP_SUMFUSE=0 SCPN_HIP_SPECJSUM=0 timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1
done 2>&1 | tee $O/ab_spec_single.log
for i in 1 2 3; do for v in 0 1; do echo -n "B=32 SPECJSUM=$v: "; SCPN_HIP_SPECJSUM=$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done; done 2>&1 | tee $O/ab_spec_32.log
for v in 0 1; do echo -n "single_bench SUMFUSE=0 SPECJSUM=$v: "; SCPN_HIP_SUMFUSE=0 SCPN_HIP_SPECJSUM=$v timeout -k 10 200 python tools/single_bench.py 513 5 1 | tr '\n' ' '; echo; done | tee $O/single_bench.log

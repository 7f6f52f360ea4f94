#!/bin/bash
# NOTE (round 6): this script drives switches / tools that are NOT in the tree at HEAD (SCPN_HIP_CURESERVE ): they exist only with the archived diffs of
# profiles/r05_negative/ applied (cu_reservation_for_the_coarse_kernel.md, issue_priority_s_setprio.md, paired_tile_launches.diff.txt) or, for the
# timing emulation, profiles/r05_emu_fuse_timing_experiment.diff.txt.  Without them the environment variables are ignored (a no-op A/B) and
# tools/pair_stamps.py / libscpn_hip.base.so are absent.  Kept as the record of how the round-5 tables were produced.
if [ -z "$SCPN_R05_DIFFS_APPLIED" ]; then echo "$0: archived experiment -- apply the round-5 diffs first (see the header), then export SCPN_R05_DIFFS_APPLIED=1" >&2; exit 2; fi
# r05 session 1: CU reservation for the coarse V-cycle kernel (SCPN_HIP_CURESERVE = CUs per group the other group's stream may not use)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s01; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || exit 1
tools/ab_fixed_env.sh SCPN_HIP_CURESERVE 0 8 16 32 0 16 > $O/ab_cureserve.log 2>&1
cat $O/ab_cureserve.log
# single equilibrium is unaffected (one group) -- sanity only
python bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err && tail -c 600 $O/bench_default.json

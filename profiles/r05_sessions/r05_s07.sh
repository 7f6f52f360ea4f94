#!/bin/bash
# NOTE (round 6): this script drives switches / tools that are NOT in the tree at HEAD (SCPN_HIP_SETPRIO ): they exist only with the archived diffs of
# profiles/r05_negative/ applied (cu_reservation_for_the_coarse_kernel.md, issue_priority_s_setprio.md, paired_tile_launches.diff.txt) or, for the
# timing emulation, profiles/r05_emu_fuse_timing_experiment.diff.txt.  Without them the environment variables are ignored (a no-op A/B) and
# tools/pair_stamps.py / libscpn_hip.base.so are absent.  Kept as the record of how the round-5 tables were produced.
if [ -z "$SCPN_R05_DIFFS_APPLIED" ]; then echo "$0: archived experiment -- apply the round-5 diffs first (see the header), then export SCPN_R05_DIFFS_APPLIED=1" >&2; exit 2; fi
# r05 session 7: s_setprio 3 for the latency-bound links of the chain (SCPN_HIP_SETPRIO bit mask: 1 tiles of levels >= 1, 2 coarse kernel, 4 scalar tails)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s07; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { cat $O/smoke.log; exit 1; }
tools/ab_fixed_env.sh SCPN_HIP_SETPRIO 0 8 10 0 8 > $O/ab_setprio_level0.log 2>&1
cat $O/ab_setprio_level0.log
for v in 0 8; do echo -n "single SETPRIO=$v: "; SCPN_HIP_SETPRIO=$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee $O/ab_setprio_level0_single.log

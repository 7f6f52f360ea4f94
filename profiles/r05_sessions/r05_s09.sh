#!/bin/bash
# NOTE (round 6): this script drives switches / tools that are NOT in the tree at HEAD (SCPN_HIP_EMU_FUSE ): they exist only with the archived diffs of
# profiles/r05_negative/ applied (cu_reservation_for_the_coarse_kernel.md, issue_priority_s_setprio.md, paired_tile_launches.diff.txt) or, for the
# timing emulation, profiles/r05_emu_fuse_timing_experiment.diff.txt.  Without them the environment variables are ignored (a no-op A/B) and
# tools/pair_stamps.py / libscpn_hip.base.so are absent.  Kept as the record of how the round-5 tables were produced.
if [ -z "$SCPN_R05_DIFFS_APPLIED" ]; then echo "$0: archived experiment -- apply the round-5 diffs first (see the header), then export SCPN_R05_DIFFS_APPLIED=1" >&2; exit 2; fi
# r05 session 9: TIMING emulation of fused [257^2 + 129^2] tile launches (SCPN_HIP_EMU_FUSE bit 0: pre-smoothers, bit 1: post-smoothers):
# the 129^2 launch dropped, its tile count added to the 257^2 launch as out-of-grid tiles -- the best a clustered launch with free hand-offs could do
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s09; mkdir -p $O
tools/ab_fixed_env.sh SCPN_HIP_EMU_FUSE 0 1 2 3 0 3 > $O/ab_emu_fuse.log 2>&1; cat $O/ab_emu_fuse.log
for v in 0 3; do echo -n "single EMU_FUSE=$v: "; SCPN_HIP_EMU_FUSE=$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee $O/ab_emu_fuse_single.log

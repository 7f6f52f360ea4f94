#!/bin/bash
# NOTE (round 6): this script drives switches / tools that are NOT in the tree at HEAD (pair_stamps ): they exist only with the archived diffs of
# profiles/r05_negative/ applied (cu_reservation_for_the_coarse_kernel.md, issue_priority_s_setprio.md, paired_tile_launches.diff.txt) or, for the
# timing emulation, profiles/r05_emu_fuse_timing_experiment.diff.txt.  Without them the environment variables are ignored (a no-op A/B) and
# tools/pair_stamps.py / libscpn_hip.base.so are absent.  Kept as the record of how the round-5 tables were produced.
if [ -z "$SCPN_R05_DIFFS_APPLIED" ]; then echo "$0: archived experiment -- apply the round-5 diffs first (see the header), then export SCPN_R05_DIFFS_APPLIED=1" >&2; exit 2; fi
# (tools/pair_stamps.py belonged to the archived experiment, profiles/r05_negative/paired_tile_launches.md; it read g_pair_stamps of the -DCV_STAMPS build through scpn_debug_pair_stamps)
# r05 session 13: in-kernel stamps of the paired launches (alone on the chip)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s13; mkdir -p $O
WHICH=pre STAMPS_OUT=$O/pre.npy timeout -k 10 200 python tools/pair_stamps.py > $O/pair_stamps_pre.txt 2>&1; cat $O/pair_stamps_pre.txt
WHICH=post STAMPS_OUT=$O/post.npy timeout -k 10 200 python tools/pair_stamps.py > $O/pair_stamps_post.txt 2>&1; cat $O/pair_stamps_post.txt

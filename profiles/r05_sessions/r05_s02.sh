#!/bin/bash
# NOTE (round 6): this script drives switches / tools that are NOT in the tree at HEAD (SCPN_HIP_CURESERVE ): they exist only with the archived diffs of
# profiles/r05_negative/ applied (cu_reservation_for_the_coarse_kernel.md, issue_priority_s_setprio.md, paired_tile_launches.diff.txt) or, for the
# timing emulation, profiles/r05_emu_fuse_timing_experiment.diff.txt.  Without them the environment variables are ignored (a no-op A/B) and
# tools/pair_stamps.py / libscpn_hip.base.so are absent.  Kept as the record of how the round-5 tables were produced.
if [ -z "$SCPN_R05_DIFFS_APPLIED" ]; then echo "$0: archived experiment -- apply the round-5 diffs first (see the header), then export SCPN_R05_DIFFS_APPLIED=1" >&2; exit 2; fi
# r05 session 2: production chain of one stream group with and without the CU reserve (kernel trace of the fixed-iteration run)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05_s02; mkdir -p $O
for R in 0 16; do
  export SCPN_HIP_CURESERVE=$R
  rocprofv3 --kernel-trace --output-format csv -d $O/tr$R -o t -- python3 tools/fixed_iters_bench.py 200 1 > $O/run$R.log 2>&1 || exit 1
  f=$(find $O/tr$R -name "*kernel_trace.csv" | head -1)
  python3 tools/chain_table.py $f > $O/chain_cureserve$R.txt && rm -rf $O/tr$R
  cat $O/run$R.log $O/chain_cureserve$R.txt
done

#!/bin/bash
# NOTE (round 6): this script drives switches / tools that are NOT in the tree at HEAD (libscpn_hip.base.so ): they exist only with the archived diffs of
# profiles/r05_negative/ applied (cu_reservation_for_the_coarse_kernel.md, issue_priority_s_setprio.md, paired_tile_launches.diff.txt) or, for the
# timing emulation, profiles/r05_emu_fuse_timing_experiment.diff.txt.  Without them the environment variables are ignored (a no-op A/B) and
# tools/pair_stamps.py / libscpn_hip.base.so are absent.  Kept as the record of how the round-5 tables were produced.
if [ -z "$SCPN_R05_DIFFS_APPLIED" ]; then echo "$0: archived experiment -- apply the round-5 diffs first (see the header), then export SCPN_R05_DIFFS_APPLIED=1" >&2; exit 2; fi
# r05 session 17: A/B of a kernel change against the previous build (libscpn_hip.base.so = the committed tree's library)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s17; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "smooth or vcycle or operators or multigrid_solve or picard" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
SCPN_HIP_TILEGEO=48 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "smooth or vcycle or operators or multigrid_solve or picard" > $O/pytest48.log 2>&1 || { tail -30 $O/pytest48.log; exit 1; }
tail -1 $O/pytest.log; tail -1 $O/pytest48.log
tools/ab_fixed.sh scpn_fusion_core_amd/libscpn_hip.base.so scpn_fusion_core_amd/libscpn_hip.so scpn_fusion_core_amd/libscpn_hip.base.so scpn_fusion_core_amd/libscpn_hip.so > $O/ab.log 2>&1; cat $O/ab.log
for l in base.so so; do echo -n "single $l: "; SCPN_HIP_LIB=$PWD/scpn_fusion_core_amd/libscpn_hip.$l timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee $O/ab_single.log

#!/bin/bash
# NOTE (round 6): this script drives switches / tools that are NOT in the tree at HEAD (SCPN_HIP_PAIR ): they exist only with the archived diffs of
# profiles/r05_negative/ applied (cu_reservation_for_the_coarse_kernel.md, issue_priority_s_setprio.md, paired_tile_launches.diff.txt) or, for the
# timing emulation, profiles/r05_emu_fuse_timing_experiment.diff.txt.  Without them the environment variables are ignored (a no-op A/B) and
# tools/pair_stamps.py / libscpn_hip.base.so are absent.  Kept as the record of how the round-5 tables were produced.
if [ -z "$SCPN_R05_DIFFS_APPLIED" ]; then echo "$0: archived experiment -- apply the round-5 diffs first (see the header), then export SCPN_R05_DIFFS_APPLIED=1" >&2; exit 2; fi
# r05 session 10: paired tile launches (k_smooth_tile_pair; SCPN_HIP_PAIR bit 0: pre-smoothers of levels 1 + 2, bit 1: post-smoothers 2 + 1)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s10; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { cat $O/smoke.log; exit 1; }
SCPN_HIP_TILEGEO=48 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vcycle or multigrid_solve or picard or free_boundary" > $O/pytest_geo48.log 2>&1 || { tail -30 $O/pytest_geo48.log; exit 1; }
tail -2 $O/pytest_geo48.log
tools/ab_fixed_env.sh SCPN_HIP_PAIR 0 1 2 3 0 3 > $O/ab_pair.log 2>&1; cat $O/ab_pair.log
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-roofline --no-newton --no-configs > $O/bench_pair.json 2> $O/bench_pair.err || { tail -5 $O/bench_pair.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_s10/bench_pair.json"))
c=d["cpu_baseline"]
print(d["value"], d["ms_per_step"], c["psi_rel_linf_vs_gpu"], c["picard_iterations_equal_on_all_compared_members"], c["members_compared"])
PY

#!/bin/bash
# Run on the GPU box from the repo root (gpurun): rocprofv3 kernel-trace stats of the default bench command, then the two
# PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, kernel trace only, as MI355X_MICROARCH.md prescribes) on one stream
# group so that every launch covers all 32 members.  Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profiles.py
# turns them into the committed files under profiles/.
set -e
tag=${1:-r01g}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/k -o k -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.log 2>&1
SCPN_HIP_GROUPS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $out/pmc_fetch.log 2>&1
SCPN_HIP_GROUPS=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/w -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $out/pmc_write.log 2>&1
find $out -name "*.csv" | head -20

#!/bin/bash
# Run on the GPU box from the repo root (gpurun): rocprofv3 kernel-trace stats of the default bench command, then the PMC
# passes -- FETCH_SIZE, WRITE_SIZE (separate runs, as MI355X_MICROARCH.md prescribes) and two SQ passes (issue / LDS /
# wait counters + GRBM_GUI_ACTIVE) -- each with --kernel-trace only and the program directly after `--`, on one stream
# group so that every launch covers all members.  Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profiles.py
# turns them into the committed files under profiles/.
set -e
tag=${1:-r02a}
EXTRA="${@:2}"  # extra bench.py arguments for every pass, e.g. `--arith exact`
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/k -o k -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-newton --no-configs --no-arith-modes $EXTRA > $out/bench_under_rocprof.log 2>&1
echo "kernel stats done"
export SCPN_HIP_GROUPS=1
PMC_CMD="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-newton --no-configs --no-arith-modes $EXTRA"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -o f -- $PMC_CMD > $out/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/w -o w -- $PMC_CMD > $out/pmc_write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/s1 -o s1 -- $PMC_CMD > $out/pmc_sq1.log 2>&1
echo "sq1 done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/s2 -o s2 -- $PMC_CMD > $out/pmc_sq2.log 2>&1 || echo "sq2 pass failed (see $out/pmc_sq2.log)"
echo "sq2 done"
# summaries (small) into gpurun_out/summary_<tag>/ -- the raw counter files exceed what gpurun copies back
python3 tools/summarize_profiles.py $tag
mkdir -p gpurun_out/summary_$tag
cp profiles/${tag}_* profiles/pmc_traffic.json gpurun_out/summary_$tag/
cp $out/bench_under_rocprof.log gpurun_out/summary_$tag/
rm -rf $out
ls gpurun_out/summary_$tag

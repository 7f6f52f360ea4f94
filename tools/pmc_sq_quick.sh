#!/bin/bash
# SQ counters of one stream group on the fixed-iteration Picard bench (GPU box, from the repo root): tools/pmc_sq_quick.sh <tag>
set -e
tag=${1:-q}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sq_$tag
mkdir -p $out
export SCPN_HIP_GROUPS=1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/s1 -o s1 -- python3 tools/fixed_iters_bench.py 20 1 > $out/s1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/s2 -o s2 -- python3 tools/fixed_iters_bench.py 20 1 > $out/s2.log 2>&1
python3 tools/sq_summary.py $out

#!/bin/bash
# r06 session 13: source on the fly (SCPN_HIP_JFLY, contracted arithmetic, L-mode): tests against the oracle, per-iteration time on / off,
# kernels alone at full fill
set -o pipefail
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s13; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_fast_arith.py -x -q -m gpu > $O/fast.txt 2>&1 || { tail -60 $O/fast.txt; exit 1; }
tail -3 $O/fast.txt
export SCPN_HIP_ARITH=contracted
bash tools/ab_fixed_env.sh SCPN_HIP_JFLY 0 1 > $O/ab_jfly.txt 2>&1; cat $O/ab_jfly.txt
for T in 1 0; do
  export SCPN_HIP_JFLY=$T
  SCPN_HIP_GROUPS=1 rocprofv3 --kernel-trace --output-format csv -d $O/iso_$T -o t -- python3 tools/fixed_iters_bench.py 100 1 > $O/iso_run_$T.log 2>&1 || exit 1
  f=$(find $O/iso_$T -name "*kernel_trace.csv" | head -1)
  python3 tools/chain_table.py $f > $O/iso_chain_jfly$T.txt && rm -rf $O/iso_$T
  grep us/iteration $O/iso_run_$T.log; cat $O/iso_chain_jfly$T.txt
done

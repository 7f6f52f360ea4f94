#!/bin/bash
# r06 session 7: whole GPU suite with the operator-solver pool + [n][batch] f4 kernels; f4 layout before / after (whole call and kernel
# durations); dispatcher / polyglot rows with the pool
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s07; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/all.txt 2>&1 || { tail -40 $O/all.txt; exit 1; }
tail -3 $O/all.txt
for L in rows cols; do
  if [ $L = rows ]; then export SCPN_HIP_F4_LAYOUT=rows; else unset SCPN_HIP_F4_LAYOUT; fi
  python3 tools/f4_layout_bench.py 1024 32 256 4096 > $O/f4_$L.txt 2>&1 || { cat $O/f4_$L.txt; exit 1; }
  cat $O/f4_$L.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/f4prof_$L -o t -- python3 tools/f4_layout_bench.py 1024 256 > /dev/null 2>&1
  f=$(find $O/f4prof_$L -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY' | tee $O/f4_kernels_$L.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_gtsv", "k_current_diffusion", "k_transpose")):
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"]) / 1e3:10.2f}')
PY
  rm -rf $O/f4prof_$L
done
python3 - <<'PY' | tee $O/configs_rows.txt
import json, sys
sys.path.insert(0, ".")
from benchmarks.configs import measure_configs
import benchmarks.configs as bc
c = measure_configs(members=2)
print(json.dumps({k: c[k] for k in ("polyglot_picard_jacobi_513", "dispatcher_multigrid_solve_257")}, indent=1))
PY

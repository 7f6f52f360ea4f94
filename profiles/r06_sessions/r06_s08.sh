#!/bin/bash
# r06 session 8: register-carried dgtsv recurrence + parallel Crank-Nicolson assembly: transport tests, then both device layouts
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s08; mkdir -p $O
python3 -m pytest tests/test_gpu_transport.py tests/test_gpu_products.py -x -q -m gpu > $O/t.txt 2>&1 || { tail -40 $O/t.txt; exit 1; }
tail -3 $O/t.txt
for L in rows cols; do
  if [ $L = rows ]; then export SCPN_HIP_F4_LAYOUT=rows; else unset SCPN_HIP_F4_LAYOUT; fi
  python3 -m pytest tests/test_gpu_transport.py -x -q -m gpu > $O/t_$L.txt 2>&1 || { tail -40 $O/t_$L.txt; exit 1; }
  python3 tools/f4_layout_bench.py 1024 32 256 4096 > $O/f4_$L.txt 2>&1 || { cat $O/f4_$L.txt; exit 1; }
  cat $O/f4_$L.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/f4prof_$L -o t -- python3 tools/f4_layout_bench.py 1024 256 > /dev/null 2>&1
  f=$(find $O/f4prof_$L -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY' | tee $O/f4_kernels_$L.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_gtsv", "k_cd_", "k_transpose")):
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"]) / 1e3:10.2f}')
PY
  rm -rf $O/f4prof_$L
done

#!/bin/bash
# round-4 session 28: relative flux change of the transport coupling on the device (config 4)
set -e
out=gpurun_out/r04ab; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
timeout -k 10 900 python -m pytest tests/test_gpu_transport.py tests/test_abi_symbols.py -x -q 2>&1 | tail -5 | tee $out/tests.txt
python - <<'PY' | tee $out/config4.txt
import sys
sys.path.insert(0, ".")
from benchmarks.bench_transport_coupled_loop import coupled_steps
run = coupled_steps(513, 32, 3)
print(round(run["member_steps_per_s"], 1), "member-steps/s;", round(1e3 * run["s_per_coupled_step_ensemble"], 2), "ms per coupled step;", run["rows"][-1])
PY

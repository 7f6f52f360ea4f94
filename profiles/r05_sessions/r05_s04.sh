#!/bin/bash
# r05 session 4: the whole GPU suite after the clean-ups (retired switches, k_gm_cycle_small out of scratch, ADVICE fixes)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s04; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?
tail -8 $O/pytest_gpu.log
exit $rc

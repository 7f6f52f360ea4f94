#!/bin/bash
# r05 session 17: A/B of a kernel change against the previous build (libscpn_hip.base.so = the committed tree's library)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s17; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "smooth or vcycle or operators or multigrid_solve or picard" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
SCPN_HIP_TILEGEO=48 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "smooth or vcycle or operators or multigrid_solve or picard" > $O/pytest48.log 2>&1 || { tail -30 $O/pytest48.log; exit 1; }
tail -1 $O/pytest.log; tail -1 $O/pytest48.log
tools/ab_fixed.sh scpn_fusion_core_amd/libscpn_hip.base.so scpn_fusion_core_amd/libscpn_hip.so scpn_fusion_core_amd/libscpn_hip.base.so scpn_fusion_core_amd/libscpn_hip.so > $O/ab.log 2>&1; cat $O/ab.log
for l in base.so so; do echo -n "single $l: "; SCPN_HIP_LIB=$PWD/scpn_fusion_core_amd/libscpn_hip.$l timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done | tee $O/ab_single.log

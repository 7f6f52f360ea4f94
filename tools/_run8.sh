set -e
cd $GRAFT_REPO_ROOT
L=$PWD/scpn_fusion_core_amd
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/parity8.log 2>&1 || { tail -30 gpurun_out/parity8.log; exit 1; }
tail -2 gpurun_out/parity8.log
SCPN_HIP_TILEREG=1 SCPN_HIP_TILEREG_TINYB=0 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/parity8b.log 2>&1 || { tail -30 gpurun_out/parity8b.log; exit 1; }
tail -2 gpurun_out/parity8b.log
for m in 1 2 4 8 12; do echo "--- m$m"; timeout -k 10 200 python tools/single_bench.py 513 3 $m; done
echo "--- m1 tiny geometry off"; SCPN_HIP_TILEREG_TINYB=0 timeout -k 10 200 python tools/single_bench.py 513 3 1
for i in 1 2; do
  echo -n "tile 2 groups: "; timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  echo -n "reg32 2 groups: "; SCPN_HIP_TILEREG=1 timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  echo -n "reg40 2 groups: "; SCPN_HIP_LIB=$L/libscpn_hip.reg40.so SCPN_HIP_TILEREG=1 timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  echo -n "reg48 2 groups: "; SCPN_HIP_LIB=$L/libscpn_hip.reg48.so SCPN_HIP_TILEREG=1 timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
  echo -n "reg32 3 groups: "; SCPN_HIP_GROUPS=3 SCPN_HIP_TILEREG=1 timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
done

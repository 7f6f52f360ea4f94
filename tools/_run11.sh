set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -x -q > gpurun_out/parity11.log 2>&1 || { tail -30 gpurun_out/parity11.log; exit 1; }
tail -2 gpurun_out/parity11.log
for m in 1 4 12; do echo "--- m$m"; timeout -k 10 200 python tools/single_bench.py 513 3 $m; done
for i in 1 2 3; do
  echo -n "tile 2 groups: "; timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
done

set -e
cd $GRAFT_REPO_ROOT
for pad in 0 8 24 56 120 248 504; do
  echo -n "pad $pad: "; SCPN_HIP_PITCH_PAD=$pad timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
done
echo -n "pad 0: "; timeout -k 10 200 python tools/fixed_iters_bench.py 200 3

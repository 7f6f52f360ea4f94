set -e
cd $GRAFT_REPO_ROOT
L=$PWD/scpn_fusion_core_amd
export SCPN_HIP_TILEREG=0
for v in lz32w6 lz32w8 lz40w8; do
  SCPN_HIP_LIB=$L/libscpn_hip.$v.so bash tools/pmc_sq_quick.sh $v > gpurun_out/sq_$v.txt 2>&1
  echo "== $v"; grep "k_smooth_tile" gpurun_out/sq_$v.txt | head -2 | cut -c1-300
done
for i in 1 2; do for v in lz32w6 lz32w8 lz40w8; do
  echo -n "$v: "; SCPN_HIP_LIB=$L/libscpn_hip.$v.so timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
done; done

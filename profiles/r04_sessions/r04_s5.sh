#!/bin/bash
# round-4 session 5: host-side cost of a launch with 2 / 4 stream groups (rocprofv3 --hip-trace --stats)
out=gpurun_out/r04e; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for g in 2 4; do
  export SCPN_HIP_GROUPS=$g
  rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $out/g$g -o t -- python3 tools/fixed_iters_bench.py 100 1 > $out/run_g$g.txt 2>&1
  find $out/g$g -name "*hip_api_stats.csv" -exec head -14 {} \; > $out/hip_stats_g$g.txt
  find $out/g$g -name "*kernel_stats.csv" -exec head -14 {} \; > $out/kernel_stats_g$g.txt
  rm -rf $out/g$g
done
grep "us/iteration" $out/run_g2.txt; cat $out/hip_stats_g2.txt; grep "us/iteration" $out/run_g4.txt; cat $out/hip_stats_g4.txt $out/kernel_stats_g4.txt

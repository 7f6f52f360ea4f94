#!/bin/bash
# A/B libraries on the fixed-iteration Picard benchmark: tools/ab_fixed.sh libA libB ...
for i in 1 2; do
 for lib in "$@"; do
  echo -n "$lib: "; SCPN_HIP_LIB=$PWD/$lib timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
 done
done

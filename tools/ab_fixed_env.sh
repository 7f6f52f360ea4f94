#!/bin/bash
# fixed-iteration Picard benchmark under different values of one environment variable: tools/ab_fixed_env.sh VAR v1 v2 ...
var=$1; shift
for i in 1 2; do
 for val in "$@"; do
  echo -n "$var=$val: "; env $var=$val timeout -k 10 200 python tools/fixed_iters_bench.py 200 3
 done
done

#!/bin/bash
# A/B one environment switch alternately on the same box: tools/ab_env.sh VAR valA valB
var=$1; shift
for i in 1 2 3; do
 for val in "$@"; do
  env $var=$val timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$var=$val', d['value'], d['ms_per_step'], d['config']['picard_iterations_per_member'])"
 done
done

#!/bin/bash
# A/B two libraries alternately on the same box
for i in 1 2 3; do
 for lib in "$@"; do
  SCPN_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"
 done
done

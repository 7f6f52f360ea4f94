#!/bin/bash
export SCPN_HIP_LIB=$PWD/build/libscpn_hip_exp.so  # hipcc ... -DTILE_EXPERIMENTS -o build/libscpn_hip_exp.so scpn_fusion_core_amd/csrc/scpn_hip.hip
for t in "1,1,1" "3,1,1" "4,1,1" "5,1,1" "6,1,1" "7,1,1" "8,1,1" "1,3,1" "1,6,6" "1,1,1"; do
  SCPN_HIP_TILE=$t timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); f=d['roofline']['families_ms']
print('tile $t', d['value'], d['ms_per_step'], ' '.join(f\"{k.replace('smooth_tile_','')}={1e3*v['ms']/v['launches']:.1f}\" for k,v in f.items() if k.startswith('smooth_tile')))"
done

#!/bin/bash
# round-4 session 18: headline check + SUMFUSE re-check with the shorter chain
set -e
out=gpurun_out/r04s; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('iteration'), d['configs'] if 'configs' in d else '')" | cut -c1-1500
for i in 1 2 3; do
  for g in 0 1; do echo -n "SUMFUSE=$g: "; SCPN_HIP_SUMFUSE=$g timeout -k 10 200 python tools/fixed_iters_bench.py 200 3; done
done 2>&1 | tee $out/ab.txt

#!/bin/bash
# r05 session 19: wall time of the default bench command (what the driver runs), and of the N = 1 driver form
set -o pipefail
O=gpurun_out/r05_s19; mkdir -p $O
SECONDS=0; python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$? default bench wall: ${SECONDS}s"
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 2 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; echo "rc=$? driver-form bench wall: ${SECONDS}s"
python -c "
import json
for f in ('bench_default','bench_driver_form'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['steps'], d['roofline']['frac'], d['cpu_baseline']['value'], d['parity']['pass'], d['strong_scaling_n1']['value'])"

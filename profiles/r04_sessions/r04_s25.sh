#!/bin/bash
# round-4 session 25: full GPU suite + the bench line with the updated byte accounting
set -e
out=gpurun_out/r04y; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $out/tests.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/r04e_bench.json 2> $out/bench.err
python -c "
import json; d=json.load(open('$out/r04e_bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['achieved'], r['traffic'], r['alg_bytes_per_launch'], r['per_level'], r['iteration']['frac'], r['one_group_all_members'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"

for i in 1 2; do for ce in ${CES:-8 16 24}; do timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --check-every $ce 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('check_every $ce', d['value'], d['ms_per_step'])"; done; done

#!/bin/bash
# r05 session 16: the N > 1 launch paths of bench.py rehearsed on one GPU after this round's changes (plan_ensemble, --total-members)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s16; mkdir -p $O
# (1) RCCL path on ONE rank, weak and strong: communicator, padded gather, allreduce barrier
SCPN_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-newton --no-configs > $O/force_dist_weak.json 2> $O/force_dist_weak.err; echo "rc=$?"; cut -c1-700 $O/force_dist_weak.json
SCPN_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --total-members 48 --no-cpu-baseline --no-roofline --no-newton --no-configs > $O/force_dist_strong.json 2> $O/force_dist_strong.err; echo "rc=$?"; cut -c1-700 $O/force_dist_strong.json
# (2) two spawned ranks sharing device 0: RCCL refuses the duplicate device on both, the documented fallback line (valid: false, exit 3)
SCPN_BENCH_DEVICE=0 timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-newton --no-configs > $O/two_ranks_weak.json 2> $O/two_ranks_weak.err; echo "rc=$?"; cut -c1-500 $O/two_ranks_weak.json; tail -2 $O/two_ranks_weak.err
SCPN_BENCH_DEVICE=0 timeout -k 10 400 python bench.py --gpus 2 --total-members 63 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-newton --no-configs > $O/two_ranks_strong.json 2> $O/two_ranks_strong.err; echo "rc=$?"; cut -c1-500 $O/two_ranks_strong.json; tail -2 $O/two_ranks_strong.err
# (3) the driver's form of the launch with torch.distributed.run, two ranks on device 0
SCPN_BENCH_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-newton --no-configs > $O/torchrun2.json 2> $O/torchrun2.err; echo "rc=$?"; grep '^{' $O/torchrun2.json | cut -c1-400
exit 0

#!/bin/bash
# round-4 session 23: coarse-kernel stamps of two builds on ONE box + single-equilibrium A/B repeated
set -e
out=gpurun_out/r04x; mkdir -p $out
export GPU_MAX_HW_QUEUES=8
P=$PWD/scpn_fusion_core_amd
timeout -k 10 200 python tools/cv_stamps.py > $out/new.txt 2>&1
# (round 5: the stamped variant is passed to tools/cv_stamps.py through its environment instead of rewriting the tracked tool with sed)
timeout -k 10 200 env STAMPS_VARIANT=stamps_prev STAMPS_DEFINES="" python tools/cv_stamps.py > $out/prev.txt 2>&1
paste $out/new.txt $out/prev.txt | tee $out/cv_stamps.txt
for i in 1 2 3; do for v in so prev.so; do echo -n "single $v: "; SCPN_HIP_LIB=$P/libscpn_hip.$v timeout -k 10 200 python tools/fixed_iters_bench.py 200 3 513 1; done; done | tee $out/ab.txt

#!/bin/bash
# round-4 session 10: in-kernel stamps of the pre-smoother's tiles on the 129^2 and 257^2 levels of a 16-member V-cycle
out=gpurun_out/r04j; mkdir -p $out
export SCPN_HIP_TILEGEO=48 STAMPS_MODE=defect STAMPS_B=16
STAMPS_DEFINES="STAMP_NZ_MIN=100 STAMP_NZ_MAX=200 STAMP_STRIDE=19" timeout -k 10 300 python tools/tile_stamps.py > $out/stamps_L2.txt 2>&1
STAMPS_DEFINES="STAMP_NZ_MIN=200 STAMP_NZ_MAX=300 STAMP_STRIDE=53" timeout -k 10 300 python tools/tile_stamps.py > $out/stamps_L1.txt 2>&1
STAMPS_DEFINES="STAMP_STRIDE=169" timeout -k 10 300 python tools/tile_stamps.py > $out/stamps_L0.txt 2>&1
tail -14 $out/stamps_L2.txt $out/stamps_L1.txt $out/stamps_L0.txt

#!/bin/bash
# (tools/pair_stamps.py belonged to the archived experiment, profiles/r05_negative/paired_tile_launches.md; it read g_pair_stamps of the -DCV_STAMPS build through scpn_debug_pair_stamps)
# r05 session 13: in-kernel stamps of the paired launches (alone on the chip)
set -o pipefail
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05_s13; mkdir -p $O
WHICH=pre STAMPS_OUT=$O/pre.npy timeout -k 10 200 python tools/pair_stamps.py > $O/pair_stamps_pre.txt 2>&1; cat $O/pair_stamps_pre.txt
WHICH=post STAMPS_OUT=$O/post.npy timeout -k 10 200 python tools/pair_stamps.py > $O/pair_stamps_post.txt 2>&1; cat $O/pair_stamps_post.txt

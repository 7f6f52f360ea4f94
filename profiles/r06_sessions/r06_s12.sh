#!/bin/bash
# r06 session 12: tile geometry under contracted arithmetic (review item 2): 48 x 8 (production) against 56 x 14 (window 40 / 44 rows: 13 / 12
# tile rows at 513 instead of 17 / 15; 4 rows per wave, 2 workgroups per CU), 64 x 16 and 40 x 10.  Experiment libraries:
#   _lib.build(variant="g56x14", defines=("TILE_BIG_LZ=56", "TILE_BIG_NW=14", "TILE_WPE=7")), g64x16: 64 / 16 / 8, g40x10: 40 / 10 / 6
set -o pipefail
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s12; mkdir -p $O
export SCPN_HIP_ARITH=contracted
bash tools/ab_fixed.sh scpn_fusion_core_amd/libscpn_hip.so scpn_fusion_core_amd/libscpn_hip.g56x14.so scpn_fusion_core_amd/libscpn_hip.g64x16.so scpn_fusion_core_amd/libscpn_hip.g40x10.so > $O/ab_geometry.txt 2>&1
cat $O/ab_geometry.txt
# bit-identity of the exact variants: one V-cycle + a short Picard solve against the production library
python3 - <<'PY' | tee $O/geometry_bit_identity.txt
import os, subprocess, sys, json
code = r'''
import sys, numpy as np, hashlib
sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import iter_like_cfg
from scpn_fusion_core_amd.ensemble import EnsembleSolver
cfg = iter_like_cfg(513, tol=1e-300, max_iter=12)
with EnsembleSolver(cfg, batch=16, arithmetic="exact") as es:
    es.solve_equilibrium()
    print(hashlib.sha256(es.get_psi().tobytes()).hexdigest())
'''
out = {}
for lib in ("libscpn_hip.so", "libscpn_hip.g56x14.so", "libscpn_hip.g64x16.so", "libscpn_hip.g40x10.so"):
    env = dict(os.environ, SCPN_HIP_LIB=os.path.abspath("scpn_fusion_core_amd/" + lib), SCPN_HIP_ARITH="exact")
    out[lib] = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip()[-64:]
print(json.dumps(out, indent=1))
print("exact arithmetic, 12 Picard iterations, 16 members, 513^2: all geometries bit-identical:", len(set(out.values())) == 1)
PY

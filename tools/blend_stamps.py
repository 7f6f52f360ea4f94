"""s_memrealtime stamps of eight sampled workgroups of k_blend_tile (-DCV_STAMPS experiment build) during a 32-member Picard run:
start -> tile staged (loads consumed, blend, LDS write) -> barrier passed -> residual / gradient done -> reductions -> end."""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
from scpn_fusion_core_amd import _lib

os.environ["SCPN_HIP_LIB"] = str(_lib.build(variant="stamps_blend", defines=("CV_STAMPS",)))
os.environ["SCPN_HIP_GROUPS"] = os.environ.get("SCPN_HIP_GROUPS", "1")
import numpy as np  # noqa: E402

from scpn_fusion_core_amd.cases import iter_like_cfg  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

B, n = int(os.environ.get("STAMPS_B", "32")), 513
cfg = iter_like_cfg(n, tol=1e-300, max_iter=40)
with EnsembleSolver(cfg, batch=B) as es:
    es.solve_equilibrium(check_every=8)
    es.sync()
    L = C.CDLL(os.environ["SCPN_HIP_LIB"])
    buf = (C.c_ulonglong * 128)()
    L.scpn_debug_tile_stamps(buf)
names = ["staged", "barrier", "resid+grad", "reduce1", "reduce2"]
print("wg   " + " ".join(f"{x:>10s}" for x in names) + "      total")
for g in range(8):
    st = list(buf)[g * 16:(g + 1) * 16]
    if st[0] == 0:
        continue
    d = [(st[j] - st[j - 1]) * 0.01 for j in range(1, 6)]
    print(f"{g:2d}   " + " ".join(f"{x:10.2f}" for x in d) + f" {(st[5] - st[0]) * 0.01:10.2f}")

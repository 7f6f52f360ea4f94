"""s_memrealtime stamps inside the in-LDS coarse V-cycle kernel (one workgroup): phase durations of one launch.
usage: [STAMPS_VARIANT=name STAMPS_DEFINES="A=1 B"] python tools/cv_stamps.py"""
import os, sys, ctypes as C
sys.path.insert(0, ".")
from scpn_fusion_core_amd import _lib
# experiment build, made on first use.  STAMPS_VARIANT / STAMPS_DEFINES select another stamped build (e.g. STAMPS_VARIANT=stamps_cvb0
# STAMPS_DEFINES="CV_BLOCKS=0": the column-pair form of the 17^2..65^2 levels) -- session scripts pass these instead of editing this file
variant = os.environ.get("STAMPS_VARIANT", "stamps_plain")
defines = ("CV_STAMPS", *os.environ.get("STAMPS_DEFINES", "").split())
os.environ["SCPN_HIP_LIB"] = str(_lib.build(variant=variant, defines=defines))
os.environ["SCPN_HIP_GROUPS"] = "1"
import numpy as np
from scpn_fusion_core_amd.ensemble import EnsembleSolver
from scpn_fusion_core_amd.cases import iter_like_cfg, bench_problem
n, B = 513, 16
es = EnsembleSolver(iter_like_cfg(n), batch=B)
psi0, src, rr, r, z = bench_problem(n, n)
es.set_psi(np.broadcast_to(psi0, (B, n, n)).copy()); es.set_source(np.broadcast_to(src, (B, n, n)).copy())
for _ in range(3): es.vcycle(1.0)
L = C.CDLL(os.environ["SCPN_HIP_LIB"])
buf = (C.c_ulonglong * 64)()
L.scpn_debug_cv_stamps(buf)
st = list(buf)
t0 = st[0]
names = {0: "start", 1: "loaded", 20: "base done", 40: "end"}
for l in range(5):
    names[2 + 2 * l] = f"pre-smooth L{l}"; names[3 + 2 * l] = f"defect+restrict L{l}"
    names[21 + 2 * l] = f"prolong L{l}"; names[22 + 2 * l] = f"post-smooth L{l}"
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 20, 27, 28, 25, 26, 23, 24, 21, 22, 40]
prev = t0
for i in order:
    if st[i] == 0: continue
    print(f"{names.get(i, i):24s} +{(st[i]-prev)*0.01:7.2f} us   (t={(st[i]-t0)*0.01:7.2f})")
    prev = st[i]

import os, sys, ctypes as C
os.environ["SCPN_HIP_LIB"] = os.path.abspath("build/libscpn_hip_stamps.so")
os.environ["SCPN_HIP_GROUPS"] = "1"
sys.path.insert(0, ".")
import numpy as np
from scpn_fusion_core_amd.ensemble import EnsembleSolver
from tests._problems import iter_like_cfg, bench_problem
for n, B in ((513, 32), (257, 16)):
    es = EnsembleSolver(iter_like_cfg(n), batch=B)
    psi0, src, rr, r, z = bench_problem(n, n)
    es.set_psi(np.broadcast_to(psi0, (B, n, n)).copy()); es.set_source(np.broadcast_to(src, (B, n, n)).copy())
    for _ in range(5): es.rb_sor(1.3, 3)
    L = C.CDLL(os.environ["SCPN_HIP_LIB"])
    buf = (C.c_ulonglong * 64)()
    L.scpn_debug_tile_stamps(buf)
    st = list(buf)
    names = ["start", "loaded"] + [f"pass {i}" for i in range(6)] + ["", "", "stored"]
    prev = st[0]
    print(f"--- n={n} B={B} tile (1,1) of member 0")
    for i in (0, 1, 2, 3, 4, 5, 6, 7, 10):
        print(f"{names[i]:10s} +{(st[i]-prev)*0.01:6.2f} us  (t={(st[i]-st[0])*0.01:6.2f})")
        prev = st[i]
    es.close()

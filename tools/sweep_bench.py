"""Per-launch time of the resident smoother for S = 1, 2, 3 sweeps per launch and several batch sizes."""
import json, sys
sys.path.insert(0, ".")
import numpy as np
from scpn_fusion_core_amd.ensemble import EnsembleSolver
from scpn_fusion_core_amd.cases import iter_like_cfg, bench_problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
for B in (32, 64):
    cfg = iter_like_cfg(n)
    es = EnsembleSolver(cfg, batch=B)
    psi0, src, rr, r, z = bench_problem(n, n)
    es.set_psi(np.broadcast_to(psi0, (B, n, n)).copy()); es.set_source(np.broadcast_to(src, (B, n, n)).copy())
    for S in (1, 2, 3):
        for _ in range(20): es.rb_sor(1.3, S)
        es.profile(True)
        for _ in range(100): es.rb_sor(1.3, S)
        rep = es.profile_report(); es.profile(False)
        k = [v for kk, v in rep.items() if kk.startswith("smooth_tile")][0]
        print(json.dumps({"n": n, "B": B, "S": S, "us_per_launch": round(1e3 * k["ms"] / k["launches"], 2)}))
    es.close()

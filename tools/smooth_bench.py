"""Time of one plain tile-smoother launch (3 RB-SOR sweeps, 513x513, B members, one stream group): smooth_bench.py [B] [calls]"""
import os
import sys
import time

os.environ.setdefault("SCPN_HIP_GROUPS", "1")
sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402
from scpn_fusion_core_amd.cases import bench_problem, iter_like_cfg  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 513
es = EnsembleSolver(iter_like_cfg(n), batch=B)
psi0, src, rr, r, z = bench_problem(n, n)
es.set_psi(np.broadcast_to(psi0, (B, n, n)).copy())
es.set_source(np.broadcast_to(src, (B, n, n)).copy())
es.rb_sor(1.0, 3)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    es.rb_sor(1.0, 3 * calls)  # `calls` launches of 3 fused sweeps each, one synchronisation at the end
    best = min(best, time.perf_counter() - t0)
print(f"B={B}: {1e6 * best / calls:7.2f} us per 3-sweep launch")
es.close()

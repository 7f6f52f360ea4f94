"""Latency of ONE 513x513 free-boundary equilibrium (BASELINE config 3 shape: unbatched): ms per solve, Picard and Newton lanes.
usage: single_bench.py [n] [reps] [members]"""
import sys
import time

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import diiid_like_cfg, iter_like_cfg  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
members = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for name, cfg in (("iter_like", iter_like_cfg(n, tol=1e-8, max_iter=1000)), ("diiid_like", diiid_like_cfg(n))):
    with EnsembleSolver(cfg, batch=members) as es:
        ts = []
        for _ in range(reps + 1):
            es.fill_psi(0.0)
            t = time.perf_counter()
            r = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
            es.sync()
            ts.append(time.perf_counter() - t)
    its = sum(r[0]["picard_iterations"])
    print(f"{name:10s} picard: {1e3 * min(ts[1:]):7.2f} ms  ({its} iterations, {1e6 * min(ts[1:]) / its:.1f} us/iteration, converged {r[0]['converged']})")

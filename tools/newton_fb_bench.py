"""Free-boundary ensemble (the bench workload) with the Newton inner solver (finite-difference Jacobian, multigrid-preconditioned
GMRES, full steps) beside the Picard solve: eq/s, iteration counts, psi difference.  usage: newton_fb_bench.py [n] [batch] [reps]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import iter_like_cfg, scan_member_table  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = iter_like_cfg(n, tol=1e-8, max_iter=1000)
currents, targets = scan_member_table(cfg, B)
out = {}
for name, extra in (("picard", {}), ("newton_fd_mg", dict(solver_method="newton", gmres_preconditioner="multigrid", newton_jacobian="finite_difference",
                                                           newton_damping=1.0, max_iterations=60))):
    c = {**cfg, "solver": {**cfg["solver"], **extra}}
    with EnsembleSolver(c, batch=B, coil_currents=currents, current_targets=targets) as es:
        best = 1e9
        for _ in range(reps + 1):
            es.fill_psi(0.0)
            t = time.perf_counter()
            res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
            best = min(best, time.perf_counter() - t)
        out[name] = (es.get_psi(), res)
    its = [sum(r["picard_iterations"]) for r in res]
    print(f"{name:14s}: {B / best:8.1f} eq/s  ({best * 1e3:.1f} ms)  converged {sum(r['converged'] for r in res)}/{B}  outer {res[0]['outer_iterations']}  "
          f"inner iterations {min(its)}..{max(its)}  per outer {res[0]['picard_iterations']}")
pa, pb = out["picard"][0], out["newton_fd_mg"][0]
print("max rel |psi_newton - psi_picard| =", float(np.abs(pa - pb).max() / np.abs(pa).max()))

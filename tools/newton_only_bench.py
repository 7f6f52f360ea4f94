"""The Newton lane alone on the bench ensemble (finite-difference Jacobian, V-cycle-preconditioned GMRES, full steps): eq/s.
usage: newton_only_bench.py [n] [batch] [reps]"""
import sys
import time

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import iter_like_cfg, scan_member_table  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = iter_like_cfg(n, tol=1e-8, max_iter=1000)
currents, targets = scan_member_table(cfg, B)
c = {**cfg, "solver": {**cfg["solver"], "solver_method": "newton", "gmres_preconditioner": "multigrid", "newton_jacobian": "finite_difference",
                       "newton_damping": 1.0, "max_iterations": 60, "newton_warmup_steps": 2}}
with EnsembleSolver(c, batch=B, coil_currents=currents, current_targets=targets) as es:
    ts = []
    for _ in range(reps + 1):
        es.fill_psi(0.0)
        t = time.perf_counter()
        res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
        ts.append(time.perf_counter() - t)
its = [sum(r["picard_iterations"]) for r in res]
best = min(ts[1:])
print(f"newton_fd_mg: {B / best:8.1f} eq/s ({best * 1e3:.2f} ms; runs {[round(1e3 * t, 1) for t in ts]}) converged {sum(r['converged'] for r in res)}/{B} "
      f"outer {res[0]['outer_iterations']} inner {min(its)}..{max(its)}")

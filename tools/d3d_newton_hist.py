"""Why the multigrid / finite-difference Newton lane needs 28 inner iterations on the DIII-D-like 513^2 case (BASELINE config 3) where an
ITER-like member needs 15: residual histories of the lane from the seeded start for a grid of knobs (warm-up steps, forcing term, Krylov
size, line search), then the whole free-boundary solve timed for the promising ones.  GPU box, repo root.
usage: d3d_newton_hist.py [n]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import diiid_like_cfg, iter_like_cfg  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513


def lane(cfg, **kw):
    sol = {**cfg["solver"], "solver_method": "newton", "gmres_preconditioner": "multigrid", "newton_jacobian": "finite_difference",
           "newton_damping": 1.0, "max_iterations": 60, "newton_warmup_steps": 2}
    sol.update(kw)
    return {**cfg, "solver": sol}


def hist(cfg, tag):
    with EnsembleSolver(cfg, batch=1) as es:
        r = es.solve_newton()[0]
        print(f"  {tag}: its {r['iterations']} conv {r['converged']} krylov columns {r['gmres_inner_iterations']}  ls rejects {r['newton_line_search_rejects']}")
        print("     diff:", " ".join(f"{v:.1e}" for v in r["residual_history"]))
        print("     gs  :", " ".join(f"{v:.1e}" for v in r["gs_residual_history"]), flush=True)


def timed(cfg, tag):
    with EnsembleSolver(cfg, batch=1) as es:
        best = None
        for _ in range(3):
            es.fill_psi(0.0)
            t = time.perf_counter()
            r = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)[0]
            es.sync()
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        psi = es.get_psi(0)
    print(f"  {tag}: {1e3 * best:.2f} ms, converged {r['converged']}, outer {r['outer_iterations']}, inner {r['picard_iterations']}", flush=True)
    return psi


for name, base in (("DIII-D-like", diiid_like_cfg(n)), ("ITER-like", iter_like_cfg(n, tol=1e-8, max_iter=1000))):
    print(name, "fixed-boundary solve from the seeded start (vacuum boundary):")
    for kw in ({}, {"newton_warmup_steps": 5}, {"newton_warmup_steps": 10}, {"newton_warmup_steps": 20}, {"gmres_rtol": 1e-2}, {"gmres_rtol": 1e-4},
               {"gmres_restart": 16}, {"newton_line_search": True}, {"newton_warmup_steps": 10, "gmres_rtol": 1e-2}):
        hist(lane(base, **kw), str(kw or "bench settings"))
print("whole free-boundary solve, DIII-D-like:")
d3d = diiid_like_cfg(n)
ref = timed(d3d, "Picard")
for kw in ({}, {"newton_warmup_steps": 5}, {"newton_warmup_steps": 10}, {"newton_warmup_steps": 20}, {"gmres_rtol": 1e-2}, {"newton_warmup_steps": 10, "gmres_rtol": 1e-2},
           {"newton_line_search": True}):
    psi = timed(lane(d3d, **kw), "Newton " + str(kw or "bench settings"))
    print(f"      psi vs Picard: {np.abs(psi - ref).max() / np.abs(ref).max():.2e}")

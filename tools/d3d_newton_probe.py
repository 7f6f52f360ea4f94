"""ONE DIII-D-like 513^2 free-boundary equilibrium with the multigrid / finite-difference Newton lane: ms per solve and inner iterations per
outer iteration (the lane is sensitive to the last bits of the current integral: profiles/r04_experiments.md).  GPU box, repo root."""
import sys, time
sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import diiid_like_cfg
from scpn_fusion_core_amd.ensemble import EnsembleSolver
cfg = diiid_like_cfg(513)
c = {**cfg, "solver": {**cfg["solver"], "solver_method": "newton", "gmres_preconditioner": "multigrid", "newton_jacobian": "finite_difference",
                       "newton_damping": 1.0, "max_iterations": 60, "newton_warmup_steps": 2}}
with EnsembleSolver(c, batch=1) as one:
    for rep in range(3):
        one.fill_psi(0.0)
        t = time.perf_counter(); r = one.solve_free_boundary(max_outer_iter=20, tol=1e-4); one.sync(); dt = time.perf_counter() - t
        print(round(dt * 1e3, 2), r[0]["converged"], r[0]["outer_iterations"], r[0]["picard_iterations"], r[0]["final_diff"])

"""Residual histories of the Newton lane (finite-difference Jacobian, multigrid GMRES) on the bench members: first solve from the
seed, then a second solve from the converged state (what the second free-boundary outer iteration does)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import iter_like_cfg, scan_member_table  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rt = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
cfg = iter_like_cfg(n, tol=1e-8, max_iter=1000)
cfg["solver"].update(solver_method="newton", gmres_preconditioner="multigrid", newton_jacobian="finite_difference", newton_damping=1.0,
                     max_iterations=60, gmres_restart=6, newton_warmup_steps=w, gmres_rtol=rt, gmres_atol=0.0)
currents, targets = scan_member_table(cfg, B)
with EnsembleSolver(cfg, batch=B, coil_currents=currents, current_targets=targets) as es:
    res = es.solve_free_boundary(max_outer_iter=1, tol=1e-4)
    bc = es.get_psi()
    for tag, pres in (("from converged", True), ("again", True)):
        r = es.solve_newton(preserve_initial_state=pres, boundary_flux=bc)
        for b in (0, B - 1):
            print(tag, b, "its", r[b]["iterations"], "conv", r[b]["converged"], "inner", r[b].get("gmres_inner_iterations"),
                  "hist", " ".join(f"{v:.2e}" for v in r[b]["residual_history"]))
    es.fill_psi(0.0)
    r = es.solve_newton(preserve_initial_state=False, boundary_flux=bc)
    for b in (0, B - 1):
        print("from seed", b, "its", r[b]["iterations"], "conv", r[b]["converged"], "inner", r[b].get("gmres_inner_iterations"),
              "hist", " ".join(f"{v:.2e}" for v in r[b]["residual_history"]), "\n   gs", " ".join(f"{v:.2e}" for v in r[b]["gs_residual_history"]))

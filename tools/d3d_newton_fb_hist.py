"""Residual history of the FIRST inner solve of the DIII-D-like free-boundary loop with the Newton lane (same start as the device loop:
psi = 0 with the Dirichlet ring of the coil flux, preserve_initial_state), for a few knobs.  GPU box, repo root."""
import sys

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import diiid_like_cfg  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
base = diiid_like_cfg(n)
for kw in ({}, {"newton_warmup_steps": 0}, {"newton_warmup_steps": 1}, {"newton_warmup_steps": 4}, {"gmres_rtol": 1e-2}, {"gmres_restart": 16, "gmres_maxiter": 2}, {"newton_damping": 0.7}):
    sol = {**base["solver"], "solver_method": "newton", "gmres_preconditioner": "multigrid", "newton_jacobian": "finite_difference",
           "newton_damping": 1.0, "max_iterations": 60, "newton_warmup_steps": 2, **kw}
    with EnsembleSolver({**base, "solver": sol}, batch=1) as es:
        es.fill_psi(0.0)
        es.solve_free_boundary(max_outer_iter=1, tol=1e-4)
        bc = es.get_boundary_flux()
        psi = np.zeros_like(bc)
        psi[:, 0, :], psi[:, -1, :], psi[:, :, 0], psi[:, :, -1] = bc[:, 0, :], bc[:, -1, :], bc[:, :, 0], bc[:, :, -1]
        es.set_psi(psi)
        r = es.solve_newton(preserve_initial_state=True, boundary_flux=bc)[0]
        print(f"{kw or 'bench settings'}: its {r['iterations']} conv {r['converged']} krylov columns {r['gmres_inner_iterations']} nonconverged linear solves {r['gmres_nonconverged_count']}")
        print("     diff:", " ".join(f"{v:.1e}" for v in r["residual_history"]))
        print("     gs  :", " ".join(f"{v:.1e}" for v in r["gs_residual_history"]), flush=True)

"""Newton-Kantorovich with the multigrid-preconditioned GMRES against the Picard solution: steps, inner iterations, psi error.
usage: newton_mg_experiment.py [n] [batch]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import iter_like_cfg  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 129
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = iter_like_cfg(n, tol=1e-8, max_iter=1000)
with EnsembleSolver(cfg, batch=B) as es:
    t = time.perf_counter()
    rp = es.solve_equilibrium()
    tp = time.perf_counter() - t
    psi_p = es.get_psi(0)
print(f"picard: {rp[0]['iterations']} iterations, converged {rp[0]['converged']}, {tp * 1e3:.1f} ms for {B} members")
for pre in ("multigrid",):
    for ls, damp, jac in ((False, 0.5, "profile"), (False, 1.0, "finite_difference"), (True, 1.0, "finite_difference"), (False, 0.7, "finite_difference")):
        c = iter_like_cfg(n, method="newton", tol=1e-8, max_iter=60)
        c["solver"].update(gmres_preconditioner=pre, newton_line_search=ls, newton_damping=damp, newton_jacobian=jac)
        with EnsembleSolver(c, batch=B) as es:
            es.solve_equilibrium()  # warm-up of the library
            es.fill_psi(0.0)
            t = time.perf_counter()
            r = es.solve_equilibrium()
            tn = time.perf_counter() - t
            psi = es.get_psi(0)
        r0 = r[0]
        err = np.abs(psi - psi_p).max() / np.abs(psi_p).max()
        print(f"newton {pre} ls={ls} damping={damp} jac={jac}: iterations {r0['iterations']} converged {r0['converged']} residual {r0['residual']:.3e} gs {r0['gs_residual']:.3e} "
              f"gmres inner {r0.get('gmres_inner_iterations')} nonconv {r0.get('gmres_nonconverged')} | rel err vs picard {err:.3e} | {tn * 1e3:.1f} ms")
        print("   hist:", " ".join(f"{v:.2e}" for v in r0["residual_history"][:40]))

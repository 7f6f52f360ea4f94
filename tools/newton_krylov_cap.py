"""The Krylov cap of the multigrid / finite-difference Newton lane (gmres_restart x gmres_maxiter; default 8 x 1): ONE DIII-D-like 513^2
free-boundary equilibrium (BASELINE config 3) and the 32-member bench ensemble, per setting: ms / eq/s, inner iterations, psi against Picard.
GPU box, repo root."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import diiid_like_cfg, iter_like_cfg, shape_scan_member_table  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

LANE = dict(solver_method="newton", gmres_preconditioner="multigrid", newton_jacobian="finite_difference", newton_damping=1.0, max_iterations=60,
            newton_warmup_steps=2)
CAPS = ((8, 1), (12, 1), (16, 1), (24, 1), (8, 2), (16, 2), (8, 4))


def run(cfg, B, reps, **kw):
    with EnsembleSolver(cfg, batch=B, **kw) as es:
        best = 1e9
        for _ in range(reps + 1):
            es.fill_psi(0.0)
            t = time.perf_counter()
            res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
            es.sync()
            best = min(best, time.perf_counter() - t)
        return best, res, es.get_psi()


d3d = diiid_like_cfg(513)
t, res, ref = run(d3d, 1, 2)
print(f"DIII-D-like, ONE equilibrium, Picard: {1e3 * t:.2f} ms, inner {res[0]['picard_iterations']}")
for rs, mi in CAPS:
    c = {**d3d, "solver": {**d3d["solver"], **LANE, "gmres_restart": rs, "gmres_maxiter": mi}}
    t, res, psi = run(c, 1, 2)
    print(f"  restart {rs:2d} x maxiter {mi}: {1e3 * t:6.2f} ms, converged {res[0]['converged']}, inner {res[0]['picard_iterations']}, "
          f"psi vs Picard {np.abs(psi - ref).max() / np.abs(ref).max():.1e}", flush=True)
cfg = iter_like_cfg(513, tol=1e-8, max_iter=1000)
tb = shape_scan_member_table(cfg, 32)
kw = dict(coil_currents=tb["currents"], current_targets=tb["targets"], coil_r=tb["coil_r"], coil_z=tb["coil_z"])
t, res, ref = run(cfg, 32, 1, **kw)
print(f"bench ensemble (32 members), Picard: {32 / t:.1f} eq/s")
for rs, mi in CAPS:
    c = {**cfg, "solver": {**cfg["solver"], **LANE, "gmres_restart": rs, "gmres_maxiter": mi}}
    t, res, psi = run(c, 32, 1, **kw)
    its = [sum(r["picard_iterations"]) for r in res]
    print(f"  restart {rs:2d} x maxiter {mi}: {32 / t:7.1f} eq/s, converged {sum(r['converged'] for r in res)}/32, inner {min(its)}..{max(its)}, "
          f"psi vs Picard {np.abs(psi - ref).max() / np.abs(ref).max():.1e}", flush=True)

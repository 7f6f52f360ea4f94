"""Where one coupled transport step (BASELINE config 4) spends its time: psi downloads, host norms, profile map, re-solve, current diffusion."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd import transport_coupling as tc  # noqa: E402
from scpn_fusion_core_amd.cases import iter_like_cfg  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

n, B = 513, 32
cfg = iter_like_cfg(n, tol=1e-6, max_iter=400)
rho = np.linspace(0, 1, 50)
scale = 1.0 + 0.1 * np.linspace(-1, 1, B)[:, None]
te = 8.0 * (1 - rho**2)[None] * scale + 0.1
ne = 9.0 * (1 - 0.8 * rho**2)[None] * np.ones((B, 1))
with EnsembleSolver(cfg, batch=B, current_targets=15.0 * scale[:, 0]) as es:
    es.solve_equilibrium()
    for k in range(3):
        te = te + 0.02 * (8.0 * (1 - rho**2)[None] * scale * 1.05 - te)
        p1 = ne * (2.0 * te)
        jbs = 1e-3 * np.gradient(p1, rho, axis=1) * rho[None] * -1.0
        t = [time.perf_counter()]
        psi_old = es.get_psi(); t.append(time.perf_counter())
        norm_old = np.linalg.norm(psi_old.reshape(B, -1), axis=1); t.append(time.perf_counter())
        tc.map_profiles_to_2d(es, rho, p1, jbs); es.sync(); t.append(time.perf_counter())
        es.external_profile_mode = True
        res = es.solve_equilibrium(); es.sync(); t.append(time.perf_counter())
        es.external_profile_mode = False
        psi = es.get_psi(); t.append(time.perf_counter())
        resid = np.linalg.norm((psi - psi_old).reshape(B, -1), axis=1) / norm_old; t.append(time.perf_counter())
        names = ["get_psi", "norm_old", "map_profiles", "re-solve", "get_psi", "diff norm"]
        print(k, {nm: round(1e3 * (b - a), 2) for nm, a, b in zip(names, t[:-1], t[1:])}, "iterations", res[0]["iterations"], res[-1]["iterations"], flush=True)

"""Where a bench step goes: cost per Picard iteration (slope of two fixed-iteration runs), per-call set-up (their intercept: seed,
50 Jacobi steps, host synchronisations) and what is left of a free-boundary solve of the bench ensemble (two outer iterations).
usage: step_breakdown.py [members]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import iter_like_cfg  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(2026)


def timed(cfg, fn, reps=4):
    ip = cfg["physics"]["plasma_current_target"]
    base = np.array([c["current"] for c in cfg["coils"]])
    with EnsembleSolver(cfg, batch=B, current_targets=ip * (1.0 + 0.1 * (2.0 * rng.random(B) - 1.0)),
                        coil_currents=base * (1.0 + 0.1 * rng.uniform(-1, 1, size=(B, base.size)))) as es:
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn(es)
            ts.append(time.perf_counter() - t0)
    return min(ts[1:]), out


t = {}
for iters in (100, 300):
    t[iters], _ = timed(iter_like_cfg(513, tol=1e-300, max_iter=iters), lambda es: es.solve_equilibrium(check_every=8))
slope = (t[300] - t[100]) / 200
setup = t[100] - 100 * slope
tfb, res = timed(iter_like_cfg(513, tol=1e-8, max_iter=1000), lambda es: es.solve_free_boundary(max_outer_iter=20, tol=1e-4, check_every=8))
its = [sum(r["picard_iterations"]) if "picard_iterations" in r else None for r in res]
print(f"per Picard iteration {1e6 * slope:.1f} us; per solve_equilibrium call set-up {1e3 * setup:.2f} ms")
print(f"free-boundary solve {1e3 * tfb:.2f} ms; keys {sorted(res[0])[:12]}")
print("iterations", its[:4])

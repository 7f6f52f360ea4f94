"""Sweep of the inexact-Newton knobs of the free-boundary ensemble (finite-difference Jacobian + multigrid-preconditioned GMRES):
Krylov cap, warm-up length; prints eq/s, inner iterations, psi difference to the Picard solution, and (last config) the kernel
families.  usage: newton_fb_sweep.py [n] [batch]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import iter_like_cfg, scan_member_table  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = iter_like_cfg(n, tol=1e-8, max_iter=1000)
currents, targets = scan_member_table(cfg, B)


def run(extra, reps=2, prof=False):
    c = {**cfg, "solver": {**cfg["solver"], **extra}}
    with EnsembleSolver(c, batch=B, coil_currents=currents, current_targets=targets) as es:
        best = 1e9
        ts = []
        for _ in range(reps + 4):
            es.fill_psi(0.0)
            t = time.perf_counter()
            res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
            ts.append(time.perf_counter() - t)
            best = min(best, ts[-1])
        print("   times ms:", " ".join(f"{1e3 * v:.1f}" for v in ts))
        fam = None
        if prof:
            es.fill_psi(0.0)
            es.profile(True)
            es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
            fam = es.profile_report()
            es.profile(False)
        return es.get_psi(), res, best, fam


ref, res, t, _ = run({})
print(f"picard: {B / t:.1f} eq/s")
base = dict(solver_method="newton", gmres_preconditioner="multigrid", newton_jacobian="finite_difference", newton_damping=1.0, max_iterations=60)
configs = [dict(newton_warmup_steps=2)]
for k, extra in enumerate(configs):
    psi, res, t, fam = run({**base, **extra}, prof=False)
    its = [sum(r["picard_iterations"]) for r in res]
    print(f"{extra}: {B / t:8.1f} eq/s ({t * 1e3:.1f} ms) converged {sum(r['converged'] for r in res)}/{B} outer {res[0]['outer_iterations']} "
          f"inner {min(its)}..{max(its)} per outer {res[0]['picard_iterations']}  dpsi {float(np.abs(psi - ref).max() / np.abs(ref).max()):.2e}", flush=True)
    if fam:
        tot = sum(v["ms"] for v in fam.values())
        for name, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:40]:
            print(f"    {name:26s} {v['ms']:8.2f} ms {v['launches']:6d} {1e3 * v['ms'] / max(v['launches'], 1):8.1f} us {100 * v['ms'] / tot:5.1f}%")
        print(f"    total {tot:.1f} ms (both groups, serialised by the profiler)")

"""Steady-state cost of one Picard iteration: 32 members, 513x513, exactly N iterations (tolerance far below reach), both
stream groups.  usage: fixed_iters_bench.py [iters] [repeats] [n] [members]   (SCPN_HIP_LIB selects the library)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import iter_like_cfg  # noqa: E402

from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(sys.argv[3]) if len(sys.argv) > 3 else 513
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
cfg = iter_like_cfg(n, tol=1e-300, max_iter=iters)
ip = cfg["physics"]["plasma_current_target"]
rng = np.random.default_rng(2026)
with EnsembleSolver(cfg, batch=B, current_targets=ip * (1.0 + 0.1 * (2.0 * rng.random(B) - 1.0))) as es:
    ts = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        res = es.solve_equilibrium(check_every=int(os.environ.get("CHECK_EVERY", "8")))
        ts.append(time.perf_counter() - t0)
    assert all(x["iterations"] == iters for x in res), [x["iterations"] for x in res]
best = min(ts[1:])
print(f"{1e6 * best / iters:8.1f} us/iteration  ({1e3 * best:.1f} ms for {iters} iterations; runs {[round(1e3 * t, 1) for t in ts]})")

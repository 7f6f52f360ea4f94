"""Helper of test_gpu_variants.py: solve one seeded 2-member Picard problem and print a digest of the result.

Run as a subprocess so that the library's read-once environment switches (SCPN_HIP_*) take effect."""
import hashlib
import json
import sys

import numpy as np

sys.path.insert(0, sys.argv[1])
from _problems import iter_like_cfg  # noqa: E402

from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

n = int(sys.argv[2])
cfg = iter_like_cfg(n, tol=1e-8, max_iter=12)
ip = cfg["physics"]["plasma_current_target"]
with EnsembleSolver(cfg, batch=2, current_targets=np.array([ip, 0.9 * ip])) as es:
    res = es.solve_equilibrium(check_every=4)
    psi = es.get_psi()
    jphi = es.get_jphi()
# one V-cycle (omega 1.0) over fields that hold signed zeros, subnormals, huge values, an infinity and a NaN: every switch must
# propagate them exactly as the default does (257^2: tile kernels + coarse kernel; 65^2: the coarse kernel alone)
from scpn_fusion_core_amd import multigrid_solve as mgs  # noqa: E402

special = hashlib.sha256()
for m in (257, 65):
    rng = np.random.default_rng(m)
    psi0 = rng.standard_normal((m, m))
    src0 = rng.standard_normal((m, m))
    psi0[rng.random((m, m)) < 0.05] = -0.0
    src0[rng.random((m, m)) < 0.05] = 0.0
    psi0[3, 5] = 5e-324; psi0[m // 2, m // 3] = 1e300; src0[7, 9] = -1e-310
    psi0[m - 5, m - 7] = np.inf; src0[m // 4, m // 4] = np.nan
    r_grid = np.broadcast_to(np.linspace(1.0, 2.0, m), (m, m))
    with np.errstate(all="ignore"):
        out = mgs.multigrid_vcycle(psi0, src0, r_grid, 1.0 / (m - 1), 2.0 / (m - 1), omega=1.0)
    out = np.where(np.isnan(out), np.nan, out)  # (which NaN a NaN result is -- payload, sign -- is not part of the contract)
    special.update(np.ascontiguousarray(out).tobytes())
print(json.dumps({"special": special.hexdigest(), "psi": hashlib.sha256(np.ascontiguousarray(psi).tobytes()).hexdigest(),
                  "jphi": hashlib.sha256(np.ascontiguousarray(jphi).tobytes()).hexdigest(),
                  "iterations": [r["iterations"] for r in res], "residual": [r["residual"] for r in res]}))

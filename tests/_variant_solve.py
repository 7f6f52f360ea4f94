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
print(json.dumps({"psi": hashlib.sha256(np.ascontiguousarray(psi).tobytes()).hexdigest(),
                  "jphi": hashlib.sha256(np.ascontiguousarray(jphi).tobytes()).hexdigest(),
                  "iterations": [r["iterations"] for r in res], "residual": [r["residual"] for r in res]}))

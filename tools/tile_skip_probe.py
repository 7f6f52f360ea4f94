"""Phase costs of the tile smoothers inside the real V-cycle: time per launch of the level-0 pre / post smoother for libraries
built with -DTILE_SKIP=<bits> (1 no global loads, 2 no sweeps, 4 no defect/restriction, 8 no psi store).  Results of such builds
are garbage; only the timing is of interest.  usage: SCPN_HIP_LIB=build/libscpn_skipN.so tile_skip_probe.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import iter_like_cfg  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

cfg = iter_like_cfg(513, tol=1e-8, max_iter=10)
with EnsembleSolver(cfg, batch=32) as es:
    rng = np.random.default_rng(1)
    es.set_psi(rng.standard_normal((32, 513, 513)))
    es.set_source(rng.standard_normal((32, 513, 513)))
    for _ in range(3):
        es.vcycle()
    es.profile(True)
    for _ in range(30):
        es.vcycle()
    rep = es.profile_report()
    es.profile(False)
print(" ".join(f"{k}={1e3 * v['ms'] / v['launches']:.1f}" for k, v in sorted(rep.items()) if k.startswith(("smooth_tile_L0", "smooth_tile_L1"))))

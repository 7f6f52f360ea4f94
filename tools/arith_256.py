"""Exact vs contracted arithmetic on the whole 256-member config-5 scan (513^2 free-boundary): psi difference and iteration counts of every member.  GPU box, repo root."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import iter_like_cfg, shape_scan_member_table  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = iter_like_cfg(513, tol=1e-8, max_iter=1000)
t = shape_scan_member_table(cfg, total)
out = {}
for mode in ("exact", "contracted"):
    with EnsembleSolver(cfg, batch=total, coil_currents=t["currents"], current_targets=t["targets"], coil_r=t["coil_r"], coil_z=t["coil_z"], arithmetic=mode) as es:
        es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
        es.fill_psi(0.0)
        t0 = time.perf_counter()
        res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
        es.sync()
        dt = time.perf_counter() - t0
        psi = np.stack([es.get_psi(m) for m in range(total)])
    out[mode] = (psi, res)
    print(f"{mode}: {total / dt:.1f} eq/s, converged {sum(r['converged'] for r in res)}/{total}", flush=True)
pe, re_ = out["exact"]
pc, rc = out["contracted"]
rel = np.abs(pe - pc).reshape(total, -1).max(axis=1) / np.abs(pe).reshape(total, -1).max(axis=1)
same = sum(r1["picard_iterations"] == r2["picard_iterations"] for r1, r2 in zip(re_, rc))
print(f"max over members of max|psi_exact - psi_contracted| / max|psi|: {rel.max():.3e} (median {np.median(rel):.3e}); members with identical Picard iteration counts: {same}/{total}")

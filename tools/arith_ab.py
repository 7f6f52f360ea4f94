"""Exact vs contracted arithmetic (scpn_gs_set_arith) of the resident solver: one V-cycle on the smoother benchmark problem for
the three tile geometries and two relaxation factors, and converged Picard solves of the ITER-like case.
usage: arith_ab.py [n_picard_grid ...]   (SCPN_HIP_LIB selects the library)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import bench_problem, iter_like_cfg  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402


def vcycle_pair(n, B, omega, cycles=1):
    psi0, source, _, r, z = bench_problem(n, n)
    cfg = iter_like_cfg(n)
    cfg["dimensions"] = {"R_min": float(r[0]), "R_max": float(r[-1]), "Z_min": float(z[0]), "Z_max": float(z[-1])}
    out = {}
    for mode in ("exact", "contracted"):
        with EnsembleSolver(cfg, batch=B, arithmetic=mode) as es:
            es.set_psi(np.broadcast_to(psi0, (B, n, n)).copy())
            es.set_source(np.broadcast_to(source, (B, n, n)).copy())
            for _ in range(cycles):
                es.vcycle(omega=omega, pre_smooth=3, post_smooth=3, min_grid=5)
            out[mode] = es.get_psi()
    d = np.abs(out["exact"] - out["contracted"]).max() / np.abs(out["exact"]).max()
    same_members = all(np.array_equal(out["contracted"][0], out["contracted"][b]) for b in range(B))
    return d, same_members


for n in (513, 300):
    for B, geo in ((1, "24x12"), (4, "32x8"), (16, "48x8")):
        for omega in (1.0, 1.6):
            d, same = vcycle_pair(n, B, omega, cycles=3)
            print(f"vcycle x3 n={n} B={B:2d} ({geo}) omega={omega}: max|exact-contracted|/max|psi| = {d:.3e}  members identical: {same}", flush=True)

for n in [int(a) for a in sys.argv[1:]] or [129, 257]:
    for omega in (1.0, 1.6):
        res = {}
        for mode in ("exact", "contracted"):
            cfg = iter_like_cfg(n, omega=omega)
            with EnsembleSolver(cfg, batch=1, arithmetic=mode) as es:
                t0 = time.perf_counter()
                r = es.solve_equilibrium()[0]
                dt = time.perf_counter() - t0
                res[mode] = (es.get_psi(0), r["iterations"], r["converged"], dt)
        pe, pc = res["exact"][0], res["contracted"][0]
        print(f"picard n={n} omega={omega}: iterations exact {res['exact'][1]} / contracted {res['contracted'][1]}, converged {res['exact'][2]} / {res['contracted'][2]}, "
              f"max|dpsi|/max|psi| = {np.abs(pe - pc).max() / np.abs(pe).max():.3e}, ms {1e3 * res['exact'][3]:.1f} / {1e3 * res['contracted'][3]:.1f}", flush=True)

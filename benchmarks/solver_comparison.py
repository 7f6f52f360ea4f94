#!/usr/bin/env python3
"""Counterpart of the reference's benchmarks/solver_comparison.py (SOR vs Newton-Kantorovich FusionKernel solves on
33x33 and 65x65, reference lines :40-117) on the HIP tier, extended to 129x129 and to 32-member ensembles.

Each row: the same config solved by (a) the HIP path (one member, wall time of FusionKernel-equivalent solve),
(b) the HIP path as a 32-member ensemble (wall time / 32), (c) the CPU oracle (C port of the reference algorithm,
one core) -- the reference's NumPy/SciPy lane does not travel to the GPU box.  Iteration counts of (a) and (c) must agree.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402


def config(n, method, max_iter):
    return {
        "reactor_name": f"bench_{method}_{n}x{n}", "grid_resolution": [n, n],
        "dimensions": {"R_min": 1.0, "R_max": 3.0, "Z_min": -1.5, "Z_max": 1.5},
        "physics": {"plasma_current_target": 1.0e6, "vacuum_permeability": 1.0},
        "coils": [{"name": "PF1", "r": 3.5, "z": 1.0, "current": 5e5}, {"name": "PF2", "r": 3.5, "z": -1.0, "current": 5e5},
                  {"name": "PF3", "r": 1.0, "z": 1.0, "current": 5e5}, {"name": "PF4", "r": 1.0, "z": -1.0, "current": 5e5},
                  {"name": "PF5", "r": 2.2, "z": 2.0, "current": 5e5}],
        "solver": {"solver_method": method, "max_iterations": max_iter, "convergence_threshold": 1e-6,
                   "relaxation_factor": 0.5, "sor_omega": 1.6},
    }


def main():
    from oracle import oracle as orc  # CPU baseline / checker only

    grids = [int(a) for a in sys.argv[1:]] or [33, 65, 129]
    rows = []
    for n in grids:
        for method, max_iter in (("sor", 200), ("newton", 20)):
            cfg = config(n, method, max_iter)
            with EnsembleSolver(cfg, batch=1) as es:
                es.solve_equilibrium()  # warm-up (module load, buffers)
                es.fill_psi(0.0)
                t = time.perf_counter()
                r1 = es.solve_equilibrium()[0]
                es.sync()
                t1 = time.perf_counter() - t
                psi1 = es.get_psi(0)
            B = 32
            ip = cfg["physics"]["plasma_current_target"] * (1.0 + 0.05 * np.linspace(-1, 1, B))
            with EnsembleSolver(cfg, batch=B, current_targets=ip) as es:
                es.solve_equilibrium()
                es.fill_psi(0.0)
                es.profile(True)
                t = time.perf_counter()
                rb = es.solve_equilibrium()
                es.sync()
                tb = time.perf_counter() - t
                fam = es.profile_report()
                es.profile(False)
            t = time.perf_counter()
            ref = orc.newton_solve(cfg) if method == "newton" else orc.picard_solve(cfg)
            tc = time.perf_counter() - t
            err = float(np.abs(psi1 - ref["psi"]).max() / np.abs(ref["psi"]).max())
            top = sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:4]
            rows.append({"solver": "Newton-K" if method == "newton" else "SOR", "grid": f"{n}x{n}", "iterations": r1["iterations"],
                         "iterations_cpu": ref["iterations"], "hip_ms": t1 * 1e3, "hip_ens32_ms_per_member": tb * 1e3 / B,
                         "cpu_port_ms": tc * 1e3, "psi_rel_linf_vs_cpu": err,
                         "gmres_inner": r1.get("gmres_inner_iterations"), "ens_iterations": [min(r["iterations"] for r in rb), max(r["iterations"] for r in rb)],
                         "top_kernels_ens": {k: round(v["ms"], 2) for k, v in top}})
            print(rows[-1], flush=True)
    print("\n| Solver | Grid | Iterations | HIP 1 member (ms) | HIP 32-member ensemble (ms / member) | CPU port, 1 core (ms) | max rel. psi diff |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['solver']} | {r['grid']} | {r['iterations']} | {r['hip_ms']:.1f} | {r['hip_ens32_ms_per_member']:.2f} | "
              f"{r['cpu_port_ms']:.1f} | {r['psi_rel_linf_vs_cpu']:.1e} |")
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "solver_comparison.json").write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()

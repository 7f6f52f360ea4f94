"""Synthetic benchmark cases of the hot path (BASELINE configs), restated from the reference's harness inputs.

Host-side, NumPy only.  Used by ``bench.py``, ``benchmarks/``, ``tools/`` and (re-exported through
``tests/_problems.py``) the parity tests.
"""
from __future__ import annotations

import numpy as np


def grid(nz, nr, r0=4.0, r1=8.0, z0=-4.0, z1=4.0):
    r = np.linspace(r0, r1, nr)
    z = np.linspace(z0, z1, nz)
    rr, zz = np.meshgrid(r, z)
    return r, z, rr, zz


def bench_problem(nz, nr, seed=2026):
    """The smoother benchmark problem of the reference (benchmarks/bench_gpu_gs_solver.py:54-63)."""
    rng = np.random.default_rng(seed)
    r, z, rr, zz = grid(nz, nr)
    source = -np.exp(-((rr - 6.0) ** 2 + zz**2) / 0.5)
    psi0 = rng.normal(0.0, 1e-3, size=(nz, nr))
    psi0[0, :] = psi0[-1, :] = psi0[:, 0] = psi0[:, -1] = 0.0
    return psi0, source, rr, r, z


def iter_like_cfg(n, method="multigrid", omega=1.0, tol=1e-8, max_iter=1000, alpha=0.12):
    """ITER-like case of validation/benchmark_vs_freegs.py:157,412-456 (values restated)."""
    R0, a, kappa, Ip = 6.2, 2.0, 1.7, 15.0
    coil_R, coil_Z = R0 + 1.5 * a, 1.2 * kappa * a
    return {
        "reactor_name": "ITER-like",
        "grid_resolution": [n, n],
        "dimensions": {"R_min": round(R0 - 2 * a, 4), "R_max": round(R0 + 2 * a, 4),
                       "Z_min": round(-2 * kappa * a, 4), "Z_max": round(2 * kappa * a, 4)},
        "physics": {"plasma_current_target": Ip, "vacuum_permeability": 1.0},
        "coils": [
            {"name": "PF1", "r": coil_R, "z": coil_Z, "current": 2.0},
            {"name": "PF2", "r": coil_R, "z": -coil_Z, "current": 2.0},
            {"name": "PF3", "r": R0 - 0.5 * a, "z": coil_Z * 1.1, "current": -1.0},
            {"name": "PF4", "r": R0 - 0.5 * a, "z": -coil_Z * 1.1, "current": -1.0},
            {"name": "CS", "r": max(R0 - 2 * a + 0.05, 0.3), "z": 0.0, "current": 0.15},
        ],
        "solver": {"max_iterations": max_iter, "convergence_threshold": tol, "relaxation_factor": alpha,
                   "solver_method": method, "sor_omega": omega},
    }


def diiid_like_cfg(n, tol=1e-8, max_iter=1000, alpha=0.12):
    """DIII-D-like case of BASELINE config 3 (validation/diiid_config.json, values restated; solver block as the parity fixtures
    use it: multigrid, sor_omega 1.0).  Coil positions are the reference's approximate ones (Luxon, Nucl. Fusion 42 (2002) 614)."""
    coils = [("OH", 0.42, 0.0, -4.0), ("F1A", 0.86, 1.01, 2.0), ("F2A", 0.86, 0.60, 1.5), ("F3A", 0.86, 0.17, 0.8),
             ("F4A", 0.86, -0.17, 0.8), ("F5A", 0.86, -0.60, 1.5), ("F6A", 0.86, -1.01, 2.0), ("F7A", 2.40, 1.10, -1.2),
             ("F8A", 2.40, -1.10, -1.2), ("F9A", 2.00, 1.55, -0.5)]
    return {
        "reactor_name": "DIII-D (General Atomics)",
        "grid_resolution": [n, n],
        "dimensions": {"R_min": 0.8, "R_max": 2.6, "Z_min": -1.8, "Z_max": 1.8},
        "physics": {"plasma_current_target": 1.6, "vacuum_permeability": 1.0},
        "target": {"R_axis": 1.67, "Z_axis": 0.0},
        "coils": [{"name": nm, "r": r, "z": z, "current": c} for nm, r, z, c in coils],
        "solver": {"max_iterations": max_iter, "convergence_threshold": tol, "relaxation_factor": alpha,
                   "solver_method": "multigrid", "sor_omega": 1.0},
    }


def scan_member_table(cfg: dict, total: int, seed: int = 2026) -> tuple[np.ndarray, np.ndarray]:
    """Coil/shape scan of BASELINE config 5: coil currents I_k*(1+0.1u) and plasma current Ip*(1+0.1u),
    u ~ U(-1, 1) from ``np.random.default_rng(seed)``; returns (currents [total, n_coils], I_p targets [total])."""
    base = np.array([c["current"] for c in cfg["coils"]], dtype=np.float64)
    rng = np.random.default_rng(seed)
    currents = base[None, :] * (1.0 + 0.1 * rng.uniform(-1.0, 1.0, size=(total, base.size)))
    targets = cfg["physics"]["plasma_current_target"] * (1.0 + 0.1 * rng.uniform(-1.0, 1.0, size=total))
    return currents, targets


def member_cfg(cfg: dict, currents_row, target) -> dict:
    """The config of one scan member on its own (what a single-equilibrium solver would be handed for it)."""
    return {**cfg, "coils": [dict(co, current=float(currents_row[i])) for i, co in enumerate(cfg["coils"])],
            "physics": dict(cfg["physics"], plasma_current_target=float(target))}

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


ITER_LIKE = {"R0": 6.2, "a": 2.0, "kappa": 1.7, "Ip": 15.0}  # validation/benchmark_vs_freegs.py:157


def iter_like_coil_positions(kappa: float = ITER_LIKE["kappa"]) -> tuple[list[float], list[float]]:
    """(r, z) of the five PF coils of ``build_config`` (validation/benchmark_vs_freegs.py:424-432) for an elongation ``kappa``:
    the coils sit at Z = +-1.2 kappa a (PF1/PF2) and +-1.1 * 1.2 kappa a (PF3/PF4) -- the shape knob of the config-5 scan."""
    R0, a = ITER_LIKE["R0"], ITER_LIKE["a"]
    coil_R, coil_Z = R0 + 1.5 * a, 1.2 * kappa * a
    return ([coil_R, coil_R, R0 - 0.5 * a, R0 - 0.5 * a, max(R0 - 2 * a + 0.05, 0.3)],
            [coil_Z, -coil_Z, coil_Z * 1.1, -coil_Z * 1.1, 0.0])


def iter_like_cfg(n, method="multigrid", omega=1.0, tol=1e-8, max_iter=1000, alpha=0.12):
    """ITER-like case of validation/benchmark_vs_freegs.py:157,412-456 (values restated)."""
    R0, a, kappa, Ip = (ITER_LIKE[k] for k in ("R0", "a", "kappa", "Ip"))
    cr, cz = iter_like_coil_positions(kappa)
    return {
        "reactor_name": "ITER-like",
        "grid_resolution": [n, n],
        "dimensions": {"R_min": round(R0 - 2 * a, 4), "R_max": round(R0 + 2 * a, 4),
                       "Z_min": round(-2 * kappa * a, 4), "Z_max": round(2 * kappa * a, 4)},
        "physics": {"plasma_current_target": Ip, "vacuum_permeability": 1.0},
        "coils": [{"name": nm, "r": cr[k], "z": cz[k], "current": cur}
                  for k, (nm, cur) in enumerate((("PF1", 2.0), ("PF2", 2.0), ("PF3", -1.0), ("PF4", -1.0), ("CS", 0.15)))],
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
    """The coil-current half of the config-5 scan: coil currents I_k*(1+0.1u) from ``np.random.default_rng(seed)`` and plasma current
    Ip*(1+0.1u) from ``np.random.default_rng([seed, 7])``, u ~ U(-1, 1); returns (currents [total, n_coils], I_p targets [total]).
    Row m of both is the same for every ``total``."""
    base = np.array([c["current"] for c in cfg["coils"]], dtype=np.float64)
    rng = np.random.default_rng(seed)
    currents = base[None, :] * (1.0 + 0.1 * rng.uniform(-1.0, 1.0, size=(total, base.size)))
    # (own stream, as kappa below: member m's I_p must not depend on `total` -- member 7 of a 32-member weak-scaling block IS member 7
    #  of the 256-member strong-scaling ensemble.  Rounds 1-5 drew the targets behind the (total, n) current block of ONE generator.)
    targets = cfg["physics"]["plasma_current_target"] * (1.0 + 0.1 * np.random.default_rng([seed, 7]).uniform(-1.0, 1.0, size=total))
    return currents, targets


def shape_scan_member_table(cfg: dict, total: int, seed: int = 2026, kappa_range: tuple[float, float] = (1.5, 1.9)) -> dict:
    """Coil/shape scan of BASELINE config 5 (SURVEY 8d cfg 5): per member, coil currents I_k*(1+0.1u), plasma current Ip*(1+0.1u),
    u ~ U(-1, 1), and an elongation kappa ~ U(kappa_range) realised through the coil heights Z = 1.2 kappa a of the ITER-like
    coil set (validation/benchmark_vs_freegs.py:424-432); grid and domain are the config's for every member.  Currents and targets
    are ``scan_member_table``'s (the rounds 1-4 table), kappa comes from ``np.random.default_rng([seed, 5])``.
    Returns ``{"currents": [total, n], "targets": [total], "kappa": [total], "coil_r": [total, n], "coil_z": [total, n]}``."""
    if len(cfg["coils"]) != 5 or cfg.get("reactor_name") != "ITER-like":
        raise ValueError("the shape scan is defined on the ITER-like five-coil set")
    currents, targets = scan_member_table(cfg, total, seed)
    kappa = np.random.default_rng([seed, 5]).uniform(kappa_range[0], kappa_range[1], size=total)  # (own stream: member m's kappa does not depend on `total`)
    pos = [iter_like_coil_positions(float(k)) for k in kappa]
    return {"currents": currents, "targets": targets, "kappa": kappa,
            "coil_r": np.array([p[0] for p in pos], dtype=np.float64).reshape(total, 5),
            "coil_z": np.array([p[1] for p in pos], dtype=np.float64).reshape(total, 5)}


def member_cfg(cfg: dict, currents_row, target, coil_r_row=None, coil_z_row=None) -> dict:
    """The config of one scan member on its own (what a single-equilibrium solver would be handed for it)."""
    coils = []
    for i, co in enumerate(cfg["coils"]):
        c = dict(co, current=float(currents_row[i]))
        if coil_r_row is not None:
            c["r"] = float(coil_r_row[i])
        if coil_z_row is not None:
            c["z"] = float(coil_z_row[i])
        coils.append(c)
    return {**cfg, "coils": coils, "physics": dict(cfg["physics"], plasma_current_target=float(target))}

"""Strict-parity gate metrics of a free-boundary equilibrium against a same-case reference flux map.

Counterpart of the acceptance block of the reference's FreeGS gate
(validation/benchmark_freegs_public_example_reconstruction.py:431-497, evaluated by
tests/test_benchmark_free_boundary_strict_parity.py:95-125): normalised-flux RMSE, magnetic-axis distance, boundary flux error
and normalised-flux error at the X-point constraints, each with the reference's threshold.  Host-side NumPy on two (nz, nr)
arrays -- the flux maps themselves come from the device solver (``EnsembleSolver.solve_free_boundary``) and from fixtures.
"""
from __future__ import annotations

import numpy as np

# thresholds of the reference gate (benchmark_freegs_public_example_reconstruction.py:491-497)
GATE_THRESHOLDS = {"psi_n_rmse": 5.0e-2, "axis_error_m": 2.5e-2, "boundary_max_abs_error_wb": 1.0e-10, "xpoint_psi_n_error": 5.0e-2}


def normalised_flux(psi: np.ndarray) -> np.ndarray:
    """(psi - min) / (max - min), span floored at 1e-30 (:318-321)."""
    lo = float(np.min(psi))
    return (psi - lo) / max(float(np.max(psi)) - lo, 1.0e-30)


def axis_location(r_axis: np.ndarray, z_axis: np.ndarray, psi: np.ndarray) -> tuple[float, float]:
    """(R, Z) of the grid maximum of psi[iz, ir], NaNs ignored (:324-330)."""
    iz, ir = np.unravel_index(int(np.nanargmax(psi)), psi.shape)
    return float(r_axis[ir]), float(z_axis[iz])


def ring_values(field: np.ndarray) -> np.ndarray:
    """The computational-boundary values of a field, every node once."""
    return np.concatenate([field[0, :], field[-1, :], field[1:-1, 0], field[1:-1, -1]])


def strict_parity_metrics(psi: np.ndarray, psi_ref: np.ndarray, r_axis, z_axis, x_points=()) -> dict:
    """Gate metrics of ``psi`` against ``psi_ref`` (both (nz, nr)) and the pass flag of each threshold."""
    psi, psi_ref = np.asarray(psi, dtype=np.float64), np.asarray(psi_ref, dtype=np.float64)
    if psi.shape != psi_ref.shape or psi.shape != (len(z_axis), len(r_axis)):
        raise ValueError("psi, psi_ref and the axes must describe the same (nz, nr) grid")
    r_axis, z_axis = np.asarray(r_axis, dtype=np.float64), np.asarray(z_axis, dtype=np.float64)
    pn, rn = normalised_flux(psi), normalised_flux(psi_ref)
    ax, ax_ref = axis_location(r_axis, z_axis, psi), axis_location(r_axis, z_axis, psi_ref)
    x_err = 0.0
    for r_x, z_x in x_points:  # nearest grid node, as the reference samples its X-point constraints (:370-378)
        ir, iz = int(np.argmin(np.abs(r_axis - float(r_x)))), int(np.argmin(np.abs(z_axis - float(z_x))))
        x_err = max(x_err, abs(float(pn[iz, ir]) - float(rn[iz, ir])))
    out = {
        "psi_n_rmse": float(np.sqrt(np.mean((pn - rn) ** 2))),
        "psi_rmse": float(np.sqrt(np.mean((psi - psi_ref) ** 2))),
        "axis_error_m": float(np.hypot(ax[0] - ax_ref[0], ax[1] - ax_ref[1])),
        "boundary_max_abs_error_wb": float(np.max(np.abs(ring_values(psi - psi_ref)))),
        "boundary_containment_fraction": float(np.mean(np.abs(ring_values(psi - psi_ref)) <= GATE_THRESHOLDS["boundary_max_abs_error_wb"])),
        "xpoint_psi_n_error_max": x_err,
        "finite": bool(np.all(np.isfinite(psi))),
    }
    out["checks"] = {"psi_n_rmse": out["psi_n_rmse"] <= GATE_THRESHOLDS["psi_n_rmse"],
                     "axis_error_m": out["axis_error_m"] <= GATE_THRESHOLDS["axis_error_m"],
                     "boundary_max_abs_error_wb": out["boundary_max_abs_error_wb"] <= GATE_THRESHOLDS["boundary_max_abs_error_wb"],
                     "xpoint_psi_n_error": x_err <= GATE_THRESHOLDS["xpoint_psi_n_error"]}
    out["pass"] = bool(out["finite"] and all(out["checks"].values()))
    return out

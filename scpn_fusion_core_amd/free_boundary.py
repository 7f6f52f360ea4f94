"""Coil Green's functions and the free-boundary outer loop on the HIP path.

Mirrors core/fusion_kernel_free_boundary.py of the reference (same names / argument meaning /
errors): ``green_function``, ``compute_external_flux``, ``build_mutual_inductance_matrix``,
``interp_psi``, ``solve_free_boundary``; plus ``CoilSet`` (core/fusion_kernel.py:60-101).
The elliptic integrals K(m), E(m) are evaluated on the device by the arithmetic-geometric mean
(the reference calls SciPy's Cephes ``ellipk/ellipe``; agreement <= 1e-12 relative, see tests).
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr

_MU0 = 4e-7 * np.pi


@dataclass
class CoilSet:
    """External coil set for the free-boundary solve (core/fusion_kernel.py:60-101)."""

    positions: list = field(default_factory=list)
    currents: np.ndarray = field(default_factory=lambda: np.array([]))
    turns: list = field(default_factory=list)
    current_limits: np.ndarray | None = None
    target_flux_points: np.ndarray | None = None
    target_flux_values: np.ndarray | None = None
    x_point_target: np.ndarray | None = None
    x_point_flux_target: float | None = None
    divertor_strike_points: np.ndarray | None = None
    divertor_flux_values: np.ndarray | None = None

    def turns_array(self) -> np.ndarray:
        n = len(self.positions)
        return np.array([self.turns[i] if i < len(self.turns) else 1 for i in range(n)], dtype=np.float64)


def coil_flux_grid(r_axis, z_axis, coil_r, coil_z, coil_i, *, turns=None, mu0=1.0, mode="external"):
    """Sum of coil Green's-function flux on the (Z, R) grid.

    mode="vacuum":   FusionKernel.calculate_vacuum_field (core/fusion_kernel.py:218-251; cfg mu0, turns ignored)
    mode="external": compute_external_flux (core/fusion_kernel_free_boundary.py:83-93; SI mu0, x turns)
    Coil arrays may carry a leading batch dimension -> output (batch, nz, nr).
    """
    L = _lib.require_gpu()
    r_axis, z_axis = as_f64(r_axis), as_f64(z_axis)
    cr, cz, ci = as_f64(coil_r), as_f64(coil_z), as_f64(coil_i)
    batched = ci.ndim == 2
    B = ci.shape[0] if batched else 1
    cr, cz, ci = (np.ascontiguousarray(np.broadcast_to(a.reshape(-1, a.shape[-1]) if a.ndim == 2 else a[None], (B, ci.shape[-1])))
                  for a in (cr, cz, ci))
    n = ci.shape[1]
    ct = None
    if turns is not None:
        t = as_f64(turns)
        ct = np.ascontiguousarray(np.broadcast_to(t if t.ndim == 2 else t[None], (B, n)))
    out = np.empty((B, z_axis.size, r_axis.size))
    check(L.scpn_coil_flux(r_axis, z_axis, z_axis.size, r_axis.size, B, n, ptr(cr), ptr(cz), ptr(ci), ptr(ct), float(mu0),
                           0 if mode == "vacuum" else 1, out))
    return out if batched else out[0]


def green_points(coil_r, coil_z, pts_r, pts_z):
    """G[coil, point] (SI mu0, per ampere-turn); NaN where the reference raises."""
    L = _lib.require_gpu()
    cr, cz, pr, pz = (np.ascontiguousarray(as_f64(a).reshape(-1)) for a in (coil_r, coil_z, pts_r, pts_z))
    out = np.empty((cr.size, pr.size))
    check(L.scpn_green_points(cr.size, cr, cz, pr.size, pr, pz, out))
    return out


def green_function(R_src: float, Z_src: float, R_obs: float, Z_obs: float) -> float:
    """core/fusion_kernel_free_boundary.py:31-55."""
    values = np.asarray([R_src, Z_src, R_obs, Z_obs], dtype=np.float64)
    if not np.all(np.isfinite(values)):
        raise ValueError("Green's-function coordinates must be finite.")
    if R_src <= 0.0 or R_obs <= 0.0:
        raise ValueError("Green's-function radii must be positive.")
    return float(green_points([R_src], [Z_src], [R_obs], [Z_obs])[0, 0])


def compute_external_flux(kernel: Any, coils: CoilSet):
    """core/fusion_kernel_free_boundary.py:83-93."""
    pos = np.array(coils.positions, dtype=np.float64).reshape(-1, 2)
    if np.any(pos[:, 0] <= 0.0) or not np.all(np.isfinite(pos)):
        raise ValueError("source coil coordinates must be finite with R_src > 0.")
    cur = as_f64(coils.currents)[: len(pos)]
    return coil_flux_grid(kernel.R, kernel.Z, pos[:, 0], pos[:, 1], cur, turns=coils.turns_array(), mode="external")


def build_mutual_inductance_matrix(kernel: Any, coils: CoilSet, obs_points) -> np.ndarray:
    """Coil-to-point flux-response matrix M[coil, point] (core/fusion_kernel_free_boundary.py:137-153)."""
    obs = as_f64(obs_points)
    pos = np.array(coils.positions, dtype=np.float64).reshape(-1, 2)
    if np.any(obs[:, 0] <= 0.0):
        raise ValueError("observation radii must be positive.")
    G = green_points(pos[:, 0], pos[:, 1], obs[:, 0], obs[:, 1])
    return coils.turns_array()[:, None] * G


def interp_psi(kernel: Any, R_pt: float, Z_pt: float) -> float:
    """Bilinear sample of kernel.Psi (core/fusion_kernel_free_boundary.py:562-580) -- host-side, 4 points."""
    ir = int(np.searchsorted(kernel.R, R_pt)) - 1
    iz = int(np.searchsorted(kernel.Z, Z_pt)) - 1
    ir = max(0, min(ir, kernel.NR - 2))
    iz = max(0, min(iz, kernel.NZ - 2))
    t_r = max(0.0, min(1.0, (R_pt - kernel.R[ir]) / kernel.dR))
    t_z = max(0.0, min(1.0, (Z_pt - kernel.Z[iz]) / kernel.dZ))
    P = kernel.Psi
    return float((1 - t_r) * (1 - t_z) * P[iz, ir] + t_r * (1 - t_z) * P[iz, ir + 1] + (1 - t_r) * t_z * P[iz + 1, ir]
                 + t_r * t_z * P[iz + 1, ir + 1])


def resolve_shape_target_flux(kernel: Any, coils: CoilSet) -> np.ndarray:
    """core/fusion_kernel_free_boundary.py:583-605: explicit target values, or the isoflux mean of bilinear psi samples."""
    if coils.target_flux_points is None:
        raise ValueError("CoilSet.target_flux_points must be set for shape optimisation.")
    obs = np.asarray(coils.target_flux_points, dtype=np.float64)
    n_pts = int(obs.shape[0])
    if obs.ndim != 2 or obs.shape[1] != 2 or n_pts == 0:
        raise ValueError("target_flux_points must have shape (n_points, 2) with n_points > 0.")
    if coils.target_flux_values is not None:
        target = np.asarray(coils.target_flux_values, dtype=np.float64).reshape(-1)
        if target.shape[0] != n_pts:
            raise ValueError("target_flux_values must have the same length as target_flux_points.")
        if not np.all(np.isfinite(target)):
            raise ValueError("target_flux_values must contain finite values only.")
        return target
    samples = np.array([float(interp_psi(kernel, R_t, Z_t)) for R_t, Z_t in obs], dtype=np.float64)
    return np.full(n_pts, float(np.mean(samples)), dtype=np.float64)


def optimize_coil_currents(kernel: Any, coils: CoilSet, target_flux, tikhonov_alpha: float = 1e-4) -> np.ndarray:
    """Bounded Tikhonov least squares for the coil currents (core/fusion_kernel_free_boundary.py:491-559).  The response
    matrix comes from the device Green's-function kernel; the (n_pts + n_coils) x n_coils ``lsq_linear`` stays on the host."""
    from scipy.optimize import lsq_linear

    if coils.target_flux_points is None:
        raise ValueError("CoilSet.target_flux_points must be set for optimisation.")
    if len(coils.positions) == 0:
        raise ValueError("CoilSet.positions must contain at least one coil.")
    if len(coils.currents) != len(coils.positions):
        raise ValueError("CoilSet.currents length must match number of coil positions.")
    if not np.isfinite(tikhonov_alpha) or tikhonov_alpha < 0.0:
        raise ValueError("tikhonov_alpha must be finite and non-negative.")
    obs = np.asarray(coils.target_flux_points, dtype=np.float64)
    if obs.ndim != 2 or obs.shape[1] != 2 or obs.shape[0] == 0:
        raise ValueError("target_flux_points must have shape (n_points, 2) with n_points > 0.")
    target = np.asarray(target_flux, dtype=np.float64).reshape(-1)
    if target.shape[0] != obs.shape[0]:
        raise ValueError("target_flux must have the same length as target_flux_points.")
    if not np.all(np.isfinite(target)):
        raise ValueError("target_flux must contain finite values only.")
    M = build_mutual_inductance_matrix(kernel, coils, obs)
    if not np.all(np.isfinite(M)):
        raise ValueError("Mutual inductance matrix contains non-finite entries.")
    n_coils = M.shape[0]
    A = np.vstack([M.T, np.sqrt(tikhonov_alpha) * np.eye(n_coils)])
    rhs = np.concatenate([target, np.zeros(n_coils)])
    if coils.current_limits is not None:
        limits = np.asarray(coils.current_limits, dtype=np.float64).reshape(-1)
        if limits.shape[0] != n_coils:
            raise ValueError("current_limits must have one entry per coil.")
        lb, ub = -np.abs(limits), np.abs(limits)
    else:
        lb, ub = -np.inf * np.ones(n_coils), np.inf * np.ones(n_coils)
    result = lsq_linear(A, rhs, bounds=(lb, ub), method="trf")
    if not bool(getattr(result, "success", False)) or not np.all(np.isfinite(result.x)):
        return np.clip(np.asarray(coils.currents, dtype=np.float64).copy(), lb, ub)
    return np.asarray(result.x, dtype=np.float64)


def solve_free_boundary(kernel: Any, coils: CoilSet, max_outer_iter: int = 20, tol: float = 1e-4,
                        optimize_shape: bool = False, tikhonov_alpha: float = 1e-4, **_unused) -> dict[str, Any]:
    """core/fusion_kernel_free_boundary.py:623-739.  ``optimize_shape=False``: the whole outer loop runs on the device.
    ``optimize_shape=True``: each outer iteration is a device Picard solve followed by the host-side bounded
    least-squares update of the coil currents (the reference's structure)."""
    if max_outer_iter < 1:
        raise ValueError("max_outer_iter must be >= 1.")
    if not np.isfinite(tol) or tol < 0.0:
        raise ValueError("tol must be finite and >= 0.")
    if not (optimize_shape and coils.target_flux_points is not None):
        return kernel._solve_free_boundary_device(coils, max_outer_iter, tol)
    psi_ext = compute_external_flux(kernel, coils)
    diff, outer, shape_optimization = float("inf"), 0, None
    for outer in range(max_outer_iter):
        kernel.Psi[0, :], kernel.Psi[-1, :] = psi_ext[0, :], psi_ext[-1, :]
        kernel.Psi[:, 0], kernel.Psi[:, -1] = psi_ext[:, 0], psi_ext[:, -1]
        psi_old = kernel.Psi.copy()
        kernel.solve_equilibrium(preserve_initial_state=True, boundary_flux=psi_ext)
        target_psi = resolve_shape_target_flux(kernel, coils)
        response = build_mutual_inductance_matrix(kernel, coils, coils.target_flux_points)
        new_currents = np.asarray(optimize_coil_currents(kernel, coils, target_psi, tikhonov_alpha=tikhonov_alpha),
                                  dtype=np.float64).reshape(-1)
        if new_currents.shape != (len(coils.positions),):
            raise ValueError("optimised coil current vector length must match coil count.")
        if not np.all(np.isfinite(new_currents)):
            raise ValueError("optimised coil currents must contain finite values only.")
        achieved = response.T @ new_currents
        residual = achieved - target_psi
        flux_rmse = float(np.sqrt(np.mean(residual**2)))
        shape_optimization = {
            "solver_mode": "free_boundary_solver_shape_current_optimization",
            "target_point_count": int(target_psi.shape[0]), "coil_count": int(len(coils.positions)),
            "flux_rmse": flux_rmse, "flux_relative_rmse": float(flux_rmse / max(float(np.sqrt(np.mean(target_psi**2))), 1.0)),
            "max_abs_flux_residual": float(np.max(np.abs(residual))), "target_flux": target_psi.copy(), "achieved_flux": achieved,
        }
        coils.currents = new_currents
        psi_ext = compute_external_flux(kernel, coils)
        diff = float(np.max(np.abs(kernel.Psi - psi_old)))
        if diff < tol:
            break
    return {"outer_iterations": outer + 1, "final_diff": diff, "coil_currents": coils.currents.copy(),
            "shape_optimization": shape_optimization}


__all__ = ["CoilSet", "coil_flux_grid", "green_points", "green_function", "compute_external_flux",
           "build_mutual_inductance_matrix", "interp_psi", "resolve_shape_target_flux", "optimize_coil_currents",
           "solve_free_boundary"]

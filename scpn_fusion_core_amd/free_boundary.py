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


def _as_finite_vector(value, *, name: str, length: int | None = None) -> np.ndarray:
    arr = np.asarray(value, dtype=np.float64).reshape(-1)
    if length is not None and arr.shape != (length,):
        raise ValueError(f"{name} must have length {length}.")
    if not np.all(np.isfinite(arr)):
        raise ValueError(f"{name} must contain finite values only.")
    return arr


def _as_finite_points(value, *, name: str) -> np.ndarray:
    arr = np.asarray(value, dtype=np.float64)
    if arr.ndim != 2 or arr.shape[1] != 2 or arr.shape[0] < 1:
        raise ValueError(f"{name} must have shape (n_points, 2) with n_points > 0.")
    if not np.all(np.isfinite(arr)):
        raise ValueError(f"{name} must contain finite values only.")
    return arr


def _points_inside_polygon(points, polygon) -> np.ndarray:
    """Even-odd containment mask (core/fusion_kernel_free_boundary.py:114-134)."""
    pts = _as_finite_points(points, name="points")
    poly = _as_finite_points(polygon, name="polygon")
    if poly.shape[0] < 3:
        raise ValueError("polygon must contain at least three points.")
    x, y = pts[:, 0], pts[:, 1]
    x0, y0 = poly[:, 0], poly[:, 1]
    x1, y1 = np.roll(x0, -1), np.roll(y0, -1)
    inside = np.zeros(pts.shape[0], dtype=bool)
    for xa, ya, xb, yb in zip(x0, y0, x1, y1):
        intersects = ((ya > y) != (yb > y)) & (x < (xb - xa) * (y - ya) / max(abs(yb - ya), 1.0e-300) + xa)
        inside ^= intersects
    return inside


def sample_flux_at_points(kernel: Any, points) -> np.ndarray:
    """core/fusion_kernel_free_boundary.py:156-159 on the device (``kernel.Psi`` is uploaded; resident ensembles use
    ``EnsembleSolver.sample_psi`` directly)."""
    obs = _as_finite_points(points, name="points")
    return sample_psi_grid(kernel.Psi, kernel.R, kernel.Z, obs)


def sample_psi_grid(psi, r_axis, z_axis, points) -> np.ndarray:
    """Device bilinear sampling of a host psi array (nz, nr) or (batch, nz, nr) at ``points`` (n, 2)."""
    from .ensemble import EnsembleSolver

    psi = as_f64(psi)
    batched = psi.ndim == 3
    p3 = psi if batched else psi[None]
    r_axis, z_axis = as_f64(r_axis), as_f64(z_axis)
    cfg = {"grid_resolution": [r_axis.size, z_axis.size],
           "dimensions": {"R_min": float(r_axis[0]), "R_max": float(r_axis[-1]), "Z_min": float(z_axis[0]), "Z_max": float(z_axis[-1])},
           "physics": {"plasma_current_target": 0.0, "vacuum_permeability": 1.0}, "coils": [],
           "solver": {"max_iterations": 1, "convergence_threshold": 1e-4}}
    with EnsembleSolver(cfg, batch=p3.shape[0], r_axis=r_axis, z_axis=z_axis) as es:
        es.set_psi(np.ascontiguousarray(p3))
        out = es.sample_psi(points)
    return out if batched else out[0]


def _validate_probe_directions(directions, *, length: int) -> list[str]:
    if len(directions) != length:
        raise ValueError("b_probe_directions must have one entry per b_probe_point.")
    out = [str(direction).upper() for direction in directions]
    if [d for d in out if d not in {"R", "Z"}]:
        raise ValueError("b_probe_directions entries must be 'R' or 'Z'.")
    return out


def build_magnetic_probe_response_matrix(kernel: Any, coils: CoilSet, *, flux_points=None, b_probe_points=None,
                                         b_probe_directions=None) -> np.ndarray:
    """Diagnostic response matrix, rows = flux loops then B probes, columns = coils
    (core/fusion_kernel_free_boundary.py:315-373); every entry is evaluated by one device kernel."""
    del kernel
    if len(coils.positions) < 1:
        raise ValueError("CoilSet.positions must contain at least one coil.")
    if len(coils.currents) != len(coils.positions):
        raise ValueError("CoilSet.currents length must match number of coil positions.")
    flux_obs = None if flux_points is None else _as_finite_points(flux_points, name="flux_points")
    b_obs = None if b_probe_points is None else _as_finite_points(b_probe_points, name="b_probe_points")
    if flux_obs is None and b_obs is None:
        raise ValueError("At least one flux point or B probe point must be provided.")
    directions: list[str] = []
    if b_obs is not None:
        if b_probe_directions is None:
            raise ValueError("b_probe_directions must be provided with b_probe_points.")
        directions = _validate_probe_directions(b_probe_directions, length=int(b_obs.shape[0]))
    pos = np.array(coils.positions, dtype=np.float64).reshape(-1, 2)
    if not np.all(np.isfinite(pos)) or np.any(pos[:, 0] <= 0.0):
        raise ValueError("Green's-function radii must be positive.")
    for obs in (flux_obs, b_obs):
        if obs is not None and np.any(obs[:, 0] <= 0.0):
            raise ValueError("Green's-function radii must be positive.")
    n_flux = 0 if flux_obs is None else flux_obs.shape[0]
    n_b = 0 if b_obs is None else b_obs.shape[0]
    L = _lib.require_gpu()
    cr, cz = np.ascontiguousarray(pos[:, 0]), np.ascontiguousarray(pos[:, 1])
    ct = np.ascontiguousarray(np.array([int(t) for t in coils.turns_array()], dtype=np.float64))
    fr = fz = br = bz = None
    if flux_obs is not None:
        fr, fz = np.ascontiguousarray(flux_obs[:, 0]), np.ascontiguousarray(flux_obs[:, 1])
    if b_obs is not None:
        br, bz = np.ascontiguousarray(b_obs[:, 0]), np.ascontiguousarray(b_obs[:, 1])
    bd = np.array([0 if d == "R" else 1 for d in directions], dtype=np.int32)
    out = np.empty((n_flux + n_b, len(pos)))
    check(L.scpn_probe_response(len(pos), cr, cz, ptr(ct), n_flux, ptr(fr), ptr(fz), n_b, ptr(br), ptr(bz), ptr(bd), out))
    if not np.all(np.isfinite(out)):
        raise ValueError("Magnetic probe response matrix contains non-finite entries.")
    return out


def reconstruct_boundary_flux_from_coils(kernel: Any, coils: CoilSet, *, boundary_points, limiter_points=None, axis_point=None,
                                         x_points=None, target_flux=None) -> dict[str, Any]:
    """Forward free-boundary vacuum contract (core/fusion_kernel_free_boundary.py:162-270): coil Green's-function flux on the
    boundary / limiter / axis / X-point sets plus residual diagnostics.  Responses come from the device kernel."""
    obs = _as_finite_points(boundary_points, name="boundary_points")
    if len(coils.positions) < 1:
        raise ValueError("CoilSet.positions must contain at least one coil.")
    currents = _as_finite_vector(coils.currents, name="currents", length=len(coils.positions))
    response = build_mutual_inductance_matrix(kernel, coils, obs)
    reconstructed = response.T @ currents
    if not np.all(np.isfinite(reconstructed)):
        raise ValueError("reconstructed boundary flux contains non-finite values.")
    diagnostics: dict[str, Any] = {
        "boundary_points": obs, "reconstructed_flux": reconstructed, "response_matrix": response,
        "response_rank": int(np.linalg.matrix_rank(response)), "point_count": int(obs.shape[0]),
        "coil_count": int(len(coils.positions)), "limiter_point_count": 0, "limiter_flux": np.array([], dtype=np.float64),
        "min_limiter_distance_m": None, "axis_point": None, "axis_flux": None, "x_point_count": 0,
        "x_point_flux": np.array([], dtype=np.float64), "x_point_flux_span": None, "x_point_pair_symmetry_abs_error": None,
    }
    if limiter_points is not None:
        limiter_obs = _as_finite_points(limiter_points, name="limiter_points")
        limiter_flux = build_mutual_inductance_matrix(kernel, coils, limiter_obs).T @ currents
        distances = np.linalg.norm(limiter_obs[:, None, :] - obs[None, :, :], axis=2)
        mask = _points_inside_polygon(obs, limiter_obs)
        frac = float(np.mean(mask)) if mask.size else 0.0
        diagnostics.update({"limiter_points": limiter_obs, "limiter_flux": limiter_flux, "limiter_point_count": int(limiter_obs.shape[0]),
                            "min_limiter_distance_m": float(np.min(distances)), "boundary_containment_fraction": frac,
                            "boundary_containment_pass": bool(frac >= 1.0)})
    if axis_point is not None:
        axis_obs = _as_finite_points(np.asarray(axis_point, dtype=np.float64).reshape(1, 2), name="axis_point")
        axis_response = build_mutual_inductance_matrix(kernel, coils, axis_obs)
        diagnostics.update({"axis_point": axis_obs[0], "axis_flux": float((axis_response.T @ currents)[0])})
    if x_points is not None:
        x_obs = _as_finite_points(x_points, name="x_points")
        x_flux = build_mutual_inductance_matrix(kernel, coils, x_obs).T @ currents
        span = float(np.max(x_flux) - np.min(x_flux)) if x_flux.size else None
        sym = None
        ax = diagnostics.get("axis_point")
        if ax is not None and x_obs.shape[0] == 2:
            same_r = abs(float(x_obs[0, 0] - x_obs[1, 0])) <= 1.0e-9
            mirrored = abs(float(x_obs[0, 1] + x_obs[1, 1] - 2.0 * ax[1])) <= 1.0e-9
            if same_r and mirrored:
                sym = float(abs(x_flux[0] - x_flux[1]))
        diagnostics.update({"x_points": x_obs, "x_point_flux": x_flux, "x_point_count": int(x_obs.shape[0]), "x_point_flux_span": span,
                            "x_point_pair_symmetry_abs_error": sym})
    if target_flux is not None:
        target = _as_finite_vector(target_flux, name="target_flux", length=int(obs.shape[0]))
        residual = reconstructed - target
        diagnostics.update({"target_flux": target, "residual": residual,
                            "rmse": float(np.sqrt(np.mean(residual**2))) if residual.size else 0.0,
                            "max_abs_error": float(np.max(np.abs(residual))) if residual.size else 0.0})
    return diagnostics


def reconstruct_coil_currents_from_magnetic_probes(kernel: Any, coils: CoilSet, *, flux_points=None, flux_measurements=None,
                                                   b_probe_points=None, b_probe_directions=None, b_probe_measurements=None,
                                                   measurement_sigma=None, tikhonov_alpha: float = 1.0e-6) -> dict[str, Any]:
    """Bounded, weighted Tikhonov fit of the coil currents to flux-loop / B-probe measurements
    (core/fusion_kernel_free_boundary.py:376-488).  The response matrix is built on the device; the
    (rows + n_coils) x n_coils ``lsq_linear`` stays on the host as in the reference."""
    response = build_magnetic_probe_response_matrix(kernel, coils, flux_points=flux_points, b_probe_points=b_probe_points,
                                                    b_probe_directions=b_probe_directions)
    targets = []
    if flux_points is not None:
        if flux_measurements is None:
            raise ValueError("flux_measurements must be provided with flux_points.")
        n_flux = _as_finite_points(flux_points, name="flux_points").shape[0]
        targets.append(_as_finite_vector(flux_measurements, name="flux_measurements", length=n_flux))
    elif flux_measurements is not None:
        raise ValueError("flux_points must be provided with flux_measurements.")
    if b_probe_points is not None:
        if b_probe_measurements is None:
            raise ValueError("b_probe_measurements must be provided with b_probe_points.")
        n_probe = _as_finite_points(b_probe_points, name="b_probe_points").shape[0]
        targets.append(_as_finite_vector(b_probe_measurements, name="b_probe_measurements", length=n_probe))
    elif b_probe_measurements is not None:
        raise ValueError("b_probe_points must be provided with b_probe_measurements.")
    if not targets:
        raise ValueError("At least one measurement vector must be provided.")
    target = np.concatenate(targets).astype(np.float64, copy=False)
    if target.shape != (response.shape[0],):
        raise ValueError("measurement vector length must match response rows.")
    weights = np.ones(response.shape[0], dtype=np.float64)
    if measurement_sigma is not None:
        sigma = _as_finite_vector(measurement_sigma, name="measurement_sigma", length=response.shape[0])
        if np.any(sigma <= 0.0):
            raise ValueError("measurement_sigma must contain finite positive values only.")
        weights = 1.0 / sigma
    alpha = float(tikhonov_alpha)
    if not np.isfinite(alpha) or alpha < 0.0:
        raise ValueError("tikhonov_alpha must be finite and non-negative.")
    n_coils = len(coils.positions)
    weighted_response = response * weights[:, None]
    weighted_target = target * weights
    prior = np.asarray(coils.currents, dtype=np.float64).reshape(-1)
    if prior.shape != (n_coils,) or not np.all(np.isfinite(prior)):
        raise ValueError("CoilSet.currents must be finite with one entry per coil.")
    if alpha > 0.0:
        A = np.vstack([weighted_response, np.sqrt(alpha) * np.eye(n_coils, dtype=np.float64)])
        b = np.concatenate([weighted_target, np.sqrt(alpha) * prior])
    else:
        A, b = weighted_response, weighted_target
    if coils.current_limits is not None:
        limits = _as_finite_vector(coils.current_limits, name="current_limits", length=n_coils)
        if np.any(limits <= 0.0):
            raise ValueError("current_limits must contain finite positive values only.")
        lb, ub = -np.abs(limits), np.abs(limits)
    else:
        lb, ub = -np.inf * np.ones(n_coils), np.inf * np.ones(n_coils)
    from scipy.optimize import lsq_linear

    result = lsq_linear(A, b, bounds=(lb, ub), method="trf")
    if not bool(getattr(result, "success", False)) or not np.all(np.isfinite(result.x)):
        raise RuntimeError(f"Magnetic probe inverse reconstruction failed: {getattr(result, 'message', '')}")
    currents = np.asarray(result.x, dtype=np.float64)
    residual = response @ currents - target
    weighted_residual = residual * weights
    return {
        "coil_currents": currents, "residual": residual, "weighted_residual": weighted_residual,
        "residual_rms": float(np.sqrt(np.mean(residual**2))) if residual.size else 0.0,
        "weighted_residual_rms": float(np.sqrt(np.mean(weighted_residual**2))) if weighted_residual.size else 0.0,
        "response_rank": int(np.linalg.matrix_rank(response)),
        "response_condition": float(np.linalg.cond(response)) if response.size else float("inf"),
        "active_bounds": int(np.count_nonzero(np.isclose(currents, lb) | np.isclose(currents, ub))),
    }


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


def _kernel_boundary_points(kernel: Any) -> np.ndarray:
    """Unique computational-boundary points in (R, Z) order: bottom row, right column, top row reversed, left column reversed
    (core/fusion_kernel_free_boundary.py:608-620)."""
    r, z = np.asarray(kernel.R, dtype=np.float64), np.asarray(kernel.Z, dtype=np.float64)
    if r.ndim != 1 or z.ndim != 1 or r.size < 2 or z.size < 2:
        raise ValueError("kernel R/Z axes must be one-dimensional with at least two points.")
    if not np.all(np.isfinite(r)) or not np.all(np.isfinite(z)):
        raise ValueError("kernel R/Z axes must contain finite values only.")
    bottom = np.column_stack([r, np.full(r.shape, z[0])])
    right = np.column_stack([np.full(z[1:].shape, r[-1]), z[1:]])
    top = np.column_stack([r[-2::-1], np.full(r[-2::-1].shape, z[-1])])
    left = np.column_stack([np.full(z[-2:0:-1].shape, r[0]), z[-2:0:-1]])
    return np.vstack([bottom, right, top, left]).astype(np.float64, copy=False)


def solve_free_boundary(kernel: Any, coils: CoilSet, max_outer_iter: int = 20, tol: float = 1e-4,
                        optimize_shape: bool = False, tikhonov_alpha: float = 1e-4, limiter_points=None, axis_point=None,
                        x_points=None) -> dict[str, Any]:
    """core/fusion_kernel_free_boundary.py:623-739.  With this package's ``FusionKernel`` and ``optimize_shape=False`` the whole
    outer loop runs on the device; with ``optimize_shape=True`` (or any other kernel object exposing ``Psi``, ``R``, ``Z`` and
    ``solve_equilibrium``) each outer iteration is the kernel's solve followed by the host-side bounded least-squares update
    of the coil currents, as in the reference.  The closing boundary-flux reconstruction diagnostics (:708-739) are
    evaluated from the device Green's-function kernel."""
    if max_outer_iter < 1:
        raise ValueError("max_outer_iter must be >= 1.")
    if not np.isfinite(tol) or tol < 0.0:
        raise ValueError("tol must be finite and >= 0.")
    shape_optimization = None
    extra: dict[str, Any] = {}
    newton = getattr(kernel, "cfg", {}).get("solver", {}).get("solver_method", "multigrid") == "newton"
    if not (optimize_shape and coils.target_flux_points is not None) and hasattr(kernel, "_solve_free_boundary_device") and not newton:
        extra = kernel._solve_free_boundary_device(coils, max_outer_iter, tol)
        outer_iterations, diff = extra["outer_iterations"], extra["final_diff"]
        psi_ext = compute_external_flux(kernel, coils)
    else:
        psi_ext = compute_external_flux(kernel, coils)
        diff, outer = float("inf"), 0
        for outer in range(max_outer_iter):
            kernel.Psi[0, :], kernel.Psi[-1, :] = psi_ext[0, :], psi_ext[-1, :]
            kernel.Psi[:, 0], kernel.Psi[:, -1] = psi_ext[:, 0], psi_ext[:, -1]
            psi_old = kernel.Psi.copy()
            kernel.solve_equilibrium(preserve_initial_state=True, boundary_flux=psi_ext)
            if optimize_shape and coils.target_flux_points is not None:
                target_psi = resolve_shape_target_flux(kernel, coils)
                response = build_mutual_inductance_matrix(kernel, coils, coils.target_flux_points)
                optimise = getattr(kernel, "optimize_coil_currents", None)  # kernel method first: keeps caller hooks working
                new_currents = (optimise(coils, target_psi, tikhonov_alpha=tikhonov_alpha) if optimise is not None
                                else optimize_coil_currents(kernel, coils, target_psi, tikhonov_alpha=tikhonov_alpha))
                new_currents = np.asarray(new_currents, dtype=np.float64).reshape(-1)
                if new_currents.shape != (len(coils.positions),):
                    raise ValueError("optimised coil current vector length must match coil count.")
                if not np.all(np.isfinite(new_currents)):
                    raise ValueError("optimised coil currents must contain finite values only.")
                achieved = response.T @ new_currents
                residual = achieved - target_psi
                flux_rmse = float(np.sqrt(np.mean(residual**2)))
                active_bounds = 0
                if coils.current_limits is not None:
                    limits = _as_finite_vector(coils.current_limits, name="current_limits", length=len(coils.positions))
                    active_bounds = int(np.count_nonzero(np.isclose(np.abs(new_currents), limits, rtol=0.0)))
                shape_optimization = {
                    "solver_mode": "free_boundary_solver_shape_current_optimization",
                    "target_point_count": int(target_psi.shape[0]), "coil_count": int(len(coils.positions)),
                    "response_rank": int(np.linalg.matrix_rank(response.T)), "response_condition": float(np.linalg.cond(response.T)),
                    "flux_rmse": flux_rmse,
                    "flux_relative_rmse": float(flux_rmse / max(float(np.sqrt(np.mean(target_psi**2))), 1.0)),
                    "max_abs_flux_residual": float(np.max(np.abs(residual))), "active_current_bounds": active_bounds,
                    "target_flux": target_psi.copy(), "achieved_flux": achieved.astype(np.float64, copy=False),
                }
                coils.currents = new_currents
                psi_ext = compute_external_flux(kernel, coils)
            diff = float(np.max(np.abs(kernel.Psi - psi_old)))
            if diff < tol:
                break
        outer_iterations = outer + 1
    boundary_target = np.concatenate([psi_ext[0, :], psi_ext[1:, -1], psi_ext[-1, -2::-1], psi_ext[-2:0:-1, 0]]).astype(np.float64)
    boundary_reconstruction = reconstruct_boundary_flux_from_coils(
        kernel, coils, boundary_points=_kernel_boundary_points(kernel), limiter_points=limiter_points, axis_point=axis_point,
        x_points=x_points, target_flux=boundary_target)
    out = {"outer_iterations": outer_iterations, "final_diff": diff, "coil_currents": np.asarray(coils.currents, dtype=np.float64).copy(),
           "vacuum_boundary_abs_error": boundary_reconstruction["max_abs_error"], "boundary_reconstruction": boundary_reconstruction,
           "shape_optimization": shape_optimization}
    for key in ("picard_iterations", "converged"):  # device-loop telemetry (not in the reference's dict)
        if key in extra:
            out[key] = extra[key]
    return out


__all__ = ["CoilSet", "coil_flux_grid", "green_points", "green_function", "compute_external_flux",
           "build_mutual_inductance_matrix", "build_magnetic_probe_response_matrix", "sample_flux_at_points", "sample_psi_grid",
           "reconstruct_boundary_flux_from_coils", "reconstruct_coil_currents_from_magnetic_probes", "interp_psi",
           "resolve_shape_target_flux", "optimize_coil_currents", "solve_free_boundary"]

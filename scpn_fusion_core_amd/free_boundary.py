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


def _require(ok: bool, message: str) -> None:
    if not ok:
        raise ValueError(message)


def green_function(R_src: float, Z_src: float, R_obs: float, Z_obs: float) -> float:
    """Flux at (R_obs, Z_obs) of a unit-current filament at (R_src, Z_src), core/fusion_kernel_free_boundary.py:31-55 --
    one point through the device kernel."""
    _require(bool(np.isfinite([R_src, Z_src, R_obs, Z_obs]).all()), "Green's-function coordinates must be finite.")
    _require(min(R_src, R_obs) > 0.0, "Green's-function radii must be positive.")
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


def _finite_array(value, name: str, *, pairs: bool = False, length: int | None = None) -> np.ndarray:
    """float64 view of a caller-supplied vector (``pairs=False``, optionally of a fixed length) or (n, 2) point list, finite."""
    arr = np.asarray(value, dtype=np.float64)
    if pairs:
        _require(arr.ndim == 2 and arr.shape[1] == 2 and arr.shape[0] >= 1, f"{name} must have shape (n_points, 2) with n_points > 0.")
    else:
        arr = arr.reshape(-1)
        _require(length is None or arr.shape == (length,), f"{name} must have length {length}.")
    _require(bool(np.all(np.isfinite(arr))), f"{name} must contain finite values only.")
    return arr


def _as_finite_vector(value, *, name: str, length: int | None = None) -> np.ndarray:
    return _finite_array(value, name, length=length)


def _as_finite_points(value, *, name: str) -> np.ndarray:
    return _finite_array(value, name, pairs=True)


def _points_inside_polygon(points, polygon) -> np.ndarray:
    """Even-odd containment mask (core/fusion_kernel_free_boundary.py:114-134), all edges at once: a point is inside when the
    ray towards +R crosses an odd number of edges (edge k runs from vertex k to vertex k+1, the last one closes the contour)."""
    pts = _finite_array(points, "points", pairs=True)
    ring = _finite_array(polygon, "polygon", pairs=True)
    _require(ring.shape[0] >= 3, "polygon must contain at least three points.")
    head, tail = ring, np.roll(ring, -1, axis=0)
    pr, pz = pts[:, 0][:, None], pts[:, 1][:, None]                # (n_pts, 1) against (n_edges,)
    straddles = (head[:, 1] > pz) != (tail[:, 1] > pz)
    rise = np.maximum(np.abs(tail[:, 1] - head[:, 1]), 1.0e-300)
    crossing_r = (tail[:, 0] - head[:, 0]) * (pz - head[:, 1]) / rise + head[:, 0]
    return (np.count_nonzero(straddles & (pr < crossing_r), axis=1) % 2).astype(bool)


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
    _require(len(directions) == length, "b_probe_directions must have one entry per b_probe_point.")
    axes = [str(d).upper() for d in directions]
    _require(set(axes) <= {"R", "Z"}, "b_probe_directions entries must be 'R' or 'Z'.")
    return axes


def build_magnetic_probe_response_matrix(kernel: Any, coils: CoilSet, *, flux_points=None, b_probe_points=None,
                                         b_probe_directions=None) -> np.ndarray:
    """Diagnostic response matrix, rows = flux loops then B probes, columns = coils
    (core/fusion_kernel_free_boundary.py:315-373); every entry is evaluated by one device kernel."""
    del kernel
    _require(len(coils.positions) >= 1, "CoilSet.positions must contain at least one coil.")
    _require(len(coils.currents) == len(coils.positions), "CoilSet.currents length must match number of coil positions.")
    flux_obs = _finite_array(flux_points, "flux_points", pairs=True) if flux_points is not None else None
    b_obs = _finite_array(b_probe_points, "b_probe_points", pairs=True) if b_probe_points is not None else None
    _require(flux_obs is not None or b_obs is not None, "At least one flux point or B probe point must be provided.")
    directions: list[str] = []
    if b_obs is not None:
        _require(b_probe_directions is not None, "b_probe_directions must be provided with b_probe_points.")
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
    boundary / limiter / axis / X-point sets plus residual diagnostics.  All point sets go through ONE device evaluation of the
    Green's functions; the per-set fluxes are products of the matching column blocks with the coil currents."""
    named = {"boundary": _finite_array(boundary_points, "boundary_points", pairs=True)}
    if limiter_points is not None:
        named["limiter"] = _finite_array(limiter_points, "limiter_points", pairs=True)
    if axis_point is not None:
        named["axis"] = _finite_array(np.asarray(axis_point, dtype=np.float64).reshape(1, 2), "axis_point", pairs=True)
    if x_points is not None:
        named["x"] = _finite_array(x_points, "x_points", pairs=True)
    n_coils = len(coils.positions)
    _require(n_coils >= 1, "CoilSet.positions must contain at least one coil.")
    currents = _finite_array(coils.currents, "currents", length=n_coils)
    every = build_mutual_inductance_matrix(kernel, coils, np.vstack(list(named.values())))
    blocks, start = {}, 0
    for key, pts in named.items():  # column blocks, contiguous copies (same matrix shapes as a per-set evaluation)
        blocks[key] = np.ascontiguousarray(every[:, start:start + pts.shape[0]])
        start += pts.shape[0]
    flux = {key: blk.T @ currents for key, blk in blocks.items()}
    obs, response, reconstructed = named["boundary"], blocks["boundary"], flux["boundary"]
    _require(bool(np.all(np.isfinite(reconstructed))), "reconstructed boundary flux contains non-finite values.")
    no_flux = np.array([], dtype=np.float64)
    report: dict[str, Any] = dict(
        boundary_points=obs, reconstructed_flux=reconstructed, response_matrix=response,
        response_rank=int(np.linalg.matrix_rank(response)), point_count=int(obs.shape[0]), coil_count=int(n_coils),
        limiter_point_count=0, limiter_flux=no_flux, min_limiter_distance_m=None, axis_point=None, axis_flux=None,
        x_point_count=0, x_point_flux=no_flux.copy(), x_point_flux_span=None, x_point_pair_symmetry_abs_error=None)
    if "limiter" in named:
        wall = named["limiter"]
        gap = np.linalg.norm(wall[:, None, :] - obs[None, :, :], axis=2)
        contained = _points_inside_polygon(obs, wall)
        share = float(np.mean(contained)) if contained.size else 0.0
        report.update(limiter_points=wall, limiter_flux=flux["limiter"], limiter_point_count=int(wall.shape[0]),
                      min_limiter_distance_m=float(np.min(gap)), boundary_containment_fraction=share,
                      boundary_containment_pass=bool(share >= 1.0))
    if "axis" in named:
        report.update(axis_point=named["axis"][0], axis_flux=float(flux["axis"][0]))
    if "x" in named:
        xp, xf = named["x"], flux["x"]
        pair_error = None
        axis = report["axis_point"]
        if axis is not None and xp.shape[0] == 2:  # an up-down mirrored X-point pair at one major radius must see the same flux
            aligned = abs(float(xp[0, 0] - xp[1, 0])) <= 1.0e-9 and abs(float(xp[0, 1] + xp[1, 1] - 2.0 * axis[1])) <= 1.0e-9
            if aligned:
                pair_error = float(abs(xf[0] - xf[1]))
        report.update(x_points=xp, x_point_flux=xf, x_point_count=int(xp.shape[0]),
                      x_point_flux_span=float(np.max(xf) - np.min(xf)) if xf.size else None, x_point_pair_symmetry_abs_error=pair_error)
    if target_flux is not None:
        wanted = _finite_array(target_flux, "target_flux", length=int(obs.shape[0]))
        miss = reconstructed - wanted
        report.update(target_flux=wanted, residual=miss, rmse=float(np.sqrt(np.mean(miss**2))) if miss.size else 0.0,
                      max_abs_error=float(np.max(np.abs(miss))) if miss.size else 0.0)
    return report


def _current_bounds(coils: CoilSet, n_coils: int, *, strictly_positive: bool) -> tuple[np.ndarray, np.ndarray]:
    """Box bounds (-|limit|, +|limit|) of the coil currents; unbounded without ``current_limits``.  ``strictly_positive`` is the
    probe-inversion contract (core/fusion_kernel_free_boundary.py:452-455), finiteness + length the shape-fit one (:531-536)."""
    if coils.current_limits is None:
        return np.full(n_coils, -np.inf), np.full(n_coils, np.inf)
    if strictly_positive:
        limits = _as_finite_vector(coils.current_limits, name="current_limits", length=n_coils)
        _require(not np.any(limits <= 0.0), "current_limits must contain finite positive values only.")
    else:
        limits = np.asarray(coils.current_limits, dtype=np.float64).reshape(-1)
        _require(limits.shape[0] == n_coils, "current_limits must have one entry per coil.")
        _require(bool(np.all(np.isfinite(limits))), "current_limits must contain finite values only.")
    mag = np.abs(limits)
    return -mag, mag


def _ridge_system(rows: np.ndarray, rhs: np.ndarray, alpha: float, prior: np.ndarray | None) -> tuple[np.ndarray, np.ndarray]:
    """Augmented least-squares system of a Tikhonov fit: ``rows`` stacked on sqrt(alpha)*I, ``rhs`` on sqrt(alpha)*prior (zeros
    when the regulariser pulls towards zero current)."""
    n = rows.shape[1]
    damp = np.sqrt(alpha)
    anchor = np.zeros(n) if prior is None else damp * prior
    return np.vstack([rows, damp * np.eye(n, dtype=np.float64)]), np.concatenate([rhs, anchor])


def reconstruct_coil_currents_from_magnetic_probes(kernel: Any, coils: CoilSet, *, flux_points=None, flux_measurements=None,
                                                   b_probe_points=None, b_probe_directions=None, b_probe_measurements=None,
                                                   measurement_sigma=None, tikhonov_alpha: float = 1.0e-6) -> dict[str, Any]:
    """Bounded, weighted Tikhonov fit of the coil currents to flux-loop / B-probe measurements
    (core/fusion_kernel_free_boundary.py:376-488).  The response matrix is built on the device; the
    (rows + n_coils) x n_coils ``lsq_linear`` stays on the host as in the reference."""
    response = build_magnetic_probe_response_matrix(kernel, coils, flux_points=flux_points, b_probe_points=b_probe_points,
                                                    b_probe_directions=b_probe_directions)
    # one measurement block per diagnostic kind, in response-row order; a block needs both its geometry and its readings
    blocks = (("flux_points", flux_points, "flux_measurements", flux_measurements),
              ("b_probe_points", b_probe_points, "b_probe_measurements", b_probe_measurements))
    readings = []
    for geo_name, geo, val_name, val in blocks:
        if geo is None:
            _require(val is None, f"{geo_name} must be provided with {val_name}.")
            continue
        _require(val is not None, f"{val_name} must be provided with {geo_name}.")
        readings.append(_as_finite_vector(val, name=val_name, length=_as_finite_points(geo, name=geo_name).shape[0]))
    _require(bool(readings), "At least one measurement vector must be provided.")
    target = np.concatenate(readings).astype(np.float64, copy=False)
    n_rows, n_coils = response.shape[0], len(coils.positions)
    _require(target.shape == (n_rows,), "measurement vector length must match response rows.")
    weights = np.ones(n_rows, dtype=np.float64)
    if measurement_sigma is not None:
        sigma = _as_finite_vector(measurement_sigma, name="measurement_sigma", length=n_rows)
        _require(not np.any(sigma <= 0.0), "measurement_sigma must contain finite positive values only.")
        weights = 1.0 / sigma
    alpha = float(tikhonov_alpha)
    _require(bool(np.isfinite(alpha)) and alpha >= 0.0, "tikhonov_alpha must be finite and non-negative.")
    prior = np.asarray(coils.currents, dtype=np.float64).reshape(-1)
    _require(prior.shape == (n_coils,) and bool(np.all(np.isfinite(prior))), "CoilSet.currents must be finite with one entry per coil.")
    A, b = response * weights[:, None], target * weights
    if alpha > 0.0:
        A, b = _ridge_system(A, b, alpha, prior)
    lb, ub = _current_bounds(coils, n_coils, strictly_positive=True)
    from scipy.optimize import lsq_linear

    fit = lsq_linear(A, b, bounds=(lb, ub), method="trf")
    if not bool(getattr(fit, "success", False)) or not np.all(np.isfinite(fit.x)):
        raise RuntimeError(f"Magnetic probe inverse reconstruction failed: {getattr(fit, 'message', '')}")
    currents = np.asarray(fit.x, dtype=np.float64)
    misfit = response @ currents - target
    rms = lambda v: float(np.sqrt(np.mean(v**2))) if v.size else 0.0  # noqa: E731
    return {
        "coil_currents": currents, "residual": misfit, "weighted_residual": misfit * weights,
        "residual_rms": rms(misfit), "weighted_residual_rms": rms(misfit * weights),
        "response_rank": int(np.linalg.matrix_rank(response)),
        "response_condition": float(np.linalg.cond(response)) if response.size else float("inf"),
        "active_bounds": int(np.count_nonzero(np.isclose(currents, lb) | np.isclose(currents, ub))),
    }


def _bilinear_psi(kernel: Any, points: np.ndarray) -> np.ndarray:
    """Bilinear samples of ``kernel.Psi`` at ``points[:, (R, Z)]`` (cell index from the sorted axes, offsets clamped to the cell;
    the four corner terms are added in the order SW, SE, NW, NE as core/fusion_kernel_free_boundary.py:562-580 does)."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    cell_r = np.clip(np.searchsorted(kernel.R, pts[:, 0]) - 1, 0, kernel.NR - 2)
    cell_z = np.clip(np.searchsorted(kernel.Z, pts[:, 1]) - 1, 0, kernel.NZ - 2)
    fr = np.clip((pts[:, 0] - np.asarray(kernel.R)[cell_r]) / kernel.dR, 0.0, 1.0)
    fz = np.clip((pts[:, 1] - np.asarray(kernel.Z)[cell_z]) / kernel.dZ, 0.0, 1.0)
    psi = np.asarray(kernel.Psi)
    out = (1 - fr) * (1 - fz) * psi[cell_z, cell_r]
    out = out + fr * (1 - fz) * psi[cell_z, cell_r + 1]
    out = out + (1 - fr) * fz * psi[cell_z + 1, cell_r]
    return out + fr * fz * psi[cell_z + 1, cell_r + 1]


def interp_psi(kernel: Any, R_pt: float, Z_pt: float) -> float:
    """Bilinear sample of kernel.Psi at one point (core/fusion_kernel_free_boundary.py:562-580) -- host-side, 4 grid values."""
    return float(_bilinear_psi(kernel, np.array([[R_pt, Z_pt]], dtype=np.float64))[0])


def _control_points(coils: CoilSet, purpose: str) -> np.ndarray:
    """The isoflux control points of a CoilSet as an (n, 2) array (n > 0)."""
    _require(coils.target_flux_points is not None, f"CoilSet.target_flux_points must be set for {purpose}.")
    pts = np.asarray(coils.target_flux_points, dtype=np.float64)
    _require(pts.ndim == 2 and pts.shape[1] == 2 and pts.shape[0] > 0, "target_flux_points must have shape (n_points, 2) with n_points > 0.")
    return pts


def resolve_shape_target_flux(kernel: Any, coils: CoilSet) -> np.ndarray:
    """core/fusion_kernel_free_boundary.py:583-605: explicit target values, or the isoflux mean of bilinear psi samples."""
    pts = _control_points(coils, "shape optimisation")
    if coils.target_flux_values is None:
        return np.full(pts.shape[0], float(np.mean(_bilinear_psi(kernel, pts))), dtype=np.float64)
    explicit = np.asarray(coils.target_flux_values, dtype=np.float64).reshape(-1)
    _require(explicit.shape[0] == pts.shape[0], "target_flux_values must have the same length as target_flux_points.")
    _require(bool(np.all(np.isfinite(explicit))), "target_flux_values must contain finite values only.")
    return explicit


def optimize_coil_currents(kernel: Any, coils: CoilSet, target_flux, tikhonov_alpha: float = 1e-4) -> np.ndarray:
    """Bounded Tikhonov least squares for the coil currents (core/fusion_kernel_free_boundary.py:491-559).  The response
    matrix comes from the device Green's-function kernel; the (n_pts + n_coils) x n_coils ``lsq_linear`` stays on the host.
    A failed fit falls back to the prior currents clipped into the bounds, as the reference does."""
    pts = _control_points(coils, "optimisation")
    _require(len(coils.positions) > 0, "CoilSet.positions must contain at least one coil.")
    _require(len(coils.currents) == len(coils.positions), "CoilSet.currents length must match number of coil positions.")
    _require(bool(np.isfinite(tikhonov_alpha)) and tikhonov_alpha >= 0.0, "tikhonov_alpha must be finite and non-negative.")
    _require(bool(np.all(np.isfinite(pts))), "target_flux_points must contain finite values only.")
    wanted = np.asarray(target_flux, dtype=np.float64).reshape(-1)
    _require(wanted.shape[0] == pts.shape[0], "target_flux must have the same length as target_flux_points.")
    _require(bool(np.all(np.isfinite(wanted))), "target_flux must contain finite values only.")
    mutual = build_mutual_inductance_matrix(kernel, coils, pts)  # (n_coils, n_pts), device Green's functions
    _require(bool(np.all(np.isfinite(mutual))), "Mutual inductance matrix contains non-finite entries.")
    A, b = _ridge_system(mutual.T, wanted, float(tikhonov_alpha), None)
    lb, ub = _current_bounds(coils, mutual.shape[0], strictly_positive=False)
    from scipy.optimize import lsq_linear

    fit = lsq_linear(A, b, bounds=(lb, ub), method="trf")
    if bool(getattr(fit, "success", False)) and np.all(np.isfinite(fit.x)):
        return np.asarray(fit.x, dtype=np.float64)
    return np.clip(np.asarray(coils.currents, dtype=np.float64), lb, ub).astype(np.float64)


def _perimeter_indices(nz: int, nr: int) -> tuple[np.ndarray, np.ndarray]:
    """(iz, ir) of every computational-boundary node once, walking bottom row -> right column -> top row backwards -> left column
    backwards (the order of core/fusion_kernel_free_boundary.py:608-620 and :716-723)."""
    iz = np.concatenate([np.zeros(nr, int), np.arange(1, nz), np.full(nr - 1, nz - 1), np.arange(nz - 2, 0, -1)])
    ir = np.concatenate([np.arange(nr), np.full(nz - 1, nr - 1), np.arange(nr - 2, -1, -1), np.zeros(nz - 2, int)])
    return iz, ir


def _kernel_boundary_points(kernel: Any) -> np.ndarray:
    """Unique computational-boundary points as (R, Z) rows in perimeter order."""
    r, z = np.asarray(kernel.R, dtype=np.float64), np.asarray(kernel.Z, dtype=np.float64)
    _require(r.ndim == 1 and z.ndim == 1 and r.size >= 2 and z.size >= 2, "kernel R/Z axes must be one-dimensional with at least two points.")
    _require(bool(np.all(np.isfinite(r)) and np.all(np.isfinite(z))), "kernel R/Z axes must contain finite values only.")
    iz, ir = _perimeter_indices(z.size, r.size)
    return np.column_stack([r[ir], z[iz]])


def _shape_step(kernel: Any, coils: CoilSet, tikhonov_alpha: float) -> dict[str, Any]:
    """One shape-control update inside the free-boundary outer loop (core/fusion_kernel_free_boundary.py:663-706): isoflux
    target from the fresh psi, bounded least-squares currents (a ``kernel.optimize_coil_currents`` hook wins over the module
    function), the achieved-flux report, and the new currents written into ``coils``."""
    wanted = resolve_shape_target_flux(kernel, coils)
    gain = build_mutual_inductance_matrix(kernel, coils, coils.target_flux_points).T  # (n_points, n_coils)
    hook = getattr(kernel, "optimize_coil_currents", None)
    fitted = hook(coils, wanted, tikhonov_alpha=tikhonov_alpha) if hook is not None else optimize_coil_currents(
        kernel, coils, wanted, tikhonov_alpha=tikhonov_alpha)
    fitted = np.asarray(fitted, dtype=np.float64).reshape(-1)
    n_coils = len(coils.positions)
    _require(fitted.shape == (n_coils,), "optimised coil current vector length must match coil count.")
    _require(bool(np.all(np.isfinite(fitted))), "optimised coil currents must contain finite values only.")
    got = gain @ fitted
    miss = got - wanted
    rmse = float(np.sqrt(np.mean(miss**2)))
    at_bound = 0
    if coils.current_limits is not None:
        caps = _finite_array(coils.current_limits, "current_limits", length=n_coils)
        at_bound = int(np.count_nonzero(np.isclose(np.abs(fitted), caps, rtol=0.0)))
    coils.currents = fitted
    return dict(solver_mode="free_boundary_solver_shape_current_optimization", target_point_count=int(wanted.shape[0]),
                coil_count=int(n_coils), response_rank=int(np.linalg.matrix_rank(gain)), response_condition=float(np.linalg.cond(gain)),
                flux_rmse=rmse, flux_relative_rmse=float(rmse / max(float(np.sqrt(np.mean(wanted**2))), 1.0)),
                max_abs_flux_residual=float(np.max(np.abs(miss))), active_current_bounds=at_bound,
                target_flux=wanted.copy(), achieved_flux=got.astype(np.float64, copy=False))


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
        ring = _perimeter_indices(*np.shape(psi_ext))
        reshape = optimize_shape and coils.target_flux_points is not None
        diff, outer_iterations = float("inf"), 0
        while outer_iterations < max_outer_iter:
            outer_iterations += 1
            kernel.Psi[ring] = psi_ext[ring]  # Dirichlet data of this pass
            before = kernel.Psi.copy()
            kernel.solve_equilibrium(preserve_initial_state=True, boundary_flux=psi_ext)
            if reshape:
                shape_optimization = _shape_step(kernel, coils, tikhonov_alpha)
                psi_ext = compute_external_flux(kernel, coils)
            diff = float(np.max(np.abs(kernel.Psi - before)))
            if diff < tol:
                break
    boundary_target = np.asarray(psi_ext, dtype=np.float64)[_perimeter_indices(*np.shape(psi_ext))]
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

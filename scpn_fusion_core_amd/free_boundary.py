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


def solve_free_boundary(kernel: Any, coils: CoilSet, max_outer_iter: int = 20, tol: float = 1e-4,
                        optimize_shape: bool = False, **_unused) -> dict[str, Any]:
    """core/fusion_kernel_free_boundary.py:623-739 for ``optimize_shape=False`` (the whole outer loop
    runs on the device).  ``optimize_shape=True`` (host ``lsq_linear`` between device solves) is
    SURVEY.md section 8 row f3 and not wired yet."""
    if max_outer_iter < 1:
        raise ValueError("max_outer_iter must be >= 1.")
    if not np.isfinite(tol) or tol < 0.0:
        raise ValueError("tol must be finite and >= 0.")
    if optimize_shape and coils.target_flux_points is not None:
        raise NotImplementedError("optimize_shape=True is not on the HIP path yet")
    return kernel._solve_free_boundary_device(coils, max_outer_iter, tol)


__all__ = ["CoilSet", "coil_flux_grid", "green_points", "green_function", "compute_external_flux",
           "build_mutual_inductance_matrix", "interp_psi", "solve_free_boundary"]

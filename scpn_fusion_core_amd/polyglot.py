"""The 'polyglot' Picard + damped-Jacobi lane (core/jax_gs_solver.py:126-167 ``gs_solve_np``) on the HIP path."""

from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr


def gs_solve_hip(R_min, R_max, Z_min, Z_max, NR, NZ, Ip_target, mu0=4e-7 * np.pi, n_picard=80, n_jacobi=200, alpha=0.1,
                 omega_j=0.667, beta_mix=0.5):
    """Same arguments and result as the reference ``gs_solve_np``; psi on the (NZ, NR) grid, zero ring."""
    L = _lib.require_gpu()
    R = np.linspace(R_min, R_max, NR)
    Z = np.linspace(Z_min, Z_max, NZ)
    RR, _ = np.meshgrid(R, Z)
    # Gaussian seed computed with NumPy exactly as the reference does (jax_gs_solver.py:151-158)
    psi = np.exp(-((RR - 0.5 * (R_min + R_max)) ** 2) / 0.5) * 0.01
    psi[0, :] = psi[-1, :] = 0.0
    psi[:, 0] = psi[:, -1] = 0.0
    psi = np.ascontiguousarray(psi)
    out = np.empty((NZ, NR))
    check(L.scpn_gs_solve_np(float(R_min), float(R_max), float(Z_min), float(Z_max), int(NR), int(NZ), R, Z, ptr(psi),
                             float(Ip_target), float(mu0), int(n_picard), int(n_jacobi), float(alpha), float(omega_j),
                             float(beta_mix), out))
    return out


class _OperatorGrid:
    """A validated flux map with the uniform grid it lives on: what the lane's diagnostics hand to the device.

    The checks are the ones the package's other operator mirrors make (a 2-D finite float64 field, a grid of at least 3 x 3,
    ordered finite bounds with R > 0); only the messages are dictated -- they are the reference's
    (core/jax_gs_solver.py:176-187), so callers and tests written against ``gs_delta_star_np`` see the same errors, in its order.
    """

    __slots__ = ("field", "nz", "nr", "d_r", "d_z", "r_axis")

    def __init__(self, psi, r_bounds, z_bounds):
        field = as_f64(psi)
        (r_lo, r_hi), (z_lo, z_hi) = r_bounds, z_bounds
        problems = (
            (field.ndim != 2, "psi must be a two-dimensional array"),
            (field.ndim == 2 and min(field.shape) < 3, "psi shape must be at least 3x3 for centered GS diagnostics"),
            (not (_ordered(r_lo, r_hi) and r_lo > 0.0), "R_min and R_max must be finite, positive, increasing values"),
            (not _ordered(z_lo, z_hi), "Z_min and Z_max must be finite, increasing values"),
            (not np.isfinite(field).all(), "psi must contain only finite values"),
        )
        for failed, message in problems:
            if failed:
                raise ValueError(message)
        self.field = np.ascontiguousarray(field)
        self.nz, self.nr = field.shape
        self.d_r, self.d_z = (r_hi - r_lo) / (self.nr - 1), (z_hi - z_lo) / (self.nz - 1)
        self.r_axis = np.linspace(r_lo, r_hi, self.nr)


def _ordered(lo, hi) -> bool:
    return bool(np.isfinite(lo) and np.isfinite(hi) and lo < hi)


def gs_delta_star_hip(psi, R_min, R_max, Z_min, Z_max):
    """``gs_delta_star_np`` (core/jax_gs_solver.py:196-215): the lane's centred cylindrical operator, zero on the ring."""
    L = _lib.require_gpu()
    g = _OperatorGrid(psi, (R_min, R_max), (Z_min, Z_max))
    out = np.empty((g.nz, g.nr))
    check(L.scpn_np_delta_star(g.field, g.nz, g.nr, g.r_axis, float(g.d_r), float(g.d_z), out))
    return out


def gs_toroidal_current_density_hip(psi, R_min, R_max, Z_min, Z_max, mu0=4e-7 * np.pi):
    """``gs_toroidal_current_density_np`` (:218-234): J_phi = -Delta* psi / (mu0 R) on the interior."""
    if not (np.isfinite(mu0) and mu0 > 0.0):
        raise ValueError("mu0 must be finite and positive")
    g = _OperatorGrid(psi, (R_min, R_max), (Z_min, Z_max))
    ds = gs_delta_star_hip(g.field, R_min, R_max, Z_min, Z_max)
    j = np.zeros_like(g.field)
    j[1:-1, 1:-1] = -ds[1:-1, 1:-1] / (mu0 * g.r_axis[None, 1:-1])
    return j


def gs_total_toroidal_current_hip(psi, R_min, R_max, Z_min, Z_max, mu0=4e-7 * np.pi):
    """``gs_total_toroidal_current_np`` (:237-248)."""
    g = _OperatorGrid(psi, (R_min, R_max), (Z_min, Z_max))
    j = gs_toroidal_current_density_hip(g.field, R_min, R_max, Z_min, Z_max, mu0=mu0)
    return float(np.sum(j[1:-1, 1:-1]) * g.d_r * g.d_z)


def gs_equation_residual_hip(psi, R_min, R_max, Z_min, Z_max, NR, NZ, Ip_target, mu0=4e-7 * np.pi, beta_mix=0.5):
    """``gs_equation_residual_np`` (:251-316): abs_max / relative_max / relative_l2 of the discrete equation imbalance with
    the lane's own source normalisation, evaluated on the device."""
    L = _lib.require_gpu()
    field = as_f64(psi)
    # the reference's checks and messages (core/jax_gs_solver.py:277-284), in its order
    for failed, message in (
            (field.shape != (NZ, NR), f"psi shape must be (NZ, NR)=({NZ}, {NR}), got {field.shape}"),
            (min(NR, NZ) < 3, "Grad-Shafranov residual requires at least a 3 x 3 grid"),
            (any((R_min <= 0.0, R_max <= R_min, Z_max <= Z_min)), "Grad-Shafranov residual requires positive, increasing R and Z bounds"),
            (not np.isfinite(field).all(), "psi must contain only finite values")):
        if failed:
            raise ValueError(message)
    norms = np.empty(3)
    check(L.scpn_np_equation_residual(field, NZ, NR, np.linspace(R_min, R_max, NR), np.linspace(Z_min, Z_max, NZ), float(Ip_target),
                                      float(mu0), float(beta_mix), norms))
    return dict(zip(("abs_max", "relative_max", "relative_l2"), (float(v) for v in norms)))

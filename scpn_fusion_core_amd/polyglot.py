"""The 'polyglot' Picard + damped-Jacobi lane (core/jax_gs_solver.py:126-167 ``gs_solve_np``) on the HIP path."""

from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import check, ptr


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


def _validate_operator_inputs(psi, R_min, R_max, Z_min, Z_max):
    """core/jax_gs_solver.py:168-193."""
    psi_arr = np.asarray(psi, dtype=float)
    if psi_arr.ndim != 2:
        raise ValueError("psi must be a two-dimensional array")
    NZ, NR = psi_arr.shape
    if NR < 3 or NZ < 3:
        raise ValueError("psi shape must be at least 3x3 for centered GS diagnostics")
    if not (np.isfinite(R_min) and np.isfinite(R_max) and 0.0 < R_min < R_max):
        raise ValueError("R_min and R_max must be finite, positive, increasing values")
    if not (np.isfinite(Z_min) and np.isfinite(Z_max) and Z_min < Z_max):
        raise ValueError("Z_min and Z_max must be finite, increasing values")
    if not np.all(np.isfinite(psi_arr)):
        raise ValueError("psi must contain only finite values")
    dR, dZ = (R_max - R_min) / (NR - 1), (Z_max - Z_min) / (NZ - 1)
    return np.ascontiguousarray(psi_arr, dtype=np.float64), NR, NZ, dR, dZ, np.linspace(R_min, R_max, NR).astype(np.float64)


def gs_delta_star_hip(psi, R_min, R_max, Z_min, Z_max):
    """``gs_delta_star_np`` (core/jax_gs_solver.py:196-215): the lane's centred cylindrical operator, zero on the ring."""
    L = _lib.require_gpu()
    psi_arr, NR, NZ, dR, dZ, R = _validate_operator_inputs(psi, R_min, R_max, Z_min, Z_max)
    out = np.empty((NZ, NR))
    check(L.scpn_np_delta_star(psi_arr, NZ, NR, R, float(dR), float(dZ), out))
    return out


def gs_toroidal_current_density_hip(psi, R_min, R_max, Z_min, Z_max, mu0=4e-7 * np.pi):
    """``gs_toroidal_current_density_np`` (:218-234): J_phi = -Delta* psi / (mu0 R) on the interior."""
    if not (np.isfinite(mu0) and mu0 > 0.0):
        raise ValueError("mu0 must be finite and positive")
    psi_arr, NR, NZ, _dR, _dZ, R = _validate_operator_inputs(psi, R_min, R_max, Z_min, Z_max)
    ds = gs_delta_star_hip(psi_arr, R_min, R_max, Z_min, Z_max)
    j = np.zeros_like(psi_arr)
    j[1:NZ - 1, 1:NR - 1] = -ds[1:-1, 1:-1] / (mu0 * R[None, 1:-1])
    return j


def gs_total_toroidal_current_hip(psi, R_min, R_max, Z_min, Z_max, mu0=4e-7 * np.pi):
    """``gs_total_toroidal_current_np`` (:237-248)."""
    _p, _NR, _NZ, dR, dZ, _R = _validate_operator_inputs(psi, R_min, R_max, Z_min, Z_max)
    j = gs_toroidal_current_density_hip(psi, R_min, R_max, Z_min, Z_max, mu0=mu0)
    return float(np.sum(j[1:-1, 1:-1]) * dR * dZ)


def gs_equation_residual_hip(psi, R_min, R_max, Z_min, Z_max, NR, NZ, Ip_target, mu0=4e-7 * np.pi, beta_mix=0.5):
    """``gs_equation_residual_np`` (:251-316): abs_max / relative_max / relative_l2 of the discrete equation imbalance with
    the lane's own source normalisation, evaluated on the device."""
    L = _lib.require_gpu()
    psi_array = np.asarray(psi, dtype=float)
    if psi_array.shape != (NZ, NR):
        raise ValueError(f"psi shape must be (NZ, NR)=({NZ}, {NR}), got {psi_array.shape}")
    if NR < 3 or NZ < 3:
        raise ValueError("Grad-Shafranov residual requires at least a 3 x 3 grid")
    if R_min <= 0.0 or R_max <= R_min or Z_max <= Z_min:
        raise ValueError("Grad-Shafranov residual requires positive, increasing R and Z bounds")
    if not np.all(np.isfinite(psi_array)):
        raise ValueError("psi must contain only finite values")
    out = np.empty(3)
    check(L.scpn_np_equation_residual(np.ascontiguousarray(psi_array, dtype=np.float64), NZ, NR, np.linspace(R_min, R_max, NR),
                                      np.linspace(Z_min, Z_max, NZ), float(Ip_target), float(mu0), float(beta_mix), out))
    return {"abs_max": float(out[0]), "relative_max": float(out[1]), "relative_l2": float(out[2])}

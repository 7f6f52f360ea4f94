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

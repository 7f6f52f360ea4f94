"""Device counterparts of the reference's elliptic operator library (core/multigrid_solve.py).

Same names, argument meaning and error behaviour as the reference free functions; every call
runs the hand-written HIP kernels through the C ABI (``scpn_mg_smooth`` ...).  Leading batch
dimensions are accepted: ``psi`` may be ``(nz, nr)`` or ``(batch, nz, nr)``.

``r_grid`` is the reference's 2-D meshgrid argument; only ``R(ir)`` enters the stencil, so the
1-D column profile is extracted (a 1-D ``r_axis`` is accepted too).  A grid whose rows differ
raises ``ValueError``.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_f64, check

FloatArray = np.ndarray


def validate_sor_omega(omega: float) -> float:
    """core/multigrid_solve.py:33-54."""
    omega_value = float(omega)
    if not np.isfinite(omega_value) or omega_value < 1.0 or omega_value >= 2.0:
        raise ValueError("omega must be finite and satisfy 1.0 <= omega < 2.0")
    return omega_value


def _r_axis(r_grid, nr: int) -> np.ndarray:
    r = np.asarray(r_grid, dtype=np.float64)
    if r.ndim == 1:
        axis = r
    elif r.ndim == 2:
        axis = r[min(1, r.shape[0] - 1)]
        if r.shape[0] > 2 and not np.array_equal(r[1:-1], np.broadcast_to(axis, r[1:-1].shape)):
            raise ValueError("r_grid must be the meshgrid of a 1-D R axis (rows identical)")
    else:
        raise ValueError("r_grid must be 1-D or 2-D")
    if axis.shape[0] != nr:
        raise ValueError(f"r_grid has {axis.shape[0]} columns, expected {nr}")
    return np.ascontiguousarray(axis)


def _batched(a, name="psi"):
    a = as_f64(a)
    if a.ndim == 2:
        return a[None], False
    if a.ndim == 3:
        return a, True
    raise ValueError(f"{name} must have shape (nz, nr) or (batch, nz, nr)")


def mg_smooth(psi, source, r_grid, dr, dz, omega, n_sweeps):
    """Red-black SOR smoother, core/multigrid_solve.py:148-208 (psi is updated in place when it is a
    C-contiguous float64 array, and returned)."""
    omega = validate_sor_omega(omega)
    L = _lib.require_gpu()
    p, _ = _batched(psi)
    s, _ = _batched(source, "source")
    if s.shape != p.shape:
        raise ValueError("source must match psi shape")
    B, nz, nr = p.shape
    work = np.ascontiguousarray(p).copy()
    check(L.scpn_mg_smooth(work, np.ascontiguousarray(s), _r_axis(r_grid, nr), nz, nr, B, float(dr), float(dz), omega,
                           int(n_sweeps)))
    out = work.reshape(np.shape(psi))
    if isinstance(psi, np.ndarray) and psi.dtype == np.float64:
        psi[...] = out
        return psi
    return out


def mg_residual(psi, source, r_grid, dr, dz):
    """core/multigrid_solve.py:211-249: ``L*[psi] - source`` (zero on the boundary)."""
    L = _lib.require_gpu()
    p, _ = _batched(psi)
    s, _ = _batched(source, "source")
    B, nz, nr = p.shape
    out = np.empty_like(p)
    check(L.scpn_mg_residual(np.ascontiguousarray(p), np.ascontiguousarray(s), _r_axis(r_grid, nr), nz, nr, B, float(dr),
                             float(dz), out))
    return out.reshape(np.shape(psi))


def residual_linf(psi, source, r_grid, dr, dz):
    """core/multigrid_solve.py:338-349."""
    L = _lib.require_gpu()
    p, batched = _batched(psi)
    s, _ = _batched(source, "source")
    B, nz, nr = p.shape
    if nz < 3 or nr < 3:
        return np.zeros(B) if batched else 0.0
    out = np.empty(B)
    check(L.scpn_residual_linf(np.ascontiguousarray(p), np.ascontiguousarray(s), _r_axis(r_grid, nr), nz, nr, B, float(dr),
                               float(dz), out))
    return out if batched else float(out[0])


def gs_residual_rms(psi, source, r_grid, dr, dz):
    """core/fusion_kernel_solver_runtime.py:45-51 compute_gs_residual_rms."""
    L = _lib.require_gpu()
    p, batched = _batched(psi)
    s, _ = _batched(source, "source")
    B, nz, nr = p.shape
    out = np.empty(B)
    check(L.scpn_gs_residual_rms(np.ascontiguousarray(p), np.ascontiguousarray(s), _r_axis(r_grid, nr), nz, nr, B, float(dr),
                                 float(dz), out))
    return out if batched else float(out[0])


def restrict_full_weight(fine):
    """core/multigrid_solve.py:57-99."""
    L = _lib.require_gpu()
    f, _ = _batched(fine, "fine")
    B, nz, nr = f.shape
    out = np.empty((B, (nz + 1) // 2, (nr + 1) // 2))
    check(L.scpn_restrict_full_weight(np.ascontiguousarray(f), nz, nr, B, out))
    return out.reshape(np.shape(fine)[:-2] + out.shape[1:])


def prolongate_bilinear(coarse, nz_f, nr_f):
    """core/multigrid_solve.py:102-145 (coarse shape must be ((nz_f+1)//2, (nr_f+1)//2))."""
    L = _lib.require_gpu()
    c, _ = _batched(coarse, "coarse")
    B, nz_c, nr_c = c.shape
    out = np.empty((B, int(nz_f), int(nr_f)))
    check(L.scpn_prolongate_bilinear(np.ascontiguousarray(c), nz_c, nr_c, int(nz_f), int(nr_f), B, out))
    return out.reshape(np.shape(coarse)[:-2] + out.shape[1:])


def multigrid_vcycle(psi, source, r_grid, dr, dz, *, omega=1.0, pre_smooth=3, post_smooth=3, min_grid=5):
    """One V-cycle, core/multigrid_solve.py:252-335 (returns a new array)."""
    omega = validate_sor_omega(omega)
    L = _lib.require_gpu()
    p, _ = _batched(psi)
    s, _ = _batched(source, "source")
    B, nz, nr = p.shape
    work = np.ascontiguousarray(p).copy()
    check(L.scpn_multigrid_vcycle(work, np.ascontiguousarray(s), _r_axis(r_grid, nr), nz, nr, B, float(dr), float(dz), omega,
                                  int(pre_smooth), int(post_smooth), int(min_grid)))
    return work.reshape(np.shape(psi))


def multigrid_solve(source, psi_bc, r_min, r_max, z_min, z_max, nr, nz, *, tol=1e-6, max_cycles=500, omega=1.0,
                    pre_smooth=3, post_smooth=3, min_grid=5):
    """Full geometric-multigrid solve, core/multigrid_solve.py:352-463.

    Returns ``(psi, residual, n_cycles, converged)``; with a leading batch dimension the last three
    are arrays.
    """
    source_arr = np.asarray(source, dtype=np.float64)
    psi = np.asarray(psi_bc, dtype=np.float64).copy()
    if source_arr.shape[-2:] != (nz, nr) or psi.shape != source_arr.shape or psi.ndim not in (2, 3):
        raise ValueError(
            f"source and psi_bc must have shape (nz, nr) = ({nz}, {nr}); got source={source_arr.shape}, psi_bc={psi.shape}.")
    if not (np.isfinite(tol) and tol > 0.0):
        raise ValueError("tol must be finite and > 0.")
    if max_cycles < 1:
        raise ValueError("max_cycles must be >= 1.")
    validate_sor_omega(omega)
    return multigrid_solve_on_axes(source_arr, psi, np.linspace(r_min, r_max, nr), np.linspace(z_min, z_max, nz), tol=tol,
                                   max_cycles=max_cycles, omega=omega, pre_smooth=pre_smooth, post_smooth=post_smooth, min_grid=min_grid)


def multigrid_solve_on_axes(source, psi_bc, r_axis, z_axis, *, tol=1e-6, max_cycles=500, omega=1.0, pre_smooth=3, post_smooth=3,
                            min_grid=5):
    """``multigrid_solve`` for callers that hold the 1-D axes themselves (an external code's grid): R at every column is
    ``r_axis[i]``, the spacings are ``r_axis[1] - r_axis[0]`` / ``z_axis[1] - z_axis[0]`` (core/multigrid_solve.py:429-432)."""
    source_arr, psi = as_f64(source), np.array(psi_bc, dtype=np.float64)
    r_axis, z_axis = np.ascontiguousarray(r_axis, dtype=np.float64), np.ascontiguousarray(z_axis, dtype=np.float64)
    nr, nz = r_axis.size, z_axis.size
    if source_arr.shape[-2:] != (nz, nr) or psi.shape != source_arr.shape or psi.ndim not in (2, 3):
        raise ValueError(f"source and psi_bc must have shape (nz, nr) = ({nz}, {nr}); got source={source_arr.shape}, psi_bc={psi.shape}.")
    if not (np.isfinite(tol) and tol > 0.0):
        raise ValueError("tol must be finite and > 0.")
    if max_cycles < 1:
        raise ValueError("max_cycles must be >= 1.")
    validate_sor_omega(omega)
    L = _lib.require_gpu()
    batched = psi.ndim == 3
    B = psi.shape[0] if batched else 1
    psi = np.ascontiguousarray(psi)
    res = np.empty(B)
    ncy = np.empty(B, dtype=np.int32)
    conv = np.empty(B, dtype=np.int32)
    check(L.scpn_multigrid_solve(np.ascontiguousarray(source_arr), psi, r_axis, z_axis, int(nr), int(nz), B, float(tol),
                                 int(max_cycles), float(omega), int(pre_smooth), int(post_smooth), int(min_grid), res, ncy,
                                 conv))
    if batched:
        return psi, res, ncy.astype(int), conv.astype(bool)
    return psi, float(res[0]), int(ncy[0]), bool(conv[0])


def jacobi_step(psi, source, r_grid, dr, dz):
    """FusionKernel._jacobi_step, core/fusion_kernel_iterative_solver.py:54-95."""
    L = _lib.require_gpu()
    p, _ = _batched(psi)
    s, _ = _batched(source, "source")
    B, nz, nr = p.shape
    out = np.empty_like(p)
    check(L.scpn_jacobi_step(np.ascontiguousarray(p), np.ascontiguousarray(s), _r_axis(r_grid, nr), nz, nr, B, float(dr),
                             float(dz), out))
    return out.reshape(np.shape(psi))


def sor_step(psi, source, r_grid, dr, dz, omega=1.6):
    """FusionKernel._sor_step, core/fusion_kernel_iterative_solver.py:97-161."""
    omega = validate_sor_omega(omega)
    L = _lib.require_gpu()
    p, _ = _batched(psi)
    s, _ = _batched(source, "source")
    B, nz, nr = p.shape
    out = np.empty_like(p)
    check(L.scpn_sor_step(np.ascontiguousarray(p), np.ascontiguousarray(s), _r_axis(r_grid, nr), nz, nr, B, float(dr), float(dz),
                          omega, out))
    return out.reshape(np.shape(psi))


__all__ = ["validate_sor_omega", "mg_smooth", "mg_residual", "residual_linf", "gs_residual_rms", "restrict_full_weight",
           "prolongate_bilinear", "multigrid_vcycle", "multigrid_solve", "multigrid_solve_on_axes", "jacobi_step", "sor_step"]

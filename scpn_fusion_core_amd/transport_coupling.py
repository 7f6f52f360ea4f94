"""Batched transport-coupling kernels: the pieces of the GS <-> transport loop that touch the
2-D flux map or solve a linear system, so an ensemble's coupled outer iteration needs no host round trip of fields.

Mirrors (reference file:line):
  * ``TransportSolverCouplingMixin.map_profiles_to_2d``  core/integrated_transport_solver_coupling.py:39-73
  * the GS step of ``run_self_consistent``                :129-147 (map -> solve_equilibrium -> relative psi change)
  * ``scipy.linalg.solve_banded((1, 1), ab, b)`` as used by core/current_diffusion.py:347-352 (LAPACK dgtsv)
  * ``CurrentDiffusionSolver``                            core/current_diffusion.py:262-353
  * ``solve_flux_evolution_nonadiabatic``                 core/current_diffusion.py:55-172 (benchmarks/bench_current_diffusion_nonadiabatic.py)

The 1-D transport physics itself (chi models, sources, bootstrap-current formulae) stays with the caller: it hands over
``ne*(Ti+Te)`` and the bootstrap profile, exactly the two arrays ``map_profiles_to_2d`` interpolates.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import as_f64, check, ptr
from .ensemble import EnsembleSolver

MU_0 = 4.0 * np.pi * 1e-7  # core/current_diffusion.py:20


def map_profiles_to_2d(es: EnsembleSolver, rho, pressure_1d, j_bs_1d, *, return_pressure: bool = False):
    """Project 1-D profiles of every member onto its resident flux map and leave ``J_phi`` on the device.

    ``pressure_1d`` (= ``ne * (Ti + Te)``) and ``j_bs_1d`` are ``(batch, n_rho)`` (or ``(n_rho,)`` for all members).
    Returns ``(I_current, Pressure_2D or None)``; ``es.get_jphi()`` reads the mapped current density.
    """
    rho = np.ascontiguousarray(as_f64(rho).reshape(-1))
    n = rho.size

    def per_member(a):
        a = as_f64(a)
        if a.ndim == 1:
            a = np.broadcast_to(a, (es.batch, n))
        if a.shape != (es.batch, n):
            raise ValueError(f"profile arrays must have shape ({es.batch}, {n}) or ({n},)")
        if not np.all(np.isfinite(a)):
            raise ValueError("profiles must contain finite values only.")
        return np.ascontiguousarray(a)

    if n < 2 or not np.all(np.diff(rho) > 0.0):
        raise ValueError("rho must be strictly increasing with at least two points")
    p1, jb = per_member(pressure_1d), per_member(j_bs_1d)
    p2 = np.empty((es.batch, es.NZ, es.NR)) if return_pressure else None
    ic = np.empty(es.batch)
    check(es._L.scpn_gs_map_profiles(es._h, float(es.cfg["dimensions"]["Z_min"]), n, rho, p1, jb, ptr(p2), ptr(ic)))
    return ic, p2


def gs_transport_outer_step(es: EnsembleSolver, rho, pressure_1d, j_bs_1d) -> tuple[np.ndarray, list[dict]]:
    """One GS leg of ``run_self_consistent`` (:129-147) for every member: map the profiles, re-solve the equilibrium in
    ``external_profile_mode`` (the transport solver's setting, ``_integrated_transport_solver_init.py:28``) and return the
    relative flux change ``||psi - psi_old|| / ||psi_old||`` per member with the solver results."""
    # psi_old and both norms stay on the device (scpn_gs_snapshot_psi / scpn_gs_flux_change): downloading the fields of a
    # 32-member 513^2 ensemble twice and taking three host norms cost 52 of the 86 ms of a coupled step
    es.snapshot_psi()
    map_profiles_to_2d(es, rho, pressure_1d, j_bs_1d)
    prev = es.external_profile_mode
    es.external_profile_mode = True
    try:
        res = es.solve_equilibrium()
    finally:
        es.external_profile_mode = prev
    return es.flux_change(), res


def solve_banded_tridiagonal(dl, d, du, b):
    """``scipy.linalg.solve_banded((1, 1), ab, b)`` for a batch of systems on the device (LAPACK dgtsv order, bit-identical).
    ``d``/``b``: (batch, n) or (n,); ``dl`` (sub-diagonal) / ``du`` (super-diagonal): (..., n-1).  Raises
    ``numpy.linalg.LinAlgError`` like SciPy when a pivot is exactly zero."""
    L = _lib.require_gpu()
    d_, b_ = as_f64(d), as_f64(b)
    single = d_.ndim == 1
    d2, b2 = np.atleast_2d(d_), np.atleast_2d(b_)
    batch, n = d2.shape
    dl2 = np.atleast_2d(as_f64(dl)) if n > 1 else None
    du2 = np.atleast_2d(as_f64(du)) if n > 1 else None
    if b2.shape != (batch, n) or (n > 1 and (dl2.shape != (batch, n - 1) or du2.shape != (batch, n - 1))):
        raise ValueError("inconsistent tridiagonal system shapes")
    x = np.empty((batch, n))
    info = np.zeros(batch, dtype=np.int32)
    check(L.scpn_tridiag_solve(batch, n, ptr(None if dl2 is None else np.ascontiguousarray(dl2)), np.ascontiguousarray(d2),
                               ptr(None if du2 is None else np.ascontiguousarray(du2)), np.ascontiguousarray(b2), x, ptr(info)))
    if np.any(info != 0):
        raise np.linalg.LinAlgError(f"singular matrix: resolution failed at diagonal {int(info[info != 0][0]) - 1}")
    return x[0] if single else x


@dataclass(frozen=True)
class FluxEvolutionTrajectory:
    """Time history of the non-adiabatic flux carrier (fields of core/current_diffusion.py:29-53); arrays carry a leading batch
    axis only when the solve was batched."""

    time_s: np.ndarray
    rho: np.ndarray
    psi: np.ndarray
    hall_drive: np.ndarray
    resistive_loss: np.ndarray
    source: np.ndarray
    damping_rate: np.ndarray
    source_increment: np.ndarray
    damping_decrement: np.ndarray
    update_residual: np.ndarray
    dt_s: float


def _as_profile(values, n: int, name: str, *, allow_infinite: bool = False) -> np.ndarray:
    prof = np.asarray(values, dtype=np.float64)
    if prof.ndim == 0:
        prof = np.full(n, float(prof), dtype=np.float64)
    if prof.shape != (n,):
        raise ValueError(f"{name} must be scalar or shape ({n},)")
    ok = np.isfinite(prof) | np.isposinf(prof) if allow_infinite else np.isfinite(prof)
    if not np.all(ok):
        raise ValueError(f"{name} must contain " + ("finite values or np.inf" if allow_infinite else "finite values"))
    return prof


def flux_evolution_batched(psi0, damping_rate, source, dt: float):
    """The recurrence psi[k+1] = psi[k] - psi[k](1 - e^{-gamma dt}) + s (1 - e^{-gamma dt}) / gamma  (dt*s where gamma = 0) of
    core/current_diffusion.py:141-158 for a batch of profiles on the device: ``psi0`` (batch, n), ``damping_rate`` / ``source``
    (batch, n_steps+1, n) sampled at the step end points.  Returns (psi, source_increment, damping_decrement, update_residual)."""
    L = _lib.require_gpu()
    rate, src, p0 = np.ascontiguousarray(as_f64(damping_rate)), np.ascontiguousarray(as_f64(source)), np.ascontiguousarray(as_f64(psi0))
    if rate.ndim != 3 or src.shape != rate.shape or p0.shape != (rate.shape[0], rate.shape[2]) or rate.shape[1] < 2:
        raise ValueError("expected psi0 (batch, n) and damping_rate / source (batch, n_steps+1, n)")
    if not np.isfinite(dt) or dt <= 0.0:
        raise ValueError("dt must be positive")
    batch, n1, n = rate.shape
    psi = np.empty((batch, n1, n))
    inc, drop, resid = np.empty((batch, n1 - 1, n)), np.empty((batch, n1 - 1, n)), np.empty((batch, n1 - 1, n))
    check(L.scpn_flux_evolution(batch, n, n1 - 1, float(dt), rate, src, p0, psi, ptr(inc), ptr(drop), ptr(resid)))
    return psi, inc, drop, resid


def solve_flux_evolution_nonadiabatic(rho, psi0, *, tau_psi_fn, R_null_t, E_theta_t, eta_spitzer_fn, J_theta_t, dt: float,
                                      n_steps: int) -> FluxEvolutionTrajectory:
    """core/current_diffusion.py:55-172 with the same call-back contract: the (Python) drive functions are sampled on the host
    at the n_steps+1 time points, the time recurrence runs on the device."""
    rho_grid = np.asarray(rho, dtype=np.float64)
    if rho_grid.ndim != 1:
        raise ValueError("rho must be one-dimensional")
    if rho_grid.size < 2:
        raise ValueError("rho must contain at least two points")
    if not np.all(np.isfinite(rho_grid)):
        raise ValueError("rho must contain finite values")
    if not np.all(np.diff(rho_grid) > 0.0):
        raise ValueError("rho must be strictly increasing")
    n = rho_grid.size
    psi_initial = _as_profile(psi0, n, "psi0")
    if not np.isfinite(dt) or dt <= 0.0:
        raise ValueError("dt must be positive")
    if n_steps <= 0:
        raise ValueError("n_steps must be positive")
    time_s = np.linspace(0.0, dt * n_steps, n_steps + 1, dtype=np.float64)
    hall, loss, rate = np.zeros((n_steps + 1, n)), np.zeros((n_steps + 1, n)), np.zeros((n_steps + 1, n))
    for k, t in enumerate(time_s):
        try:
            tau_raw = tau_psi_fn(t, rho_grid)
        except TypeError:
            tau_raw = tau_psi_fn(t)
        tau = _as_profile(tau_raw, n, "tau", allow_infinite=True)
        if np.any(np.isneginf(tau) | ((tau <= 0.0) & ~np.isposinf(tau))):
            raise ValueError("tau_psi_fn must return positive finite values or np.inf")
        with np.errstate(divide="ignore"):
            rate[k] = np.where(np.isinf(tau), 0.0, 1.0 / tau)
        r_null = float(R_null_t(float(t)))
        if not np.isfinite(r_null) or r_null < 0.0:
            raise ValueError("R_null_t must return a finite non-negative radius")
        e_theta = _as_profile(E_theta_t(float(t), rho_grid), n, "E_theta_t")
        eta = _as_profile(eta_spitzer_fn(rho_grid), n, "eta_spitzer_fn")
        if np.any(eta < 0.0):
            raise ValueError("eta_spitzer_fn must return non-negative resistivity")
        j_theta = _as_profile(J_theta_t(float(t), rho_grid), n, "J_theta_t")
        hall[k] = r_null * e_theta
        loss[k] = eta * j_theta
    source = hall - loss
    psi, inc, drop, resid = flux_evolution_batched(psi_initial[None], rate[None], source[None], dt)
    return FluxEvolutionTrajectory(time_s=time_s, rho=rho_grid, psi=psi[0], hall_drive=hall, resistive_loss=loss, source=source,
                                   damping_rate=rate, source_increment=inc[0], damping_decrement=drop[0], update_residual=resid[0],
                                   dt_s=float(dt))


class CurrentDiffusionSolver:
    """Batched implicit Crank-Nicolson poloidal-flux diffusion (core/current_diffusion.py:262-353): ``batch`` independent
    profiles on one ``rho`` grid advance per ``step`` in one device launch."""

    def __init__(self, rho, R0: float, a: float, B0: float, batch: int = 1):
        self.rho = np.ascontiguousarray(as_f64(rho).reshape(-1))
        self.R0, self.a, self.B0 = float(R0), float(a), float(B0)
        self.nr = len(self.rho)
        self.batch = int(batch)
        self.drho = self.rho[1] - self.rho[0]
        psi = np.zeros(self.nr)
        for i in range(1, self.nr):  # :268-275 initial q = 1 + 2 rho^2 flux profile
            r = self.rho[i]
            q_r = 1.0 + 2.0 * r**2
            dpsi = -r * self.a**2 * self.B0 / (self.R0 * q_r)
            psi[i] = psi[i - 1] + dpsi * self.drho
        psi -= psi[-1]
        self.psi = np.tile(psi, (self.batch, 1))

    def step(self, dt: float, Te, ne, Z_eff: float, j_bs, j_cd, j_ext=None) -> np.ndarray:
        L = _lib.require_gpu()

        def per_member(arr):
            arr = as_f64(arr)
            if arr.ndim == 1:
                arr = np.broadcast_to(arr, (self.batch, self.nr))
            if arr.shape != (self.batch, self.nr):
                raise ValueError(f"profiles must have shape ({self.batch}, {self.nr}) or ({self.nr},)")
            return np.ascontiguousarray(arr)

        te, ne_, jb, jc = per_member(Te), per_member(ne), per_member(j_bs), per_member(j_cd)
        je = None if j_ext is None else per_member(j_ext)
        psi = np.ascontiguousarray(self.psi)
        info = np.zeros(self.batch, dtype=np.int32)
        check(L.scpn_current_diffusion_step(self.batch, self.nr, self.rho, psi, self.R0, self.a, self.B0, float(dt), te, ne_,
                                            float(Z_eff), jb, jc, ptr(je), ptr(info)))
        if np.any(info != 0):
            raise np.linalg.LinAlgError("singular Crank-Nicolson system in current diffusion step")
        self.psi = psi
        return self.psi

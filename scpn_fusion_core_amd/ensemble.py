"""Resident ensemble solver: ``batch`` independent Grad-Shafranov equilibria on one MI355X.

Thin Python face of the ``scpn_gs_*`` handle API (include/scpn_hip.h, group 3).  Fields are uploaded
once, the whole Picard / free-boundary iteration runs on the device (per-member scalars, convergence
flags and ping-pong state never leave HBM) and results are downloaded once.  The reference has no
NumPy ensemble driver; the per-member semantics are exactly those of
``FusionKernel.solve_equilibrium`` / ``solve_free_boundary``
(core/fusion_kernel_newton_solver.py:390-615, core/fusion_kernel_free_boundary.py:623-739).
"""

from __future__ import annotations

import ctypes as C
import logging
import os
import warnings
from typing import Any, Sequence

import numpy as np

from . import _lib
from ._lib import NewtonOpts, NewtonResult, PicardParams, PicardResult, as_f64, check, ptr

_METHODS = {"multigrid": 0, "sor": 1, "jacobi": 2, "anderson": 3, "newton": 1}  # newton: its warm-up is the SOR branch
_ARITHMETIC_NAMES = ("exact", "contracted")  # scpn_gs_set_arith modes 0 / 1
_PED_DEFAULT = {"ped_top": 0.92, "ped_width": 0.05, "ped_height": 1.0, "core_alpha": 0.3}  # core/fusion_kernel.py:181-192


def picard_params_from_cfg(cfg: dict, *, preserve_initial_state=False, has_boundary_flux=False,
                           external_profile_mode=False, check_every=8) -> PicardParams:
    """Translate a reference-style config dict (core/config_schema.py) into the C parameter block."""
    dims, phys, sol = cfg["dimensions"], cfg["physics"], cfg.get("solver", {})
    method = sol.get("solver_method", "multigrid")
    if method not in _METHODS:
        raise NotImplementedError(f"solver_method={method!r} is not on the HIP path (multigrid, sor, anderson, jacobi are)")
    p = PicardParams()
    p.z_min_cfg = dims["Z_min"]
    p.r_center = (dims["R_min"] + dims["R_max"]) / 2.0
    p.mu0 = phys["vacuum_permeability"]
    p.method = _METHODS[method]
    p.sor_omega = sol.get("sor_omega", 1.6)
    p.alpha = sol.get("relaxation_factor", 0.1)
    p.tol = sol["convergence_threshold"]
    p.max_iter = sol["max_iterations"]
    p.require_gs_residual = int(bool(sol.get("require_gs_residual", False)))
    p.gs_tol = float(sol.get("gs_residual_threshold", p.tol))
    if p.require_gs_residual and p.gs_tol <= 0.0:
        raise ValueError("solver.gs_residual_threshold must be > 0")
    prof = phys.get("profiles") or {}
    p.hmode = int(prof.get("mode", "l-mode") in ("h-mode", "H-mode", "hmode"))
    pp, pf = dict(_PED_DEFAULT), dict(_PED_DEFAULT)
    pp.update(prof.get("p_prime", {}))
    pf.update(prof.get("ff_prime", {}))
    keys = ("ped_top", "ped_width", "ped_height", "core_alpha")
    p.ped_p = (C.c_double * 4)(*[pp[k] for k in keys])
    p.ped_ff = (C.c_double * 4)(*[pf[k] for k in keys])
    p.preserve_initial_state = int(preserve_initial_state)
    p.has_boundary_flux = int(has_boundary_flux)
    p.external_profile_mode = int(external_profile_mode)
    p.check_every = int(check_every)
    p.fail_on_diverge = int(bool(sol.get("fail_on_diverge", False)))
    p.skip_seed = 0
    p.anderson_m = int(sol.get("anderson_depth", 5))
    p.xpoint_saddle = int(bool(sol.get("xpoint_use_saddle_detection", False)))
    return p


def newton_opts_from_cfg(cfg: dict, *, poll_every: int = 10) -> NewtonOpts:
    """core/fusion_kernel_newton_runtime.py:75-150 parse_newton_dispatch_config (line-search / GMRES keys)."""
    sol = cfg["solver"]
    c = float(sol.get("newton_line_search_c", 1e-4))
    if (not np.isfinite(c)) or c <= 0.0 or c >= 1.0:
        raise ValueError("solver.newton_line_search_c must be finite and in (0, 1)")
    mb = int(sol.get("newton_line_search_max_backtracks", 6))
    if mb <= 0:
        raise ValueError("solver.newton_line_search_max_backtracks must be >= 1")
    mode = str(sol.get("gmres_preconditioner", "none")).strip().lower()
    if mode in {"", "auto", "default", "legacy"}:
        mode = "none"
    if mode == "none" and bool(sol.get("gmres_diagonal_preconditioner", False)):
        mode = "diagonal"
    if mode not in {"none", "diagonal", "ilu", "multigrid"}:
        raise ValueError("solver.gmres_preconditioner must be one of {'none', 'diagonal', 'ilu', 'multigrid'}")
    if mode == "ilu":
        # the reference degrades to the diagonal preconditioner when SuperLU's spilu is unavailable
        # (core/fusion_kernel_solver_runtime.py:289-293); there is no incomplete LU on the device, so this is that path
        logging.getLogger(__name__).warning("ILU preconditioner unavailable on the HIP path (%s); falling back to diagonal.",
                                            "no device spilu")
        mode = "diagonal"
    drop, fill = float(sol.get("gmres_ilu_drop_tol", 1e-4)), float(sol.get("gmres_ilu_fill_factor", 8.0))
    if (not np.isfinite(drop)) or drop <= 0.0:
        raise ValueError("solver.gmres_ilu_drop_tol must be finite and > 0")
    if (not np.isfinite(fill)) or fill < 1.0:
        raise ValueError("solver.gmres_ilu_fill_factor must be finite and >= 1")
    budget = sol.get("gmres_nonconverged_budget")
    if budget is not None and (isinstance(budget, bool) or int(budget) < 0):
        raise ValueError("solver.gmres_nonconverged_budget must be an int >= 0")
    o = NewtonOpts()
    o.precond = {"none": 0, "diagonal": 1, "multigrid": 2}[mode]  # 2: one V-cycle of the GS* operator (extension of this package)
    o.line_search = int(bool(sol.get("newton_line_search", False)))
    o.line_search_c = c
    o.max_backtracks = mb
    # the reference always warms up with min(15, max_iter // 2) Picard iterations (fusion_kernel_newton_runtime.py:92) = -1 here;
    # ``solver.newton_warmup_steps`` overrides it (the finite-difference / multigrid lane needs far fewer)
    o.warmup_steps = int(sol.get("newton_warmup_steps", -1))
    if o.warmup_steps < -1:
        raise ValueError("solver.newton_warmup_steps must be >= 0 (or -1 for the reference's rule)")
    # the reference applies psi += 0.5 * delta (core/fusion_kernel_newton_solver.py:297); ``solver.newton_damping`` is this
    # package's opt-in for the full Newton step, useful once the linear solves are accurate (multigrid-preconditioned GMRES)
    damping = float(sol.get("newton_damping", 0.5))
    if (not np.isfinite(damping)) or damping <= 0.0 or damping > 1.0:
        raise ValueError("solver.newton_damping must be finite and in (0, 1]")
    o.newton_alpha = damping
    o.gmres_restart, o.gmres_maxiter, o.gmres_atol, o.gmres_rtol = 50, 100, 1e-8, 1e-6  # the reference's call (solver_runtime.py:296-312)
    if mode == "multigrid":
        # inexact Newton: a V-cycle-preconditioned solve gains about a decade per column, and a forcing term of 1e-3 keeps the
        # outer iteration superlinear at a third of the columns of rtol 1e-6 (measured: tools/newton_fb_sweep.py); a step whose
        # Jacobian action is at its finite-difference noise floor must not spin through 100 x 50 columns, so the Krylov space
        # is capped.  All four are overridable (solver.gmres_restart / gmres_maxiter / gmres_rtol / gmres_atol).
        o.gmres_restart, o.gmres_maxiter = int(sol.get("gmres_restart", 8)), int(sol.get("gmres_maxiter", 1))
        o.gmres_rtol = 1e-3
        if o.gmres_restart < 1 or o.gmres_maxiter < 1:
            raise ValueError("solver.gmres_restart and solver.gmres_maxiter must be >= 1")
        o.gmres_rtol, o.gmres_atol = float(sol.get("gmres_rtol", o.gmres_rtol)), float(sol.get("gmres_atol", o.gmres_atol))
        if not (np.isfinite(o.gmres_rtol) and o.gmres_rtol > 0.0 and np.isfinite(o.gmres_atol) and o.gmres_atol >= 0.0):
            raise ValueError("solver.gmres_rtol must be finite and > 0, solver.gmres_atol finite and >= 0")
    o.poll_every = int(poll_every)
    jac = str(sol.get("newton_jacobian", "profile")).strip().lower()
    if jac not in {"profile", "finite_difference"}:
        raise ValueError("solver.newton_jacobian must be 'profile' (the reference's) or 'finite_difference'")
    o.reserved = 1 if jac == "finite_difference" else 0
    return o


def _result_dict(r: PicardResult) -> dict[str, Any]:
    return {"converged": bool(r.converged), "iterations": int(r.iterations), "diverged": bool(r.diverged),
            "residual": r.residual, "gs_residual": r.gs_residual, "gs_residual_best": r.gs_residual_best,
            "psi_axis": r.psi_axis, "psi_boundary": r.psi_bnd, "x_point": (r.x_r, r.x_z), "i_current": r.i_current}


_live_streams = 0
_warned_queues = False


def _note_live_handle(sign: int, groups: int) -> None:
    """Count the solver streams alive in this process and say so -- once -- when they outnumber the hardware queues HIP will use.

    Every stream group of a handle is its own HIP stream; the runtime multiplexes streams onto ``GPU_MAX_HW_QUEUES`` hardware queues
    (default 4), and two group streams that share a queue run one after the other (measured: 37.5 -> 50.7 ms per Newton-lane
    solve with two handles alive).  The library never touches the process environment, so an embedding
    application that keeps several handles alive has to export ``GPU_MAX_HW_QUEUES`` itself before its first HIP call."""
    global _live_streams, _warned_queues
    _live_streams += sign * max(int(groups), 1)
    if sign > 0 and not _warned_queues and "GPU_MAX_HW_QUEUES" not in os.environ and _live_streams > 3:
        _warned_queues = True
        warnings.warn(f"{_live_streams} solver streams are alive but GPU_MAX_HW_QUEUES is unset (HIP default: 4 hardware queues, one of "
                      "them the null stream's): stream groups will share queues and serialise; export GPU_MAX_HW_QUEUES=8 before the "
                      "first HIP call of the process", RuntimeWarning, stacklevel=3)


class EnsembleSolver:
    """``batch`` equilibria sharing one (R, Z) grid and one solver configuration.

    Parameters
    ----------
    cfg : dict
        Reference-style configuration (grid_resolution, dimensions, physics, coils, solver).
    batch : int
        Number of ensemble members resident on this GPU.
    coil_currents : array (batch, n_coils), optional
        Per-member currents of ``cfg["coils"]`` (default: the config's currents for every member).
    current_targets : array (batch,), optional
        Per-member ``plasma_current_target``.
    coil_r, coil_z : array (batch, n_coils), optional
        Per-member positions of ``cfg["coils"]`` (a shape scan: every member its own coil geometry on the shared grid); they
        feed the vacuum field of the seed and are the default coil set of ``solve_free_boundary``.
    arithmetic : ``"exact"`` | ``"contracted"``, optional
        Arithmetic of the handle's hot kernels (``scpn_gs_set_arith``, include/scpn_hip.h): ``"exact"`` keeps the reference's
        operation order (bit-exact linear operators), ``"contracted"`` runs the same iteration with fused multiply-adds
        (agreement to rounding, <= 1e-10 max|psi| on converged equilibria).  Default: ``cfg["solver"]["arithmetic"]`` if present,
        else the library's default (environment variable ``SCPN_HIP_ARITH``, else exact).
    """

    def __init__(self, cfg: dict, batch: int = 1, coil_currents=None, current_targets=None, device: int | None = None,
                 r_axis=None, z_axis=None, coil_r=None, coil_z=None, arithmetic: str | None = None):
        self._L = _lib.require_gpu()
        if device is not None:
            check(self._L.scpn_set_device(int(device)))
        self.cfg = cfg
        self.batch = int(batch)
        self.NR, self.NZ = (int(v) for v in cfg["grid_resolution"])
        d = cfg["dimensions"]
        self.R = np.linspace(d["R_min"], d["R_max"], self.NR)  # core/fusion_kernel.py:165-168
        self.Z = np.linspace(d["Z_min"], d["Z_max"], self.NZ)
        if r_axis is not None:  # caller-supplied axes (sampling a host array whose axes are already materialised)
            self.R = np.ascontiguousarray(as_f64(r_axis).reshape(self.NR))
        if z_axis is not None:
            self.Z = np.ascontiguousarray(as_f64(z_axis).reshape(self.NZ))
        self.dR = float(self.R[1] - self.R[0])
        self.dZ = float(self.Z[1] - self.Z[0])
        h = C.c_void_p()
        check(self._L.scpn_gs_create(C.byref(h), self.NR, self.NZ, self.batch, self.R, self.Z, self.dR, self.dZ))
        self._h = h
        _note_live_handle(+1, self._L.scpn_gs_num_groups(h))
        mode = arithmetic if arithmetic is not None else cfg.get("solver", {}).get("arithmetic")
        if mode is not None:
            self.arithmetic = mode
        coils = cfg.get("coils", [])
        self.n_coils = len(coils)
        self.coil_r = np.tile(np.array([c["r"] for c in coils], dtype=np.float64), (self.batch, 1)) if coil_r is None else as_f64(coil_r).reshape(self.batch, -1)
        self.coil_z = np.tile(np.array([c["z"] for c in coils], dtype=np.float64), (self.batch, 1)) if coil_z is None else as_f64(coil_z).reshape(self.batch, -1)
        if self.coil_r.shape != (self.batch, self.n_coils) or self.coil_z.shape != (self.batch, self.n_coils):
            raise ValueError("coil_r / coil_z must have shape (batch, n_coils)")
        base_i = np.array([c["current"] for c in coils], dtype=np.float64)
        self.coil_i = np.tile(base_i, (self.batch, 1)) if coil_currents is None else as_f64(coil_currents).reshape(self.batch, -1)
        if self.coil_i.shape != (self.batch, self.n_coils):
            raise ValueError("coil_currents must have shape (batch, n_coils)")
        self._push_vacuum_coils()
        it = cfg["physics"]["plasma_current_target"]
        self.current_targets = np.full(self.batch, float(it)) if current_targets is None else as_f64(current_targets).reshape(-1)
        if self.current_targets.shape != (self.batch,):
            raise ValueError("current_targets must have shape (batch,)")
        check(self._L.scpn_gs_set_current_target(self._h, -1, self.current_targets))
        self.external_profile_mode = False

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            _note_live_handle(-1, self._L.scpn_gs_num_groups(self._h))
            self._L.scpn_gs_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- state transfer -------------------------------------------------------------------------
    def _push_vacuum_coils(self):
        if self.n_coils:
            self.coil_r, self.coil_z, self.coil_i = (np.ascontiguousarray(a) for a in (self.coil_r, self.coil_z, self.coil_i))
            check(self._L.scpn_gs_set_vacuum_coils(self._h, self.n_coils, ptr(self.coil_r), ptr(self.coil_z), ptr(self.coil_i)))
        else:
            check(self._L.scpn_gs_set_vacuum_coils(self._h, 0, None, None, None))

    def _field(self, arr, member):
        a = as_f64(arr)
        want = (self.NZ, self.NR) if member >= 0 else (self.batch, self.NZ, self.NR)
        if a.shape != want:
            raise ValueError(f"field shape {a.shape} must be {want}")
        return np.ascontiguousarray(a)

    def _get(self, fn, member):
        out = np.empty((self.NZ, self.NR) if member >= 0 else (self.batch, self.NZ, self.NR))
        check(fn(self._h, member, out))
        return out

    def set_psi(self, psi, member=-1):
        check(self._L.scpn_gs_set_psi(self._h, member, self._field(psi, member)))

    def get_psi(self, member=-1):
        return self._get(self._L.scpn_gs_get_psi, member)

    def get_jphi(self, member=-1):
        return self._get(self._L.scpn_gs_get_jphi, member)

    def set_jphi(self, j, member=-1):
        check(self._L.scpn_gs_set_jphi(self._h, member, self._field(j, member)))

    def get_source(self, member=-1):
        return self._get(self._L.scpn_gs_get_source, member)

    def set_source(self, s, member=-1):
        check(self._L.scpn_gs_set_source(self._h, member, self._field(s, member)))

    def set_boundary_flux(self, bc, member=-1):
        check(self._L.scpn_gs_set_boundary_flux(self._h, member, self._field(bc, member)))

    def get_boundary_flux(self, member=-1):
        return self._get(self._L.scpn_gs_get_boundary_flux, member)

    def set_seed(self, seed_j):
        """Optional host-computed Gaussian seed (core/fusion_kernel_iterative_solver.py:400-402)."""
        if seed_j is None:
            check(self._L.scpn_gs_set_seed(self._h, None))
        else:
            s = self._field(seed_j, 0)
            check(self._L.scpn_gs_set_seed(self._h, ptr(s)))

    # -- solves ---------------------------------------------------------------------------------
    def solve_equilibrium(self, preserve_initial_state=False, boundary_flux=None, *, history=False, trace=False,
                          check_every=8) -> list[dict[str, Any]]:
        """Picard solve of every member (core/fusion_kernel_newton_solver.py:390-615); ``solver_method="newton"`` dispatches
        to :meth:`solve_newton` as the reference does (:435-439)."""
        if self.cfg.get("solver", {}).get("solver_method", "multigrid") == "newton":
            return self.solve_newton(preserve_initial_state=preserve_initial_state, boundary_flux=boundary_flux)
        if boundary_flux is not None:
            bf = as_f64(boundary_flux)
            if bf.shape == (self.NZ, self.NR):
                bf = np.broadcast_to(bf, (self.batch, self.NZ, self.NR))
            if bf.shape != (self.batch, self.NZ, self.NR):
                raise ValueError(f"boundary_flux shape {bf.shape} must match Psi shape {(self.NZ, self.NR)}")
            self.set_boundary_flux(np.ascontiguousarray(bf))
        p = picard_params_from_cfg(self.cfg, preserve_initial_state=preserve_initial_state,
                                   has_boundary_flux=boundary_flux is not None,
                                   external_profile_mode=self.external_profile_mode, check_every=check_every)
        res = (PicardResult * self.batch)()
        mi = max(int(p.max_iter), 1)
        hd = np.zeros((self.batch, mi)) if history else None
        hg = np.zeros((self.batch, mi)) if history else None
        tr = np.zeros((self.batch, mi, 6)) if trace else None
        check(self._L.scpn_gs_picard_solve(self._h, C.byref(p), res, ptr(hd), ptr(hg), ptr(tr)))
        out = []
        for b in range(self.batch):
            d = _result_dict(res[b])
            n = d["iterations"]
            if history:
                d["residual_history"] = hd[b, :n].tolist()
                d["gs_residual_history"] = hg[b, :n].tolist()
            if trace:
                d["trace"] = tr[b, :n].copy()
            out.append(d)
        return out

    def solve_newton(self, preserve_initial_state=False, boundary_flux=None, *, poll_every=10) -> list[dict[str, Any]]:
        """Newton-Kantorovich + GMRES solve of every member (core/fusion_kernel_newton_solver.py:104-372).

        Result dicts carry the reference's keys (``residual_history``, ``gs_residual_history``, ``newton_line_search*``,
        ``gmres_*``, ``solver_method="newton"``).  The GMRES non-convergence budget / fail-on-breakdown policy of
        ``GmresTelemetry`` (core/fusion_kernel_newton_runtime.py:44-72) is applied to the returned counters.
        """
        if boundary_flux is not None:
            bf = as_f64(boundary_flux)
            if bf.shape == (self.NZ, self.NR):
                bf = np.broadcast_to(bf, (self.batch, self.NZ, self.NR))
            if bf.shape != (self.batch, self.NZ, self.NR):
                raise ValueError(f"boundary_flux shape {bf.shape} must match Psi shape {(self.NZ, self.NR)}")
            self.set_boundary_flux(np.ascontiguousarray(bf))
        p = picard_params_from_cfg({**self.cfg, "solver": {**self.cfg["solver"], "solver_method": "newton"}},
                                   preserve_initial_state=preserve_initial_state, has_boundary_flux=boundary_flux is not None,
                                   external_profile_mode=self.external_profile_mode)
        o = newton_opts_from_cfg(self.cfg, poll_every=poll_every)
        res = (NewtonResult * self.batch)()
        cap = max(int(p.max_iter), 0) + 2
        hd, hg = np.zeros((self.batch, cap)), np.zeros((self.batch, cap))
        check(self._L.scpn_gs_newton_solve(self._h, C.byref(p), C.byref(o), res, ptr(hd), ptr(hg)))
        sol = self.cfg["solver"]
        budget = sol.get("gmres_nonconverged_budget")
        out = []
        for b in range(self.batch):
            r = res[b]
            if budget is not None and r.gmres_nonconverged > int(budget):
                raise RuntimeError(f"GMRES non-convergence budget exceeded: {r.gmres_nonconverged}>{int(budget)} "
                                   f"(last info={r.gmres_last_info}, member={b})")
            out.append({"converged": bool(r.converged), "iterations": int(r.iterations), "diverged": bool(r.diverged),
                        "residual": r.residual, "residual_history": hd[b, : r.n_hist].tolist(), "gs_residual": r.gs_residual,
                        "gs_residual_best": r.gs_residual_best, "gs_residual_history": hg[b, : r.n_gs_hist].tolist(),
                        "newton_line_search": bool(o.line_search), "newton_line_search_attempts": int(r.ls_attempts),
                        "newton_line_search_accepts": int(r.ls_accepts), "newton_line_search_rejects": int(r.ls_rejects),
                        "gmres_nonconverged_count": int(r.gmres_nonconverged), "gmres_breakdown_count": int(r.gmres_breakdown),
                        "gmres_last_info": int(r.gmres_last_info), "gmres_inner_iterations": int(r.gmres_inner),
                        "gmres_fail_on_breakdown": bool(sol.get("gmres_fail_on_breakdown", True)),
                        "gmres_nonconverged_budget": None if budget is None else int(budget), "solver_method": "newton"})
        return out

    def solve_free_boundary(self, coil_r=None, coil_z=None, coil_currents=None, turns=None, *, max_outer_iter=20, tol=1e-4,
                            check_every=8) -> list[dict[str, Any]]:
        """``solve_free_boundary(optimize_shape=False)`` of every member
        (core/fusion_kernel_free_boundary.py:623-739).  Coil arrays are ``(batch, n)`` (1-D arrays are
        broadcast to every member); defaults: the config's coil set, one turn each."""
        if max_outer_iter < 1:
            raise ValueError("max_outer_iter must be >= 1.")
        if not np.isfinite(tol) or tol < 0.0:
            raise ValueError("tol must be finite and >= 0.")
        newton = self.cfg.get("solver", {}).get("solver_method", "multigrid") == "newton"

        def bc(a, default):
            a = default if a is None else as_f64(a)
            if a.ndim == 1:
                a = np.tile(a, (self.batch, 1))
            return np.ascontiguousarray(a)

        cr, cz, ci = bc(coil_r, self.coil_r), bc(coil_z, self.coil_z), bc(coil_currents, self.coil_i)
        n = cr.shape[1]
        ct = bc(turns, np.ones((self.batch, n)))
        for a in (cz, ci, ct):
            if a.shape != (self.batch, n):
                raise ValueError("coil arrays must share the shape (batch, n_coils)")
        p = picard_params_from_cfg(self.cfg, preserve_initial_state=True, has_boundary_flux=True,
                                   external_profile_mode=self.external_profile_mode, check_every=check_every)
        res = (PicardResult * self.batch)()
        outer = np.zeros(self.batch, dtype=np.int32)
        fdiff = np.zeros(self.batch)
        pits = np.zeros((self.batch, max_outer_iter), dtype=np.int32)
        if newton:  # the same outer loop, Newton-Kantorovich as the inner solver (picard_iterations = its iteration counts)
            o = newton_opts_from_cfg(self.cfg)
            check(self._L.scpn_gs_free_boundary_solve_newton(self._h, C.byref(p), C.byref(o), n, ptr(cr), ptr(cz), ptr(ci), ptr(ct),
                                                             int(max_outer_iter), float(tol), ptr(outer), ptr(fdiff), ptr(pits), res))
        else:
            check(self._L.scpn_gs_free_boundary_solve(self._h, C.byref(p), n, ptr(cr), ptr(cz), ptr(ci), ptr(ct), int(max_outer_iter),
                                                      float(tol), ptr(outer), ptr(fdiff), ptr(pits), res))
        out = []
        for b in range(self.batch):
            d = _result_dict(res[b])
            d.update(outer_iterations=int(outer[b]), final_diff=float(fdiff[b]),
                     picard_iterations=[int(v) for v in pits[b, : outer[b]]], coil_currents=ci[b].copy())
            out.append(d)
        return out

    # -- resident operators (zero-copy smoother benchmark path) ------------------------------------
    def solve_free_boundary_shape(self, coils_per_member, *, max_outer_iter=20, tol=1e-4, tikhonov_alpha=1e-4) -> list[dict[str, Any]]:
        """``solve_free_boundary(optimize_shape=True)`` of every member (core/fusion_kernel_free_boundary.py:623-739): the shape
        scan of an ensemble.  ``coils_per_member`` is one :class:`~scpn_fusion_core_amd.free_boundary.CoilSet` per member
        (positions / turns / limits / target points may differ; ``currents`` are updated in place like the reference does).

        Per outer iteration the equilibrium solves of all members that still iterate run in ONE device call
        (per-member Dirichlet maps, frozen members keep their state through ``preserve_initial_state``), the isoflux targets
        are sampled on the device, the coil responses come from the device Green's-function kernel, and only the tiny bounded
        least-squares problems (n_points + n_coils rows) are solved on the host, member by member, as in the reference."""
        from . import free_boundary as fb

        if len(coils_per_member) != self.batch:
            raise ValueError("one CoilSet per ensemble member is required")
        if max_outer_iter < 1:
            raise ValueError("max_outer_iter must be >= 1.")
        if not np.isfinite(tol) or tol < 0.0:
            raise ValueError("tol must be finite and >= 0.")
        for c in coils_per_member:
            if c.target_flux_points is None:
                raise ValueError("CoilSet.target_flux_points must be set for shape optimisation.")

        class _Axes:  # what the free-boundary helpers read from a kernel
            R, Z, NR, NZ, dR, dZ = self.R, self.Z, self.NR, self.NZ, self.dR, self.dZ

        def ext_flux(b):
            return fb.compute_external_flux(_Axes, coils_per_member[b])

        B = self.batch
        psi_ext = np.stack([ext_flux(b) for b in range(B)])
        diff = np.full(B, np.inf)
        outer_done = np.zeros(B, dtype=int)
        active = np.ones(B, dtype=bool)
        shape_info: list[Any] = [None] * B
        last_res: list[Any] = [None] * B
        for outer in range(max_outer_iter):
            psi = self.get_psi()
            psi[:, 0, :], psi[:, -1, :], psi[:, :, 0], psi[:, :, -1] = psi_ext[:, 0, :], psi_ext[:, -1, :], psi_ext[:, :, 0], psi_ext[:, :, -1]
            psi_old = psi.copy()
            saved = psi.copy()
            self.set_psi(psi)
            res = self.solve_equilibrium(preserve_initial_state=True, boundary_flux=psi_ext)
            new = self.get_psi()
            if not active.all():  # members that already converged keep their final state
                new[~active] = saved[~active]
                self.set_psi(new)
            for b in np.flatnonzero(active):
                cs = coils_per_member[b]
                pts = np.asarray(cs.target_flux_points, dtype=np.float64)
                if cs.target_flux_values is not None:
                    target = fb.resolve_shape_target_flux(None, cs)
                else:  # isoflux: mean of the bilinear samples (device sampling of member b)
                    target = None
                shape_info[b] = (pts, target)
            need = [b for b in np.flatnonzero(active) if shape_info[b][1] is None]
            if need:
                # one device sampling call per distinct point set; members usually share it
                cache: dict[bytes, np.ndarray] = {}
                for b in need:
                    key = shape_info[b][0].tobytes()
                    if key not in cache:
                        cache[key] = self.sample_psi(shape_info[b][0])
                    samples = cache[key][b]
                    shape_info[b] = (shape_info[b][0], np.full(samples.shape[0], float(np.mean(samples))))
            for b in np.flatnonzero(active):
                cs = coils_per_member[b]
                pts, target = shape_info[b]
                response = fb.build_mutual_inductance_matrix(None, cs, pts)
                new_cur = np.asarray(fb.optimize_coil_currents(None, cs, target, tikhonov_alpha=tikhonov_alpha), dtype=np.float64).reshape(-1)
                achieved = response.T @ new_cur
                residual = achieved - target
                rmse = float(np.sqrt(np.mean(residual**2)))
                shape_info[b] = {"solver_mode": "free_boundary_solver_shape_current_optimization", "target_point_count": int(target.shape[0]),
                                 "coil_count": len(cs.positions), "flux_rmse": rmse,
                                 "flux_relative_rmse": float(rmse / max(float(np.sqrt(np.mean(target**2))), 1.0)),
                                 "max_abs_flux_residual": float(np.max(np.abs(residual))), "target_flux": target.copy(),
                                 "achieved_flux": achieved}
                cs.currents = new_cur
                psi_ext[b] = ext_flux(b)
                diff[b] = float(np.max(np.abs(new[b] - psi_old[b])))
                outer_done[b] = outer + 1
                last_res[b] = res[b]
                if diff[b] < tol:
                    active[b] = False
            if not active.any():
                break
        return [{"outer_iterations": int(outer_done[b]), "final_diff": float(diff[b]),
                 "coil_currents": np.asarray(coils_per_member[b].currents, dtype=np.float64).copy(), "shape_optimization": shape_info[b],
                 "converged": bool(last_res[b]["converged"]) if last_res[b] else False} for b in range(B)]

    def rb_sor(self, omega: float, n_sweeps: int):
        check(self._L.scpn_gs_rb_sor(self._h, float(omega), int(n_sweeps)))

    def vcycle(self, omega=1.0, pre_smooth=3, post_smooth=3, min_grid=5):
        check(self._L.scpn_gs_vcycle(self._h, float(omega), int(pre_smooth), int(post_smooth), int(min_grid)))

    def sample_psi(self, points) -> np.ndarray:
        """Bilinear psi samples of every member at ``points`` (n, 2) = (R, Z): ``sample_flux_at_points`` / ``interp_psi``
        (core/fusion_kernel_free_boundary.py:156-159, 562-580) evaluated on the device; returns (batch, n)."""
        pts = as_f64(points)
        if pts.ndim != 2 or pts.shape[1] != 2 or pts.shape[0] < 1:
            raise ValueError("points must have shape (n_points, 2) with n_points > 0.")
        if not np.all(np.isfinite(pts)):
            raise ValueError("points must contain finite values only.")
        pr, pz = np.ascontiguousarray(pts[:, 0]), np.ascontiguousarray(pts[:, 1])
        out = np.empty((self.batch, pts.shape[0]))
        check(self._L.scpn_gs_sample_psi(self._h, pts.shape[0], pr, pz, out))
        return out

    def compute_b_field_all(self):
        """(B_R, B_Z) of every member, each (batch, NZ, NR) (core/fusion_kernel.py:450-456)."""
        br, bz = np.empty((self.batch, self.NZ, self.NR)), np.empty((self.batch, self.NZ, self.NR))
        check(self._L.scpn_gs_compute_b_field(self._h, -1, br, bz))
        return br, bz

    def compute_b_field(self, member=0):
        br, bz = np.empty((self.NZ, self.NR)), np.empty((self.NZ, self.NR))
        check(self._L.scpn_gs_compute_b_field(self._h, int(member), br, bz))
        return br, bz

    def snapshot_psi(self):
        """Keep a device copy of every member's current psi (the ``psi_old`` of the transport coupling's flux-change test)."""
        check(self._L.scpn_gs_snapshot_psi(self._h))

    def flux_change(self) -> np.ndarray:
        """``||psi - psi_old|| / ||psi_old||`` per member against the last :meth:`snapshot_psi`, evaluated on the device
        (core/integrated_transport_solver_coupling.py:131-146; ``||psi_old|| < 1e-30`` counts as 1)."""
        out = np.empty(self.batch)
        check(self._L.scpn_gs_flux_change(self._h, out))
        return out

    def fill_psi(self, value: float = 0.0):
        """psi <- value on every member, on the device."""
        check(self._L.scpn_gs_fill_psi(self._h, float(value)))

    def pack_psi_device(self) -> int:
        """Dense [batch][nz][nr] device copy of the current psi; returns the device address (RCCL send buffer)."""
        p = C.c_void_p()
        check(self._L.scpn_gs_pack_psi_device(self._h, C.byref(p)))
        return int(p.value)

    # -- device timing (bench.py) -------------------------------------------------------------------
    def profile(self, enable: bool):
        check(self._L.scpn_gs_profile(self._h, int(bool(enable))))

    def profile_read(self, name: str):
        ms, n = C.c_double(), C.c_long()
        check(self._L.scpn_gs_profile_read(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_report(self) -> dict:
        """{family: {"ms": total device ms, "launches": n}} for every kernel family timed since profile(True)."""
        buf = C.create_string_buffer(4096)
        check(self._L.scpn_gs_profile_names(self._h, buf, 4096))
        out = {}
        for name in filter(None, buf.value.decode().split(",")):
            ms, n = self.profile_read(name)
            if n:
                out[name] = {"ms": round(ms, 3), "launches": int(n)}
        return out

    @property
    def arithmetic(self) -> str:
        return _ARITHMETIC_NAMES[int(self._L.scpn_gs_get_arith(self._h))]

    @arithmetic.setter
    def arithmetic(self, mode: str):
        if mode not in _ARITHMETIC_NAMES:
            raise ValueError(f"arithmetic must be one of {_ARITHMETIC_NAMES}, not {mode!r}")
        check(self._L.scpn_gs_set_arith(self._h, _ARITHMETIC_NAMES.index(mode)))

    @property
    def num_groups(self) -> int:
        return int(self._L.scpn_gs_num_groups(self._h))

    def sync(self):
        check(self._L.scpn_gs_sync(self._h))

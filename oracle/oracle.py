"""ctypes wrapper over oracle/libgs_oracle.so -- the CPU restatement of the reference path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg; the product package never imports this module.
Signatures mirror the reference free functions (file:line in gs_oracle.h).
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libgs_oracle.so"
_REF_PATH = _HERE / "_ref" / "libscpn_solver_ref.so"

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_d = C.c_double
_i = C.c_int


class PicardParams(C.Structure):
    _fields_ = [
        ("nr", _i), ("nz", _i),
        ("r_min", _d), ("r_max", _d), ("z_min", _d), ("z_max", _d),
        ("mu0", _d), ("i_target", _d),
        ("method", _i), ("sor_omega", _d), ("alpha", _d), ("tol", _d), ("max_iter", _i),
        ("require_gs_residual", _i), ("gs_tol", _d),
        ("hmode", _i), ("ped_p", _d * 4), ("ped_ff", _d * 4),
        ("preserve_initial_state", _i), ("has_boundary_flux", _i), ("external_profile_mode", _i),
        ("n_coils", _i),
        ("coil_r", C.POINTER(_d)), ("coil_z", C.POINTER(_d)), ("coil_i", C.POINTER(_d)),
        ("anderson_m", _i), ("xpoint_saddle", _i),
    ]


class PicardResult(C.Structure):
    _fields_ = [
        ("converged", _i), ("iterations", _i), ("diverged", _i),
        ("residual", _d), ("gs_residual", _d), ("gs_residual_best", _d),
        ("psi_axis", _d), ("psi_bnd", _d), ("x_idx", C.c_long),
    ]


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when the reference checkout is present)."""
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < (_HERE / "gs_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "libgs_oracle.so"], check=True, capture_output=True)
    if Path("/root/reference/src/scpn_fusion/hpc/solver.cpp").exists() and (force or not _REF_PATH.exists()):
        subprocess.run(["make", "-C", str(_HERE), "ref"], check=True, capture_output=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        L.orc_np_sum.restype = _d
        L.orc_np_sum.argtypes = [_dp, C.c_long]
        L.orc_linspace.argtypes = [_d, _d, _i, _dp]
        L.orc_mg_smooth.argtypes = [_dp, _dp, _dp, _i, _i, _d, _d, _d, _i]
        L.orc_mg_residual.argtypes = [_dp, _dp, _dp, _i, _i, _d, _d, _dp]
        L.orc_residual_linf.restype = _d
        L.orc_residual_linf.argtypes = [_dp, _dp, _dp, _i, _i, _d, _d]
        L.orc_stable_rms.restype = _d
        L.orc_stable_rms.argtypes = [_dp, C.c_long]
        L.orc_gs_residual_rms.restype = _d
        L.orc_gs_residual_rms.argtypes = [_dp, _dp, _dp, _i, _i, _d, _d]
        L.orc_restrict_full_weight.argtypes = [_dp, _i, _i, _dp]
        L.orc_prolongate_bilinear.argtypes = [_dp, _i, _i, _i, _i, _dp]
        L.orc_multigrid_vcycle.argtypes = [_dp, _dp, _dp, _i, _i, _d, _d, _d, _i, _i, _i]
        L.orc_multigrid_solve.argtypes = [_dp, _dp, _d, _d, _d, _d, _i, _i, _d, _i, _d, _i, _i, _i,
                                          C.POINTER(_d), C.POINTER(_i), C.POINTER(_i)]
        L.orc_jacobi_step.argtypes = [_dp, _dp, _dp, _i, _i, _d, _d, _dp]
        L.orc_sor_step.argtypes = [_dp, _dp, _dp, _i, _i, _d, _d, _d, _dp]
        L.orc_ellip_ke.argtypes = [_d, C.POINTER(_d), C.POINTER(_d)]
        L.orc_vacuum_field.argtypes = [_dp, _dp, _i, _i, _i, _dp, _dp, _dp, _d, _dp]
        L.orc_green_function.restype = _d
        L.orc_green_function.argtypes = [_d, _d, _d, _d]
        L.orc_external_flux.argtypes = [_dp, _dp, _i, _i, _i, _dp, _dp, _dp, _dp, _dp]
        L.orc_find_axis.argtypes = [_dp, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_d)]
        L.orc_find_x_point.argtypes = [_dp, _dp, _i, _i, _d, _d, _d, C.POINTER(C.c_long), C.POINTER(_d)]
        L.orc_find_x_point_saddle.argtypes = [_dp, _dp, _i, _i, _d, _d, _d, C.POINTER(C.c_long), C.POINTER(_d)]
        L.orc_update_source.restype = _d
        L.orc_update_source.argtypes = [_dp, _dp, _i, _i, _d, _d, _d, _d, _d, _d, _i, _dp, _dp, _dp]
        L.orc_picard_solve.argtypes = [C.POINTER(PicardParams), _dp, C.c_void_p, _dp,
                                       C.POINTER(PicardResult), C.c_void_p, _i]
        L.orc_free_boundary_solve.argtypes = [C.POINTER(PicardParams), _dp, _dp, _i, _dp, _dp, _dp, _dp,
                                              _i, _d, C.POINTER(_d), C.POINTER(_i), C.POINTER(PicardResult)]
        L.orc_gs_solve_np.argtypes = [_d, _d, _d, _d, _i, _i, _d, _d, _i, _i, _d, _d, _d, _dp]
        L.orc_np_delta_star.argtypes = [_dp, _i, _i, _d, _d, _d, _d, _dp]
        L.orc_np_equation_residual.argtypes = [_dp, _i, _i, _d, _d, _d, _d, _d, _d, _d, _dp]
        L.orc_fastsolver_sweeps.restype = _d
        L.orc_fastsolver_sweeps.argtypes = [_dp, _dp, _i, _i, _d, _d, _d, _d, _d, _d, _i, _d, C.POINTER(_i)]
        L.orc_interp_psi.restype = _d
        L.orc_interp_psi.argtypes = [_dp, _dp, _dp, _i, _i, _d, _d, _d, _d]
        L.orc_b_field.argtypes = [_dp, _dp, _i, _i, _d, _d, _dp, _dp]
        L.orc_probe_response.argtypes = [_i, _dp, _dp, C.c_void_p, _i, C.c_void_p, C.c_void_p, _i, C.c_void_p, C.c_void_p,
                                         C.c_void_p, _dp]
        L.orc_np_interp.argtypes = [_dp, C.c_long, _dp, _dp, _i, _dp]
        L.orc_map_profiles_to_2d.restype = _d
        L.orc_map_profiles_to_2d.argtypes = [_dp, _dp, _dp, _i, _i, _d, _d, _d, _d, _dp, _dp, _dp, _i, _dp, _dp]
        L.orc_gtsv.argtypes = [_i, _dp, _dp, _dp, _dp]
        L.orc_neoclassical_resistivity.restype = _d
        L.orc_neoclassical_resistivity.argtypes = [_d, _d, _d, _d]
        L.orc_q_from_psi.argtypes = [_dp, _dp, _i, _d, _d, _d, _dp]
        L.orc_current_diffusion_step.argtypes = [_dp, _dp, _i, _d, _d, _d, _d, _dp, _dp, _d, _dp, _dp, C.c_void_p]
        L.orc_flux_evolution.argtypes = [_i, _i, _d, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_dlartg.argtypes = [_d, _d, C.POINTER(_d), C.POINTER(_d), C.POINTER(_d)]
        L.orc_gmres_gs.argtypes = [_dp, _dp, _i, _i, _d, _d, _dp, C.c_void_p, _i, _i, _d, _d, _dp, C.POINTER(C.c_long)]
        L.orc_profile_jacobian.argtypes = [_dp, _dp, _dp, _i, _i, _d, _d, _d, _d, _d, _i, _dp]
        L.orc_newton_mg_fd.restype = _i
        L.orc_newton_mg_fd.argtypes = [C.POINTER(PicardParams), C.POINTER(NewtonMgOpts), _dp, C.c_void_p, _dp,
                                       C.POINTER(NewtonResult), C.c_void_p, C.c_void_p]
        L.orc_newton_solve.argtypes = [C.POINTER(PicardParams), C.POINTER(NewtonOpts), _dp, C.c_void_p, _dp,
                                       C.POINTER(NewtonResult), C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def linspace(a, b, n):
    out = np.empty(n)
    lib().orc_linspace(a, b, n, out)
    return out


def np_sum(a):
    a = _c(a).ravel()
    return lib().orc_np_sum(a, a.size)


def mg_smooth(psi, source, r_grid, dr, dz, omega, n_sweeps):
    psi = _c(psi).copy()
    nz, nr = psi.shape
    if lib().orc_mg_smooth(psi, _c(source), _c(r_grid), nz, nr, dr, dz, omega, n_sweeps):
        raise ValueError("omega must be finite and satisfy 1.0 <= omega < 2.0")
    return psi


def mg_residual(psi, source, r_grid, dr, dz):
    psi = _c(psi)
    nz, nr = psi.shape
    out = np.empty_like(psi)
    lib().orc_mg_residual(psi, _c(source), _c(r_grid), nz, nr, dr, dz, out)
    return out


def residual_linf(psi, source, r_grid, dr, dz):
    psi = _c(psi)
    nz, nr = psi.shape
    return lib().orc_residual_linf(psi, _c(source), _c(r_grid), nz, nr, dr, dz)


def gs_residual_rms(psi, source, r_grid, dr, dz):
    psi = _c(psi)
    nz, nr = psi.shape
    return lib().orc_gs_residual_rms(psi, _c(source), _c(r_grid), nz, nr, dr, dz)


def stable_rms(v):
    v = _c(v).ravel()
    return lib().orc_stable_rms(v, v.size)


def restrict_full_weight(fine):
    fine = _c(fine)
    nz, nr = fine.shape
    out = np.empty(((nz + 1) // 2, (nr + 1) // 2))
    lib().orc_restrict_full_weight(fine, nz, nr, out)
    return out


def prolongate_bilinear(coarse, nz_f, nr_f):
    coarse = _c(coarse)
    nz_c, nr_c = coarse.shape
    out = np.empty((nz_f, nr_f))
    if lib().orc_prolongate_bilinear(coarse, nz_c, nr_c, nz_f, nr_f, out):
        raise ValueError("inconsistent coarse/fine sizes")
    return out


def multigrid_vcycle(psi, source, r_grid, dr, dz, *, omega=1.0, pre_smooth=3, post_smooth=3, min_grid=5):
    psi = _c(psi).copy()
    nz, nr = psi.shape
    if lib().orc_multigrid_vcycle(psi, _c(source), _c(r_grid), nz, nr, dr, dz, omega, pre_smooth, post_smooth, min_grid):
        raise ValueError("omega must be finite and satisfy 1.0 <= omega < 2.0")
    return psi


def multigrid_solve(source, psi_bc, r_min, r_max, z_min, z_max, nr, nz, *, tol=1e-6, max_cycles=500,
                    omega=1.0, pre_smooth=3, post_smooth=3, min_grid=5):
    psi = _c(psi_bc).copy()
    source = _c(source)
    if source.shape != (nz, nr) or psi.shape != (nz, nr):
        raise ValueError("source and psi_bc must have shape (nz, nr)")
    res, ncy, conv = _d(), _i(), _i()
    rc = lib().orc_multigrid_solve(source, psi, r_min, r_max, z_min, z_max, nr, nz, tol, max_cycles, omega,
                                   pre_smooth, post_smooth, min_grid, C.byref(res), C.byref(ncy), C.byref(conv))
    if rc:
        raise ValueError("bad tol / max_cycles / omega")
    return psi, res.value, ncy.value, bool(conv.value)


def native_profile_source_reconstruction(r_axis, z_axis, boundary_psi, jtor, rel_tol=1e-13, max_cycles=40):
    """CPU counterpart of the strict-parity harness's native lane (validation/benchmark_freegs_public_example_reconstruction.py:275-315):
    the linear system the reference factorises with SuperLU -- rows 1/dZ^2, 1/dR^2 +- 1/(2 R dR), -2(1/dR^2 + 1/dZ^2), right-hand side
    -mu0 R J inside the ring of ``boundary_psi`` (R-slow (len(R), len(Z)) arrays) -- relaxed to round-off by ``orc_multigrid_solve``
    on linspace axes.  Test infrastructure: checked against the reference's own output in tests/test_oracle_golden.py."""
    r_axis, z_axis = np.asarray(r_axis, dtype=np.float64), np.asarray(z_axis, dtype=np.float64)
    src = np.ascontiguousarray((-(4.0e-7 * np.pi) * r_axis[:, None] * np.asarray(jtor, dtype=np.float64)).T)
    ring = np.asarray(boundary_psi, dtype=np.float64).T
    start = np.zeros_like(src)
    start[0, :], start[-1, :], start[:, 0], start[:, -1] = ring[0, :], ring[-1, :], ring[:, 0], ring[:, -1]
    scale = float(np.abs(src).max())
    psi, res, _, _ = multigrid_solve(src, start, r_axis[0], r_axis[-1], z_axis[0], z_axis[-1], r_axis.size, z_axis.size,
                                     tol=rel_tol * scale, max_cycles=max_cycles)
    if res > 1e-10 * scale:
        raise RuntimeError(f"residual {res:.3e} did not reach round-off")
    return np.ascontiguousarray(psi.T)


def jacobi_step(psi, source, r_grid, dr, dz):
    psi = _c(psi)
    nz, nr = psi.shape
    out = np.empty_like(psi)
    lib().orc_jacobi_step(psi, _c(source), _c(r_grid), nz, nr, dr, dz, out)
    return out


def sor_step(psi, source, r_grid, dr, dz, omega):
    psi = _c(psi)
    nz, nr = psi.shape
    out = np.empty_like(psi)
    if lib().orc_sor_step(psi, _c(source), _c(r_grid), nz, nr, dr, dz, omega, out):
        raise ValueError("omega must be finite and satisfy 1.0 <= omega < 2.0")
    return out


def ellip_ke(m):
    k, e = _d(), _d()
    lib().orc_ellip_ke(float(m), C.byref(k), C.byref(e))
    return k.value, e.value


def vacuum_field(r_axis, z_axis, coils_r, coils_z, coils_i, mu0):
    r_axis, z_axis = _c(r_axis), _c(z_axis)
    out = np.empty((z_axis.size, r_axis.size))
    lib().orc_vacuum_field(r_axis, z_axis, z_axis.size, r_axis.size, len(coils_r), _c(coils_r), _c(coils_z),
                           _c(coils_i), mu0, out)
    return out


def green_function(r_src, z_src, r_obs, z_obs):
    return lib().orc_green_function(r_src, z_src, r_obs, z_obs)


def external_flux(r_axis, z_axis, coils_r, coils_z, coils_i, turns):
    r_axis, z_axis = _c(r_axis), _c(z_axis)
    out = np.empty((z_axis.size, r_axis.size))
    lib().orc_external_flux(r_axis, z_axis, z_axis.size, r_axis.size, len(coils_r), _c(coils_r), _c(coils_z),
                            _c(coils_i), _c(turns), out)
    return out


def find_axis(psi):
    psi = _c(psi)
    iz, ir, v = _i(), _i(), _d()
    lib().orc_find_axis(psi, psi.shape[0], psi.shape[1], C.byref(iz), C.byref(ir), C.byref(v))
    return iz.value, ir.value, v.value


def find_x_point_saddle(psi, z_axis, dr, dz, z_min_cfg):
    psi = _c(psi)
    idx, px = C.c_long(), _d()
    lib().orc_find_x_point_saddle(psi, _c(z_axis), psi.shape[0], psi.shape[1], dr, dz, z_min_cfg, C.byref(idx), C.byref(px))
    return idx.value, px.value


def find_x_point(psi, z_axis, dr, dz, z_min_cfg):
    psi = _c(psi)
    idx, v = C.c_long(), _d()
    lib().orc_find_x_point(psi, _c(z_axis), psi.shape[0], psi.shape[1], dr, dz, z_min_cfg, C.byref(idx), C.byref(v))
    return idx.value, v.value


_PED = (0.92, 0.05, 1.0, 0.3)


def update_source(psi, r_axis, dr, dz, psi_axis, psi_bnd, mu0, i_target, hmode=False, ped_p=_PED, ped_ff=_PED):
    psi = _c(psi)
    j = np.empty_like(psi)
    i_cur = lib().orc_update_source(psi, _c(r_axis), psi.shape[0], psi.shape[1], dr, dz, psi_axis, psi_bnd, mu0,
                                    i_target, int(hmode), _c(ped_p), _c(ped_ff), j)
    return j, i_cur


_METHODS = {"multigrid": 0, "sor": 1, "jacobi": 2, "anderson": 3, "newton": 4}


def make_params(cfg: dict, *, preserve=False, has_bc=False, external_profile_mode=False):
    """Build PicardParams from a reference-style config dict (core/config_schema.py)."""
    dims, phys, sol = cfg["dimensions"], cfg["physics"], cfg.get("solver", {})
    p = PicardParams()
    p.nr, p.nz = cfg["grid_resolution"]
    p.r_min, p.r_max, p.z_min, p.z_max = dims["R_min"], dims["R_max"], dims["Z_min"], dims["Z_max"]
    p.mu0 = phys["vacuum_permeability"]
    p.i_target = phys["plasma_current_target"]
    p.method = _METHODS[sol.get("solver_method", "multigrid")]
    p.sor_omega = sol.get("sor_omega", 1.6)
    p.alpha = sol.get("relaxation_factor", 0.1)
    p.tol = sol["convergence_threshold"]
    p.max_iter = sol["max_iterations"]
    p.require_gs_residual = int(bool(sol.get("require_gs_residual", False)))
    p.gs_tol = float(sol.get("gs_residual_threshold", p.tol))
    prof = phys.get("profiles") or {}
    p.hmode = int(prof.get("mode", "l-mode") in ("h-mode", "H-mode", "hmode"))
    pp = dict(zip(("ped_top", "ped_width", "ped_height", "core_alpha"), _PED))
    pf = dict(pp)
    pp.update(prof.get("p_prime", {}))
    pf.update(prof.get("ff_prime", {}))
    p.ped_p = (_d * 4)(pp["ped_top"], pp["ped_width"], pp["ped_height"], pp["core_alpha"])
    p.ped_ff = (_d * 4)(pf["ped_top"], pf["ped_width"], pf["ped_height"], pf["core_alpha"])
    p.preserve_initial_state = int(preserve)
    p.has_boundary_flux = int(has_bc)
    p.external_profile_mode = int(external_profile_mode)
    coils = cfg.get("coils", [])
    p.n_coils = len(coils)
    p.anderson_m = int(sol.get("anderson_depth", 5))
    p.xpoint_saddle = int(bool(sol.get("xpoint_use_saddle_detection", False)))
    keep = [np.array([c[k] for c in coils], dtype=np.float64) for k in ("r", "z", "current")]
    p.coil_r, p.coil_z, p.coil_i = [a.ctypes.data_as(C.POINTER(_d)) for a in keep]
    p._keep = keep  # keep the arrays alive
    return p


def picard_solve(cfg, psi0=None, boundary_flux=None, *, preserve=False, external_profile_mode=False,
                 trace=False, max_iter_budget=-1):
    p = make_params(cfg, preserve=preserve, has_bc=boundary_flux is not None,
                    external_profile_mode=external_profile_mode)
    nz, nr = p.nz, p.nr
    psi = np.zeros((nz, nr)) if psi0 is None else _c(psi0).copy()
    j = np.zeros((nz, nr))
    res = PicardResult()
    tr = np.full((p.max_iter, 6), np.nan) if trace else None
    bf = None if boundary_flux is None else _c(boundary_flux)
    lib().orc_picard_solve(C.byref(p), psi, None if bf is None else bf.ctypes.data, j, C.byref(res),
                           None if tr is None else tr.ctypes.data, max_iter_budget)
    out = {"psi": psi, "j_phi": j, "converged": bool(res.converged), "iterations": res.iterations,
           "diverged": bool(res.diverged), "residual": res.residual, "gs_residual": res.gs_residual,
           "gs_residual_best": res.gs_residual_best, "psi_axis": res.psi_axis, "psi_bnd": res.psi_bnd,
           "x_idx": res.x_idx}
    if trace:
        out["trace"] = tr[: res.iterations]
    return out


class NewtonOpts(C.Structure):
    _fields_ = [("precond", _i), ("line_search", _i), ("line_search_c", _d), ("max_backtracks", _i)]


class NewtonMgOpts(C.Structure):
    _fields_ = [("warmup_steps", _i), ("newton_alpha", _d), ("gmres_restart", _i), ("gmres_maxiter", _i), ("gmres_atol", _d),
                ("gmres_rtol", _d)]


class NewtonResult(C.Structure):
    _fields_ = [
        ("converged", _i), ("iterations", _i), ("diverged", _i),
        ("residual", _d), ("gs_residual", _d), ("gs_residual_best", _d),
        ("ls_attempts", _i), ("ls_accepts", _i), ("ls_rejects", _i),
        ("gmres_nonconverged", _i), ("gmres_breakdown", _i), ("gmres_last_info", _i),
        ("n_hist", _i), ("n_gs_hist", _i), ("gmres_matvecs", C.c_long),
    ]


def newton_opts_from_cfg(cfg: dict) -> NewtonOpts:
    """core/fusion_kernel_newton_runtime.py:75-150 parse_newton_dispatch_config (the GMRES / line-search keys)."""
    sol = cfg["solver"]
    mode = str(sol.get("gmres_preconditioner", "none")).strip().lower()
    if mode in ("", "auto", "default", "legacy"):
        mode = "none"
    if mode == "none" and bool(sol.get("gmres_diagonal_preconditioner", False)):
        mode = "diagonal"
    if mode not in ("none", "diagonal"):
        raise ValueError("oracle restates gmres_preconditioner none|diagonal only")
    o = NewtonOpts()
    o.precond = 1 if mode == "diagonal" else 0
    o.line_search = int(bool(sol.get("newton_line_search", False)))
    o.line_search_c = float(sol.get("newton_line_search_c", 1e-4))
    o.max_backtracks = int(sol.get("newton_line_search_max_backtracks", 6))
    return o


def dlartg(f, g):
    c, s, r = _d(), _d(), _d()
    lib().orc_dlartg(float(f), float(g), C.byref(c), C.byref(s), C.byref(r))
    return c.value, s.value, r.value


def profile_jacobian(psi, j_phi, r_axis, dr, dz, psi_axis, psi_bnd, i_target, external_profile_mode=False):
    psi = _c(psi)
    out = np.zeros_like(psi)
    lib().orc_profile_jacobian(psi, _c(j_phi), _c(r_axis), psi.shape[0], psi.shape[1], dr, dz, psi_axis, psi_bnd,
                               i_target, int(external_profile_mode), out)
    return out


def gmres_gs(diag_term, r_grid, dr, dz, rhs, inv_diag=None, restart=50, maxiter=100, atol=1e-8, rtol=1e-6):
    diag_term = _c(diag_term)
    nz, nr = diag_term.shape
    x = np.zeros((nz - 2) * (nr - 2))
    inner = C.c_long()
    info = lib().orc_gmres_gs(diag_term, _c(r_grid), nz, nr, dr, dz, _c(rhs).ravel(),
                              None if inv_diag is None else _c(inv_diag).ravel().ctypes.data, restart, maxiter, atol, rtol,
                              x, C.byref(inner))
    return x, info, inner.value


def newton_solve(cfg, psi0=None, boundary_flux=None, *, preserve=False, external_profile_mode=False):
    p = make_params(cfg, preserve=preserve, has_bc=boundary_flux is not None,
                    external_profile_mode=external_profile_mode)
    o = newton_opts_from_cfg(cfg)
    nz, nr = p.nz, p.nr
    psi = np.zeros((nz, nr)) if psi0 is None else _c(psi0).copy()
    j = np.zeros((nz, nr))
    res = NewtonResult()
    hist = np.full(2 * p.max_iter + 2, np.nan)
    gsh = np.full(2 * p.max_iter + 2, np.nan)
    bf = None if boundary_flux is None else _c(boundary_flux)
    lib().orc_newton_solve(C.byref(p), C.byref(o), psi, None if bf is None else bf.ctypes.data, j, C.byref(res),
                           hist.ctypes.data, gsh.ctypes.data)
    return {"psi": psi, "j_phi": j, "converged": bool(res.converged), "iterations": res.iterations,
            "diverged": bool(res.diverged), "residual": res.residual, "gs_residual": res.gs_residual,
            "gs_residual_best": res.gs_residual_best, "residual_history": hist[: res.n_hist].copy(),
            "gs_residual_history": gsh[: res.n_gs_hist].copy(), "newton_line_search": bool(o.line_search),
            "newton_line_search_attempts": res.ls_attempts, "newton_line_search_accepts": res.ls_accepts,
            "newton_line_search_rejects": res.ls_rejects, "gmres_nonconverged_count": res.gmres_nonconverged,
            "gmres_breakdown_count": res.gmres_breakdown, "gmres_last_info": res.gmres_last_info,
            "gmres_matvecs": res.gmres_matvecs, "solver_method": "newton"}


def newton_mg_fd(cfg, psi0=None, boundary_flux=None, *, preserve=False):
    """CPU restatement of the package's multigrid-preconditioned / finite-difference-Jacobian Newton lane (orc_newton_mg_fd):
    the keys of ``cfg["solver"]`` that lane reads (newton_warmup_steps, newton_damping, gmres_restart / maxiter / rtol / atol),
    with the package's defaults for that lane (GMRES(8) x 1, rtol 1e-3, atol 1e-8; damping as given, default 0.5)."""
    sol = cfg["solver"]
    p = make_params(cfg, preserve=preserve, has_bc=boundary_flux is not None)
    o = NewtonMgOpts(int(sol.get("newton_warmup_steps", -1)), float(sol.get("newton_damping", 0.5)), int(sol.get("gmres_restart", 8)),
                     int(sol.get("gmres_maxiter", 1)), float(sol.get("gmres_atol", 1e-8)), float(sol.get("gmres_rtol", 1e-3)))
    nz, nr = p.nz, p.nr
    psi = np.zeros((nz, nr)) if psi0 is None else _c(psi0).copy()
    j = np.zeros((nz, nr))
    res = NewtonResult()
    hist = np.full(2 * p.max_iter + 2, np.nan)
    gsh = np.full(2 * p.max_iter + 2, np.nan)
    bf = None if boundary_flux is None else _c(boundary_flux)
    lib().orc_newton_mg_fd(C.byref(p), C.byref(o), psi, None if bf is None else bf.ctypes.data, j, C.byref(res), hist.ctypes.data,
                           gsh.ctypes.data)
    return {"psi": psi, "converged": bool(res.converged), "iterations": res.iterations, "diverged": bool(res.diverged),
            "residual": res.residual, "gs_residual": res.gs_residual, "residual_history": hist[: res.n_hist].copy(),
            "gs_residual_history": gsh[: res.n_gs_hist].copy(), "gmres_nonconverged_count": res.gmres_nonconverged,
            "gmres_last_info": res.gmres_last_info, "gmres_matvecs": res.gmres_matvecs}


def interp_psi(psi, r_axis, z_axis, dr, dz, r_pt, z_pt):
    psi = _c(psi)
    return lib().orc_interp_psi(psi, _c(r_axis), _c(z_axis), psi.shape[0], psi.shape[1], dr, dz, float(r_pt), float(z_pt))


def b_field(psi, r_axis, dr, dz):
    psi = _c(psi)
    br, bz = np.zeros_like(psi), np.zeros_like(psi)
    lib().orc_b_field(psi, _c(r_axis), psi.shape[0], psi.shape[1], dr, dz, br, bz)
    return br, bz


def probe_response(coil_r, coil_z, turns, flux_points=None, b_points=None, b_dirs=None):
    cr, cz = _c(coil_r), _c(coil_z)
    ct = None if turns is None else _c(turns)
    fp = np.zeros((0, 2)) if flux_points is None else _c(flux_points).reshape(-1, 2)
    bp = np.zeros((0, 2)) if b_points is None else _c(b_points).reshape(-1, 2)
    fr, fz, br, bz = (np.ascontiguousarray(a) for a in (fp[:, 0], fp[:, 1], bp[:, 0], bp[:, 1]))
    bd = np.array([0 if str(d).upper() == "R" else 1 for d in (b_dirs or [])], dtype=np.int32)
    out = np.zeros((len(fr) + len(br), len(cr)))
    lib().orc_probe_response(len(cr), cr, cz, None if ct is None else ct.ctypes.data, len(fr), fr.ctypes.data, fz.ctypes.data,
                             len(br), br.ctypes.data, bz.ctypes.data, bd.ctypes.data, out)
    return out


def np_interp(x, xp, fp):
    x = _c(x)
    out = np.zeros_like(x)
    lib().orc_np_interp(x.ravel(), x.size, _c(xp), _c(fp), len(xp), out.reshape(-1))
    return out


def map_profiles_to_2d(psi, r_axis, z_axis, dr, dz, z_min_cfg, i_target, rho, pressure_1d, jbs_1d):
    psi = _c(psi)
    p2, j = np.zeros_like(psi), np.zeros_like(psi)
    ic = lib().orc_map_profiles_to_2d(psi, _c(r_axis), _c(z_axis), psi.shape[0], psi.shape[1], dr, dz, z_min_cfg, i_target,
                                      _c(rho), _c(pressure_1d), _c(jbs_1d), len(rho), p2, j)
    return p2, j, ic


def gtsv(dl, d, du, b):
    dl, d, du, b = (_c(a).copy() for a in (dl, d, du, b))
    info = lib().orc_gtsv(len(d), dl, d, du, b)
    return b, info


def current_diffusion_step(rho, psi, r0, a, b0, dt, te, ne, z_eff, j_bs, j_cd, j_ext=None):
    psi = _c(psi).copy()
    je = None if j_ext is None else _c(j_ext)
    info = lib().orc_current_diffusion_step(_c(rho), psi, len(psi), r0, a, b0, dt, _c(te), _c(ne), z_eff, _c(j_bs), _c(j_cd),
                                            None if je is None else je.ctypes.data)
    return psi, info


def flux_evolution(psi0, rate, source, dt):
    rate, source = _c(rate), _c(source)
    n_steps, n = rate.shape[0] - 1, rate.shape[1]
    psi = np.zeros((n_steps + 1, n))
    psi[0] = psi0
    inc, drop, resid = np.zeros((n_steps, n)), np.zeros((n_steps, n)), np.zeros((n_steps, n))
    lib().orc_flux_evolution(n, n_steps, float(dt), rate, source, psi, inc, drop, resid)
    return psi, inc, drop, resid


def free_boundary_solve(cfg, coils_r, coils_z, coils_i, turns, *, psi0=None, max_outer=20, tol=1e-4):
    p = make_params(cfg)
    nz, nr = p.nz, p.nr
    psi = np.zeros((nz, nr)) if psi0 is None else _c(psi0).copy()
    j = np.zeros((nz, nr))
    fd = _d()
    its = (_i * max_outer)()
    last = PicardResult()
    outer = lib().orc_free_boundary_solve(C.byref(p), psi, j, len(coils_r), _c(coils_r), _c(coils_z), _c(coils_i),
                                          _c(turns), max_outer, tol, C.byref(fd), its, C.byref(last))
    return {"psi": psi, "j_phi": j, "outer_iterations": outer, "final_diff": fd.value,
            "picard_iterations": list(its)[:outer], "converged": bool(last.converged)}


def gs_solve_np(R_min, R_max, Z_min, Z_max, NR, NZ, Ip_target, mu0=4e-7 * np.pi, n_picard=80, n_jacobi=200,
                alpha=0.1, omega_j=0.667, beta_mix=0.5):
    out = np.empty((NZ, NR))
    lib().orc_gs_solve_np(R_min, R_max, Z_min, Z_max, NR, NZ, Ip_target, mu0, n_picard, n_jacobi, alpha, omega_j,
                          beta_mix, out)
    return out


def np_delta_star(psi, R_min, R_max, Z_min, Z_max):
    """gs_delta_star_np, core/jax_gs_solver.py:196-217."""
    psi = _c(psi)
    out = np.empty_like(psi)
    lib().orc_np_delta_star(psi, psi.shape[0], psi.shape[1], R_min, R_max, Z_min, Z_max, out)
    return out


def np_equation_residual(psi, R_min, R_max, Z_min, Z_max, Ip_target, mu0=4e-7 * np.pi, beta_mix=0.5):
    """gs_equation_residual_np, core/jax_gs_solver.py:251-316."""
    psi = _c(psi)
    out = np.empty(3)
    lib().orc_np_equation_residual(psi, psi.shape[0], psi.shape[1], R_min, R_max, Z_min, Z_max, Ip_target, mu0, beta_mix, out)
    return {"abs_max": float(out[0]), "relative_max": float(out[1]), "relative_l2": float(out[2])}


def fastsolver_sweeps(psi_state, j, nr, nz, r_min, r_max, z_min, z_max, boundary_value, omega, iterations, tol=-1.0):
    psi_state = _c(psi_state).copy()
    done = _i()
    last = lib().orc_fastsolver_sweeps(psi_state, _c(j), nr, nz, r_min, r_max, z_min, z_max, boundary_value, omega,
                                       iterations, tol, C.byref(done))
    return psi_state, last, done.value


class RefFastSolver:
    """The reference's own compiled hpc/solver.cpp (oracle/_ref), for validating the restatement."""

    def __init__(self, nr, nz, r_min, r_max, z_min, z_max):
        if not _REF_PATH.exists():
            build()
        L = C.CDLL(str(_REF_PATH))
        L.create_solver.restype = C.c_void_p
        L.create_solver.argtypes = [_i, _i, _d, _d, _d, _d]
        L.run_step.argtypes = [C.c_void_p, _dp, _dp, _i, _i]
        L.run_step_converged.restype = _i
        L.run_step_converged.argtypes = [C.c_void_p, _dp, _dp, _i, _i, _d, _d, C.POINTER(_d)]
        L.set_boundary_dirichlet.argtypes = [C.c_void_p, _d]
        L.destroy_solver.argtypes = [C.c_void_p]
        self.L, self.nr, self.nz = L, nr, nz
        self.h = L.create_solver(nr, nz, r_min, r_max, z_min, z_max)

    def set_boundary(self, v):
        self.L.set_boundary_dirichlet(self.h, v)

    def run_step(self, j, iterations):
        out = np.zeros((self.nz, self.nr))
        self.L.run_step(self.h, _c(j), out, out.size, iterations)
        return out

    def run_step_converged(self, j, max_it, omega, tol):
        out = np.zeros((self.nz, self.nr))
        fd = _d()
        n = self.L.run_step_converged(self.h, _c(j), out, out.size, max_it, omega, tol, C.byref(fd))
        return out, n, fd.value

    def close(self):
        if self.h:
            self.L.destroy_solver(self.h)
            self.h = None

    @staticmethod
    def available():
        return _REF_PATH.exists()

"""Strict-parity gate metrics of a free-boundary equilibrium against a same-case reference flux map.

Counterpart of the reference's FreeGS strict-parity harness (validation/benchmark_freegs_public_example_reconstruction.py,
evaluated by tests/test_benchmark_free_boundary_strict_parity.py:95-125):

* ``native_profile_source_reconstruction`` -- the harness's NATIVE LANE (:275-315): the fixed-boundary Grad-Shafranov solve of a
  supplied toroidal current density inside a supplied Dirichlet ring on the external code's own grid.  The reference factorises the
  5-point matrix with SuperLU; here the same linear system is relaxed to round-off by the device multigrid solver
  (``scpn_multigrid_solve``: its stencil coefficients 1/dR^2 -+ 1/(2 R dR), 1/dZ^2 are the matrix rows of :298-303).
* ``same_case_profile_source_comparison`` -- the comparison record of :431-497 with its FIVE gated metrics: normalised-flux RMSE,
  magnetic-axis distance, boundary flux error, current closure of the supplied current density, and normalised-flux error at the
  X-point constraints, each with the reference's threshold.
* ``strict_parity_metrics`` -- the same metrics for two (nz, nr) flux maps of this package's own free-boundary solver.

The metric arithmetic is host-side NumPy; every flux map comes from the device (this module's native lane,
``EnsembleSolver.solve_free_boundary``) or from a fixture.
"""
from __future__ import annotations

import numpy as np

from . import multigrid_solve as _mg

# thresholds of the reference gate (benchmark_freegs_public_example_reconstruction.py:491-497)
GATE_THRESHOLDS = {"psi_n_rmse": 5.0e-2, "axis_error_m": 2.5e-2, "boundary_max_abs_error_wb": 1.0e-10,
                   "current_closure_relative_error": 5.0e-2, "xpoint_psi_n_error": 5.0e-2}
MU0 = 4.0e-7 * np.pi  # :65


def normalised_flux(psi: np.ndarray) -> np.ndarray:
    """(psi - min) / (max - min), span floored at 1e-30 (:318-321)."""
    lo = float(np.min(psi))
    return (psi - lo) / max(float(np.max(psi)) - lo, 1.0e-30)


def axis_location(r_axis: np.ndarray, z_axis: np.ndarray, psi: np.ndarray) -> tuple[float, float]:
    """(R, Z) of the grid maximum of psi[iz, ir], NaNs ignored (:324-330)."""
    iz, ir = np.unravel_index(int(np.nanargmax(psi)), psi.shape)
    return float(r_axis[ir]), float(z_axis[iz])


def ring_values(field: np.ndarray) -> np.ndarray:
    """The computational-boundary values of a field, every node once."""
    return np.concatenate([field[0, :], field[-1, :], field[1:-1, 0], field[1:-1, -1]])


def native_profile_source_reconstruction(r_axis, z_axis, boundary_psi, jtor, *, rel_tol: float = 1.0e-13, max_cycles: int = 40):
    """Native fixed-boundary Grad-Shafranov solve of a supplied current density on an external code's grid (:275-315).

    Arguments as the reference's ``_native_profile_source_reconstruction``: ``boundary_psi`` and ``jtor`` are
    ``(len(R), len(Z))`` arrays (R slow -- the external code's layout); only the ring of ``boundary_psi`` is used (the Dirichlet
    data, :305-309).  A leading batch axis on both solves several maps in one device call.  The linear system of :288-303
    is solved on the device by V-cycles of the same 5-point operator until the L-infinity residual is below
    ``rel_tol * max|mu0 R J|`` (round-off level; the reference's sparse LU leaves 1e-13 relative); a stall above
    ``1e-10 * max|mu0 R J|`` raises ``RuntimeError``.  Same ``ValueError``s as the reference.
    """
    r_axis, z_axis = np.asarray(r_axis, dtype=np.float64), np.asarray(z_axis, dtype=np.float64)
    boundary_psi, jtor = np.asarray(boundary_psi, dtype=np.float64), np.asarray(jtor, dtype=np.float64)
    steps = {"R": np.diff(r_axis), "Z": np.diff(z_axis)}

    def uniform(d):  # the device operator takes ONE spacing per direction (axis[1] - axis[0]); the reference's matrix uses mean(diff)
        return d.size > 0 and bool(np.all(np.abs(d - d[0]) <= 1.0e-9 * np.abs(d[0])))

    # (condition that must hold, the reference's message when it does not) -- evaluated in the reference's order (:280-289)
    requirements = (
        (boundary_psi.shape == jtor.shape, "boundary_psi and jtor must have matching shapes"),
        (boundary_psi.ndim in (2, 3) and boundary_psi.shape[-2:] == (r_axis.size, z_axis.size), "boundary_psi must have shape len(R) x len(Z)"),
        (min(r_axis.size, z_axis.size) >= 3, "native reconstruction requires at least 3x3 grid"),
        (bool(np.isfinite(boundary_psi).all() and np.isfinite(jtor).all()), "native reconstruction inputs must be finite"),
        (all(bool((d > 0.0).all()) for d in steps.values()), "native reconstruction axes must be strictly increasing"),
        # this lane only (the reference accepts any increasing axes and builds its matrix from the mean spacing): the device stencil is
        # that of a uniform grid, so non-uniform axes would silently solve a different system
        (all(uniform(d) for d in steps.values()), "native reconstruction on the device requires uniformly spaced axes"),
    )
    for holds, message in requirements:
        if not holds:
            raise ValueError(message)
    # device layout is (nz, nr): transpose the R-slow maps; right-hand side -mu0 R J (:305), ring from boundary_psi (:306-309)
    source = np.ascontiguousarray(np.swapaxes(-MU0 * r_axis[:, np.newaxis] * jtor, -1, -2))
    ring = np.swapaxes(boundary_psi, -1, -2)
    start = np.zeros_like(source)
    start[..., 0, :], start[..., -1, :], start[..., :, 0], start[..., :, -1] = ring[..., 0, :], ring[..., -1, :], ring[..., :, 0], ring[..., :, -1]
    scale = max(float(np.max(np.abs(source))), 1.0e-300)
    psi, residual, _, _ = _mg.multigrid_solve_on_axes(source, start, r_axis, z_axis, tol=rel_tol * scale, max_cycles=max_cycles)
    if float(np.max(residual)) > 1.0e-10 * scale:
        raise RuntimeError(f"native reconstruction did not reach round-off: residual {float(np.max(residual)):.3e} for |source| {scale:.3e}")
    return np.ascontiguousarray(np.swapaxes(psi, -1, -2))


def integrate_current(r_axis, z_axis, jtor) -> float:
    """Trapezoid integral of an R-slow (len(R), len(Z)) current density over Z then R (:360-364, scipy ``trapezoid``)."""
    r_axis, z_axis, jtor = np.asarray(r_axis, dtype=np.float64), np.asarray(z_axis, dtype=np.float64), np.asarray(jtor, dtype=np.float64)
    over_z = np.sum(np.diff(z_axis) * (jtor[:, 1:] + jtor[:, :-1]) / 2.0, axis=1)
    return float(np.sum(np.diff(r_axis) * (over_z[1:] + over_z[:-1]) / 2.0))


def same_case_profile_source_comparison(r_axis, z_axis, external_psi, plasma_psi, jtor, plasma_current_a, xpoints=(), *, native_plasma_psi=None) -> dict:
    """The record of the reference's ``_native_same_case_profile_source_comparison`` (:431-497) from plain arrays.

    ``external_psi`` / ``plasma_psi`` / ``jtor`` are the external code's total flux, plasma flux and current density, R slow
    (what the harness reads off a FreeGS equilibrium: ``eq.psi()``, ``eq.plasma_psi``, ``eq.Jtor``).  The native plasma flux is
    solved on the device (``native_profile_source_reconstruction``) unless ``native_plasma_psi`` hands one in.  Keys and
    thresholds are the reference's (the FreeGS-only ``q_profile_sanity`` entry is absent); ``checks`` / ``pass`` add the verdict
    of the five thresholds.
    """
    r_axis, z_axis = np.asarray(r_axis, dtype=np.float64), np.asarray(z_axis, dtype=np.float64)
    external_psi, plasma_psi, jtor = (np.asarray(a, dtype=np.float64) for a in (external_psi, plasma_psi, jtor))
    if plasma_psi.shape != external_psi.shape:
        raise ValueError("plasma_psi and external_psi must have matching shapes")
    vacuum_psi = external_psi - plasma_psi
    if native_plasma_psi is None:
        native_plasma_psi = native_profile_source_reconstruction(r_axis, z_axis, plasma_psi, jtor)
    native_psi = native_plasma_psi + vacuum_psi
    # the helpers below work on (nz, nr) maps: hand them the transposed views
    m = strict_parity_metrics(native_psi.T, external_psi.T, r_axis, z_axis, x_points=xpoints, jtor=jtor.T, plasma_current_a=plasma_current_a)
    ax_n, ax_e = axis_location(r_axis, z_axis, native_psi.T), axis_location(r_axis, z_axis, external_psi.T)
    return {
        "accepted_full_fidelity": False,
        "axis_error_m": m["axis_error_m"],
        "boundary_containment_fraction": m["boundary_containment_fraction"],
        "boundary_max_abs_error_wb": m["boundary_max_abs_error_wb"],
        "current_closure_relative_error": m["current_closure_relative_error"],
        "external_axis_r_m": ax_e[0], "external_axis_z_m": ax_e[1],
        "finite_external_psi": bool(external_psi.size and np.all(np.isfinite(external_psi))),
        "finite_native_psi": bool(native_psi.size and np.all(np.isfinite(native_psi))),
        "native_axis_r_m": ax_n[0], "native_axis_z_m": ax_n[1],
        "native_current_a": m["native_current_a"],
        "native_plasma_psi_rmse": float(np.sqrt(np.mean((native_plasma_psi - plasma_psi) * (native_plasma_psi - plasma_psi)))),
        "psi_n_rmse": m["psi_n_rmse"],
        "schema": "native-freegs-profile-source-comparison.v1",
        "thresholds": dict(GATE_THRESHOLDS),
        "xpoint_constraint_count": len(tuple(xpoints)),
        "xpoint_psi_n_error_max": m["xpoint_psi_n_error_max"],
        "checks": m["checks"],
        "pass": m["pass"],
    }


def strict_parity_metrics(psi: np.ndarray, psi_ref: np.ndarray, r_axis, z_axis, x_points=(), *, jtor=None, plasma_current_a=None) -> dict:
    """Gate metrics of ``psi`` against ``psi_ref`` (both (nz, nr)) and the pass flag of each threshold.

    ``jtor`` ((nz, nr)) with ``plasma_current_a`` adds the fifth gated metric, the current closure
    |trapezoid integral of J_tor - I_p| / max(|I_p|, 1) (:457-461); without them the record says so (``checks`` then holds
    four entries and ``current_closure_relative_error`` is None)."""
    psi, psi_ref = np.asarray(psi, dtype=np.float64), np.asarray(psi_ref, dtype=np.float64)
    if psi.shape != psi_ref.shape or psi.shape != (len(z_axis), len(r_axis)):
        raise ValueError("psi, psi_ref and the axes must describe the same (nz, nr) grid")
    r_axis, z_axis = np.asarray(r_axis, dtype=np.float64), np.asarray(z_axis, dtype=np.float64)
    pn, rn = normalised_flux(psi), normalised_flux(psi_ref)
    ax, ax_ref = axis_location(r_axis, z_axis, psi), axis_location(r_axis, z_axis, psi_ref)
    x_err = 0.0
    for r_x, z_x in x_points:  # nearest grid node, as the reference samples its X-point constraints (:370-378)
        ir, iz = int(np.argmin(np.abs(r_axis - float(r_x)))), int(np.argmin(np.abs(z_axis - float(z_x))))
        x_err = max(x_err, abs(float(pn[iz, ir]) - float(rn[iz, ir])))
    out = {
        "psi_n_rmse": float(np.sqrt(np.mean((pn - rn) ** 2))),
        "psi_rmse": float(np.sqrt(np.mean((psi - psi_ref) ** 2))),
        "axis_error_m": float(np.hypot(ax[0] - ax_ref[0], ax[1] - ax_ref[1])),
        "boundary_max_abs_error_wb": float(np.max(np.abs(ring_values(psi - psi_ref)))),
        "boundary_containment_fraction": float(np.mean(np.abs(ring_values(psi - psi_ref)) <= GATE_THRESHOLDS["boundary_max_abs_error_wb"])),
        "xpoint_psi_n_error_max": x_err,
        "finite": bool(np.all(np.isfinite(psi))),
    }
    out["current_closure_relative_error"], out["native_current_a"] = None, None
    if jtor is not None:
        if plasma_current_a is None:
            raise ValueError("jtor needs plasma_current_a for the current-closure metric")
        jt = np.asarray(jtor, dtype=np.float64)
        if jt.shape != psi.shape:
            raise ValueError("jtor must have the shape of psi")
        out["native_current_a"] = integrate_current(r_axis, z_axis, jt.T)
        out["current_closure_relative_error"] = abs(out["native_current_a"] - float(plasma_current_a)) / max(abs(float(plasma_current_a)), 1.0)
    out["checks"] = {"psi_n_rmse": out["psi_n_rmse"] <= GATE_THRESHOLDS["psi_n_rmse"],
                     "axis_error_m": out["axis_error_m"] <= GATE_THRESHOLDS["axis_error_m"],
                     "boundary_max_abs_error_wb": out["boundary_max_abs_error_wb"] <= GATE_THRESHOLDS["boundary_max_abs_error_wb"],
                     "xpoint_psi_n_error": x_err <= GATE_THRESHOLDS["xpoint_psi_n_error"]}
    if out["current_closure_relative_error"] is not None:
        out["checks"]["current_closure_relative_error"] = out["current_closure_relative_error"] <= GATE_THRESHOLDS["current_closure_relative_error"]
    out["pass"] = bool(out["finite"] and all(out["checks"].values()))
    return out

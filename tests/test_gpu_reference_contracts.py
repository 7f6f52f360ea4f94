"""Behavioural contracts the reference's own tests pin for the path, re-stated against the HIP implementation (fresh code,
same scenarios; the reference test each one mirrors is cited).  These complement the golden-vector parity tests with the
edge cases: non-finite inputs, overflow, boundary enforcement, residual gate, argument validation, duck-typed kernels."""
import json
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import scpn_fusion_core_amd as pkg
    from scpn_fusion_core_amd import _lib

    _lib.require_gpu()
    return pkg


BASE = {  # tests/test_gs_residual_gate.py BASE_CONFIG / tests/test_fusion_kernel_fail_on_diverge.py _write_cfg shape
    "reactor_name": "contract", "grid_resolution": [20, 20],
    "dimensions": {"R_min": 4.0, "R_max": 8.0, "Z_min": -4.0, "Z_max": 4.0},
    "physics": {"plasma_current_target": 15.0, "vacuum_permeability": 1.0},
    "coils": [{"name": "PF1", "r": 3.0, "z": 7.6, "current": 8.0}, {"name": "PF2", "r": 8.2, "z": 6.7, "current": -1.32},
              {"name": "CS", "r": 1.7, "z": 0.0, "current": 0.15}],
    "solver": {"max_iterations": 30, "convergence_threshold": 1e-4, "relaxation_factor": 0.1, "solver_method": "sor"},
}


def _kernel(tmp_path, name="cfg.json", **solver):
    from scpn_fusion_core_amd.fusion_kernel import FusionKernel

    cfg = json.loads(json.dumps(BASE))
    cfg["solver"].update(solver)
    p = tmp_path / name
    p.write_text(json.dumps(cfg))
    return FusionKernel(str(p))


def _boundary_map(shape):
    b = np.zeros(shape)
    b[0, :], b[-1, :], b[:, 0], b[:, -1] = 10.0, 20.0, 30.0, 40.0
    return b


def _assert_ring(field, ref):
    for sl in ((0, slice(None)), (-1, slice(None)), (slice(None), 0), (slice(None), -1)):
        np.testing.assert_array_equal(field[sl], ref[sl])


# ---- tests/test_fusion_kernel_iterative_solver_contracts.py ---------------------------------------------
def test_jacobi_step_sanitizes_inputs_and_preserves_boundaries(hip, tmp_path):  # :90-102
    k = _kernel(tmp_path)
    psi = _boundary_map(k.Psi.shape)
    psi[2, 2] = np.nan
    source = np.ones_like(psi)
    source[1, 1] = np.inf
    out = k._jacobi_step(psi, source)
    _assert_ring(out, psi)
    assert np.all(np.isfinite(out)) and out.shape == psi.shape


def test_sor_step_sanitizes_non_finite_inputs(hip, tmp_path):  # tests/test_fusion_kernel_fail_on_diverge.py:138-150
    k = _kernel(tmp_path)
    psi = np.zeros_like(k.Psi)
    psi[3, 3], psi[4, 4] = np.nan, np.inf
    source = np.ones_like(psi)
    source[5, 5] = -np.inf
    out = k._sor_step(psi, source, omega=1.6)
    assert np.all(np.isfinite(out))


def test_restrict_prolong_wrappers_preserve_shape(hip):  # :105-115
    from scpn_fusion_core_amd import multigrid_solve as mgs

    fine = np.arange(81, dtype=np.float64).reshape(9, 9)
    coarse = mgs.restrict_full_weight(fine)
    back = mgs.prolongate_bilinear(coarse, 9, 9)
    assert coarse.shape == (5, 5) and back.shape == fine.shape and np.all(np.isfinite(back))


@pytest.mark.parametrize("method", ["jacobi", "multigrid", "sor", "anderson"])
def test_elliptic_solve_methods_enforce_boundary(hip, tmp_path, method):  # :213-224
    k = _kernel(tmp_path, solver_method=method, sor_omega=1.1, max_iterations=3)
    boundary = _boundary_map(k.Psi.shape)
    res = k.solve_equilibrium(boundary_flux=boundary)
    _assert_ring(k.Psi, boundary)
    assert np.all(np.isfinite(k.Psi)) and res["iterations"] == 3


def test_boundary_flux_shape_mismatch_rejected(hip, tmp_path):  # :253-258
    k = _kernel(tmp_path)
    with pytest.raises(ValueError, match="boundary_flux shape"):
        k.solve_equilibrium(boundary_flux=np.zeros((3, 3)))


def test_seed_normalises_gaussian_current(hip):  # :240-250 -- observed through a zero-iteration solve of the resident ensemble
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = json.loads(json.dumps(BASE))
    cfg["solver"]["max_iterations"] = 0
    with EnsembleSolver(cfg, batch=2, current_targets=np.array([2.5, 0.0])) as es:
        es.solve_equilibrium()
        j = es.get_jphi()
        assert float(j[0].sum()) * es.dR * es.dZ == pytest.approx(2.5)
        assert np.array_equal(j[1], np.zeros_like(j[1]))  # :227-237 zero target clears the current
        assert np.all(np.isfinite(es.get_psi()))


# ---- tests/test_fusion_kernel_fail_on_diverge.py ----------------------------------------------------------
def test_residual_rms_avoids_overflow_for_extreme_fields(hip, tmp_path):  # :75-92
    k = _kernel(tmp_path)
    psi = np.zeros_like(k.Psi)
    inner = psi[1:-1, 1:-1]
    psi[1:-1, 1:-1] = np.where((np.indices(inner.shape).sum(axis=0) % 2) == 0, 1e300, -1e300)
    k.Psi = psi
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        rms = k._compute_gs_residual_rms(np.zeros_like(psi))
    assert np.isfinite(rms) and rms > 0.0


def test_find_x_point_extreme_and_nonfinite_fields(hip, tmp_path):  # :95-135
    k = _kernel(tmp_path)
    psi = np.outer(np.linspace(-1e300, 1e300, k.NZ), np.ones(k.NR))
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        (rx, zx), px = k.find_x_point(psi)
    assert np.isfinite(rx) and np.isfinite(zx) and np.isfinite(px)
    bad = np.full_like(k.Psi, np.nan)
    bad[1, 1], bad[2, 2] = np.inf, -np.inf
    (rx, zx), px = k.find_x_point(bad)
    assert np.isfinite(rx) and np.isfinite(zx) and np.isfinite(px)
    k.cfg["solver"]["xpoint_use_saddle_detection"] = True  # :109-119 prefers the saddle
    r0, z0 = float(k.R[len(k.R) // 2]), float(k.Z[int(0.2 * (len(k.Z) - 1))])
    (rx, zx), _ = k.find_x_point((k.RR - r0) ** 2 - (k.ZZ - z0) ** 2)
    assert abs(rx - r0) <= k.dR and abs(zx - z0) <= k.dZ


# ---- tests/test_gs_residual_gate.py ---------------------------------------------------------------------------
def test_solve_reports_gs_residual_metrics(hip, tmp_path):  # :47-63
    res = _kernel(tmp_path).solve_equilibrium()
    hist = res["gs_residual_history"]
    assert len(hist) == len(res["residual_history"]) >= 1
    assert np.isfinite(res["gs_residual"]) and np.isfinite(res["gs_residual_best"])
    assert all(np.isfinite(x) and x >= 0.0 for x in hist)
    assert res["gs_residual"] == pytest.approx(hist[-1]) and res["gs_residual_best"] <= res["gs_residual"] + 1e-12


def test_gs_residual_gate_blocks_update_only_convergence(hip, tmp_path):  # :66-87
    loose = _kernel(tmp_path, "loose.json", max_iterations=5, convergence_threshold=1e6).solve_equilibrium()
    assert loose["converged"] and loose["iterations"] == 1
    gate = _kernel(tmp_path, "gate.json", max_iterations=5, convergence_threshold=1e6, require_gs_residual=True,
                   gs_residual_threshold=1e-12).solve_equilibrium()
    assert not gate["converged"] and gate["iterations"] == 5 and gate["gs_residual"] > 1e-12


def test_gs_residual_gate_rejects_non_positive_threshold(hip, tmp_path):  # :90-98
    k = _kernel(tmp_path, require_gs_residual=True, gs_residual_threshold=0.0)
    with pytest.raises(ValueError, match="gs_residual_threshold"):
        k.solve_equilibrium()


# ---- tests/test_fusion_kernel_free_boundary.py ---------------------------------------------------------------
class _KernelStub:  # the duck-typed kernel of :27-40: axes, Psi and a solve_equilibrium that only counts calls
    def __init__(self):
        self.R, self.Z = np.linspace(5.0, 6.0, 5), np.linspace(-0.5, 0.5, 5)
        self.NR, self.NZ = 5, 5
        self.dR, self.dZ = float(self.R[1] - self.R[0]), float(self.Z[1] - self.Z[0])
        self.Psi = np.zeros((5, 5))
        self.solve_calls = 0

    def solve_equilibrium(self, **kwargs):
        self.solve_calls += 1
        return {"converged": True, "kwargs": kwargs}


def test_free_boundary_helpers_contract(hip):
    from scpn_fusion_core_amd import free_boundary as fb

    k = _KernelStub()
    assert np.isfinite(fb.green_function(5.2, 0.1, 5.5, -0.2))  # :43-45
    k.Psi[0, 0], k.Psi[0, 1], k.Psi[1, 0], k.Psi[1, 1] = 0.0, 1.0, 2.0, 3.0  # :48-56
    assert abs(fb.interp_psi(k, float(k.R[0] + 0.5 * k.dR), float(k.Z[0] + 0.5 * k.dZ)) - 1.5) < 1e-12
    assert abs(fb.sample_flux_at_points(k, [[k.R[0] + 0.5 * k.dR, k.Z[0] + 0.5 * k.dZ]])[0] - 1.5) < 1e-12
    coils = fb.CoilSet(positions=[(5.2, 0.0)], currents=np.array([1.0]), target_flux_points=np.array([[5.2, 0.0], [5.4, 0.1]]),
                       target_flux_values=np.array([0.1, 0.2]))
    np.testing.assert_allclose(fb.resolve_shape_target_flux(k, coils), [0.1, 0.2])  # :59-68
    coils = fb.CoilSet(positions=[(5.3, 0.0)], currents=np.array([0.0]), turns=[10], current_limits=np.array([2.0]),
                       target_flux_points=np.array([[5.35, 0.0]]))
    sol = fb.optimize_coil_currents(k, coils, target_flux=np.array([0.02]), tikhonov_alpha=1e-4)  # :71-87
    assert sol.shape == (1,) and np.all(np.isfinite(sol))


def test_inverse_probe_reconstruction_contract(hip):
    from scpn_fusion_core_amd import free_boundary as fb

    k = _KernelStub()
    coils = fb.CoilSet(positions=[(5.1, -0.35), (5.9, 0.35)], currents=np.array([0.0, 0.0]), turns=[6, 8],
                       current_limits=np.array([2.0, 2.0]))
    fp, bp, dirs = np.array([[5.25, -0.15], [5.75, 0.15]]), np.array([[5.35, 0.0], [5.65, 0.0]]), ["R", "Z"]
    true = np.array([0.55, -0.35])
    resp = fb.build_magnetic_probe_response_matrix(k, coils, flux_points=fp, b_probe_points=bp, b_probe_directions=dirs)
    meas = resp @ true
    r = fb.reconstruct_coil_currents_from_magnetic_probes(k, coils, flux_points=fp, flux_measurements=meas[:2], b_probe_points=bp,
                                                          b_probe_directions=dirs, b_probe_measurements=meas[2:], tikhonov_alpha=0.0)
    np.testing.assert_allclose(r["coil_currents"], true, atol=1e-9)  # :90-124
    assert r["residual_rms"] < 1e-12 and r["response_rank"] == 2
    lim = fb.CoilSet(positions=[(5.1, -0.35)], currents=np.array([0.0]), current_limits=np.array([0.2]))  # :182-199
    r = fb.reconstruct_coil_currents_from_magnetic_probes(k, lim, flux_points=np.array([[5.25, -0.15]]),
                                                          flux_measurements=np.array([1.0]), tikhonov_alpha=0.0)
    assert np.all(np.abs(r["coil_currents"]) <= lim.current_limits + 1e-12) and r["active_bounds"] >= 1
    bare = fb.CoilSet(positions=[(5.1, -0.35)], currents=np.array([0.0]))  # :202-224
    with pytest.raises(ValueError):
        fb.reconstruct_coil_currents_from_magnetic_probes(k, bare, flux_points=np.array([[5.25, -0.15]]),
                                                          flux_measurements=np.array([0.0, 1.0]))
    with pytest.raises(ValueError):
        fb.reconstruct_coil_currents_from_magnetic_probes(k, bare, b_probe_points=np.array([[5.25, -0.15]]),
                                                          b_probe_directions=["phi"], b_probe_measurements=np.array([0.0]))


def test_boundary_flux_reconstruction_contract(hip):
    from scpn_fusion_core_amd import free_boundary as fb

    k = _KernelStub()
    coils = fb.CoilSet(positions=[(5.1, -0.35), (5.9, 0.35)], currents=np.array([0.55, -0.35]), turns=[6, 8])
    pts = np.array([[5.15, -0.45], [5.45, -0.48], [5.85, 0.42], [5.55, 0.47]])
    target = fb.build_mutual_inductance_matrix(k, coils, pts).T @ coils.currents
    r = fb.reconstruct_boundary_flux_from_coils(k, coils, boundary_points=pts, target_flux=target)  # :127-148
    np.testing.assert_allclose(r["reconstructed_flux"], target, atol=1e-13)
    assert r["rmse"] < 1e-13 and r["max_abs_error"] < 1e-13 and r["response_rank"] == 2 and r["point_count"] == 4
    lim = np.array([[5.0, -0.5], [6.0, -0.5], [6.0, 0.5], [5.0, 0.5]])  # :151-179
    r = fb.reconstruct_boundary_flux_from_coils(k, coils, boundary_points=np.array([[5.2, -0.4], [5.8, 0.4]]), limiter_points=lim,
                                                axis_point=np.array([5.5, 0.0]), x_points=np.array([[5.25, -0.45], [5.75, 0.45]]))
    assert r["limiter_point_count"] == 4 and r["x_point_count"] == 2 and r["axis_flux"] is not None
    assert r["limiter_flux"].shape == (4,) and r["x_point_flux"].shape == (2,) and r["min_limiter_distance_m"] > 0.0
    assert np.all(np.isfinite(r["limiter_flux"])) and np.all(np.isfinite(r["x_point_flux"]))


def test_solve_free_boundary_returns_contract_for_duck_typed_kernel(hip):  # :227-255
    from scpn_fusion_core_amd import free_boundary as fb

    k = _KernelStub()
    coils = fb.CoilSet(positions=[(5.2, 0.0)], currents=np.array([0.0]))
    lim = np.array([[5.0, -0.5], [6.0, -0.5], [6.0, 0.5], [5.0, 0.5]])
    r = fb.solve_free_boundary(k, coils, max_outer_iter=1, tol=0.0, limiter_points=lim, axis_point=np.array([5.5, 0.0]),
                               x_points=np.array([[5.25, -0.45]]))
    assert r["outer_iterations"] == 1 and "final_diff" in r and "coil_currents" in r
    assert r["vacuum_boundary_abs_error"] == 0.0
    br = r["boundary_reconstruction"]
    assert br["point_count"] == 4 * (k.NR - 1) and br["coil_count"] == 1 and br["max_abs_error"] == 0.0
    assert br["limiter_point_count"] == 4 and br["x_point_count"] == 1 and br["axis_flux"] is not None
    assert k.solve_calls == 1
    with pytest.raises(ValueError):
        fb.solve_free_boundary(k, coils, max_outer_iter=0)
    with pytest.raises(ValueError):
        fb.solve_free_boundary(k, coils, tol=-1.0)


def test_fusion_kernel_free_boundary_reports_boundary_reconstruction(hip, tmp_path):
    """Device outer loop + the closing Green's-function reconstruction of the ring (core/fusion_kernel_free_boundary.py:708-739):
    the Dirichlet ring the device applied equals the coil flux at the boundary points to rounding."""
    from scpn_fusion_core_amd import free_boundary as fb

    k = _kernel(tmp_path, solver_method="multigrid", sor_omega=1.0, max_iterations=60)
    coils = fb.CoilSet(positions=[(c["r"], c["z"]) for c in BASE["coils"]], currents=np.array([c["current"] for c in BASE["coils"]]) * 1e6)
    r = k.solve_free_boundary(coils, max_outer_iter=2, tol=1e-4)
    br = r["boundary_reconstruction"]
    assert br["point_count"] == 2 * (k.NR - 1) + 2 * (k.NZ - 1) and r["vacuum_boundary_abs_error"] <= 1e-12 * np.abs(br["target_flux"]).max()
    assert r["outer_iterations"] in (1, 2) and np.all(np.isfinite(k.Psi))


# ---- tests/test_multigrid_solve.py ------------------------------------------------------------------------------
def _solovev_problem(nr=65, nz=65):
    r_min, r_max, z_min, z_max = 1.2, 2.2, -0.5, 0.5
    rr, zz = np.meshgrid(np.linspace(r_min, r_max, nr), np.linspace(z_min, z_max, nz))
    return -rr * np.exp(-((rr - 1.7) ** 2 + zz**2) / 0.05), r_min, r_max, z_min, z_max, nr, nz


def test_multigrid_solve_contract(hip):
    from scpn_fusion_core_amd import multigrid_solve as mgs

    source, r_min, r_max, z_min, z_max, nr, nz = _solovev_problem()
    psi, residual, n_cycles, converged = mgs.multigrid_solve(source, np.zeros((nz, nr)), r_min, r_max, z_min, z_max, nr, nz, tol=1e-6,
                                                             max_cycles=200)  # :34-43
    assert converged is True and residual < 1e-6 and 0 < n_cycles < 50 and np.all(np.isfinite(psi))
    rr, _ = np.meshgrid(np.linspace(r_min, r_max, nr), np.linspace(z_min, z_max, nz))
    dr, dz = (r_max - r_min) / (nr - 1), (z_max - z_min) / (nz - 1)
    assert residual == pytest.approx(mgs.residual_linf(psi, source, rr, dr, dz), rel=1e-12)  # :62-75
    bc = np.zeros((nz, nr))
    bc[0, :] = bc[-1, :] = bc[:, 0] = bc[:, -1] = 0.4  # :46-59
    psi, *_ = mgs.multigrid_solve(source, bc, r_min, r_max, z_min, z_max, nr, nz, tol=1e-6, max_cycles=200)
    assert float(psi[0, 0]) == 0.4 and float(psi[-1, -1]) == 0.4
    np.testing.assert_array_equal(psi[0, :], 0.4)
    np.testing.assert_array_equal(psi[:, -1], 0.4)
    for src_shape, psi_shape in (((10, 12), (65, 65)), ((65, 65), (10, 12))):  # :78-83
        with pytest.raises(ValueError, match="shape"):
            mgs.multigrid_solve(np.zeros(src_shape), np.zeros(psi_shape), 1.0, 2.0, -1.0, 1.0, 65, 65)
    for tol in (0.0, -1e-6, np.inf, np.nan):  # :86-90
        with pytest.raises(ValueError, match="tol"):
            mgs.multigrid_solve(np.zeros((33, 33)), np.zeros((33, 33)), 1.0, 9.0, -5.0, 5.0, 33, 33, tol=tol)
    with pytest.raises(ValueError, match="max_cycles"):  # :93-97
        mgs.multigrid_solve(np.zeros((33, 33)), np.zeros((33, 33)), 1.0, 9.0, -5.0, 5.0, 33, 33, max_cycles=0)
    for omega in (0.0, 0.999999, 2.0, np.inf, np.nan):  # :121-123
        with pytest.raises(ValueError, match="omega"):
            mgs.validate_sor_omega(omega)


def test_residual_linf_exact_solution_and_empty_interior(hip):
    from scpn_fusion_core_amd import multigrid_solve as mgs

    r, z = np.linspace(1.0, 2.0, 17), np.linspace(-0.5, 0.5, 17)  # :100-109
    rr, zz = np.meshgrid(r, z)
    dr, dz = float(r[1] - r[0]), float(z[1] - z[0])
    psi = 0.03125 * rr**4 - 0.125 * zz**2 + 0.05 * rr**2 * zz**2
    source = mgs.mg_residual(psi, np.zeros_like(psi), rr, dr, dz)
    assert mgs.residual_linf(psi, source, rr, dr, dz) < 1e-12
    tiny = np.zeros((2, 2))  # :126-130
    assert mgs.residual_linf(tiny, tiny, np.array([[1.0, 2.0], [1.0, 2.0]]), 1.0, 1.0) == 0.0


# ---- tests/test_jax_gs_solver.py TestGsSolveNumpy (the CPU "polyglot" lane) -----------------------------------------
def test_polyglot_lane_properties(hip):
    from scpn_fusion_core_amd.polyglot import gs_solve_hip

    args = (0.1, 2.0, -1.5, 1.5, 17, 17, 1e6)  # :32-38 R_MIN, R_MAX, Z_MIN, Z_MAX, NR, NZ, IP_TARGET
    psi = gs_solve_hip(*args, n_picard=20, n_jacobi=50)  # :44-69
    assert psi.shape == (17, 17) and np.all(np.isfinite(psi)) and np.max(np.abs(psi[1:-1, 1:-1])) > 0.0
    for edge in (psi[0, :], psi[-1, :], psi[:, 0], psi[:, -1]):
        assert np.allclose(edge, 0.0)
    sym = gs_solve_hip(*args, n_picard=40, n_jacobi=100)  # :71-76
    np.testing.assert_allclose(sym, sym[::-1, :], atol=1e-4 * max(1.0, np.abs(sym).max()))
    hi = gs_solve_hip(0.1, 2.0, -1.5, 1.5, 17, 17, 2e6, n_picard=20, n_jacobi=50)  # :78-94
    assert np.max(np.abs(hi)) > np.max(np.abs(psi))
    assert np.all(np.isfinite(gs_solve_hip(0.1, 2.0, -1.5, 1.5, 33, 33, 1e6, n_picard=20, n_jacobi=50)))  # :96-101


# ---- tests/test_jax_gs_solver.py:104-265 (lane diagnostics: equation residual, Delta*, J_phi, total current) --------------
@pytest.mark.gpu
def test_polyglot_equation_residual_contract(hip):
    from scpn_fusion_core_amd.polyglot import gs_equation_residual_hip

    mu0 = 4e-7 * np.pi
    imbalanced = np.zeros((3, 3))
    imbalanced[1, 1] = 1.0
    bal = gs_equation_residual_hip(np.zeros((3, 3)), 1.0, 3.0, -1.0, 1.0, 3, 3, 0.0, mu0, 0.5)
    imb = gs_equation_residual_hip(imbalanced, 1.0, 3.0, -1.0, 1.0, 3, 3, 0.0, mu0, 0.5)
    assert bal["abs_max"] == 0.0
    assert imb["abs_max"] == pytest.approx(4.0)
    assert imb["relative_max"] > 1.0
    with pytest.raises(ValueError, match="psi shape"):
        gs_equation_residual_hip(np.zeros((4, 3)), 1.0, 3.0, -1.0, 1.0, 3, 3, 0.0)
    with pytest.raises(ValueError, match="positive, increasing"):
        gs_equation_residual_hip(np.zeros((3, 3)), 0.0, 3.0, -1.0, 1.0, 3, 3, 0.0)
    bad = np.zeros((3, 3))
    bad[1, 1] = np.nan
    with pytest.raises(ValueError, match="finite"):
        gs_equation_residual_hip(bad, 1.0, 3.0, -1.0, 1.0, 3, 3, 0.0)


@pytest.mark.gpu
def test_polyglot_delta_star_manufactured_fluxes(hip):
    from scpn_fusion_core_amd import polyglot as pg

    mu0 = 4e-7 * np.pi
    box = (1.0, 3.0, -1.0, 1.0)
    # z-quadratic flux: Delta* = 2 c, zero ring, J = -2c/(mu0 R)
    nr, nz, c = 11, 13, -0.25
    r, z = np.linspace(1.0, 3.0, nr), np.linspace(-1.0, 1.0, nz)
    _, zz = np.meshgrid(r, z)
    psi = c * zz**2
    ds, jd = pg.gs_delta_star_hip(psi, *box), pg.gs_toroidal_current_density_hip(psi, *box)
    np.testing.assert_allclose(ds[1:-1, 1:-1], 2.0 * c, rtol=1e-12, atol=1e-12)
    for edge in (ds[0, :], ds[-1, :], ds[:, 0], ds[:, -1]):
        np.testing.assert_allclose(edge, 0.0, atol=0.0)
    np.testing.assert_allclose(jd[1:-1, 1:-1], np.broadcast_to(-2.0 * c / (mu0 * r[None, 1:-1]), jd[1:-1, 1:-1].shape), rtol=1e-12, atol=1e-6)
    # total current closure
    nr, nz, c = 17, 19, -0.125
    r, z = np.linspace(1.0, 3.0, nr), np.linspace(-1.0, 1.0, nz)
    _, zz = np.meshgrid(r, z)
    dr, dz = 2.0 / (nr - 1), 2.0 / (nz - 1)
    expected = float(np.sum(-2.0 * c / (mu0 * r[None, 1:-1])) * (nz - 2) * dr * dz)
    assert pg.gs_total_toroidal_current_hip(c * zz**2, *box) == pytest.approx(expected, rel=1e-12)
    # radial quartic and mixed Solov'ev fluxes
    nr, nz = 15, 17
    r, z = np.linspace(1.0, 3.0, nr), np.linspace(-1.0, 1.0, nz)
    rr, zz = np.meshgrid(r, z)
    dr = 2.0 / (nr - 1)
    a, b = 0.03125, -0.125
    ds = pg.gs_delta_star_hip(a * rr**4 + b * zz**2, *box)
    np.testing.assert_allclose(ds[1:-1, 1:-1], 8.0 * a * rr[1:-1, 1:-1] ** 2 + 2.0 * b - 2.0 * a * dr**2, rtol=1e-12, atol=1e-12)
    m = 0.05
    psi = m * rr**2 * zz**2 + b * zz**2
    ds, jd = pg.gs_delta_star_hip(psi, *box), pg.gs_toroidal_current_density_hip(psi, *box)
    exp = 2.0 * m * rr[1:-1, 1:-1] ** 2 + 2.0 * b
    np.testing.assert_allclose(ds[1:-1, 1:-1], exp, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(jd[1:-1, 1:-1], -exp / (mu0 * rr[1:-1, 1:-1]), rtol=1e-12, atol=1e-6)
    # validation contract (core/jax_gs_solver.py:168-193)
    with pytest.raises(ValueError, match="two-dimensional"):
        pg.gs_delta_star_hip(np.zeros(5), *box)
    with pytest.raises(ValueError, match="at least 3x3"):
        pg.gs_delta_star_hip(np.zeros((2, 5)), *box)
    with pytest.raises(ValueError, match="R_min and R_max"):
        pg.gs_delta_star_hip(np.zeros((5, 5)), 0.0, 3.0, -1.0, 1.0)
    with pytest.raises(ValueError, match="mu0"):
        pg.gs_toroidal_current_density_hip(np.zeros((5, 5)), *box, mu0=0.0)

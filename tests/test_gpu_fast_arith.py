"""Contracted arithmetic (``scpn_gs_set_arith(handle, 1)``, include/scpn_hip.h) against the CPU ORACLE -- never against the exact
device build -- on the BASELINE configurations the metric is quoted on.

The contracted kernels run the same red-black Gauss-Seidel / V-cycle / Picard iteration as the reference
(core/multigrid_solve.py:184-206,211-249, core/fusion_kernel_newton_solver.py:390-615) with fused multiply-adds and
reciprocal multiplies, so they are not bit-identical to it.  Bars (stated here, asserted below):

* one V-cycle (the linear operator):          |psi - psi_oracle| <= 1e-11 * max|psi|
* converged Picard / free-boundary solves:     |psi - psi_oracle| <= 1e-10 * max|psi|  (north_star: "stated psi tolerance"),
                                               Picard iteration counts equal or +-1 (the difference is reported in the
                                               assertion message), outer iteration counts equal
* the five-threshold strict-parity gate of the contracted flux map against the oracle's passes.
"""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests._problems import bench_problem, configs_from, iter_like_cfg  # noqa: E402

PSI_TOL = 1e-10  # converged equilibria, relative to max|psi_oracle|
OP_TOL = 1e-11   # one V-cycle


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def ES():
    from scpn_fusion_core_amd import _lib
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    _lib.require_gpu()
    return EnsembleSolver


def _counts_close(got, want):
    got, want = list(np.atleast_1d(got)), list(np.atleast_1d(want))
    return len(got) == len(want) and all(abs(int(a) - int(b)) <= 1 for a, b in zip(got, want))


def test_arithmetic_switch_api(ES, monkeypatch):
    from scpn_fusion_core_amd import _lib

    monkeypatch.delenv("SCPN_HIP_ARITH", raising=False)
    cfg = iter_like_cfg(33)
    with ES(cfg, batch=2) as es:
        assert es.arithmetic == "exact"  # the default keeps the reference's operation order
        es.arithmetic = "contracted"
        assert es.arithmetic == "contracted"
        with pytest.raises(ValueError):
            es.arithmetic = "fast-and-loose"
        assert _lib.lib().scpn_gs_set_arith(es._h, 7) == _lib.SCPN_E_ARG
        assert es.arithmetic == "contracted"
    with ES({**cfg, "solver": dict(cfg["solver"], arithmetic="contracted")}, batch=1) as es:
        assert es.arithmetic == "contracted"  # the config key
    monkeypatch.setenv("SCPN_HIP_ARITH", "contracted")
    with ES(cfg, batch=1) as es:
        assert es.arithmetic == "contracted"  # the library default of new handles
    with ES(cfg, batch=1, arithmetic="exact") as es:
        assert es.arithmetic == "exact"


@pytest.mark.parametrize("shape,batch", [((513, 513), 16), ((513, 513), 4), ((513, 513), 1), ((257, 257), 16), ((200, 141), 13), ((130, 67), 2),
                                         ((66, 300), 1), ((65, 65), 3)])
@pytest.mark.parametrize("omega", [1.0, 1.6])
def test_vcycle_contracted_vs_oracle(ES, orc, shape, batch, omega):
    """Every tile geometry (48 x 8 from 13 members, 32 x 8, 24 x 12 for a lone member), non-square and even-sized grids, both
    relaxation forms (omega == 1: four FMAs with the column predicate folded into the operands; general omega: five)."""
    nz, nr = shape
    psi0, source, rr, r, z = bench_problem(nz, nr)
    cfg = iter_like_cfg(nr)
    cfg["grid_resolution"] = [nr, nz]
    cfg["dimensions"] = {"R_min": float(r[0]), "R_max": float(r[-1]), "Z_min": float(z[0]), "Z_max": float(z[-1])}
    ref = orc.multigrid_vcycle(psi0, source, rr, float(r[1] - r[0]), float(z[1] - z[0]), omega=omega)
    with ES(cfg, batch=batch, arithmetic="contracted", r_axis=r, z_axis=z) as es:
        es.set_psi(np.broadcast_to(psi0, (batch, nz, nr)).copy())
        es.set_source(np.broadcast_to(source, (batch, nz, nr)).copy())
        es.vcycle(omega=omega, pre_smooth=3, post_smooth=3, min_grid=5)
        out = es.get_psi()
    err = np.abs(out - ref[None]).max() / np.abs(ref).max()
    assert err <= OP_TOL, err
    assert all(np.array_equal(out[0], out[b]) for b in range(batch))  # members are independent and deterministic
    assert np.array_equal(out[0][0], psi0[0]) and np.array_equal(out[0][:, -1], psi0[:, -1])  # the Dirichlet ring is untouched, exactly


@pytest.mark.parametrize("omega", [1.0, 1.6])
def test_config2_fixed_boundary_257_contracted_vs_oracle(ES, orc, omega):
    """BASELINE config 2: ITER-like fixed-boundary 257 x 257, Picard + multigrid, tol 1e-8, sor_omega 1.0 / 1.6."""
    cfg = iter_like_cfg(257, omega=omega, tol=1e-8, max_iter=600)
    ref = orc.picard_solve(cfg)
    with ES(cfg, batch=2, arithmetic="contracted") as es:
        res = es.solve_equilibrium(check_every=7)
        psi = es.get_psi()
    assert res[0]["converged"] and ref["converged"]
    assert _counts_close(res[0]["iterations"], ref["iterations"]), (res[0]["iterations"], ref["iterations"])
    err = np.abs(psi[0] - ref["psi"]).max() / np.abs(ref["psi"]).max()
    assert err <= PSI_TOL, (err, res[0]["iterations"], ref["iterations"])
    assert np.array_equal(psi[0], psi[1])


def test_hmode_contracted_vs_oracle(ES, orc):
    cfg = iter_like_cfg(129, tol=1e-8, max_iter=600)
    cfg["physics"]["profiles"] = {"mode": "h-mode", "p_prime": {"ped_top": 0.9}}
    ref = orc.picard_solve(cfg)
    with ES(cfg, batch=1, arithmetic="contracted") as es:
        res = es.solve_equilibrium()
        psi = es.get_psi(0)
    assert _counts_close(res[0]["iterations"], ref["iterations"]), (res[0]["iterations"], ref["iterations"])
    assert np.abs(psi - ref["psi"]).max() <= PSI_TOL * np.abs(ref["psi"]).max()


def test_config3_diiid_like_free_boundary_513_contracted_vs_oracle(ES, orc, golden):
    """BASELINE config 3: ONE DIII-D-like 513 x 513 free-boundary equilibrium (validation/diiid_config.json geometry from the golden
    config, tol 1e-8, 20 outer iterations allowed) -- and the five-threshold strict-parity gate of the contracted map against
    the oracle's (validation/benchmark_freegs_public_example_reconstruction.py:491-497)."""
    from scpn_fusion_core_amd import parity_gate

    cfg = copy.deepcopy(configs_from(golden("free_boundary.npz"))["d3d33"])
    cfg["grid_resolution"] = [513, 513]
    co = cfg["coils"]
    ref = orc.free_boundary_solve(cfg, [c["r"] for c in co], [c["z"] for c in co], [c["current"] for c in co], [1.0] * len(co))
    with ES(cfg, batch=1, arithmetic="contracted") as es:
        res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
        psi = es.get_psi(0)
        jphi = es.get_jphi(0)
        r_axis, z_axis = es.R.copy(), es.Z.copy()
    assert res[0]["outer_iterations"] == ref["outer_iterations"]
    assert _counts_close(res[0]["picard_iterations"], ref["picard_iterations"]), (res[0]["picard_iterations"], ref["picard_iterations"])
    err = np.abs(psi - ref["psi"]).max() / np.abs(ref["psi"]).max()
    assert err <= PSI_TOL, err
    ip = cfg["physics"]["plasma_current_target"]
    m = parity_gate.strict_parity_metrics(psi, ref["psi"], r_axis, z_axis, x_points=[res[0]["x_point"]],
                                          jtor=jphi, plasma_current_a=ip)
    assert m["pass"] and len(m["checks"]) == 5, m
    assert m["psi_n_rmse"] <= 1e-9 and m["axis_error_m"] == 0.0 and m["boundary_max_abs_error_wb"] <= 1e-10


def test_config5_shape_scan_members_contracted_vs_oracle(ES, orc):
    """BASELINE config 5 / the bench workload: a 32-member block of the 513 x 513 coil / shape scan solved with contracted
    arithmetic; six members (first, last, lowest / highest kappa, two drawn) also by the CPU oracle, one per host core."""
    from concurrent.futures import ThreadPoolExecutor

    from scpn_fusion_core_amd.cases import member_cfg, shape_scan_member_table

    n, total = 513, 32
    cfg = iter_like_cfg(n, method="multigrid", omega=1.0, tol=1e-8, max_iter=1000, alpha=0.12)
    t = shape_scan_member_table(cfg, total)
    check = sorted({0, total - 1, int(np.argmin(t["kappa"])), int(np.argmax(t["kappa"])), 11, 20})
    with ES(cfg, batch=total, coil_currents=t["currents"], current_targets=t["targets"], coil_r=t["coil_r"], coil_z=t["coil_z"],
            arithmetic="contracted") as es:
        assert es.num_groups == 2
        res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4, check_every=8)
        picked = {m: es.get_psi(m) for m in check}
    assert all(r["converged"] for r in res)

    def cpu(m):
        cm = member_cfg(cfg, t["currents"][m], t["targets"][m], t["coil_r"][m], t["coil_z"][m])
        co = cm["coils"]
        return orc.free_boundary_solve(cm, [c["r"] for c in co], [c["z"] for c in co], [c["current"] for c in co], [1.0] * len(co),
                                       max_outer=20, tol=1e-4)

    with ThreadPoolExecutor(max_workers=6) as pool:
        refs = dict(zip(check, pool.map(cpu, check)))
    worst = 0.0
    for m, psi in picked.items():
        ref = refs[m]
        assert res[m]["outer_iterations"] == ref["outer_iterations"], m
        assert _counts_close(res[m]["picard_iterations"], ref["picard_iterations"]), (m, res[m]["picard_iterations"], ref["picard_iterations"])
        worst = max(worst, np.abs(psi - ref["psi"]).max() / np.abs(ref["psi"]).max())
    assert worst <= PSI_TOL, worst


def test_contracted_mode_leaves_the_stateless_operators_exact(ES, orc, monkeypatch):
    """SCPN_HIP_ARITH only sets the default of resident handles: the operator API stays bit-exact."""
    from scpn_fusion_core_amd import multigrid_solve as mgs

    monkeypatch.setenv("SCPN_HIP_ARITH", "contracted")
    psi0, source, rr, r, z = bench_problem(129, 129)
    dr, dz = float(r[1] - r[0]), float(z[1] - z[0])
    assert np.array_equal(mgs.multigrid_vcycle(psi0.copy(), source, rr, dr, dz), orc.multigrid_vcycle(psi0, source, rr, dr, dz))
    assert np.array_equal(mgs.mg_smooth(psi0.copy(), source, rr, dr, dz, 1.3, 6), orc.mg_smooth(psi0, source, rr, dr, dz, 1.3, 6))


@pytest.mark.parametrize("nr,nz,batch", [(97, 161, 3), (200, 141, 5), (130, 67, 2), (65, 257, 7), (513, 300, 13)])
def test_contracted_non_square_grids_odd_batches_vs_oracle(ES, orc, nr, nz, batch):
    """Edge tiles (grids that are no multiple of the tile window, even sizes), ensembles that split unevenly over the stream groups, the
    source-on-the-fly path in every tile geometry: every member against its own CPU solve, fixed boundary then three free-boundary outer iterations."""
    cfg = iter_like_cfg(33, tol=1e-6, max_iter=300)
    cfg["grid_resolution"] = [nr, nz]
    targets = 15.0 * (1.0 + 0.05 * np.linspace(-1, 1, batch))
    with ES(cfg, batch=batch, current_targets=targets, arithmetic="contracted") as es:
        res = es.solve_equilibrium(check_every=7)
        psi = es.get_psi()
        fres = es.solve_free_boundary(max_outer_iter=3, tol=1e-4)
        fpsi = es.get_psi()
    co = cfg["coils"]
    for b in sorted({0, batch // 2, batch - 1}):
        c = {**cfg, "physics": dict(cfg["physics"], plasma_current_target=float(targets[b]))}
        ref = orc.picard_solve(c)
        assert _counts_close(res[b]["iterations"], ref["iterations"]) and res[b]["converged"] == ref["converged"], (b, res[b]["iterations"], ref["iterations"])
        assert np.abs(psi[b] - ref["psi"]).max() <= PSI_TOL * np.abs(ref["psi"]).max(), b
        fref = orc.free_boundary_solve(c, [k["r"] for k in co], [k["z"] for k in co], [k["current"] for k in co], [1.0] * len(co),
                                       psi0=ref["psi"], max_outer=3, tol=1e-4)
        assert _counts_close(fres[b]["picard_iterations"], fref["picard_iterations"]), (b, fres[b]["picard_iterations"], fref["picard_iterations"])
        assert np.abs(fpsi[b] - fref["psi"]).max() <= 1e-9 * np.abs(fref["psi"]).max(), b  # (three unconverged outer iterations: the Picard tolerance 1e-6 is what both reach)


def test_contracted_external_profile_mode_and_frozen_members(ES, orc):
    """Paths of the contracted build that do NOT rebuild the source on the fly: external_profile_mode (the source field holds the mapped
    current: BASELINE config 4's re-solve) and an ensemble in which members converge at different iterations (frozen members are skipped)."""
    from scpn_fusion_core_amd import transport_coupling as tc

    n = 129
    cfg = iter_like_cfg(n, tol=1e-8, max_iter=1000)
    rho = np.linspace(0, 1, 64)
    p1 = np.stack([5.0 * (1 - rho**2) ** 1.5, 5.05 * (1 - rho**2) ** 1.5])
    jb = np.stack([1e-3 * rho * (1 - rho), 1.01e-3 * rho * (1 - rho)])
    with ES(cfg, batch=2, arithmetic="contracted") as es:
        es.solve_equilibrium()
        _resid, res = tc.gs_transport_outer_step(es, rho, p1, jb)
        psi1 = es.get_psi()
    ref = orc.picard_solve(cfg, external_profile_mode=True)
    for b in range(2):
        assert _counts_close(res[b]["iterations"], ref["iterations"]) and res[b]["converged"], (res[b]["iterations"], ref["iterations"])
        assert np.abs(psi1[b] - ref["psi"]).max() <= PSI_TOL * np.abs(ref["psi"]).max(), b
    # members with very different targets: the loose one converges (and freezes) dozens of iterations before the tight one
    cfg2 = iter_like_cfg(65, tol=1e-7, max_iter=600)
    targets = np.array([15.0, 3.0, 15.0, 0.0])  # (member 3: the zero-current short-circuit)
    with ES(cfg2, batch=4, current_targets=targets, arithmetic="contracted") as es:
        res = es.solve_equilibrium(check_every=5)
        psi = es.get_psi()
    for b in range(3):
        ref = orc.picard_solve({**cfg2, "physics": dict(cfg2["physics"], plasma_current_target=float(targets[b]))})
        assert _counts_close(res[b]["iterations"], ref["iterations"]), (b, res[b]["iterations"], ref["iterations"])
        assert np.abs(psi[b] - ref["psi"]).max() <= PSI_TOL * np.abs(ref["psi"]).max(), b
    assert res[3]["iterations"] == 0 and res[3]["converged"] and res[0]["iterations"] != res[1]["iterations"]
    assert np.array_equal(psi[0], psi[2])


def test_contracted_hmode_ensemble_keeps_the_stored_source(ES, orc):
    """H-mode (mtanh pedestal: two tanh per point) is not rebuilt on the fly -- the contracted kernels read the stored J_raw field."""
    cfg = iter_like_cfg(129, tol=1e-8, max_iter=600)
    cfg["physics"]["profiles"] = {"mode": "h-mode", "p_prime": {"ped_top": 0.9}}
    ref = orc.picard_solve(cfg)
    with ES(cfg, batch=16, arithmetic="contracted") as es:  # (16 members: two stream groups, the 48 x 8 geometry)
        res = es.solve_equilibrium()
        psi = es.get_psi()
    assert _counts_close(res[0]["iterations"], ref["iterations"]), (res[0]["iterations"], ref["iterations"])
    assert np.abs(psi[0] - ref["psi"]).max() <= PSI_TOL * np.abs(ref["psi"]).max()
    assert all(np.array_equal(psi[0], psi[b]) for b in range(16))


@pytest.mark.parametrize("name", ["iter65_saddle", "d3d33_saddle"])
def test_contracted_saddle_detection_vs_reference_fixture(ES, golden, name):
    """xpoint_use_saddle_detection (core/fusion_kernel.py:284-329): the iteration then takes the separate topology kernels every iteration
    (no topology partials from the tail) -- contracted arithmetic against the REFERENCE's own fixture (tests/golden/saddle.npz)."""
    g = golden("saddle.npz")
    cfg = configs_from(g)[name]
    with ES(cfg, batch=2, arithmetic="contracted") as es:
        res = es.solve_equilibrium(check_every=4)
        psi = es.get_psi()
    st = g[f"{name}_stats"]
    for b in range(2):
        assert _counts_close(res[b]["iterations"], int(st[1])) and res[b]["converged"] == bool(st[0]), (res[b]["iterations"], int(st[1]))
        assert np.abs(psi[b] - g[f"{name}_psi"]).max() <= PSI_TOL * np.abs(g[f"{name}_psi"]).max()


def test_contracted_newton_lane_reaches_the_oracle_picard_solution(ES, orc):
    """The multigrid / finite-difference Newton lane with contracted V-cycles and source evaluations: the fixed point of the reference's
    Picard iteration (CPU oracle) to 1e-8 (the lane's own bar in exact arithmetic, tests/test_gpu_newton.py)."""
    n = 129
    ref = orc.picard_solve(iter_like_cfg(n, tol=1e-8, max_iter=1000))
    c = iter_like_cfg(n, method="newton", tol=1e-8, max_iter=60)
    c["solver"].update(gmres_preconditioner="multigrid", newton_jacobian="finite_difference", newton_damping=1.0)
    with ES(c, batch=2, arithmetic="contracted") as es:
        rn = es.solve_equilibrium()
        psi = es.get_psi()
    assert all(r["converged"] for r in rn) and 1 <= rn[0]["iterations"] - 15 <= 9, rn[0]["iterations"]
    assert np.abs(psi[0] - ref["psi"]).max() <= 1e-8 * np.abs(ref["psi"]).max() and np.array_equal(psi[0], psi[1])


def test_contracted_shape_optimisation_through_the_config_key(golden):
    """`solver.arithmetic = "contracted"` travels with the case: the shape-optimisation variant of config 3 (device Picard solves + host
    bounded least squares, FusionKernel mirror) against the REFERENCE's fixture at its own tolerances."""
    from scpn_fusion_core_amd.free_boundary import CoilSet
    from scpn_fusion_core_amd.fusion_kernel import FusionKernel

    g = golden("shape_d3d.npz")
    cfg = copy.deepcopy(configs_from(g)["shape_d3d129"])
    cfg["solver"]["arithmetic"] = "contracted"
    k = FusionKernel(cfg)
    coils = CoilSet(positions=[(c["r"], c["z"]) for c in cfg["coils"]], currents=np.array([c["current"] for c in cfg["coils"]], dtype=float),
                    turns=[1] * len(cfg["coils"]), current_limits=np.full(len(cfg["coils"]), 5.0e7), target_flux_points=np.array(g["points"]))
    out = k.solve_free_boundary(coils, max_outer_iter=4, tol=1e-6, optimize_shape=True)
    assert k._es.arithmetic == "contracted"
    assert out["outer_iterations"] == int(g["stats"][0])
    np.testing.assert_allclose(out["coil_currents"], g["currents"], rtol=1e-6, atol=1e-12)
    assert np.abs(k.Psi - g["psi"]).max() <= 1e-8 * np.abs(g["psi"]).max()
    k.close()


@pytest.mark.parametrize("value,where", [(np.inf, (20, 24)), (np.nan, (20, 24)), (np.inf, (0, 9)), (np.nan, (64, 13)), (-np.inf, (31, 0))])
def test_contracted_non_finite_entry_in_a_preserved_start(ES, orc, value, where):
    """The divergence path on the TILE kernels under contracted arithmetic (65 x 65; core/fusion_kernel_newton_solver.py:441-470,518-532): an
    interior non-finite entry is sanitised before the first iteration, a non-finite RING value is Dirichlet data and makes the first elliptic
    solve non-finite -- same converged / diverged / iteration outcome and the same finite pattern as the CPU port (in a column that is never
    updated the folded operands return psi itself: 0 * x + psi, also for a non-finite psi)."""
    cfg = iter_like_cfg(65, tol=1e-6, max_iter=60)
    start = orc.picard_solve(cfg)["psi"].copy()
    start[where] = value
    ref = orc.picard_solve(cfg, psi0=start, preserve=True)
    with ES(cfg, batch=2, arithmetic="contracted") as es:
        es.set_psi(np.stack([start, start]))
        res = es.solve_equilibrium(preserve_initial_state=True)
        psi = es.get_psi()
    for k in ("converged", "diverged"):
        assert res[0][k] == ref[k] == res[1][k], (k, res[0][k], ref[k])
    assert _counts_close(res[0]["iterations"], ref["iterations"]), (res[0]["iterations"], ref["iterations"])
    fin = np.isfinite(ref["psi"])
    assert np.array_equal(np.isfinite(psi[0]), fin)
    assert np.abs(psi[0][fin] - ref["psi"][fin]).max() <= PSI_TOL * np.abs(ref["psi"][fin]).max()

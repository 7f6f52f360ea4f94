"""GPU parity tests: the HIP path (through the C ABI) against the golden fixtures generated from the
reference and against the CPU oracle on seeded inputs.

Bars: bit-exact (array_equal) for the linear elliptic operators (smoother, residual, restriction,
prolongation, V-cycle, multigrid_solve, Jacobi/SOR steps, legacy FastSolver ABI); for the nonlinear
Picard / free-boundary solves -- whose global sums are reduced in a different (fixed) order than
NumPy's pairwise sum -- psi within 1e-11 * max|psi| and identical iteration counts; Green's
functions within 1e-12 relative (SciPy's Cephes polynomials vs AGM).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests._problems import bench_problem, configs_from, grid, iter_like_cfg, operator_problem  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    import scpn_fusion_core_amd as pkg
    from scpn_fusion_core_amd import _lib

    _lib.require_gpu()  # fail loudly: no fallback path exists
    return pkg


@pytest.fixture(scope="module")
def mgs(hip):
    from scpn_fusion_core_amd import multigrid_solve

    return multigrid_solve


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle

    return oracle


# ----------------------------------------------------------------------------- a1 smoother
def test_smooth_golden_bit_exact(mgs, golden):
    g = golden("smooth.npz")
    for nz, nr, omega, sweeps in g["cases"]:
        nz, nr, sweeps = int(nz), int(nr), int(sweeps)
        psi0, source, rr, r, z = bench_problem(nz, nr)
        dr, dz = 4.0 / (nr - 1), 8.0 / (nz - 1)
        out = mgs.mg_smooth(psi0.copy(), source, rr, dr, dz, omega, sweeps)
        assert np.array_equal(out, g[f"out_{nz}x{nr}_w{omega}_s{sweeps}"]), (nz, nr, omega, sweeps)


@pytest.mark.parametrize("n", [65, 129, 257])
def test_smooth_vs_oracle_bit_exact(mgs, orc, n):
    psi0, source, rr, r, z = bench_problem(n, n)
    dr, dz = 4.0 / (n - 1), 8.0 / (n - 1)
    ref = orc.mg_smooth(psi0, source, rr, dr, dz, 1.3, 20)
    out = mgs.mg_smooth(psi0.copy(), source, rr, dr, dz, 1.3, 20)
    assert np.array_equal(out, ref)


def test_smooth_input_semantics_and_errors(mgs):
    psi0, source, rr, r, z = bench_problem(17, 17)
    p = psi0.copy()
    out = mgs.mg_smooth(p, source, rr, 0.25, 0.5, 1.0, 1)
    assert out is p  # in place + returned, like the reference
    assert np.array_equal(out[0], psi0[0]) and np.array_equal(out[:, -1], psi0[:, -1])  # ring untouched
    for w in (0.5, 2.0, float("nan")):
        with pytest.raises(ValueError):
            mgs.mg_smooth(psi0.copy(), source, rr, 0.25, 0.5, w, 1)


def test_smooth_batch_members_independent(mgs, orc):
    a = bench_problem(33, 33, seed=1)
    b = bench_problem(33, 33, seed=2)
    psi = np.stack([a[0], b[0], a[0]])
    src = np.stack([a[1], 2.0 * b[1], a[1]])
    out = mgs.mg_smooth(psi.copy(), src, a[2], 0.125, 0.25, 1.6, 7)
    assert np.array_equal(out[0], out[2])
    assert np.array_equal(out[0], orc.mg_smooth(a[0], a[1], a[2], 0.125, 0.25, 1.6, 7))
    assert np.array_equal(out[1], orc.mg_smooth(b[0], 2.0 * b[1], a[2], 0.125, 0.25, 1.6, 7))


def test_discrete_solution_is_fixed_point(mgs):
    # reference tests/test_fusion_kernel_solver_mixins.py:113-134 (fixed point to 1e-12)
    n = 65
    psi0, _, rr, r, z = bench_problem(n, n)
    dr, dz = float(r[1] - r[0]), float(z[1] - z[0])
    psi = np.sin(rr) * np.cos(0.5 * np.repeat(z[:, None], n, axis=1))
    src = mgs.mg_residual(psi, np.zeros_like(psi), rr, dr, dz)  # L*psi
    out = mgs.mg_smooth(psi.copy(), src, rr, dr, dz, 1.6, 3)
    assert np.abs(out - psi).max() < 1e-12


# ----------------------------------------------------------------------------- a2-a6 operators
def test_operators_golden_bit_exact(mgs, golden):
    g = golden("operators.npz")
    for nz, nr in g["shapes"]:
        nz, nr = int(nz), int(nr)
        tag = f"{nz}x{nr}"
        psi, source, rr, dr, dz = operator_problem(nz, nr)
        assert np.array_equal(mgs.mg_residual(psi, source, rr, dr, dz), g[f"res_{tag}"]), tag
        assert mgs.residual_linf(psi, source, rr, dr, dz) == float(g[f"linf_{tag}"]), tag
        fine = g[f"fine_{tag}"]
        c = mgs.restrict_full_weight(fine)
        assert np.array_equal(c, g[f"restrict_{tag}"]), tag
        assert np.array_equal(mgs.prolongate_bilinear(c, nz, nr), g[f"prolong_{tag}"]), tag
        if min(nz, nr) >= 5:
            for omega in (1.0, 1.6):
                v = mgs.multigrid_vcycle(psi, source, rr, dr, dz, omega=omega)
                assert np.array_equal(v, g[f"vcycle_{tag}_w{omega}"]), (tag, omega)


def test_restrict_prolong_shape_contract(mgs):
    with pytest.raises(ValueError):
        mgs.prolongate_bilinear(np.zeros((5, 5)), 17, 17)
    assert mgs.restrict_full_weight(np.zeros((3, 33, 65))).shape == (3, 17, 33)


@pytest.mark.parametrize("shape", [(129, 129), (257, 257), (200, 141), (513, 513), (128, 256), (257, 65), (66, 300), (130, 130)])
def test_vcycle_vs_oracle_bit_exact(mgs, orc, shape):
    nz, nr = shape
    psi, source, rr, dr, dz = operator_problem(nz, nr)
    ref = orc.multigrid_vcycle(psi, source, rr, dr, dz, omega=1.0)
    out = mgs.multigrid_vcycle(psi, source, rr, dr, dz, omega=1.0)
    assert np.array_equal(out, ref)
    assert mgs.residual_linf(out, source, rr, dr, dz) == orc.residual_linf(ref, source, rr, dr, dz)


def test_vcycle_random_shapes_vs_oracle_bit_exact(mgs, orc):
    """48 seeded shapes between 9 and 150 points per side (odd, even, non-square), omega 1.0 / 1.25 / 1.6: every branch of the in-LDS
    coarse kernel (register smoother with wave-uniform and mixed column choice, one-point-per-lane levels with wave shuffles or DPP
    row moves, scratch-field defect, per-cell prolongation) and of the tile path is visited by some shape."""
    rng = np.random.default_rng(20260410)
    shapes = [(int(a), int(b)) for a, b in rng.integers(9, 151, size=(40, 2))] + [(9, 9), (10, 10), (11, 12), (12, 37), (33, 65), (65, 33), (17, 150), (150, 17)]
    for k, (nz, nr) in enumerate(shapes):
        psi, source, rr, dr, dz = operator_problem(nz, nr)
        omega = (1.0, 1.25, 1.6)[k % 3]
        ref = orc.multigrid_vcycle(psi, source, rr, dr, dz, omega=omega)
        out = mgs.multigrid_vcycle(psi, source, rr, dr, dz, omega=omega)
        assert np.array_equal(out, ref), (nz, nr, omega)


def test_vcycle_block_levels_vs_oracle_bit_exact(mgs, orc):
    """Every 2^k+1 level shape the coarse kernel smooths as 2x2 register blocks (17 .. 65 points per side, square and not, one wave
    or many), below tile levels or as the cycle's top level, with omega 1.0 (one-fma relaxation) and 1.3, and with sweep counts
    that leave the fused prolongation / defect without sweeps around them."""
    sides = (9, 17, 33, 65)
    shapes = [(a, b) for a in sides for b in sides if max(a, b) >= 17] + [(129, 65), (65, 129), (257, 33), (33, 257), (129, 257)]
    for k, (nz, nr) in enumerate(shapes):
        psi, source, rr, dr, dz = operator_problem(nz, nr)
        for omega in (1.0, 1.3):
            ref = orc.multigrid_vcycle(psi, source, rr, dr, dz, omega=omega)
            out = mgs.multigrid_vcycle(psi, source, rr, dr, dz, omega=omega)
            assert np.array_equal(out, ref), (nz, nr, omega)
    psi, source, rr, dr, dz = operator_problem(65, 65)
    for pre, post in ((0, 3), (3, 0), (1, 2), (5, 4)):
        ref = orc.multigrid_vcycle(psi, source, rr, dr, dz, omega=1.0, pre_smooth=pre, post_smooth=post)
        out = mgs.multigrid_vcycle(psi, source, rr, dr, dz, omega=1.0, pre_smooth=pre, post_smooth=post)
        assert np.array_equal(out, ref), (pre, post)


def test_vcycle_batched_with_frozen_members_matches_oracle(mgs, orc):
    # multigrid_solve with members that converge at different cycles exercises the pass-through of frozen members
    n = 129
    r, z, rr, zz = grid(n, n, 1.2, 2.2, -0.5, 0.5)
    base = -rr * np.exp(-((rr - 1.7) ** 2 + zz**2) / 0.05)
    srcs = np.stack([base, 1e-4 * base, 0.0 * base, 7.0 * base])
    psi, res, ncy, conv = mgs.multigrid_solve(srcs, np.zeros_like(srcs), 1.2, 2.2, -0.5, 0.5, n, n, tol=1e-7)
    assert len(set(ncy.tolist())) >= 3  # members really stop at different cycles
    for b in range(4):
        p, rs, nc, cv = orc.multigrid_solve(srcs[b], np.zeros((n, n)), 1.2, 2.2, -0.5, 0.5, n, n, tol=1e-7)
        assert np.array_equal(psi[b], p) and res[b] == rs and ncy[b] == nc and conv[b] == cv, b


@pytest.mark.parametrize("pre,post", [(1, 1), (2, 5), (4, 0), (0, 3), (7, 2)])
def test_vcycle_sweep_counts_vs_oracle(mgs, orc, pre, post):
    psi, source, rr, dr, dz = operator_problem(257, 257)
    ref = orc.multigrid_vcycle(psi, source, rr, dr, dz, omega=1.3, pre_smooth=pre, post_smooth=post)
    out = mgs.multigrid_vcycle(psi, source, rr, dr, dz, omega=1.3, pre_smooth=pre, post_smooth=post)
    assert np.array_equal(out, ref)


def test_gs_residual_rms(mgs, golden):
    g = golden("pieces.npz")
    psi = g["psi"]
    r0, r1, z0, z1, dr, dz, _ = g["dims"]
    r = np.linspace(r0, r1, psi.shape[1])
    assert mgs.gs_residual_rms(psi, g["source"], r, dr, dz) == pytest.approx(float(g["gs_rms"]), rel=1e-13)


# ----------------------------------------------------------------------------- a7 multigrid_solve
def test_multigrid_solve_golden_bit_exact(mgs, golden):
    g = golden("mgsolve.npz")
    for n in (33, 65):
        psi, res, ncy, conv = mgs.multigrid_solve(g[f"gauss{n}_source"], np.zeros((n, n)), 1.2, 2.2, -0.5, 0.5, n, n, tol=1e-6)
        assert np.array_equal(psi, g[f"gauss{n}_psi"])
        assert [res, ncy, float(conv)] == list(g[f"gauss{n}_stats"])
    for n in (33, 65, 97):
        r_min, r_max, z_min, z_max = g[f"solovev{n}_extent"]
        r, z, rr, zz = grid(n, n, r_min, r_max, z_min, z_max)
        exact = rr**4 / 8.0 + 0.5 * zz**2
        source = rr**2 + 1.0
        bc = np.zeros((n, n))
        bc[0, :], bc[-1, :], bc[:, 0], bc[:, -1] = exact[0, :], exact[-1, :], exact[:, 0], exact[:, -1]
        psi, res, ncy, conv = mgs.multigrid_solve(source, bc, r_min, r_max, z_min, z_max, n, n, tol=1e-9, max_cycles=200)
        assert np.array_equal(psi, g[f"solovev{n}_psi"]), n
        assert [res, ncy, float(conv)] == list(g[f"solovev{n}_stats"])
    ext = g["solovev65_extent"]
    source65 = grid(65, 65, *ext)[2] ** 2 + 1.0
    psi2, res2, ncy2, conv2 = mgs.multigrid_solve(source65, _solovev_bc(65, ext), *ext, 65, 65, tol=1e-9, max_cycles=200,
                                                 omega=1.6, pre_smooth=2, post_smooth=1)
    assert np.array_equal(psi2, g["solovev65_w16_psi"])
    assert [res2, ncy2, float(conv2)] == list(g["solovev65_w16_stats"])


def _solovev_bc(n, extent):
    r, z, rr, zz = grid(n, n, *extent)
    exact = rr**4 / 8.0 + 0.5 * zz**2
    bc = np.zeros((n, n))
    bc[0, :], bc[-1, :], bc[:, 0], bc[:, -1] = exact[0, :], exact[-1, :], exact[:, 0], exact[:, -1]
    return bc


def test_multigrid_solve_batch_and_errors(mgs, golden):
    g = golden("mgsolve.npz")
    src = g["gauss33_source"]
    srcs = np.stack([src, 0.0 * src, 3.0 * src])
    psi, res, ncy, conv = mgs.multigrid_solve(srcs, np.zeros_like(srcs), 1.2, 2.2, -0.5, 0.5, 33, 33, tol=1e-6)
    assert np.array_equal(psi[0], g["gauss33_psi"]) and ncy[0] == int(g["gauss33_stats"][1])
    assert ncy[1] == 0 and conv[1] and np.all(psi[1] == 0.0)  # already converged member does no cycle
    assert conv.all()
    z = np.zeros((9, 9))
    for kw in (dict(tol=0.0), dict(tol=float("inf")), dict(max_cycles=0), dict(omega=2.0)):
        with pytest.raises(ValueError):
            mgs.multigrid_solve(z, z, 1.0, 2.0, -1.0, 1.0, 9, 9, **kw)
    with pytest.raises(ValueError):
        mgs.multigrid_solve(z, z, 1.0, 2.0, -1.0, 1.0, 9, 8)


def test_multigrid_solve_513_properties(mgs):
    # full BASELINE size: grid-independent cycle count (<12, reference multigrid.rs:343-460) and exact ring
    n = 513
    r, z, rr, zz = grid(n, n, 1.2, 2.2, -0.5, 0.5)
    source = -rr * np.exp(-((rr - 1.7) ** 2 + zz**2) / 0.05)
    rng = np.random.default_rng(5)
    bc = np.zeros((n, n))
    bc[0, :], bc[-1, :], bc[:, 0], bc[:, -1] = rng.normal(size=n) * 1e-3, 0.0, 0.0, 0.0
    bc[0, 0] = bc[0, -1] = 0.0
    psi, res, ncy, conv = mgs.multigrid_solve(source, bc, 1.2, 2.2, -0.5, 0.5, n, n, tol=1e-8)
    assert conv and ncy < 16 and res < 1e-8
    assert np.array_equal(psi[0], bc[0]) and np.array_equal(psi[-1], bc[-1]) and np.array_equal(psi[:, 0], bc[:, 0])
    assert mgs.residual_linf(psi, source, rr, float(r[1] - r[0]), float(z[1] - z[0])) == res
    # linearity: solving for 2*source with 2*bc gives 2*psi up to rounding of the stopping rule
    psi2, _, ncy2, _ = mgs.multigrid_solve(2.0 * source, 2.0 * bc, 1.2, 2.2, -0.5, 0.5, n, n, tol=2e-8)
    assert ncy2 == ncy and np.array_equal(psi2, 2.0 * psi)


# ----------------------------------------------------------------------------- Jacobi / SOR steps
def test_jacobi_sor_steps_golden(mgs, golden):
    g = golden("pieces.npz")
    psi = g["psi"]
    r0, r1, z0, z1, dr, dz, _ = g["dims"]
    r = np.linspace(r0, r1, psi.shape[1])
    assert np.array_equal(mgs.jacobi_step(psi, g["source"], r, dr, dz), g["jacobi"])
    assert np.array_equal(mgs.sor_step(psi, g["source"], r, dr, dz, 1.6), g["sor_w16"])
    bad = psi.copy()
    bad[5, 5] = np.nan
    bad[6, 6] = np.inf
    out = mgs.jacobi_step(bad, g["source"], r, dr, dz)
    assert np.isfinite(out).all()


# ----------------------------------------------------------------------------- a11 Green's functions
def test_coil_flux_golden(hip, golden):
    from scpn_fusion_core_amd import free_boundary as fb

    g = golden("green.npz")
    co = g["coils"]
    r, z = np.linspace(2.2, 10.2, 33), np.linspace(-6.8, 6.8, 33)
    vac = fb.coil_flux_grid(r, z, co[:, 0], co[:, 1], co[:, 2], mu0=1.0, mode="vacuum")
    np.testing.assert_allclose(vac, g["vacuum33"], rtol=1e-12, atol=1e-14)
    ext = fb.coil_flux_grid(r, z, co[:, 0], co[:, 1], co[:, 2], turns=g["turns"], mode="external")
    np.testing.assert_allclose(ext, g["ext33"], rtol=1e-12, atol=1e-20)
    for p, v in zip(g["green_pts"], g["green_vals"]):
        assert fb.green_function(*p) == pytest.approx(v, rel=1e-12, abs=1e-22)
    assert fb.green_function(9.2, 4.08, 9.2, 4.08) == 0.0
    with pytest.raises(ValueError):
        fb.green_function(-1.0, 0.0, 1.0, 0.0)


# ----------------------------------------------------------------------------- a8-a10 Picard
PICARD_CASES = ["iter33_mg_w10", "iter65_mg_w10", "iter33_mg_w16", "iter33_sor", "iter33_jacobi", "iter33_hmode",
                "d3d33_mg", "iter32_even_mg"]


@pytest.mark.parametrize("name", PICARD_CASES)
def test_picard_golden(hip, golden, name):
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    g = golden("picard.npz")
    cfg = configs_from(g)[name]
    with EnsembleSolver(cfg, batch=1) as es:
        res = es.solve_equilibrium(history=True, trace=True, check_every=1)[0]
        psi = es.get_psi(0)
        jphi = es.get_jphi(0)
    conv, its, resid, gs, gs_best = g[f"{name}_stats"]
    assert res["iterations"] == int(its) and res["converged"] == bool(conv)
    tr = res["trace"]
    np.testing.assert_allclose(tr[:, 0:2], g[f"{name}_axis_bnd"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(res["residual_history"], g[f"{name}_diff"], rtol=1e-6, atol=1e-15)
    np.testing.assert_allclose(res["gs_residual_history"], g[f"{name}_gs"], rtol=1e-5, atol=1e-13)
    scale = np.abs(g[f"{name}_psi"]).max()
    assert np.abs(psi - g[f"{name}_psi"]).max() <= 1e-11 * scale
    jscale = np.abs(g[f"{name}_jphi"]).max()
    assert np.abs(jphi - g[f"{name}_jphi"]).max() <= 1e-9 * jscale
    assert res["residual"] == pytest.approx(resid, rel=1e-5)


@pytest.mark.parametrize("name", ["iter33_saddle", "iter65_saddle", "d3d33_saddle"])
def test_picard_saddle_detection_golden(hip, golden, name):
    """solver.xpoint_use_saddle_detection (core/fusion_kernel.py:284-329) on the device: same X-point sequence, iteration count
    and psi as the reference."""
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    g = golden("saddle.npz")
    cfg = configs_from(g)[name]
    with EnsembleSolver(cfg, batch=2, current_targets=np.array([cfg["physics"]["plasma_current_target"]] * 2)) as es:
        res = es.solve_equilibrium(history=True, trace=True, check_every=4)
        psi = es.get_psi()
    st = g[f"{name}_stats"]
    for b in range(2):
        assert res[b]["iterations"] == int(st[1]) and res[b]["converged"] == bool(st[0])
        np.testing.assert_allclose(res[b]["trace"][:, 0:2], g[f"{name}_axis_bnd"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(res[b]["residual_history"], g[f"{name}_diff"], rtol=1e-6, atol=1e-15)
        assert np.abs(psi[b] - g[f"{name}_psi"]).max() <= 1e-11 * np.abs(g[f"{name}_psi"]).max()


def test_find_x_point_saddle_host_mirror(hip, golden, tmp_path):
    import json

    from scpn_fusion_core_amd.fusion_kernel import FusionKernel

    g = golden("saddle.npz")
    cfg = configs_from(g)["iter33_saddle"]
    p = tmp_path / "c.json"
    p.write_text(json.dumps(cfg))
    k = FusionKernel(str(p))
    for name in ("solved", "noisy", "monotone", "bowl", "saddle", "nanhole"):
        (rx, zx), px = k.find_x_point(g[f"unit_{name}_psi"])
        assert [rx, zx, px] == list(g[f"unit_{name}_x"]), name
    k.cfg["solver"]["xpoint_use_saddle_detection"] = False
    (rx, zx), px = k.find_x_point(g["unit_solved_psi"])
    assert [rx, zx, px] == list(g["unit_solved_x_plain"])


def test_picard_ensemble_members_match_single_solves(hip, orc):
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = iter_like_cfg(33, tol=1e-7, max_iter=400)
    base = np.array([c["current"] for c in cfg["coils"]])
    rng = np.random.default_rng(2026)
    currents = base * (1.0 + 0.1 * rng.uniform(-1, 1, size=(4, base.size)))
    targets = np.array([15.0, 14.0, 0.0, 16.0])  # member 2: zero-current vacuum short-circuit
    with EnsembleSolver(cfg, batch=4, coil_currents=currents, current_targets=targets) as es:
        res = es.solve_equilibrium(check_every=5)
        psi = es.get_psi()
        jphi = es.get_jphi()
    for b in range(4):
        c = {**cfg, "coils": [dict(co, current=float(currents[b, i])) for i, co in enumerate(cfg["coils"])],
             "physics": dict(cfg["physics"], plasma_current_target=float(targets[b]))}
        ref = orc.picard_solve(c)
        assert res[b]["iterations"] == ref["iterations"] and res[b]["converged"] == ref["converged"], b
        scale = max(np.abs(ref["psi"]).max(), 1e-30)
        assert np.abs(psi[b] - ref["psi"]).max() <= 1e-10 * scale, b
        assert np.abs(jphi[b] - ref["j_phi"]).max() <= 1e-9 * max(np.abs(ref["j_phi"]).max(), 1e-30), b
    assert res[2]["iterations"] == 0 and res[2]["converged"]


@pytest.mark.parametrize("n,kw", [(129, {}), (257, {"tol": 1e-6}), (257, {"tol": 1e-8}), (129, {"hmode": True}), (96, {"tol": 1e-6})])
def test_picard_vs_oracle_larger_grids(hip, orc, n, kw):
    """(257, tol 1e-8) is BASELINE config 2 at its stated tolerance: ITER-like fixed-boundary 257x257 Picard + multigrid."""
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = iter_like_cfg(n, tol=kw.get("tol", 1e-8), max_iter=600)
    if kw.get("hmode"):
        cfg["physics"]["profiles"] = {"mode": "h-mode", "p_prime": {"ped_top": 0.9}}
    ref = orc.picard_solve(cfg)
    with EnsembleSolver(cfg, batch=2) as es:
        res = es.solve_equilibrium(check_every=7)
        psi = es.get_psi()
    assert res[0]["iterations"] == ref["iterations"] and res[0]["converged"] == ref["converged"]
    assert np.abs(psi[0] - ref["psi"]).max() <= 1e-10 * np.abs(ref["psi"]).max()
    assert np.array_equal(psi[0], psi[1])
    assert res[0]["residual"] == pytest.approx(ref["residual"], rel=1e-5)


def test_external_profile_mode_matches_reference_semantics(hip, orc):
    """with external_profile_mode the loop keeps the (seeded) J_phi and becomes a damped
    fixed-source iteration; the HIP path reproduces that reference behaviour."""
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = iter_like_cfg(33, tol=1e-7, max_iter=300)
    ref = orc.picard_solve(cfg, external_profile_mode=True)
    with EnsembleSolver(cfg, batch=1) as es:
        es.external_profile_mode = True
        res = es.solve_equilibrium()[0]
        psi, jphi = es.get_psi(0), es.get_jphi(0)
    assert res["iterations"] == ref["iterations"] and res["converged"] == ref["converged"]
    assert np.abs(psi - ref["psi"]).max() <= 1e-10 * np.abs(ref["psi"]).max()
    assert np.abs(jphi - ref["j_phi"]).max() <= 1e-12 * np.abs(ref["j_phi"]).max()


def test_anderson_picard_golden(hip, golden, orc):
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    g = golden("anderson.npz")
    cfgs = configs_from(g)
    cfg = cfgs["iter65_anderson_m3"]
    with EnsembleSolver(cfg, batch=2) as es:
        res = es.solve_equilibrium(history=True, check_every=4)
        psi = es.get_psi()
    assert res[0]["iterations"] == int(g["iter65_anderson_m3_stats"][1]) and not res[0]["converged"]
    np.testing.assert_allclose(res[0]["residual_history"], g["iter65_anderson_m3_diff"], rtol=1e-6, atol=1e-11)
    assert np.abs(psi[0] - g["iter65_anderson_m3_psi"]).max() <= 1e-8 * np.abs(g["iter65_anderson_m3_psi"]).max()
    assert np.array_equal(psi[0], psi[1])
    # depth 5: ill-conditioned mixing (see tests/test_oracle_golden.py) -- the first iterations agree
    with EnsembleSolver(cfgs["iter33_anderson"], batch=1) as es:
        r5 = es.solve_equilibrium(history=True)[0]
    np.testing.assert_allclose(r5["residual_history"][:9], g["iter33_anderson_diff"][:9], rtol=1e-6)
    assert r5["iterations"] == 150


@pytest.mark.parametrize("name", ["iter33", "d3d33", "iter65"])
def test_free_boundary_golden(hip, golden, name):
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    g = golden("free_boundary.npz")
    cfg = configs_from(g)[name]
    with EnsembleSolver(cfg, batch=1) as es:
        res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)[0]
        psi = es.get_psi(0)
    st = g[f"{name}_stats"]
    assert res["outer_iterations"] == int(st[0])
    assert res["picard_iterations"] == [int(v) for v in st[2:]]
    scale = np.abs(g[f"{name}_psi"]).max()
    assert np.abs(psi - g[f"{name}_psi"]).max() <= 1e-11 * scale
    assert res["final_diff"] < 1e-4


@pytest.mark.parametrize("name,n", [("d3d33", 129), ("iter33", 129), ("d3d33", 257), ("d3d33", 513)])
def test_free_boundary_larger_grids_vs_oracle(hip, golden, orc, name, n):
    """BASELINE config 3 geometry (DIII-D, validation/diiid_config.json, taken from the golden config) and the ITER-like case on
    finer grids: device free-boundary loop vs the oracle on the same config -- identical outer / Picard iteration counts,
    psi within 1e-10 of max|psi|."""
    import copy

    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = copy.deepcopy(configs_from(golden("free_boundary.npz"))[name])
    cfg["grid_resolution"] = [n, n]
    if n == 257:
        cfg["solver"]["convergence_threshold"] = 1e-6  # (513: the full BASELINE config 3 setting, tol 1e-8, ~5 s on the CPU oracle)
    co = cfg["coils"]
    ref = orc.free_boundary_solve(cfg, [c["r"] for c in co], [c["z"] for c in co], [c["current"] for c in co], [1.0] * len(co))
    with EnsembleSolver(cfg, batch=2) as es:
        res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
        psi = es.get_psi()
    for b in range(2):
        assert res[b]["outer_iterations"] == ref["outer_iterations"] and res[b]["picard_iterations"] == ref["picard_iterations"]
        assert np.abs(psi[b] - ref["psi"]).max() <= 1e-10 * np.abs(ref["psi"]).max()


@pytest.mark.parametrize("nr,nz,batch", [(97, 161, 3), (200, 141, 5), (130, 67, 2), (65, 257, 7)])
def test_picard_and_free_boundary_non_square_grids_odd_batches(hip, orc, nr, nz, batch):
    """Non-square / even-sized grids (tile and coarse-level edge handling, XCD order for grids that are not multiples of 8) and
    ensemble sizes that split unevenly over the two stream groups: every member must match its own CPU solve."""
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = iter_like_cfg(33, tol=1e-6, max_iter=300)
    cfg["grid_resolution"] = [nr, nz]
    targets = 15.0 * (1.0 + 0.05 * np.linspace(-1, 1, batch))
    with EnsembleSolver(cfg, batch=batch, current_targets=targets) as es:
        res = es.solve_equilibrium(check_every=7)
        psi = es.get_psi()
        fres = es.solve_free_boundary(max_outer_iter=3, tol=1e-4)
        fpsi = es.get_psi()
    assert psi.shape == (batch, nz, nr)
    co = cfg["coils"]
    for b in range(batch):
        c = {**cfg, "physics": dict(cfg["physics"], plasma_current_target=float(targets[b]))}
        ref = orc.picard_solve(c)
        assert res[b]["iterations"] == ref["iterations"] and res[b]["converged"] == ref["converged"], b
        assert np.abs(psi[b] - ref["psi"]).max() <= 1e-10 * np.abs(ref["psi"]).max(), b
        fref = orc.free_boundary_solve(c, [k["r"] for k in co], [k["z"] for k in co], [k["current"] for k in co], [1.0] * len(co),
                                       psi0=ref["psi"], max_outer=3, tol=1e-4)
        assert fres[b]["picard_iterations"] == fref["picard_iterations"], b
        assert np.abs(fpsi[b] - fref["psi"]).max() <= 1e-10 * np.abs(fref["psi"]).max(), b


@pytest.mark.parametrize("n", [4, 5, 8, 9, 16, 17, 20])
@pytest.mark.parametrize("method", ["multigrid", "sor", "jacobi", "anderson"])
def test_picard_tiny_grids_all_methods(hip, orc, n, method):
    """Smallest grids the config schema allows (4x4) up to 20x20 -- base-case-only V-cycles, single-workgroup smoothers,
    1-2 interior rows -- for every Picard method, vs the oracle."""
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = iter_like_cfg(n, method=method, omega=1.3 if method != "multigrid" else 1.0, tol=1e-7, max_iter=40, alpha=0.2)
    ref = orc.picard_solve(cfg)
    with EnsembleSolver(cfg, batch=2) as es:
        res = es.solve_equilibrium(check_every=3)
        psi = es.get_psi()
    for b in range(2):
        assert res[b]["iterations"] == ref["iterations"] and res[b]["converged"] == ref["converged"]
        tol = 1e-10 if method != "anderson" else 1e-6  # Anderson: small Gram systems amplify the reduction-order rounding
        assert np.abs(psi[b] - ref["psi"]).max() <= tol * max(np.abs(ref["psi"]).max(), 1e-30)


def test_polyglot_lane_golden(hip, golden):
    from scpn_fusion_core_amd.polyglot import gs_solve_hip

    g = golden("polyglot.npz")
    p = g["params"]
    for n in (17, 65):
        psi = gs_solve_hip(p[0], p[1], p[2], p[3], n, n, p[4], mu0=p[5], n_picard=int(p[6]), n_jacobi=int(p[7]), alpha=p[8],
                           omega_j=p[9], beta_mix=p[10])
        np.testing.assert_allclose(psi, g[f"psi{n}"], rtol=1e-12, atol=0)
        if n == 17:  # published golden: validation/reports/polyglot_gs_solver_comparison.json:117
            assert abs(psi.max() - 0.042651828900105285) < 1e-13


@pytest.mark.gpu
def test_polyglot_lane_diagnostics_golden(hip, golden, orc):
    """Device gs_delta_star / current density / total current / equation residual vs the reference's outputs."""
    from scpn_fusion_core_amd import polyglot as pg

    g = golden("polyglot.npz")
    p = g["params"]
    box = (p[0], p[1], p[2], p[3])
    for tag, f in (("s17", g["psi17"]), ("s65", g["psi65"]), ("rnd", g["rnd_psi"])):
        nz, nr = f.shape
        np.testing.assert_array_equal(pg.gs_delta_star_hip(f, *box), g[f"ds_{tag}"])
        np.testing.assert_array_equal(pg.gs_toroidal_current_density_hip(f, *box, mu0=p[5]), g[f"jd_{tag}"])
        assert pg.gs_total_toroidal_current_hip(f, *box, mu0=p[5]) == float(g[f"it_{tag}"])
        m = pg.gs_equation_residual_hip(f, *box, nr, nz, p[4], p[5], p[10])
        er = g[f"er_{tag}"]
        assert m["abs_max"] == er[0] and m["relative_max"] == er[1]
        assert m["relative_l2"] == pytest.approx(er[2], rel=1e-12)
        mo = orc.np_equation_residual(f, *box, p[4], p[5], p[10])
        assert m["abs_max"] == mo["abs_max"] and m["relative_max"] == mo["relative_max"]



# ----------------------------------------------------------------------------- a14 legacy ABI
def test_legacy_abi_golden_bit_exact(hip, golden):
    from scpn_fusion_core_amd.hpc_bridge import HPCBridge

    g = golden("fastsolver.npz")
    j = g["j"]
    nz, nr = j.shape
    ext = g["extent"]
    br = HPCBridge()
    assert br.is_available()
    br.initialize(nr, nz, (ext[0], ext[1]), (ext[2], ext[3]))
    assert np.array_equal(br.solve(j, iterations=50), g["run50_a"])
    assert np.array_equal(br.solve(j, iterations=50), g["run50_b"])  # warm psi state inside the handle
    br.set_boundary_dirichlet(0.25)
    assert np.array_equal(br.solve(j, iterations=7), g["run7_bc"])
    psi, n, fd = br.solve_until_converged(j, max_iterations=500, tolerance=1e-9, omega=1.5)
    assert np.array_equal(psi, g["conv_psi"]) and [n, fd] == list(g["conv_stats"])
    br.close()


def test_config5_scan_256_members_on_one_gpu(hip, orc):
    """BASELINE config 5 on ONE device: the 256-member 513x513 coil / shape scan of SURVEY 8(d) cfg 5 -- coil currents and I_p
    +-10 % (rng 2026) AND kappa in [1.5, 1.9] through the coil heights Z = 1.2 kappa a (validation/benchmark_vs_freegs.py:424-432),
    every member its own coil geometry -- free-boundary solve of every member.  All 256 converge; eight seeded members (first,
    last, lowest and highest kappa, four drawn) are solved by the CPU oracle too, one per host core (~4 s each), and agree in
    iteration counts and psi."""
    from concurrent.futures import ThreadPoolExecutor

    from scpn_fusion_core_amd.cases import member_cfg, shape_scan_member_table
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    n, total = 513, 256
    cfg = iter_like_cfg(n, method="multigrid", omega=1.0, tol=1e-8, max_iter=1000, alpha=0.12)
    t = shape_scan_member_table(cfg, total)
    assert 1.5 <= t["kappa"].min() and t["kappa"].max() <= 1.9 and np.ptp(t["coil_z"][:, 0]) > 0.9  # the coil heights really move (2.4 m x 0.4)
    check = sorted({0, total - 1, int(np.argmin(t["kappa"])), int(np.argmax(t["kappa"])),
                    *(int(m) for m in np.random.default_rng(5).choice(total, size=4, replace=False))})
    assert len(check) >= 8
    with EnsembleSolver(cfg, batch=total, coil_currents=t["currents"], current_targets=t["targets"], coil_r=t["coil_r"], coil_z=t["coil_z"]) as es:
        res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4, check_every=8)
        picked = {m: es.get_psi(m) for m in check}
    assert len(res) == total and all(r["converged"] for r in res)
    assert all(r["outer_iterations"] == 2 for r in res)

    def cpu(m):
        cm = member_cfg(cfg, t["currents"][m], t["targets"][m], t["coil_r"][m], t["coil_z"][m])
        co = cm["coils"]
        return orc.free_boundary_solve(cm, [c["r"] for c in co], [c["z"] for c in co], [c["current"] for c in co], [1.0] * len(co),
                                       max_outer=20, tol=1e-4)

    with ThreadPoolExecutor(max_workers=8) as pool:  # (ctypes releases the GIL inside the C solve)
        refs = dict(zip(check, pool.map(cpu, check)))
    for m, psi in picked.items():
        ref = refs[m]
        assert res[m]["outer_iterations"] == ref["outer_iterations"] and res[m]["picard_iterations"] == ref["picard_iterations"], m
        assert np.abs(psi - ref["psi"]).max() <= 1e-10 * np.abs(ref["psi"]).max(), m
    assert len({tuple(r["picard_iterations"]) for r in res}) > 1  # (members differ in their iteration counts: a real scan)


def test_picard_on_a_tall_grid_with_more_than_256_tail_tiles_per_member(hip, orc):
    """250 x 1900 (nr x nz): 3 x 136 = 408 tiles of the fused Picard tail per member, so every per-member final reduction
    (k_picard_final, k_jsum_final) walks its partials in more than one chunk of 256; non-square, not of the 2^k + 1 family."""
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = iter_like_cfg(65, tol=1e-12, max_iter=7)
    cfg["grid_resolution"] = [250, 1900]
    ref = orc.picard_solve(cfg, trace=True)
    with EnsembleSolver(cfg, batch=2) as es:
        res = es.solve_equilibrium(history=True, trace=True, check_every=3)
        psi = es.get_psi()
    assert res[0]["iterations"] == ref["iterations"] == 7
    assert np.abs(psi[0] - ref["psi"]).max() <= 1e-11 * np.abs(ref["psi"]).max()
    assert np.array_equal(psi[0], psi[1])
    tr = np.asarray(res[0]["trace"])
    assert np.array_equal(tr[:, 2], ref["trace"][:, 2])  # the X-point index of every iteration
    np.testing.assert_allclose(tr[:, :2], ref["trace"][:, :2], rtol=1e-11)  # psi_axis, psi_boundary
    np.testing.assert_allclose(tr[:, 4:], ref["trace"][:, 4:], rtol=1e-9)   # mean |dpsi|, GS residual RMS
    assert res[0]["gs_residual"] == pytest.approx(ref["gs_residual"], rel=1e-9)


def test_picard_tail_cold_path_sanitises_an_overflowing_residual(hip, orc):
    """A start whose GS residual exceeds the sanitising cap (1e250) in part of the grid: the fused tail takes its cold path (NaN -> 0,
    clip, power-of-two rescaling of the squares) in the waves that see it and reports the reference's stable RMS."""
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = iter_like_cfg(65, tol=1e-12, max_iter=3)
    cfg["solver"]["relaxation_factor"] = 0.01  # the blended iterate stays the rough start: |L* psi| ~ 1e254 in the patch
    nz = nr = 65
    psi0 = np.zeros((nz, nr))
    zz, rr = np.indices((nz, nr))
    psi0[8:30, 8:40] = np.where((zz[8:30, 8:40] + rr[8:30, 8:40]) % 2 == 0, 1e251, -1e251)
    ref = orc.picard_solve(cfg, psi0=psi0, preserve=True, trace=True)
    with EnsembleSolver(cfg, batch=1) as es:
        es.set_psi(psi0[None])
        res = es.solve_equilibrium(preserve_initial_state=True, history=True, trace=True)[0]
    assert res["iterations"] == ref["iterations"]
    got, want = np.asarray(res["trace"])[:, 5], ref["trace"][:, 5]
    assert np.all(np.isfinite(got)) and 1e249 < want.max() <= 1e250  # residuals beyond the cap were clipped, the squares did not overflow
    np.testing.assert_allclose(got, want, rtol=1e-9)


@pytest.mark.parametrize("value,where", [(np.inf, (10, 12)), (np.nan, (10, 12)), (-np.inf, (16, 16)), (np.inf, (0, 5)), (np.nan, (32, 7))])
def test_non_finite_entry_in_a_preserved_start(hip, orc, value, where):
    """``preserve_initial_state`` with a non-finite entry in the supplied flux (core/fusion_kernel_newton_solver.py:441-470: the
    interior is sanitised by ``_prepare_initial_flux``, a non-finite RING value stays Dirichlet data and makes the first elliptic
    solve non-finite -> the divergence path).  The fused Picard tail reads its non-finite verdict off sum |new - cur| (see its
    header): same converged / diverged / iteration outcome and the same flux as the CPU port in every case."""
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = iter_like_cfg(33, tol=1e-6, max_iter=50)
    start = orc.picard_solve(cfg)["psi"].copy()
    start[where] = value
    ref = orc.picard_solve(cfg, psi0=start, preserve=True)
    with EnsembleSolver(cfg, batch=2) as es:
        es.set_psi(np.stack([start, start]))
        res = es.solve_equilibrium(preserve_initial_state=True)
        psi = es.get_psi()
    for k in ("converged", "diverged", "iterations"):
        assert res[0][k] == ref[k] == res[1][k], (k, res[0][k], ref[k])
    fin = np.isfinite(ref["psi"])
    assert np.array_equal(np.isfinite(psi[0]), fin)
    assert np.abs(psi[0][fin] - ref["psi"][fin]).max() <= 1e-10 * np.abs(ref["psi"][fin]).max()

"""GPU parity tests of the batched transport-coupling kernels against fixtures generated from
the reference (tests/golden/transport.npz: TransportSolver.map_profiles_to_2d, scipy.linalg.solve_banded, CurrentDiffusionSolver)
and the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests._problems import configs_from, iter_like_cfg  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    import scpn_fusion_core_amd as pkg
    from scpn_fusion_core_amd import _lib

    _lib.require_gpu()
    return pkg


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle

    return oracle


def test_map_profiles_to_2d_golden(hip, golden):
    """Pressure_2D is bit-exact (np.interp restated exactly); J_phi carries one global sum (device order != NumPy pairwise),
    so it is compared at 1e-13 relative.  Members with different flux maps / profiles are mapped in one call."""
    from scpn_fusion_core_amd import transport_coupling as tc
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    g = golden("transport.npz")
    cfg = configs_from(g)["iter33"]
    with EnsembleSolver(cfg, batch=2) as es:
        es.set_psi(np.stack([g["map_a_psi"], g["map_b_psi"]]))
        ic, p2 = tc.map_profiles_to_2d(es, g["map_a_rho"], np.stack([g["map_a_p1d"], g["map_b_p1d"]]),
                                       np.stack([g["map_a_jbs1d"], g["map_b_jbs1d"]]), return_pressure=True)
        j = es.get_jphi()
    for b, t in enumerate("ab"):
        assert np.array_equal(p2[b], g[f"map_{t}_p2d"]), t
        ref = g[f"map_{t}_jphi"]
        assert np.abs(j[b] - ref).max() <= 1e-13 * np.abs(ref).max(), t
        assert (j[b].sum() * es.dR * es.dZ) == pytest.approx(cfg["physics"]["plasma_current_target"], rel=1e-12)
    with pytest.raises(ValueError):
        tc.map_profiles_to_2d(es, [0.0, 0.5, 0.4], np.zeros(3), np.zeros(3))


def test_map_profiles_large_grid_vs_oracle(hip, orc):
    """513^2, 3 members: against the oracle on the same inputs, plus the size-independent property sum(J) dR dZ = I_target."""
    from scpn_fusion_core_amd import transport_coupling as tc
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    n = 513
    cfg = iter_like_cfg(n)
    d = cfg["dimensions"]
    r, z = np.linspace(d["R_min"], d["R_max"], n), np.linspace(d["Z_min"], d["Z_max"], n)
    rr, zz = np.meshgrid(r, z)
    rng = np.random.default_rng(12)
    base = 40.0 * np.exp(-((rr - 6.0) ** 2 / 3.0 + zz**2 / 8.0)) - 0.3 * np.exp(-((rr - 5.0) ** 2 + (zz + 5.2) ** 2) / 0.5)
    psi = np.stack([base, 0.7 * base + 0.01 * rng.normal(size=base.shape), base[::-1].copy()])
    rho = np.linspace(0, 1, 64)
    p1 = np.stack([5.0 * (1 - rho**2) * (s + 1 - rho**2) for s in (1.0, 2.0, 0.5)])
    jb = np.stack([1e-2 * rho * (1 - rho) * s for s in (1.0, -1.0, 3.0)])
    targets = np.array([15.0, 12.0, 9.0])
    with EnsembleSolver(cfg, batch=3, current_targets=targets) as es:
        es.set_psi(psi)
        ic, p2 = tc.map_profiles_to_2d(es, rho, p1, jb, return_pressure=True)
        j = es.get_jphi()
    for b in range(3):
        rp2, rj, ric = orc.map_profiles_to_2d(psi[b], r, z, r[1] - r[0], z[1] - z[0], d["Z_min"], targets[b], rho, p1[b], jb[b])
        assert np.array_equal(p2[b], rp2), b
        assert np.abs(j[b] - rj).max() <= 1e-12 * np.abs(rj).max(), b
        assert ic[b] == pytest.approx(ric, rel=1e-12)
        assert j[b].sum() * es.dR * es.dZ == pytest.approx(targets[b], rel=1e-11)


def test_tridiagonal_solve_golden_bit_exact(hip, golden):
    from scpn_fusion_core_amd import transport_coupling as tc

    g = golden("transport.npz")
    for name in ("dom50", "piv50", "piv3", "piv2", "dom7"):
        x = tc.solve_banded_tridiagonal(g[f"tri_{name}_dl"], g[f"tri_{name}_d"], g[f"tri_{name}_du"], g[f"tri_{name}_b"])
        assert np.array_equal(x, g[f"tri_{name}_x"]), name
    # batched: both 50-point systems at once
    x = tc.solve_banded_tridiagonal(np.stack([g["tri_dom50_dl"], g["tri_piv50_dl"]]), np.stack([g["tri_dom50_d"], g["tri_piv50_d"]]),
                                    np.stack([g["tri_dom50_du"], g["tri_piv50_du"]]), np.stack([g["tri_dom50_b"], g["tri_piv50_b"]]))
    assert np.array_equal(x[0], g["tri_dom50_x"]) and np.array_equal(x[1], g["tri_piv50_x"])
    with pytest.raises(np.linalg.LinAlgError):
        tc.solve_banded_tridiagonal([0.0, 0.0], [0.0, 1.0, 1.0], [0.0, 0.0], [1.0, 1.0, 1.0])


def test_tridiagonal_solve_large_batch_vs_oracle(hip, orc):
    from scpn_fusion_core_amd import transport_coupling as tc

    rng = np.random.default_rng(21)
    B, n = 4096, 200
    dl, du = rng.normal(size=(B, n - 1)), rng.normal(size=(B, n - 1))
    d = rng.normal(size=(B, n)) + 3.0 * np.sign(rng.normal(size=(B, n)))
    b = rng.normal(size=(B, n))
    x = tc.solve_banded_tridiagonal(dl, d, du, b)
    for s in rng.integers(0, B, 40):
        ref, info = orc.gtsv(dl[s], d[s], du[s], b[s])
        assert info == 0 and np.array_equal(x[s], ref), s
    # residual property on every system
    ax = d * x
    ax[:, 1:] += dl * x[:, :-1]
    ax[:, :-1] += du * x[:, 1:]
    assert np.abs(ax - b).max() <= 1e-9 * max(1.0, np.abs(x).max())


def test_current_diffusion_step_golden(hip, golden, orc):
    """Six Crank-Nicolson steps against the reference history.  Floating point with one libm call (Te**1.5) per cell whose
    device implementation may differ in the last bit: tolerance 1e-13 relative."""
    from scpn_fusion_core_amd import transport_coupling as tc

    g = golden("transport.npz")
    cd = tc.CurrentDiffusionSolver(g["cd_rho"], R0=6.2, a=2.0, B0=5.3, batch=3)
    assert np.array_equal(cd.psi[0], g["cd_psi0"])
    te = np.stack([g["cd_te"], 0.5 * g["cd_te"], g["cd_te"]])
    for k in range(6):
        psi = cd.step(0.5, te, g["cd_ne"], 1.7, g["cd_jbs"], g["cd_jcd"], j_ext=None if k % 2 else np.zeros_like(g["cd_rho"]))
        ref = g["cd_hist"][k]
        assert np.abs(psi[0] - ref).max() <= 1e-13 * np.abs(ref).max(), k
        assert np.array_equal(psi[0], psi[2])
    p1 = g["cd_psi0"].copy()
    for k in range(6):
        p1, info = orc.current_diffusion_step(g["cd_rho"], p1, 6.2, 2.0, 5.3, 0.5, 0.5 * g["cd_te"], g["cd_ne"], 1.7, g["cd_jbs"], g["cd_jcd"])
    assert np.abs(cd.psi[1] - p1).max() <= 1e-13 * np.abs(p1).max()


def test_gs_transport_outer_step(hip, orc):
    """map -> external-profile re-solve -> relative flux change, for 2 members, against the same sequence on the oracle."""
    from scpn_fusion_core_amd import transport_coupling as tc
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    cfg = iter_like_cfg(33, tol=1e-6, max_iter=120)
    rho = np.linspace(0, 1, 50)
    p1 = np.stack([5.0 * (1 - rho**2) ** 1.5 * 2.0, 4.0 * (1 - rho**2)])
    jb = np.stack([1e-3 * rho * (1 - rho), np.zeros_like(rho)])
    with EnsembleSolver(cfg, batch=2) as es:
        es.solve_equilibrium()
        psi0 = es.get_psi()
        resid, res = tc.gs_transport_outer_step(es, rho, p1, jb)
        psi1 = es.get_psi()
    d = cfg["dimensions"]
    r, z = np.linspace(d["R_min"], d["R_max"], 33), np.linspace(d["Z_min"], d["Z_max"], 33)
    for b in range(2):
        ref = orc.picard_solve(cfg, external_profile_mode=True)  # the reference's seed overrides the mapped J
        assert res[b]["iterations"] == ref["iterations"]
        assert np.abs(psi1[b] - ref["psi"]).max() <= 1e-10 * np.abs(ref["psi"]).max()
        assert resid[b] == pytest.approx(np.linalg.norm(ref["psi"] - psi0[b]) / np.linalg.norm(psi0[b]), rel=1e-6)


def test_flux_change_on_device_matches_host_norms(hip):
    """scpn_gs_snapshot_psi / scpn_gs_flux_change: ||psi - psi_old|| / ||psi_old|| per member without field downloads, against
    np.linalg.norm of the downloaded fields (core/integrated_transport_solver_coupling.py:131-146), incl. the rule for a zero
    psi_old, for a handle with two stream groups' worth of members and a grid that is no multiple of the block size."""
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    n, B = 97, 5
    cfg = iter_like_cfg(n, tol=1e-6, max_iter=30)
    rng = np.random.default_rng(7)
    with EnsembleSolver(cfg, batch=B) as es:
        with pytest.raises(ValueError):
            es.flux_change()  # no snapshot yet
        old = rng.standard_normal((B, n, n))
        old[3] = 0.0  # ||psi_old|| < 1e-30 counts as 1
        es.set_psi(old)
        es.snapshot_psi()
        new = old + 1e-3 * rng.standard_normal((B, n, n))
        new[1] = old[1]  # no change: exactly 0
        es.set_psi(new)
        got = es.flux_change()
    norm_old = np.linalg.norm(old.reshape(B, -1), axis=1)
    norm_old = np.where(norm_old < 1e-30, 1.0, norm_old)
    want = np.linalg.norm((new - old).reshape(B, -1), axis=1) / norm_old
    assert got[1] == 0.0 and got[3] == pytest.approx(np.linalg.norm(new[3]), rel=1e-12)
    assert np.allclose(got, want, rtol=1e-12, atol=0.0)


def test_config4_coupled_step_513(hip, orc):
    """BASELINE config 4 at full size: one coupled transport step (map_profiles_to_2d -> external-profile re-solve -> relative
    flux change -> current-diffusion step) on the 513x513 ITER-like grid for 2 members, against the same sequence on the CPU
    oracle (core/integrated_transport_solver_coupling.py:114-146; ~10 s of oracle time)."""
    from scpn_fusion_core_amd import transport_coupling as tc
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    n = 513
    cfg = iter_like_cfg(n, tol=1e-8, max_iter=1000)
    rho = np.linspace(0, 1, 64)
    p1 = np.stack([5.0 * (1 - rho**2) ** 1.5, 5.05 * (1 - rho**2) ** 1.5])  # the benchmark's stub profile and its 1 % drift
    jb = np.stack([1e-3 * rho * (1 - rho), 1.01e-3 * rho * (1 - rho)])
    te = np.stack([8.0 * (1 - 0.9 * rho**2), 8.1 * (1 - 0.9 * rho**2)])
    ne, jcd = 5.0 * (1 - 0.5 * rho**2), np.zeros_like(rho)
    with EnsembleSolver(cfg, batch=2) as es:
        es.solve_equilibrium()
        psi0 = es.get_psi()
        resid, res = tc.gs_transport_outer_step(es, rho, p1, jb)
        psi1 = es.get_psi()
    cd = tc.CurrentDiffusionSolver(rho, R0=6.2, a=2.0, B0=5.3, batch=2)
    cd_psi0 = cd.psi.copy()
    psi_cd = cd.step(0.5, te, ne, 1.7, jb, jcd)
    ref = orc.picard_solve(cfg, external_profile_mode=True)  # the reference's seed overrides the mapped J
    assert ref["converged"]
    for b in range(2):
        assert res[b]["iterations"] == ref["iterations"] and res[b]["converged"] == ref["converged"], b
        assert np.abs(psi1[b] - ref["psi"]).max() <= 1e-10 * np.abs(ref["psi"]).max(), b
        assert resid[b] == pytest.approx(np.linalg.norm(ref["psi"] - psi0[b]) / np.linalg.norm(psi0[b]), rel=1e-6)
        p_ref, _info = orc.current_diffusion_step(rho, cd_psi0[b], 6.2, 2.0, 5.3, 0.5, te[b], ne, 1.7, jb[b], jcd)
        assert np.abs(psi_cd[b] - p_ref).max() <= 1e-13 * np.abs(p_ref).max(), b


def test_flux_evolution_nonadiabatic_golden(hip, golden, orc):
    """solve_flux_evolution_nonadiabatic (core/current_diffusion.py:55-172) with the reference benchmark's drive functions
    (benchmarks/bench_current_diffusion_nonadiabatic.py:43-72), an undamped case (tau = inf) and a rho-dependent tau.  One exp per
    cell and step: 1e-14 relative against NumPy (SIMD exp) and the oracle (libm exp); the exact-balance contract
    psi[n+1] = psi[n] - damping_decrement[n] + source_increment[n] holds bit-exactly (update_residual == 0)."""
    from scpn_fusion_core_amd import transport_coupling as tc

    g = golden("transport.npz")
    for tag, n_rho, n_steps, tau in (("bench", 64, 128, lambda t, gr: 2.0e-6), ("inf", 33, 20, lambda t, gr: np.inf),
                                     ("rho", 48, 40, lambda t, gr: np.where(gr < 0.5, 1.0e-6 * (1 + gr), np.inf))):
        rho = np.linspace(0.0, 1.0, n_rho)
        dt = 1.0e-8
        tr = tc.solve_flux_evolution_nonadiabatic(
            rho, 0.08 - 0.02 * rho**2, tau_psi_fn=tau, R_null_t=lambda t: 0.18, E_theta_t=lambda t, gr: 25.0 + t / (n_steps * dt) + gr,
            eta_spitzer_fn=lambda gr: np.full_like(gr, 2.5e-6), J_theta_t=lambda t, gr: 1.0e5 * (1.0 - 0.3 * gr), dt=dt, n_steps=n_steps)
        assert np.array_equal(tr.time_s, g[f"fx_{tag}_time"]) and np.array_equal(tr.damping_rate, g[f"fx_{tag}_rate"])
        assert np.array_equal(tr.source, g[f"fx_{tag}_source"]) and np.array_equal(tr.hall_drive, g[f"fx_{tag}_hall"])
        assert np.array_equal(tr.resistive_loss, g[f"fx_{tag}_loss"])
        scale = np.abs(g[f"fx_{tag}_psi"]).max()
        assert np.abs(tr.psi - g[f"fx_{tag}_psi"]).max() <= 1e-14 * scale, tag
        np.testing.assert_allclose(tr.source_increment, g[f"fx_{tag}_inc"], rtol=1e-12, atol=1e-30)
        np.testing.assert_allclose(tr.damping_decrement, g[f"fx_{tag}_drop"], rtol=1e-12, atol=1e-30)
        assert np.array_equal(tr.update_residual, np.zeros_like(tr.update_residual))
        assert np.array_equal(tr.psi[1:], tr.psi[:-1] - tr.damping_decrement + tr.source_increment)
        ref = orc.flux_evolution(g[f"fx_{tag}_psi"][0], g[f"fx_{tag}_rate"], g[f"fx_{tag}_source"], dt)[0]
        assert np.abs(tr.psi - ref).max() <= 1e-14 * scale
    # zero drive, zero damping: psi stays constant (current_diffusion.py:76-78)
    rho = np.linspace(0, 1, 16)
    tr = tc.solve_flux_evolution_nonadiabatic(rho, 0.05 + rho, tau_psi_fn=lambda t: np.inf, R_null_t=lambda t: 0.0,
                                              E_theta_t=lambda t, gr: 0.0, eta_spitzer_fn=lambda gr: 0.0, J_theta_t=lambda t, gr: 0.0,
                                              dt=1e-6, n_steps=7)
    assert np.array_equal(tr.psi, np.tile(0.05 + rho, (8, 1)))
    for bad in ({"dt": 0.0}, {"n_steps": 0}):
        kw = dict(tau_psi_fn=lambda t: 1.0, R_null_t=lambda t: 0.1, E_theta_t=lambda t, gr: 0.0, eta_spitzer_fn=lambda gr: 0.0,
                  J_theta_t=lambda t, gr: 0.0, dt=1e-6, n_steps=3)
        kw.update(bad)
        with pytest.raises(ValueError):
            tc.solve_flux_evolution_nonadiabatic(rho, rho, **kw)
    with pytest.raises(ValueError):
        tc.solve_flux_evolution_nonadiabatic(rho, rho, tau_psi_fn=lambda t: -1.0, R_null_t=lambda t: 0.1, E_theta_t=lambda t, gr: 0.0,
                                             eta_spitzer_fn=lambda gr: 0.0, J_theta_t=lambda t, gr: 0.0, dt=1e-6, n_steps=3)


def test_flux_evolution_large_batch_properties(hip, orc):
    """4096 profiles x 1024 rho x 128 steps in one launch: members are independent (equal inputs -> bit-equal outputs), sampled
    members equal the oracle to 1e-14, and the discrete balance holds exactly."""
    from scpn_fusion_core_amd import transport_coupling as tc

    rng = np.random.default_rng(8)
    B, n, steps, dt = 512, 1024, 128, 1e-8
    rate = np.abs(rng.normal(size=(B, 1, n))) * 5e5 * np.ones((1, steps + 1, 1))
    rate[::7] = 0.0
    src = rng.normal(size=(B, steps + 1, n)) * 1e3
    psi0 = rng.normal(size=(B, n))
    rate[1], src[1], psi0[1] = rate[0], src[0], psi0[0]
    psi, inc, drop, resid = tc.flux_evolution_batched(psi0, rate, src, dt)
    assert np.array_equal(psi[0], psi[1]) and np.array_equal(resid, np.zeros_like(resid))
    assert np.array_equal(psi[:, 1:], psi[:, :-1] - drop + inc)
    for b in (0, 7, 100, 511):
        ref = orc.flux_evolution(psi0[b], rate[b], src[b], dt)[0]
        assert np.abs(psi[b] - ref).max() <= 1e-14 * np.abs(ref).max(), b

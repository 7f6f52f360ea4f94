"""GPU parity tests of the on-device field products: psi sampling, B field of all members,
diagnostic response matrices and the probe-based reconstructions built on them, against fixtures generated from the
reference (tests/golden/products.npz) and the CPU oracle."""
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


class _K:  # the attributes interp_psi / sample_flux_at_points read from a kernel
    def __init__(self, psi, cfg):
        d = cfg["dimensions"]
        self.NR, self.NZ = cfg["grid_resolution"]
        self.R = np.linspace(d["R_min"], d["R_max"], self.NR)
        self.Z = np.linspace(d["Z_min"], d["Z_max"], self.NZ)
        self.dR, self.dZ = self.R[1] - self.R[0], self.Z[1] - self.Z[0]
        self.Psi = psi


def _coils(g, fb):
    return fb.CoilSet(positions=[tuple(p) for p in g["coil_pos"]], currents=g["coil_cur"].copy(),
                      turns=[int(t) for t in g["coil_turns"]], current_limits=np.full(len(g["coil_pos"]), 5.0e7))


def test_sample_psi_and_b_field_golden_bit_exact(hip, golden):
    from scpn_fusion_core_amd import free_boundary as fb
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    g = golden("products.npz")
    cfg = configs_from(g)["iter33"]
    k = _K(g["psi"], cfg)
    assert np.array_equal(fb.sample_flux_at_points(k, g["sample_pts"]), g["sample_psi"])
    rng = np.random.default_rng(3)
    stack = np.stack([g["psi"], 2.0 * g["psi"], g["psi"] + rng.normal(size=g["psi"].shape)])
    with EnsembleSolver(cfg, batch=3) as es:
        es.set_psi(stack)
        s = es.sample_psi(g["sample_pts"])
        br, bz = es.compute_b_field_all()
        br1, bz1 = es.compute_b_field(1)
    assert np.array_equal(s[0], g["sample_psi"])
    for b in range(3):
        kb = _K(stack[b], cfg)
        assert np.array_equal(s[b], [fb.interp_psi(kb, r, z) for r, z in g["sample_pts"]])
    assert np.array_equal(br[0], g["b_r"]) and np.array_equal(bz[0], g["b_z"])
    assert np.array_equal(br[1], br1) and np.array_equal(bz[1], bz1)
    with pytest.raises(ValueError):
        fb.sample_flux_at_points(k, np.zeros((0, 2)))
    with pytest.raises(ValueError):
        fb.sample_flux_at_points(k, [[np.nan, 0.0]])


def test_probe_response_and_reconstructions_golden(hip, golden, orc):
    from scpn_fusion_core_amd import free_boundary as fb

    g = golden("products.npz")
    coils = _coils(g, fb)
    dirs = ["R" if v == 0 else "Z" for v in g["b_dirs"]]
    n_flux = len(g["flux_pts"])
    resp = fb.build_magnetic_probe_response_matrix(None, coils, flux_points=g["flux_pts"], b_probe_points=g["b_pts"],
                                                   b_probe_directions=[d.lower() for d in dirs])
    ref = orc.probe_response(g["coil_pos"][:, 0], g["coil_pos"][:, 1], g["coil_turns"], g["flux_pts"], g["b_pts"], dirs)
    np.testing.assert_allclose(resp, ref, rtol=1e-13, atol=1e-300)  # same AGM Green's function: device == oracle
    np.testing.assert_allclose(resp[:n_flux], g["response"][:n_flux], rtol=1e-12)
    np.testing.assert_allclose(resp[n_flux:], g["response"][n_flux:], rtol=1e-6, atol=1e-7 * np.abs(g["response"][n_flux:]).max())
    only_flux = fb.build_magnetic_probe_response_matrix(None, coils, flux_points=g["flux_pts"])
    assert np.array_equal(only_flux, resp[:n_flux])
    only_b = fb.build_magnetic_probe_response_matrix(None, coils, b_probe_points=g["b_pts"], b_probe_directions=dirs)
    assert np.array_equal(only_b, resp[n_flux:])
    np.testing.assert_allclose(fb.build_mutual_inductance_matrix(None, coils, g["flux_pts"]), g["mutual"], rtol=1e-12)
    for kw in ({}, {"b_probe_points": g["b_pts"]}, {"b_probe_points": g["b_pts"], "b_probe_directions": ["R"]},
               {"b_probe_points": g["b_pts"], "b_probe_directions": ["Q"] * 6}, {"flux_points": np.zeros((3, 3))}):
        with pytest.raises(ValueError):  # tests/test_free_boundary_*.py argument contract
            fb.build_magnetic_probe_response_matrix(None, coils, **kw)

    rec = fb.reconstruct_coil_currents_from_magnetic_probes(
        None, coils, flux_points=g["flux_pts"], flux_measurements=g["meas"][:n_flux], b_probe_points=g["b_pts"],
        b_probe_directions=dirs, b_probe_measurements=g["meas"][n_flux:], measurement_sigma=g["sigma"], tikhonov_alpha=1e-12)
    np.testing.assert_allclose(rec["coil_currents"], g["rec_currents"], rtol=1e-5)
    np.testing.assert_allclose(rec["coil_currents"], g["true_cur"], rtol=1e-4)
    assert rec["response_rank"] == int(g["rec_stats"][2]) and rec["active_bounds"] == int(g["rec_stats"][4])
    assert rec["response_condition"] == pytest.approx(g["rec_stats"][3], rel=1e-6)
    with pytest.raises(ValueError):
        fb.reconstruct_coil_currents_from_magnetic_probes(None, coils, flux_points=g["flux_pts"])
    with pytest.raises(ValueError):
        fb.reconstruct_coil_currents_from_magnetic_probes(None, coils, flux_points=g["flux_pts"], flux_measurements=g["meas"][:n_flux],
                                                          measurement_sigma=np.zeros(n_flux))

    diag = fb.reconstruct_boundary_flux_from_coils(None, coils, boundary_points=g["bnd_pts"], limiter_points=g["lim_pts"],
                                                   axis_point=np.array([6.2, 0.0]), x_points=np.array([[5.0, -3.4], [5.0, 3.4]]),
                                                   target_flux=g["bnd_target"])
    np.testing.assert_allclose(diag["reconstructed_flux"], g["bnd_flux"], rtol=1e-11)
    np.testing.assert_allclose(diag["limiter_flux"], g["lim_flux"], rtol=1e-11)
    np.testing.assert_allclose(diag["x_point_flux"], g["x_flux"], rtol=1e-11)
    st = g["bnd_stats"]
    assert diag["response_rank"] == int(st[0]) and diag["min_limiter_distance_m"] == st[1]
    assert diag["boundary_containment_fraction"] == st[2] and diag["boundary_containment_pass"] == bool(st[2] >= 1.0)
    assert diag["axis_flux"] == pytest.approx(st[3], rel=1e-11) and diag["rmse"] == pytest.approx(st[6], rel=1e-10)
    assert diag["x_point_pair_symmetry_abs_error"] == pytest.approx(st[5], abs=1e-9 * abs(st[3]))


def test_sample_psi_large_grid_properties(hip, orc):
    """513^2, 4 members, 10^4 points: samples at grid nodes reproduce the node values (to the rounding of t = (R - R[i-1])/dR), samples of a bilinear
    function are exact to rounding, points outside the grid clamp to the edge cell, and every value matches the oracle."""
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    n = 513
    cfg = iter_like_cfg(n)
    d = cfg["dimensions"]
    r, z = np.linspace(d["R_min"], d["R_max"], n), np.linspace(d["Z_min"], d["Z_max"], n)
    rr, zz = np.meshgrid(r, z)
    rng = np.random.default_rng(9)
    fields = np.stack([1.5 + 0.25 * rr - 0.5 * zz + 0.125 * rr * zz, rng.normal(size=(n, n)), np.sin(rr) * np.cos(zz), rr**2])
    iz, ir = rng.integers(0, n, 2000), rng.integers(0, n, 2000)
    pts = np.concatenate([np.column_stack([r[ir], z[iz]]),
                          np.column_stack([rng.uniform(d["R_min"] - 1, d["R_max"] + 1, 8000), rng.uniform(d["Z_min"] - 1, d["Z_max"] + 1, 8000)])])
    with EnsembleSolver(cfg, batch=4) as es:
        es.set_psi(fields)
        s = es.sample_psi(pts)
    np.testing.assert_allclose(s[1, :2000], fields[1][iz, ir], rtol=0, atol=1e-9)  # t = (R - R[i-1])/dR is 1 only to rounding
    inside = (pts[:, 0] >= d["R_min"]) & (pts[:, 0] <= d["R_max"]) & (pts[:, 1] >= d["Z_min"]) & (pts[:, 1] <= d["Z_max"])
    exact = 1.5 + 0.25 * pts[:, 0] - 0.5 * pts[:, 1] + 0.125 * pts[:, 0] * pts[:, 1]
    np.testing.assert_allclose(s[0, inside], exact[inside], rtol=1e-12)
    dr, dz = r[1] - r[0], z[1] - z[0]
    sel = rng.integers(0, len(pts), 300)
    for b in (1, 2):
        assert np.array_equal(s[b, sel], [orc.interp_psi(fields[b], r, z, dr, dz, *pts[i]) for i in sel])

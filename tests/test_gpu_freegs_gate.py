"""FreeGS pin and strict-parity gate counterpart (north_star: "FreeGS strict-parity psi RMSE within the repo's accepted threshold").

(1) The device Green's function against the FreeGS 0.8.2 vacuum-psi samples the reference holds
    (tests/golden/freegs.npz <- validation/reference_data/.../freegs_public_example_reconstruction_attempt.json).
(2) The gate of validation/benchmark_freegs_public_example_reconstruction.py:491-497 -- psi_N RMSE <= 0.05, axis <= 0.025 m,
    boundary <= 1e-10 Wb, X-point psi_N <= 0.05 -- evaluated for the HIP free-boundary solve against the reference's own native
    lane at 65^2 and 129^2 (same DIII-D-like case, validation/diiid_config.json).
(3) The harness's own native lane (``_native_profile_source_reconstruction``, :275-315) on the device, against the flux the
    imported reference function produced on both public-example grids (tests/golden/freegs_native.npz), and the five-threshold
    comparison record of :431-497 built from the device solve."""
import json

import numpy as np
import pytest

from tests._freegs_machines import tokamak_filaments
from tests._problems import configs_from

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import scpn_fusion_core_amd as pkg
    from scpn_fusion_core_amd import _lib

    _lib.require_gpu()  # fail loudly: no fallback path exists
    return pkg


def test_device_green_function_reproduces_freegs_vacuum_flux(hip, golden):
    from scpn_fusion_core_amd.free_boundary import green_points

    g = golden("freegs.npz")
    cid = "freegs_01_test_tokamak_freeboundary"
    coils = {c["name"]: c["effective_current_a_turns"] for c in json.loads(str(g[f"{cid}_coils"]))}
    fil = tokamak_filaments(coils)
    pts = g[f"{cid}_points"]
    G = green_points(fil[:, 0], fil[:, 1], pts[:, 0], pts[:, 1])  # (n_filaments, 9) on the device
    psi = fil[:, 2] @ G
    want = g[f"{cid}_freegs_vacuum_psi"]
    # the artifact's own acceptance thresholds (max |error| and NRMSE both 1e-12): the restated machine (tests/_freegs_machines.py)
    # with the device Green's function meets them
    tol_abs, tol_nrmse = (float(t) for t in g[f"{cid}_thresholds"])
    assert tol_abs == tol_nrmse == 1e-12
    assert np.abs(psi - want).max() <= tol_abs, np.abs(psi - want).max()
    assert np.sqrt(np.mean((psi - want) ** 2)) / (want.max() - want.min()) <= tol_nrmse
    assert np.abs(g[f"{cid}_native_vacuum_psi"] - want).max() <= float(g[f"{cid}_thresholds"][0])  # the reference's own gate, on its data


@pytest.mark.parametrize("n", [65, 129])
def test_strict_parity_gate_vs_reference_native_lane(hip, golden, n):
    from scpn_fusion_core_amd.ensemble import EnsembleSolver
    from scpn_fusion_core_amd.parity_gate import GATE_THRESHOLDS, strict_parity_metrics

    g = golden("freegs.npz")
    cfg = configs_from(g)[f"d3d{n}"]
    assert json.loads(str(g["gate_thresholds"]))["psi_n_rmse"] == GATE_THRESHOLDS["psi_n_rmse"]
    with EnsembleSolver(cfg, batch=2) as es:
        res = es.solve_free_boundary(max_outer_iter=20, tol=1e-4)
        psi = es.get_psi()
    d = cfg["dimensions"]
    r, z = np.linspace(d["R_min"], d["R_max"], n), np.linspace(d["Z_min"], d["Z_max"], n)
    ax = g[f"d3d{n}_axis_x"]
    m = strict_parity_metrics(psi[0], g[f"d3d{n}_psi"], r, z, x_points=[(ax[2], ax[3])])
    assert m["pass"], m
    assert res[0]["outer_iterations"] == int(g[f"d3d{n}_stats"][0])
    # far inside the gate: the device lane is the same algorithm in the same arithmetic
    assert m["psi_n_rmse"] <= 1e-12 and m["axis_error_m"] == 0.0 and m["boundary_max_abs_error_wb"] <= 1e-13 and m["xpoint_psi_n_error_max"] <= 1e-12
    assert np.array_equal(psi[0], psi[1])


NATIVE_CASES = ("freegs_01_test_tokamak_freeboundary", "freegs_16_diiid_public_example")


@pytest.mark.parametrize("cid", NATIVE_CASES)
def test_native_profile_source_lane_on_the_device(hip, golden, cid):
    """|psi_HIP - psi_reference| <= 1e-10 Wb everywhere (the gate's own boundary threshold), psi_N RMSE <= 1e-9; the three rings of
    the fixture are solved in ONE batched device call and each equals the single solve."""
    from scpn_fusion_core_amd.parity_gate import native_profile_source_reconstruction, normalised_flux

    g = golden("freegs_native.npz")
    r, z, jt, vac = g[f"{cid}_r"], g[f"{cid}_z"], g[f"{cid}_jtor"], g[f"{cid}_vacuum_psi"]
    rings = np.stack([g[f"{cid}_boundary_psi"], g[f"{cid}_external_psi"] - vac, g[f"{cid}_external_psi_far"] - vac])
    wants = np.stack([g[f"{cid}_native_plasma_psi"], g[f"{cid}_native_plasma_psi_cmp"], g[f"{cid}_native_plasma_psi_far"]])
    got = native_profile_source_reconstruction(r, z, rings, np.broadcast_to(jt, rings.shape))
    assert got.shape == wants.shape
    for k in range(3):
        assert np.abs(got[k] - wants[k]).max() <= 1e-10, (k, np.abs(got[k] - wants[k]).max())
        total, total_ref = got[k] + vac, wants[k] + vac
        assert np.sqrt(np.mean((normalised_flux(total) - normalised_flux(total_ref)) ** 2)) <= 1e-9
        for edge in (np.s_[0, :], np.s_[-1, :], np.s_[:, 0], np.s_[:, -1]):
            assert np.array_equal(got[k][edge], rings[k][edge])
    single = native_profile_source_reconstruction(r, z, rings[0], jt)
    assert np.array_equal(single, got[0])


@pytest.mark.parametrize("cid", NATIVE_CASES)
def test_five_threshold_record_from_the_device_solve(hip, golden, cid):
    """same_case_profile_source_comparison with the native flux solved on the device reproduces the reference's record (metrics to
    1e-9 of their value or 1e-12 absolute) and its verdict on all five thresholds -- pass for the near map; for the far map of the
    TestTokamak grid the axis threshold fails, exactly as in the reference's record."""
    from scpn_fusion_core_amd.parity_gate import GATE_THRESHOLDS, same_case_profile_source_comparison

    g = golden("freegs_native.npz")
    spec, vac = json.loads(str(g[f"{cid}_spec"])), g[f"{cid}_vacuum_psi"]
    for which in ("", "_far"):
        want = json.loads(str(g[f"{cid}_record{which}"]))
        ext = g[f"{cid}_external_psi{which}"]
        rec = same_case_profile_source_comparison(g[f"{cid}_r"], g[f"{cid}_z"], ext, ext - vac, g[f"{cid}_jtor"], spec["plasma_current_a"], spec["xpoints"])
        assert set(rec["checks"]) == set(GATE_THRESHOLDS) and len(rec["checks"]) == 5
        for key in ("psi_n_rmse", "axis_error_m", "boundary_max_abs_error_wb", "current_closure_relative_error", "xpoint_psi_n_error_max",
                    "native_plasma_psi_rmse", "native_current_a", "boundary_containment_fraction"):
            assert rec[key] == pytest.approx(want[key], rel=1e-9, abs=1e-12), (which, key)
        assert rec["checks"] == {k: bool(want[{"xpoint_psi_n_error": "xpoint_psi_n_error_max"}.get(k, k)] <= v) for k, v in GATE_THRESHOLDS.items()}
        assert rec["pass"] is all(rec["checks"].values()) and rec["finite_native_psi"]
    assert rec["pass"] is (not cid.startswith("freegs_01"))  # (the far map: only the TestTokamak grid's axis moves)

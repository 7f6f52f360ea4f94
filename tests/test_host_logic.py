"""CPU: host-side logic of the package -- dispatcher seam, trust gate, config contract, sharding helpers."""
import os
import shutil

import numpy as np
import pytest

from scpn_fusion_core_amd import _lib
from scpn_fusion_core_amd import multi_compat as multi
from tests._problems import iter_like_cfg


def test_backend_tier_order_and_registration():
    assert [t.name for t in sorted(multi.BackendTier)] == ["RUST", "GPU", "MOJO", "JULIA", "GO", "JAX", "NUMPY"]
    multi.register_kernel("unit_test_kernel", multi.BackendTier.NUMPY, lambda: "numpy")
    multi.register_kernel("unit_test_kernel", multi.BackendTier.GPU, lambda: "gpu")
    assert multi.registered_tiers("unit_test_kernel") == [multi.BackendTier.GPU, multi.BackendTier.NUMPY]
    try:
        multi.set_availability(multi.BackendTier.GPU, False)
        multi.set_availability(multi.BackendTier.NUMPY, True)
        assert multi.dispatch("unit_test_kernel")() == "numpy"
        assert multi.available_backends()["numpy"] is True and multi.available_backends()["gpu"] is False
        multi.set_availability(multi.BackendTier.GPU, True)
        assert multi.dispatch("unit_test_kernel")() == "gpu"  # lower enum value preferred
        # same (name, tier) again: the FIRST registration stays (the reference appends and stable-sorts, core/_multi_compat.py:256-260)
        multi.register_kernel("unit_test_kernel", multi.BackendTier.GPU, lambda: "gpu2")
        assert multi.dispatch("unit_test_kernel")() == "gpu"
        # registering a kernel CLASS of the same name leaves the kernel's memoised choice alone
        multi.register_kernel_class("unit_test_kernel", multi.BackendTier.GPU, lambda: dict)
        assert multi.dispatch("unit_test_kernel")() == "gpu" and multi.dispatch_kernel_class("unit_test_kernel") is dict
        multi.set_availability(multi.BackendTier.GPU, False)
        multi.set_availability(multi.BackendTier.NUMPY, False)
        with pytest.raises(RuntimeError, match="Registered tiers: \\['gpu', 'numpy'\\]"):
            multi.dispatch("unit_test_kernel")
        with pytest.raises(KeyError):
            multi.dispatch("no_such_kernel")
        with pytest.raises(KeyError):
            multi.dispatch_kernel_class("no_such_class")
    finally:
        multi.set_availability(multi.BackendTier.GPU, None)
        multi.set_availability(multi.BackendTier.NUMPY, None)
    assert multi.is_available(multi.BackendTier.NUMPY) is False  # this package carries no CPU tier
    assert multi.is_available(multi.BackendTier.GPU) == _lib.gpu_available()


def test_gpu_tier_registered_for_the_three_seams(monkeypatch):
    assert multi.BackendTier.GPU in multi.registered_tiers("gs_rb_sor_smooth")
    assert multi.BackendTier.GPU in multi.registered_tiers("multigrid_solve")
    assert multi.BackendTier.GPU in multi.registered_tiers("equilibrium_kernel", kernel_class=True)
    monkeypatch.setenv("SCPN_DISABLE_GPU", "1")
    assert not _lib.gpu_available()  # core/_multi_compat.py:109-113
    multi.reset_probe_cache()
    try:
        with pytest.raises(RuntimeError):
            multi.dispatch("multigrid_solve")
        with pytest.raises(RuntimeError):
            multi.dispatch_kernel_class("equilibrium_kernel")
    finally:
        monkeypatch.delenv("SCPN_DISABLE_GPU")
        multi.reset_probe_cache()


def test_gpu_tier_surface_without_device():
    from scpn_fusion_core_amd import gpu_tier

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    assert gpu_tier.py_gpu_available() is False and gpu_tier.py_gpu_info() is None
    with pytest.raises(RuntimeError):
        gpu_tier.PyGpuSolver(65, 65, 4.0, 8.0, -4.0, 4.0)
    with pytest.raises(RuntimeError):
        gpu_tier.PyGpuSolver(3, 65, 4.0, 8.0, -4.0, 4.0)


def test_trust_gate(tmp_path, monkeypatch):
    from scpn_fusion_core_amd import hpc_bridge as hb

    for var in ("SCPN_SOLVER_LIB", "SCPN_SOLVER_LIB_SHA256", "SCPN_SOLVER_TRUST_MANIFEST"):
        monkeypatch.delenv(var, raising=False)
    lib = tmp_path / "libscpn_solver.so"
    shutil.copy(_lib.LIB_PATH, lib)
    with pytest.raises(ValueError, match="trust metadata is required"):
        hb.HPCBridge(lib_path=str(lib))
    hb.write_sha256_sidecar(lib)
    br = hb.HPCBridge(lib_path=str(lib))
    assert br.loaded
    assert br.is_available() == (_lib.device_count() > 0)
    monkeypatch.setenv("SCPN_SOLVER_LIB_SHA256", "0" * 64)
    with pytest.raises(ValueError, match="does not match"):
        hb.HPCBridge(lib_path=str(lib))
    monkeypatch.delenv("SCPN_SOLVER_LIB_SHA256")
    with pytest.raises(ValueError, match="absolute"):
        hb.HPCBridge(lib_path="libscpn_solver.so")
    link = tmp_path / "link.so"
    os.symlink(lib, link)
    with pytest.raises(ValueError, match="symlink"):
        hb.HPCBridge(lib_path=str(link))
    with pytest.raises(ValueError, match="does not exist"):
        hb.HPCBridge(lib_path=str(tmp_path / "missing.so"))
    # manifest route
    (tmp_path / "libscpn_solver.so.sha256").unlink()
    man = tmp_path / "manifest.json"
    man.write_text('{"libraries": {"libscpn_solver.so": "%s"}}' % hb.LibraryTrust(lib).actual_digest())
    monkeypatch.setenv("SCPN_SOLVER_TRUST_MANIFEST", str(man))
    assert hb.HPCBridge(lib_path=str(lib)).loaded
    man.write_text('{"%s": 5}' % lib)  # flat manifest keyed by path, with a digest of the wrong type
    with pytest.raises(ValueError, match="must be a string"):
        hb.HPCBridge(lib_path=str(lib))
    man.write_text("[1, 2]")
    with pytest.raises(ValueError, match="JSON object"):
        hb.HPCBridge(lib_path=str(lib))
    monkeypatch.delenv("SCPN_SOLVER_TRUST_MANIFEST")
    with pytest.raises(ValueError, match="regular file"):
        hb.HPCBridge(lib_path=str(tmp_path))
    with pytest.raises(ValueError, match="64 hexadecimal"):
        hb.LibraryTrust.parse_digest("abc")
    assert hb.LibraryTrust.parse_digest("A" * 64 + "  libx.so\n") == "a" * 64


def test_bridge_input_validation():
    from scpn_fusion_core_amd import hpc_bridge as hb

    br = hb.HPCBridge()
    br.nr, br.nz = 5, 4
    with pytest.raises(ValueError):
        br._prepare_inputs(np.zeros((5, 4)))
    with pytest.raises(ValueError):
        br._prepare_inputs(np.full((4, 5), np.nan))
    with pytest.raises(ValueError):
        br._prepare_inputs(np.zeros(20))
    assert br._prepare_inputs(np.zeros((4, 5))).flags.c_contiguous


def test_config_contract_and_param_block():
    from scpn_fusion_core_amd.ensemble import picard_params_from_cfg
    from scpn_fusion_core_amd.fusion_kernel import validate_config

    cfg = validate_config(iter_like_cfg(33))
    assert cfg["grid_resolution"] == [33, 33] and cfg["physics"]["vacuum_permeability"] == 1.0
    for bad in ({"grid_resolution": [3, 33]}, {"dimensions": {"R_min": -1.0, "R_max": 2.0, "Z_min": -1, "Z_max": 1}},
                {"dimensions": {"R_min": 2.0, "R_max": 1.0, "Z_min": -1, "Z_max": 1}}):
        with pytest.raises(ValueError):
            validate_config({**iter_like_cfg(33), **bad})
    with pytest.raises(ValueError):
        validate_config({"reactor_name": "x"})
    p = picard_params_from_cfg(cfg, preserve_initial_state=True, has_boundary_flux=True)
    assert p.method == 0 and p.sor_omega == 1.0 and p.alpha == 0.12 and p.tol == 1e-8 and p.max_iter == 1000
    assert p.z_min_cfg == -6.8 and p.r_center == (2.2 + 10.2) / 2.0 and p.preserve_initial_state == 1 and p.hmode == 0
    assert list(p.ped_p) == [0.92, 0.05, 1.0, 0.3]
    cfg["physics"]["profiles"] = {"mode": "h-mode", "p_prime": {"ped_top": 0.9}}
    cfg["solver"]["solver_method"] = "sor"
    p = picard_params_from_cfg(cfg)
    assert p.hmode == 1 and p.ped_p[0] == 0.9 and p.ped_ff[0] == 0.92 and p.method == 1 and p.sor_omega == 1.0
    cfg["solver"]["solver_method"] = "rust_multigrid"
    with pytest.raises(NotImplementedError):
        picard_params_from_cfg(cfg)
    from scpn_fusion_core_amd.ensemble import newton_opts_from_cfg

    cfg["solver"].update(solver_method="newton", gmres_preconditioner="diagonal", newton_line_search=True,
                         newton_line_search_max_backtracks=8)
    o = newton_opts_from_cfg(cfg)
    assert (o.precond, o.line_search, o.max_backtracks, o.gmres_restart, o.gmres_maxiter) == (1, 1, 8, 50, 100)
    assert o.line_search_c == 1e-4 and o.newton_alpha == 0.5 and o.warmup_steps == -1
    for bad in ({"gmres_preconditioner": "bogus"}, {"newton_line_search_c": 1.5}, {"newton_line_search_max_backtracks": 0},
                {"gmres_nonconverged_budget": -1}, {"gmres_ilu_drop_tol": 0.0}):  # tests/test_newton_equilibrium.py:176-230
        with pytest.raises(ValueError):
            newton_opts_from_cfg({**cfg, "solver": {**cfg["solver"], **bad}})
    # "ilu" degrades to the diagonal preconditioner (the reference's fallback when spilu is unavailable, solver_runtime.py:289-293)
    assert newton_opts_from_cfg({**cfg, "solver": {**cfg["solver"], "gmres_preconditioner": "ilu"}}).precond == 1
    mgo = newton_opts_from_cfg({**cfg, "solver": {**cfg["solver"], "gmres_preconditioner": "multigrid", "newton_jacobian": "finite_difference",
                                                  "newton_damping": 1.0}})
    assert (mgo.precond, mgo.reserved, mgo.newton_alpha, mgo.gmres_restart, mgo.gmres_maxiter) == (2, 1, 1.0, 8, 1)
    assert (mgo.gmres_rtol, mgo.gmres_atol, mgo.warmup_steps) == (1e-3, 1e-8, -1)  # inexact-Newton forcing term; the reference's warm-up rule
    tuned = newton_opts_from_cfg({**cfg, "solver": {**cfg["solver"], "gmres_preconditioner": "multigrid", "gmres_restart": 4, "gmres_rtol": 1e-2,
                                                    "gmres_atol": 0.0, "newton_warmup_steps": 2}})
    assert (tuned.gmres_restart, tuned.gmres_rtol, tuned.gmres_atol, tuned.warmup_steps) == (4, 1e-2, 0.0, 2)
    for bad in ({"newton_jacobian": "exact"}, {"newton_damping": 0.0}, {"newton_damping": 1.5}, {"newton_warmup_steps": -2},
                {"gmres_preconditioner": "multigrid", "gmres_rtol": 0.0}, {"gmres_preconditioner": "multigrid", "gmres_restart": 0}):
        with pytest.raises(ValueError):
            newton_opts_from_cfg({**cfg, "solver": {**cfg["solver"], **bad}})
    for k in ("gmres_preconditioner", "newton_line_search", "newton_line_search_max_backtracks"):
        cfg["solver"].pop(k)
    cfg["solver"].update(solver_method="multigrid", require_gs_residual=True, gs_residual_threshold=0.0)
    with pytest.raises(ValueError):
        picard_params_from_cfg(cfg)


def test_operator_argument_validation_precedes_the_device():
    from scpn_fusion_core_amd import multigrid_solve as mgs
    from scpn_fusion_core_amd import free_boundary as fb

    z = np.zeros((9, 9))
    for kw in (dict(tol=0.0), dict(tol=float("nan")), dict(max_cycles=0), dict(omega=0.5)):
        with pytest.raises(ValueError):
            mgs.multigrid_solve(z, z, 1.0, 2.0, -1.0, 1.0, 9, 9, **kw)
    with pytest.raises(ValueError):
        mgs.multigrid_solve(z, z, 1.0, 2.0, -1.0, 1.0, 9, 8)
    assert mgs.validate_sor_omega(1.0) == 1.0
    for w in (0.99, 2.0, float("inf")):
        with pytest.raises(ValueError):
            mgs.validate_sor_omega(w)
    with pytest.raises(ValueError):
        fb.green_function(-1.0, 0.0, 1.0, 0.0)
    with pytest.raises(ValueError):
        fb.green_function(1.0, float("nan"), 1.0, 0.0)
    cs = fb.CoilSet(positions=[(1.0, 0.0), (2.0, 1.0), (3.0, 0.0)], currents=np.ones(3), turns=[4, 2])
    assert list(cs.turns_array()) == [4.0, 2.0, 1.0]


def test_shard_ranges():
    from scpn_fusion_core_amd.distributed import shard_range, shard_sizes

    assert [shard_range(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    assert shard_sizes(10, 4) == [3, 3, 3, 1] and shard_sizes(3, 4) == [1, 1, 1, 0]
    cover = []
    for r in range(3):
        a, b = shard_range(7, r, 3)
        cover += list(range(a, b))
    assert cover == list(range(7))
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def test_second_handle_without_hardware_queues_warns_once(monkeypatch):
    """Stream groups beyond HIP's default of four hardware queues share queues and serialise: the package says so once
    (the library itself never touches the environment)."""
    import warnings

    from scpn_fusion_core_amd import ensemble

    monkeypatch.setattr(ensemble, "_live_streams", 0)
    monkeypatch.setattr(ensemble, "_warned_queues", False)
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        ensemble._note_live_handle(+1, 2)      # one handle, two groups: fits
        assert not seen
        ensemble._note_live_handle(+1, 2)      # a second handle alive: four streams + the null stream
        assert len(seen) == 1 and "GPU_MAX_HW_QUEUES" in str(seen[0].message)
        ensemble._note_live_handle(+1, 2)      # said once
        assert len(seen) == 1
    ensemble._note_live_handle(-1, 2)
    assert ensemble._live_streams == 4
    monkeypatch.setattr(ensemble, "_warned_queues", False)
    monkeypatch.setattr(ensemble, "_live_streams", 0)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        for _ in range(4):
            ensemble._note_live_handle(+1, 2)
        assert not seen


# ---- strict-parity gate, host side (validation/benchmark_freegs_public_example_reconstruction.py:318-378,431-497)
@pytest.mark.parametrize("cid", ["freegs_01_test_tokamak_freeboundary", "freegs_16_diiid_public_example"])
@pytest.mark.parametrize("which", ["", "_far"])
def test_same_case_comparison_record_equals_the_reference_record(golden, cid, which):
    """The five gated metrics (and every other numeric entry) of the comparison record, from the reference's own native-lane flux,
    against the record the reference function wrote into the fixture: equal to the last bit or two."""
    import json

    from scpn_fusion_core_amd import parity_gate as pg

    g = golden("freegs_native.npz")
    spec, want = json.loads(str(g[f"{cid}_spec"])), json.loads(str(g[f"{cid}_record{which}"]))
    ext, vac = g[f"{cid}_external_psi{which}"], g[f"{cid}_vacuum_psi"]
    native = g[f"{cid}_native_plasma_psi{which or '_cmp'}"]
    rec = pg.same_case_profile_source_comparison(g[f"{cid}_r"], g[f"{cid}_z"], ext, ext - vac, g[f"{cid}_jtor"], spec["plasma_current_a"],
                                                 spec["xpoints"], native_plasma_psi=native)
    assert rec["thresholds"] == want["thresholds"] == pg.GATE_THRESHOLDS and len(pg.GATE_THRESHOLDS) == 5
    for key, val in want.items():
        if key == "thresholds":
            continue
        if isinstance(val, float):
            assert rec[key] == pytest.approx(val, rel=1e-14, abs=1e-300), key
        else:
            assert rec[key] == val, key
    assert set(rec["checks"]) == set(pg.GATE_THRESHOLDS)
    far_axis_moved = which == "_far" and cid.startswith("freegs_01")
    assert rec["pass"] is (not far_axis_moved) and rec["checks"]["axis_error_m"] is (not far_axis_moved)
    assert want["current_closure_relative_error"] == pytest.approx(2e-3, rel=1e-9)  # (the fixture's current density is 0.2 % off I_p by construction)


def test_native_lane_validation_precedes_the_device():
    """The ValueErrors of _native_profile_source_reconstruction (:282-291), raised before any device call."""
    from scpn_fusion_core_amd.parity_gate import native_profile_source_reconstruction as lane, strict_parity_metrics

    r, z = np.linspace(1.0, 2.0, 9), np.linspace(-1.0, 1.0, 7)
    ok = np.zeros((9, 7))
    for args, msg in (((r, z, ok, np.zeros((9, 6))), "matching shapes"), ((r, z, ok.T, ok.T), "len\\(R\\) x len\\(Z\\)"),
                      ((r[:2], z, ok[:2], ok[:2]), "at least 3x3"), ((r, z, np.full((9, 7), np.nan), ok), "must be finite"),
                      ((r[::-1], z, ok, ok), "strictly increasing"),
                      ((np.concatenate([r[:-1], [r[-1] + 0.3]]), z, ok, ok), "uniformly spaced")):  # (this lane only: one spacing per direction)
        with pytest.raises(ValueError, match=msg):
            lane(*args)
    with pytest.raises(ValueError, match="plasma_current_a"):
        strict_parity_metrics(ok.T, ok.T, r, z, jtor=ok.T)


def test_shape_scan_table_contract():
    """cases.shape_scan_member_table (BASELINE config 5 / SURVEY 8(d) cfg 5): the current half equals scan_member_table, kappa has a
    stream of its own (member m does not depend on the ensemble size), the coil heights are the reference's build_config expression
    (validation/benchmark_vs_freegs.py:424-432), and the scan is refused on any other coil set."""
    from scpn_fusion_core_amd.cases import (diiid_like_cfg, iter_like_coil_positions, member_cfg, scan_member_table,
                                            shape_scan_member_table)

    cfg = iter_like_cfg(33)
    t = shape_scan_member_table(cfg, 40)
    cur, tgt = scan_member_table(cfg, 40)
    assert np.array_equal(t["currents"], cur) and np.array_equal(t["targets"], tgt)
    assert np.array_equal(shape_scan_member_table(cfg, 7)["kappa"], t["kappa"][:7])
    assert t["kappa"].min() >= 1.5 and t["kappa"].max() <= 1.9 and np.unique(t["kappa"]).size == 40
    r17, z17 = iter_like_coil_positions(1.7)
    assert [c["r"] for c in cfg["coils"]] == r17 and [c["z"] for c in cfg["coils"]] == z17  # the config's own coil set is kappa = 1.7
    for m in (0, 39):
        k = float(t["kappa"][m])
        assert list(t["coil_z"][m]) == [1.2 * k * 2.0, -(1.2 * k * 2.0), 1.2 * k * 2.0 * 1.1, -(1.2 * k * 2.0) * 1.1, 0.0]
        one = member_cfg(cfg, t["currents"][m], t["targets"][m], t["coil_r"][m], t["coil_z"][m])
        assert [c["z"] for c in one["coils"]] == list(t["coil_z"][m]) and one["physics"]["plasma_current_target"] == t["targets"][m]
        assert cfg["coils"][0]["z"] == z17[0]  # (member_cfg copies: the ensemble's config is untouched)
    with pytest.raises(ValueError, match="ITER-like five-coil"):
        shape_scan_member_table(diiid_like_cfg(33), 4)

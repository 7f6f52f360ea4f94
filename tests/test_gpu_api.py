"""GPU: the drop-in surfaces (FusionKernel class, dispatcher seam, scpn_fusion_rs-shaped GPU tier) against the golden
fixtures generated from the reference and the CPU oracle."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests._problems import bench_problem, configs_from, iter_like_cfg  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    from scpn_fusion_core_amd import _lib

    _lib.require_gpu()


def test_fusion_kernel_class_solve_equilibrium(golden, tmp_path):
    from scpn_fusion_core_amd.fusion_kernel import FusionKernel

    g = golden("picard.npz")
    cfg = configs_from(g)["iter33_mg_w10"]
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    k = FusionKernel(path)  # same constructor as the reference: a JSON config path
    assert k.Psi.shape == (33, 33) and k.RR.shape == (33, 33) and k.dR == pytest.approx(8.0 / 32)
    res = k.solve_equilibrium()
    st = g["iter33_mg_w10_stats"]
    assert set(res) == {"psi", "converged", "iterations", "residual", "residual_history", "gs_residual", "gs_residual_best",
                        "gs_residual_history", "wall_time_s", "solver_method"}  # newton_solver.py:604-615
    assert res["converged"] == bool(st[0]) and res["iterations"] == int(st[1]) and res["solver_method"] == "multigrid"
    assert len(res["residual_history"]) == res["iterations"] == len(res["gs_residual_history"])
    scale = np.abs(g["iter33_mg_w10_psi"]).max()
    assert np.abs(k.Psi - g["iter33_mg_w10_psi"]).max() <= 1e-11 * scale and res["psi"] is k.Psi
    np.testing.assert_allclose(k.J_phi, g["iter33_mg_w10_jphi"], rtol=0, atol=1e-9 * np.abs(g["iter33_mg_w10_jphi"]).max())
    # compute_b_field: core/fusion_kernel.py:450-456
    dz, dr = np.gradient(k.Psi, k.dZ, k.dR)
    rs = np.maximum(k.RR, 1e-6)
    assert np.array_equal(k.B_R, -(1.0 / rs) * dz) and np.array_equal(k.B_Z, (1.0 / rs) * dr)
    # warm restart keeps the interior and re-converges to the same fixed point
    res2 = k.solve_equilibrium(preserve_initial_state=True)
    assert res2["converged"] and np.abs(k.Psi - g["iter33_mg_w10_psi"]).max() <= 1e-6 * scale
    with pytest.raises(ValueError):
        k.solve_equilibrium(boundary_flux=np.zeros((5, 5)))
    k.save_results(str(tmp_path / "eq.npz"))
    assert set(np.load(tmp_path / "eq.npz").files) == {"R", "Z", "Psi", "J_phi"}
    k.close()


def test_fusion_kernel_topology_and_vacuum(golden):
    from scpn_fusion_core_amd.fusion_kernel import FusionKernel

    g = golden("pieces.npz")
    gg = golden("green.npz")
    k = FusionKernel(iter_like_cfg(33))
    np.testing.assert_allclose(k.calculate_vacuum_field(), gg["vacuum33"], rtol=1e-12, atol=1e-14)
    (rx, zx), px = k.find_x_point(g["psi"])
    assert [rx, zx, px] == list(g["xpoint"])
    k.Psi = g["psi"].copy()
    assert list(k._find_magnetic_axis()) == list(g["axis"])
    assert np.array_equal(k._jacobi_step(g["psi"], g["source"]), g["jacobi"])
    assert np.array_equal(k._sor_step(g["psi"], g["source"], omega=1.6), g["sor_w16"])
    k.close()


def test_fusion_kernel_free_boundary_and_zero_current(golden):
    from scpn_fusion_core_amd.free_boundary import CoilSet, compute_external_flux
    from scpn_fusion_core_amd.fusion_kernel import FusionKernel

    g = golden("free_boundary.npz")
    cfg = configs_from(g)["d3d33"]
    k = FusionKernel(cfg)
    coils = CoilSet(positions=[(c["r"], c["z"]) for c in cfg["coils"]], currents=np.array([c["current"] for c in cfg["coils"]]),
                    turns=[1] * len(cfg["coils"]))
    out = k.solve_free_boundary(coils, max_outer_iter=20, tol=1e-4, optimize_shape=False)
    st = g["d3d33_stats"]
    assert out["outer_iterations"] == int(st[0]) and out["picard_iterations"] == [int(v) for v in st[2:]]
    assert np.abs(k.Psi - g["d3d33_psi"]).max() <= 1e-11 * np.abs(g["d3d33_psi"]).max()
    ext = compute_external_flux(k, coils)
    # Dirichlet ring = coil flux up to the rounding of the under-relaxation blend (1-a)*x + a*x, as in the reference
    np.testing.assert_allclose(k.Psi[0], ext[0], rtol=4e-16, atol=0)
    np.testing.assert_allclose(k.Psi[:, -1], ext[:, -1], rtol=4e-16, atol=0)
    with pytest.raises(ValueError):
        k.solve_free_boundary(coils, max_outer_iter=0)
    k.close()
    # zero plasma current: the vacuum field is returned untouched (newton_solver.py:448-465)
    cfg0 = iter_like_cfg(33)
    cfg0["physics"]["plasma_current_target"] = 0.0
    k0 = FusionKernel(cfg0)
    r0 = k0.solve_equilibrium()
    assert r0["converged"] and r0["iterations"] == 0 and r0["residual"] == 0.0
    assert np.array_equal(k0.Psi, k0.calculate_vacuum_field()) and not k0.J_phi.any()
    k0.close()


def test_shape_optimisation_outer_loop_golden(golden):
    """solve_free_boundary(optimize_shape=True): device Picard + host bounded least squares per outer iteration
    (core/fusion_kernel_free_boundary.py:664-705)."""
    from scpn_fusion_core_amd.free_boundary import CoilSet
    from scpn_fusion_core_amd.fusion_kernel import FusionKernel

    g = golden("shape.npz")
    cfg = configs_from(g)["shape33"]
    k = FusionKernel(cfg)
    coils = CoilSet(positions=[(c["r"], c["z"]) for c in cfg["coils"]], currents=np.array([c["current"] for c in cfg["coils"]]),
                    turns=[1] * len(cfg["coils"]), current_limits=np.full(len(cfg["coils"]), 5.0e7), target_flux_points=g["points"])
    out = k.solve_free_boundary(coils, max_outer_iter=4, tol=1e-6, optimize_shape=True)
    assert out["outer_iterations"] == int(g["stats"][0])
    np.testing.assert_allclose(out["coil_currents"], g["currents"], rtol=1e-6, atol=1e-9)
    assert np.abs(k.Psi - g["psi"]).max() <= 1e-8 * np.abs(g["psi"]).max()
    assert out["shape_optimization"]["flux_rmse"] == pytest.approx(float(g["stats"][2]), rel=1e-5, abs=1e-14)
    k.close()


def test_ensemble_shape_scan_matches_single_kernel_runs(golden):
    """EnsembleSolver.solve_free_boundary_shape: three members with different starting coil currents / targets; member 0
    is the golden case (reference run), the others are checked against the single-member FusionKernel path."""
    import copy

    from scpn_fusion_core_amd.ensemble import EnsembleSolver
    from scpn_fusion_core_amd.free_boundary import CoilSet
    from scpn_fusion_core_amd.fusion_kernel import FusionKernel

    g = golden("shape.npz")
    cfg = configs_from(g)["shape33"]
    base = np.array([c["current"] for c in cfg["coils"]])
    starts = [base, base * 1.1, base * np.array([0.9, 1.05, 1.0, 1.1, 0.95])[: base.size]]
    pts = [g["points"], g["points"], g["points"] * np.array([1.0, 0.9])]

    def coilset(i):
        return CoilSet(positions=[(c["r"], c["z"]) for c in cfg["coils"]], currents=starts[i].copy(), turns=[1] * base.size,
                       current_limits=np.full(base.size, 5.0e7), target_flux_points=pts[i])

    sets = [coilset(i) for i in range(3)]
    with EnsembleSolver(cfg, batch=3) as es:
        out = es.solve_free_boundary_shape(sets, max_outer_iter=4, tol=1e-6)
        psi = es.get_psi()
    assert out[0]["outer_iterations"] == int(g["stats"][0])
    np.testing.assert_allclose(out[0]["coil_currents"], g["currents"], rtol=1e-6, atol=1e-9)
    assert np.abs(psi[0] - g["psi"]).max() <= 1e-8 * np.abs(g["psi"]).max()
    for i in (1, 2):
        k = FusionKernel(copy.deepcopy(cfg))
        ref = k.solve_free_boundary(coilset(i), max_outer_iter=4, tol=1e-6, optimize_shape=True)
        assert out[i]["outer_iterations"] == ref["outer_iterations"]
        np.testing.assert_allclose(out[i]["coil_currents"], ref["coil_currents"], rtol=1e-8, atol=1e-9)
        assert np.abs(psi[i] - k.Psi).max() <= 1e-9 * np.abs(k.Psi).max()
        assert out[i]["shape_optimization"]["flux_rmse"] == pytest.approx(ref["shape_optimization"]["flux_rmse"], rel=1e-6, abs=1e-14)
        k.close()


def test_fail_on_diverge_and_revert(orc):
    from scpn_fusion_core_amd.fusion_kernel import FusionKernel

    cfg = iter_like_cfg(17, tol=1e-12, max_iter=5)
    k = FusionKernel(cfg)
    bf = np.zeros((17, 17))
    bf[0, 3] = np.nan  # a non-finite Dirichlet value makes Psi_new non-finite in the first iteration
    res = k.solve_equilibrium(boundary_flux=bf)
    assert not res["converged"] and res["iterations"] == 1
    ref = orc.picard_solve(cfg, boundary_flux=bf)
    assert ref["diverged"] and ref["iterations"] == 1
    assert np.array_equal(np.isnan(k.Psi), np.isnan(ref["psi"]))  # both revert to Psi_best (the pre-seed state)
    cfg["solver"]["fail_on_diverge"] = True
    k2 = FusionKernel(cfg)
    with pytest.raises(RuntimeError):
        k2.solve_equilibrium(boundary_flux=bf)
    k.close()
    k2.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle

    return oracle


def test_dispatcher_seam(golden):
    from scpn_fusion_core_amd import multi_compat as multi

    multi.reset_probe_cache()
    assert multi.is_available(multi.BackendTier.GPU) and multi.available_backends()["gpu"]
    smooth = multi.dispatch("gs_rb_sor_smooth")
    g = golden("smooth.npz")
    psi0, source, rr, r, z = bench_problem(33, 33)
    keep = psi0.copy()
    out = smooth(psi0, source, 4.0, 8.0, -4.0, 4.0, omega=1.3, n_sweeps=50)
    assert np.array_equal(psi0, keep)  # input not mutated (tests/test_multi_compat_providers.py:387-411)
    assert out.dtype == np.float64 and np.array_equal(out, g["out_33x33_w1.3_s50"])  # == NumPy tier bit for bit
    assert np.array_equal(out, smooth(psi0, source, 4.0, 8.0, -4.0, 4.0, omega=1.3, n_sweeps=50))  # reproducible
    solve = multi.dispatch("multigrid_solve")
    gm = golden("mgsolve.npz")
    psi, res, ncy, conv = solve(gm["gauss65_source"], np.zeros((65, 65)), 1.2, 2.2, -0.5, 0.5, 65, 65, tol=1e-6)
    assert np.array_equal(psi, gm["gauss65_psi"]) and [res, ncy, float(conv)] == list(gm["gauss65_stats"])
    K = multi.dispatch_kernel_class("equilibrium_kernel")
    assert K.__name__ == "FusionKernel"
    for attr in ("solve_equilibrium", "find_x_point", "compute_b_field", "calculate_vacuum_field", "save_results"):
        assert callable(getattr(K, attr))


def test_gpu_tier_surface_like_bench_gpu_gs_solver():
    """benchmarks/bench_gpu_gs_solver.py:54-113: list-of-f32 call path, rel-L2 vs the fp64 lane (< 1e-4 gate)."""
    from scpn_fusion_core_amd import gpu_tier
    from scpn_fusion_core_amd.multigrid_solve import mg_smooth

    assert gpu_tier.py_gpu_available() and "gfx950" in gpu_tier.py_gpu_info()
    n = 129
    psi0, source, rr, r, z = bench_problem(n, n)
    s = gpu_tier.PyGpuSolver(n, n, 4.0, 8.0, -4.0, 4.0)
    assert s.grid_shape() == (n, n)
    flat = s.solve(psi0.astype(np.float32).ravel().tolist(), source.astype(np.float32).ravel().tolist(), 200, 1.3)
    assert flat.dtype == np.float32 and flat.shape == (n * n,)
    ref = mg_smooth(psi0.copy(), source, rr, 4.0 / (n - 1), 8.0 / (n - 1), 1.3, 200)
    rel = np.linalg.norm(flat.astype(np.float64).reshape(n, n) - ref) / np.linalg.norm(ref)
    assert rel < 1e-4
    with pytest.raises(RuntimeError):
        s.solve([0.0] * 5, [0.0] * 5, 1, 1.3)
    with pytest.raises(RuntimeError):
        s.solve(psi0.ravel().tolist(), source.ravel().tolist(), 1, 2.0)
    gm = gpu_tier.multigrid_vcycle(source, np.zeros_like(psi0), 4.0, 8.0, -4.0, 4.0, n, n, tol=1e-8)
    assert gm[3] and gm[1] < 1e-8


def test_resident_smoother_matches_operator_path():
    from scpn_fusion_core_amd.ensemble import EnsembleSolver
    from scpn_fusion_core_amd.multigrid_solve import mg_smooth, multigrid_vcycle

    n, B = 257, 3
    cfg = iter_like_cfg(n)
    cfg["dimensions"] = {"R_min": 4.0, "R_max": 8.0, "Z_min": -4.0, "Z_max": 4.0}
    psi0, source, rr, r, z = bench_problem(n, n)
    with EnsembleSolver(cfg, batch=B) as es:
        es.set_psi(np.stack([psi0, 2 * psi0, 0 * psi0]))
        es.set_source(np.stack([source, source, 3 * source]))
        es.rb_sor(1.3, 7)
        got = es.get_psi()
        ref = mg_smooth(np.stack([psi0, 2 * psi0, 0 * psi0]), np.stack([source, source, 3 * source]), r, es.dR, es.dZ, 1.3, 7)
        assert np.array_equal(got, ref)
        es.vcycle(omega=1.0)
        ref2 = multigrid_vcycle(ref, np.stack([source, source, 3 * source]), r, es.dR, es.dZ, omega=1.0)
        assert np.array_equal(es.get_psi(), ref2)


def test_rccl_ensemble_gather_through_the_c_abi_on_one_rank():
    """The exchange step of the multi-GPU path (include/scpn_hip.h section 4) with world_size 1: librccl is loaded by the library,
    the communicator is initialised from a unique id, the all-gather of the resident psi blocks lands in a device buffer that
    equals the solver's psi, and the max-reduction returns its input.  (More ranks need more GPUs than a test box has.)"""
    from scpn_fusion_core_amd.distributed import EnsembleComm
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    comm = EnsembleComm(0, 1, 0, addr="127.0.0.1", port=0)  # world 1: no socket is opened
    try:
        info = comm.info()
        assert (info["rank"], info["world"]) == (0, 1) and info["rccl_version"] > 20000
        cfg = iter_like_cfg(65, tol=1e-6, max_iter=40)
        with EnsembleSolver(cfg, batch=3, current_targets=np.array([15.0, 14.0, 16.0])) as es:
            es.solve_equilibrium()
            comm.allgather_psi(es)
            got = comm.download((1, 3, 65, 65))
            assert np.array_equal(got[0], es.get_psi())
            comm.allgather_psi(es)  # a second gather reuses the receive buffer
            comm.wait()
            # the padded block of an uneven ensemble: this rank owns 3 members of a common block of 5 -> two zero members behind
            comm.allgather_psi(es, total=5)
            padded = comm.download((1, 5, 65, 65))
            assert np.array_equal(padded[0, :3], es.get_psi()) and not padded[0, 3:].any()
            assert np.array_equal(EnsembleComm.unpad(padded, 3), es.get_psi())
            with pytest.raises(ValueError):
                comm.allgather_psi(es, total=2)  # a common block smaller than this rank's batch
        assert comm.allreduce_max(3.25) == 3.25
        comm.barrier()
    finally:
        comm.close()


def test_stateless_operator_pool_reuses_solvers_without_changing_results():
    """The solver object of a stateless operator call is pooled per grid (scpn_operator_pool_stats): a second call on the same grid is a
    pool hit, another grid / batch a miss; results are bit-identical whether the object is fresh or reused, and a call with other inputs
    on a reused object sees nothing of the previous one."""
    import ctypes as C

    from scpn_fusion_core_amd import _lib, multigrid_solve as mgs
    from tests._problems import bench_problem

    def stats():
        h, m, i = C.c_long(), C.c_long(), C.c_int()
        assert _lib.lib().scpn_operator_pool_stats(C.byref(h), C.byref(m), C.byref(i)) == 0
        return h.value, m.value, i.value

    psi0, source, rr, r, z = bench_problem(97, 141)
    dr, dz = float(r[1] - r[0]), float(z[1] - z[0])
    first = mgs.multigrid_vcycle(psi0.copy(), source, rr, dr, dz)
    h0, m0, _ = stats()
    again = mgs.multigrid_vcycle(psi0.copy(), source, rr, dr, dz)
    h1, m1, idle = stats()
    assert h1 == h0 + 1 and m1 == m0 and idle >= 1 and np.array_equal(first, again)
    other = mgs.multigrid_vcycle(2.0 * psi0, 3.0 * source, rr, dr, dz)  # same grid, other data: reused object, fresh result
    assert not np.array_equal(other, first) and np.array_equal(other, mgs.multigrid_vcycle(2.0 * psi0, 3.0 * source, rr, dr, dz))
    assert np.array_equal(mgs.multigrid_vcycle(psi0.copy(), source, rr, dr, dz), first)
    batched = mgs.multigrid_vcycle(np.stack([psi0, psi0]), np.stack([source, source]), rr, dr, dz)  # another batch size: its own object
    h2, m2, _ = stats()
    assert m2 == m1 + 1 and np.array_equal(batched[0], first) and np.array_equal(batched[1], first)
    smooth = mgs.mg_smooth(psi0.copy(), source, rr, dr, dz, 1.3, 4)  # another operator on the pooled object of this grid
    assert stats()[0] > h2 and np.array_equal(smooth, mgs.mg_smooth(psi0.copy(), source, rr, dr, dz, 1.3, 4))

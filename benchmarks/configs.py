"""One figure per BASELINE config, measured on the device (called by bench.py outside its timed region).

  config 2  ITER-like fixed-boundary 257x257 Picard + multigrid, tol 1e-8      -> converged solves/s of a resident ensemble
  config 3  ONE DIII-D-like free-boundary equilibrium 513x513 (unbatched)       -> ms per solve (Picard lane, Newton lane)
            + ONE member of the bench workload (ITER-like 513x513 free-boundary) -> ms per solve
  config 4  transport + current-diffusion loop re-solving the equilibrium, 513^2 -> member-steps/s
  smoother  benchmarks/bench_gpu_gs_solver.py's 200 red-black SOR sweeps, 513^2  -> ms, next to the reference's published numbers
"""
from __future__ import annotations

import time

import numpy as np

# validation/reports/gpu_gs_solver_benchmark.json:35 (the reference's own published smoother benchmark, 513^2, 200 sweeps)
PUBLISHED_SMOOTHER_513 = {"numpy_s": 18.85, "wgpu_f32_s": 0.66, "source": "validation/reports/gpu_gs_solver_benchmark.json:35"}


# validation/polyglot/gs_picard_reference.toml (values restated) on a 513 x 513 grid
POLYGLOT_CASE_513 = {"R_min": 1.0, "R_max": 3.0, "Z_min": -1.2, "Z_max": 1.2, "NR": 513, "NZ": 513, "Ip_target": 1.0e6,
                     "mu0": 1.2566370614359173e-6, "n_picard": 8, "n_jacobi": 16, "alpha": 0.1, "omega_j": 0.6666666666666666, "beta_mix": 0.5}


def _best_of(fn, reps=3):
    ts, out = [], None
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t)
    return min(ts), out


def measure_configs(device: int = 0, members: int = 32, check_every: int = 8, arithmetic: str = "contracted") -> dict:
    """``arithmetic`` (``scpn_gs_set_arith``) applies to the resident-handle rows (configs 2, 3, 4); the smoother benchmark and the
    dispatcher seam run the bit-exact operators whatever it says."""
    from scpn_fusion_core_amd.cases import bench_problem, diiid_like_cfg, iter_like_cfg
    from scpn_fusion_core_amd.ensemble import EnsembleSolver

    out: dict = {}
    # ---- config 2
    cfg2 = iter_like_cfg(257, tol=1e-8, max_iter=1000)
    ip = cfg2["physics"]["plasma_current_target"]
    rng = np.random.default_rng(2026)
    out["arithmetic"] = arithmetic
    with EnsembleSolver(cfg2, batch=members, current_targets=ip * (1.0 + 0.1 * (2.0 * rng.random(members) - 1.0)), device=device, arithmetic=arithmetic) as es:
        def run2():
            es.fill_psi(0.0)
            r = es.solve_equilibrium(check_every=check_every)
            es.sync()
            return r
        run2()
        dt, res = _best_of(run2, 2)
    out["config2_fixed_boundary_257"] = {
        "value": round(members / dt, 2), "unit": "converged solves/s", "members": members, "ms_per_ensemble_solve": round(dt * 1e3, 2),
        "converged_members": sum(int(r["converged"]) for r in res),
        "picard_iterations": [min(r["iterations"] for r in res), max(r["iterations"] for r in res)],
        "workload": "ITER-like fixed-boundary 257x257, Picard + V-cycle, tol 1e-8, plasma current +-10 % over the members",
        "memoised_setup": "vacuum field per handle: calculate_vacuum_field of an unchanged coil set (1.8 ms for 32 members at 513^2) is evaluated once per resident handle, the reference pays it in every solve_equilibrium call"}

    # ---- config 3: one equilibrium at a time
    def single(cfg, newton=False, arith=None, **lane):
        c = cfg if not newton else {**cfg, "solver": {**cfg["solver"], "solver_method": "newton", "gmres_preconditioner": "multigrid",
                                                       "newton_jacobian": "finite_difference", "newton_damping": 1.0,
                                                       "max_iterations": 60, "newton_warmup_steps": 2, **lane}}
        with EnsembleSolver(c, batch=1, device=device, arithmetic=arith or arithmetic) as one:
            def run():
                one.fill_psi(0.0)
                r = one.solve_free_boundary(max_outer_iter=20, tol=1e-4, check_every=check_every)
                one.sync()
                return r
            run()
            t, r = _best_of(run, 3)
        return {"ms_per_solve": round(t * 1e3, 2), "converged": bool(r[0]["converged"]), "outer_iterations": r[0]["outer_iterations"],
                "inner_iterations": int(sum(r[0]["picard_iterations"]))}

    d3d, it1 = diiid_like_cfg(513), iter_like_cfg(513, tol=1e-8, max_iter=1000)
    out["config3_single_free_boundary_513"] = {
        "unit": "ms per equilibrium (ONE member, unbatched)",
        "diiid_like": {"picard": single(d3d), "picard_exact_arithmetic": single(d3d, arith="exact"), "newton_fd_mg": single(d3d, True),
                       # the lane's Krylov cap (gmres_restart, default 8: the best setting of the 32-member ensemble) is what costs this case
                       # its extra Newton steps: 13 of its 20 linear solves end at the cap (tools/d3d_newton_fb_hist.py, tools/newton_krylov_cap.py)
                       "newton_fd_mg_krylov24": single(d3d, True, gmres_restart=24)},
        "iter_like_bench_member": {"picard": single(it1), "picard_exact_arithmetic": single(it1, arith="exact"), "newton_fd_mg": single(it1, True)},
        "workload": "solve_free_boundary(max_outer_iter=20, tol=1e-4, optimize_shape=False), 513x513, Picard tol 1e-8 "
                    "(DIII-D-like: validation/diiid_config.json; the real shot g145419 is not in the reference checkout)"}

    # ---- config 3, second variant: the shape-optimisation outer loop (8 isoflux control points on the DIII-D-like grid, SI units as in
    #      tests/golden/make_golden.py::gen_shape_d3d; device Picard + host bounded least squares per outer iteration, as the reference)
    try:
        import copy

        from scpn_fusion_core_amd.free_boundary import CoilSet
        from scpn_fusion_core_amd.fusion_kernel import FusionKernel

        cs = copy.deepcopy(diiid_like_cfg(513, tol=1e-8, max_iter=1000))
        cs["physics"]["vacuum_permeability"] = 4e-7 * np.pi
        cs["physics"]["plasma_current_target"] = 1.6e6
        for c in cs["coils"]:
            c["current"] = c["current"] * 1.0e6
        th = np.linspace(0.0, 2.0 * np.pi, 8, endpoint=False)
        pts = np.column_stack([1.67 + 0.6 * np.cos(th), 1.8 * 0.6 * np.sin(th)])

        def shape_run():
            k = FusionKernel(copy.deepcopy(cs))
            coils = CoilSet(positions=[(c["r"], c["z"]) for c in cs["coils"]], currents=np.array([c["current"] for c in cs["coils"]], dtype=float),
                            turns=[1] * len(cs["coils"]), current_limits=np.full(len(cs["coils"]), 5.0e7), target_flux_points=pts)
            t = time.perf_counter()
            r = k.solve_free_boundary(coils, max_outer_iter=4, tol=1e-6, optimize_shape=True)
            dt = time.perf_counter() - t
            k.close()
            return dt, r
        shape_run()
        dts, rs = zip(*(shape_run() for _ in range(2)))
        out["config3_single_free_boundary_513"]["shape_optimisation"] = {
            "ms_per_solve": round(min(dts) * 1e3, 2), "outer_iterations": int(rs[0]["outer_iterations"]),
            "flux_rmse": float(rs[0]["shape_optimization"]["flux_rmse"]),
            "workload": "solve_free_boundary(max_outer_iter=4, tol=1e-6, optimize_shape=True), 8 isoflux points (ellipse R0 1.67, a 0.6, kappa 1.8), "
                        "DIII-D-like 513x513, one member; includes the host lsq_linear and the diagnostics of every outer iteration"}
    except Exception as exc:
        out["config3_single_free_boundary_513"]["shape_optimisation"] = {"error": str(exc)}

    # ---- config 4
    try:
        from benchmarks.bench_transport_coupled_loop import coupled_steps

        run = coupled_steps(513, members, 3, device=device, arithmetic=arithmetic)
        out["config4_transport_coupled_loop_513"] = {
            "value": round(run["member_steps_per_s"], 2), "unit": "member-steps/s", "members": members,
            "s_per_coupled_step_ensemble": round(run["s_per_coupled_step_ensemble"], 4),
            "picard_iterations_per_step": run["rows"][-1]["picard_iterations"],
            "workload": "map_profiles_to_2d -> external-profile re-solve (tol 1e-6) -> flux change -> current-diffusion step, 513x513",
            "memoised_setup": "vacuum field per handle: calculate_vacuum_field of an unchanged coil set (1.8 ms for 32 members at 513^2) is evaluated once per resident handle, the reference pays it in every solve_equilibrium call"}
    except Exception as exc:  # an optional figure must not cost the bench line
        out["config4_transport_coupled_loop_513"] = {"error": str(exc)}

    # ---- the polyglot Picard + damped-Jacobi lane at 513^2 (north_star's literal comparator; harness
    #      benchmarks/polyglot_gs_solver_comparison.py times gs_solve_np(**case) of validation/polyglot/gs_picard_reference.toml)
    try:
        from scpn_fusion_core_amd.polyglot import gs_solve_hip

        out["polyglot_picard_jacobi_513"] = {"unit": "ms per solve (ONE 513x513 equilibrium, host arrays in and out)", "case": dict(POLYGLOT_CASE_513)}
        for label, kw in (("toml_parameters", {}), ("lane_defaults", {"n_picard": 80, "n_jacobi": 200})):
            case = {**POLYGLOT_CASE_513, **kw}
            gs_solve_hip(**case)
            t, psi = _best_of(lambda: gs_solve_hip(**case), 3)
            out["polyglot_picard_jacobi_513"][label] = {"n_picard": case["n_picard"], "n_jacobi": case["n_jacobi"], "hip_ms": round(t * 1e3, 3),
                                                        "psi_axis": float(psi.max())}
    except Exception as exc:
        out["polyglot_picard_jacobi_513"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- the dispatcher seam (counterpart of benchmarks/bench_dispatcher_kernel_tiers.py:35,138: multigrid_solve through dispatch()):
    #      host buffers in and out per call (seam B3) vs the same V-cycles on a resident handle
    try:
        from scpn_fusion_core_amd import multi_compat

        n = 257
        psi0, source, _rr, r, z = bench_problem(n, n)
        fn = multi_compat.dispatch("multigrid_solve")
        args = (source, np.zeros_like(source), float(r[0]), float(r[-1]), float(z[0]), float(z[-1]), n, n)
        fn(*args, tol=1e-8, max_cycles=50)
        t_seam, res = _best_of(lambda: fn(*args, tol=1e-8, max_cycles=50), 3)
        cyc = int(res[2])
        cfgd = iter_like_cfg(n)
        cfgd["dimensions"] = {"R_min": float(r[0]), "R_max": float(r[-1]), "Z_min": float(z[0]), "Z_max": float(z[-1])}
        with EnsembleSolver(cfgd, batch=1, device=device, arithmetic="exact") as es:
            es.set_source(source, 0)

            def resident():
                es.fill_psi(0.0)
                for _ in range(cyc):
                    es.vcycle(omega=1.0, pre_smooth=3, post_smooth=3, min_grid=5)
                es.sync()
            resident()
            t_res, _ = _best_of(resident, 3)
        out["dispatcher_multigrid_solve_257"] = {
            "unit": "ms per call", "tier": "GPU", "cycles": cyc, "residual": float(res[1]), "converged": bool(res[3]),
            "dispatch_host_buffers_ms": round(t_seam * 1e3, 3), "resident_handle_same_cycles_ms": round(t_res * 1e3, 3),
            "seam_overhead_ms": round((t_seam - t_res) * 1e3, 3),
            "workload": "dispatch('multigrid_solve')(source, psi_bc, ...) at 257x257, tol 1e-8 (host arrays copied in and out, hierarchy built per call, "
                        "residual checked per cycle) vs the same number of V-cycles on a resident handle"}
    except Exception as exc:
        out["dispatcher_multigrid_solve_257"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- smoother benchmark (resident fp64, 200 sweeps, omega 1.3)
    n = 513
    psi0, source, _rr, _r, _z = bench_problem(n, n)
    cfgs = iter_like_cfg(n)
    cfgs["dimensions"] = {"R_min": 4.0, "R_max": 8.0, "Z_min": -4.0, "Z_max": 4.0}
    with EnsembleSolver(cfgs, batch=1, device=device) as es:
        es.set_source(source, 0)

        def sweeps():
            es.set_psi(psi0, 0)
            t = time.perf_counter()
            es.rb_sor(1.3, 200)
            return time.perf_counter() - t
        sweeps()
        ts = float(np.median([sweeps() for _ in range(3)]))
    out["smoother_200_sweeps_513"] = {"value": round(ts * 1e3, 3), "unit": "ms", "published_reference": PUBLISHED_SMOOTHER_513,
                                      "workload": "benchmarks/bench_gpu_gs_solver.py: 200 red-black SOR sweeps, omega 1.3, 513x513, fp64, "
                                                  "fields resident (bit-exact vs the reference: tests/test_gpu_parity.py)"}
    return out

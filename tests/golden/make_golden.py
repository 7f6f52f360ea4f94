#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the fixtures it writes are plain
input/expected-output arrays -- no reference source or bytecode is stored.  Re-run with

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Recorded alongside each file: numpy / scipy versions used.
Reference functions exercised (file:line in the docstring of each block).
"""

from __future__ import annotations

import json
import os
import sys
import tempfile
from pathlib import Path

sys.dont_write_bytecode = True
REF = Path("/root/reference")
sys.path.insert(0, str(REF / "src"))
sys.path.insert(0, str(REF))

import numpy as np  # noqa: E402
import scipy  # noqa: E402
from scipy.special import ellipe, ellipk  # noqa: E402

from scpn_fusion.core import multigrid_solve as mg  # noqa: E402
from scpn_fusion.core.fusion_kernel import CoilSet, FusionKernel  # noqa: E402
from scpn_fusion.core.fusion_kernel_free_boundary import (  # noqa: E402
    compute_external_flux,
    green_function,
)
from scpn_fusion.core.fusion_kernel_solver_runtime import compute_gs_residual_rms  # noqa: E402
from scpn_fusion.core.jax_gs_solver import (  # noqa: E402
    gs_delta_star_np, gs_equation_residual_np, gs_solve_np, gs_toroidal_current_density_np, gs_total_toroidal_current_np)
from validation.benchmark_vs_freegs import TokamakCase, build_config  # noqa: E402

OUT = Path(__file__).resolve().parent
META = {"numpy": np.__version__, "scipy": scipy.__version__, "python": sys.version.split()[0]}


def save(name, **arrs):
    np.savez_compressed(OUT / name, _meta=json.dumps(META), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


def grid(nz, nr, r0=4.0, r1=8.0, z0=-4.0, z1=4.0):
    r = np.linspace(r0, r1, nr)
    z = np.linspace(z0, z1, nz)
    rr, zz = np.meshgrid(r, z)
    return r, z, rr, zz


def bench_problem(nz, nr, seed=2026):
    """benchmarks/bench_gpu_gs_solver.py:54-63 (generalised to nz != nr)."""
    rng = np.random.default_rng(seed)
    r, z, rr, zz = grid(nz, nr)
    source = -np.exp(-((rr - 6.0) ** 2 + zz**2) / 0.5)
    psi0 = rng.normal(0.0, 1e-3, size=(nz, nr))
    psi0[0, :] = psi0[-1, :] = psi0[:, 0] = psi0[:, -1] = 0.0
    return psi0, source, rr, r, z


# ── (a) mg_smooth  core/multigrid_solve.py:148-208 ────────────────────────────
def gen_smooth():
    out = {}
    cases = []
    for (nz, nr) in [(17, 17), (33, 33), (20, 37), (16, 16)]:
        psi0, source, rr, r, z = bench_problem(nz, nr)
        dr = (8.0 - 4.0) / (nr - 1)
        dz = (4.0 + 4.0) / (nz - 1)
        out[f"psi0_{nz}x{nr}"] = psi0
        out[f"src_{nz}x{nr}"] = source
        for omega in (1.0, 1.3, 1.6):
            for sweeps in (1, 2, 50):
                res = mg.mg_smooth(psi0.copy(), source, rr, dr, dz, omega, sweeps)
                key = f"out_{nz}x{nr}_w{omega}_s{sweeps}"
                out[key] = res
                cases.append([nz, nr, omega, sweeps])
    out["cases"] = np.array(cases)
    save("smooth.npz", **out)


# ── (b) residual / restrict / prolong / vcycle  core/multigrid_solve.py:57-335 ─
def gen_operators():
    out = {}
    shapes = [(17, 17), (33, 33), (65, 65), (16, 16), (128, 128), (65, 33), (20, 37), (6, 7), (5, 5), (9, 9)]
    out["shapes"] = np.array(shapes)
    rng = np.random.default_rng(7)
    for (nz, nr) in shapes:
        tag = f"{nz}x{nr}"
        psi0, source, rr, r, z = bench_problem(nz, nr)
        dr = float(r[1] - r[0])
        dz = float(z[1] - z[0])
        psi = psi0 + 0.01 * np.sin(rr) * np.cos(0.3 * zz_of(z, nr))
        out[f"psi_{tag}"] = psi
        out[f"src_{tag}"] = source
        out[f"res_{tag}"] = mg.mg_residual(psi, source, rr, dr, dz)
        out[f"linf_{tag}"] = np.array(mg.residual_linf(psi, source, rr, dr, dz))
        f = rng.normal(size=(nz, nr))
        out[f"fine_{tag}"] = f
        c = mg.restrict_full_weight(f)
        out[f"restrict_{tag}"] = c
        out[f"restrict_r_{tag}"] = mg.restrict_full_weight(rr)
        out[f"prolong_{tag}"] = mg.prolongate_bilinear(c, nz, nr)
        if min(nz, nr) >= 5:
            for omega in (1.0, 1.6):
                out[f"vcycle_{tag}_w{omega}"] = mg.multigrid_vcycle(psi.copy(), source, rr, dr, dz, omega=omega)
    save("operators.npz", **out)


def zz_of(z, nr):
    return np.repeat(z[:, None], nr, axis=1)


# ── (c) multigrid_solve  core/multigrid_solve.py:352-463 ──────────────────────
def gen_mgsolve():
    out = {}
    # tests/test_multigrid_solve.py:26-44 problem
    for n in (33, 65):
        r, z, rr, zz = grid(n, n, 1.2, 2.2, -0.5, 0.5)
        source = -rr * np.exp(-((rr - 1.7) ** 2 + zz**2) / 0.05)
        psi_bc = np.zeros((n, n))
        psi, res, ncy, conv = mg.multigrid_solve(source, psi_bc, 1.2, 2.2, -0.5, 0.5, n, n, tol=1e-6)
        out[f"gauss{n}_source"] = source
        out[f"gauss{n}_psi"] = psi
        out[f"gauss{n}_stats"] = np.array([res, ncy, float(conv)])
    # Solov'ev gate  validation/validate_grad_shafranov_solovev.py:98-178,353-425
    for n in (33, 65, 97):
        r0, a = 1.7, 0.5
        r_min, r_max, z_min, z_max = r0 - 1.5 * a, r0 + 1.5 * a, -1.5 * a, 1.5 * a
        r, z, rr, zz = grid(n, n, r_min, r_max, z_min, z_max)
        exact = rr**4 / 8.0 + 0.5 * zz**2
        source = rr**2 + 1.0
        psi_bc = np.zeros((n, n))
        psi_bc[0, :], psi_bc[-1, :], psi_bc[:, 0], psi_bc[:, -1] = exact[0, :], exact[-1, :], exact[:, 0], exact[:, -1]
        psi, res, ncy, conv = mg.multigrid_solve(source, psi_bc, r_min, r_max, z_min, z_max, n, n, tol=1e-9, max_cycles=200)
        out[f"solovev{n}_extent"] = np.array([r_min, r_max, z_min, z_max])
        out[f"solovev{n}_psi"] = psi
        out[f"solovev{n}_stats"] = np.array([res, ncy, float(conv)])
        if n == 65:
            psi2, res2, ncy2, conv2 = mg.multigrid_solve(source, psi_bc, r_min, r_max, z_min, z_max, n, n, tol=1e-9,
                                                        max_cycles=200, omega=1.6, pre_smooth=2, post_smooth=1)
            out["solovev65_w16_psi"] = psi2
            out["solovev65_w16_stats"] = np.array([res2, ncy2, float(conv2)])
    save("mgsolve.npz", **out)


# ── (d) gs_solve_np  core/jax_gs_solver.py:126-167; validation/polyglot/gs_picard_reference.toml ─
def gen_polyglot():
    kw = dict(R_min=1.0, R_max=3.0, Z_min=-1.2, Z_max=1.2, Ip_target=1.0e6, mu0=1.2566370614359173e-6,
              n_picard=8, n_jacobi=16, alpha=0.1, omega_j=0.6666666666666666, beta_mix=0.5)
    out = {}
    for n in (17, 65):
        out[f"psi{n}"] = gs_solve_np(NR=n, NZ=n, **kw)
    out["params"] = np.array([kw[k] for k in ("R_min", "R_max", "Z_min", "Z_max", "Ip_target", "mu0", "n_picard",
                                             "n_jacobi", "alpha", "omega_j", "beta_mix")])
    # lane diagnostics (core/jax_gs_solver.py:196-316) on the solved fields and on a seeded random field
    box = (kw["R_min"], kw["R_max"], kw["Z_min"], kw["Z_max"])
    rng = np.random.default_rng(1305)
    fields = {"s17": out["psi17"], "s65": out["psi65"], "rnd": rng.normal(size=(29, 41)) * 0.3}
    out["rnd_psi"] = fields["rnd"]
    for tag, f in fields.items():
        nz, nr = f.shape
        out[f"ds_{tag}"] = gs_delta_star_np(f, *box)
        out[f"jd_{tag}"] = gs_toroidal_current_density_np(f, *box, mu0=kw["mu0"])
        out[f"it_{tag}"] = np.array(gs_total_toroidal_current_np(f, *box, mu0=kw["mu0"]))
        m = gs_equation_residual_np(f, *box, nr, nz, kw["Ip_target"], kw["mu0"], kw["beta_mix"])
        out[f"er_{tag}"] = np.array([m["abs_max"], m["relative_max"], m["relative_l2"]])
    save("polyglot.npz", **out)


# ── configs ──────────────────────────────────────────────────────────────────
def iter_cfg(n, method="multigrid", omega=1.0, tol=1e-8, max_iter=1000, alpha=0.12, hmode=False):
    cfg = build_config(TokamakCase("ITER-like", 6.2, 2.0, 5.3, 15.0, 1.7, n, n))
    cfg["solver"].update(solver_method=method, sor_omega=omega, convergence_threshold=tol, max_iterations=max_iter,
                         relaxation_factor=alpha)
    if hmode:
        cfg["physics"]["profiles"] = {"mode": "h-mode", "p_prime": {"ped_top": 0.9}, "ff_prime": {"core_alpha": 0.25}}
    return cfg


def d3d_cfg(n, tol=1e-8, max_iter=1000):
    cfg = json.loads((REF / "validation" / "diiid_config.json").read_text())
    cfg.pop("_reference", None)
    cfg["grid_resolution"] = [n, n]
    cfg["solver"].update(solver_method="multigrid", sor_omega=1.0, convergence_threshold=tol, max_iterations=max_iter,
                         relaxation_factor=0.12)
    cfg["solver"].pop("fail_on_diverge", None)
    return cfg


def make_kernel(cfg):
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False, dir="/tmp") as f:
        json.dump(cfg, f)
        path = f.name
    k = FusionKernel(path)
    os.unlink(path)
    assert not k.hpc.is_available()
    return k


def traced_solve(k, **kw):
    """Run solve_equilibrium recording (psi_axis, psi_bnd) per iteration via the source-update hook."""
    rec = []
    orig = k.update_plasma_source_nonlinear

    def hook(pa, pb):
        rec.append((pa, pb))
        return orig(pa, pb)

    k.update_plasma_source_nonlinear = hook
    res = k.solve_equilibrium(**kw)
    k.update_plasma_source_nonlinear = orig
    return res, np.array(rec)


# ── (e) Picard  core/fusion_kernel_newton_solver.py:390-615 ───────────────────
def gen_picard():
    out = {}
    cases = {
        "iter33_mg_w10": iter_cfg(33),
        "iter65_mg_w10": iter_cfg(65),
        "iter33_mg_w16": iter_cfg(33, omega=1.6),
        "iter33_sor": iter_cfg(33, method="sor", omega=1.6, tol=1e-4, max_iter=120),
        "iter33_jacobi": iter_cfg(33, method="jacobi", tol=1e-4, max_iter=60),
        "iter33_hmode": iter_cfg(33, hmode=True),
        "d3d33_mg": d3d_cfg(33),
        "iter32_even_mg": iter_cfg(32, tol=1e-6),
    }
    cfgs = {}
    for name, cfg in cases.items():
        k = make_kernel(cfg)
        res, rec = traced_solve(k)
        cfgs[name] = cfg
        out[f"{name}_psi"] = res["psi"]
        out[f"{name}_jphi"] = k.J_phi
        out[f"{name}_axis_bnd"] = rec
        out[f"{name}_diff"] = np.array(res["residual_history"])
        out[f"{name}_gs"] = np.array(res["gs_residual_history"])
        out[f"{name}_stats"] = np.array([float(res["converged"]), res["iterations"], res["residual"], res["gs_residual"],
                                         res["gs_residual_best"]])
        print(name, res["iterations"], res["converged"], res["residual"])
    out["_configs"] = np.array(json.dumps(cfgs))
    save("picard.npz", **out)


# ── Anderson-accelerated Picard  core/fusion_kernel_newton_solver.py:540-550, iterative_solver.py:248-314 ──
def gen_anderson():
    out, cfgs = {}, {}
    cases = {
        "iter33_anderson": iter_cfg(33, method="anderson", omega=1.6, tol=1e-6, max_iter=150),
        "iter65_anderson_m3": iter_cfg(65, method="anderson", omega=1.3, tol=1e-5, max_iter=100),
    }
    cases["iter65_anderson_m3"]["solver"]["anderson_depth"] = 3
    for name, cfg in cases.items():
        k = make_kernel(cfg)
        res, rec = traced_solve(k)
        cfgs[name] = cfg
        out[f"{name}_psi"] = res["psi"]
        out[f"{name}_axis_bnd"] = rec
        out[f"{name}_diff"] = np.array(res["residual_history"])
        out[f"{name}_gs"] = np.array(res["gs_residual_history"])
        out[f"{name}_stats"] = np.array([float(res["converged"]), res["iterations"], res["residual"], res["gs_residual"],
                                         res["gs_residual_best"]])
        print(name, res["iterations"], res["converged"], res["residual"])
    out["_configs"] = np.array(json.dumps(cfgs))
    save("anderson.npz", **out)


# ── (j,k) unit pieces: axis / x-point / source / jacobi / sor ────────────────
def gen_pieces():
    out = {}
    cfg = iter_cfg(33, max_iter=12, tol=1e-12)
    k = make_kernel(cfg)
    k.solve_equilibrium()
    psi = k.Psi.copy()
    out["psi"] = psi
    iz, ir, pa = k._find_magnetic_axis()
    (rx, zx), px = k.find_x_point(psi)
    out["axis"] = np.array([iz, ir, pa])
    out["xpoint"] = np.array([rx, zx, px])
    j = k.update_plasma_source_nonlinear(pa, px).copy()
    out["jphi_lmode"] = j
    k.profile_mode = "h-mode"
    k.ped_params_p.update({"ped_top": 0.9})
    out["jphi_hmode"] = k.update_plasma_source_nonlinear(pa, px).copy()
    src = -1.0 * k.RR * j
    out["source"] = src
    out["jacobi"] = k._jacobi_step(psi, src)
    out["sor_w16"] = k._sor_step(psi, src, omega=1.6)
    out["gs_rms"] = np.array(compute_gs_residual_rms(k, src))
    out["dims"] = np.array([k.R[0], k.R[-1], k.Z[0], k.Z[-1], k.dR, k.dZ, cfg["dimensions"]["Z_min"]])
    # x-point fallback (Z_min >= 0 -> empty mask) is a host-logic path, tested without a fixture
    save("pieces.npz", **out)


# ── (f) Green's functions  core/fusion_kernel.py:218-251; core/fusion_kernel_free_boundary.py:31-93 ─
def gen_green():
    out = {}
    m = np.concatenate([np.array([1e-12, 1e-9, 1e-6, 1e-3, 0.01]), np.linspace(0.05, 0.95, 19),
                        1.0 - np.array([1e-2, 1e-3, 1e-6, 1e-9, 1e-12])])
    out["m"] = m
    out["ellipk"] = ellipk(m)
    out["ellipe"] = ellipe(m)
    cfg = iter_cfg(33)
    k = make_kernel(cfg)
    out["vacuum33"] = k.calculate_vacuum_field()
    coils = CoilSet(positions=[(c["r"], c["z"]) for c in cfg["coils"]],
                    currents=np.array([c["current"] for c in cfg["coils"]], dtype=float), turns=[1, 2, 3, 1, 10])
    out["ext33"] = compute_external_flux(k, coils)
    out["coils"] = np.array([[c["r"], c["z"], c["current"]] for c in cfg["coils"]])
    out["turns"] = np.array(coils.turns, dtype=float)
    pts = np.array([[9.2, 4.08, 6.2, 0.0], [9.2, 4.08, 9.2, 4.08], [2.25, 0.0, 10.2, -6.8], [5.2, 4.488, 5.3, 4.4],
                    [0.42, 0.0, 0.8, 1.8]])
    out["green_pts"] = pts
    out["green_vals"] = np.array([green_function(*p) for p in pts])
    kd = make_kernel(d3d_cfg(33))
    out["vacuum33_d3d"] = kd.calculate_vacuum_field()
    save("green.npz", **out)


# ── (h) free boundary  core/fusion_kernel_free_boundary.py:623-739 ────────────
def gen_free_boundary():
    out = {}
    cfgs = {}
    for name, cfg in {"iter33": iter_cfg(33), "d3d33": d3d_cfg(33), "iter65": iter_cfg(65)}.items():
        k = make_kernel(cfg)
        its = []
        orig = k.solve_equilibrium

        def wrapped(**kw):
            r = orig(**kw)
            its.append(r["iterations"])
            return r

        k.solve_equilibrium = wrapped
        coils = CoilSet(positions=[(c["r"], c["z"]) for c in cfg["coils"]],
                        currents=np.array([c["current"] for c in cfg["coils"]], dtype=float), turns=[1] * len(cfg["coils"]))
        r = k.solve_free_boundary(coils, max_outer_iter=20, tol=1e-4, optimize_shape=False)
        cfgs[name] = cfg
        out[f"{name}_psi"] = k.Psi
        out[f"{name}_stats"] = np.array([r["outer_iterations"], r["final_diff"]] + its, dtype=float)
        print(name, r["outer_iterations"], r["final_diff"], its)
    out["_configs"] = np.array(json.dumps(cfgs))
    save("free_boundary.npz", **out)


# ── BASELINE config 3, second variant: solve_free_boundary(optimize_shape=True) with 8 isoflux control points on the DIII-D-like
#    grid (validation/diiid_config.json; ellipse R0 1.67 m, a 0.6 m, kappa 1.8)  core/fusion_kernel_free_boundary.py:664-705 ─
def gen_shape_d3d():
    out = {}
    cfg = d3d_cfg(129, tol=1e-6, max_iter=400)
    cfg["physics"]["vacuum_permeability"] = 4e-7 * np.pi  # SI-consistent units, as in gen_shape
    cfg["physics"]["plasma_current_target"] = 1.6e6
    for c in cfg["coils"]:
        c["current"] = c["current"] * 1.0e6
    k = make_kernel(cfg)
    th = np.linspace(0.0, 2.0 * np.pi, 8, endpoint=False)
    pts = np.column_stack([1.67 + 0.6 * np.cos(th), 1.8 * 0.6 * np.sin(th)])
    coils = CoilSet(positions=[(c["r"], c["z"]) for c in cfg["coils"]],
                    currents=np.array([c["current"] for c in cfg["coils"]], dtype=float), turns=[1] * len(cfg["coils"]),
                    current_limits=np.full(len(cfg["coils"]), 5.0e7), target_flux_points=pts)
    r = k.solve_free_boundary(coils, max_outer_iter=4, tol=1e-6, optimize_shape=True)
    out["psi"] = k.Psi
    out["points"] = pts
    out["currents"] = r["coil_currents"]
    out["stats"] = np.array([r["outer_iterations"], r["final_diff"], r["shape_optimization"]["flux_rmse"]])
    out["_configs"] = np.array(json.dumps({"shape_d3d129": cfg}))
    print("shape d3d", r["outer_iterations"], r["final_diff"], r["coil_currents"], r["shape_optimization"])
    save("shape_d3d.npz", **out)


# ── shape optimisation  core/fusion_kernel_free_boundary.py:491-559,583-605,664-705 ─
def gen_shape():
    out = {}
    cfg = iter_cfg(33, tol=1e-6, max_iter=400)
    # SI-consistent units so the coil flux matters: mu0 = 4*pi*1e-7, currents in ampere-turns
    cfg["physics"]["vacuum_permeability"] = 4e-7 * np.pi
    cfg["physics"]["plasma_current_target"] = 15.0e6
    for c in cfg["coils"]:
        c["current"] = c["current"] * 1.0e6
    k = make_kernel(cfg)
    th = np.linspace(0.0, 2.0 * np.pi, 8, endpoint=False)
    pts = np.column_stack([6.2 + 1.6 * np.cos(th), 2.6 * np.sin(th)])
    coils = CoilSet(positions=[(c["r"], c["z"]) for c in cfg["coils"]],
                    currents=np.array([c["current"] for c in cfg["coils"]], dtype=float), turns=[1] * len(cfg["coils"]),
                    current_limits=np.full(len(cfg["coils"]), 5.0e7), target_flux_points=pts)
    r = k.solve_free_boundary(coils, max_outer_iter=4, tol=1e-6, optimize_shape=True)
    out["psi"] = k.Psi
    out["points"] = pts
    out["currents"] = r["coil_currents"]
    out["stats"] = np.array([r["outer_iterations"], r["final_diff"], r["shape_optimization"]["flux_rmse"]])
    out["_configs"] = np.array(json.dumps({"shape33": cfg}))
    print("shape", r["outer_iterations"], r["final_diff"], r["coil_currents"])
    save("shape.npz", **out)


# ── GEQDSK  core/eqdsk.py:128-196,349-541; data file validation/reference_data/diiid/diiid_lmode_1MA.geqdsk ──
def gen_eqdsk():
    from scpn_fusion.core.eqdsk import read_geqdsk

    src = REF / "validation" / "reference_data" / "diiid" / "diiid_lmode_1MA.geqdsk"
    eq = read_geqdsk(src)
    cfg = eq.to_config("diiid_lmode_1MA")
    cfg["solver"].update(solver_method="sor", sor_omega=1.6, max_iterations=400, convergence_threshold=1e-4, relaxation_factor=0.1)
    k = make_kernel({key: val for key, val in cfg.items()})
    res = k.solve_equilibrium()
    out = {"scalars": np.array([eq.nw, eq.nh, eq.rdim, eq.zdim, eq.rcentr, eq.rleft, eq.zmid, eq.rmaxis, eq.zmaxis, eq.simag, eq.sibry,
                                eq.bcentr, eq.current]),
           "fpol": eq.fpol, "pres": eq.pres, "ffprime": eq.ffprime, "pprime": eq.pprime, "qpsi": eq.qpsi, "psirz": eq.psirz,
           "rbdry": eq.rbdry, "zbdry": eq.zbdry, "rlim": eq.rlim, "zlim": eq.zlim,
           "solve_psi": res["psi"], "solve_stats": np.array([float(res["converged"]), res["iterations"], res["residual"]]),
           "_configs": np.array(json.dumps({"diiid_lmode_1MA": cfg}))}
    sp = REF / "validation" / "reference_data" / "sparc" / "sparc_1305.eqdsk"  # public SPARC file (MIT, see sparc/LICENSE)
    raw, ad = read_geqdsk(sp), read_geqdsk(sp, source_convention_mode="public_sparc_named_adapter")
    none = read_geqdsk(src, source_convention_mode="public_sparc_named_adapter")
    out.update(sparc_scalars=np.array([raw.nw, raw.nh, raw.rdim, raw.zdim, raw.rcentr, raw.rleft, raw.zmid, raw.rmaxis, raw.zmaxis,
                                       raw.simag, raw.sibry, raw.bcentr, raw.current]),
               sparc_psirz=raw.psirz, sparc_ffprime_raw=raw.ffprime, sparc_pprime_raw=raw.pprime, sparc_ffprime_adapted=ad.ffprime,
               sparc_pprime_adapted=ad.pprime, sparc_rbdry=raw.rbdry, sparc_zlim=raw.zlim,
               sparc_meta=np.array(json.dumps({"adapted": [ad.source_convention, ad.source_convention_adapter,
                                                           ad.source_convention_adapter_pass, ad.source_convention_metadata],
                                               "no_match": [none.source_convention, none.source_convention_adapter,
                                                            none.source_convention_adapter_pass, none.source_convention_metadata]})))
    print("eqdsk", eq.nw, eq.nh, eq.current, res["iterations"], res["converged"], res["residual"], "sparc", raw.nw, raw.nh)
    save("eqdsk.npz", **out)


# ── (g) compiled hpc/solver.cpp through oracle/_ref  hpc/solver.cpp:200-335 ──
def gen_fastsolver():
    sys.path.insert(0, str(OUT.parents[1]))
    from oracle.oracle import RefFastSolver

    out = {}
    nr, nz = 33, 29
    rng = np.random.default_rng(11)
    j = rng.normal(size=(nz, nr))
    s = RefFastSolver(nr, nz, 1.0, 3.0, -1.5, 1.5)
    out["j"] = j
    out["extent"] = np.array([1.0, 3.0, -1.5, 1.5])
    out["run50_a"] = s.run_step(j, 50)
    out["run50_b"] = s.run_step(j, 50)  # warm state persists in the handle
    s.set_boundary(0.25)
    out["run7_bc"] = s.run_step(j, 7)
    psi, n, fd = s.run_step_converged(j, 500, 1.5, 1e-9)
    out["conv_psi"] = psi
    out["conv_stats"] = np.array([n, fd])
    s.close()
    save("fastsolver.npz", **out)


# ── (a8, optional) X-point by saddle detection  core/fusion_kernel.py:284-329 ──
def gen_saddle():
    out, cfgs = {}, {}
    rng = np.random.default_rng(17)
    # unit: find_x_point with the saddle test on assorted flux maps (solved state, perturbed, monotone -> no saddle, NaN hole)
    cfg = iter_cfg(33, tol=1e-6, max_iter=200)
    cfg["solver"]["xpoint_use_saddle_detection"] = True
    k = make_kernel(cfg)
    k.solve_equilibrium()
    base = k.Psi.copy()
    fields = {"solved": base, "noisy": base + 0.02 * rng.normal(size=base.shape), "monotone": 0.3 * k.RR + 0.1 * k.ZZ,
              "bowl": (k.RR - 6.0) ** 2 + (k.ZZ + 5.0) ** 2, "saddle": (k.RR - 6.0) ** 2 - (k.ZZ + 5.0) ** 2}
    hole = base.copy()
    hole[4, 7] = np.nan
    fields["nanhole"] = hole
    for name, f in fields.items():
        (rx, zx), px = k.find_x_point(f)
        out[f"unit_{name}_psi"] = f
        out[f"unit_{name}_x"] = np.array([rx, zx, px])
        k.cfg["solver"]["xpoint_use_saddle_detection"] = False
        (r0, z0), p0 = k.find_x_point(f)
        k.cfg["solver"]["xpoint_use_saddle_detection"] = True
        out[f"unit_{name}_x_plain"] = np.array([r0, z0, p0])
        print(name, (rx, zx, px), (r0, z0, p0))
    out["unit_dims"] = np.array([k.R[0], k.R[-1], k.Z[0], k.Z[-1], k.dR, k.dZ, cfg["dimensions"]["Z_min"]])
    for name, c in {"iter33_saddle": iter_cfg(33), "iter65_saddle": iter_cfg(65), "d3d33_saddle": d3d_cfg(33)}.items():
        c["solver"]["xpoint_use_saddle_detection"] = True
        kk = make_kernel(c)
        res, rec = traced_solve(kk)
        cfgs[name] = c
        out[f"{name}_psi"], out[f"{name}_axis_bnd"] = res["psi"], rec
        out[f"{name}_diff"] = np.array(res["residual_history"])
        out[f"{name}_stats"] = np.array([float(res["converged"]), res["iterations"], res["residual"]])
        print(name, res["iterations"], res["converged"])
    out["_configs"] = np.array(json.dumps(cfgs))
    save("saddle.npz", **out)


# ── (f3) field products: psi sampling, B field, diagnostic response matrices, probe-based reconstruction ──
def gen_products():
    """core/fusion_kernel.py:450-456; core/fusion_kernel_free_boundary.py:137-159, 162-270, 285-488, 562-580."""
    from scpn_fusion.core import fusion_kernel_free_boundary as fb

    out = {}
    cfg = iter_cfg(33, tol=1e-6, max_iter=300)
    k = make_kernel(cfg)
    k.solve_equilibrium()
    out["psi"], out["b_r"], out["b_z"] = k.Psi.copy(), k.B_R.copy(), k.B_Z.copy()
    rng = np.random.default_rng(5)
    d = cfg["dimensions"]
    pts = np.column_stack([rng.uniform(d["R_min"] - 0.3, d["R_max"] + 0.3, 40), rng.uniform(d["Z_min"] - 0.3, d["Z_max"] + 0.3, 40)])
    pts[:6] = [[k.R[0], k.Z[0]], [k.R[-1], k.Z[-1]], [k.R[3], k.Z[7]], [k.R[3] + 1e-13, k.Z[7] - 1e-13], [d["R_min"] - 1.0, 0.0],
               [6.2, d["Z_max"] + 2.0]]
    out["sample_pts"] = pts
    out["sample_psi"] = fb.sample_flux_at_points(k, pts)
    pos = [(c["r"], c["z"]) for c in cfg["coils"]]
    cur = np.array([c["current"] for c in cfg["coils"]], dtype=float) * 1e6
    turns = [1, 2, 1, 3, 1][: len(pos)] + [1] * max(0, len(pos) - 5)
    coils = CoilSet(positions=pos, currents=cur, turns=turns, current_limits=np.full(len(pos), 5.0e7))
    th = np.linspace(0.0, 2.0 * np.pi, 12, endpoint=False)
    flux_pts = np.column_stack([6.2 + 2.4 * np.cos(th), 3.8 * np.sin(th)])
    b_pts = np.column_stack([6.2 + 2.6 * np.cos(th[::2] + 0.2), 4.0 * np.sin(th[::2] + 0.2)])
    dirs = ["R", "Z", "R", "Z", "z", "r"]
    resp = fb.build_magnetic_probe_response_matrix(k, coils, flux_points=flux_pts, b_probe_points=b_pts, b_probe_directions=dirs)
    out["coil_pos"], out["coil_cur"], out["coil_turns"] = np.array(pos), cur, np.array(turns, dtype=float)
    out["flux_pts"], out["b_pts"], out["b_dirs"] = flux_pts, b_pts, np.array([0 if x.upper() == "R" else 1 for x in dirs])
    out["response"] = resp
    out["mutual"] = fb.build_mutual_inductance_matrix(k, coils, flux_pts)
    true_cur = cur * (1.0 + 0.2 * rng.uniform(-1, 1, cur.size))
    meas = resp @ true_cur
    sigma = np.concatenate([np.full(len(flux_pts), 1e-3), np.full(len(b_pts), 1e-4)])
    rec = fb.reconstruct_coil_currents_from_magnetic_probes(
        k, coils, flux_points=flux_pts, flux_measurements=meas[: len(flux_pts)], b_probe_points=b_pts, b_probe_directions=dirs,
        b_probe_measurements=meas[len(flux_pts):], measurement_sigma=sigma, tikhonov_alpha=1e-12)
    out["true_cur"], out["meas"], out["sigma"] = true_cur, meas, sigma
    out["rec_currents"], out["rec_residual"] = rec["coil_currents"], rec["residual"]
    out["rec_stats"] = np.array([rec["residual_rms"], rec["weighted_residual_rms"], rec["response_rank"], rec["response_condition"],
                                 rec["active_bounds"]])
    lim = np.column_stack([6.2 + 3.2 * np.cos(th), 5.0 * np.sin(th)])
    bnd = np.column_stack([6.2 + 1.8 * np.cos(th), 3.0 * np.sin(th)])
    target = rng.normal(size=len(bnd)) * 1e-2
    diag = fb.reconstruct_boundary_flux_from_coils(k, coils, boundary_points=bnd, limiter_points=lim, axis_point=np.array([6.2, 0.0]),
                                                   x_points=np.array([[5.0, -3.4], [5.0, 3.4]]), target_flux=target)
    out["bnd_pts"], out["lim_pts"], out["bnd_target"] = bnd, lim, target
    out["bnd_flux"], out["lim_flux"], out["x_flux"] = diag["reconstructed_flux"], diag["limiter_flux"], diag["x_point_flux"]
    out["bnd_stats"] = np.array([diag["response_rank"], diag["min_limiter_distance_m"], diag["boundary_containment_fraction"],
                                 diag["axis_flux"], diag["x_point_flux_span"], diag["x_point_pair_symmetry_abs_error"], diag["rmse"],
                                 diag["max_abs_error"]])
    out["_configs"] = np.array(json.dumps({"iter33": cfg}))
    print("products", resp.shape, rec["coil_currents"], rec["active_bounds"], diag["rmse"])
    save("products.npz", **out)


# ── (f4) transport coupling: map_profiles_to_2d, solve_banded tridiagonal, CurrentDiffusionSolver.step ──
def gen_transport():
    """core/integrated_transport_solver_coupling.py:39-73; core/current_diffusion.py:175-233, 262-353."""
    from scipy.linalg import solve_banded

    from scpn_fusion.core.current_diffusion import CurrentDiffusionSolver, neoclassical_resistivity, q_from_psi
    from scpn_fusion.core.integrated_transport_solver import TransportSolver

    out = {}
    cfg = iter_cfg(33, tol=1e-6, max_iter=300)
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False, dir="/tmp") as f:
        json.dump(cfg, f)
        path = f.name
    ts = TransportSolver(path, nr=50)
    os.unlink(path)
    ts.solve_equilibrium()
    rng = np.random.default_rng(4)
    for tag, scale in (("a", 1.0), ("b", 2.5)):
        ts.Te = scale * (1 - ts.rho**2) + 0.05
        ts.Ti = 0.8 * scale * (1 - ts.rho**2) ** 1.5 + 0.05
        ts.ne = 5.0 * (1 - 0.9 * ts.rho**2) ** 0.5
        r0 = (cfg["dimensions"]["R_min"] + cfg["dimensions"]["R_max"]) / 2.0
        b_pol = (1.256e-6 * cfg["physics"]["plasma_current_target"]) / (
            2 * np.pi * 0.5 * (cfg["dimensions"]["R_max"] - cfg["dimensions"]["R_min"]))
        jbs = ts.calculate_bootstrap_current(r0, b_pol)
        psi_in = ts.Psi.copy()
        ts.map_profiles_to_2d()
        out[f"map_{tag}_psi"], out[f"map_{tag}_rho"] = psi_in, ts.rho.copy()
        out[f"map_{tag}_p1d"], out[f"map_{tag}_jbs1d"] = ts.ne * (ts.Ti + ts.Te), np.asarray(jbs, dtype=float)
        out[f"map_{tag}_p2d"], out[f"map_{tag}_jphi"] = ts.Pressure_2D.copy(), ts.J_phi.copy()
        ts.Psi = ts.Psi + 0.05 * rng.normal(size=ts.Psi.shape)  # second case: perturbed flux map (different topology branch)
    out["_configs"] = np.array(json.dumps({"iter33": cfg}))
    # np.interp unit vectors incl. the exact-knot, out-of-range and NaN branches
    xp = np.linspace(0, 1, 17)
    fp = rng.normal(size=17)
    x = np.concatenate([rng.uniform(-0.2, 1.2, 200), xp, [np.nan, 0.0, 1.0, -1.0, 2.0]])
    out["interp_x"], out["interp_xp"], out["interp_fp"], out["interp_y"] = x, xp, fp, np.interp(x, xp, fp)
    # tridiagonal systems: diagonally dominant (no interchanges) and general (row interchanges), n = 2, 3, 50
    tri = {}
    for name, n, dom in (("dom50", 50, True), ("piv50", 50, False), ("piv3", 3, False), ("piv2", 2, False), ("dom7", 7, True)):
        dl, du = rng.normal(size=n - 1), rng.normal(size=n - 1)
        d = rng.normal(size=n) + (4.0 if dom else 0.0) * np.sign(rng.normal(size=n))
        b = rng.normal(size=n)
        ab = np.zeros((3, n))
        ab[0, 1:], ab[1, :], ab[2, :-1] = du, d, dl
        out[f"tri_{name}_dl"], out[f"tri_{name}_d"], out[f"tri_{name}_du"], out[f"tri_{name}_b"] = dl, d, du, b
        out[f"tri_{name}_x"] = solve_banded((1, 1), ab, b)
    # current diffusion: 6 steps
    rho = np.linspace(0, 1, 40)
    cd = CurrentDiffusionSolver(rho, R0=6.2, a=2.0, B0=5.3)
    te, ne = 8.0 * (1 - rho**2) + 0.1, 9.0 * (1 - 0.8 * rho**2)
    jbs, jcd = 2e5 * rho * (1 - rho), 1e5 * np.exp(-((rho - 0.3) / 0.1) ** 2)
    out["cd_rho"], out["cd_psi0"], out["cd_te"], out["cd_ne"], out["cd_jbs"], out["cd_jcd"] = rho, cd.psi.copy(), te, ne, jbs, jcd
    out["cd_q0"] = q_from_psi(rho, cd.psi, 6.2, 2.0, 5.3)
    out["cd_eta"] = np.array([neoclassical_resistivity(float(te[i]), float(ne[i]), 1.7, rho[i] * 2.0 / 6.2, float(out["cd_q0"][i]), 6.2)
                              for i in range(rho.size)])
    hist = []
    for _ in range(6):
        hist.append(cd.step(0.5, te, ne, 1.7, jbs, jcd).copy())
    out["cd_hist"] = np.array(hist)
    # non-adiabatic flux carrier (benchmarks/bench_current_diffusion_nonadiabatic.py:43-72 case + space-dependent / infinite tau)
    from scpn_fusion.core.current_diffusion import solve_flux_evolution_nonadiabatic

    for tag, n_rho, n_steps, tau in (("bench", 64, 128, lambda t, g: 2.0e-6), ("inf", 33, 20, lambda t, g: np.inf),
                                     ("rho", 48, 40, lambda t, g: np.where(g < 0.5, 1.0e-6 * (1 + g), np.inf))):
        rho = np.linspace(0.0, 1.0, n_rho)
        psi0 = 0.08 - 0.02 * rho**2
        dt = 1.0e-8
        tr = solve_flux_evolution_nonadiabatic(
            rho, psi0, tau_psi_fn=tau, R_null_t=lambda t: 0.18, E_theta_t=lambda t, g: 25.0 + t / (n_steps * dt) + g,
            eta_spitzer_fn=lambda g: np.full_like(g, 2.5e-6), J_theta_t=lambda t, g: 1.0e5 * (1.0 - 0.3 * g), dt=dt, n_steps=n_steps)
        out[f"fx_{tag}_psi"], out[f"fx_{tag}_rate"], out[f"fx_{tag}_source"] = tr.psi, tr.damping_rate, tr.source
        out[f"fx_{tag}_inc"], out[f"fx_{tag}_drop"], out[f"fx_{tag}_resid"] = tr.source_increment, tr.damping_decrement, tr.update_residual
        out[f"fx_{tag}_hall"], out[f"fx_{tag}_loss"], out[f"fx_{tag}_time"] = tr.hall_drive, tr.resistive_loss, tr.time_s
    print("transport", out["map_a_jphi"].sum(), out["cd_hist"][-1][:3], out["fx_bench_psi"][-1][:2])
    save("transport.npz", **out)


# ── (f1) Newton-Kantorovich + GMRES  core/fusion_kernel_newton_solver.py:104-372 ──
def gen_newton():
    """Full Newton dispatch runs (residual histories, psi) + unit vectors for the profile Jacobian,
    one scipy GMRES solve on the Newton operator, and LAPACK dlartg samples."""
    from scipy.linalg.lapack import dlartg
    from scipy.sparse.linalg import LinearOperator, gmres

    from scpn_fusion.core.fusion_kernel_solver_runtime import apply_gs_operator, compute_profile_jacobian

    out, cfgs = {}, {}
    mock = {  # tests/test_newton_equilibrium.py:21-38 MOCK_CONFIG
        "reactor_name": "Newton-Test", "grid_resolution": [20, 20],
        "dimensions": {"R_min": 4.0, "R_max": 8.0, "Z_min": -4.0, "Z_max": 4.0},
        "physics": {"plasma_current_target": 15.0, "vacuum_permeability": 1.0},
        "coils": [{"name": "PF1", "r": 3.0, "z": 7.6, "current": 8.0}, {"name": "PF2", "r": 8.2, "z": 6.7, "current": -1.32},
                  {"name": "CS", "r": 1.7, "z": 0.0, "current": 0.15}],
        "solver": {"max_iterations": 60, "convergence_threshold": 1e-3, "relaxation_factor": 0.1, "solver_method": "newton"},
    }

    def variant(**solver):
        c = json.loads(json.dumps(mock))
        c["solver"].update(solver)
        return c

    hm = variant()
    hm["physics"]["profiles"] = {"mode": "h-mode",
                                 "p_prime": {"ped_top": 0.92, "ped_width": 0.05, "ped_height": 1.0, "core_alpha": 0.3},
                                 "ff_prime": {"ped_top": 0.92, "ped_width": 0.05, "ped_height": 1.0, "core_alpha": 0.3}}
    it33 = iter_cfg(33, method="newton", omega=1.6, tol=1e-6, max_iter=40, alpha=0.1)
    it65 = iter_cfg(65, method="newton", omega=1.6, tol=1e-6, max_iter=30, alpha=0.1)
    it65d = iter_cfg(65, method="newton", omega=1.6, tol=1e-6, max_iter=30, alpha=0.1)
    it65d["solver"]["gmres_preconditioner"] = "diagonal"
    cases = {
        "mock20": variant(),
        "mock20_diag": variant(gmres_preconditioner="diagonal"),
        "mock20_ls": variant(newton_line_search=True, newton_line_search_c=1e-4, newton_line_search_max_backtracks=8),
        "mock20_hmode": hm,
        "mock20_gsreq": variant(require_gs_residual=True, gs_residual_threshold=1e-4, max_iterations=40),
        "iter33": it33,
        "iter65": it65,
        "iter65_diag": it65d,
    }
    for name, cfg in cases.items():
        k = make_kernel(cfg)
        res = k.solve_equilibrium()
        cfgs[name] = cfg
        out[f"{name}_psi"] = res["psi"]
        out[f"{name}_jphi"] = k.J_phi
        out[f"{name}_hist"] = np.array(res["residual_history"])
        out[f"{name}_gs"] = np.array(res["gs_residual_history"])
        out[f"{name}_stats"] = np.array([float(res["converged"]), res["iterations"], res["residual"], res["gs_residual"],
                                         res["gs_residual_best"], res["newton_line_search_attempts"],
                                         res["newton_line_search_accepts"], res["newton_line_search_rejects"],
                                         res["gmres_nonconverged_count"], res["gmres_breakdown_count"], res["gmres_last_info"]])
        print(name, res["iterations"], res["converged"], res["residual"], res["gmres_nonconverged_count"])
    out["_configs"] = np.array(json.dumps(cfgs))

    # unit: profile Jacobian + one GMRES solve on a mid-solve state (33^2 Picard-converged psi)
    k = make_kernel(iter_cfg(33, tol=1e-4, max_iter=200))
    k.solve_equilibrium()
    _, _, pa = k._find_magnetic_axis()
    _, pb = k.find_x_point(k.Psi)
    mu0 = k.cfg["physics"]["vacuum_permeability"]
    dj = compute_profile_jacobian(k, pa, pb, mu0)
    k.external_profile_mode = True
    dj_ext = compute_profile_jacobian(k, pa, pb, mu0)
    k.external_profile_mode = False
    out["jac_psi"], out["jac_jphi"], out["jac_axis_bnd"] = k.Psi.copy(), k.J_phi.copy(), np.array([pa, pb])
    out["jac_dj"], out["jac_dj_ext"] = dj, dj_ext
    out["jac_itarget"] = np.array(k.cfg["physics"]["plasma_current_target"])
    diag_term = -mu0 * k.RR * dj
    NZ, NR = k.Psi.shape
    n_int = (NZ - 2) * (NR - 2)

    def matvec(v):
        v2 = np.zeros((NZ, NR))
        v2[1:-1, 1:-1] = v.reshape(NZ - 2, NR - 2)
        lv = apply_gs_operator(k, v2)
        lv[1:-1, 1:-1] -= diag_term[1:-1, 1:-1] * v2[1:-1, 1:-1]
        return lv[1:-1, 1:-1].ravel()

    rng = np.random.default_rng(7)
    rhs = rng.normal(size=n_int)
    J = LinearOperator((n_int, n_int), matvec=matvec, dtype=np.float64)
    resn = []
    x, info = gmres(J, rhs, maxiter=100, restart=50, atol=1e-8, rtol=1e-6, callback=resn.append, callback_type="pr_norm")
    dl = -2.0 / k.dR**2 - 2.0 / k.dZ**2
    inv_diag = 1.0 / (dl - diag_term[1:-1, 1:-1]).ravel()
    M = LinearOperator((n_int, n_int), matvec=lambda v: inv_diag * v, dtype=np.float64)
    resn_d = []
    xd, info_d = gmres(J, rhs, maxiter=100, restart=50, atol=1e-8, rtol=1e-6, M=M, callback=resn_d.append,
                       callback_type="pr_norm")
    x3, info3 = gmres(J, rhs, maxiter=3, restart=50, atol=1e-8, rtol=1e-6)
    out["gm_diag_term"], out["gm_rhs"], out["gm_r"] = diag_term, rhs, k.RR.copy()
    out["gm_drdz"] = np.array([k.dR, k.dZ])
    out["gm_x"], out["gm_info"], out["gm_prnorm"] = x, np.array(info), np.array(resn)
    out["gm_xd"], out["gm_info_d"], out["gm_prnorm_d"], out["gm_inv_diag"] = xd, np.array(info_d), np.array(resn_d), inv_diag
    out["gm_x3"], out["gm_info3"] = x3, np.array(info3)
    print("gmres", info, len(resn), "diag", info_d, len(resn_d), "3 cycles", info3)

    fg = np.array([[3.0, 4.0], [-3.0, 4.0], [3.0, -4.0], [0.0, 2.0], [0.0, -2.0], [5.0, 0.0], [1e-200, 1e-180], [1e200, -1e190],
                   [1e-320, 1.0], [-2.5, 1e-17], [0.3, 0.7]])
    out["lartg_in"] = fg
    out["lartg_out"] = np.array([dlartg(f, g) for f, g in fg])
    save("newton.npz", **out)


# ── FreeGS pin + strict-parity gate fixtures ───────────────────────────────────────────────────────────────────
# validation/benchmark_freegs_public_example_reconstruction.py:248-270,491-523 and the tracked artifact
# validation/reference_data/full_fidelity_public_artifacts/freegs_public_example_reconstruction_attempt.json
def gen_freegs():
    """(1) The FreeGS 0.8.2 vacuum-psi samples the reference holds (9 points x 2 machines, data only) with the sample points of
    ``_sample_points`` (:248-259), the coil records and the acceptance thresholds stored beside them.  (2) The gate thresholds
    of ``_native_same_case_profile_source_comparison`` (:491-497).  (3) psi of the reference's own NumPy lane
    (``FusionKernel.solve_free_boundary``) for the DIII-D-like config (validation/diiid_config.json) at 65^2 and 129^2: what the
    HIP free-boundary solve is gated against (tests/test_gpu_freegs_gate.py, bench.py ``psi_rmse``)."""
    art = json.loads((REF / "validation" / "reference_data" / "full_fidelity_public_artifacts"
                      / "freegs_public_example_reconstruction_attempt.json").read_text())
    out = {"freegs_version": np.array(art["freegs_version"])}
    for case in art["cases"]:
        cid, dom, vac = case["case_id"], case["domain"], case["vacuum_green_function_comparison"]
        quarter = lambda lo, hi: np.linspace(lo + 0.25 * (hi - lo), hi - 0.25 * (hi - lo), 3)  # noqa: E731
        pts = np.asarray([(float(r), float(z)) for z in quarter(dom["z_min"], dom["z_max"]) for r in quarter(dom["r_min"], dom["r_max"])])
        out[f"{cid}_points"] = pts
        out[f"{cid}_freegs_vacuum_psi"] = np.asarray(vac["freegs_vacuum_psi_sample"], dtype=float)
        out[f"{cid}_native_vacuum_psi"] = np.asarray(vac["native_vacuum_psi_sample"], dtype=float)
        out[f"{cid}_thresholds"] = np.array([vac["thresholds"]["max_abs_error_wb"], vac["thresholds"]["nrmse"]])
        out[f"{cid}_coils"] = np.array(json.dumps(vac["coils"]))
        out[f"{cid}_source_contract"] = np.array(json.dumps(case["source_contract"]))
        gate = case["nonlinear_solve_attempt"]["native_same_case_profile_source_comparison"]["thresholds"]
        out["gate_thresholds"] = np.array(json.dumps(gate))
    cfgs = {}
    for n in (65, 129):
        cfg = d3d_cfg(n)
        k = make_kernel(cfg)
        coils = CoilSet(positions=[(c["r"], c["z"]) for c in cfg["coils"]],
                        currents=np.array([c["current"] for c in cfg["coils"]], dtype=float), turns=[1] * len(cfg["coils"]))
        r = k.solve_free_boundary(coils, max_outer_iter=20, tol=1e-4, optimize_shape=False)
        (xr, xz), psi_x = k.find_x_point(k.Psi)
        iz, ir = np.unravel_index(int(np.argmax(k.Psi)), k.Psi.shape)
        cfgs[f"d3d{n}"] = cfg
        out[f"d3d{n}_psi"] = k.Psi
        out[f"d3d{n}_axis_x"] = np.array([k.R[ir], k.Z[iz], xr, xz, psi_x])
        out[f"d3d{n}_stats"] = np.array([r["outer_iterations"], r["final_diff"]])
        print("native lane", n, r["outer_iterations"], r["final_diff"], (k.R[ir], k.Z[iz]), (xr, xz))
    out["_configs"] = np.array(json.dumps(cfgs))
    save("freegs.npz", **out)
    gen_freegs_native()


def _synthetic_same_case(spec, fg):
    """Inputs of the harness's native lane on one public FreeGS example grid, built WITHOUT FreeGS (not installed): a synthetic
    toroidal current density (Gaussian core + pedestal shoulder, sign and total of the case's plasma current up to a known
    0.2 % closure error), its own free-space flux on the grid's ring from the reference's ``green_function`` (what FreeGS's
    von-Hagenow boundary delivers as ``plasma_psi`` there) and a vacuum flux of four point coils outside the domain.  Layout
    is the harness's: arrays are (len(R), len(Z)), R slow."""
    r = np.linspace(spec.r_min, spec.r_max, spec.nx)
    z = np.linspace(spec.z_min, spec.z_max, spec.ny)
    rc, zc = 0.5 * (spec.xpoints[0][0] + spec.xpoints[1][0]) + 0.12, 0.5 * (spec.xpoints[0][1] + spec.xpoints[1][1])
    a, kappa = 0.30 * (spec.r_max - spec.r_min) / 1.9 * 1.6, 1.35
    rr, zz = np.meshgrid(r, z, indexing="ij")
    rho = np.sqrt(((rr - rc) / a) ** 2 + ((zz - zc) / (kappa * a)) ** 2)
    shape = np.exp(-rho**2 / 0.18) + 0.25 * 0.5 * (1.0 - np.tanh((rho - 0.85) / 0.06))
    shape[rho > 1.25] = 0.0
    shape[0, :] = shape[-1, :] = 0.0
    shape[:, 0] = shape[:, -1] = 0.0
    jtor = shape * (spec.plasma_current_a * (1.0 + 2.0e-3) / fg._integrate_current(r, z, shape))
    d_r, d_z = r[1] - r[0], z[1] - z[0]
    src = [(i, j) for i in range(spec.nx) for j in range(spec.ny) if jtor[i, j] != 0.0]
    plasma = 0.37 * np.sin(3.0 * rr) * np.cos(2.0 * zz)  # interior values: the lane must ignore them
    ring = [(i, j) for i in range(spec.nx) for j in range(spec.ny) if i in (0, spec.nx - 1) or j in (0, spec.ny - 1)]
    for i, j in ring:
        plasma[i, j] = sum(jtor[a_, b_] * green_function(r[a_], z[b_], r[i], z[j]) for a_, b_ in src) * d_r * d_z
    span_r, span_z = spec.r_max - spec.r_min, spec.z_max - spec.z_min
    coils = [(spec.r_min + 0.45 * span_r, spec.z_max + 0.15 * span_z, 0.35), (spec.r_min + 0.45 * span_r, spec.z_min - 0.15 * span_z, 0.35),
             (spec.r_max + 0.12 * span_r, 0.35 * spec.z_max, -0.2), (spec.r_max + 0.12 * span_r, 0.35 * spec.z_min, -0.2)]
    vacuum = np.zeros_like(plasma)
    for cr, cz, frac in coils:
        cur = frac * spec.plasma_current_a
        for i in range(spec.nx):
            for j in range(spec.ny):
                vacuum[i, j] += cur * green_function(cr, cz, r[i], z[j])
    return r, z, jtor, plasma, vacuum


# validation/benchmark_freegs_public_example_reconstruction.py:275-315 (native lane), :318-378 (metric helpers), :431-497 (comparison)
def gen_freegs_native():
    """The strict-parity harness's OWN native lane, called as the harness calls it, on both public-example grids (65 x 65, the
    artifact's domains).  Stored: inputs, ``_native_profile_source_reconstruction``'s flux, and the full metric record of
    ``_native_same_case_profile_source_comparison`` against an "external" flux map = that lane's total flux + a smooth 1e-3
    perturbation + 3e-12 Wb on the ring, so every one of the five gated metrics is non-zero in the fixture."""
    from types import SimpleNamespace

    import validation.benchmark_freegs_public_example_reconstruction as fg

    out = {}
    for spec in fg.PUBLIC_CASES:
        cid = spec.case_id
        r, z, jtor, plasma, vacuum = _synthetic_same_case(spec, fg)
        native_plasma = fg._native_profile_source_reconstruction(r, z, plasma, jtor)
        rr, zz = np.meshgrid(r, z, indexing="ij")
        bump = np.sin(np.pi * (rr - r[0]) / (r[-1] - r[0])) * np.sin(np.pi * (zz - z[0]) / (z[-1] - z[0]))
        total = native_plasma + vacuum
        external = total + 1.0e-3 * (total.max() - total.min()) * bump * np.cos(2.0 * rr)
        for edge in (external[0, :], external[-1, :], external[:, 0], external[:, -1]):
            edge += 3.0e-12
        # the harness takes R_1D / Z_1D / Jtor / plasma_psi from a FreeGS equilibrium object; here they are plain arrays
        eq = SimpleNamespace(R_1D=r, Z_1D=z, Jtor=jtor, plasma_psi=external - vacuum)
        rec = fg._native_same_case_profile_source_comparison(eq, None, spec, external)
        rec.pop("q_profile_sanity")  # FreeGS-only (eq.q)
        out[f"{cid}_r"], out[f"{cid}_z"], out[f"{cid}_jtor"] = r, z, jtor
        out[f"{cid}_boundary_psi"], out[f"{cid}_vacuum_psi"], out[f"{cid}_external_psi"] = plasma, vacuum, external
        out[f"{cid}_native_plasma_psi"] = native_plasma
        # the comparison solves with eq.plasma_psi (= external - vacuum) on the ring: its own solution is stored too
        out[f"{cid}_native_plasma_psi_cmp"] = fg._native_profile_source_reconstruction(r, z, external - vacuum, jtor)
        out[f"{cid}_record"] = np.array(json.dumps(rec))
        # a second "external" map far enough away (6 % of the flux range, skewed in R) that the axis moves and thresholds fail
        ia, ja = np.unravel_index(int(np.argmax(total)), total.shape)
        hill = np.exp(-(((rr - r[min(ia + 4, spec.nx - 2)]) / (3.0 * (r[1] - r[0]))) ** 2 + ((zz - z[min(ja + 3, spec.ny - 2)]) / (3.0 * (z[1] - z[0]))) ** 2))
        far = total + (total.max() - total.min()) * (6.0e-2 * bump * (rr - r[0]) / (r[-1] - r[0]) + 0.15 * hill * bump)
        rec_far = fg._native_same_case_profile_source_comparison(
            SimpleNamespace(R_1D=r, Z_1D=z, Jtor=jtor, plasma_psi=far - vacuum), None, spec, far)
        rec_far.pop("q_profile_sanity")
        out[f"{cid}_external_psi_far"], out[f"{cid}_record_far"] = far, np.array(json.dumps(rec_far))
        out[f"{cid}_native_plasma_psi_far"] = fg._native_profile_source_reconstruction(r, z, far - vacuum, jtor)
        print("  far:", {k: rec_far[k] for k in ("psi_n_rmse", "axis_error_m", "xpoint_psi_n_error_max")})
        out[f"{cid}_spec"] = np.array(json.dumps({"plasma_current_a": spec.plasma_current_a, "xpoints": spec.xpoints}))
        print(cid, {k: rec[k] for k in ("psi_n_rmse", "axis_error_m", "boundary_max_abs_error_wb", "current_closure_relative_error",
                                        "xpoint_psi_n_error_max", "native_plasma_psi_rmse")})
    save("freegs_native.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["smooth", "operators", "mgsolve", "polyglot", "picard", "pieces", "green",
                             "free_boundary", "fastsolver", "anderson", "shape", "shape_d3d", "eqdsk", "newton", "products", "transport", "saddle", "freegs"]  # "freegs" also writes freegs_native.npz
    for w in which:
        globals()[f"gen_{w}"]()

#!/usr/bin/env python3
"""BASELINE config 4: transport + current-diffusion time-stepping loop that re-solves the equilibrium every step, 513x513,
one MI355X, for a resident ensemble.  One coupled step (reference: TransportSolverCouplingMixin.run_self_consistent,
core/integrated_transport_solver_coupling.py:129-147, plus CurrentDiffusionSolver.step, core/current_diffusion.py:277-353):

    1-D profile update (the caller's transport model -- here a prescribed relaxation, host side, O(n_rho) per member)
    -> map_profiles_to_2d (device)  -> solve_equilibrium in external_profile_mode (device, fields stay resident)
    -> relative flux change (host norm of the downloaded psi, as the reference)  -> current-diffusion step (device)

Reports seconds per coupled step for the ensemble and the CPU oracle's time for ONE member doing the same step.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from scpn_fusion_core_amd import transport_coupling as tc  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402
from scpn_fusion_core_amd.cases import iter_like_cfg  # noqa: E402


def coupled_steps(n: int = 513, B: int = 32, steps: int = 4, device: int | None = None, verbose: bool = False, arithmetic: str | None = None) -> dict:
    """`steps` coupled steps (after one warm-up step) of a B-member ensemble on the n x n ITER-like grid; returns the timing rows and
    the mean seconds per coupled step.  Device work only (no oracle)."""
    cfg = iter_like_cfg(n, tol=1e-6, max_iter=400)
    rho = np.linspace(0, 1, 50)
    scale = 1.0 + 0.1 * np.linspace(-1, 1, B)[:, None]
    te = 8.0 * (1 - rho**2)[None] * scale + 0.1
    ne = 9.0 * (1 - 0.8 * rho**2)[None] * np.ones((B, 1))
    cd = tc.CurrentDiffusionSolver(rho, R0=6.2, a=2.0, B0=5.3, batch=B)
    rows = []
    kw = {} if device is None else {"device": device}
    if arithmetic is not None:
        kw["arithmetic"] = arithmetic
    with EnsembleSolver(cfg, batch=B, current_targets=15.0 * scale[:, 0], **kw) as es:
        es.solve_equilibrium()
        for k in range(steps + 1):
            t0 = time.perf_counter()
            te = te + 0.02 * (8.0 * (1 - rho**2)[None] * scale * 1.05 - te)  # stand-in for evolve_profiles
            p1 = ne * (2.0 * te)
            jbs = 1e-3 * np.gradient(p1, rho, axis=1) * rho[None] * -1.0
            t1 = time.perf_counter()
            resid, res = tc.gs_transport_outer_step(es, rho, p1, jbs)
            es.sync()
            t2 = time.perf_counter()
            cd.step(0.5, te, ne, 1.7, 2e5 * rho * (1 - rho), 1e5 * np.exp(-((rho - 0.3) / 0.1) ** 2))
            t3 = time.perf_counter()
            if k:  # step 0 warms buffers
                rows.append({"step": k, "profiles_host_s": t1 - t0, "map_and_resolve_s": t2 - t1, "current_diffusion_s": t3 - t2,
                             "picard_iterations": [min(r["iterations"] for r in res), max(r["iterations"] for r in res)],
                             "psi_residual_max": float(resid.max())})
                if verbose:
                    print(rows[-1], flush=True)
    mean = float(np.mean([x["map_and_resolve_s"] + x["current_diffusion_s"] + x["profiles_host_s"] for x in rows]))
    return {"cfg": cfg, "rho": rho, "te": te, "ne": ne, "rows": rows, "s_per_coupled_step_ensemble": mean, "member_steps_per_s": B / mean}


def main():
    from oracle import oracle as orc  # CPU baseline / checker only

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    run = coupled_steps(n, B, steps, verbose=True)
    cfg, rho, te, ne, rows, mean = run["cfg"], run["rho"], run["te"], run["ne"], run["rows"], run["s_per_coupled_step_ensemble"]
    d = cfg["dimensions"]
    r, z = np.linspace(d["R_min"], d["R_max"], n), np.linspace(d["Z_min"], d["Z_max"], n)
    t = time.perf_counter()
    ref = orc.picard_solve(cfg, external_profile_mode=True)
    orc.map_profiles_to_2d(ref["psi"], r, z, r[1] - r[0], z[1] - z[0], d["Z_min"], 15.0, rho, ne[0] * 2.0 * te[0], np.zeros_like(rho))
    t_cpu = time.perf_counter() - t
    mean = float(np.mean([x["map_and_resolve_s"] + x["current_diffusion_s"] + x["profiles_host_s"] for x in rows]))
    summary = {"grid": n, "members": B, "steps": steps, "s_per_coupled_step_ensemble": mean, "member_steps_per_s": B / mean,
               "cpu_port_one_member_s_per_step": t_cpu, "cpu_picard_iterations": ref["iterations"], "rows": rows}
    print(json.dumps({k: v for k, v in summary.items() if k != "rows"}))
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "transport_coupled_loop.json").write_text(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()

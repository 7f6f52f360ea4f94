#!/usr/bin/env python3
"""Counterpart of the reference's benchmarks/bench_current_diffusion_nonadiabatic.py (non-adiabatic MIF/FRC flux carrier,
cases (n_rho, n_steps) = (64,128), (256,128), (1024,128), 5 repeats; reference lines :40-92) on the HIP tier.

Rows per case: (i) the drop-in call `solve_flux_evolution_nonadiabatic` (Python drive call-backs sampled on the host as in the
reference, recurrence on the device), (ii) the device recurrence alone for a 4096-profile ensemble (arrays resident in host
memory, one launch), (iii) the CPU oracle (C port of the recurrence, one core) for one profile.  Checksums are compared.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from scpn_fusion_core_amd import transport_coupling as tc  # noqa: E402

REPEATS, CASES, ENSEMBLE = 5, [(64, 128), (256, 128), (1024, 128)], 4096


def main():
    from oracle import oracle as orc  # CPU baseline / checker only

    rows = []
    for n_rho, n_steps in CASES:
        rho = np.linspace(0.0, 1.0, n_rho)
        psi0 = 0.08 - 0.02 * rho**2
        dt = 1.0e-8
        kw = dict(tau_psi_fn=lambda _: 2.0e-6, R_null_t=lambda _: 0.18, E_theta_t=lambda t, g: 25.0 + t / (n_steps * dt) + g,
                  eta_spitzer_fn=lambda g: np.full_like(g, 2.5e-6), J_theta_t=lambda _, g: 1.0e5 * (1.0 - 0.3 * g), dt=dt, n_steps=n_steps)
        tc.solve_flux_evolution_nonadiabatic(rho, psi0, **kw)  # warm-up
        t_call = []
        for _ in range(REPEATS):
            t = time.perf_counter()
            tr = tc.solve_flux_evolution_nonadiabatic(rho, psi0, **kw)
            t_call.append(time.perf_counter() - t)
        rate = np.broadcast_to(tr.damping_rate, (ENSEMBLE, n_steps + 1, n_rho)).copy()
        src = np.broadcast_to(tr.source, (ENSEMBLE, n_steps + 1, n_rho)).copy()
        p0 = np.broadcast_to(psi0, (ENSEMBLE, n_rho)).copy()
        t_ens = []
        for _ in range(3):
            t = time.perf_counter()
            psi_e = tc.flux_evolution_batched(p0, rate, src, dt)[0]
            t_ens.append(time.perf_counter() - t)
        t_cpu = []
        for _ in range(REPEATS):
            t = time.perf_counter()
            ref = orc.flux_evolution(psi0, tr.damping_rate, tr.source, dt)[0]
            t_cpu.append(time.perf_counter() - t)
        rows.append({"grid_points": n_rho, "steps": n_steps, "hip_call_mean_s": float(np.mean(t_call)),
                     "hip_ensemble_s_per_profile": float(np.min(t_ens)) / ENSEMBLE, "ensemble": ENSEMBLE,
                     "cpu_port_mean_s": float(np.mean(t_cpu)), "final_psi_checksum": float(np.sum(tr.psi[-1])),
                     "checksum_cpu": float(np.sum(ref[-1])), "ensemble_max_dev": float(np.abs(psi_e - tr.psi).max()),
                     "max_abs_update_residual": float(np.max(np.abs(tr.update_residual)))})
        print(rows[-1], flush=True)
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "current_diffusion_nonadiabatic.json").write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Counterpart of the reference's benchmarks/bench_gpu_gs_solver.py on the HIP tier.

Same problem (seed 2026, R in [4,8], Z in [-4,4], Gaussian source, omega 1.3, 200 sweeps, n in {129,257,513},
median of 3; reference lines :35-63) through the same provider signature, timed three ways:
  (i)   the scpn_fusion_rs-shaped list-of-f32 call the reference benchmark times (gpu_tier.PyGpuSolver.solve),
  (ii)  the fp64 ndarray dispatcher provider (multi_compat "gs_rb_sor_smooth": host buffers in / out),
  (iii) the resident zero-copy path (EnsembleSolver.rb_sor, fields already in HBM).
The "numpy" column is the CPU oracle (C port of mg_smooth) -- the reference's NumPy does not travel to the GPU box.
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from scpn_fusion_core_amd import gpu_tier, multi_compat  # noqa: E402
from scpn_fusion_core_amd.ensemble import EnsembleSolver  # noqa: E402
from scpn_fusion_core_amd.cases import bench_problem, iter_like_cfg  # noqa: E402

R_LEFT, R_RIGHT, Z_BOTTOM, Z_TOP, OMEGA, SWEEPS, REPEATS = 4.0, 8.0, -4.0, 4.0, 1.3, 200, 3


def median_time(fn):
    ts, out = [], None
    for _ in range(REPEATS):
        t = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)), out


def main():
    from oracle import oracle as orc  # CPU baseline / checker only

    rows = []
    for n in (129, 257, 513):
        psi0, source, rr, r, z = bench_problem(n, n)
        dr, dz = (R_RIGHT - R_LEFT) / (n - 1), (Z_TOP - Z_BOTTOM) / (n - 1)
        t_cpu, ref = median_time(lambda: orc.mg_smooth(psi0, source, rr, dr, dz, OMEGA, SWEEPS))
        solver = gpu_tier.PyGpuSolver(n, n, R_LEFT, R_RIGHT, Z_BOTTOM, Z_TOP)
        t_list, flat = median_time(lambda: solver.solve(psi0.astype(np.float32).ravel().tolist(),
                                                        source.astype(np.float32).ravel().tolist(), SWEEPS, OMEGA))
        prov = multi_compat.dispatch("gs_rb_sor_smooth")
        t_nd, out = median_time(lambda: prov(psi0, source, R_LEFT, R_RIGHT, Z_BOTTOM, Z_TOP, omega=OMEGA, n_sweeps=SWEEPS))
        cfg = iter_like_cfg(n)
        cfg["dimensions"] = {"R_min": R_LEFT, "R_max": R_RIGHT, "Z_min": Z_BOTTOM, "Z_max": Z_TOP}
        with EnsembleSolver(cfg, batch=1) as es:
            es.set_source(source, 0)

            def resident():
                es.set_psi(psi0, 0)
                t = time.perf_counter()
                es.rb_sor(OMEGA, SWEEPS)
                return time.perf_counter() - t

            t_res = float(np.median([resident() for _ in range(REPEATS)]))
            got = es.get_psi(0)
        rows.append({"grid": f"{n}x{n}", "sweeps": SWEEPS, "cpu_oracle_median_s": t_cpu, "gpu_list_f32_median_s": t_list,
                     "gpu_ndarray_f64_median_s": t_nd, "gpu_resident_median_s": t_res,
                     "list_f32_rel_l2": float(np.linalg.norm(flat.astype(np.float64).reshape(n, n) - ref) / np.linalg.norm(ref)),
                     "ndarray_bit_exact": bool(np.array_equal(out, ref)), "resident_bit_exact": bool(np.array_equal(got, ref))})
    print(json.dumps({"kernel": "gs_rb_sor_smooth", "omega": OMEGA, "gpu_adapter": gpu_tier.py_gpu_info(), "grids": rows}, indent=1))


if __name__ == "__main__":
    main()

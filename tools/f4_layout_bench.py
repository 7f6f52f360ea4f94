"""Batched transport-coupling kernels: ms per call of the tridiagonal solve (k_gtsv) and of
CurrentDiffusionSolver.step (k_current_diffusion) for n = 1024 and several batch sizes.  Run it twice --
SCPN_HIP_F4_LAYOUT=rows (rounds 1-5: [batch][n], a wave's accesses strided by n doubles) and unset ([n][batch], coalesced; the
transposes on the device are inside the timing) -- or under `rocprofv3 --kernel-trace --stats` for the kernel durations alone.
usage: f4_layout_bench.py [n] [batch ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd import transport_coupling as tc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batches = [int(a) for a in sys.argv[2:]] or [32, 256, 4096]
rng = np.random.default_rng(7)
print(f"layout: {os.environ.get('SCPN_HIP_F4_LAYOUT', 'transposed [n][batch]')}")
for B in batches:
    d = 4.0 + rng.random((B, n))
    dl, du, b = rng.random((B, n - 1)) - 0.5, rng.random((B, n - 1)) - 0.5, rng.random((B, n))
    tc.solve_banded_tridiagonal(dl, d, du, b)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        x = tc.solve_banded_tridiagonal(dl, d, du, b)
        ts.append(time.perf_counter() - t)
    r = d * x
    r[:, 1:] += dl * x[:, :-1]
    r[:, :-1] += du * x[:, 1:]
    rho = np.linspace(0.0, 1.0, n)
    cd = tc.CurrentDiffusionSolver(rho, 6.2, 2.0, 5.3, batch=B)
    te = 8.0 * (1 - rho**2)[None] * (1.0 + 0.1 * rng.random((B, 1))) + 0.1
    ne = 9.0 * np.ones((B, n))
    jb = 1e5 * rho[None] * (1 - rho)[None] * np.ones((B, 1))
    cd.step(1e-3, te, ne, 1.5, jb, 0.0 * jb)
    tcd = []
    for _ in range(5):
        t = time.perf_counter()
        cd.step(1e-3, te, ne, 1.5, jb, 0.0 * jb)
        tcd.append(time.perf_counter() - t)
    print(f"n={n} batch={B:5d}: tridiag {1e3 * min(ts):8.3f} ms/call (max |Ax-b| {np.abs(r - b).max():.1e}), current-diffusion step {1e3 * min(tcd):8.3f} ms/call", flush=True)

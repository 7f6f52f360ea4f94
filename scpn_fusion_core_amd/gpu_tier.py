"""``scpn_fusion_rs``-shaped GPU-tier surface served by libscpn_hip.so.

What ``benchmarks/bench_gpu_gs_solver.py`` and the dispatcher's GPU tier import from the reference's PyO3 module
(scpn-fusion-rs/crates/fusion-python/src/bindings/gpu.rs:19-90, bindings/equilibrium.rs:369-403):
``py_gpu_available``, ``py_gpu_info``, ``PyGpuSolver(nr, nz, r_left, r_right, z_bottom, z_top)`` with
``.solve(psi, source, iterations, omega)``, ``.vcycle(...)``, ``.grid_shape()``, and ``multigrid_vcycle``.
Differences from the wgpu tier, both in the direction of the NumPy lane: arithmetic is fp64 (the result array is
still float32 for interface compatibility unless ``out_dtype`` is overridden) and restriction is full weighting.
"""

from __future__ import annotations

import numpy as np

from . import _lib
from . import multigrid_solve as _mgs


def py_gpu_available() -> bool:
    return _lib.gpu_available()


def py_gpu_info() -> str | None:
    return _lib.device_info() if _lib.gpu_available() else None


class PyGpuSolver:
    def __init__(self, nr: int, nz: int, r_left: float, r_right: float, z_bottom: float, z_top: float, out_dtype=np.float32):
        if nr < 4 or nz < 4:  # fusion-gpu/src/lib.rs:84-88
            raise RuntimeError("grid must be at least 4x4")
        if not (r_right > r_left and z_top > z_bottom):
            raise RuntimeError("invalid grid extent")
        if not _lib.gpu_available():
            raise RuntimeError("no GPU adapter available")
        self.nr, self.nz = int(nr), int(nz)
        self.r_axis = np.linspace(r_left, r_right, self.nr)
        self.dr = (r_right - r_left) / (self.nr - 1)
        self.dz = (z_top - z_bottom) / (self.nz - 1)
        self.out_dtype = out_dtype

    def grid_shape(self) -> tuple[int, int]:
        return (self.nz, self.nr)

    def _field(self, a, name):
        arr = np.asarray(a, dtype=np.float64).reshape(-1)
        if arr.size != self.nz * self.nr:  # fusion-gpu/src/lib.rs:400-409
            raise RuntimeError(f"{name} length {arr.size} != nz*nr = {self.nz * self.nr}")
        return arr.reshape(self.nz, self.nr)

    def solve(self, psi, source, iterations: int, omega: float):
        """``iterations`` red-black SOR sweeps; flat (nz*nr) array out."""
        if not (np.isfinite(omega) and 1.0 <= omega < 2.0):  # fusion-gpu/src/lib.rs:423-428
            raise RuntimeError("omega must satisfy 1.0 <= omega < 2.0")
        out = _mgs.mg_smooth(self._field(psi, "psi").copy(), self._field(source, "source"), self.r_axis, self.dr, self.dz,
                             float(omega), int(iterations))
        return out.astype(self.out_dtype).ravel()

    def vcycle(self, psi, source, pre: int = 3, post: int = 3, omega: float = 1.0):
        out = _mgs.multigrid_vcycle(self._field(psi, "psi"), self._field(source, "source"), self.r_axis, self.dr, self.dz,
                                    omega=float(omega), pre_smooth=int(pre), post_smooth=int(post))
        return out.astype(self.out_dtype).ravel()


def multigrid_vcycle(source, psi_bc, r_min, r_max, z_min, z_max, nr, nz, tol=1e-6, max_cycles=500):
    """``scpn_fusion_rs.multigrid_vcycle`` (full multigrid solve): ``(psi f64 (nz, nr), residual, n_cycles, converged)``."""
    return _mgs.multigrid_solve(source, psi_bc, r_min, r_max, z_min, z_max, nr, nz, tol=tol, max_cycles=max_cycles)

"""Backend dispatch seam of the reference (core/_multi_compat.py:52-313) with the HIP GPU tier registered.

Same surface: ``BackendTier`` (lower = preferred), ``register_kernel(name, tier, fn)``, ``dispatch(name)``,
``is_available(tier)``, ``available_backends()``, ``register_kernel_class`` / ``dispatch_kernel_class``.  The
reference's Python does not travel to the GPU box, so the registry itself is restated here; a maintainer
wiring libscpn_hip into the real package only needs the three ``register_*`` calls at the bottom
(INTEGRATION.md).  Only the GPU tier is registered: there is deliberately no NumPy/CPU tier in this package.
"""

from __future__ import annotations

import os
import threading
from enum import IntEnum
from typing import Any, Callable

import numpy as np

from . import _lib


class BackendTier(IntEnum):
    RUST = 0
    GPU = 1
    MOJO = 2
    JULIA = 3
    GO = 4
    JAX = 5
    NUMPY = 6


_TIER_NAMES = {t: t.name.lower() for t in BackendTier}
_registry: dict[str, list[tuple[BackendTier, Callable[..., Any]]]] = {}
_class_registry: dict[str, list[tuple[BackendTier, Callable[[], type]]]] = {}
_dispatch_cache: dict[str, tuple[BackendTier, Callable[..., Any]]] = {}
_availability: dict[BackendTier, bool] = {}
_lock = threading.RLock()
_probed = False


def _probe_gpu() -> bool:
    """core/_multi_compat.py:101-120: honours SCPN_DISABLE_GPU, otherwise asks the native library."""
    return _lib.gpu_available()


def _ensure_probed() -> None:
    global _probed
    if _probed:
        return
    with _lock:
        if _probed:
            return
        for t in BackendTier:
            _availability[t] = False
        _availability[BackendTier.GPU] = _probe_gpu()
        _probed = True


def reset_probe_cache() -> None:
    global _probed
    with _lock:
        _probed = False
        _availability.clear()
        _dispatch_cache.clear()


def is_available(tier: BackendTier) -> bool:
    _ensure_probed()
    return bool(_availability.get(tier, False))


def available_backends() -> dict[str, bool]:
    _ensure_probed()
    return {_TIER_NAMES[t]: bool(_availability.get(t, False)) for t in BackendTier}


def register_kernel(name: str, tier: BackendTier, implementation: Callable[..., Any]) -> None:
    with _lock:
        _registry.setdefault(name, []).append((tier, implementation))
        _registry[name].sort(key=lambda x: x[0])
        _dispatch_cache.pop(name, None)


def dispatch(name: str) -> Callable[..., Any]:
    """Fastest available implementation; KeyError if nothing registered, RuntimeError if all unavailable."""
    _ensure_probed()
    cached = _dispatch_cache.get(name)
    if cached is not None:
        return cached[1]
    with _lock:
        entries = _registry.get(name)
        if not entries:
            raise KeyError(f"No implementations registered for kernel {name!r}")
        for tier, impl in entries:
            if _availability.get(tier, False):
                _dispatch_cache[name] = (tier, impl)
                return impl
        raise RuntimeError(f"All registered backends for {name!r} are unavailable. "
                           f"Registered tiers: {[_TIER_NAMES[t] for t, _ in entries]}")


def register_kernel_class(name: str, tier: BackendTier, loader: Callable[[], type]) -> None:
    with _lock:
        _class_registry.setdefault(name, []).append((tier, loader))
        _class_registry[name].sort(key=lambda x: x[0])


def dispatch_kernel_class(name: str) -> type:
    _ensure_probed()
    entries = _class_registry.get(name)
    if not entries:
        raise KeyError(f"No kernel classes registered for {name!r}")
    for tier, loader in entries:
        if _availability.get(tier, False):
            return loader()
    raise RuntimeError(f"All registered backends for kernel class {name!r} are unavailable.")


# ── GPU-tier providers (canonical signatures: core/_multi_compat_providers.py:322-344, 723-792) ───────────
def _gpu_gs_rb_sor_smooth(psi: Any, source: Any, r_left: float, r_right: float, z_bottom: float, z_top: float, *,
                          omega: float = 1.3, n_sweeps: int = 50) -> Any:
    """``gs_rb_sor_smooth``: n_sweeps red-black SOR sweeps on a copy of psi, float64 (nz, nr) out; input not mutated.
    fp64 on the device (the reference's wgpu tier is f32), so it matches the NumPy tier bit for bit."""
    from .multigrid_solve import mg_smooth

    psi_arr = np.array(psi, dtype=np.float64, copy=True)
    source_arr = np.asarray(source, dtype=np.float64)
    nz, nr = psi_arr.shape
    r_axis = np.linspace(r_left, r_right, nr)
    dr = (r_right - r_left) / (nr - 1)
    dz = (z_top - z_bottom) / (nz - 1)
    return mg_smooth(psi_arr, source_arr, r_axis, dr, dz, omega, n_sweeps)


def _gpu_multigrid_solve(source: Any, psi_bc: Any, r_min: float, r_max: float, z_min: float, z_max: float, nr: int, nz: int,
                         *, tol: float = 1e-6, max_cycles: int = 500) -> Any:
    """``multigrid_solve``: ``(psi, residual, n_cycles, converged)``."""
    from .multigrid_solve import multigrid_solve

    return multigrid_solve(source, psi_bc, r_min, r_max, z_min, z_max, nr, nz, tol=tol, max_cycles=max_cycles)


def _load_equilibrium_kernel() -> type:
    from .fusion_kernel import FusionKernel

    return FusionKernel


register_kernel("gs_rb_sor_smooth", BackendTier.GPU, _gpu_gs_rb_sor_smooth)
register_kernel("multigrid_solve", BackendTier.GPU, _gpu_multigrid_solve)
register_kernel_class("equilibrium_kernel", BackendTier.GPU, _load_equilibrium_kernel)

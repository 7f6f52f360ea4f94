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


class _Seam:
    """The registry behind the module-level functions.

    Layout: one table per kind (``"fn"`` for kernels, ``"cls"`` for lazily loaded kernel classes) mapping a kernel name to a
    ``{tier: implementation}`` dict -- the FIRST registration of a (name, tier) pair stays, as in the reference, whose registry
    appends and stable-sorts by tier (core/_multi_compat.py:240-262) -- and a table
    of tier probes (``{tier: () -> bool}``, only the GPU tier has one in this package) whose answers are memoised until
    :meth:`forget`.  Selection is ``min()`` over the available tiers of a name (the enum value IS the preference)."""

    def __init__(self, probes: dict[BackendTier, Callable[[], bool]]) -> None:
        self._probes = dict(probes)
        self._tables: dict[str, dict[str, dict[BackendTier, Any]]] = {"fn": {}, "cls": {}}
        self._up: dict[BackendTier, bool] = {}      # memoised probe answers (and test overrides)
        self._chosen: dict[str, Callable[..., Any]] = {}  # kernel name -> implementation picked by the last dispatch
        self._guard = threading.RLock()

    # ---- availability
    def up(self, tier: BackendTier) -> bool:
        with self._guard:
            if tier not in self._up:
                probe = self._probes.get(tier)
                self._up[tier] = bool(probe()) if probe else False
            return self._up[tier]

    def force(self, tier: BackendTier, state: bool | None) -> None:
        """Pin a tier's availability (``None``: back to its probe).  For tests and for ``SCPN_DISABLE_GPU``-style switches."""
        with self._guard:
            if state is None:
                self._up.pop(tier, None)
            else:
                self._up[tier] = bool(state)
            self._chosen.clear()

    def forget(self) -> None:
        with self._guard:
            self._up.clear()
            self._chosen.clear()

    # ---- registration / selection
    def add(self, kind: str, name: str, tier: BackendTier, impl: Any) -> None:
        with self._guard:
            self._tables[kind].setdefault(name, {}).setdefault(BackendTier(tier), impl)
            if kind == "fn":  # (kernel classes are not memoised: a class registration leaves the kernels' choices alone)
                self._chosen.pop(name, None)

    def chosen(self, name: str, missing: str, all_down: str) -> Callable[..., Any]:
        """The memoised pick of kernel ``name`` (read and written under the registry lock)."""
        with self._guard:
            hit = self._chosen.get(name)
            if hit is None:
                hit = self._chosen[name] = self.pick("fn", name, missing, all_down)
            return hit

    def tiers(self, kind: str, name: str) -> list[BackendTier]:
        return sorted(self._tables[kind].get(name, ()))

    def pick(self, kind: str, name: str, missing: str, all_down: str) -> Any:
        with self._guard:
            offered = self._tables[kind].get(name)
            if not offered:
                raise KeyError(missing)
            live = [t for t in offered if self.up(t)]
            if not live:
                raise RuntimeError(all_down)
            return offered[min(live)]


_seam = _Seam({BackendTier.GPU: _lib.gpu_available})  # core/_multi_compat.py:101-120: SCPN_DISABLE_GPU, then the native library


def reset_probe_cache() -> None:
    _seam.forget()


def set_availability(tier: BackendTier, state: bool | None) -> None:
    """Override (``True`` / ``False``) or restore (``None``) what :func:`is_available` reports for ``tier``."""
    _seam.force(tier, state)


def is_available(tier: BackendTier) -> bool:
    return _seam.up(BackendTier(tier))


def available_backends() -> dict[str, bool]:
    return {t.name.lower(): _seam.up(t) for t in BackendTier}


def registered_tiers(name: str, *, kernel_class: bool = False) -> list[BackendTier]:
    """Tiers that offer ``name`` (ascending = preference order), whether or not they are available here."""
    return _seam.tiers("cls" if kernel_class else "fn", name)


def register_kernel(name: str, tier: BackendTier, implementation: Callable[..., Any]) -> None:
    _seam.add("fn", name, tier, implementation)


def dispatch(name: str) -> Callable[..., Any]:
    """Fastest available implementation; KeyError if nothing registered, RuntimeError if all unavailable."""
    return _seam.chosen(name, f"No implementations registered for kernel {name!r}",
                        f"All registered backends for {name!r} are unavailable. "
                        f"Registered tiers: {[t.name.lower() for t in registered_tiers(name)]}")


def register_kernel_class(name: str, tier: BackendTier, loader: Callable[[], type]) -> None:
    _seam.add("cls", name, tier, loader)


def dispatch_kernel_class(name: str) -> type:
    loader = _seam.pick("cls", name, f"No kernel classes registered for {name!r}",
                        f"All registered backends for kernel class {name!r} are unavailable.")
    return loader()


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

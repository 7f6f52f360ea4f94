"""Loader for the legacy ``libscpn_solver`` C ABI served by libscpn_hip.so.

Mirror of the reference's ``HPCBridge`` (hpc/hpc_bridge.py:89-455) and of its trust gate
(hpc/_hpc_native_trust.py:186-221): the library path comes from ``lib_path`` / ``SCPN_SOLVER_LIB``
(absolute, not a symlink, regular file) or the package-local libscpn_hip.so, and its SHA-256 must
match ``SCPN_SOLVER_LIB_SHA256``, a ``SCPN_SOLVER_TRUST_MANIFEST`` entry or a ``<lib>.sha256`` sidecar
when an explicit path is given.  Semantics of the six symbols are those of hpc/solver.cpp:200-335.
"""

from __future__ import annotations

import ctypes as C
import hashlib
import hmac
import json
import math
import os
from pathlib import Path

import numpy as np

from . import _lib

_SOLVER_LIB_ENV = "SCPN_SOLVER_LIB"
_HEX = frozenset("0123456789abcdef")


class LibraryTrust:
    """The trust rule of hpc/_hpc_native_trust.py:186-221 for an explicitly named solver library, as one object.

    ``LibraryTrust(path)`` validates the path (absolute, no symlink, existing regular file) and ``.verify()`` compares the
    file's SHA-256 with the FIRST pinned digest found in this order: ``SCPN_SOLVER_LIB_SHA256``, the JSON manifest named by
    ``SCPN_SOLVER_TRUST_MANIFEST`` (flat ``{path-or-name: digest}`` or under ``"libraries"``), the ``<lib>.sha256`` sidecar.
    Environment-variable names and error texts are the reference's (its tests match on them); the structure is not."""

    # (test on the candidate path / its resolution, error text) -- evaluated in order, first failure raises
    _PATH_RULES = (
        (lambda raw, res: raw.is_absolute(), lambda raw, res: f"{_SOLVER_LIB_ENV}/lib_path must be an absolute path"),
        (lambda raw, res: not raw.is_symlink(), lambda raw, res: f"explicit native solver library must not be a symlink: {raw}"),
        (lambda raw, res: res.exists(), lambda raw, res: f"explicit native solver library does not exist: {res}"),
        (lambda raw, res: res.is_file(), lambda raw, res: f"explicit native solver library must be a regular file: {res}"),
    )

    def __init__(self, path) -> None:
        raw = Path(path).expanduser()
        resolved = raw.resolve(strict=False) if raw.is_absolute() else raw
        for ok, message in self._PATH_RULES:
            if not ok(raw, resolved):
                raise ValueError(message(raw, resolved))
        self.path = resolved

    # ---- digests
    @staticmethod
    def parse_digest(text: str) -> str:
        """First whitespace-separated token of ``text`` as a lower-case SHA-256 (the ``sha256sum`` line format)."""
        token = (text.split() or [""])[0].lower()
        if len(token) != 64 or not set(token) <= _HEX:
            raise ValueError("SHA-256 digest must be 64 hexadecimal characters")
        return token

    def actual_digest(self) -> str:
        h = hashlib.sha256()
        with open(self.path, "rb") as f:
            while chunk := f.read(1 << 20):
                h.update(chunk)
        return h.hexdigest()

    def _from_env(self):
        return os.environ.get("SCPN_SOLVER_LIB_SHA256") or None

    def _from_manifest(self):
        where = os.environ.get("SCPN_SOLVER_TRUST_MANIFEST")
        if not where:
            return None
        table = json.loads(Path(where).read_text(encoding="utf-8"))
        table = table.get("libraries", table) if isinstance(table, dict) else table
        if not isinstance(table, dict):
            raise ValueError("native library trust manifest must be a JSON object")
        hit = next((table[k] for k in (str(self.path), str(self.path.resolve()), self.path.name) if k in table), None)
        if hit is not None and not isinstance(hit, str):
            raise ValueError("native library trust manifest digest must be a string")
        return hit

    def _from_sidecar(self):
        side = self.sidecar_path(self.path)
        return side.read_text(encoding="utf-8") if side.exists() else None

    def pinned_digest(self) -> str | None:
        for source in (self._from_env, self._from_manifest, self._from_sidecar):
            text = source()
            if text:
                return self.parse_digest(text)
        return None

    def verify(self) -> str:
        pinned = self.pinned_digest()
        if pinned is None:
            raise ValueError("native solver library trust metadata is required; provide SCPN_SOLVER_LIB_SHA256, "
                             "SCPN_SOLVER_TRUST_MANIFEST, or a .sha256 sidecar")
        actual = self.actual_digest()
        if not hmac.compare_digest(actual, pinned):
            raise ValueError("native solver library SHA-256 does not match trusted metadata")
        return actual

    @staticmethod
    def sidecar_path(path: Path) -> Path:
        return path.with_name(path.name + ".sha256")


def require_explicit_library_path(path) -> Path:
    """The validated, resolved path of an explicit library override (raises ``ValueError`` with the reference's texts)."""
    return LibraryTrust(path).path


def verify_native_library_trust(path) -> str:
    """SHA-256 of ``path`` after it matched its pinned digest (``ValueError`` otherwise)."""
    return LibraryTrust(path).verify()


def write_sha256_sidecar(path) -> None:
    """Pin ``path`` with a ``sha256sum``-style sidecar next to it."""
    t = LibraryTrust(path)
    LibraryTrust.sidecar_path(Path(path)).write_text(f"{t.actual_digest()}  {Path(path).name}\n", encoding="utf-8")


_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class HPCBridge:
    """ctypes face of create_solver / run_step / run_step_converged / set_boundary_dirichlet / destroy_solver."""

    def __init__(self, lib_path: str | None = None) -> None:
        self.lib = None
        self.solver_ptr = None
        self.loaded = False
        self.nr = self.nz = 0
        explicit = lib_path or os.environ.get(_SOLVER_LIB_ENV)
        try:
            if explicit:
                trust = LibraryTrust(explicit)
                trust.verify()
                path = trust.path
                L = C.CDLL(str(path))
                L.create_solver.restype = C.c_void_p
                L.create_solver.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
                L.run_step.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_int]
                L.destroy_solver.argtypes = [C.c_void_p]
                if hasattr(L, "run_step_converged"):
                    L.run_step_converged.restype = C.c_int
                    L.run_step_converged.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double,
                                                     C.POINTER(C.c_double)]
                if hasattr(L, "set_boundary_dirichlet"):
                    L.set_boundary_dirichlet.argtypes = [C.c_void_p, C.c_double]
                self.lib = L
            else:
                self.lib = _lib.lib()
            self.loaded = True
        except (OSError, ValueError, _lib.HipUnavailableError):
            self.lib = None
            self.loaded = False
            if explicit:
                raise

    def is_available(self) -> bool:
        return bool(self.loaded and self.lib is not None and _lib.device_count() > 0)

    def initialize(self, nr: int, nz: int, r_range, z_range, boundary_value: float = 0.0) -> None:
        if not self.loaded:
            return
        self.close()
        self.nr, self.nz = int(nr), int(nz)
        self.solver_ptr = self.lib.create_solver(self.nr, self.nz, float(r_range[0]), float(r_range[1]), float(z_range[0]),
                                                 float(z_range[1]))
        if self.solver_ptr and boundary_value != 0.0:
            self.set_boundary_dirichlet(boundary_value)

    def set_boundary_dirichlet(self, boundary_value: float) -> None:
        if self.solver_ptr and hasattr(self.lib, "set_boundary_dirichlet"):
            self.lib.set_boundary_dirichlet(self.solver_ptr, float(boundary_value))

    def _prepare_inputs(self, j_phi):
        j = np.asarray(j_phi, dtype=np.float64)
        if j.ndim != 2 or j.size == 0:
            raise ValueError("j_phi must be a non-empty 2-D array")
        if not np.all(np.isfinite(j)):
            raise ValueError("j_phi must contain finite values only")
        if j.shape != (self.nz, self.nr):
            raise ValueError(f"j_phi shape mismatch: expected {(self.nz, self.nr)}, received {j.shape}")
        return np.ascontiguousarray(j)

    def solve(self, j_phi, iterations: int = 100):
        if not self.solver_ptr:
            return None
        j = self._prepare_inputs(j_phi)
        psi = np.zeros((self.nz, self.nr))
        self.lib.run_step(self.solver_ptr, j, psi, psi.size, int(iterations))
        return psi

    def solve_until_converged(self, j_phi, max_iterations: int = 1000, tolerance: float = 1e-6, omega: float = 1.8):
        if not self.solver_ptr:
            return None
        max_iters = int(max_iterations)
        if max_iters < 1:
            raise ValueError("max_iterations must be >= 1.")
        tol = float(tolerance)
        if not math.isfinite(tol) or tol < 0.0:
            raise ValueError("tolerance must be finite and >= 0.")
        om = float(omega)
        if not math.isfinite(om) or om <= 0.0 or om >= 2.0:
            raise ValueError("omega must be finite and in (0, 2).")
        j = self._prepare_inputs(j_phi)
        psi = np.zeros((self.nz, self.nr))
        fd = C.c_double()
        n = self.lib.run_step_converged(self.solver_ptr, j, psi, psi.size, max_iters, om, tol, C.byref(fd))
        return psi, int(n), float(fd.value)

    def close(self) -> None:
        if self.solver_ptr and self.lib is not None:
            self.lib.destroy_solver(self.solver_ptr)
        self.solver_ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

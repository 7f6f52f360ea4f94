"""``FusionKernel``-compatible equilibrium kernel on the HIP path.

Drop-in for the reference's ``scpn_fusion.core.fusion_kernel.FusionKernel`` on the equilibrium path
(core/fusion_kernel.py:104-467 and its mixins): same constructor (a JSON config path -- a dict is also
accepted), same attributes (``Psi J_phi B_R B_Z R Z RR ZZ dR dZ NR NZ cfg``) and the methods the
``equilibrium_kernel`` class seam requires (core/_rust_compat.py:83-212): ``solve_equilibrium``,
``solve_free_boundary``, ``find_x_point``, ``calculate_vacuum_field``, ``compute_b_field``,
``save_results``.  The Picard outer loop's control flow lives on the host side of the C ABI; all field
work runs in the gfx950 kernels (one resident ensemble member).
"""

from __future__ import annotations

import json
import time
from pathlib import Path
from typing import Any

import numpy as np

from . import free_boundary as _fb
from . import multigrid_solve as _mgs
from .ensemble import EnsembleSolver

MAX_CONFIG_BYTES = 8 * 1024 * 1024


def validate_config(raw: dict) -> dict:
    """Minimal restatement of the reference's pydantic contract (core/config_schema.py:31-102):
    required dimensions with R_min > 0 and R_max > R_min, grid >= 4x4, finite numbers, defaults filled in."""
    if not isinstance(raw, dict):
        raise ValueError("configuration must be a JSON object")
    cfg = json.loads(json.dumps(raw))  # deep copy, rejects non-JSON values
    if "dimensions" not in cfg:
        raise ValueError("dimensions: field required")
    d = cfg["dimensions"]
    for k in ("R_min", "R_max", "Z_min", "Z_max"):
        if k not in d:
            raise ValueError(f"dimensions.{k}: field required")
        if not np.isfinite(float(d[k])):
            raise ValueError(f"dimensions.{k} must be finite")
    if not (d["R_min"] > 0 and d["R_max"] > 0):
        raise ValueError("R_min and R_max must be > 0")
    if d["R_max"] <= d["R_min"]:
        raise ValueError("R_max must be greater than R_min")
    cfg.setdefault("reactor_name", "Unnamed-Reactor")
    res = tuple(int(v) for v in cfg.get("grid_resolution", (129, 129)))
    if len(res) != 2 or res[0] < 4 or res[1] < 4:
        raise ValueError("Grid resolution must be at least 4x4")
    cfg["grid_resolution"] = list(res)
    coils = cfg.setdefault("coils", [])
    for c in coils:
        c.setdefault("name", "unnamed")
        c.setdefault("current", 0.0)
        if not (float(c["r"]) > 0) or not all(np.isfinite(float(c[k])) for k in ("r", "z", "current")):
            raise ValueError("coil r must be > 0 and coil values finite")
    phys = cfg.setdefault("physics", {})
    phys.setdefault("plasma_current_target", 5.0)
    phys.setdefault("vacuum_permeability", 1.25663706e-6)
    if phys["vacuum_permeability"] < 0 or not np.isfinite(phys["plasma_current_target"]):
        raise ValueError("physics parameters out of range")
    sol = cfg.setdefault("solver", {})
    sol.setdefault("max_iterations", 1000)
    sol.setdefault("convergence_threshold", 1e-4)
    sol.setdefault("relaxation_factor", 0.1)
    if not (int(sol["max_iterations"]) > 0 and sol["convergence_threshold"] > 0 and 0 < sol["relaxation_factor"] <= 1.0):
        raise ValueError("solver parameters out of range")
    return cfg


class FusionKernel:
    """Non-linear Grad-Shafranov equilibrium solver backed by libscpn_hip.so."""

    def __init__(self, config_path: str | Path | dict) -> None:
        self._config_path = str(config_path) if not isinstance(config_path, dict) else "<dict>"
        self.load_config(config_path)
        self.initialize_grid()
        self.external_profile_mode = False
        self._es: EnsembleSolver | None = None

    # -- construction (core/fusion_kernel.py:135-214) --------------------------------------------
    def load_config(self, path: str | Path | dict) -> None:
        if isinstance(path, dict):
            raw = path
        else:
            p = Path(path)
            if p.stat().st_size > MAX_CONFIG_BYTES:
                raise ValueError(f"configuration file exceeds {MAX_CONFIG_BYTES} byte limit: {p}")
            raw = json.loads(p.read_text(encoding="utf-8"))
        self.cfg = validate_config(raw)

    def initialize_grid(self) -> None:
        dims, res = self.cfg["dimensions"], self.cfg["grid_resolution"]
        self.NR, self.NZ = int(res[0]), int(res[1])
        self.R = np.linspace(dims["R_min"], dims["R_max"], self.NR)
        self.Z = np.linspace(dims["Z_min"], dims["Z_max"], self.NZ)
        self.dR = float(self.R[1] - self.R[0])
        self.dZ = float(self.Z[1] - self.Z[0])
        self.RR, self.ZZ = np.meshgrid(self.R, self.Z)
        self.Psi = np.zeros((self.NZ, self.NR))
        self.J_phi = np.zeros((self.NZ, self.NR))
        self.B_R = np.zeros((self.NZ, self.NR))
        self.B_Z = np.zeros((self.NZ, self.NR))
        prof = self.cfg.get("physics", {}).get("profiles") or {}
        self.profile_mode = prof.get("mode", "l-mode")

    def _solver(self) -> EnsembleSolver:
        if self._es is None:
            self._es = EnsembleSolver(self.cfg, batch=1)
        self._es.cfg = self.cfg
        self._es.external_profile_mode = bool(self.external_profile_mode)
        it = np.array([float(self.cfg["physics"]["plasma_current_target"])])
        from ._lib import check

        check(self._es._L.scpn_gs_set_current_target(self._es._h, -1, it))
        coils = self.cfg.get("coils", [])
        if len(coils) == self._es.n_coils:
            self._es.coil_r = np.array([[c["r"] for c in coils]], dtype=np.float64)
            self._es.coil_z = np.array([[c["z"] for c in coils]], dtype=np.float64)
            self._es.coil_i = np.array([[c["current"] for c in coils]], dtype=np.float64)
            self._es._push_vacuum_coils()
        return self._es

    def close(self) -> None:
        if self._es is not None:
            self._es.close()
            self._es = None

    # -- vacuum field / topology (core/fusion_kernel.py:218-355) ------------------------------------
    def calculate_vacuum_field(self) -> np.ndarray:
        coils = self.cfg["coils"]
        mu0 = self.cfg["physics"].get("vacuum_permeability", 1.0)
        if not coils:
            return np.zeros((self.NZ, self.NR))
        return _fb.coil_flux_grid(self.R, self.Z, [c["r"] for c in coils], [c["z"] for c in coils],
                                  [c["current"] for c in coils], mu0=mu0, mode="vacuum")

    def find_x_point(self, Psi) -> tuple[tuple[float, float], float]:
        """core/fusion_kernel.py:255-340 (incl. ``solver.xpoint_use_saddle_detection``), evaluated on the host for a single
        field (inside the solve it runs on the device every iteration)."""
        psi_raw = np.asarray(Psi, dtype=np.float64)
        finite = np.isfinite(psi_raw)
        if not np.any(finite):
            return (0.0, 0.0), 0.0
        psi_safe = np.nan_to_num(psi_raw, nan=0.0, posinf=1e300, neginf=-1e300)
        dz, dr = np.gradient(psi_safe, self.dZ, self.dR)
        bmag = np.hypot(dr, dz)
        mask = (self.cfg["dimensions"]["Z_min"] * 0.5) > self.ZZ
        if np.any(mask):
            masked = np.where(mask, bmag, np.inf)
            if int(np.isfinite(masked).sum()) > 0:
                iz, ir = np.unravel_index(int(np.argmin(masked)), psi_raw.shape)
                if self.cfg.get("solver", {}).get("xpoint_use_saddle_detection", False):
                    # the 16 smallest |grad psi| points in ascending (value, index) order; first interior saddle wins
                    nzz, nrr = psi_raw.shape
                    flat = masked.ravel()
                    order = np.lexsort((np.arange(flat.size), flat))[: min(16, int(np.isfinite(flat).sum()))]
                    for q in order:
                        a, c = divmod(int(q), nrr)
                        if a <= 0 or a >= nzz - 1 or c <= 0 or c >= nrr - 1:
                            continue
                        d2r = (psi_safe[a, c + 1] - 2.0 * psi_safe[a, c] + psi_safe[a, c - 1]) / (self.dR**2)
                        d2z = (psi_safe[a + 1, c] - 2.0 * psi_safe[a, c] + psi_safe[a - 1, c]) / (self.dZ**2)
                        drz = (psi_safe[a + 1, c + 1] - psi_safe[a + 1, c - 1] - psi_safe[a - 1, c + 1] + psi_safe[a - 1, c - 1]) / (
                            4.0 * self.dR * self.dZ)
                        det = float(d2r * d2z - drz * drz)
                        if np.isfinite(det) and det < 0.0:
                            iz, ir = a, c
                            break
                px = float(psi_raw[iz, ir])
                if not np.isfinite(px):
                    px = float(psi_safe[iz, ir])
                return (float(self.R[ir]), float(self.Z[iz])), px
        return (0.0, 0.0), float(np.min(psi_raw[finite]))

    def _find_magnetic_axis(self) -> tuple[int, int, float]:
        iz, ir = np.unravel_index(int(np.argmax(self.Psi)), self.Psi.shape)
        pa = float(self.Psi[iz, ir])
        if abs(pa) < 1e-6:
            pa = 1e-6
        return int(iz), int(ir), pa

    # -- solves ----------------------------------------------------------------------------------------
    def solve_equilibrium(self, preserve_initial_state: bool = False, boundary_flux=None) -> dict[str, Any]:
        """core/fusion_kernel_newton_solver.py:390-615 (Picard methods multigrid / sor / jacobi)."""
        t0 = time.time()
        method = self.cfg["solver"].get("solver_method", "multigrid")
        if method == "rust_multigrid":
            raise NotImplementedError("solver_method='rust_multigrid' delegates to the Rust crate; not on the HIP path")
        if method == "newton":
            return self._newton_solve_dispatch(preserve_initial_state, boundary_flux)
        if boundary_flux is not None:
            bf = np.asarray(boundary_flux, dtype=np.float64)
            if bf.shape != self.Psi.shape:
                raise ValueError(f"boundary_flux shape {bf.shape} must match Psi shape {self.Psi.shape}")
        es = self._solver()
        if preserve_initial_state:
            es.set_psi(np.ascontiguousarray(self.Psi, dtype=np.float64), 0)
        if self.external_profile_mode:
            es.set_jphi(np.ascontiguousarray(self.J_phi, dtype=np.float64), 0)
        res = es.solve_equilibrium(preserve_initial_state=preserve_initial_state, boundary_flux=boundary_flux, history=True,
                                   check_every=8)[0]
        if res["diverged"] and self.cfg["solver"].get("fail_on_diverge", False):
            raise RuntimeError(f"Equilibrium solver diverged at iter={res['iterations'] - 1}")
        self.Psi = es.get_psi(0)
        self.J_phi = es.get_jphi(0)
        self.compute_b_field()
        gs_hist = res.get("gs_residual_history", [])
        return {
            "psi": self.Psi,
            "converged": res["converged"],
            "iterations": res["iterations"],
            "residual": res["residual"],
            "residual_history": res.get("residual_history", []),
            "gs_residual": gs_hist[-1] if gs_hist else res["gs_residual"],
            "gs_residual_best": res["gs_residual_best"],
            "gs_residual_history": gs_hist,
            "wall_time_s": time.time() - t0,
            "solver_method": method,
        }

    def _newton_solve_dispatch(self, preserve_initial_state: bool = False, boundary_flux=None) -> dict[str, Any]:
        """core/fusion_kernel_newton_solver.py:104-372 on the device (Picard warm-up + Newton/GMRES)."""
        t0 = time.time()
        if boundary_flux is not None:
            bf = np.asarray(boundary_flux, dtype=np.float64)
            if bf.shape != self.Psi.shape:
                raise ValueError(f"boundary_flux shape {bf.shape} must match Psi shape {self.Psi.shape}")
        es = self._solver()
        if preserve_initial_state:
            es.set_psi(np.ascontiguousarray(self.Psi, dtype=np.float64), 0)
        if self.external_profile_mode:
            es.set_jphi(np.ascontiguousarray(self.J_phi, dtype=np.float64), 0)
        res = es.solve_newton(preserve_initial_state=preserve_initial_state, boundary_flux=boundary_flux)[0]
        if res["diverged"] and self.cfg["solver"].get("fail_on_diverge", False):
            raise RuntimeError(f"Newton solver diverged at iter={res['iterations'] - 1}")
        self.Psi = es.get_psi(0)
        self.J_phi = es.get_jphi(0)
        self.compute_b_field()
        res.pop("diverged", None)
        res.pop("gmres_inner_iterations", None)
        return {"psi": self.Psi, **res, "wall_time_s": time.time() - t0}

    def _solve_free_boundary_device(self, coils: _fb.CoilSet, max_outer_iter: int, tol: float) -> dict[str, Any]:
        es = self._solver()
        es.set_psi(np.ascontiguousarray(self.Psi, dtype=np.float64), 0)
        pos = np.array(coils.positions, dtype=np.float64).reshape(-1, 2)
        cur = np.asarray(coils.currents, dtype=np.float64)
        r = es.solve_free_boundary(pos[:, 0], pos[:, 1], cur, coils.turns_array(), max_outer_iter=max_outer_iter, tol=tol)[0]
        self.Psi = es.get_psi(0)
        self.J_phi = es.get_jphi(0)
        self.compute_b_field()
        return {"outer_iterations": r["outer_iterations"], "final_diff": r["final_diff"], "coil_currents": cur.copy(),
                "picard_iterations": r["picard_iterations"], "converged": r["converged"], "shape_optimization": None}

    def solve_free_boundary(self, coils: _fb.CoilSet, max_outer_iter: int = 20, tol: float = 1e-4,
                            optimize_shape: bool = False, tikhonov_alpha: float = 1e-4, **kw) -> dict[str, Any]:
        return _fb.solve_free_boundary(self, coils, max_outer_iter=max_outer_iter, tol=tol, optimize_shape=optimize_shape,
                                       tikhonov_alpha=tikhonov_alpha, **kw)

    def optimize_coil_currents(self, coils: _fb.CoilSet, target_flux, tikhonov_alpha: float = 1e-4):
        return _fb.optimize_coil_currents(self, coils, target_flux, tikhonov_alpha=tikhonov_alpha)

    # -- post-processing (core/fusion_kernel.py:450-467) ---------------------------------------------
    def compute_b_field(self) -> None:
        if self._es is not None:
            self.B_R, self.B_Z = self._es.compute_b_field(0)
        else:
            dz, dr = np.gradient(self.Psi, self.dZ, self.dR)
            rs = np.maximum(self.RR, 1e-6)
            self.B_R, self.B_Z = -(1.0 / rs) * dz, (1.0 / rs) * dr

    def save_results(self, filename: str = "equilibrium_nonlinear.npz") -> None:
        np.savez(filename, R=self.R, Z=self.Z, Psi=self.Psi, J_phi=self.J_phi)

    # -- elliptic sub-solvers kept for API parity (core/fusion_kernel_iterative_solver.py) ------------
    def _jacobi_step(self, Psi, Source):
        return _mgs.jacobi_step(Psi, Source, self.R, self.dR, self.dZ)

    def _sor_step(self, Psi, Source, omega: float = 1.6):
        return _mgs.sor_step(Psi, Source, self.R, self.dR, self.dZ, omega)

    def _mg_smooth(self, Psi, Source, R_grid, dR, dZ, omega, n_sweeps):
        return _mgs.mg_smooth(Psi, Source, R_grid, dR, dZ, omega, n_sweeps)

    def _multigrid_vcycle(self, Psi, Source, R_grid, dR, dZ, *, omega=1.6, pre_smooth=3, post_smooth=3, min_grid=5):
        return _mgs.multigrid_vcycle(Psi, Source, R_grid, dR, dZ, omega=omega, pre_smooth=pre_smooth, post_smooth=post_smooth,
                                     min_grid=min_grid)

    def _compute_gs_residual_rms(self, Source) -> float:
        return _mgs.gs_residual_rms(self.Psi, Source, self.R, self.dR, self.dZ)


HipFusionKernel = FusionKernel

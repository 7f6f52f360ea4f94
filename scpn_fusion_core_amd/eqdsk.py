"""G-EQDSK (EFIT) equilibrium files in / out, and conversion to a solver config.

Host-side Python, mirroring the public surface of the reference's ``core/eqdsk.py``: ``GEqdsk`` container
(``:61-126``), ``GEqdsk.to_config`` (``:128-196``), ``read_geqdsk`` (``:349-541``), ``write_geqdsk``.  The format is
the standard one of Lao et al., Nucl. Fusion 25 (1985) 1611: a header line (48-character description, idum, nw, nh),
20 scalars, the 1-D arrays fpol / pres / ffprime / pprime (nw each), psirz (nh x nw, row-major), qpsi (nw), the
boundary / limiter point counts and their interleaved (R, Z) lists; numbers are Fortran ``5e16.9`` fields that may run
together without blanks.
"""

from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

MAX_GEQDSK_BYTES = 10 * 1024 * 1024
MAX_GEQDSK_GRID_POINTS = 1_000_000
MAX_GEQDSK_CONTOUR_POINTS = 100_000
_FLOAT_RE = re.compile(r"[+-]?\d*\.?\d+(?:[eEdD][+-]?\d+)?")


GEQDSK_SOURCE_CONVENTION_MODES = {"raw_canonical": "raw_canonical", "public_sparc_named_adapter": "public_sparc_named_adapter"}
# public SPARC files whose ffprime / pprime are stored per-radian (core/eqdsk.py:44-53)
GEQDSK_PUBLIC_SPARC_SOURCE_ADAPTERS = {"sparc_1305.eqdsk": "scaled_by_2pi", "sparc_1310.eqdsk": "scaled_by_2pi",
                                       "sparc_1315.eqdsk": "scaled_by_2pi", "sparc_1349.eqdsk": "scaled_by_2pi"}
GEQDSK_SOURCE_CONVENTION_ADAPTERS = {"scaled_by_2pi": 2.0 * np.pi}


def _empty():
    return np.array([], dtype=np.float64)


@dataclass
class GEqdsk:
    description: str = ""
    nw: int = 0
    nh: int = 0
    rdim: float = 0.0
    zdim: float = 0.0
    rcentr: float = 0.0
    rleft: float = 0.0
    zmid: float = 0.0
    rmaxis: float = 0.0
    zmaxis: float = 0.0
    simag: float = 0.0
    sibry: float = 0.0
    bcentr: float = 0.0
    current: float = 0.0
    fpol: np.ndarray = field(default_factory=_empty)
    pres: np.ndarray = field(default_factory=_empty)
    ffprime: np.ndarray = field(default_factory=_empty)
    pprime: np.ndarray = field(default_factory=_empty)
    qpsi: np.ndarray = field(default_factory=_empty)
    psirz: np.ndarray = field(default_factory=_empty)
    rbdry: np.ndarray = field(default_factory=_empty)
    zbdry: np.ndarray = field(default_factory=_empty)
    rlim: np.ndarray = field(default_factory=_empty)
    zlim: np.ndarray = field(default_factory=_empty)
    source_convention: str = "raw_canonical"
    source_convention_adapter: str = "none"
    source_convention_adapter_pass: bool = True
    source_convention_metadata: dict = field(default_factory=dict)

    @property
    def psi_norm(self) -> np.ndarray:
        return np.linspace(0.0, 1.0, self.nw)

    @property
    def r(self) -> np.ndarray:
        return np.linspace(self.rleft, self.rleft + self.rdim, self.nw)

    @property
    def z(self) -> np.ndarray:
        return np.linspace(self.zmid - self.zdim / 2, self.zmid + self.zdim / 2, self.nh)

    def psi_to_norm(self, psi):
        return (psi - self.simag) / (self.sibry - self.simag)

    def to_config(self, name: str = "eqdsk") -> dict:
        """FusionKernel config: grid = the file's grid, I_p in MA, normalised mu0 = 1, no coils, the EFIT boundary /
        limiter exported as an isoflux shape contract (core/eqdsk.py:128-196)."""
        r, z = self.r, self.z
        fb: dict = {"magnetic_axis": [float(self.rmaxis), float(self.zmaxis)], "psi_axis": float(self.simag),
                    "psi_boundary": float(self.sibry), "boundary_source": "geqdsk_rbdry_zbdry"}
        if self.rbdry.size and self.zbdry.size:
            if self.rbdry.shape != self.zbdry.shape:
                raise ValueError("GEQDSK boundary R/Z arrays must have matching lengths.")
            pts = np.column_stack([self.rbdry, self.zbdry]).astype(np.float64)
            fb["target_flux_points"] = pts.tolist()
            fb["target_flux_values"] = np.full(pts.shape[0], float(self.sibry)).tolist()
        if self.rlim.size or self.zlim.size:
            if self.rlim.shape != self.zlim.shape:
                raise ValueError("GEQDSK limiter R/Z arrays must have matching lengths.")
            fb["limiter_points"] = np.column_stack([self.rlim, self.zlim]).astype(np.float64).tolist()
        return {
            "reactor_name": name,
            "grid_resolution": [self.nw, self.nh],
            "dimensions": {"R_min": float(r[0]), "R_max": float(r[-1]), "Z_min": float(z[0]), "Z_max": float(z[-1])},
            "physics": {"plasma_current_target": float(self.current / 1e6), "vacuum_permeability": 1.0},
            "coils": [],
            "free_boundary": fb,
            "solver": {"max_iterations": 1000, "convergence_threshold": 1e-4, "relaxation_factor": 0.1},
        }


def _validate(eq: GEqdsk) -> None:
    if eq.nw < 2 or eq.nh < 2:
        raise ValueError(f"GEQDSK grid dimensions must be >= 2x2, got {(eq.nw, eq.nh)}")
    if eq.nw * eq.nh > MAX_GEQDSK_GRID_POINTS:
        raise ValueError("GEQDSK grid dimensions exceed safety limit")
    if eq.rdim <= 0.0 or eq.zdim <= 0.0:
        raise ValueError("GEQDSK rdim and zdim must be positive.")
    if eq.rcentr <= 0.0:
        raise ValueError("GEQDSK rcentr must be positive.")
    if eq.sibry == eq.simag:
        raise ValueError("GEQDSK psi boundary must differ from psi axis.")
    for name in ("fpol", "pres", "ffprime", "pprime", "qpsi"):
        a = getattr(eq, name)
        if a.shape != (eq.nw,) or not np.all(np.isfinite(a)):
            raise ValueError(f"GEQDSK {name} must be a finite array of shape {(eq.nw,)}.")
    if eq.psirz.shape != (eq.nh, eq.nw) or not np.all(np.isfinite(eq.psirz)):
        raise ValueError(f"GEQDSK psirz must be a finite array of shape {(eq.nh, eq.nw)}.")


def _apply_source_adapter(eq: GEqdsk, *, convention: str, source_file: Path, mode: str) -> GEqdsk:
    """Documented source-convention adapter with provenance (core/eqdsk.py:295-346)."""
    multiplier = GEQDSK_SOURCE_CONVENTION_ADAPTERS.get(convention)
    if multiplier is None:
        eq.source_convention, eq.source_convention_adapter, eq.source_convention_adapter_pass = "unsupported", convention, False
        eq.source_convention_metadata = {"requested_mode": mode, "source_file": source_file.name, "adapter": convention,
                                         "applied_scale": float("nan"), "provenance": "unsupported_named_adapter",
                                         "error": "unsupported source convention adapter"}
        return eq
    eq.ffprime = np.round(multiplier * eq.ffprime, decimals=11)
    eq.pprime = np.round(multiplier * eq.pprime, decimals=11)
    eq.source_convention, eq.source_convention_adapter, eq.source_convention_adapter_pass = "canonical", convention, True
    eq.source_convention_metadata = {"requested_mode": mode, "source_file": source_file.name, "adapter": convention,
                                     "applied_scale": float(multiplier), "provenance": "public_sparc_named_adapter",
                                     "public_case": source_file.name}
    return eq


def read_geqdsk(path, *, source_convention_mode: str = "raw_canonical") -> GEqdsk:
    """core/eqdsk.py:349-539.  ``source_convention_mode="public_sparc_named_adapter"`` rescales ffprime / pprime of the
    explicitly recognised public SPARC files by 2*pi and records the provenance; anything else stays raw."""
    path = Path(path)
    if source_convention_mode not in GEQDSK_SOURCE_CONVENTION_MODES:
        raise ValueError(f"Unsupported source_convention_mode: {source_convention_mode}. Expected one of: "
                         + ", ".join(sorted(GEQDSK_SOURCE_CONVENTION_MODES)))
    if path.stat().st_size > MAX_GEQDSK_BYTES:
        raise ValueError("GEQDSK file too large")
    lines = path.read_text(encoding="utf-8").splitlines()
    if not lines:
        raise ValueError("GEQDSK file is empty")
    head = lines[0].split()
    if len(head) < 3:
        raise ValueError("GEQDSK header must contain idum, nw, and nh")
    nh, nw = int(head[-1]), int(head[-2])
    desc = " ".join(head[:-3]) if len(head) > 3 else ""
    if nw < 2 or nh < 2 or nw * nh > MAX_GEQDSK_GRID_POINTS:
        raise ValueError(f"GEQDSK grid dimensions out of range: {(nw, nh)}")
    tokens: list[str] = []
    for line in lines[1:]:
        tokens.extend(_FLOAT_RE.findall(line))
    pos = 0

    def take(n: int) -> np.ndarray:
        nonlocal pos
        if n < 0 or pos + n > len(tokens):
            raise ValueError("GEQDSK file ended before all required values were present")
        vals = [float(t.replace("D", "E").replace("d", "e")) for t in tokens[pos:pos + n]]
        if not all(math.isfinite(v) for v in vals):
            raise ValueError("GEQDSK values must be finite.")
        pos += n
        return np.array(vals, dtype=np.float64)

    s = take(20)
    eq = GEqdsk(description=desc, nw=nw, nh=nh, rdim=s[0], zdim=s[1], rcentr=s[2], rleft=s[3], zmid=s[4], rmaxis=s[5],
                zmaxis=s[6], simag=s[7], sibry=s[8], bcentr=s[9], current=s[10])
    eq.fpol, eq.pres, eq.ffprime, eq.pprime = take(nw), take(nw), take(nw), take(nw)
    eq.psirz = take(nh * nw).reshape(nh, nw)
    eq.qpsi = take(nw)
    if pos + 2 <= len(tokens):
        nb, nl = int(take(1)[0]), int(take(1)[0])
        if not (0 <= nb <= MAX_GEQDSK_CONTOUR_POINTS and 0 <= nl <= MAX_GEQDSK_CONTOUR_POINTS):
            raise ValueError("GEQDSK contour count out of range")
        bd = take(2 * nb)
        eq.rbdry, eq.zbdry = bd[0::2].copy(), bd[1::2].copy()
        lm = take(2 * nl)
        eq.rlim, eq.zlim = lm[0::2].copy(), lm[1::2].copy()
    _validate(eq)
    if source_convention_mode == "public_sparc_named_adapter":
        adapter = GEQDSK_PUBLIC_SPARC_SOURCE_ADAPTERS.get(path.name.lower())
        if adapter is None:
            eq.source_convention_adapter, eq.source_convention_adapter_pass, eq.source_convention = "no_named_adapter", False, "raw_canonical"
            eq.source_convention_metadata = {"requested_mode": source_convention_mode, "source_file": path.name,
                                             "adapter": "no_named_adapter", "applied_scale": 1.0,
                                             "provenance": "public_sparc_named_adapter_no_match",
                                             "error": "no recognized public SPARC convention mapping"}
        else:
            eq = _apply_source_adapter(eq, convention=adapter, source_file=path, mode=source_convention_mode)
    return eq


def write_geqdsk(eq: GEqdsk, path) -> None:
    """Standard (5e16.9) layout; read_geqdsk(write_geqdsk(eq)) reproduces every value to the 10 printed digits."""
    _validate(eq)

    def block(values) -> str:
        v = np.asarray(values, dtype=np.float64).ravel()
        out = []
        for i in range(0, v.size, 5):
            out.append("".join(f"{x:16.9E}" for x in v[i:i + 5]))
        return "\n".join(out) + ("\n" if v.size else "")

    with open(path, "w", encoding="utf-8") as f:
        f.write(f"{eq.description[:48]:48s}{0:4d}{eq.nw:4d}{eq.nh:4d}\n")
        f.write(block([eq.rdim, eq.zdim, eq.rcentr, eq.rleft, eq.zmid]))
        f.write(block([eq.rmaxis, eq.zmaxis, eq.simag, eq.sibry, eq.bcentr]))
        f.write(block([eq.current, eq.simag, 0.0, eq.rmaxis, 0.0]))
        f.write(block([eq.zmaxis, 0.0, eq.sibry, 0.0, 0.0]))
        for arr in (eq.fpol, eq.pres, eq.ffprime, eq.pprime, eq.psirz, eq.qpsi):
            f.write(block(arr))
        f.write(f"{eq.rbdry.size:5d}{eq.rlim.size:5d}\n")
        f.write(block(np.column_stack([eq.rbdry, eq.zbdry]) if eq.rbdry.size else []))
        f.write(block(np.column_stack([eq.rlim, eq.zlim]) if eq.rlim.size else []))


__all__ = ["GEqdsk", "read_geqdsk", "write_geqdsk"]

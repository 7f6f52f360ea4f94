"""Restatement of the FreeGS 0.8.2 ``TestTokamak`` coil set as point filaments -- test infrastructure.

FreeGS (third-party, pinned 0.8.2 by the reference's tracked artifact ``freegs_public_example_reconstruction_attempt.json``) is
neither vendored under /root/reference nor installed, so the machine is restated from its published definition
(freegs/machine.py ``TestTokamak``: shaped coils P1L / P1U = the unit squares of side 0.1 m centred at (1.0, -1.1) / (1.0, +1.1),
point coils P2L / P2U at (1.75, -0.6) / (1.75, +0.6); freegs/quadrature.py: a shaped coil is represented by the six-point
degree-4 Gauss rule on each triangle of its polygon, weights scaled by the triangles' area shares -- 12 filaments per square,
as the artifact's ``filament_count`` records).  A square has two triangulations; which diagonal FreeGS's ear clipping cuts
depends on the polygon's winding, and the two squares are mirror images: P1L is fanned from its vertex 1, P1U from its vertex 0.
With the coil currents stored in the artifact this reproduces the nine ``freegs_vacuum_psi_sample`` values to 6e-16 Wb -- inside
the artifact's own 1e-12 thresholds (rounds 1-4 fanned both squares from vertex 1 and stopped at 3.75e-10; the other three
combinations of diagonals give 3.75e-10 ... 6.96e-10, a degree-3 six-point rule 1e-7).
"""
import numpy as np

# barycentric orbit (a, b, b) and weight of the two orbits of the six-point rule
_TRI6 = ((0.108103018168070, 0.445948490915965, 0.223381589678011), (0.816847572980459, 0.091576213509771, 0.109951743655322))


def _rule():
    pts = []
    for a, b, w in _TRI6:
        pts += [(a, b, b, w), (b, a, b, w), (b, b, a, w)]
    return pts


def _shaped(polygon, apex=1):
    """Filaments (r, z, weight) of a convex polygon: fan triangulation from vertex ``apex``, six Gauss points per triangle."""
    n = len(polygon)
    tris = [(polygon[apex], polygon[(apex + i) % n], polygon[(apex + 1 + i) % n]) for i in range(1, n - 1)]
    area = [0.5 * abs((t[1][0] - t[0][0]) * (t[2][1] - t[0][1]) - (t[2][0] - t[0][0]) * (t[1][1] - t[0][1])) for t in tris]
    out = []
    for t, a in zip(tris, area):
        for l0, l1, l2, w in _rule():
            out.append((l0 * t[0][0] + l1 * t[1][0] + l2 * t[2][0], l0 * t[0][1] + l1 * t[1][1] + l2 * t[2][1], w * a / sum(area)))
    return out


def tokamak_filaments(currents: dict) -> np.ndarray:
    """(n, 3) array of (R, Z, current) filaments of TestTokamak for the given coil currents {"P1L": ..., ...}."""
    p1l = [(0.95, -1.15), (0.95, -1.05), (1.05, -1.05), (1.05, -1.15)]
    p1u = [(0.95, 1.15), (0.95, 1.05), (1.05, 1.05), (1.05, 1.15)]
    fil = [(r, z, currents["P1L"] * w) for r, z, w in _shaped(p1l, 1)] + [(r, z, currents["P1U"] * w) for r, z, w in _shaped(p1u, 0)]
    fil += [(1.75, -0.6, currents["P2L"]), (1.75, 0.6, currents["P2U"])]
    return np.asarray(fil, dtype=np.float64)

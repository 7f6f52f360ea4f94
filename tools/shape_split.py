"""Where the shape-optimisation variant of BASELINE config 3 (8 isoflux points, DIII-D-like 513x513, ONE member) spends its time: cProfile of one run."""
import copy
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from scpn_fusion_core_amd.cases import diiid_like_cfg  # noqa: E402
from scpn_fusion_core_amd.free_boundary import CoilSet  # noqa: E402
from scpn_fusion_core_amd.fusion_kernel import FusionKernel  # noqa: E402

cs = copy.deepcopy(diiid_like_cfg(513, tol=1e-8, max_iter=1000))
cs["physics"]["vacuum_permeability"] = 4e-7 * np.pi
cs["physics"]["plasma_current_target"] = 1.6e6
for c in cs["coils"]:
    c["current"] = c["current"] * 1.0e6
th = np.linspace(0.0, 2.0 * np.pi, 8, endpoint=False)
pts = np.column_stack([1.67 + 0.6 * np.cos(th), 1.8 * 0.6 * np.sin(th)])


def run():
    k = FusionKernel(copy.deepcopy(cs))
    coils = CoilSet(positions=[(c["r"], c["z"]) for c in cs["coils"]], currents=np.array([c["current"] for c in cs["coils"]], dtype=float),
                    turns=[1] * len(cs["coils"]), current_limits=np.full(len(cs["coils"]), 5.0e7), target_flux_points=pts)
    t = time.perf_counter()
    r = k.solve_free_boundary(coils, max_outer_iter=4, tol=1e-6, optimize_shape=True)
    dt = time.perf_counter() - t
    k.close()
    return dt, r


run()
print("ms per solve:", [round(1e3 * run()[0], 1) for _ in range(3)])
pr = cProfile.Profile()
pr.enable()
dt, r = run()
pr.disable()
print("outer iterations", r["outer_iterations"], "profiled run ms", round(1e3 * dt, 1))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

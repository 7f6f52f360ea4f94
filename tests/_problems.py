"""Seeded problem builders shared by the CPU and GPU parity tests (no reference import).  The benchmark cases themselves
live in the product package (scpn_fusion_core_amd/cases.py); this module adds the fixture helpers."""
import json

import numpy as np

from scpn_fusion_core_amd.cases import bench_problem, grid, iter_like_cfg, member_cfg, scan_member_table  # noqa: F401


def operator_problem(nz, nr):
    """Inputs of tests/golden/operators.npz (same construction as make_golden.gen_operators)."""
    psi0, source, rr, r, z = bench_problem(nz, nr)
    zz = np.repeat(z[:, None], nr, axis=1)
    psi = psi0 + 0.01 * np.sin(rr) * np.cos(0.3 * zz)
    return psi, source, rr, float(r[1] - r[0]), float(z[1] - z[0])


def configs_from(npz):
    return json.loads(str(npz["_configs"]))

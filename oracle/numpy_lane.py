"""NumPy restatement of the elliptic kernels of the hot path -- TEST INFRASTRUCTURE and CPU-baseline lane, never the product.

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import this module ("a NumPy restatement of the 8a formulas
written in the build's repo, validated against the imported reference via fixtures").  Each function keeps the floating-point
expression (operand order, division) of the reference function it names, so the results are bit-identical to the fixtures of
``tests/golden`` (checked by tests/test_oracle_golden.py::test_numpy_lane_*), but the array traversal is this repo's own:
the red-black passes run on the four strided sub-lattices of a colour instead of boolean-mask gathers, which is ~6x faster
than the reference's NumPy lane and therefore the stronger CPU baseline of the two.

  rb_sor_sweeps      core/multigrid_solve.py:148-208   (mg_smooth)
  gs_residual        core/multigrid_solve.py:211-249   (mg_residual)
  restrict_fw        core/multigrid_solve.py:57-99     (restrict_full_weight)
  prolong_bilinear   core/multigrid_solve.py:102-145   (prolongate_bilinear)
  vcycle             core/multigrid_solve.py:252-335   (multigrid_vcycle)
  picard_iteration   core/fusion_kernel_newton_solver.py:499-582 with fusion_kernel.py:255-444 (one iteration, L-mode)
"""
from __future__ import annotations

import numpy as np


def _coefficients(r_grid: np.ndarray, dr: float, dz: float):
    dr2, dz2 = dr**2, dz**2
    r_safe = np.maximum(r_grid[1:-1, 1:-1], 1e-10)
    half = 1.0 / (2.0 * r_safe * dr)
    return 1.0 / dr2 - half, 1.0 / dr2 + half, 1.0 / dz2, 2.0 / dr2 + 2.0 / dz2


def rb_sor_sweeps(psi: np.ndarray, source: np.ndarray, r_grid: np.ndarray, dr: float, dz: float, omega: float, n_sweeps: int) -> np.ndarray:
    """``n_sweeps`` red-black SOR sweeps of the toroidal 5-point stencil, in place (parity of the global index sum, colour 0 first)."""
    if not (np.isfinite(omega) and 1.0 <= omega < 2.0):
        raise ValueError("omega must satisfy 1 <= omega < 2")
    a_e, a_w, a_ns, a_c = _coefficients(r_grid, dr, dz)
    nz, nr = psi.shape
    om1 = 1.0 - omega
    for _ in range(int(n_sweeps)):
        for parity in (0, 1):
            for i0 in (1, 2):  # interior rows i0, i0+2, ...; their columns of this colour start at j0
                j0 = 1 + ((i0 + 1 + parity) % 2)
                rows, cols = slice(i0, nz - 1, 2), slice(j0, nr - 1, 2)
                c_rows, c_cols = slice(i0 - 1, nz - 2, 2), slice(j0 - 1, nr - 2, 2)  # the same points in the interior-sized tables
                if psi[rows, cols].size == 0:
                    continue
                east = psi[rows, j0 + 1 : nr : 2]
                west = psi[rows, j0 - 1 : nr - 2 : 2]
                south = psi[i0 - 1 : nz - 2 : 2, cols]
                north = psi[i0 + 1 : nz : 2, cols]
                gs = (a_e[c_rows, c_cols] * east + a_w[c_rows, c_cols] * west + a_ns * south + a_ns * north
                      - source[rows, cols]) / a_c
                psi[rows, cols] = om1 * psi[rows, cols] + omega * gs
    return psi


def gs_residual(psi: np.ndarray, source: np.ndarray, r_grid: np.ndarray, dr: float, dz: float) -> np.ndarray:
    """r = L*psi - source on the interior, 0 on the ring."""
    out = np.zeros_like(psi)
    mid = psi[1:-1, 1:-1]
    r_safe = np.maximum(r_grid[1:-1, 1:-1], 1e-10)
    d2r = (psi[1:-1, 2:] - 2.0 * mid + psi[1:-1, :-2]) / dr**2
    d1r = (psi[1:-1, 2:] - psi[1:-1, :-2]) / (2.0 * dr)
    d2z = (psi[2:, 1:-1] - 2.0 * mid + psi[:-2, 1:-1]) / dz**2
    out[1:-1, 1:-1] = d2r - d1r / r_safe + d2z - source[1:-1, 1:-1]
    return out


def restrict_fw(fine: np.ndarray) -> np.ndarray:
    """Full weighting 1/16 [1 2 1; 2 4 2; 1 2 1] on the interior of the ((n+1)//2)^2 coarse grid, ring injected."""
    nz_c, nr_c = (fine.shape[0] + 1) // 2, (fine.shape[1] + 1) // 2
    coarse = np.zeros((nz_c, nr_c))
    c = lambda dz, dr: fine[2 + dz : 2 + dz + 2 * (nz_c - 2) : 2, 2 + dr : 2 + dr + 2 * (nr_c - 2) : 2]  # noqa: E731
    coarse[1:-1, 1:-1] = (4.0 * c(0, 0) + 2.0 * (c(-1, 0) + c(1, 0) + c(0, -1) + c(0, 1))
                          + (c(-1, -1) + c(-1, 1) + c(1, -1) + c(1, 1))) / 16.0
    # ring: injection of every second fine ring value (first / last fine row and column)
    # (rows first, then columns: for even fine sizes the corners take the column's value)
    for edge in (0, -1):
        coarse[edge, :] = fine[edge, ::2][:nr_c]
    for edge in (0, -1):
        coarse[:, edge] = fine[::2, edge][:nz_c]
    return coarse


def prolong_bilinear(coarse: np.ndarray, nz_f: int, nr_f: int) -> np.ndarray:
    """Bilinear interpolation to (nz_f, nr_f): coincident copy, 2-point midpoints, 4-point centres (writes the ring too)."""
    nz_c, nr_c = coarse.shape
    fine = np.zeros((nz_f, nr_f))
    ez, er = min(2 * (nz_c - 1), nz_f - 1), min(2 * (nr_c - 1), nr_f - 1)  # last fine index with a coarse partner on both sides
    fine[: 2 * nz_c : 2, : 2 * nr_c : 2][: (nz_f + 1) // 2, : (nr_f + 1) // 2] = coarse[: (nz_f + 1) // 2, : (nr_f + 1) // 2]
    kz, kr = len(range(1, ez, 2)), len(range(1, er, 2))
    fine[0 : 2 * nz_c : 2, 1:er:2][: (nz_f + 1) // 2] = 0.5 * (coarse[: (nz_f + 1) // 2, :kr] + coarse[: (nz_f + 1) // 2, 1 : kr + 1])
    fine[1:ez:2, 0 : 2 * nr_c : 2][:, : (nr_f + 1) // 2] = 0.5 * (coarse[:kz, : (nr_f + 1) // 2] + coarse[1 : kz + 1, : (nr_f + 1) // 2])
    fine[1:ez:2, 1:er:2] = 0.25 * (coarse[:kz, :kr] + coarse[1 : kz + 1, :kr] + coarse[:kz, 1 : kr + 1] + coarse[1 : kz + 1, 1 : kr + 1])
    return fine


def vcycle(psi: np.ndarray, source: np.ndarray, r_grid: np.ndarray, dr: float, dz: float, *, omega: float = 1.0, pre: int = 3,
           post: int = 3, min_grid: int = 5) -> np.ndarray:
    """One recursive V-cycle: base case (min_grid >= nz or nr) = 50 sweeps on a copy; else pre-smooth, restrict the negated
    residual and R, solve the coarse correction from zero at doubled spacing, add its prolongation, post-smooth."""
    nz, nr = psi.shape
    if min_grid >= nz or min_grid >= nr:
        return rb_sor_sweeps(psi.copy(), source, r_grid, dr, dz, omega, 50)
    psi = rb_sor_sweeps(psi.copy(), source, r_grid, dr, dz, omega, pre)
    defect = restrict_fw(-gs_residual(psi, source, r_grid, dr, dz))
    r_coarse = restrict_fw(r_grid)
    err = vcycle(np.zeros_like(defect), defect, r_coarse, 2.0 * dr, 2.0 * dz, omega=omega, pre=pre, post=post, min_grid=min_grid)
    psi = psi + prolong_bilinear(err, nz, nr)
    return rb_sor_sweeps(psi, source, r_grid, dr, dz, omega, post)


def picard_iteration(psi: np.ndarray, psi_bc: np.ndarray, rr: np.ndarray, zz: np.ndarray, dr: float, dz: float, *, i_target: float,
                     mu0: float, z_min_cfg: float, alpha: float, omega: float = 1.0) -> tuple[np.ndarray, float]:
    """One multigrid Picard iteration of the L-mode lane: axis / X-point, J_phi renormalised to ``i_target``, one V-cycle,
    ring from ``psi_bc``, mean |d psi| and the under-relaxed blend.  Returns (psi_next, diff)."""
    psi_axis = float(np.max(psi))
    if abs(psi_axis) < 1e-6:
        psi_axis = 1e-6
    g_z, g_r = np.gradient(np.nan_to_num(psi, nan=0.0, posinf=1e300, neginf=-1e300), dz, dr)
    b_mag = np.hypot(g_r, g_z)
    mask = zz < 0.5 * z_min_cfg
    psi_bnd = float(psi.flat[int(np.argmin(np.where(mask, b_mag, np.inf)))]) if mask.any() else float(np.min(psi))
    if abs(psi_axis - psi_bnd) < 0.1:
        psi_bnd = 0.1 * psi_axis
    den = psi_bnd - psi_axis
    if abs(den) < 1e-9:
        den = 1e-9
    psi_n = (psi - psi_axis) / den
    prof = np.where((psi_n >= 0.0) & (psi_n < 1.0), 1.0 - psi_n, 0.0)
    j_raw = 0.5 * (rr * prof) + 0.5 * (prof / (mu0 * rr))
    total = float(np.sum(j_raw)) * dr * dz
    j_phi = j_raw * (i_target / total) if abs(total) > 1e-9 else np.zeros_like(j_raw)
    new = vcycle(psi, -mu0 * rr * j_phi, rr, dr, dz, omega=omega)
    new[0, :], new[-1, :], new[:, 0], new[:, -1] = psi_bc[0, :], psi_bc[-1, :], psi_bc[:, 0], psi_bc[:, -1]
    diff = float(np.mean(np.abs(new - psi)))
    return (1.0 - alpha) * psi + alpha * new, diff


def polyglot_picard_jacobi(r_min: float, r_max: float, z_min: float, z_max: float, nr: int, nz: int, ip_target: float, *, mu0: float = 4e-7 * np.pi,
                           n_picard: int = 80, n_jacobi: int = 200, alpha: float = 0.1, omega_j: float = 0.667, beta_mix: float = 0.5) -> np.ndarray:
    """The "polyglot" fixed-boundary lane in NumPy (core/jax_gs_solver.py:58-167 ``gs_solve_np``: zero ring, Gaussian seed,
    ``n_picard`` x [L-mode source normalised to ``ip_target`` from the interior maximum, ``n_jacobi`` damped-Jacobi steps, blend]).
    The lane the reference's benchmarks/polyglot_gs_solver_comparison.py times as its Python row -- here as the NumPy comparator of
    ``polyglot.gs_solve_hip`` at 513 x 513 (bench.py ``configs.polyglot_picard_jacobi_513``).  The stencil coefficients are rebuilt
    in every Jacobi step, as the reference does (its per-step cost is what is being compared)."""
    r_axis, z_axis = np.linspace(r_min, r_max, nr), np.linspace(z_min, z_max, nz)
    rr = np.meshgrid(r_axis, z_axis)[0]
    dr, dz = r_axis[1] - r_axis[0], z_axis[1] - z_axis[0]
    inner = (slice(1, -1), slice(1, -1))
    psi = 0.01 * np.exp(-((rr - 0.5 * (r_min + r_max)) ** 2) / 0.5)
    psi[[0, -1], :] = 0.0
    psi[:, [0, -1]] = 0.0
    for _ in range(n_picard):
        # source: psi_N against the interior maximum and a zero boundary flux, clipped to [0, 1]
        span = 0.0 - float(psi[inner].max())
        if abs(span) < 1e-9:
            span = float(np.sign(span)) * 1e-9 if span != 0 else 1e-9
        psi_n = np.clip((psi - float(psi[inner].max())) / span, 0.0, 1.0)
        shape = np.where((psi_n >= 0.0) & (psi_n < 1.0), 1.0 - psi_n, 0.0)
        j_raw = beta_mix * (rr * shape) + (1.0 - beta_mix) * (shape / (mu0 * np.maximum(rr, 1e-10)))
        rhs = -mu0 * rr * (j_raw * (ip_target / max(abs(np.sum(j_raw) * dr * dz), 1e-9)))
        work = psi.copy()
        for _ in range(n_jacobi):
            r_safe = np.maximum(rr[inner], 1e-10)
            dr2, dz2 = dr * dr, dz * dz
            east, west = 1.0 / dr2 - 1.0 / (2.0 * r_safe * dr), 1.0 / dr2 + 1.0 / (2.0 * r_safe * dr)
            gs = (east * work[1:-1, 2:] + west * work[1:-1, :-2] + (1.0 / dz2) * (work[:-2, 1:-1] + work[2:, 1:-1]) - rhs[inner]) / (2.0 / dr2 + 2.0 / dz2)
            nxt = work.copy()
            nxt[inner] = (1.0 - omega_j) * work[inner] + omega_j * gs
            work = nxt
        psi = (1.0 - alpha) * psi + alpha * work
    return psi

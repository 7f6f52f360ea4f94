/*
 * gs_oracle.c -- CPU restatement of the reference Grad-Shafranov hot path (see gs_oracle.h).
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker; never by the product path.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout, src/scpn_fusion/...).  Operation order follows the NumPy expressions
 * exactly; compile with -ffp-contract=off.
 */
#include "gs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CAP 1.0e250 /* core/fusion_kernel_numerics.py:16 NUMERIC_SANITIZE_CAP */
#define IDX(iz, ir) ((long)(iz) * nr + (ir))

static double *dalloc(long n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

/* numpy/_core/function_base.py linspace: y = arange(n)*step + start; y[-1] = stop */
void orc_linspace(double start, double stop, int n, double *out) {
    if (n <= 0) return;
    if (n == 1) { out[0] = start; return; }
    double div = (double)(n - 1);
    double delta = stop - start;
    double step = delta / div;
    for (int i = 0; i < n; ++i) {
        double y = (double)i;
        if (step != 0.0) y = y * step;
        else y = (y / div) * delta;
        out[i] = y + start;
    }
    out[n - 1] = stop;
}

/* numpy/_core/src/umath/loops_utils.h.src  DOUBLE_pairwise_sum (PW_BLOCKSIZE 128) */
static double pairwise(const double *a, long n) {
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise(a, n2) + pairwise(a + n2, n - n2);
    }
}
/* np.add.reduce over a contiguous vector: the ufunc machinery hands the inner loop at most
 * NPY_BUFSIZE (8192) elements at a time and accumulates the pairwise sums of the chunks
 * sequentially (verified against numpy 2.2.6 in tests/test_oracle_golden.py). */
double orc_np_sum(const double *a, long n) {
    double s = 0.0;
    for (long i = 0; i < n; i += 8192) s = s + pairwise(a + i, (n - i) < 8192 ? (n - i) : 8192);
    return s;
}

static int omega_ok(double omega) { /* core/multigrid_solve.py:33-54 */
    return isfinite(omega) && omega >= 1.0 && omega < 2.0;
}

/* interior coefficient tables a_E, a_W  -- core/multigrid_solve.py:184-189 */
static void coeffs(const double *r_grid, int nz, int nr, double dr, double dr2, double *a_e,
                   double *a_w) {
    int ni = nr - 2;
    for (int iz = 1; iz < nz - 1; ++iz)
        for (int ir = 1; ir < nr - 1; ++ir) {
            double r_safe = fmax(r_grid[IDX(iz, ir)], 1e-10);
            double t = 1.0 / ((2.0 * r_safe) * dr);
            a_e[(long)(iz - 1) * ni + (ir - 1)] = 1.0 / dr2 - t;
            a_w[(long)(iz - 1) * ni + (ir - 1)] = 1.0 / dr2 + t;
        }
}

/* one red-black sweep; clip!=0 adds the +-CAP clip of _sor_step (iterative_solver.py:158) */
static void rb_sweep(double *psi, const double *source, const double *a_e, const double *a_w,
                     double a_ns, double a_c, int nz, int nr, double omega, int clip) {
    int ni = nr - 2;
    for (int parity = 0; parity < 2; ++parity)
        for (int iz = 1; iz < nz - 1; ++iz)
            for (int ir = 1; ir < nr - 1; ++ir) {
                /* mask = ((ii + jj) % 2) == parity with ii,jj the GLOBAL indices
                 * (np.mgrid[1:nz-1, 1:nr-1]) -- core/multigrid_solve.py:191-195 */
                if (((iz + ir) & 1) != parity) continue;
                long c = (long)(iz - 1) * ni + (ir - 1);
                double gs = (a_e[c] * psi[IDX(iz, ir + 1)] + a_w[c] * psi[IDX(iz, ir - 1)] +
                             a_ns * psi[IDX(iz - 1, ir)] + a_ns * psi[IDX(iz + 1, ir)] -
                             source[IDX(iz, ir)]) /
                            a_c;
                double old = psi[IDX(iz, ir)];
                double v = (1.0 - omega) * old + omega * gs;
                if (clip) v = fmin(fmax(v, -CAP), CAP);
                psi[IDX(iz, ir)] = v;
            }
}

int orc_mg_smooth(double *psi, const double *source, const double *r_grid, int nz, int nr,
                  double dr, double dz, double omega, int n_sweeps) {
    if (!omega_ok(omega)) return -1;
    if (nz < 3 || nr < 3) return 0;
    double dr2 = pow(dr, 2.0), dz2 = pow(dz, 2.0);
    long ni = (long)(nz - 2) * (nr - 2);
    double *a_e = dalloc(ni), *a_w = dalloc(ni);
    coeffs(r_grid, nz, nr, dr, dr2, a_e, a_w);
    double a_ns = 1.0 / dz2;
    double a_c = 2.0 / dr2 + 2.0 / dz2;
    for (int s = 0; s < n_sweeps; ++s) rb_sweep(psi, source, a_e, a_w, a_ns, a_c, nz, nr, omega, 0);
    free(a_e);
    free(a_w);
    return 0;
}

void orc_mg_residual(const double *psi, const double *source, const double *r_grid, int nz,
                     int nr, double dr, double dz, double *out) {
    double dr2 = pow(dr, 2.0), dz2 = pow(dz, 2.0);
    memset(out, 0, sizeof(double) * (size_t)nz * nr);
    for (int iz = 1; iz < nz - 1; ++iz)
        for (int ir = 1; ir < nr - 1; ++ir) {
            double r_safe = fmax(r_grid[IDX(iz, ir)], 1e-10);
            double c = psi[IDX(iz, ir)];
            double e = psi[IDX(iz, ir + 1)], w = psi[IDX(iz, ir - 1)];
            double n = psi[IDX(iz + 1, ir)], s = psi[IDX(iz - 1, ir)];
            double d2r = (e - 2.0 * c + w) / dr2;
            double d1r = (e - w) / (2.0 * dr);
            double d2z = (n - 2.0 * c + s) / dz2;
            double lpsi = d2r - d1r / r_safe + d2z;
            out[IDX(iz, ir)] = lpsi - source[IDX(iz, ir)];
        }
}

double orc_residual_linf(const double *psi, const double *source, const double *r_grid, int nz,
                         int nr, double dr, double dz) {
    if (nz < 3 || nr < 3) return 0.0;
    double *r = dalloc((long)nz * nr);
    orc_mg_residual(psi, source, r_grid, nz, nr, dr, dz, r);
    double m = 0.0;
    int any = 0;
    /* np.max(np.abs(residual[1:-1,1:-1])): NaN propagates */
    for (int iz = 1; iz < nz - 1; ++iz)
        for (int ir = 1; ir < nr - 1; ++ir) {
            double a = fabs(r[IDX(iz, ir)]);
            if (!any) { m = a; any = 1; }
            else if (isnan(a) || a > m) { if (!isnan(m)) m = a; }
        }
    free(r);
    return m;
}

double orc_stable_rms(const double *v, long n) {
    if (n == 0) return 0.0;
    double *vals = dalloc(n);
    double max_abs = 0.0;
    for (long i = 0; i < n; ++i) { /* sanitize_numeric_array */
        double x = v[i];
        if (isnan(x)) x = 0.0;
        else if (isinf(x)) x = x > 0 ? CAP : -CAP;
        vals[i] = x;
        if (fabs(x) > max_abs) max_abs = fabs(x);
    }
    if (max_abs > CAP) {
        for (long i = 0; i < n; ++i) vals[i] = fmin(fmax(vals[i], -CAP), CAP);
        max_abs = CAP;
    }
    if (max_abs <= 0.0) { free(vals); return 0.0; }
    for (long i = 0; i < n; ++i) {
        double s = vals[i] / max_abs;
        vals[i] = s * s;
    }
    double mean = orc_np_sum(vals, n) / (double)n;
    free(vals);
    return max_abs * sqrt(mean);
}

double orc_gs_residual_rms(const double *psi, const double *source, const double *r_grid, int nz,
                           int nr, double dr, double dz) {
    if (nz < 3 || nr < 3) return 0.0;
    double *r = dalloc((long)nz * nr);
    orc_mg_residual(psi, source, r_grid, nz, nr, dr, dz, r);
    long ni = (long)(nz - 2) * (nr - 2);
    double *in = dalloc(ni);
    for (int iz = 1; iz < nz - 1; ++iz)
        memcpy(in + (long)(iz - 1) * (nr - 2), r + IDX(iz, 1), sizeof(double) * (size_t)(nr - 2));
    double rms = orc_stable_rms(in, ni);
    free(in);
    free(r);
    return rms;
}

void orc_restrict_full_weight(const double *fine, int nz_f, int nr_f, double *coarse) {
    int nz_c = (nz_f + 1) / 2, nr_c = (nr_f + 1) / 2;
    int nr = nr_f; /* for IDX on the fine grid */
    memset(coarse, 0, sizeof(double) * (size_t)nz_c * nr_c);
    for (int ic = 1; ic < nz_c - 1; ++ic)
        for (int jc = 1; jc < nr_c - 1; ++jc) {
            int i = 2 * ic, j = 2 * jc;
            double v = (4.0 * fine[IDX(i, j)] +
                        2.0 * (fine[IDX(i - 1, j)] + fine[IDX(i + 1, j)] + fine[IDX(i, j - 1)] +
                               fine[IDX(i, j + 1)]) +
                        (fine[IDX(i - 1, j - 1)] + fine[IDX(i - 1, j + 1)] +
                         fine[IDX(i + 1, j - 1)] + fine[IDX(i + 1, j + 1)])) /
                       16.0;
            coarse[(long)ic * nr_c + jc] = v;
        }
    /* boundary injection, rows first then columns (core/multigrid_solve.py:94-97) */
    for (int jc = 0; jc < nr_c; ++jc) coarse[jc] = fine[IDX(0, 2 * jc)];
    for (int jc = 0; jc < nr_c; ++jc) coarse[(long)(nz_c - 1) * nr_c + jc] = fine[IDX(nz_f - 1, 2 * jc)];
    for (int ic = 0; ic < nz_c; ++ic) coarse[(long)ic * nr_c] = fine[IDX(2 * ic, 0)];
    for (int ic = 0; ic < nz_c; ++ic) coarse[(long)ic * nr_c + nr_c - 1] = fine[IDX(2 * ic, nr_f - 1)];
}

int orc_prolongate_bilinear(const double *coarse, int nz_c, int nr_c, int nz_f, int nr_f,
                            double *fine) {
    if (nz_c != (nz_f + 1) / 2 || nr_c != (nr_f + 1) / 2) return -1;
    int nr = nr_f;
#define C(i, j) coarse[(long)(i) * nr_c + (j)]
    memset(fine, 0, sizeof(double) * (size_t)nz_f * nr_f);
    int nz_used = nz_c, nr_used = nr_c; /* = min(n_c, (n_f+1)//2) */
    int h_end = 2 * (nr_c - 1) < nr_f - 1 ? 2 * (nr_c - 1) : nr_f - 1;
    int v_end = 2 * (nz_c - 1) < nz_f - 1 ? 2 * (nz_c - 1) : nz_f - 1;
    for (int i = 0; i < nz_used; ++i)
        for (int j = 0; j < nr_used; ++j) fine[IDX(2 * i, 2 * j)] = C(i, j);
    for (int i = 0; i < nz_used; ++i)
        for (int jf = 1; jf < h_end; jf += 2) {
            int j = (jf - 1) / 2;
            fine[IDX(2 * i, jf)] = 0.5 * (C(i, j) + C(i, j + 1));
        }
    for (int iff = 1; iff < v_end; iff += 2) {
        int i = (iff - 1) / 2;
        for (int j = 0; j < nr_used; ++j) fine[IDX(iff, 2 * j)] = 0.5 * (C(i, j) + C(i + 1, j));
    }
    for (int iff = 1; iff < v_end; iff += 2) {
        int i = (iff - 1) / 2;
        for (int jf = 1; jf < h_end; jf += 2) {
            int j = (jf - 1) / 2;
            fine[IDX(iff, jf)] = 0.25 * (C(i, j) + C(i + 1, j) + C(i, j + 1) + C(i + 1, j + 1));
        }
    }
#undef C
    return 0;
}

int orc_multigrid_vcycle(double *psi, const double *source, const double *r_grid, int nz, int nr,
                         double dr, double dz, double omega, int pre_smooth, int post_smooth,
                         int min_grid) {
    if (!omega_ok(omega)) return -1;
    if (min_grid >= nz || min_grid >= nr) /* base case, core/multigrid_solve.py:292-293 */
        return orc_mg_smooth(psi, source, r_grid, nz, nr, dr, dz, omega, 50);
    orc_mg_smooth(psi, source, r_grid, nz, nr, dr, dz, omega, pre_smooth);
    long n = (long)nz * nr;
    double *defect = dalloc(n);
    orc_mg_residual(psi, source, r_grid, nz, nr, dr, dz, defect);
    for (long i = 0; i < n; ++i) defect[i] = -defect[i];
    int nz_c = (nz + 1) / 2, nr_c = (nr + 1) / 2;
    long nc = (long)nz_c * nr_c;
    double *d_c = dalloc(nc), *r_c = dalloc(nc), *e_c = dalloc(nc);
    orc_restrict_full_weight(defect, nz, nr, d_c);
    orc_restrict_full_weight(r_grid, nz, nr, r_c);
    orc_multigrid_vcycle(e_c, d_c, r_c, nz_c, nr_c, dr * 2.0, dz * 2.0, omega, pre_smooth,
                         post_smooth, min_grid);
    double *corr = defect; /* reuse */
    orc_prolongate_bilinear(e_c, nz_c, nr_c, nz, nr, corr);
    for (long i = 0; i < n; ++i) psi[i] = psi[i] + corr[i];
    orc_mg_smooth(psi, source, r_grid, nz, nr, dr, dz, omega, post_smooth);
    free(defect);
    free(d_c);
    free(r_c);
    free(e_c);
    return 0;
}

static void meshgrid_r(const double *r_axis, int nz, int nr, double *rr) {
    for (int iz = 0; iz < nz; ++iz) memcpy(rr + (long)iz * nr, r_axis, sizeof(double) * (size_t)nr);
}

static void apply_bc(double *psi, const double *bc, int nz, int nr) {
    /* core/fusion_kernel_iterative_solver.py:316-329 */
    for (int ir = 0; ir < nr; ++ir) {
        psi[IDX(0, ir)] = bc[IDX(0, ir)];
        psi[IDX(nz - 1, ir)] = bc[IDX(nz - 1, ir)];
    }
    for (int iz = 0; iz < nz; ++iz) {
        psi[IDX(iz, 0)] = bc[IDX(iz, 0)];
        psi[IDX(iz, nr - 1)] = bc[IDX(iz, nr - 1)];
    }
}

int orc_multigrid_solve(const double *source, double *psi, double r_min, double r_max,
                        double z_min, double z_max, int nr, int nz, double tol, int max_cycles,
                        double omega, int pre_smooth, int post_smooth, int min_grid,
                        double *residual_out, int *n_cycles_out, int *converged_out) {
    if (!(isfinite(tol) && tol > 0.0)) return -1;
    if (max_cycles < 1) return -1;
    if (!omega_ok(omega)) return -1;
    long n = (long)nz * nr;
    double *r_axis = dalloc(nr), *z_axis = dalloc(nz), *rr = dalloc(n), *bc = dalloc(n);
    orc_linspace(r_min, r_max, nr, r_axis);
    orc_linspace(z_min, z_max, nz, z_axis);
    meshgrid_r(r_axis, nz, nr, rr);
    double dr = nr > 1 ? r_axis[1] - r_axis[0] : 1.0;
    double dz = nz > 1 ? z_axis[1] - z_axis[0] : 1.0;
    memcpy(bc, psi, sizeof(double) * (size_t)n);
    int n_cycles = 0;
    double residual = orc_residual_linf(psi, source, rr, nz, nr, dr, dz);
    int converged = residual < tol;
    while (!converged && n_cycles < max_cycles) {
        orc_multigrid_vcycle(psi, source, rr, nz, nr, dr, dz, omega, pre_smooth, post_smooth,
                             min_grid);
        apply_bc(psi, bc, nz, nr);
        n_cycles += 1;
        residual = orc_residual_linf(psi, source, rr, nz, nr, dr, dz);
        converged = residual < tol;
    }
    *residual_out = residual;
    *n_cycles_out = n_cycles;
    *converged_out = converged;
    free(r_axis);
    free(z_axis);
    free(rr);
    free(bc);
    return 0;
}

static double sanitize1(double x) { /* core/fusion_kernel_numerics.py:19-25 */
    if (isnan(x)) return 0.0;
    if (isinf(x)) return x > 0 ? CAP : -CAP;
    return fmin(fmax(x, -CAP), CAP);
}

void orc_jacobi_step(const double *psi_in, const double *source_in, const double *r_grid, int nz,
                     int nr, double dr, double dz, double *out) {
    long n = (long)nz * nr;
    double *psi = dalloc(n), *src = dalloc(n);
    for (long i = 0; i < n; ++i) { psi[i] = sanitize1(psi_in[i]); src[i] = sanitize1(source_in[i]); }
    memcpy(out, psi, sizeof(double) * (size_t)n);
    double dr2 = pow(dr, 2.0), dz2 = pow(dz, 2.0);
    double a_ns = 1.0 / dz2, a_c = 2.0 / dr2 + 2.0 / dz2;
    for (int iz = 1; iz < nz - 1; ++iz)
        for (int ir = 1; ir < nr - 1; ++ir) {
            double r_safe = fmax(r_grid[IDX(iz, ir)], 1e-10);
            double t = 1.0 / ((2.0 * r_safe) * dr);
            double a_e = 1.0 / dr2 - t, a_w = 1.0 / dr2 + t;
            double v = (a_e * psi[IDX(iz, ir + 1)] + a_w * psi[IDX(iz, ir - 1)] +
                        a_ns * psi[IDX(iz - 1, ir)] + a_ns * psi[IDX(iz + 1, ir)] -
                        src[IDX(iz, ir)]) /
                       a_c;
            out[IDX(iz, ir)] = fmin(fmax(v, -CAP), CAP);
        }
    free(psi);
    free(src);
}

int orc_sor_step(const double *psi_in, const double *source_in, const double *r_grid, int nz,
                 int nr, double dr, double dz, double omega, double *out) {
    if (!omega_ok(omega)) return -1;
    long n = (long)nz * nr;
    double *src = dalloc(n);
    for (long i = 0; i < n; ++i) { out[i] = sanitize1(psi_in[i]); src[i] = sanitize1(source_in[i]); }
    if (nz >= 3 && nr >= 3) {
        double dr2 = pow(dr, 2.0), dz2 = pow(dz, 2.0);
        long ni = (long)(nz - 2) * (nr - 2);
        double *a_e = dalloc(ni), *a_w = dalloc(ni);
        coeffs(r_grid, nz, nr, dr, dr2, a_e, a_w);
        rb_sweep(out, src, a_e, a_w, 1.0 / dz2, 2.0 / dr2 + 2.0 / dz2, nz, nr, omega, 1);
        free(a_e);
        free(a_w);
    }
    free(src);
    return 0;
}

/* K(m), E(m) by the arithmetic-geometric mean (A&S 17.6.3-17.6.4) */
void orc_ellip_ke(double m, double *K, double *E) {
    double a = 1.0, b = sqrt(1.0 - m);
    double s = 0.5 * m, pw = 0.5;
    for (int it = 0; it < 40; ++it) {
        double c = 0.5 * (a - b);
        double an = 0.5 * (a + b);
        b = sqrt(a * b);
        a = an;
        pw *= 2.0;
        s += pw * c * c;
        if (fabs(c) <= 1e-17 * a) break;
    }
    double k = M_PI / (2.0 * a);
    *K = k;
    *E = k * (1.0 - s);
}

void orc_vacuum_field(const double *r_axis, const double *z_axis, int nz, int nr, int n_coils,
                      const double *coil_r, const double *coil_z, const double *coil_i, double mu0,
                      double *psi_vac) {
    memset(psi_vac, 0, sizeof(double) * (size_t)nz * nr);
    for (int c = 0; c < n_coils; ++c) {
        double rc = coil_r[c], zc = coil_z[c], cur = coil_i[c];
        double prefactor = (mu0 * cur) / (2.0 * M_PI);
        for (int iz = 0; iz < nz; ++iz)
            for (int ir = 0; ir < nr; ++ir) {
                double R = r_axis[ir], dZ = z_axis[iz] - zc;
                double rp = (R + rc) * (R + rc);
                double k2 = ((4.0 * R) * rc) / (rp + dZ * dZ);
                k2 = fmin(fmax(k2, 1e-12), 1.0 - 1e-12);
                double K, E;
                orc_ellip_ke(k2, &K, &E);
                double sqrt_term = sqrt(R * rc);
                double k = sqrt(k2);
                double term = ((2.0 - k2) * K - 2.0 * E) / k;
                psi_vac[IDX(iz, ir)] += (prefactor * sqrt_term) * term;
            }
    }
}

#define MU0_SI (4e-7 * M_PI)

static double green_vec(double r_src, double z_src, double r_obs, double z_obs) {
    /* core/fusion_kernel_free_boundary.py:58-80 */
    double dr = r_obs - r_src, dz = z_obs - z_src;
    int self_mask = (dr * dr + dz * dz) < 1e-24;
    double denom = (r_obs + r_src) * (r_obs + r_src) + dz * dz;
    double k2 = denom > 1e-30 ? ((4.0 * r_obs) * r_src) / fmax(denom, 1e-30) : 0.0;
    k2 = fmin(fmax(k2, 1e-12), 1.0 - 1e-12);
    double k = sqrt(k2), K, E;
    orc_ellip_ke(k2, &K, &E);
    double prefactor = (MU0_SI / (2.0 * M_PI)) * sqrt(r_obs * r_src);
    double flux = (prefactor * ((2.0 - k2) * K - 2.0 * E)) / k;
    return self_mask ? 0.0 : flux;
}

double orc_green_function(double r_src, double z_src, double r_obs, double z_obs) {
    if (!isfinite(r_src) || !isfinite(z_src) || !isfinite(r_obs) || !isfinite(z_obs)) return NAN;
    if (r_src <= 0.0 || r_obs <= 0.0) return NAN;
    return green_vec(r_src, z_src, r_obs, z_obs);
}

void orc_external_flux(const double *r_axis, const double *z_axis, int nz, int nr, int n_coils,
                       const double *coil_r, const double *coil_z, const double *coil_i,
                       const double *coil_turns, double *psi_ext) {
    memset(psi_ext, 0, sizeof(double) * (size_t)nz * nr);
    for (int c = 0; c < n_coils; ++c) {
        double i_eff = coil_i[c] * (coil_turns ? coil_turns[c] : 1.0);
        for (int iz = 0; iz < nz; ++iz)
            for (int ir = 0; ir < nr; ++ir)
                psi_ext[IDX(iz, ir)] += i_eff * green_vec(coil_r[c], coil_z[c], r_axis[ir], z_axis[iz]);
    }
}

void orc_find_axis(const double *psi, int nz, int nr, int *iz, int *ir, double *psi_axis) {
    long n = (long)nz * nr, best = 0;
    /* np.argmax: first occurrence; a NaN wins */
    for (long i = 1; i < n; ++i) {
        if (isnan(psi[best])) break;
        if (isnan(psi[i]) || psi[i] > psi[best]) best = i;
    }
    *iz = (int)(best / nr);
    *ir = (int)(best % nr);
    double v = psi[best];
    if (fabs(v) < 1e-6) v = 1e-6;
    *psi_axis = v;
}

static double nan_to_num300(double x) {
    if (isnan(x)) return 0.0;
    if (isinf(x)) return x > 0 ? 1e300 : -1e300;
    return x;
}

void orc_find_x_point(const double *psi, const double *z_axis, int nz, int nr, double dr, double dz,
                      double z_min_cfg, long *idx_out, double *psi_x) {
    long n = (long)nz * nr;
    int any_finite = 0;
    for (long i = 0; i < n; ++i) if (isfinite(psi[i])) { any_finite = 1; break; }
    if (!any_finite) { *idx_out = -2; *psi_x = 0.0; return; }
    double zthr = z_min_cfg * 0.5;
    long best = -1;
    double bestv = INFINITY;
    for (int iz = 0; iz < nz; ++iz) {
        if (!(zthr > z_axis[iz])) continue;
        for (int ir = 0; ir < nr; ++ir) {
#define P(a, b) nan_to_num300(psi[IDX(a, b)])
            double gz, gr; /* np.gradient(psi, dZ, dR), edge_order=1 */
            if (iz == 0) gz = (P(1, ir) - P(0, ir)) / dz;
            else if (iz == nz - 1) gz = (P(nz - 1, ir) - P(nz - 2, ir)) / dz;
            else gz = (P(iz + 1, ir) - P(iz - 1, ir)) / (2.0 * dz);
            if (ir == 0) gr = (P(iz, 1) - P(iz, 0)) / dr;
            else if (ir == nr - 1) gr = (P(iz, nr - 1) - P(iz, nr - 2)) / dr;
            else gr = (P(iz, ir + 1) - P(iz, ir - 1)) / (2.0 * dr);
#undef P
            double b = hypot(gr, gz);
            if (isfinite(b) && (best < 0 || b < bestv)) { best = IDX(iz, ir); bestv = b; }
            else if (isnan(b) && best >= 0 && !isnan(bestv)) { /* np.argmin: first NaN wins */
                best = IDX(iz, ir); bestv = b;
            }
        }
    }
    if (best >= 0) {
        double v = psi[best];
        if (!isfinite(v)) v = nan_to_num300(v);
        *idx_out = best;
        *psi_x = v;
        return;
    }
    double m = INFINITY; /* fallback: min over finite psi, core/fusion_kernel.py:339-340 */
    for (long i = 0; i < n; ++i) if (isfinite(psi[i]) && psi[i] < m) m = psi[i];
    *idx_out = -1;
    *psi_x = m;
}

/* masked |grad psi| at one point (the quantity find_x_point minimises), +inf outside the divertor mask / non-finite */
static double masked_b_at(const double *psi, const double *z_axis, int nz, int nr, double dr, double dz, double zthr, int iz, int ir) {
    if (!(zthr > z_axis[iz])) return INFINITY;
#define P(a, b) nan_to_num300(psi[IDX(a, b)])
    double gz, gr;
    if (iz == 0) gz = (P(1, ir) - P(0, ir)) / dz;
    else if (iz == nz - 1) gz = (P(nz - 1, ir) - P(nz - 2, ir)) / dz;
    else gz = (P(iz + 1, ir) - P(iz - 1, ir)) / (2.0 * dz);
    if (ir == 0) gr = (P(iz, 1) - P(iz, 0)) / dr;
    else if (ir == nr - 1) gr = (P(iz, nr - 1) - P(iz, nr - 2)) / dr;
    else gr = (P(iz, ir + 1) - P(iz, ir - 1)) / (2.0 * dr);
#undef P
    double b = hypot(gr, gz);
    return isfinite(b) ? b : INFINITY;
}

/* core/fusion_kernel.py:284-329: solver.xpoint_use_saddle_detection -- among the (up to) 16 smallest masked |grad psi| points
 * prefer the smallest one that is an interior saddle (det Hessian < 0); otherwise the plain minimum.  Candidates are taken in
 * ascending (|grad psi|, index) order, so the first saddle met is the reference's min over saddle_hits. */
void orc_find_x_point_saddle(const double *psi, const double *z_axis, int nz, int nr, double dr, double dz,
                             double z_min_cfg, long *idx_out, double *psi_x) {
    long plain_idx;
    double plain_psi;
    orc_find_x_point(psi, z_axis, nz, nr, dr, dz, z_min_cfg, &plain_idx, &plain_psi);
    *idx_out = plain_idx;
    *psi_x = plain_psi;
    if (plain_idx < 0) return;
    double zthr = z_min_cfg * 0.5;
    long n = (long)nz * nr;
    double last_b = -1.0;
    long last_i = -1;
    for (int k = 0; k < 16; ++k) {
        long best = -1;
        double bestv = INFINITY;
        for (long q = 0; q < n; ++q) {
            int iz = (int)(q / nr), ir = (int)(q % nr);
            double b = masked_b_at(psi, z_axis, nz, nr, dr, dz, zthr, iz, ir);
            if (!isfinite(b)) continue;
            if (b < last_b || (b == last_b && q <= last_i)) continue; /* already taken */
            if (best < 0 || b < bestv) { best = q; bestv = b; }
        }
        if (best < 0) break; /* fewer than 16 finite candidates */
        last_b = bestv;
        last_i = best;
        int iz = (int)(best / nr), ir = (int)(best % nr);
        if (iz <= 0 || iz >= nz - 1 || ir <= 0 || ir >= nr - 1) continue;
#define P(a, b) nan_to_num300(psi[IDX(a, b)])
        double d2r = (P(iz, ir + 1) - 2.0 * P(iz, ir) + P(iz, ir - 1)) / (dr * dr);
        double d2z = (P(iz + 1, ir) - 2.0 * P(iz, ir) + P(iz - 1, ir)) / (dz * dz);
        double drz = (P(iz + 1, ir + 1) - P(iz + 1, ir - 1) - P(iz - 1, ir + 1) + P(iz - 1, ir - 1)) / (4.0 * dr * dz);
#undef P
        double det = d2r * d2z - drz * drz;
        if (isfinite(det) && det < 0.0) {
            double v = psi[best];
            if (!isfinite(v)) v = nan_to_num300(v);
            *idx_out = best;
            *psi_x = v;
            return;
        }
    }
}

static double mtanh1(double x, const double *p) { /* core/fusion_kernel.py:359-390 */
    double y = (p[0] - x) / p[1];
    y = fmin(fmax(y, -20.0), 20.0);
    double pedestal = (0.5 * p[2]) * (1.0 + tanh(y));
    double core = 0.0;
    if (x < p[0]) {
        double q = x / p[0];
        core = fmax(0.0, 1.0 - q * q);
    }
    return pedestal + p[3] * core;
}

double orc_update_source(const double *psi, const double *r_axis, int nz, int nr, double dr,
                         double dz, double psi_axis, double psi_bnd, double mu0, double i_target,
                         int hmode, const double *ped_p, const double *ped_ff, double *j_phi) {
    long n = (long)nz * nr;
    double denom = psi_bnd - psi_axis;
    if (fabs(denom) < 1e-9) denom = 1e-9;
    const double beta_mix = 0.5;
    for (int iz = 0; iz < nz; ++iz)
        for (int ir = 0; ir < nr; ++ir) {
            double R = r_axis[ir];
            double pn = (psi[IDX(iz, ir)] - psi_axis) / denom;
            int mask = (pn >= 0) && (pn < 1.0);
            double p = 0.0, ff = 0.0;
            if (mask) {
                if (hmode) { p = mtanh1(pn, ped_p); ff = mtanh1(pn, ped_ff); }
                else { p = 1.0 - pn; ff = p; }
            }
            double j_p = R * p;
            double j_f = (1.0 / (mu0 * R)) * ff;
            j_phi[IDX(iz, ir)] = beta_mix * j_p + (1 - beta_mix) * j_f;
        }
    double i_current = (orc_np_sum(j_phi, n) * dr) * dz;
    if (fabs(i_current) > 1e-9) {
        double s = i_target / i_current;
        for (long i = 0; i < n; ++i) j_phi[i] = j_phi[i] * s;
    } else {
        memset(j_phi, 0, sizeof(double) * (size_t)n);
    }
    return i_current;
}

/* core/fusion_kernel_iterative_solver.py:248-314 _anderson_step.  hist arrays hold `len` fields (oldest first).
 * Gram products are plain left-to-right sums (NumPy delegates dF.T @ dF to BLAS; order differs, tolerance parity). */
static void anderson_step(double **psi_hist, double **res_hist, int len, int m, long n, double *mixed) {
    int mk = m < len ? m : len;
    double **P = psi_hist + (len - mk), **F = res_hist + (len - mk);
    if (mk < 2) { memcpy(mixed, psi_hist[len - 1], sizeof(double) * (size_t)n); return; }
    int q = mk - 1;
    double G[16][16], v[16], gamma[16], alpha[16];
    for (int a = 0; a < q; ++a) {
        for (int b2 = 0; b2 < q; ++b2) {
            double acc = 0.0;
            for (long i = 0; i < n; ++i) acc += (F[a + 1][i] - F[a][i]) * (F[b2 + 1][i] - F[b2][i]);
            G[a][b2] = acc + (a == b2 ? 1e-10 : 0.0);
        }
        double acc = 0.0;
        for (long i = 0; i < n; ++i) acc += (F[a + 1][i] - F[a][i]) * F[mk - 1][i];
        v[a] = acc;
    }
    /* np.linalg.solve: LU with partial pivoting */
    int singular = 0;
    for (int c = 0; c < q; ++c) {
        int piv = c;
        for (int r = c + 1; r < q; ++r) if (fabs(G[r][c]) > fabs(G[piv][c])) piv = r;
        if (G[piv][c] == 0.0) { singular = 1; break; }
        if (piv != c) { for (int k = 0; k < q; ++k) { double t = G[c][k]; G[c][k] = G[piv][k]; G[piv][k] = t; } double t = v[c]; v[c] = v[piv]; v[piv] = t; }
        for (int r = c + 1; r < q; ++r) {
            double f = G[r][c] / G[c][c];
            for (int k = c; k < q; ++k) G[r][k] -= f * G[c][k];
            v[r] -= f * v[c];
        }
    }
    if (singular) { memcpy(mixed, psi_hist[len - 1], sizeof(double) * (size_t)n); return; }
    for (int r = q - 1; r >= 0; --r) {
        double acc = v[r];
        for (int k = r + 1; k < q; ++k) acc -= G[r][k] * gamma[k];
        gamma[r] = acc / G[r][r];
    }
    double gsum = 0.0;
    for (int a = 0; a < q; ++a) gsum += gamma[a];
    for (int a = 0; a < mk; ++a) alpha[a] = 0.0;
    alpha[mk - 1] = 1.0 - gsum;
    for (int a = 0; a < q; ++a) alpha[a] -= gamma[a];
    double asum = 0.0;
    for (int a = 0; a < mk; ++a) asum += alpha[a];
    if (fabs(asum) < 1e-12) { memcpy(mixed, psi_hist[len - 1], sizeof(double) * (size_t)n); return; }
    for (int a = 0; a < mk; ++a) alpha[a] /= asum;
    for (long i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int a = 0; a < mk; ++a) acc += alpha[a] * P[a][i];
        mixed[i] = acc;
    }
}

int orc_picard_solve(const orc_picard_params *p, double *psi, const double *boundary_flux,
                     double *j_phi, orc_picard_result *res, double *trace, int max_iter_budget) {
    int nz = p->nz, nr = p->nr;
    long n = (long)nz * nr;
    memset(res, 0, sizeof(*res));
    res->x_idx = -1;
    double *r_axis = dalloc(nr), *z_axis = dalloc(nz);
    orc_linspace(p->r_min, p->r_max, nr, r_axis);
    orc_linspace(p->z_min, p->z_max, nz, z_axis);
    double dr = r_axis[1] - r_axis[0], dz = z_axis[1] - z_axis[0];
    double *rr = dalloc(n);
    meshgrid_r(r_axis, nz, nr, rr);
    int rc = 0;

    /* zero-current short-circuit, core/fusion_kernel_newton_solver.py:448-465 */
    if (fabs(p->i_target) < 1e-12 && !p->preserve_initial_state) {
        orc_vacuum_field(r_axis, z_axis, nz, nr, p->n_coils, p->coil_r, p->coil_z, p->coil_i, p->mu0, psi);
        memset(j_phi, 0, sizeof(double) * (size_t)n);
        res->converged = 1;
        res->iterations = 0;
        free(r_axis); free(z_axis); free(rr);
        return 0;
    }

    /* _prepare_initial_flux, core/fusion_kernel_iterative_solver.py:412-451 */
    double *bc = dalloc(n);
    if (p->has_boundary_flux && boundary_flux) memcpy(bc, boundary_flux, sizeof(double) * (size_t)n);
    else if (p->preserve_initial_state) memcpy(bc, psi, sizeof(double) * (size_t)n);
    else orc_vacuum_field(r_axis, z_axis, nz, nr, p->n_coils, p->coil_r, p->coil_z, p->coil_i, p->mu0, bc);
    if (p->preserve_initial_state) apply_bc(psi, bc, nz, nr);
    else memcpy(psi, bc, sizeof(double) * (size_t)n);

    double *psi_best = dalloc(n), *source = dalloc(n), *psi_new = dalloc(n), *tmp = dalloc(n);
    memcpy(psi_best, psi, sizeof(double) * (size_t)n);
    double diff_best = 1e9, gs_best = INFINITY, gs_last = INFINITY;
    int converged = 0, final_iter = 0, ran = 0;

    /* _seed_plasma, core/fusion_kernel_iterative_solver.py:384-410 */
    if (fabs(p->i_target) < 1e-12) {
        memset(j_phi, 0, sizeof(double) * (size_t)n);
    } else {
        double r_center = (p->r_min + p->r_max) / 2.0;
        for (int iz = 0; iz < nz; ++iz)
            for (int ir = 0; ir < nr; ++ir) {
                double a = r_axis[ir] - r_center;
                double dist_sq = a * a + z_axis[iz] * z_axis[iz];
                j_phi[IDX(iz, ir)] = exp(-dist_sq / 2.0);
            }
        double i_seed = (orc_np_sum(j_phi, n) * dr) * dz;
        if (i_seed > 0) {
            double s = p->i_target / i_seed;
            for (long i = 0; i < n; ++i) j_phi[i] *= s;
        }
        for (int iz = 0; iz < nz; ++iz)
            for (int ir = 0; ir < nr; ++ir) source[IDX(iz, ir)] = ((-p->mu0) * r_axis[ir]) * j_phi[IDX(iz, ir)];
        for (int s = 0; s < 50; ++s) {
            orc_jacobi_step(psi, source, rr, nz, nr, dr, dz, tmp);
            memcpy(psi, tmp, sizeof(double) * (size_t)n);
        }
    }

    int max_iter = p->max_iter;
    if (max_iter_budget >= 0 && max_iter_budget < max_iter) max_iter = max_iter_budget;
    double *and_psi[64], *and_res[64];
    int and_len = 0;
    for (int k = 0; k < max_iter; ++k) {
        final_iter = k;
        ran = 1;
        int aiz, air;
        double psi_axis, psi_bnd;
        long xidx;
        orc_find_axis(psi, nz, nr, &aiz, &air, &psi_axis);
        if (p->xpoint_saddle) orc_find_x_point_saddle(psi, z_axis, nz, nr, dr, dz, p->z_min, &xidx, &psi_bnd);
        else orc_find_x_point(psi, z_axis, nz, nr, dr, dz, p->z_min, &xidx, &psi_bnd);
        if (fabs(psi_axis - psi_bnd) < 0.1) psi_bnd = psi_axis * 0.1;
        double i_current = NAN;
        if (!p->external_profile_mode)
            i_current = orc_update_source(psi, r_axis, nz, nr, dr, dz, psi_axis, psi_bnd, p->mu0,
                                          p->i_target, p->hmode, p->ped_p, p->ped_ff, j_phi);
        for (int iz = 0; iz < nz; ++iz)
            for (int ir = 0; ir < nr; ++ir) source[IDX(iz, ir)] = ((-p->mu0) * r_axis[ir]) * j_phi[IDX(iz, ir)];
        /* _elliptic_solve, core/fusion_kernel_iterative_solver.py:361-380 */
        if (p->method == 2) {
            orc_jacobi_step(psi, source, rr, nz, nr, dr, dz, psi_new);
        } else if (p->method == 0) {
            memcpy(psi_new, psi, sizeof(double) * (size_t)n);
            rc = orc_multigrid_vcycle(psi_new, source, rr, nz, nr, dr, dz, p->sor_omega, 3, 3, 5);
        } else {
            rc = orc_sor_step(psi, source, rr, nz, nr, dr, dz, p->sor_omega, psi_new);
        }
        if (rc) break;
        apply_bc(psi_new, bc, nz, nr);
        int bad = 0;
        for (long i = 0; i < n; ++i) if (!isfinite(psi_new[i])) { bad = 1; break; }
        res->psi_axis = psi_axis;
        res->psi_bnd = psi_bnd;
        res->x_idx = xidx;
        if (bad) {
            memcpy(psi, psi_best, sizeof(double) * (size_t)n);
            res->diverged = 1;
            break;
        }
        for (long i = 0; i < n; ++i) tmp[i] = fabs(psi_new[i] - psi[i]);
        double diff = orc_np_sum(tmp, n) / (double)n;
        for (long i = 0; i < n; ++i) psi[i] = (1.0 - p->alpha) * psi[i] + p->alpha * psi_new[i];
        if (p->method == 3) { /* Anderson acceleration, core/fusion_kernel_newton_solver.py:540-550 */
            double *ph = dalloc(n), *rh = dalloc(n);
            memcpy(ph, psi, sizeof(double) * (size_t)n);
            for (long i = 0; i < n; ++i) rh[i] = psi_new[i] - psi[i];
            and_psi[and_len] = ph;
            and_res[and_len] = rh;
            and_len += 1;
            if (and_len >= 3 && k % 3 == 0) {
                anderson_step(and_psi, and_res, and_len, p->anderson_m, n, tmp);
                apply_bc(tmp, bc, nz, nr);
                memcpy(psi, tmp, sizeof(double) * (size_t)n);
            }
            if (and_len > p->anderson_m + 2) {
                free(and_psi[0]);
                free(and_res[0]);
                for (int j = 1; j < and_len; ++j) { and_psi[j - 1] = and_psi[j]; and_res[j - 1] = and_res[j]; }
                and_len -= 1;
            }
        }
        double gs = orc_gs_residual_rms(psi, source, rr, nz, nr, dr, dz);
        gs_last = gs;
        if (gs < gs_best) gs_best = gs;
        if (diff < diff_best) {
            diff_best = diff;
            memcpy(psi_best, psi, sizeof(double) * (size_t)n);
        }
        if (trace) {
            double *t = trace + 6L * k;
            t[0] = psi_axis; t[1] = psi_bnd; t[2] = (double)xidx; t[3] = i_current; t[4] = diff; t[5] = gs;
        }
        int upd = diff < p->tol;
        int gsc = (!p->require_gs_residual) || (gs < p->gs_tol);
        if (upd && gsc) { converged = 1; break; }
    }
    for (int j = 0; j < and_len; ++j) { free(and_psi[j]); free(and_res[j]); }
    res->converged = converged;
    res->iterations = ran ? final_iter + 1 : 1;
    res->residual = diff_best;
    res->gs_residual = gs_last;
    res->gs_residual_best = gs_best;
    free(r_axis); free(z_axis); free(rr); free(bc);
    free(psi_best); free(source); free(psi_new); free(tmp);
    return rc;
}

int orc_free_boundary_solve(const orc_picard_params *p, double *psi, double *j_phi, int n_fb_coils,
                            const double *fb_r, const double *fb_z, const double *fb_i,
                            const double *fb_turns, int max_outer, double tol, double *final_diff,
                            int *picard_iters, orc_picard_result *last) {
    int nz = p->nz, nr = p->nr;
    long n = (long)nz * nr;
    double *r_axis = dalloc(nr), *z_axis = dalloc(nz), *psi_ext = dalloc(n), *psi_old = dalloc(n);
    orc_linspace(p->r_min, p->r_max, nr, r_axis);
    orc_linspace(p->z_min, p->z_max, nz, z_axis);
    orc_external_flux(r_axis, z_axis, nz, nr, n_fb_coils, fb_r, fb_z, fb_i, fb_turns, psi_ext);
    orc_picard_params q = *p;
    q.preserve_initial_state = 1;
    q.has_boundary_flux = 1;
    double diff = INFINITY;
    int outer = 0;
    for (outer = 0; outer < max_outer; ++outer) {
        apply_bc(psi, psi_ext, nz, nr);
        memcpy(psi_old, psi, sizeof(double) * (size_t)n);
        orc_picard_result r;
        orc_picard_solve(&q, psi, psi_ext, j_phi, &r, NULL, -1);
        if (picard_iters) picard_iters[outer] = r.iterations;
        if (last) *last = r;
        diff = 0.0;
        for (long i = 0; i < n; ++i) {
            double d = fabs(psi[i] - psi_old[i]);
            if (isnan(d) || d > diff) diff = d;
        }
        if (diff < tol) { outer += 1; break; }
    }
    if (outer > max_outer) outer = max_outer;
    *final_diff = diff;
    free(r_axis); free(z_axis); free(psi_ext); free(psi_old);
    return outer;
}

/* _compute_source_np, core/jax_gs_solver.py:89-123 */
static void np_lane_source(const double *psi, const double *r_axis, int nz, int nr, double dr, double dz,
                           double mu0, double ip_target, double beta_mix, double *jraw, double *source) {
    long n = (long)nz * nr;
    double psi_axis = -INFINITY;
    for (int iz = 1; iz < nz - 1; ++iz)
        for (int ir = 1; ir < nr - 1; ++ir)
            if (psi[IDX(iz, ir)] > psi_axis) psi_axis = psi[IDX(iz, ir)];
    double denom = 0.0 - psi_axis;
    if (fabs(denom) < 1e-9) denom = denom != 0 ? (denom > 0 ? 1e-9 : -1e-9) : 1e-9;
    for (int iz = 0; iz < nz; ++iz)
        for (int ir = 0; ir < nr; ++ir) {
            double R = r_axis[ir];
            double x = (psi[IDX(iz, ir)] - psi_axis) / denom;
            x = fmin(fmax(x, 0.0), 1.0);
            double prof = (x >= 0.0 && x < 1.0) ? 1.0 - x : 0.0;
            double r_safe = fmax(R, 1e-10);
            double j_p = R * prof;
            double j_f = prof / (mu0 * r_safe);
            jraw[IDX(iz, ir)] = beta_mix * j_p + (1.0 - beta_mix) * j_f;
        }
    double i_cur = (orc_np_sum(jraw, n) * dr) * dz;
    double scale = ip_target / fmax(fabs(i_cur), 1e-9);
    for (int iz = 0; iz < nz; ++iz)
        for (int ir = 0; ir < nr; ++ir)
            source[IDX(iz, ir)] = ((-mu0) * r_axis[ir]) * (jraw[IDX(iz, ir)] * scale);
}

/* gs_delta_star_np, core/jax_gs_solver.py:196-217: centred operator with this lane's association, zero ring */
void orc_np_delta_star(const double *psi, int nz, int nr, double r_min, double r_max, double z_min,
                       double z_max, double *out) {
    double *r_axis = dalloc(nr);
    orc_linspace(r_min, r_max, nr, r_axis);
    double dr = (r_max - r_min) / (nr - 1), dz = (z_max - z_min) / (nz - 1);
    for (long i = 0; i < (long)nz * nr; ++i) out[i] = 0.0;
    for (int iz = 1; iz < nz - 1; ++iz)
        for (int ir = 1; ir < nr - 1; ++ir) {
            double c = psi[IDX(iz, ir)];
            double d2r = (psi[IDX(iz, ir + 1)] - 2.0 * c + psi[IDX(iz, ir - 1)]) / (dr * dr);
            double d1r = (psi[IDX(iz, ir + 1)] - psi[IDX(iz, ir - 1)]) / ((2.0 * dr) * r_axis[ir]);
            double d2z = (psi[IDX(iz + 1, ir)] - 2.0 * c + psi[IDX(iz - 1, ir)]) / (dz * dz);
            out[IDX(iz, ir)] = d2r - d1r + d2z;
        }
    free(r_axis);
}

/* gs_equation_residual_np, core/jax_gs_solver.py:251-316; out3 = abs_max, relative_max, relative_l2.
 * The L2 norms are plain left-to-right sums here (the reference's go through BLAS ddot, whose order is unspecified). */
void orc_np_equation_residual(const double *psi, int nz, int nr, double r_min, double r_max, double z_min,
                              double z_max, double ip_target, double mu0, double beta_mix, double *out3) {
    long n = (long)nz * nr;
    double *r_axis = dalloc(nr), *z_axis = dalloc(nz), *jraw = dalloc(n), *source = dalloc(n);
    orc_linspace(r_min, r_max, nr, r_axis);
    orc_linspace(z_min, z_max, nz, z_axis);
    double dr = r_axis[1] - r_axis[0], dz = z_axis[1] - z_axis[0];
    np_lane_source(psi, r_axis, nz, nr, dr, dz, mu0, ip_target, beta_mix, jraw, source);
    double dr2 = dr * dr, dz2 = dz * dz;
    double a_ns = 1.0 / dz2, a_c = 2.0 / dr2 + 2.0 / dz2;
    double rm = 0.0, r2 = 0.0, sm = 0.0, s2 = 0.0;
    for (int iz = 1; iz < nz - 1; ++iz)
        for (int ir = 1; ir < nr - 1; ++ir) {
            double r_safe = fmax(r_axis[ir], 1.0e-10);
            double t = 1.0 / ((2.0 * r_safe) * dr);
            double a_e = 1.0 / dr2 - t, a_w = 1.0 / dr2 + t;
            double sv = source[IDX(iz, ir)];
            double res = a_e * psi[IDX(iz, ir + 1)] + a_w * psi[IDX(iz, ir - 1)] +
                         a_ns * (psi[IDX(iz - 1, ir)] + psi[IDX(iz + 1, ir)]) - a_c * psi[IDX(iz, ir)] - sv;
            if (fabs(res) > rm) rm = fabs(res);
            if (fabs(sv) > sm) sm = fabs(sv);
            r2 += res * res;
            s2 += sv * sv;
        }
    out3[0] = rm;
    out3[1] = rm / (sm + 1.0e-30);
    out3[2] = sqrt(r2) / (sqrt(s2) + 1.0e-30);
    free(r_axis); free(z_axis); free(jraw); free(source);
}

void orc_gs_solve_np(double r_min, double r_max, double z_min, double z_max, int nr, int nz,
                     double ip_target, double mu0, int n_picard, int n_jacobi, double alpha,
                     double omega_j, double beta_mix, double *psi) {
    long n = (long)nz * nr;
    double *r_axis = dalloc(nr), *z_axis = dalloc(nz);
    orc_linspace(r_min, r_max, nr, r_axis);
    orc_linspace(z_min, z_max, nz, z_axis);
    double dr = r_axis[1] - r_axis[0], dz = z_axis[1] - z_axis[0];
    double r_center = 0.5 * (r_min + r_max);
    for (int iz = 0; iz < nz; ++iz)
        for (int ir = 0; ir < nr; ++ir) {
            double a = r_axis[ir] - r_center;
            psi[IDX(iz, ir)] = exp(-(a * a) / 0.5) * 0.01;
        }
    for (int ir = 0; ir < nr; ++ir) { psi[IDX(0, ir)] = 0.0; psi[IDX(nz - 1, ir)] = 0.0; }
    for (int iz = 0; iz < nz; ++iz) { psi[IDX(iz, 0)] = 0.0; psi[IDX(iz, nr - 1)] = 0.0; }
    double *source = dalloc(n), *pe = dalloc(n), *pn = dalloc(n), *jraw = dalloc(n);
    double dr2 = dr * dr, dz2 = dz * dz;
    double a_ns = 1.0 / dz2, a_c = 2.0 / dr2 + 2.0 / dz2;
    for (int it = 0; it < n_picard; ++it) {
        np_lane_source(psi, r_axis, nz, nr, dr, dz, mu0, ip_target, beta_mix, jraw, source);
        memcpy(pe, psi, sizeof(double) * (size_t)n);
        for (int jt = 0; jt < n_jacobi; ++jt) {
            /* _jacobi_gs_step_np, core/jax_gs_solver.py:58-86 */
            memcpy(pn, pe, sizeof(double) * (size_t)n);
            for (int iz = 1; iz < nz - 1; ++iz)
                for (int ir = 1; ir < nr - 1; ++ir) {
                    double r_safe = fmax(r_axis[ir], 1e-10);
                    double t = 1.0 / ((2.0 * r_safe) * dr);
                    double a_e = 1.0 / dr2 - t, a_w = 1.0 / dr2 + t;
                    double upd = (a_e * pe[IDX(iz, ir + 1)] + a_w * pe[IDX(iz, ir - 1)] +
                                  a_ns * (pe[IDX(iz - 1, ir)] + pe[IDX(iz + 1, ir)]) -
                                  source[IDX(iz, ir)]) /
                                 a_c;
                    pn[IDX(iz, ir)] = (1.0 - omega_j) * pe[IDX(iz, ir)] + omega_j * upd;
                }
            double *sw = pe; pe = pn; pn = sw;
        }
        for (long i = 0; i < n; ++i) psi[i] = (1.0 - alpha) * psi[i] + alpha * pe[i];
    }
    free(r_axis); free(z_axis); free(source); free(pe); free(pn); free(jraw);
}

double orc_fastsolver_sweeps(double *psi, const double *j, int nr, int nz, double r_min,
                             double r_max, double z_min, double z_max, double boundary_value,
                             double omega, int iterations, double tol, int *performed) {
    /* hpc/solver.cpp:63-199: dr from the extent (not linspace), R = r_min + r*dr, mu = 1.0 */
    double dr = (r_max - r_min) / (nr - 1), dz = (z_max - z_min) / (nz - 1);
    double dr_sq = dr * dr, dz_sq = dz * dz;
    double last = 0.0;
    int done = 0;
    for (int it = 0; it < iterations; ++it) {
        for (int r = 0; r < nr; ++r) { psi[r] = boundary_value; psi[(long)(nz - 1) * nr + r] = boundary_value; }
        for (int z = 0; z < nz; ++z) { psi[(long)z * nr] = boundary_value; psi[(long)z * nr + nr - 1] = boundary_value; }
        double mx[2] = {0.0, 0.0};
        for (int parity = 0; parity < 2; ++parity)
            for (int z = 1; z < nz - 1; ++z)
                for (int r = 1; r < nr - 1; ++r) {
                    if (((z + r) % 2) != parity) continue;
                    long idx = (long)z * nr + r;
                    double R = r_min + r * dr;
                    double source = -1.0 * R * j[idx];
                    double c_rp = 1.0 / dr_sq - 1.0 / (2.0 * R * dr);
                    double c_rm = 1.0 / dr_sq + 1.0 / (2.0 * R * dr);
                    double c_z = 1.0 / dz_sq;
                    double center = 2.0 / dr_sq + 2.0 / dz_sq;
                    double p_gs = (source + c_z * (psi[idx + nr] + psi[idx - nr]) +
                                   c_rp * psi[idx + 1] + c_rm * psi[idx - 1]) /
                                  center;
                    double old = psi[idx];
                    double nw = (1.0 - omega) * old + omega * p_gs;
                    psi[idx] = nw;
                    double d = fabs(nw - old);
                    if (d > mx[parity]) mx[parity] = d;
                }
        last = mx[0] > mx[1] ? mx[0] : mx[1];
        done = it + 1;
        if (tol >= 0.0 && last <= tol) break;
    }
    if (performed) *performed = done;
    return last;
}

/* =====================================================================================
 * Newton-Kantorovich + GMRES  
 * ===================================================================================== */

void orc_dlartg(double f, double g, double *c, double *s, double *r) {
    /* LAPACK 3.11 SRC/la_lartg.f90 (real double) */
    const double safmin = 2.2250738585072014e-308, safmax = 1.0 / safmin;
    const double rtmin = sqrt(safmin), rtmax = sqrt(safmax / 2.0);
    double f1 = fabs(f), g1 = fabs(g);
    if (g == 0.0) { *c = 1.0; *s = 0.0; *r = f; }
    else if (f == 0.0) { *c = 0.0; *s = copysign(1.0, g); *r = g1; }
    else if (f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax) {
        double d = sqrt(f * f + g * g);
        *c = f1 / d;
        *r = copysign(d, f);
        *s = g / *r;
    } else {
        double u = fmin(safmax, fmax(safmin, fmax(f1, g1)));
        double fs = f / u, gs = g / u;
        double d = sqrt(fs * fs + gs * gs);
        *c = fabs(fs) / d;
        *r = copysign(d, f);
        *s = gs / *r;
        *r = *r * u;
    }
}

/* BLAS-like dot: 32 interleaved partial sums (the reference's OpenBLAS kernel order is not
 * specified; parity on this path is to tolerance) */
static double dotn(const double *a, const double *b, long n) {
    double acc[32] = {0};
    long i = 0;
    for (; i + 32 <= n; i += 32)
        for (int j = 0; j < 32; ++j) acc[j] += a[i + j] * b[i + j];
    for (int j = 0; i < n; ++i, ++j) acc[j] += a[i] * b[i];
    for (int w = 16; w >= 1; w >>= 1)
        for (int j = 0; j < w; ++j) acc[j] += acc[j + w];
    return acc[0];
}
static double nrm2n(const double *a, long n) { return sqrt(dotn(a, a, n)); }

/* matvec of core/fusion_kernel_newton_solver.py:217-222: (L v - diag_term*v) on the interior */
static void newton_matvec(const double *v_flat, const double *diag_term, const double *r_grid, int nz, int nr,
                          double dr, double dz, double *v2d, double *lv, const double *zero, double *out) {
    int ni = nr - 2;
    memset(v2d, 0, sizeof(double) * (size_t)nz * nr);
    for (int iz = 1; iz < nz - 1; ++iz)
        memcpy(v2d + IDX(iz, 1), v_flat + (long)(iz - 1) * ni, sizeof(double) * (size_t)ni);
    orc_mg_residual(v2d, zero, r_grid, nz, nr, dr, dz, lv); /* apply_gs_operator: same stencil, zero source */
    for (int iz = 1; iz < nz - 1; ++iz)
        for (int ir = 1; ir < nr - 1; ++ir)
            out[(long)(iz - 1) * ni + (ir - 1)] = lv[IDX(iz, ir)] - diag_term[IDX(iz, ir)] * v2d[IDX(iz, ir)];
}

int orc_gmres_gs(const double *diag_term, const double *r_grid, int nz, int nr, double dr, double dz,
                 const double *rhs, const double *inv_diag, int restart, int maxiter, double atol_in,
                 double rtol, double *x, long *inner_out) {
    long n = (long)(nz - 2) * (nr - 2), ng = (long)nz * nr;
    double *v2d = dalloc(ng), *lv = dalloc(ng), *zero = dalloc(ng);
    double *r = dalloc(n), *w = dalloc(n), *av = dalloc(n);
    memset(x, 0, sizeof(double) * (size_t)n);
    long inner = 0;
    if (inner_out) *inner_out = 0;
    double bnrm2 = nrm2n(rhs, n);
    double atol = fmax(atol_in, rtol * bnrm2);
    if (bnrm2 == 0.0) { free(v2d); free(lv); free(zero); free(r); free(w); free(av); return 0; }
    const double eps = 2.220446049250313e-16;
    if (restart > n) restart = (int)n;
    double Mb;
    if (inv_diag) { for (long i = 0; i < n; ++i) w[i] = inv_diag[i] * rhs[i]; Mb = nrm2n(w, n); }
    else Mb = bnrm2;
    double ptol_max_factor = 1.0;
    double ptol = Mb * fmin(ptol_max_factor, atol / bnrm2);
    double presid = 0.0, rnorm = 0.0;
    double *v = dalloc((long)(restart + 1) * n);
    double *h = dalloc((long)restart * (restart + 1));
    double *giv = dalloc(2L * restart), *S = dalloc(restart + 1), *y = dalloc(restart + 1);
#define H(c, k) h[(long)(c) * (restart + 1) + (k)]
    int info = maxiter, done = 0;
    for (int it = 0; it < maxiter && !done; ++it) {
        if (it == 0) {
            memcpy(r, rhs, sizeof(double) * (size_t)n); /* x0 = 0 -> r = b.copy() */
            if (nrm2n(r, n) < atol) { info = 0; rnorm = 0.0; done = 2; break; }
        }
        if (inv_diag) for (long i = 0; i < n; ++i) v[i] = inv_diag[i] * r[i];
        else memcpy(v, r, sizeof(double) * (size_t)n);
        double tmp = nrm2n(v, n);
        double inv = 1.0 / tmp;
        for (long i = 0; i < n; ++i) v[i] *= inv;
        for (int i = 0; i <= restart; ++i) S[i] = 0.0;
        S[0] = tmp;
        int breakdown = 0, col = 0;
        for (col = 0; col < restart; ++col) {
            newton_matvec(v + (long)col * n, diag_term, r_grid, nz, nr, dr, dz, v2d, lv, zero, av);
            if (inv_diag) for (long i = 0; i < n; ++i) w[i] = inv_diag[i] * av[i];
            else memcpy(w, av, sizeof(double) * (size_t)n);
            double h0 = nrm2n(w, n);
            for (int k = 0; k <= col; ++k) {
                const double *vk = v + (long)k * n;
                double t = dotn(vk, w, n);
                H(col, k) = t;
                for (long i = 0; i < n; ++i) w[i] -= t * vk[i];
            }
            double h1 = nrm2n(w, n);
            H(col, col + 1) = h1;
            double *vn = v + (long)(col + 1) * n;
            memcpy(vn, w, sizeof(double) * (size_t)n);
            if (h1 <= eps * h0) { H(col, col + 1) = 0.0; breakdown = 1; }
            else { double ih = 1.0 / h1; for (long i = 0; i < n; ++i) vn[i] *= ih; }
            for (int k = 0; k < col; ++k) {
                double c = giv[2 * k], s = giv[2 * k + 1];
                double n0 = H(col, k), n1 = H(col, k + 1);
                H(col, k) = c * n0 + s * n1;
                H(col, k + 1) = -s * n0 + c * n1;
            }
            double c, s, mag;
            orc_dlartg(H(col, col), H(col, col + 1), &c, &s, &mag);
            giv[2 * col] = c; giv[2 * col + 1] = s;
            H(col, col) = mag; H(col, col + 1) = 0.0;
            double t2 = -s * S[col];
            S[col] = c * S[col];
            S[col + 1] = t2;
            presid = fabs(t2);
            inner += 1;
            if (presid <= ptol || breakdown) break;
        }
        if (col == restart) col = restart - 1; /* python's loop variable after exhaustion */
        if (H(col, col) == 0.0) S[col] = 0.0;
        for (int k = 0; k <= col; ++k) y[k] = S[k];
        for (int k = col; k > 0; --k)
            if (y[k] != 0.0) {
                y[k] /= H(k, k);
                double t = y[k];
                for (int j = 0; j < k; ++j) y[j] -= t * H(k, j);
            }
        if (y[0] != 0.0) y[0] /= H(0, 0);
        for (long i = 0; i < n; ++i) { /* x += y @ v[:col+1] */
            double acc = 0.0;
            for (int k = 0; k <= col; ++k) acc += y[k] * v[(long)k * n + i];
            x[i] += acc;
        }
        newton_matvec(x, diag_term, r_grid, nz, nr, dr, dz, v2d, lv, zero, av);
        for (long i = 0; i < n; ++i) r[i] = rhs[i] - av[i];
        rnorm = nrm2n(r, n);
        if (rnorm <= atol) break;
        else if (breakdown) break;
        else if (presid <= ptol) ptol_max_factor = fmax(eps, 0.25 * ptol_max_factor);
        else ptol_max_factor = fmin(1.0, 1.5 * ptol_max_factor);
        ptol = presid * fmin(ptol_max_factor, atol / rnorm);
    }
#undef H
    if (done != 2) info = (rnorm <= atol) ? 0 : maxiter;
    if (inner_out) *inner_out = inner;
    free(v2d); free(lv); free(zero); free(r); free(w); free(av); free(v); free(h); free(giv); free(S); free(y);
    return info;
}

void orc_profile_jacobian(const double *psi, const double *j_phi, const double *r_axis, int nz, int nr,
                          double dr, double dz, double psi_axis_in, double psi_bnd_in, double i_target,
                          int external_profile_mode, double *dj) {
    long n = (long)nz * nr;
    double psi_axis = isfinite(psi_axis_in) ? psi_axis_in : 0.0;
    double psi_bnd = psi_bnd_in;
    if (isnan(psi_bnd) || (isinf(psi_bnd) && psi_bnd > 0)) psi_bnd = psi_axis + 1e-9;
    else if (isinf(psi_bnd)) psi_bnd = psi_axis - 1e-9;
    double denom = psi_bnd - psi_axis;
    if (!isfinite(denom)) denom = 1e-9;
    if (fabs(denom) < 1e-9) denom = denom >= 0.0 ? 1e-9 : -1e-9;
    double *tmp = dalloc(n);
    unsigned char *mask = (unsigned char *)calloc((size_t)n, 1);
    for (int iz = 0; iz < nz; ++iz)
        for (int ir = 0; ir < nr; ++ir) {
            double ps = psi[IDX(iz, ir)];
            if (isnan(ps)) ps = psi_axis;
            else if (isinf(ps)) ps = ps > 0 ? psi_axis + 1e9 : psi_axis - 1e9;
            double rr = r_axis[ir];
            if (isnan(rr)) rr = 0.0; else if (isinf(rr)) rr = rr > 0 ? 1e6 : 0.0;
            rr = fmin(fmax(rr, 0.0), 1e6);
            double pn = (ps - psi_axis) / denom;
            int m = (pn >= 0) && (pn < 1.0);
            mask[IDX(iz, ir)] = (unsigned char)m;
            tmp[IDX(iz, ir)] = m ? (1 - pn) * rr : 0.0;
        }
    if (external_profile_mode) {
        double scale = fmax(fabs(denom), 1e-9);
        for (long i = 0; i < n; ++i) {
            double ja = j_phi[i];
            if (isnan(ja)) ja = 0.0; else if (isinf(ja)) ja = 1e12;
            ja = fabs(ja);
            double d = mask[i] ? -ja / scale : 0.0;
            if (isnan(d)) d = 0.0; else if (isinf(d)) d = d > 0 ? 0.0 : -1e12;
            dj[i] = d;
        }
        free(tmp); free(mask);
        return;
    }
    double it = isfinite(i_target) ? i_target : 0.0;
    double s = orc_np_sum(tmp, n) * dr * dz;
    if (!isfinite(s)) s = 0.0;
    double c = it / fmax(fabs(s), 1e-9);
    for (int iz = 0; iz < nz; ++iz)
        for (int ir = 0; ir < nr; ++ir) {
            double rr = r_axis[ir];
            if (isnan(rr)) rr = 0.0; else if (isinf(rr)) rr = rr > 0 ? 1e6 : 0.0;
            rr = fmin(fmax(rr, 0.0), 1e6);
            double d = mask[IDX(iz, ir)] ? (-c * rr) / denom : 0.0;
            if (isnan(d)) d = 0.0; else if (isinf(d)) d = d > 0 ? 1e12 : -1e12;
            dj[IDX(iz, ir)] = d;
        }
    free(tmp); free(mask);
}

/* sqrt(sum(r[1:-1,1:-1]**2)) and sqrt(mean(...)) of core/fusion_kernel_newton_solver.py:199-200 */
static void interior_norms(const double *r, int nz, int nr, double *sum_norm, double *rms) {
    long ni = (long)(nz - 2) * (nr - 2);
    double *sq = dalloc(ni);
    for (int iz = 1; iz < nz - 1; ++iz)
        for (int ir = 1; ir < nr - 1; ++ir) {
            double v = r[IDX(iz, ir)];
            sq[(long)(iz - 1) * (nr - 2) + (ir - 1)] = v * v;
        }
    double s = orc_np_sum(sq, ni);
    free(sq);
    if (sum_norm) *sum_norm = sqrt(s);
    if (rms) *rms = sqrt(s / (double)ni);
}

typedef struct {
    const orc_picard_params *p;
    const double *r_axis, *z_axis, *rr;
    double dr, dz;
} nctx;

/* axis + X-point + 0.1 heuristic + source update (when not external_profile_mode) */
static void newton_topology_source(const nctx *c, const double *psi, double *j_phi, double *psi_axis, double *psi_bnd) {
    const orc_picard_params *p = c->p;
    int aiz, air;
    long xidx;
    orc_find_axis(psi, p->nz, p->nr, &aiz, &air, psi_axis);
    if (p->xpoint_saddle) orc_find_x_point_saddle(psi, c->z_axis, p->nz, p->nr, c->dr, c->dz, p->z_min, &xidx, psi_bnd);
    else orc_find_x_point(psi, c->z_axis, p->nz, p->nr, c->dr, c->dz, p->z_min, &xidx, psi_bnd);
    if (fabs(*psi_axis - *psi_bnd) < 0.1) *psi_bnd = *psi_axis * 0.1;
    if (!p->external_profile_mode)
        orc_update_source(psi, c->r_axis, p->nz, p->nr, c->dr, c->dz, *psi_axis, *psi_bnd, p->mu0, p->i_target,
                          p->hmode, p->ped_p, p->ped_ff, j_phi);
}

int orc_newton_solve(const orc_picard_params *p, const orc_newton_opts *o, double *psi,
                     const double *boundary_flux, double *j_phi, orc_newton_result *res, double *hist,
                     double *gs_hist) {
    int nz = p->nz, nr = p->nr;
    long n = (long)nz * nr, ni = (long)(nz - 2) * (nr - 2);
    memset(res, 0, sizeof(*res));
    double *r_axis = dalloc(nr), *z_axis = dalloc(nz), *rr = dalloc(n);
    orc_linspace(p->r_min, p->r_max, nr, r_axis);
    orc_linspace(p->z_min, p->z_max, nz, z_axis);
    double dr = r_axis[1] - r_axis[0], dz = z_axis[1] - z_axis[0];
    meshgrid_r(r_axis, nz, nr, rr);
    nctx cx = {p, r_axis, z_axis, rr, dr, dz};

    double *bc = dalloc(n);
    if (p->has_boundary_flux && boundary_flux) memcpy(bc, boundary_flux, sizeof(double) * (size_t)n);
    else if (p->preserve_initial_state) memcpy(bc, psi, sizeof(double) * (size_t)n);
    else orc_vacuum_field(r_axis, z_axis, nz, nr, p->n_coils, p->coil_r, p->coil_z, p->coil_i, p->mu0, bc);
    if (p->preserve_initial_state) apply_bc(psi, bc, nz, nr);
    else memcpy(psi, bc, sizeof(double) * (size_t)n);

    double *source = dalloc(n), *psi_new = dalloc(n), *tmp = dalloc(n), *rk = dalloc(n), *dj = dalloc(n),
           *diag = dalloc(n), *rhs = dalloc(ni), *delta = dalloc(ni), *invd = dalloc(ni), *psi_prev = dalloc(n),
           *src_trial = dalloc(n);
    int nh = 0, ngh = 0;
    double last_hist = INFINITY, gs_best = INFINITY, gs_last = INFINITY;
    int converged = 0, final_iter = 0, have_source = 0;
    int max_iter = p->max_iter;
    int warmup = max_iter / 2 < 15 ? max_iter / 2 : 15;
#define PUSH_H(v) do { last_hist = (v); if (hist) hist[nh] = (v); nh++; } while (0)
#define PUSH_G(v) do { gs_last = (v); if (gs_hist) gs_hist[ngh] = (v); ngh++; if ((v) < gs_best) gs_best = (v); } while (0)
#define MAKE_SOURCE(dst) do { for (int iz = 0; iz < nz; ++iz) for (int ir = 0; ir < nr; ++ir) \
        (dst)[IDX(iz, ir)] = ((-p->mu0) * r_axis[ir]) * j_phi[IDX(iz, ir)]; } while (0)

    /* _seed_plasma */
    if (fabs(p->i_target) < 1e-12) {
        memset(j_phi, 0, sizeof(double) * (size_t)n);
    } else {
        double r_center = (p->r_min + p->r_max) / 2.0;
        for (int iz = 0; iz < nz; ++iz)
            for (int ir = 0; ir < nr; ++ir) {
                double a = r_axis[ir] - r_center;
                j_phi[IDX(iz, ir)] = exp(-(a * a + z_axis[iz] * z_axis[iz]) / 2.0);
            }
        double i_seed = (orc_np_sum(j_phi, n) * dr) * dz;
        if (i_seed > 0) {
            double s = p->i_target / i_seed;
            for (long i = 0; i < n; ++i) j_phi[i] *= s;
        }
        MAKE_SOURCE(source);
        for (int s = 0; s < 50; ++s) {
            orc_jacobi_step(psi, source, rr, nz, nr, dr, dz, tmp);
            memcpy(psi, tmp, sizeof(double) * (size_t)n);
        }
    }

    /* Phase A: Picard warm-up (the "newton" method string falls to the SOR branch of _elliptic_solve) */
    for (int k = 0; k < warmup; ++k) {
        final_iter = k;
        double psi_axis, psi_bnd;
        newton_topology_source(&cx, psi, j_phi, &psi_axis, &psi_bnd);
        MAKE_SOURCE(source);
        have_source = 1;
        orc_sor_step(psi, source, rr, nz, nr, dr, dz, p->sor_omega, psi_new);
        apply_bc(psi_new, bc, nz, nr);
        int bad = 0;
        for (long i = 0; i < n; ++i) if (!isfinite(psi_new[i])) { bad = 1; break; }
        if (bad) { res->diverged = 1; break; }
        for (long i = 0; i < n; ++i) tmp[i] = fabs(psi_new[i] - psi[i]);
        double diff = orc_np_sum(tmp, n) / (double)n;
        PUSH_H(diff);
        for (long i = 0; i < n; ++i) psi[i] = (1.0 - p->alpha) * psi[i] + p->alpha * psi_new[i];
        apply_bc(psi, bc, nz, nr);
        double gs = orc_gs_residual_rms(psi, source, rr, nz, nr, dr, dz);
        PUSH_G(gs);
        if (diff < p->tol && ((!p->require_gs_residual) || gs < p->gs_tol)) { converged = 1; break; }
    }

    /* Phase B: Newton */
    if (!converged) {
        for (int k = warmup; k < max_iter; ++k) {
            final_iter = k;
            double psi_axis, psi_bnd;
            newton_topology_source(&cx, psi, j_phi, &psi_axis, &psi_bnd);
            MAKE_SOURCE(source);
            have_source = 1;
            orc_mg_residual(psi, source, rr, nz, nr, dr, dz, rk);
            double res_norm, gs;
            interior_norms(rk, nz, nr, &res_norm, &gs);
            if (!o->line_search) PUSH_H(res_norm);
            PUSH_G(gs);
            if (res_norm < p->tol && ((!p->require_gs_residual) || gs < p->gs_tol)) {
                if (o->line_search) PUSH_H(res_norm);
                converged = 1;
                break;
            }
            orc_profile_jacobian(psi, j_phi, r_axis, nz, nr, dr, dz, psi_axis, psi_bnd, p->i_target,
                                 p->external_profile_mode, dj);
            for (int iz = 0; iz < nz; ++iz)
                for (int ir = 0; ir < nr; ++ir) diag[IDX(iz, ir)] = ((-p->mu0) * r_axis[ir]) * dj[IDX(iz, ir)];
            for (int iz = 1; iz < nz - 1; ++iz)
                for (int ir = 1; ir < nr - 1; ++ir) rhs[(long)(iz - 1) * (nr - 2) + (ir - 1)] = -rk[IDX(iz, ir)];
            const double *pre = NULL;
            if (o->precond == 1) {
                double dl = -2.0 / (dr * dr) - 2.0 / (dz * dz);
                for (int iz = 1; iz < nz - 1; ++iz)
                    for (int ir = 1; ir < nr - 1; ++ir) {
                        double dj_ = dl - diag[IDX(iz, ir)];
                        double sd = fabs(dj_) > 1e-12 ? dj_ : (dj_ >= 0.0 ? 1e-12 : -1e-12);
                        invd[(long)(iz - 1) * (nr - 2) + (ir - 1)] = 1.0 / sd;
                    }
                pre = invd;
            }
            long inner = 0;
            int info = orc_gmres_gs(diag, rr, nz, nr, dr, dz, rhs, pre, 50, 100, 1e-8, 1e-6, delta, &inner);
            res->gmres_matvecs += inner;
            res->gmres_last_info = info;
            if (info > 0) res->gmres_nonconverged += 1;
            else if (info < 0) res->gmres_breakdown += 1;

            if (o->line_search) {
                memcpy(psi_prev, psi, sizeof(double) * (size_t)n);
                double alpha = 0.5;
                int accepted = 0;
                memcpy(src_trial, source, sizeof(double) * (size_t)n);
                for (int b = 0; b < o->max_backtracks; ++b) {
                    res->ls_attempts += 1;
                    memcpy(psi, psi_prev, sizeof(double) * (size_t)n);
                    for (int iz = 1; iz < nz - 1; ++iz)
                        for (int ir = 1; ir < nr - 1; ++ir)
                            psi[IDX(iz, ir)] = psi_prev[IDX(iz, ir)] + alpha * delta[(long)(iz - 1) * (nr - 2) + (ir - 1)];
                    apply_bc(psi, bc, nz, nr);
                    if (!p->external_profile_mode) {
                        double ta, tb;
                        newton_topology_source(&cx, psi, j_phi, &ta, &tb);
                        MAKE_SOURCE(src_trial);
                    }
                    orc_mg_residual(psi, src_trial, rr, nz, nr, dr, dz, tmp);
                    double tn;
                    interior_norms(tmp, nz, nr, &tn, NULL);
                    if (tn <= (1.0 - o->line_search_c * alpha) * res_norm) { accepted = 1; res->ls_accepts += 1; break; }
                    res->ls_rejects += 1;
                    alpha *= 0.5;
                }
                if (!accepted) {
                    for (int iz = 1; iz < nz - 1; ++iz)
                        for (int ir = 1; ir < nr - 1; ++ir)
                            psi[IDX(iz, ir)] = psi_prev[IDX(iz, ir)] + alpha * delta[(long)(iz - 1) * (nr - 2) + (ir - 1)];
                    for (int ir = 0; ir < nr; ++ir) { psi[IDX(0, ir)] = psi_prev[IDX(0, ir)]; psi[IDX(nz - 1, ir)] = psi_prev[IDX(nz - 1, ir)]; }
                    for (int iz = 0; iz < nz; ++iz) { psi[IDX(iz, 0)] = psi_prev[IDX(iz, 0)]; psi[IDX(iz, nr - 1)] = psi_prev[IDX(iz, nr - 1)]; }
                    apply_bc(psi, bc, nz, nr);
                    if (!p->external_profile_mode) {
                        double ta, tb;
                        newton_topology_source(&cx, psi, j_phi, &ta, &tb);
                    }
                    /* Source stays the pre-step one (reference: only J_phi is refreshed here) */
                } else {
                    memcpy(source, src_trial, sizeof(double) * (size_t)n);
                }
                orc_mg_residual(psi, source, rr, nz, nr, dr, dz, tmp);
                double pn;
                interior_norms(tmp, nz, nr, &pn, NULL);
                PUSH_H(pn);
            } else {
                for (int iz = 1; iz < nz - 1; ++iz)
                    for (int ir = 1; ir < nr - 1; ++ir)
                        psi[IDX(iz, ir)] += 0.5 * delta[(long)(iz - 1) * (nr - 2) + (ir - 1)];
                apply_bc(psi, bc, nz, nr);
            }
            int bad = 0;
            for (long i = 0; i < n; ++i) if (!isfinite(psi[i])) { bad = 1; break; }
            if (bad) { res->diverged = 1; break; }
        }
    }
#undef PUSH_H
#undef PUSH_G
#undef MAKE_SOURCE
    res->converged = converged;
    res->iterations = final_iter + 1;
    res->residual = last_hist;
    res->gs_residual = have_source ? gs_last : INFINITY;
    res->gs_residual_best = have_source ? gs_best : INFINITY;
    res->n_hist = nh;
    res->n_gs_hist = ngh;
    free(r_axis); free(z_axis); free(rr); free(bc); free(source); free(psi_new); free(tmp); free(rk); free(dj);
    free(diag); free(rhs); free(delta); free(invd); free(psi_prev); free(src_trial);
    return 0;
}

/* -------------------------------------------------------------------------------------
 * Multigrid-preconditioned, finite-difference-Jacobian Newton lane (an extension of the HIP package, scpn_gs_newton_solve with
 * gmres_preconditioner = "multigrid", newton_jacobian = "finite_difference"; precedent in the reference:
 * core/jax_multigrid_precond.py:208-276).  CPU restatement of what the device lane computes, same arithmetic up to the order of the
 * global reductions:
 *   warm-up   : `warmup` Picard iterations of the reference's Newton dispatch (phase A above);
 *   each step : axis / X-point / source at psi_k, r_k = L psi_k - S(psi_k), history + convergence test as orc_newton_solve;
 *               solve J dx = -r_k by SciPy-style restarted GMRES (restart columns, maxiter cycles) LEFT-preconditioned by
 *               z = M^-1 v = ONE V-cycle(3,3), omega 1, min_grid 5, of the GS* operator from a zero start with v as source;
 *               J v = L v - (S(psi + eps v) - S(psi)) / eps with the FULL source map (axis, X-point, 0.1 heuristic, profile,
 *               sum-J renormalisation) and eps = 1e-7 sqrt(n_interior) max(|psi_axis|, 1e-6) / ||v||_2 (||v|| := 1 for the unit
 *               Krylov basis vectors, as the device does); x0 = 0; the LAST cycle of a capped solve exits on the Arnoldi
 *               estimate (no true residual);
 *               psi_interior += newton_alpha * dx, Dirichlet ring re-applied, NaN / Inf check.
 * Vectors live on the full (nz, nr) grid with a zero ring.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    const nctx *cx;
    const double *psi, *source0; /* state of this Newton step */
    double psi_axis;
    double *psip, *jtmp, *srcp, *lv, *zero; /* scratch, full grids */
} fdctx;

static double grid_norm2(const double *a, long n) { return nrm2n(a, n); }

/* out = J v on the interior (ring 0); unit: v is a unit-norm Krylov basis vector */
static void fd_matvec(const fdctx *f, const double *v, int unit, double *out) {
    const orc_picard_params *p = f->cx->p;
    int nz = p->nz, nr = p->nr;
    long n = (long)nz * nr, ni = (long)(nz - 2) * (nr - 2);
    double nv2 = unit ? 1.0 : dotn(v, v, n);
    double eps = nv2 > 0.0 ? 1e-7 * sqrt((double)ni) * fmax(fabs(f->psi_axis), 1e-6) / sqrt(nv2) : 0.0;
    for (long i = 0; i < n; ++i) f->psip[i] = f->psi[i] + eps * v[i];
    double pa, pb;
    newton_topology_source(f->cx, f->psip, f->jtmp, &pa, &pb);
    for (int iz = 0; iz < nz; ++iz)
        for (int ir = 0; ir < nr; ++ir) f->srcp[IDX(iz, ir)] = ((-p->mu0) * f->cx->r_axis[ir]) * f->jtmp[IDX(iz, ir)];
    orc_mg_residual(v, f->zero, f->cx->rr, nz, nr, f->cx->dr, f->cx->dz, f->lv); /* L v (zero source, zero ring) */
    memset(out, 0, sizeof(double) * (size_t)n);
    for (int iz = 1; iz < nz - 1; ++iz)
        for (int ir = 1; ir < nr - 1; ++ir) {
            double ds = eps != 0.0 ? (f->srcp[IDX(iz, ir)] - f->source0[IDX(iz, ir)]) / eps : 0.0;
            out[IDX(iz, ir)] = f->lv[IDX(iz, ir)] - ds;
        }
}
/* z = one V-cycle from zero with v as source; ring of z zeroed (Krylov-grid convention) */
static void mg_precond(const nctx *cx, const double *v, double *z) {
    const orc_picard_params *p = cx->p;
    long n = (long)p->nz * p->nr;
    memset(z, 0, sizeof(double) * (size_t)n);
    orc_multigrid_vcycle(z, v, cx->rr, p->nz, p->nr, cx->dr, cx->dz, 1.0, 3, 3, 5);
}

static int gmres_mg_fd(const fdctx *f, const double *rhs, int restart, int maxiter, double atol_in, double rtol, double *x,
                       long *inner_out) {
    const orc_picard_params *p = f->cx->p;
    const int nr = p->nr;
    long n = (long)p->nz * p->nr;
    double *r = dalloc(n), *w = dalloc(n), *av = dalloc(n), *v = dalloc((long)(restart + 1) * n);
    double *h = dalloc((long)restart * (restart + 1)), *giv = dalloc(2L * restart), *S = dalloc(restart + 1), *y = dalloc(restart + 1);
    memset(x, 0, sizeof(double) * (size_t)n);
    long inner = 0;
    int info = maxiter;
    const double eps = 2.220446049250313e-16;
    double bnrm2 = grid_norm2(rhs, n);
    double atol = fmax(atol_in, rtol * bnrm2);
    if (bnrm2 == 0.0 || bnrm2 < atol) { info = 0; goto done; } /* x = 0 is the answer */
    {
        mg_precond(f->cx, rhs, w);
        double Mb = grid_norm2(w, n);
        double ptol_max_factor = 1.0, ptol = Mb * fmin(1.0, atol / bnrm2), presid = 0.0, rnorm = bnrm2;
        memcpy(r, rhs, sizeof(double) * (size_t)n);
#define H(c, k) h[(long)(c) * (restart + 1) + (k)]
        for (int it = 0; it < maxiter; ++it) {
            if (it == 0) memcpy(v, w, sizeof(double) * (size_t)n); /* M^-1 b is the first (unscaled) Krylov vector */
            else mg_precond(f->cx, r, v);
            double tmp = grid_norm2(v, n), inv = 1.0 / tmp;
            for (long i = 0; i < n; ++i) v[i] *= inv;
            for (int i = 0; i <= restart; ++i) S[i] = 0.0;
            S[0] = tmp;
            int breakdown = 0, col = 0;
            for (col = 0; col < restart; ++col) {
                fd_matvec(f, v + (long)col * n, 1, av);
                mg_precond(f->cx, av, w);
                double h0 = grid_norm2(w, n);
                for (int k = 0; k <= col; ++k) {
                    const double *vk = v + (long)k * n;
                    double t = dotn(vk, w, n);
                    H(col, k) = t;
                    for (long i = 0; i < n; ++i) w[i] -= t * vk[i];
                }
                double h1 = grid_norm2(w, n);
                H(col, col + 1) = h1;
                double *vn = v + (long)(col + 1) * n;
                memcpy(vn, w, sizeof(double) * (size_t)n);
                if (h1 <= eps * h0) { H(col, col + 1) = 0.0; breakdown = 1; }
                else { double ih = 1.0 / h1; for (long i = 0; i < n; ++i) vn[i] *= ih; }
                for (int k = 0; k < col; ++k) {
                    double c = giv[2 * k], s_ = giv[2 * k + 1], n0 = H(col, k), n1 = H(col, k + 1);
                    H(col, k) = c * n0 + s_ * n1;
                    H(col, k + 1) = -s_ * n0 + c * n1;
                }
                double c, s_, mag;
                orc_dlartg(H(col, col), H(col, col + 1), &c, &s_, &mag);
                giv[2 * col] = c; giv[2 * col + 1] = s_;
                H(col, col) = mag; H(col, col + 1) = 0.0;
                double t2 = -s_ * S[col];
                S[col] = c * S[col];
                S[col + 1] = t2;
                presid = fabs(t2);
                inner += 1;
                if (presid <= ptol || breakdown) break;
            }
            if (col == restart) col = restart - 1;
            if (H(col, col) == 0.0) S[col] = 0.0;
            for (int k = 0; k <= col; ++k) y[k] = S[k];
            for (int k = col; k > 0; --k)
                if (y[k] != 0.0) {
                    y[k] /= H(k, k);
                    double t = y[k];
                    for (int j = 0; j < k; ++j) y[j] -= t * H(k, j);
                }
            if (y[0] != 0.0) y[0] /= H(0, 0);
            for (long i = 0; i < n; ++i) {
                double acc = 0.0;
                for (int k = 0; k <= col; ++k) acc += y[k] * v[(long)k * n + i];
                x[i] += acc;
            }
            if (it + 1 == maxiter) { /* capped inexact solve: judged on the Arnoldi estimate of the preconditioned residual */
                info = (presid <= ptol || breakdown) ? 0 : maxiter;
                break;
            }
            fd_matvec(f, x, 0, av);
            for (long i = 0; i < n; ++i) r[i] = rhs[i] - av[i];
            for (int ir = 0; ir < p->nr; ++ir) { r[IDX(0, ir)] = 0.0; r[IDX(p->nz - 1, ir)] = 0.0; }
            for (int iz = 0; iz < p->nz; ++iz) { r[IDX(iz, 0)] = 0.0; r[IDX(iz, p->nr - 1)] = 0.0; }
            rnorm = grid_norm2(r, n);
            if (rnorm <= atol) { info = 0; break; }
            else if (breakdown) { info = maxiter; break; }
            else if (presid <= ptol) ptol_max_factor = fmax(eps, 0.25 * ptol_max_factor);
            else ptol_max_factor = fmin(1.0, 1.5 * ptol_max_factor);
            ptol = presid * fmin(ptol_max_factor, atol / rnorm);
        }
#undef H
    }
done:
    if (inner_out) *inner_out = inner;
    free(r); free(w); free(av); free(v); free(h); free(giv); free(S); free(y);
    return info;
}

int orc_newton_mg_fd(const orc_picard_params *p, const orc_newton_mg_opts *o, double *psi, const double *boundary_flux,
                     double *j_phi, orc_newton_result *res, double *hist, double *gs_hist) {
    int nz = p->nz, nr = p->nr;
    long n = (long)nz * nr;
    memset(res, 0, sizeof(*res));
    double *r_axis = dalloc(nr), *z_axis = dalloc(nz), *rr = dalloc(n);
    orc_linspace(p->r_min, p->r_max, nr, r_axis);
    orc_linspace(p->z_min, p->z_max, nz, z_axis);
    double dr = r_axis[1] - r_axis[0], dz = z_axis[1] - z_axis[0];
    meshgrid_r(r_axis, nz, nr, rr);
    nctx cx = {p, r_axis, z_axis, rr, dr, dz};
    double *bc = dalloc(n);
    if (p->has_boundary_flux && boundary_flux) memcpy(bc, boundary_flux, sizeof(double) * (size_t)n);
    else if (p->preserve_initial_state) memcpy(bc, psi, sizeof(double) * (size_t)n);
    else orc_vacuum_field(r_axis, z_axis, nz, nr, p->n_coils, p->coil_r, p->coil_z, p->coil_i, p->mu0, bc);
    if (p->preserve_initial_state) apply_bc(psi, bc, nz, nr);
    else memcpy(psi, bc, sizeof(double) * (size_t)n);
    double *source = dalloc(n), *psi_new = dalloc(n), *tmp = dalloc(n), *rk = dalloc(n), *rhs = dalloc(n), *dx = dalloc(n);
    double *psip = dalloc(n), *jtmp = dalloc(n), *srcp = dalloc(n), *lv = dalloc(n), *zero = dalloc(n);
    memset(zero, 0, sizeof(double) * (size_t)n);
    int nh = 0, ngh = 0, converged = 0, final_iter = 0, have_source = 0;
    double last_hist = INFINITY, gs_best = INFINITY, gs_last = INFINITY;
    int max_iter = p->max_iter;
    int warmup = o->warmup_steps >= 0 ? (o->warmup_steps < max_iter ? o->warmup_steps : max_iter) : (max_iter / 2 < 15 ? max_iter / 2 : 15);
#define PUSH_H(v) do { last_hist = (v); if (hist) hist[nh] = (v); nh++; } while (0)
#define PUSH_G(v) do { gs_last = (v); if (gs_hist) gs_hist[ngh] = (v); ngh++; if ((v) < gs_best) gs_best = (v); } while (0)
#define MAKE_SOURCE(dst) do { for (int iz = 0; iz < nz; ++iz) for (int ir = 0; ir < nr; ++ir) \
        (dst)[IDX(iz, ir)] = ((-p->mu0) * r_axis[ir]) * j_phi[IDX(iz, ir)]; } while (0)
    /* _seed_plasma */
    if (fabs(p->i_target) < 1e-12) {
        memset(j_phi, 0, sizeof(double) * (size_t)n);
    } else {
        double r_center = (p->r_min + p->r_max) / 2.0;
        for (int iz = 0; iz < nz; ++iz)
            for (int ir = 0; ir < nr; ++ir) {
                double a = r_axis[ir] - r_center;
                j_phi[IDX(iz, ir)] = exp(-(a * a + z_axis[iz] * z_axis[iz]) / 2.0);
            }
        double i_seed = (orc_np_sum(j_phi, n) * dr) * dz;
        if (i_seed > 0) { double s = p->i_target / i_seed; for (long i = 0; i < n; ++i) j_phi[i] *= s; }
        MAKE_SOURCE(source);
        for (int s = 0; s < 50; ++s) { orc_jacobi_step(psi, source, rr, nz, nr, dr, dz, tmp); memcpy(psi, tmp, sizeof(double) * (size_t)n); }
    }
    /* phase A: Picard warm-up, as orc_newton_solve */
    for (int k = 0; k < warmup; ++k) {
        final_iter = k;
        double psi_axis, psi_bnd;
        newton_topology_source(&cx, psi, j_phi, &psi_axis, &psi_bnd);
        MAKE_SOURCE(source);
        have_source = 1;
        orc_sor_step(psi, source, rr, nz, nr, dr, dz, p->sor_omega, psi_new);
        apply_bc(psi_new, bc, nz, nr);
        int bad = 0;
        for (long i = 0; i < n; ++i) if (!isfinite(psi_new[i])) { bad = 1; break; }
        if (bad) { res->diverged = 1; break; }
        for (long i = 0; i < n; ++i) tmp[i] = fabs(psi_new[i] - psi[i]);
        double diff = orc_np_sum(tmp, n) / (double)n;
        PUSH_H(diff);
        for (long i = 0; i < n; ++i) psi[i] = (1.0 - p->alpha) * psi[i] + p->alpha * psi_new[i];
        apply_bc(psi, bc, nz, nr);
        double gs = orc_gs_residual_rms(psi, source, rr, nz, nr, dr, dz);
        PUSH_G(gs);
        if (diff < p->tol && ((!p->require_gs_residual) || gs < p->gs_tol)) { converged = 1; break; }
    }
    /* phase B */
    if (!converged && !res->diverged) {
        for (int k = warmup; k < max_iter; ++k) {
            final_iter = k;
            double psi_axis, psi_bnd;
            newton_topology_source(&cx, psi, j_phi, &psi_axis, &psi_bnd);
            MAKE_SOURCE(source);
            have_source = 1;
            orc_mg_residual(psi, source, rr, nz, nr, dr, dz, rk);
            double res_norm, gs;
            interior_norms(rk, nz, nr, &res_norm, &gs);
            PUSH_H(res_norm);
            PUSH_G(gs);
            if (res_norm < p->tol && ((!p->require_gs_residual) || gs < p->gs_tol)) { converged = 1; break; }
            memset(rhs, 0, sizeof(double) * (size_t)n);
            for (int iz = 1; iz < nz - 1; ++iz)
                for (int ir = 1; ir < nr - 1; ++ir) rhs[IDX(iz, ir)] = -rk[IDX(iz, ir)];
            fdctx f = {&cx, psi, source, psi_axis, psip, jtmp, srcp, lv, zero};
            long inner = 0;
            int info = gmres_mg_fd(&f, rhs, o->gmres_restart, o->gmres_maxiter, o->gmres_atol, o->gmres_rtol, dx, &inner);
            res->gmres_matvecs += inner;
            res->gmres_last_info = info;
            if (info > 0) res->gmres_nonconverged += 1;
            for (int iz = 1; iz < nz - 1; ++iz)
                for (int ir = 1; ir < nr - 1; ++ir) psi[IDX(iz, ir)] += o->newton_alpha * dx[IDX(iz, ir)];
            apply_bc(psi, bc, nz, nr);
            int bad = 0;
            for (long i = 0; i < n; ++i) if (!isfinite(psi[i])) { bad = 1; break; }
            if (bad) { res->diverged = 1; break; }
        }
    }
#undef PUSH_H
#undef PUSH_G
#undef MAKE_SOURCE
    res->converged = converged;
    res->iterations = final_iter + 1;
    res->residual = last_hist;
    res->gs_residual = have_source ? gs_last : INFINITY;
    res->gs_residual_best = have_source ? gs_best : INFINITY;
    res->n_hist = nh;
    res->n_gs_hist = ngh;
    free(r_axis); free(z_axis); free(rr); free(bc); free(source); free(psi_new); free(tmp); free(rk); free(rhs); free(dx);
    free(psip); free(jtmp); free(srcp); free(lv); free(zero);
    return 0;
}

/* =====================================================================================
 * f3: field products  
 * ===================================================================================== */
static int searchsorted_left(const double *a, int n, double x) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}
double orc_interp_psi(const double *psi, const double *r_axis, const double *z_axis, int nz, int nr, double dr,
                      double dz, double r_pt, double z_pt) {
    int ir = searchsorted_left(r_axis, nr, r_pt) - 1, iz = searchsorted_left(z_axis, nz, z_pt) - 1;
    if (ir < 0) ir = 0;
    if (ir > nr - 2) ir = nr - 2;
    if (iz < 0) iz = 0;
    if (iz > nz - 2) iz = nz - 2;
    double tr = (r_pt - r_axis[ir]) / dr, tz = (z_pt - z_axis[iz]) / dz;
    tr = fmax(0.0, fmin(1.0, tr));
    tz = fmax(0.0, fmin(1.0, tz));
    double v = ((1 - tr) * (1 - tz)) * psi[IDX(iz, ir)];
    v = v + (tr * (1 - tz)) * psi[IDX(iz, ir + 1)];
    v = v + ((1 - tr) * tz) * psi[IDX(iz + 1, ir)];
    v = v + (tr * tz) * psi[IDX(iz + 1, ir + 1)];
    return v;
}
void orc_b_field(const double *psi, const double *r_axis, int nz, int nr, double dr, double dz, double *b_r,
                 double *b_z) {
    for (int iz = 0; iz < nz; ++iz)
        for (int ir = 0; ir < nr; ++ir) {
            double gz, gr;
            if (iz == 0) gz = (psi[IDX(1, ir)] - psi[IDX(0, ir)]) / dz;
            else if (iz == nz - 1) gz = (psi[IDX(nz - 1, ir)] - psi[IDX(nz - 2, ir)]) / dz;
            else gz = (psi[IDX(iz + 1, ir)] - psi[IDX(iz - 1, ir)]) / (2.0 * dz);
            if (ir == 0) gr = (psi[IDX(iz, 1)] - psi[IDX(iz, 0)]) / dr;
            else if (ir == nr - 1) gr = (psi[IDX(iz, nr - 1)] - psi[IDX(iz, nr - 2)]) / dr;
            else gr = (psi[IDX(iz, ir + 1)] - psi[IDX(iz, ir - 1)]) / (2.0 * dr);
            double rs = fmax(r_axis[ir], 1e-6);
            b_r[IDX(iz, ir)] = -(1.0 / rs) * gz;
            b_z[IDX(iz, ir)] = (1.0 / rs) * gr;
        }
}
void orc_probe_response(int n_coils, const double *coil_r, const double *coil_z, const double *coil_turns, int n_flux,
                        const double *fr, const double *fz, int n_b, const double *br, const double *bz,
                        const int *bdir, double *out) {
    for (int c = 0; c < n_coils; ++c) {
        double turns = coil_turns ? coil_turns[c] : 1.0, rc = coil_r[c], zc = coil_z[c];
        for (int i = 0; i < n_flux; ++i) out[(long)i * n_coils + c] = turns * orc_green_function(rc, zc, fr[i], fz[i]);
        for (int q = 0; q < n_b; ++q) {
            double r_obs = br[q], z_obs = bz[q];
            double eps_r = fmax(1.0e-5, 1.0e-5 * fabs(r_obs)), eps_z = fmax(1.0e-5, 1.0e-5 * (1.0 + fabs(z_obs)));
            double r_safe = fmax(r_obs, eps_r), v;
            if (bdir[q] == 0) {
                double plus = turns * orc_green_function(rc, zc, r_safe, z_obs + eps_z);
                double minus = turns * orc_green_function(rc, zc, r_safe, z_obs - eps_z);
                v = -(plus - minus) / ((2.0 * eps_z) * r_safe);
            } else {
                double plus = turns * orc_green_function(rc, zc, r_safe + eps_r, z_obs);
                double minus = turns * orc_green_function(rc, zc, r_safe - eps_r, z_obs);
                v = (plus - minus) / ((2.0 * eps_r) * r_safe);
            }
            out[(long)(n_flux + q) * n_coils + c] = v;
        }
    }
}

/* =====================================================================================
 * f4: transport coupling  
 * ===================================================================================== */
/* numpy.interp (numpy/_core/src/multiarray/compiled_base.c arr_interp; third-party NumPy 2.2.6): xp ascending, left = fp[0],
 * right = fp[m-1]; slope form with the non-finite fall-backs */
void orc_np_interp(const double *x, long n, const double *xp, const double *fp, int m, double *out) {
    for (long i = 0; i < n; ++i) {
        double xv = x[i];
        if (isnan(xv)) { out[i] = xv; continue; }
        if (xv > xp[m - 1]) { out[i] = fp[m - 1]; continue; }
        if (xv < xp[0]) { out[i] = fp[0]; continue; }
        int lo = 0, hi = m; /* largest j with xp[j] <= xv */
        while (hi - lo > 1) { int mid = (lo + hi) / 2; if (xp[mid] <= xv) lo = mid; else hi = mid; }
        int j = lo;
        if (j == m - 1) { out[i] = fp[j]; continue; }
        if (xp[j] == xv) { out[i] = fp[j]; continue; }
        double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
        double r = slope * (xv - xp[j]) + fp[j];
        if (isnan(r)) {
            r = slope * (xv - xp[j + 1]) + fp[j + 1];
            if (isnan(r) && fp[j] == fp[j + 1]) r = fp[j];
        }
        out[i] = r;
    }
}

/* core/integrated_transport_solver_coupling.py:39-73 map_profiles_to_2d.  pressure_1d = ne*(Ti+Te), jbs_1d = the bootstrap
 * profile (both evaluated by the transport side).  Writes Pressure_2D and J_phi; returns I_current before renormalisation */
double orc_map_profiles_to_2d(const double *psi, const double *r_axis, const double *z_axis, int nz, int nr, double dr,
                              double dz, double z_min_cfg, double i_target, const double *rho, const double *pressure_1d,
                              const double *jbs_1d, int n_rho, double *pressure_2d, double *j_phi) {
    long n = (long)nz * nr;
    int aiz, air;
    double psi_axis, psi_x;
    long xidx;
    /* np.argmax(self.Psi) -- plain argmax, no |psi_axis| floor here */
    long imax = 0;
    for (long i = 1; i < n; ++i) if (psi[i] > psi[imax] || (isnan(psi[i]) && !isnan(psi[imax]))) imax = i;
    (void)aiz; (void)air;
    psi_axis = psi[imax];
    orc_find_x_point(psi, z_axis, nz, nr, dr, dz, z_min_cfg, &xidx, &psi_x);
    double psi_edge = psi_x;
    if (fabs(psi_edge - psi_axis) < 1.0) {
        double mn = psi[0];
        for (long i = 1; i < n; ++i) if (psi[i] < mn || isnan(psi[i])) mn = psi[i];
        psi_edge = mn;
    }
    double denom = psi_edge - psi_axis;
    if (fabs(denom) < 1e-9) denom = 1e-9;
    double *rho2 = dalloc(n), *jb = dalloc(n);
    for (long i = 0; i < n; ++i) {
        double pn = (psi[i] - psi_axis) / denom;
        pn = fmin(fmax(pn, 0.0), 1.0);
        rho2[i] = sqrt(pn);
    }
    orc_np_interp(rho2, n, rho, pressure_1d, n_rho, pressure_2d);
    orc_np_interp(rho2, n, rho, jbs_1d, n_rho, jb);
    for (int iz = 0; iz < nz; ++iz)
        for (int ir = 0; ir < nr; ++ir) j_phi[IDX(iz, ir)] = (pressure_2d[IDX(iz, ir)] * r_axis[ir]) + jb[IDX(iz, ir)];
    double i_curr = orc_np_sum(j_phi, n) * dr * dz;
    if (fabs(i_curr) > 1e-9) {
        double s = i_target / i_curr;
        for (long i = 0; i < n; ++i) j_phi[i] *= s;
    }
    free(rho2); free(jb);
    return i_curr;
}

/* LAPACK dgtsv (one right-hand side) -- what scipy.linalg.solve_banded((1,1), ...) calls (SciPy 1.15.3 linalg/_basic.py).
 * dl[n-1], d[n], du[n-1], b[n] are overwritten; returns info (0 ok, i>0: U(i,i) exactly zero) */
int orc_gtsv(int n, double *dl, double *d, double *du, double *b) {
    if (n == 0) return 0;
    for (int i = 0; i < n - 2; ++i) {
        if (fabs(d[i]) >= fabs(dl[i])) {
            if (d[i] != 0.0) {
                double fact = dl[i] / d[i];
                d[i + 1] = d[i + 1] - fact * du[i];
                b[i + 1] = b[i + 1] - fact * b[i];
            } else return i + 1;
            dl[i] = 0.0;
        } else {
            double fact = d[i] / dl[i];
            d[i] = dl[i];
            double temp = d[i + 1];
            d[i + 1] = du[i] - fact * temp;
            dl[i] = du[i + 1];
            du[i + 1] = -fact * dl[i];
            du[i] = temp;
            temp = b[i];
            b[i] = b[i + 1];
            b[i + 1] = temp - fact * b[i + 1];
        }
    }
    if (n > 1) {
        int i = n - 2;
        if (fabs(d[i]) >= fabs(dl[i])) {
            if (d[i] != 0.0) {
                double fact = dl[i] / d[i];
                d[i + 1] = d[i + 1] - fact * du[i];
                b[i + 1] = b[i + 1] - fact * b[i];
            } else return i + 1;
        } else {
            double fact = d[i] / dl[i];
            d[i] = dl[i];
            double temp = d[i + 1];
            d[i + 1] = du[i] - fact * temp;
            du[i] = temp;
            temp = b[i];
            b[i] = b[i + 1];
            b[i + 1] = temp - fact * b[i + 1];
        }
    }
    if (d[n - 1] == 0.0) return n;
    b[n - 1] = b[n - 1] / d[n - 1];
    if (n > 1) b[n - 2] = (b[n - 2] - du[n - 2] * b[n - 1]) / d[n - 2];
    for (int i = n - 3; i >= 0; --i) b[i] = (b[i] - du[i] * b[i + 1] - dl[i] * b[i + 2]) / d[i];
    return 0;
}

/* core/current_diffusion.py:175-207 neoclassical_resistivity (Sauter) */
double orc_neoclassical_resistivity(double te_kev, double ne_19, double z_eff, double epsilon) {
    te_kev = fmax(te_kev, 1e-3);
    ne_19 = fmax(ne_19, 1e-3);
    epsilon = fmax(epsilon, 1e-6);
    (void)ne_19;
    double ln_lambda = 17.0;
    double eta_sp = 1.65e-9 * z_eff * ln_lambda / pow(te_kev, 1.5);
    double f_t = 1.0 - pow(1.0 - epsilon, 2.0) / (sqrt(1.0 - pow(epsilon, 2.0)) * (1.0 + 1.46 * sqrt(epsilon)));
    f_t = fmax(0.0, fmin(f_t, 1.0));
    double c_r = 1.0 - (1.0 + 0.36 / z_eff) * f_t + (0.59 / z_eff) * pow(f_t, 2.0);
    double eta_neo = eta_sp / (1.0 - f_t) * c_r;
    return fmax(eta_neo, eta_sp);
}

/* core/current_diffusion.py:210-233 q_from_psi */
void orc_q_from_psi(const double *rho, const double *psi, int n, double r0, double a, double b0, double *q) {
    double drho = rho[1] - rho[0];
    double *g = dalloc(n);
    g[0] = (psi[1] - psi[0]) / drho;
    g[n - 1] = (psi[n - 1] - psi[n - 2]) / drho;
    for (int i = 1; i < n - 1; ++i) g[i] = (psi[i + 1] - psi[i - 1]) / (2.0 * drho);
    q[0] = 0.0;
    for (int i = 1; i < n; ++i) {
        double denom = r0 * g[i];
        if (fabs(denom) < 1e-12) q[i] = i > 1 ? q[i - 1] : 1.0;
        else q[i] = -rho[i] * pow(a, 2.0) * b0 / denom;
    }
    double d2 = (psi[2] - 2 * psi[1] + psi[0]) / pow(drho, 2.0);
    if (fabs(d2) > 1e-12) q[0] = -pow(a, 2.0) * b0 / (r0 * d2);
    else q[0] = q[1];
    for (int i = 0; i < n; ++i) q[i] = fabs(q[i]);
    free(g);
}

/* core/current_diffusion.py:277-353 CurrentDiffusionSolver.step; psi in/out; returns dgtsv info */
int orc_current_diffusion_step(const double *rho, double *psi, int n, double r0, double a, double b0, double dt,
                               const double *te, const double *ne, double z_eff, const double *j_bs, const double *j_cd,
                               const double *j_ext) {
    const double mu0 = 4.0 * M_PI * 1e-7;
    double drho = rho[1] - rho[0];
    double *q = dalloc(n), *eta = dalloc(n), *D = dalloc(n), *sub = dalloc(n), *diag = dalloc(n), *sup = dalloc(n),
           *rhs = dalloc(n), *jt = dalloc(n);
    for (int i = 0; i < n; ++i) jt[i] = (j_bs[i] + j_cd[i]) + (j_ext ? j_ext[i] : 0.0);
    orc_q_from_psi(rho, psi, n, r0, a, b0, q);
    for (int i = 0; i < n; ++i) {
        double eps = rho[i] * a / r0;
        eta[i] = orc_neoclassical_resistivity(te[i], ne[i], z_eff, eps);
        D[i] = eta[i] / (mu0 * pow(a, 2.0));
    }
    double alpha = dt / 2.0, drho2 = pow(drho, 2.0);
    diag[0] = 1.0 + alpha * 4.0 * D[0] / drho2;
    sup[0] = -alpha * 4.0 * D[0] / drho2;
    rhs[0] = psi[0] + alpha * 4.0 * D[0] * (psi[1] - psi[0]) / drho2 + dt * r0 * eta[0] * jt[0];
    for (int i = 1; i < n - 1; ++i) {
        double r = rho[i];
        double cp = D[i] * (1.0 / drho2 - 1.0 / (2.0 * r * drho));
        double cc = D[i] * (-2.0 / drho2);
        double cn = D[i] * (1.0 / drho2 + 1.0 / (2.0 * r * drho));
        sub[i] = -alpha * cp;
        diag[i] = 1.0 - alpha * cc;
        sup[i] = -alpha * cn;
        double l = cp * psi[i - 1] + cc * psi[i] + cn * psi[i + 1];
        rhs[i] = psi[i] + alpha * l + dt * r0 * eta[i] * jt[i];
    }
    diag[n - 1] = 1.0;
    sub[n - 1] = 0.0;
    rhs[n - 1] = psi[n - 1];
    /* ab[0,1:] = sup[:-1]; ab[1,:] = diag; ab[2,:-1] = sub[1:]  ->  gtsv(dl = sub[1:], d = diag, du = sup[:-1], b = rhs) */
    int info = orc_gtsv(n, sub + 1, diag, sup, rhs);
    memcpy(psi, rhs, sizeof(double) * (size_t)n);
    free(q); free(eta); free(D); free(sub); free(diag); free(sup); free(rhs); free(jt);
    return info;
}

/* core/current_diffusion.py:55-172 solve_flux_evolution_nonadiabatic, the time recurrence (:141-158) given the sampled
 * damping rate 1/tau_psi and source R_null*E_theta - eta*J_theta at the n_steps+1 time points ((n_steps+1) x n rows):
 *   gamma = (rate[k]+rate[k+1])/2, s = (source[k]+source[k+1])/2, decay = exp(-gamma dt),
 *   inc = gamma > 0 ? s (1-decay)/gamma : dt s,  drop = psi (1-decay),  psi' = psi - drop + inc.
 * psi: (n_steps+1) x n (row 0 = psi0 on entry); inc / drop / resid: n_steps x n */
void orc_flux_evolution(int n, int n_steps, double dt, const double *rate, const double *source, double *psi, double *inc,
                        double *drop, double *resid) {
    for (int k = 0; k < n_steps; ++k)
        for (int i = 0; i < n; ++i) {
            double gamma = 0.5 * (rate[(long)k * n + i] + rate[(long)(k + 1) * n + i]);
            double s = 0.5 * (source[(long)k * n + i] + source[(long)(k + 1) * n + i]);
            double decay = exp(-gamma * dt);
            double di = gamma > 0.0 ? s * (1.0 - decay) / gamma : dt * s;
            double p = psi[(long)k * n + i];
            double dd = p * (1.0 - decay);
            double nx = p - dd + di;
            psi[(long)(k + 1) * n + i] = nx;
            inc[(long)k * n + i] = di;
            drop[(long)k * n + i] = dd;
            resid[(long)k * n + i] = nx - nx;
        }
}

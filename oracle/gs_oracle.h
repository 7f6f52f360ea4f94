/*
 * gs_oracle.h -- CPU restatement (plain C, fp64) of the reference's Grad-Shafranov hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker / the CPU baseline.  The product path (scpn_fusion_core_amd + libscpn_hip.so)
 * never links, imports or calls this code.
 *
 * Parity status: PINNED.  Every function here is checked against golden vectors produced
 * by importing the reference Python package in the build container
 * (tests/golden/make_golden.py -> tests/golden/ npz files; tests/test_oracle_golden.py).
 *
 * All fields are C-contiguous float64, shape (nz, nr), index [iz*nr + ir]  (Z slow, R fast),
 * exactly as the reference (core/fusion_kernel.py:163-174).
 * Build with -ffp-contract=off: the restatement keeps NumPy's operation order and
 * must not be fused into FMAs.
 */
#ifndef GS_ORACLE_H
#define GS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* numpy.linspace(start, stop, n) (endpoint=True) -- used by core/fusion_kernel.py:165-166,
 * core/multigrid_solve.py:428-429 */
void orc_linspace(double start, double stop, int n, double *out);

/* numpy pairwise add.reduce over a contiguous float64 vector (np.sum / np.mean numerator) */
double orc_np_sum(const double *a, long n);

/* core/multigrid_solve.py:148-208  mg_smooth (in place).  returns 0, or -1 if omega invalid */
int orc_mg_smooth(double *psi, const double *source, const double *r_grid, int nz, int nr,
                  double dr, double dz, double omega, int n_sweeps);

/* core/multigrid_solve.py:211-249  mg_residual  (out: zero ring) */
void orc_mg_residual(const double *psi, const double *source, const double *r_grid, int nz,
                     int nr, double dr, double dz, double *out);

/* core/multigrid_solve.py:338-349  residual_linf */
double orc_residual_linf(const double *psi, const double *source, const double *r_grid, int nz,
                         int nr, double dr, double dz);

/* core/fusion_kernel_numerics.py:28-37  stable_rms over a contiguous vector */
double orc_stable_rms(const double *v, long n);

/* core/fusion_kernel_solver_runtime.py:45-51  compute_gs_residual_rms (interior RMS) */
double orc_gs_residual_rms(const double *psi, const double *source, const double *r_grid, int nz,
                           int nr, double dr, double dz);

/* core/multigrid_solve.py:57-99  restrict_full_weight; coarse is ((nz+1)/2, (nr+1)/2) */
void orc_restrict_full_weight(const double *fine, int nz_f, int nr_f, double *coarse);

/* core/multigrid_solve.py:102-145  prolongate_bilinear; requires nz_c==(nz_f+1)/2 etc.
 * returns 0 or -1 on inconsistent sizes */
int orc_prolongate_bilinear(const double *coarse, int nz_c, int nr_c, int nz_f, int nr_f,
                            double *fine);

/* core/multigrid_solve.py:252-335  multigrid_vcycle; psi is replaced by the improved estimate */
int orc_multigrid_vcycle(double *psi, const double *source, const double *r_grid, int nz, int nr,
                         double dr, double dz, double omega, int pre_smooth, int post_smooth,
                         int min_grid);

/* core/multigrid_solve.py:352-463  multigrid_solve.  psi: in = psi_bc, out = solution.
 * returns 0 ok, -1 bad tol / max_cycles / omega */
int orc_multigrid_solve(const double *source, double *psi, double r_min, double r_max,
                        double z_min, double z_max, int nr, int nz, double tol, int max_cycles,
                        double omega, int pre_smooth, int post_smooth, int min_grid,
                        double *residual_out, int *n_cycles_out, int *converged_out);

/* core/fusion_kernel_iterative_solver.py:54-95 _jacobi_step (out != psi) */
void orc_jacobi_step(const double *psi, const double *source, const double *r_grid, int nz, int nr,
                     double dr, double dz, double *out);

/* core/fusion_kernel_iterative_solver.py:97-161 _sor_step (one red-black sweep, with sanitize
 * and clip; out != psi) */
int orc_sor_step(const double *psi, const double *source, const double *r_grid, int nz, int nr,
                 double dr, double dz, double omega, double *out);

/* complete elliptic integrals K(m), E(m), parameter convention m = k^2
 * (scipy.special.ellipk / ellipe, third-party: SciPy 1.15.3 (Cephes ellpk/ellpe);
 * restated here with the arithmetic-geometric-mean algorithm, Abramowitz & Stegun 17.6) */
void orc_ellip_ke(double m, double *K, double *E);

/* core/fusion_kernel.py:218-251 calculate_vacuum_field (cfg mu0, no turns) */
void orc_vacuum_field(const double *r_axis, const double *z_axis, int nz, int nr, int n_coils,
                      const double *coil_r, const double *coil_z, const double *coil_i, double mu0,
                      double *psi_vac);

/* core/fusion_kernel_free_boundary.py:31-55 green_function (SI mu0); returns NaN on bad input */
double orc_green_function(double r_src, double z_src, double r_obs, double z_obs);

/* core/fusion_kernel_free_boundary.py:58-93 compute_external_flux (SI mu0, x turns) */
void orc_external_flux(const double *r_axis, const double *z_axis, int nz, int nr, int n_coils,
                       const double *coil_r, const double *coil_z, const double *coil_i,
                       const double *coil_turns, double *psi_ext);

/* core/fusion_kernel.py:342-355 _find_magnetic_axis */
void orc_find_axis(const double *psi, int nz, int nr, int *iz, int *ir, double *psi_axis);

/* core/fusion_kernel.py:255-340 find_x_point (default path, no saddle detection).
 * z_min_cfg is cfg["dimensions"]["Z_min"].  idx_out = flat index or -1 for the fallback */
void orc_find_x_point(const double *psi, const double *z_axis, int nz, int nr, double dr, double dz,
                      double z_min_cfg, long *idx_out, double *psi_x);

/* core/fusion_kernel.py:284-329 find_x_point with solver.xpoint_use_saddle_detection = True */
void orc_find_x_point_saddle(const double *psi, const double *z_axis, int nz, int nr, double dr, double dz,
                             double z_min_cfg, long *idx_out, double *psi_x);

/* core/fusion_kernel.py:359-444 update_plasma_source_nonlinear; hmode!=0 selects mtanh.
 * ped_p / ped_ff = {ped_top, ped_width, ped_height, core_alpha}.  Writes J_phi; returns
 * I_current (before renormalisation) */
double orc_update_source(const double *psi, const double *r_axis, int nz, int nr, double dr,
                         double dz, double psi_axis, double psi_bnd, double mu0, double i_target,
                         int hmode, const double *ped_p, const double *ped_ff, double *j_phi);

typedef struct orc_picard_params {
    int nr, nz;
    double r_min, r_max, z_min, z_max; /* cfg["dimensions"] */
    double mu0;                        /* physics.vacuum_permeability */
    double i_target;                   /* physics.plasma_current_target */
    int method;                        /* 0 multigrid, 1 sor, 2 jacobi, 3 anderson (sor + Anderson mixing) */
    double sor_omega;                  /* solver.sor_omega (default 1.6) */
    double alpha;                      /* solver.relaxation_factor (default 0.1) */
    double tol;                        /* solver.convergence_threshold */
    int max_iter;                      /* solver.max_iterations */
    int require_gs_residual;
    double gs_tol;
    int hmode;
    double ped_p[4], ped_ff[4];
    int preserve_initial_state;
    int has_boundary_flux;
    int external_profile_mode; /* skip source update, keep seeded J */
    int n_coils;               /* cfg["coils"] used for calculate_vacuum_field */
    const double *coil_r, *coil_z, *coil_i;
    int anderson_m; /* solver.anderson_depth (default 5) */
    int xpoint_saddle; /* solver.xpoint_use_saddle_detection (default 0) */
} orc_picard_params;

typedef struct orc_picard_result {
    int converged, iterations, diverged;
    double residual;      /* best diff */
    double gs_residual;   /* last */
    double gs_residual_best;
    double psi_axis, psi_bnd; /* of the last iteration's topology step */
    long x_idx;
} orc_picard_result;

/* core/fusion_kernel_newton_solver.py:390-615 solve_equilibrium (Picard path).
 * psi: in (used when preserve_initial_state) / out.  boundary_flux may be NULL.
 * j_phi out.  trace (may be NULL): max_iter rows of 6 doubles
 *   {psi_axis, psi_bnd, x_idx, I_current, diff, gs_rms}.
 * max_iter_budget <0: use params->max_iter; otherwise stop after that many iterations
 * (used by bench.py's bounded cpu_baseline sample). */
int orc_picard_solve(const orc_picard_params *p, double *psi, const double *boundary_flux,
                     double *j_phi, orc_picard_result *res, double *trace, int max_iter_budget);

/* core/fusion_kernel_free_boundary.py:623-739 solve_free_boundary with optimize_shape=False.
 * psi in/out (kernel.Psi state).  Returns outer iterations; final_diff out.
 * picard_iters (may be NULL): per-outer Picard iteration counts (max_outer entries). */
int orc_free_boundary_solve(const orc_picard_params *p, double *psi, double *j_phi, int n_fb_coils,
                            const double *fb_r, const double *fb_z, const double *fb_i,
                            const double *fb_turns, int max_outer, double tol, double *final_diff,
                            int *picard_iters, orc_picard_result *last);

/* core/jax_gs_solver.py:126-167 gs_solve_np (polyglot Picard + damped Jacobi CPU lane) */
void orc_gs_solve_np(double r_min, double r_max, double z_min, double z_max, int nr, int nz,
                     double ip_target, double mu0, int n_picard, int n_jacobi, double alpha,
                     double omega_j, double beta_mix, double *psi_out);
void orc_np_delta_star(const double *psi, int nz, int nr, double r_min, double r_max, double z_min,
                       double z_max, double *out);
void orc_np_equation_residual(const double *psi, int nz, int nr, double r_min, double r_max, double z_min,
                              double z_max, double ip_target, double mu0, double beta_mix, double *out3);

/* hpc/solver.cpp:63-199 FastSolver semantics (legacy C ABI): `iterations` sweeps at omega on the
 * persistent psi state with constant Dirichlet ring; source = -1.0*R*J; returns last max|dpsi| */
double orc_fastsolver_sweeps(double *psi_state, const double *j, int nr, int nz, double r_min,
                             double r_max, double z_min, double z_max, double boundary_value,
                             double omega, int iterations, double tol, int *performed);

/* ---- Newton-Kantorovich + GMRES ---------------------------------------------
 * core/fusion_kernel_newton_solver.py:104-372 _newton_solve_dispatch; helpers
 * core/fusion_kernel_solver_runtime.py:25-68 (residual / operator), :71-136 (profile Jacobian),
 * :206-315 (linear solve).  The linear solve is scipy.sparse.linalg.gmres -- third-party,
 * SciPy 1.15.3 (scipy/sparse/linalg/_isolve/iterative.py, pure-Python restarted GMRES with
 * modified Gram-Schmidt + Givens rotations from LAPACK dlartg, LAPACK 3.11 la_lartg.f90 as
 * bundled by the OpenBLAS 0.3.28 wheel) -- restated here from its published algorithm with
 * restart=50, maxiter=100, atol=1e-8, rtol=1e-6.  Dot products / norms go through BLAS in the
 * reference, so parity is to floating-point tolerance, not bit-exact.  The "ilu" preconditioner
 * (SuperLU spilu) is not restated. */
typedef struct orc_newton_opts {
    int precond;        /* 0 none, 1 diagonal (solver.gmres_preconditioner) */
    int line_search;    /* solver.newton_line_search */
    double line_search_c;
    int max_backtracks;
} orc_newton_opts;

typedef struct orc_newton_result {
    int converged, iterations, diverged;
    double residual;          /* residual_history[-1] (inf when empty) */
    double gs_residual, gs_residual_best;
    int ls_attempts, ls_accepts, ls_rejects;
    int gmres_nonconverged, gmres_breakdown, gmres_last_info;
    int n_hist, n_gs_hist;    /* entries written to the history buffers */
    long gmres_matvecs;       /* total inner GMRES iterations (telemetry for the bench) */
} orc_newton_result;

/* LAPACK 3.11 dlartg (plane rotation), exposed for the pin test against scipy.linalg.lapack.dlartg */
void orc_dlartg(double f, double g, double *c, double *s, double *r);

/* restarted GMRES exactly as SciPy 1.15.3 runs it for the Newton step.  op: y = L*x - diag*x on the
 * interior of an (nz, nr) grid (vectors have (nz-2)*(nr-2) entries, row-major).  inv_diag may be
 * NULL (no preconditioner).  returns info (0 or maxiter); *inner_out = inner iterations done */
int orc_gmres_gs(const double *diag_term, const double *r_grid, int nz, int nr, double dr, double dz,
                 const double *rhs, const double *inv_diag, int restart, int maxiter, double atol,
                 double rtol, double *x_out, long *inner_out);

/* core/fusion_kernel_solver_runtime.py:71-136 compute_profile_jacobian (dJ/dpsi field) */
void orc_profile_jacobian(const double *psi, const double *j_phi, const double *r_axis, int nz, int nr,
                          double dr, double dz, double psi_axis, double psi_bnd, double i_target,
                          int external_profile_mode, double *dj_dpsi);

/* _newton_solve_dispatch.  hist / gs_hist: caller buffers of >= 2*max_iter+2 doubles (may be NULL) */
int orc_newton_solve(const orc_picard_params *p, const orc_newton_opts *o, double *psi,
                     const double *boundary_flux, double *j_phi, orc_newton_result *res, double *hist,
                     double *gs_hist);

/* Multigrid-preconditioned, finite-difference-Jacobian Newton lane of the HIP package (gmres_preconditioner = "multigrid",
 * newton_jacobian = "finite_difference") restated on the CPU: the checker of that lane (tests/test_gpu_newton.py).
 * warmup_steps < 0 -> the reference's rule min(15, max_iter / 2). */
typedef struct orc_newton_mg_opts {
    int warmup_steps;
    double newton_alpha;      /* solver.newton_damping */
    int gmres_restart, gmres_maxiter;
    double gmres_atol, gmres_rtol;
} orc_newton_mg_opts;
int orc_newton_mg_fd(const orc_picard_params *p, const orc_newton_mg_opts *o, double *psi, const double *boundary_flux,
                     double *j_phi, orc_newton_result *res, double *hist, double *gs_hist);

/* ---- field products ----------------------------------------------------------
 * core/fusion_kernel_free_boundary.py:562-580 interp_psi (bilinear sample of psi) */
double orc_interp_psi(const double *psi, const double *r_axis, const double *z_axis, int nz, int nr, double dr,
                      double dz, double r_pt, double z_pt);
/* core/fusion_kernel.py:450-456 compute_b_field (numpy.gradient, edge_order 1) */
void orc_b_field(const double *psi, const double *r_axis, int nz, int nr, double dr, double dz, double *b_r,
                 double *b_z);
/* core/fusion_kernel_free_boundary.py:285-373 build_magnetic_probe_response_matrix; bdir 0 = "R", 1 = "Z";
 * out [(n_flux+n_b)][n_coils] */
void orc_probe_response(int n_coils, const double *coil_r, const double *coil_z, const double *coil_turns, int n_flux,
                        const double *fr, const double *fz, int n_b, const double *br, const double *bz,
                        const int *bdir, double *out);

/* ---- transport coupling --------------------------------------------------------
 * numpy.interp (third-party NumPy 2.2.6, compiled_base.c arr_interp) */
void orc_np_interp(const double *x, long n, const double *xp, const double *fp, int m, double *out);
/* core/integrated_transport_solver_coupling.py:39-73 map_profiles_to_2d */
double orc_map_profiles_to_2d(const double *psi, const double *r_axis, const double *z_axis, int nz, int nr, double dr,
                              double dz, double z_min_cfg, double i_target, const double *rho, const double *pressure_1d,
                              const double *jbs_1d, int n_rho, double *pressure_2d, double *j_phi);
/* LAPACK dgtsv, the routine behind scipy.linalg.solve_banded((1, 1), ...) (third-party SciPy 1.15.3 / LAPACK 3.11) */
int orc_gtsv(int n, double *dl, double *d, double *du, double *b);
/* core/current_diffusion.py:175-207, 210-233, 277-353 */
double orc_neoclassical_resistivity(double te_kev, double ne_19, double z_eff, double epsilon);
void orc_q_from_psi(const double *rho, const double *psi, int n, double r0, double a, double b0, double *q);
int orc_current_diffusion_step(const double *rho, double *psi, int n, double r0, double a, double b0, double dt,
                               const double *te, const double *ne, double z_eff, const double *j_bs, const double *j_cd,
                               const double *j_ext);

/* core/current_diffusion.py:141-158 time recurrence of solve_flux_evolution_nonadiabatic (rate = 1/tau_psi, source =
 * R_null*E_theta - eta*J_theta sampled at the n_steps+1 time points) */
void orc_flux_evolution(int n, int n_steps, double dt, const double *rate, const double *source, double *psi, double *inc,
                        double *drop, double *resid);

#ifdef __cplusplus
}
#endif
#endif

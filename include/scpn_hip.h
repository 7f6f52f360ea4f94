/*
 * scpn_hip.h -- C ABI of libscpn_hip.so, the MI355X (gfx950) Grad-Shafranov hot path.
 *
 * Plain C: pointers and sizes only, no torch / no C++ types.  Every field is C-contiguous
 * float64 of shape (nz, nr), index [iz*nr + ir] (Z slow, R fast) -- the reference layout
 * (core/fusion_kernel.py:163-174).  "batch" functions take `batch` such fields back to back.
 * All host pointers are borrowed for the duration of the call; every call is synchronous
 * (returns after results are copied out).  A handle must not be shared between threads;
 * distinct handles are independent.
 *
 * Three groups:
 *   1. the six legacy symbols of the reference's libscpn_solver.so (hpc/solver.cpp:200-335,
 *      bound by hpc/hpc_bridge.py:190-250) with identical semantics;
 *   2. the elliptic operator library: device counterparts of the free functions of
 *      core/multigrid_solve.py:57-463 (what _multi_compat.dispatch("multigrid_solve") /
 *      ("gs_rb_sor_smooth") providers call, core/_multi_compat_providers.py:322-375,720-792);
 *   3. the resident ensemble solver: Picard / free-boundary solve of `batch` independent
 *      equilibria entirely on the device (core/fusion_kernel_newton_solver.py:390-615,
 *      core/fusion_kernel_free_boundary.py:623-739).
 *
 * Return codes (groups 2, 3): 0 ok; <0 error, see SCPN_E_*.  scpn_last_error() gives text.
 */
#ifndef SCPN_HIP_H
#define SCPN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SCPN_OK 0
#define SCPN_E_ARG (-1)     /* invalid argument (the reference raises ValueError) */
#define SCPN_E_DEVICE (-2)  /* no usable HIP device / HIP runtime error */
#define SCPN_E_DIVERGED (-3)

/* ------------------------------------------------------------------------------------------
 * 1. Legacy libscpn_solver ABI  (replaces hpc/solver.cpp:205-333)
 * ---------------------------------------------------------------------------------------- */
/* solver.cpp:205-222: NULL when nr<2 || nz<2 || rmin>=rmax || zmin>=zmax (or no device) */
void *create_solver(int nr, int nz, double rmin, double rmax, double zmin, double zmax);
/* solver.cpp:227-234: constant Dirichlet value forced on all four edges every sweep */
void set_boundary_dirichlet(void *solver, double boundary_value);
/* solver.cpp:243-268: input is J_phi; source = -1.0*R*J; `iterations` (>=1) red-black SOR sweeps
 * at omega=1.8 on the psi state kept inside the handle; silent no-op on NULL / size mismatch */
void run_step(void *solver, const double *j_array, double *psi_array, int size, int iterations);
/* solver.cpp:273-321: sweeps until max|dpsi| <= tolerance; omega clamped to [0.1,1.99];
 * returns sweeps executed (0 on invalid input) */
int run_step_converged(void *solver, const double *j_array, double *psi_array, int size,
                       int max_iterations, double omega, double tolerance, double *final_delta_out);
/* solver.cpp:325-333 */
void destroy_solver(void *solver);
void delete_solver(void *solver);

/* ------------------------------------------------------------------------------------------
 * Device probes (replace scpn_fusion_rs.py_gpu_available / py_gpu_info,
 * scpn-fusion-rs/crates/fusion-python/src/bindings/gpu.rs:82-90)
 * ---------------------------------------------------------------------------------------- */
int scpn_device_count(void);                /* number of HIP devices, 0 if none / no runtime */
int scpn_device_info(char *buf, int buflen); /* "name | gfx arch | CUs | HBM GiB"; SCPN_E_DEVICE if none */
int scpn_set_device(int device);            /* select the device used by subsequently created handles */
const char *scpn_last_error(void);
const char *scpn_version(void);

/* ------------------------------------------------------------------------------------------
 * 2. Elliptic operator library (host buffers in, host buffers out; `batch` fields per call)
 *    r_axis[nr] is the 1-D R coordinate of the columns (the reference passes the 2-D meshgrid
 *    of it; only R(ir) enters the stencil).
 * ---------------------------------------------------------------------------------------- */
/* core/multigrid_solve.py:148-208 mg_smooth: n_sweeps red-black SOR sweeps, in place.
 * SCPN_E_ARG unless 1 <= omega < 2 (multigrid_solve.py:33-54) */
int scpn_mg_smooth(double *psi, const double *source, const double *r_axis, int nz, int nr,
                   int batch, double dr, double dz, double omega, int n_sweeps);
/* core/multigrid_solve.py:211-249 mg_residual: out = L*psi - source, zero ring */
int scpn_mg_residual(const double *psi, const double *source, const double *r_axis, int nz, int nr,
                     int batch, double dr, double dz, double *out);
/* core/multigrid_solve.py:338-349 residual_linf -> linf[batch] */
int scpn_residual_linf(const double *psi, const double *source, const double *r_axis, int nz,
                       int nr, int batch, double dr, double dz, double *linf);
/* core/fusion_kernel_solver_runtime.py:45-51 compute_gs_residual_rms -> rms[batch] */
int scpn_gs_residual_rms(const double *psi, const double *source, const double *r_axis, int nz,
                         int nr, int batch, double dr, double dz, double *rms);
/* core/multigrid_solve.py:57-99 restrict_full_weight: coarse is ((nz_f+1)/2, (nr_f+1)/2) */
int scpn_restrict_full_weight(const double *fine, int nz_f, int nr_f, int batch, double *coarse);
/* core/multigrid_solve.py:102-145 prolongate_bilinear (nz_c == (nz_f+1)/2 and nr_c == (nr_f+1)/2) */
int scpn_prolongate_bilinear(const double *coarse, int nz_c, int nr_c, int nz_f, int nr_f,
                             int batch, double *fine);
/* core/multigrid_solve.py:252-335 multigrid_vcycle: psi replaced by the improved estimate */
int scpn_multigrid_vcycle(double *psi, const double *source, const double *r_axis, int nz, int nr,
                          int batch, double dr, double dz, double omega, int pre_smooth,
                          int post_smooth, int min_grid);
/* core/multigrid_solve.py:352-463 multigrid_solve.  psi: in = psi_bc (Dirichlet ring + initial
 * interior), out = solution.  r_axis/z_axis = numpy.linspace of the extent (host side).
 * residual/n_cycles/converged: [batch].  SCPN_E_ARG on tol<=0 / non-finite, max_cycles<1, omega */
int scpn_multigrid_solve(const double *source, double *psi, const double *r_axis,
                         const double *z_axis, int nr, int nz, int batch, double tol, int max_cycles,
                         double omega, int pre_smooth, int post_smooth, int min_grid,
                         double *residual, int *n_cycles, int *converged);
/* core/fusion_kernel_iterative_solver.py:54-95 _jacobi_step / :97-161 _sor_step (sanitize+clip) */
int scpn_jacobi_step(const double *psi, const double *source, const double *r_axis, int nz, int nr,
                     int batch, double dr, double dz, double *out);
int scpn_sor_step(const double *psi, const double *source, const double *r_axis, int nz, int nr,
                  int batch, double dr, double dz, double omega, double *out);
/* Coil Green's-function flux on the grid, sum over coils of  scale*I_c*turns_c*G(coil; R,Z).
 * mode 0: core/fusion_kernel.py:218-251 calculate_vacuum_field (mu0 = `mu0`, turns ignored)
 * mode 1: core/fusion_kernel_free_boundary.py:58-93 compute_external_flux (SI mu0, x turns,
 *         self-point -> 0).  coil arrays are [batch][n_coils]; out is [batch][nz][nr]. */
int scpn_coil_flux(const double *r_axis, const double *z_axis, int nz, int nr, int batch,
                   int n_coils, const double *coil_r, const double *coil_z, const double *coil_i,
                   const double *coil_turns, double mu0, int mode, double *out);
/* core/fusion_kernel_free_boundary.py:31-55,137-153 green_function at arbitrary points (SI mu0):
 * out[c*n_pts + p] = G(coil c -> point p); NaN where the reference raises ValueError */
int scpn_green_points(int n_coils, const double *coil_r, const double *coil_z, int n_pts,
                      const double *pts_r, const double *pts_z, double *out);
/* core/fusion_kernel_free_boundary.py:285-373 build_magnetic_probe_response_matrix: out[(n_flux+n_b)][n_coils];
 * flux-loop rows turns*G, B-probe rows by central differences of the same kernel; b_dir[i] 0 = "R", 1 = "Z";
 * coil_turns may be NULL (all 1) */
int scpn_probe_response(int n_coils, const double *coil_r, const double *coil_z, const double *coil_turns,
                        int n_flux, const double *flux_r, const double *flux_z, int n_b, const double *b_r,
                        const double *b_z, const int *b_dir, double *out);
/* Verification hook: the two division-by-reciprocal sequences of the kernels (5 and 4 instructions, see csrc/gs_kernels.hip.h
 * divc / divc4) applied to caller-supplied x[i] / d[i]; both must equal IEEE division (NumPy's `/`) bit for bit. */
int scpn_verify_division(const double *x, const double *d, int n, double *q5, double *q4);
/* Measurement hook (bench.py, SURVEY 8d "device-copy bandwidth probe"): GB/s (read + written bytes) of a 16-B/lane device
 * copy of `bytes` bytes, HIP-event timed over `reps` launches -- the measured HBM denominator printed next to the nominal one. */
int scpn_copy_bandwidth_probe(long bytes, int reps, double *gb_per_s);
/* the same with a mode: 0 = grid-stride copy, 1 = read-only sum, 2 = copy in contiguous 32-KiB chunks per workgroup with eight
 * 16-B loads in flight per lane, 3 = mode 2 with non-temporal stores, 4 = hipMemcpyDtoDAsync, 5 = three read streams + one write stream; all moved bytes count;
 * best over several grid sizes */
int scpn_bandwidth_probe(int mode, long bytes, int reps, double *gb_per_s);
/* polyglot lane diagnostics, core/jax_gs_solver.py:196-316: gs_delta_star_np (zero ring; out [nz][nr]) and
 * gs_equation_residual_np (out3 = abs_max, relative_max, relative_l2 of the discrete equation imbalance with the lane's own
 * source normalisation) */
int scpn_np_delta_star(const double *psi, int nz, int nr, const double *r_axis, double dr, double dz, double *out);
int scpn_np_equation_residual(const double *psi, int nz, int nr, const double *r_axis, const double *z_axis,
                              double ip_target, double mu0, double beta_mix, double *out3);
/* ---- transport coupling ----
 * scipy.linalg.solve_banded((1, 1), ab, b) of `batch` independent tridiagonal systems, i.e. LAPACK dgtsv (partial pivoting,
 * bit-identical operation order): dl/du [batch][n-1], d/b/x [batch][n]; info[batch] (0, or i>0: U(i,i) == 0) may be NULL */
int scpn_tridiag_solve(int batch, int n, const double *dl, const double *d, const double *du, const double *b,
                       double *x, int *info);
/* core/current_diffusion.py:277-353 CurrentDiffusionSolver.step for `batch` profiles on one rho grid: psi [batch][n] in/out,
 * te/ne/j_* [batch][n], j_ext may be NULL */
int scpn_current_diffusion_step(int batch, int n, const double *rho, double *psi, double r0, double a, double b0,
                                double dt, const double *te, const double *ne, double z_eff, const double *j_bs,
                                const double *j_cd, const double *j_ext, int *info);
/* core/current_diffusion.py:141-158: the time recurrence of solve_flux_evolution_nonadiabatic for `batch` profiles.  rate =
 * 1/tau_psi and source = R_null*E_theta - eta*J_theta sampled at the n_steps+1 time points ([batch][n_steps+1][n]); psi0
 * [batch][n]; psi out [batch][n_steps+1][n]; inc / drop / resid out [batch][n_steps][n] (may be NULL) */
int scpn_flux_evolution(int batch, int n, int n_steps, double dt, const double *rate, const double *source,
                        const double *psi0, double *psi, double *inc, double *drop, double *resid);
/* core/jax_gs_solver.py:126-167 gs_solve_np (polyglot Picard + damped-Jacobi lane), batch=1 */
int scpn_gs_solve_np(double r_min, double r_max, double z_min, double z_max, int nr, int nz,
                     const double *r_axis, const double *z_axis, const double *psi_seed,
                     double ip_target, double mu0, int n_picard, int n_jacobi, double alpha,
                     double omega_j, double beta_mix, double *psi_out);

/* ------------------------------------------------------------------------------------------
 * 3. Resident ensemble solver
 * ---------------------------------------------------------------------------------------- */
typedef struct scpn_gs scpn_gs;

typedef struct scpn_picard_params {
    double z_min_cfg;    /* cfg["dimensions"]["Z_min"]  (X-point mask, core/fusion_kernel.py:279) */
    double r_center;     /* (R_min + R_max)/2 of the config (seed, iterative_solver.py:400) */
    double mu0;          /* physics.vacuum_permeability */
    int method;          /* 0 multigrid (one V-cycle / iteration), 1 sor, 2 jacobi, 3 anderson (sor + Anderson mixing) */
    double sor_omega;    /* solver.sor_omega */
    double alpha;        /* solver.relaxation_factor */
    double tol;          /* solver.convergence_threshold */
    int max_iter;        /* solver.max_iterations */
    int require_gs_residual;
    double gs_tol;
    int hmode;           /* physics.profiles.mode == h-mode */
    double ped_p[4];     /* ped_top, ped_width, ped_height, core_alpha */
    double ped_ff[4];
    int preserve_initial_state;
    int has_boundary_flux; /* use the field set by scpn_gs_set_boundary_flux as Dirichlet map */
    int external_profile_mode;
    int check_every;     /* host polls the device convergence word every this many iterations (>=1).  ONLY used by the blocking-poll
                          * loop (environment SCPN_HIP_RUNAHEAD=0, or no host-mapped memory): the default run-ahead loop follows the device
                          * through two host-mapped words {solve id, launches started} / {solve id, done} every iteration and ignores it.
                          * Measured gain of the run-ahead loop on the 32-member bench: +0.1 ... +0.3 %, i.e. inside box noise
                          * (profiles/r05_ab_runahead_bench.txt); it matters for short solves of few members */
    int fail_on_diverge;
    int skip_seed;       /* 1: do not run _seed_plasma (not a reference behaviour; benchmarking aid) */
    int anderson_m;      /* solver.anderson_depth (default 5, at most 5) */
    int xpoint_saddle;   /* solver.xpoint_use_saddle_detection (core/fusion_kernel.py:284-329), default 0 */
} scpn_picard_params;

typedef struct scpn_picard_result {
    int converged;
    int iterations;
    int diverged;
    int reserved;
    double residual;         /* best mean|dpsi| */
    double gs_residual;      /* last GS RMS */
    double gs_residual_best;
    double psi_axis, psi_bnd; /* last topology step */
    double x_r, x_z;          /* last X-point position */
    double i_current;         /* last unnormalised current integral */
} scpn_picard_result;

/* grid: r_axis[nr], z_axis[nz] (numpy.linspace of the config extent), dr = R[1]-R[0], dz likewise */
int scpn_gs_create(scpn_gs **out, int nr, int nz, int batch, const double *r_axis,
                   const double *z_axis, double dr, double dz);
void scpn_gs_destroy(scpn_gs *h);
/* member = -1 addresses all members (buffer then holds `batch` fields) */
int scpn_gs_set_psi(scpn_gs *h, int member, const double *psi);
int scpn_gs_get_psi(scpn_gs *h, int member, double *psi);
int scpn_gs_get_jphi(scpn_gs *h, int member, double *jphi);
int scpn_gs_set_jphi(scpn_gs *h, int member, const double *jphi);
int scpn_gs_get_source(scpn_gs *h, int member, double *source);
int scpn_gs_set_boundary_flux(scpn_gs *h, int member, const double *psi_bc);
int scpn_gs_get_boundary_flux(scpn_gs *h, int member, double *psi_bc);
/* per-member plasma current target (physics.plasma_current_target); member=-1: i_target[batch] */
int scpn_gs_set_current_target(scpn_gs *h, int member, const double *i_target);
/* cfg["coils"] of each member for calculate_vacuum_field: arrays [batch][n_coils] */
int scpn_gs_set_vacuum_coils(scpn_gs *h, int n_coils, const double *coil_r, const double *coil_z,
                             const double *coil_i);
/* optional host-computed Gaussian seed J (shape (nz,nr), before normalisation); NULL = computed
 * inside with libm exp (core/fusion_kernel_iterative_solver.py:400-402) */
int scpn_gs_set_seed(scpn_gs *h, const double *seed_j);

/* core/fusion_kernel_newton_solver.py:390-615 on all members; results[batch].
 * histories (may be NULL): [batch][max_iter] residual_history / gs_residual_history,
 * trace (may be NULL): [batch][max_iter][6] = psi_axis, psi_bnd, x_idx, I_current, diff, gs */
int scpn_gs_picard_solve(scpn_gs *h, const scpn_picard_params *p, scpn_picard_result *results,
                         double *residual_history, double *gs_history, double *trace);

/* ---- Newton-Kantorovich + GMRES ------------------------------------------------
 * core/fusion_kernel_newton_solver.py:104-372 _newton_solve_dispatch on all members:
 * Picard warm-up (min(15, max_iter/2) SOR-step iterations), then Newton steps, each solving
 * (L - diag) delta = -r with restarted GMRES exactly as scipy.sparse.linalg.gmres (SciPy 1.15.3) runs
 * for core/fusion_kernel_solver_runtime.py:206-315 (restart 50, maxiter 100, atol 1e-8, rtol 1e-6;
 * preconditioner none | diagonal; "ilu": no incomplete LU on the device -- the host mirror takes the reference's
 * own fallback to diagonal, solver_runtime.py:289-293), psi += 0.5*delta or the Armijo backtracking line search
 * (:246-294).  p->method is ignored (the warm-up is the SOR branch).
 * Extension lane of this library (opt-in): precond = 2 (one V-cycle(3,3) as left preconditioner;
 * the first Krylov vector is the preconditioned right-hand side, a solve capped at gmres_restart x gmres_maxiter columns
 * exits its last cycle on the Arnoldi estimate), reserved = 1 (finite-difference Jacobian action), newton_alpha = 1. */
typedef struct scpn_newton_opts {
    int precond;            /* 0 none, 1 diagonal (solver.gmres_preconditioner), 2 one geometric V-cycle of the GS* operator
                             * (extension: the reference's JAX lane has this preconditioner, core/jax_multigrid_precond.py:208-276) */
    int line_search;        /* solver.newton_line_search */
    double line_search_c;   /* solver.newton_line_search_c (1e-4) */
    int max_backtracks;     /* solver.newton_line_search_max_backtracks (6) */
    int warmup_steps;       /* < 0: min(15, max_iter/2) as the reference */
    double newton_alpha;    /* 0.5 */
    int gmres_restart;      /* 50 (at most 50) */
    int gmres_maxiter;      /* 100 restart cycles */
    double gmres_atol;      /* 1e-8 */
    double gmres_rtol;      /* 1e-6 */
    int poll_every;         /* host polls the device "inner loop finished" word every this many columns (>=1) */
    int reserved;           /* Jacobian action: 0 = the reference's profile Jacobian (diagonal dS/dpsi term); 1 = finite difference of the
                             * full source map (axis / X-point / I_p renormalisation included) -- extension, use with newton_alpha 1 */
} scpn_newton_opts;

typedef struct scpn_newton_result {
    int converged;
    int iterations;
    int diverged;
    int n_hist;               /* entries of residual_history */
    int n_gs_hist;            /* entries of gs_residual_history */
    int ls_attempts, ls_accepts, ls_rejects;
    int gmres_nonconverged, gmres_breakdown, gmres_last_info;
    int reserved;
    long gmres_inner;         /* total GMRES inner iterations (operator applications) */
    double residual;          /* residual_history[-1], +inf when empty */
    double gs_residual;       /* gs_residual_history[-1] */
    double gs_residual_best;
} scpn_newton_result;

/* histories (may be NULL): [batch][max_iter + 2], entry counts in the results */
int scpn_gs_newton_solve(scpn_gs *h, const scpn_picard_params *p, const scpn_newton_opts *o,
                         scpn_newton_result *results, double *residual_history, double *gs_history);

/* core/fusion_kernel_free_boundary.py:623-739 with optimize_shape=False: coil arrays [batch][n];
 * outer_iterations[batch], final_diff[batch], picard_iters [batch][max_outer] (may be NULL) */
int scpn_gs_free_boundary_solve(scpn_gs *h, const scpn_picard_params *p, int n_coils,
                                const double *coil_r, const double *coil_z, const double *coil_i,
                                const double *coil_turns, int max_outer, double tol,
                                int *outer_iterations, double *final_diff, int *picard_iters,
                                scpn_picard_result *results);
/* the same outer loop with the Newton-Kantorovich solve as the inner solver (solver_method = "newton" inside
 * FusionKernel.solve_free_boundary, core/fusion_kernel_free_boundary.py:659-662 -> newton_solver.py:435-439); `inner_iters`
 * receives the inner solver's iteration count (warm-up + Newton steps) per outer iteration */
int scpn_gs_free_boundary_solve_newton(scpn_gs *h, const scpn_picard_params *p, const scpn_newton_opts *o, int n_coils,
                                       const double *coil_r, const double *coil_z, const double *coil_i, const double *coil_turns,
                                       int max_outer, double tol, int *outer_iterations, double *final_diff, int *inner_iters,
                                       scpn_picard_result *results);

/* resident smoother / V-cycle on the handle state (psi, source): the zero-copy path the
 * smoother benchmark's roofline figure is taken on */
int scpn_gs_set_source(scpn_gs *h, int member, const double *source);
int scpn_gs_rb_sor(scpn_gs *h, double omega, int n_sweeps);
int scpn_gs_vcycle(scpn_gs *h, double omega, int pre_smooth, int post_smooth, int min_grid);
/* post-solve products core/fusion_kernel.py:450-456 compute_b_field; member = -1: all members, [batch][nz][nr] */
int scpn_gs_compute_b_field(scpn_gs *h, int member, double *b_r, double *b_z);
/* core/integrated_transport_solver_coupling.py:39-73 map_profiles_to_2d for every member: from the resident psi and the 1-D
 * profiles rho[n_rho] (shared), pressure_1d = ne*(Ti+Te) and the bootstrap profile jbs_1d ([batch][n_rho]) to the device
 * J_phi (read it with scpn_gs_get_jphi; a following Picard solve in external_profile_mode consumes it in place).
 * z_min_cfg as in scpn_picard_params.  pressure_2d_out [batch][nz][nr] and i_current_out [batch] may be NULL */
int scpn_gs_map_profiles(scpn_gs *h, double z_min_cfg, int n_rho, const double *rho, const double *pressure_1d,
                         const double *jbs_1d, double *pressure_2d_out, double *i_current_out);
/* core/fusion_kernel_free_boundary.py:156-159 sample_flux_at_points / :562-580 interp_psi of every member's psi at
 * n_pts (R, Z) points; out [batch][n_pts] */
int scpn_gs_sample_psi(scpn_gs *h, int n_pts, const double *pts_r, const double *pts_z, double *out);

/* psi <- value on every member, on the device (ensemble re-runs without a host upload) */
/* Relative flux change of the transport coupling (core/integrated_transport_solver_coupling.py:131-146: psi_old = Psi.copy() ...
 * norm(Psi - psi_old) / norm(psi_old), norm(psi_old) < 1e-30 -> 1) without a download of the fields: scpn_gs_snapshot_psi keeps a
 * device copy of every member's current iterate, scpn_gs_flux_change returns the ratio per member (rel_change: batch doubles)
 * for the iterate the handle holds then. */
int scpn_gs_snapshot_psi(scpn_gs *h);
int scpn_gs_flux_change(scpn_gs *h, double *rel_change);
int scpn_gs_fill_psi(scpn_gs *h, double value);
/* pack the current psi of all members into one dense device buffer [batch][nz][nr] owned by the
 * handle and return its DEVICE address (valid until the next pack / destroy): the send buffer of the
 * RCCL ensemble gather */
int scpn_gs_pack_psi_device(scpn_gs *h, void **device_ptr);

/* timing hooks for bench.py: average device time (ms) of the named kernel family since the last
 * reset, measured with hipEvents on the handle's stream when profiling is enabled */
int scpn_gs_profile(scpn_gs *h, int enable);
int scpn_gs_profile_read(scpn_gs *h, const char *name, double *total_ms, long *launches);
int scpn_gs_profile_names(scpn_gs *h, char *buf, int buflen); /* comma-separated family names seen so far */
int scpn_gs_sync(scpn_gs *h);
/* the ensemble is split into this many independent groups (own HIP stream + host thread each; env
 * SCPN_HIP_GROUPS overrides the default of 2 for batch >= 16) so latency-bound coarse-level kernels of one
 * group overlap bandwidth-bound fine-level kernels of another */
int scpn_gs_num_groups(scpn_gs *h);
/* Arithmetic of the handle's hot kernels (tile smoother, in-LDS coarse V-cycle, fused Picard tail):
 *   0 = exact (default): every expression keeps the operation order of the reference NumPy code
 *       (core/multigrid_solve.py:184-206,211-249) -- the linear operators are bit-identical to it;
 *   1 = contracted: the same red-black Gauss-Seidel / V-cycle / Picard iteration with fused multiply-adds and
 *       reciprocal multiplies (the update is one chain of four FMAs) -- results agree with mode 0 to rounding
 *       (<= 1e-10 max|psi| on converged equilibria, asserted by tests/test_gpu_fast_arith.py against the CPU oracle).
 * The default of new handles comes from the environment variable SCPN_HIP_ARITH (exact | contracted).  The stateless
 * operator API (section 2), the dispatcher providers and the legacy ABI (section 1) are always exact.
 * Returns SCPN_E_ARG for another mode. */
/* The stateless operators (section 2) keep the solver objects of the last few grids in a pool (as the reference's GPU provider caches
 * its PyGpuSolver per grid, core/_multi_compat_providers.py:720,755-792); SCPN_HIP_OPCACHE=0 turns it off.  Counters for tests / tools. */
int scpn_operator_pool_stats(long *hits, long *misses, int *idle);
int scpn_gs_set_arith(scpn_gs *h, int mode);
int scpn_gs_get_arith(scpn_gs *h);

/* ------------------------------------------------------------------------------------------
 * 4. Ensemble gather over RCCL / xGMI (one process per GPU).  The path shards across members, so the only
 *    collective is the all-gather of the dense psi blocks; librccl is dlopen'ed at the first call.  Rendezvous: rank 0 calls
 *    scpn_comm_unique_id and the host passes the 128 bytes to the other ranks (scpn_fusion_core_amd/distributed.py: TCP on
 *    MASTER_ADDR:MASTER_PORT), every rank then calls scpn_comm_init.  The reference has no counterpart to replace (its only
 *    multi-device code is the domain-decomposition prototype scpn-fusion-rs/crates/fusion-core/src/mpi_domain.rs:741-770).
 * ---------------------------------------------------------------------------------------- */
#define SCPN_COMM_ID_BYTES 128
int scpn_comm_unique_id(char *id128);
/* the LOCAL part of scpn_comm_init -- librccl loaded, `device` visible and selected, NCCL 2.x ABI -- without entering a
 * collective: the host exchanges the ranks' results of this step and calls scpn_comm_init only if every rank passed, so a
 * rank that fails here cannot leave its peers blocked inside ncclCommInitRank */
int scpn_comm_prepare(int device, int *rccl_version);
int scpn_comm_init(int rank, int world, const char *id128, int device); /* selects `device` for this process; collective */
int scpn_comm_info(int *rank, int *world, int *rccl_version);           /* as RCCL sees them */
/* pack + ncclAllGather of psi: every rank sends a block of `per_rank_members` members (>= its own batch; the rank that owns the
 * short last block of an uneven ensemble is padded with zeros, so all ranks issue the same count), recv_device =
 * [world][per_rank_members][nz][nr] doubles on this GPU; asynchronous on the communicator's stream, ordered behind the solver
 * streams by events (overlaps the next solve) -- scpn_comm_wait() before the buffer is read.  The send block is the
 * communicator's own buffer (not the staging buffer of scpn_gs_pack_psi_device); a second gather waits for the first */
int scpn_comm_allgather_psi(scpn_gs *h, int per_rank_members, void *recv_device);
int scpn_comm_wait(void);
int scpn_comm_allreduce_max(double *value); /* in place, blocking: the max-over-ranks of a timing and the barrier */
int scpn_comm_finalize(void);
int scpn_device_alloc(long bytes, void **ptr);
int scpn_device_free(void *ptr);
int scpn_device_download(void *host, const void *device_ptr, long bytes);

#ifdef __cplusplus
}
#endif
#endif

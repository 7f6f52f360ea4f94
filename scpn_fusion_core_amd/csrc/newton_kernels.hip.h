// newton_kernels.hip.h -- device side of the Newton-Kantorovich + GMRES equilibrium solve.
//
// Reference: core/fusion_kernel_newton_solver.py:104-372 (_newton_solve_dispatch),
// core/fusion_kernel_solver_runtime.py:25-136 (residual / operator / profile Jacobian), :206-315 (linear
// solve = scipy.sparse.linalg.gmres, SciPy 1.15.3: restarted GMRES, modified Gram-Schmidt, Givens rotations
// from LAPACK dlartg).  Everything is batched over ensemble members; each member carries its own GMRES state
// (Hessenberg column, rotations, tolerances, activity flags) in HBM and all members advance in lock-step,
// finished members being masked out.
//
// Layout: Krylov vectors are stored on the padded (nz, ld) grid of the solver with a ZERO ring and ZERO pad
// columns, so a vector of the reference's interior space (nz-2)*(nr-2) is the same numbers plus zeros: dots
// and norms run flat over the member stride (16-B loads), the operator is the plain 5-point stencil.
// Reductions: every block writes a partial; the CONSUMER kernel (the next launch on the stream) has each of its blocks
// add the partials in index order at its start, so a Gram-Schmidt step is one launch, no device-scope fence or atomic
// is needed (on gfx950 an agent-scope release/acquire per block costs an L2 write-back + invalidate: measured 100x
// slower), and results do not depend on block scheduling.  Scalar tails (rotations, tolerance control) run in
// one-block-per-member kernels once per column / cycle.
#pragma once

namespace gs {

constexpr int GM_R = 50;  // restart length used by the reference (fusion_kernel_solver_runtime.py:297-300)

struct GmState {
    double bnrm2, atol, ptol, ptol_max_factor, presid, rnorm, h0, tmp, Mb;
    double S[GM_R + 1], y[GM_R + 1], giv[GM_R * 2], h[GM_R * (GM_R + 1)];
    int inner_on, solve_on, breakdown, info, col_final, inner_total, cycles, restart;
    int pad0, pad1;
};

// per-member Newton state
struct NtMember {
    double res_norm, gs, gs_best, last_hist;
    double jac_c, jac_denom;     // profile-Jacobian scalars (fusion_kernel_solver_runtime.py:119-131)
    double ls_alpha, ls_res0;    // line search
    int on;                      // still iterating in phase B
    int converged, diverged, iterations;
    int n_hist, n_gs_hist;
    int ls_on, ls_accepted, ls_attempts, ls_accepts, ls_rejects;
    int gm_nonconv, gm_breakdown, gm_last_info;
    long gm_inner;
    int pad0, pad1;
};

struct NtCtl {
    double tol, gs_tol;
    int require_gs, line_search, hist_cap, pad;
    long n_interior;
};

// ---------------------------------------------------------------------------------------------------------
// sum of `n` partials in index order by the 256 threads of a block; the result is returned to EVERY thread
__device__ __forceinline__ double sum_partials(const double* part, int n, double* sh) {
    double s = 0.0;
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    for (int i = t; i < n; i += 256) s += part[i];
    s = wave_sum(s);
    __syncthreads();
    if ((t & 63) == 0) sh[t >> 6] = s;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ double block_sum_flat(double v, double* sh) {
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    v = wave_sum(v);
    if ((t & 63) == 0) sh[t >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (t == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;
}

// LAPACK 3.11 la_lartg.f90 (double): plane rotation with r = sign(f)*hypot(f,g)
__device__ __forceinline__ void dlartg(double f, double g, double& c, double& s, double& r) {
    const double safmin = 2.2250738585072014e-308, safmax = 1.0 / safmin;
    const double rtmin = 1.4916681462400413e-154;  // sqrt(safmin)
    const double rtmax = 4.7403759540545887e+153;  // sqrt(safmax/2)
    double f1 = fabs(f), g1 = fabs(g);
    if (g == 0.0) { c = 1.0; s = 0.0; r = f; }
    else if (f == 0.0) { c = 0.0; s = copysign(1.0, g); r = g1; }
    else if (f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax) {
        double d = sqrt(f * f + g * g);
        c = f1 / d;
        r = copysign(d, f);
        s = g / r;
    } else {
        double u = fmin(safmax, fmax(safmin, fmax(f1, g1)));
        double fs = f / u, gs_ = g / u;
        double d = sqrt(fs * fs + gs_ * gs_);
        c = fabs(fs) / d;
        r = copysign(d, f);
        s = gs_ / r;
        r = r * u;
    }
}

// L*v at an interior point (fusion_kernel_solver_runtime.py:54-68 apply_gs_operator); same arithmetic as resid_at
// (c = iz * L.ld + ir, the point's linear index)
__device__ __forceinline__ double lop_at_lin(const double* v, const Lvl& L, long c, int ir) {
    double p = v[c], e = v[c + 1], w = v[c - 1], n = v[c + L.ld], s = v[c - L.ld];
    double d2r = divc(e - 2.0 * p + w, L.dr2, L.inv_dr2);
    double d1r = divc(e - w, L.two_dr, L.inv_two_dr);
    double d2z = divc(n - 2.0 * p + s, L.dz2, L.inv_dz2);
    return d2r - divc(d1r, L.r_safe[ir], L.r_inv[ir]) + d2z;
}
__device__ __forceinline__ double lop_at(const double* v, const Lvl& L, int iz, int ir) { return lop_at_lin(v, L, (long)iz * L.ld + ir, ir); }

// ------------------------------------------------------------------------------------------------ Newton level
// r_k = L psi - Source on the interior (zero ring) + per-block sum r^2.  2-D grid (ceil(nr/64), ceil(nz/4), B), block 64x4.
// which 0: members nt.on            which 1: members nt.ls_on (line-search trial)
__global__ void __launch_bounds__(256) k_nt_resid(Fld outF, Fld psiF, Fld srcF, Lvl L, const NtMember* nt, int which, double* part) {
    __shared__ double sh[4];
    const int b = blockIdx.z;
    if (which == 1 ? !nt[b].ls_on : !nt[b].on) return;
    const int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    double v = 0.0;
    const bool in = ir < L.nr && iz < L.nz;
    if (in && iz > 0 && iz < L.nz - 1 && ir > 0 && ir < L.nr - 1) v = resid_at(psiF.at(b), srcF.at(b), L, iz, ir);
    if (in && outF.base) outF.at(b)[(long)iz * L.ld + ir] = v;
    const int nblk = gridDim.x * gridDim.y;
    double bs = block_sum_flat(v * v, sh);
    if (threadIdx.x == 0 && threadIdx.y == 0) part[(long)b * nblk + blockIdx.y * gridDim.x + blockIdx.x] = bs;
}
// one block per member: finish the norm and act on it
// mode 0: Newton residual of the iteration (history, convergence test, newton_solver.py:197-214)
// mode 1: line-search trial norm (accept test :267-274)      mode 2: post-step norm of the line search (:292-294)
__global__ void __launch_bounds__(256) k_nt_resid_final(Member* st, NtMember* nt, NtCtl ctl, int mode, double ls_c, double* hist,
                                                        double* gs_hist, const double* part, int nblk, int* f_active, int* n_on) {
    __shared__ double sh[4];
    const int b = blockIdx.x;
    if (mode == 1 ? !nt[b].ls_on : !nt[b].on) return;
    const double tot = sum_partials(part + (long)b * nblk, nblk, sh);
    if (threadIdx.x != 0) return;
    NtMember& m = nt[b];
    const double nrm = sqrt(tot);
    if (mode == 0) {
        const double gsr = sqrt(tot / (double)ctl.n_interior);
        m.res_norm = nrm;
        m.gs = gsr;
        if (!ctl.line_search) { if (m.n_hist < ctl.hist_cap) hist[(long)b * ctl.hist_cap + m.n_hist] = nrm; m.n_hist++; m.last_hist = nrm; }
        if (m.n_gs_hist < ctl.hist_cap) gs_hist[(long)b * ctl.hist_cap + m.n_gs_hist] = gsr;
        m.n_gs_hist++;
        if (gsr < m.gs_best) m.gs_best = gsr;
        const bool upd = nrm < ctl.tol;
        const bool gsc = (!ctl.require_gs) || (gsr < ctl.gs_tol);
        if (upd && gsc) {
            if (ctl.line_search) { if (m.n_hist < ctl.hist_cap) hist[(long)b * ctl.hist_cap + m.n_hist] = nrm; m.n_hist++; m.last_hist = nrm; }
            m.converged = 1;
            m.on = 0;
            st[b].active = 0;
            st[b].converged = 1;
            f_active[b] = 0;
            atomicSub(n_on, 1);
        }
    } else if (mode == 1) {
        m.ls_attempts++;
        if (nrm <= (1.0 - ls_c * m.ls_alpha) * m.res_norm) { m.ls_accepted = 1; m.ls_accepts++; m.ls_on = 0; }
        else { m.ls_rejects++; m.ls_alpha *= 0.5; }
    } else {
        if (m.n_hist < ctl.hist_cap) hist[(long)b * ctl.hist_cap + m.n_hist] = nrm;
        m.n_hist++;
        m.last_hist = nrm;
    }
}

__device__ __forceinline__ double jac_denom_of(const Member& m, double& pa_out) {
    double pa = m.psi_axis, pb = m.psi_bnd;  // fusion_kernel_solver_runtime.py:75-86
    if (!finite_d(pa)) pa = 0.0;
    if (pb != pb || pb > 1.79e308) pb = pa + 1e-9; else if (pb < -1.79e308) pb = pa - 1e-9;
    double denom = pb - pa;
    if (!finite_d(denom)) denom = 1e-9;
    if (fabs(denom) < 1e-9) denom = denom >= 0.0 ? 1e-9 : -1e-9;
    pa_out = pa;
    return denom;
}
// partial sums of mask * (1 - psi_N) * R  (fusion_kernel_solver_runtime.py:124); flat over nz*nr, grid (nblk, B)
__global__ void __launch_bounds__(256) k_nt_jac_sum(Fld psiF, const Member* st, const NtMember* nt, int nz, int nr, int ld,
                                                    const double* r_axis, double* part, int nblk) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    if (!nt[b].on) return;
    const double* psi = psiF.at(b);
    double pa;
    const double denom = jac_denom_of(st[b], pa);
    const long n = (long)nz * nr;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (q < n) {
            int iz = (int)(q / nr), ir = (int)(q - (long)iz * nr);
            double p = psi[(long)iz * ld + ir];
            if (p != p) p = pa; else if (p > 1.79e308) p = pa + 1e9; else if (p < -1.79e308) p = pa - 1e9;
            double R = fmin(fmax(r_axis[ir], 0.0), 1e6);
            double pn = (p - pa) / denom;
            if (pn >= 0 && pn < 1.0) s += (1 - pn) * R;
        }
    }
    double bs = block_sum<256>(s, sh);
    if (threadIdx.x == 0) part[(long)b * nblk + blockIdx.x] = bs;
}
// diag_term = (-mu0 R) * dJ/dpsi (newton_solver.py:215) and, for the diagonal preconditioner,
// inv_diag = 1 / safe(-2/dR^2 - 2/dZ^2 - diag_term) (fusion_kernel_solver_runtime.py:222-232); ring / pads stay 0.
// Every block first finishes s = sum(...) * dR * dZ and c = I_target / max(|s|, 1e-9) from the partials (:124-128).
__global__ void __launch_bounds__(256) k_nt_jac_diag(Fld dF, Fld ivdF, Fld psiF, Fld jF, const Member* st, const NtMember* nt, int nz,
                                                     int nr, int ld, const double* r_axis, const double* neg_mu0R, int external_mode,
                                                     int precond, double diag_lap, double dr, double dz, const double* part, int nblk) {
    __shared__ double sh[4];
    const int b = blockIdx.z;
    if (!nt[b].on) return;
    double jac_c = 0.0;
    if (!external_mode) {
        double sv = sum_partials(part + (long)b * nblk, nblk, sh) * dr * dz;
        if (!finite_d(sv)) sv = 0.0;
        double it = st[b].i_target;
        if (!finite_d(it)) it = 0.0;
        jac_c = it / fmax(fabs(sv), 1e-9);
    }
    const int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir < 1 || iz < 1 || ir >= nr - 1 || iz >= nz - 1) return;
    const long c = (long)iz * ld + ir;
    double pa;
    const double denom = jac_denom_of(st[b], pa);
    double p = psiF.at(b)[c];
    if (p != p) p = pa; else if (p > 1.79e308) p = pa + 1e9; else if (p < -1.79e308) p = pa - 1e9;
    const double R = fmin(fmax(r_axis[ir], 0.0), 1e6);
    const double pn = (p - pa) / denom;
    const bool mask = pn >= 0 && pn < 1.0;
    double dj = 0.0;
    if (mask) {
        if (external_mode) {
            double ja = jF.at(b)[c];
            if (ja != ja) ja = 0.0; else if (!finite_d(ja)) ja = 1e12;
            dj = -fabs(ja) / fmax(fabs(denom), 1e-9);
            if (dj != dj) dj = 0.0; else if (dj > 1.79e308) dj = 0.0; else if (dj < -1.79e308) dj = -1e12;
        } else {
            dj = (-jac_c * R) / denom;
            if (dj != dj) dj = 0.0; else if (dj > 1.79e308) dj = 1e12; else if (dj < -1.79e308) dj = -1e12;
        }
    }
    const double d = neg_mu0R[ir] * dj;
    dF.at(b)[c] = d;
    if (precond) {
        const double dja = diag_lap - d;
        const double sd = fabs(dja) > 1e-12 ? dja : (dja >= 0.0 ? 1e-12 : -1e-12);
        ivdF.at(b)[c] = 1.0 / sd;
    }
}

// psi_trial = base + alpha * delta on the interior, ring from bc (newton_solver.py:253-256 / :296-297); per-block NaN/Inf flag.
// which: 0 fixed step (members nt.on, alpha = fixed_alpha, base == psi in place)
//        1 line-search trial (members nt.ls_on, alpha = nt.ls_alpha)   2 rejected-all fallback (members on && !accepted)
__global__ void __launch_bounds__(256) k_nt_step(Fld psiF, Fld baseF, Fld dxF, Fld bcF, const NtMember* nt, int nz, int nr, int ld,
                                                 int which, double fixed_alpha, int* part_bad) {
    __shared__ int shbad;
    const int b = blockIdx.z;
    const NtMember& m0 = nt[b];
    const bool go = which == 0 ? m0.on : (which == 1 ? m0.ls_on : (m0.on && !m0.ls_accepted));
    if (!go) return;
    if (threadIdx.x == 0 && threadIdx.y == 0) shbad = 0;
    __syncthreads();
    const int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    const double alpha = which == 0 ? fixed_alpha : m0.ls_alpha;
    if (ir < nr && iz < nz) {
        const long c = (long)iz * ld + ir;
        double v;
        if (iz == 0 || iz == nz - 1 || ir == 0 || ir == nr - 1) v = bcF.at(b)[c];
        else v = baseF.at(b)[c] + alpha * dxF.at(b)[c];
        psiF.at(b)[c] = v;
        if (!finite_d(v)) atomicOr(&shbad, 1);
    }
    __syncthreads();
    if (part_bad && threadIdx.x == 0 && threadIdx.y == 0)
        part_bad[(long)b * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x] = shbad;
}
// NaN / Inf verdict of a Newton step (newton_solver.py:301-306): stop iterating, keep the state.  grid (B), 256 threads
__global__ void __launch_bounds__(256) k_nt_step_final(Member* st, NtMember* nt, const int* part_bad, int nblk, int* f_active, int* n_on) {
    __shared__ int shbad;
    const int b = blockIdx.x;
    if (!nt[b].on) return;
    if (threadIdx.x == 0) shbad = 0;
    __syncthreads();
    int bad = 0;
    for (int i = threadIdx.x; i < nblk; i += 256) bad |= part_bad[(long)b * nblk + i];
    if (bad) atomicOr(&shbad, 1);
    __syncthreads();
    if (threadIdx.x == 0 && shbad) {
        NtMember& m = nt[b];
        m.diverged = 1;
        m.on = 0;
        st[b].active = 0;
        st[b].diverged = 1;
        f_active[b] = 0;
        atomicSub(n_on, 1);
    }
}
// line-search bookkeeping; one thread per member
// op 0: start of a Newton step (ls_on = on, alpha = newton_alpha, accepted = 0)
// op 1: Member.active := ls_on          op 2: Member.active := on && !accepted        op 3: Member.active := on
// op 4: iteration counter (iterations = k + 1 for members still on)
__global__ void k_nt_flags(Member* st, NtMember* nt, int* f_active, int B, int op, double alpha0, int k) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    NtMember& m = nt[b];
    if (op == 0) { m.ls_on = m.on; m.ls_alpha = alpha0; m.ls_accepted = 0; return; }
    if (op == 4) { if (m.on) m.iterations = k + 1; return; }
    int a = op == 1 ? m.ls_on : (op == 2 ? (m.on && !m.ls_accepted) : m.on);
    st[b].active = a;
    f_active[b] = a;
}
// masked field copy keyed on NtMember flags: which 0 -> members on, 1 -> members on && accepted
__global__ void k_nt_copy(Fld dst, Fld src, const NtMember* nt, int which, int nz, int nr, int ld) {
    const int b = blockIdx.z;
    if (!(which == 0 ? nt[b].on : (nt[b].on && nt[b].ls_accepted))) return;
    const int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    const long c = (long)iz * ld + ir;
    dst.at(b)[c] = src.at(b)[c];
}

// ------------------------------------------------------------------------------------------------ GMRES
// b = -r_k, x = 0, r = b; per-block sum b^2 (scipy _isolve/iterative.py gmres prologue).  flat grid (nbf, B)
__global__ void __launch_bounds__(256) k_gm_init(double* rhs, double* x, double* r, const double* rk, long stride, GmState* gm,
                                                 const NtMember* nt, double* part, int nbf) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    if (!nt[b].on) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { gm[b].solve_on = 0; gm[b].inner_on = 0; }
        return;
    }
    const long off = (long)b * stride;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        long i = ((long)blockIdx.x * 512 + k * 256 + threadIdx.x) * 2;
        if (i < stride) {
            double2 v = *reinterpret_cast<const double2*>(rk + off + i);
            double2 nb{-v.x, -v.y};
            *reinterpret_cast<double2*>(rhs + off + i) = nb;
            *reinterpret_cast<double2*>(r + off + i) = nb;
            *reinterpret_cast<double2*>(x + off + i) = double2{0.0, 0.0};
            s += nb.x * nb.x;
            s += nb.y * nb.y;
        }
    }
    double bs = block_sum<256>(s, sh);
    if (threadIdx.x == 0) part[(long)b * nbf + blockIdx.x] = bs;
}
// scalar tails of the GMRES driver; one block per member.
// mode 0 (after k_gm_init): ||b||, atol, solve_on                      mode 1 (after k_gm_norm): Mb = ||M^-1 b||, first ptol
// mode 2 (after the end-of-cycle residual): rnorm, restart / exit logic, inner-tolerance adaptation, telemetry
// mode 3 (multigrid-preconditioned capped solve, last cycle): exit without the true residual
__global__ void __launch_bounds__(256) k_gm_tail(GmState* gm, NtMember* nt, int mode, double atol_in, double rtol, int restart,
                                                 int precond, int maxiter, const double* part, int nblk, int* n_solve) {
    __shared__ double sh[4];
    const int b = blockIdx.x;
    if (mode == 0 ? !nt[b].on : !gm[b].solve_on) return;
    const double tot = mode == 3 ? 0.0 : sum_partials(part + (long)b * nblk, nblk, sh);
    if (threadIdx.x != 0) return;
    GmState& g = gm[b];
    const double eps = 2.220446049250313e-16;
    if (mode == 3) {  // final cycle of a capped (inexact-Newton) solve: judged on the Arnoldi estimate of the preconditioned residual
        g.cycles += 1;
        g.solve_on = 0;
        g.info = (g.presid <= g.ptol || g.breakdown) ? 0 : maxiter;
        NtMember& m = nt[b];
        m.gm_last_info = g.info;
        if (g.info > 0) m.gm_nonconv += 1;
        m.gm_inner += g.inner_total;
        atomicSub(n_solve, 1);
        return;
    }
    if (mode == 0) {
        g.bnrm2 = sqrt(tot);
        g.atol = fmax(atol_in, rtol * g.bnrm2);
        g.ptol_max_factor = 1.0;
        g.presid = 0.0;
        g.rnorm = g.bnrm2;
        g.restart = restart;
        g.inner_total = 0;
        g.cycles = 0;
        g.breakdown = 0;
        g.inner_on = 0;
        g.col_final = 0;
        if (g.bnrm2 == 0.0 || g.bnrm2 < g.atol) { g.solve_on = 0; g.info = 0; nt[b].gm_last_info = 0; }  // x = 0 is the answer
        else { g.solve_on = 1; g.info = -1; }
        g.Mb = g.bnrm2;
        if (!precond && g.solve_on) g.ptol = g.Mb * fmin(1.0, g.atol / g.bnrm2);
    } else if (mode == 1) {
        g.Mb = sqrt(tot);
        g.ptol = g.Mb * fmin(1.0, g.atol / g.bnrm2);
    } else {
        g.rnorm = sqrt(tot);
        g.cycles += 1;
        bool stop = false;
        if (g.rnorm <= g.atol) stop = true;
        else if (g.breakdown) stop = true;
        else if (g.presid <= g.ptol) g.ptol_max_factor = fmax(eps, 0.25 * g.ptol_max_factor);
        else g.ptol_max_factor = fmin(1.0, 1.5 * g.ptol_max_factor);
        if (!stop) g.ptol = g.presid * fmin(g.ptol_max_factor, g.atol / g.rnorm);
        if (stop || g.cycles >= maxiter) {
            g.solve_on = 0;
            g.info = (g.rnorm <= g.atol) ? 0 : maxiter;
            NtMember& m = nt[b];
            m.gm_last_info = g.info;
            if (g.info > 0) m.gm_nonconv += 1;
            m.gm_inner += g.inner_total;
            atomicSub(n_solve, 1);
        }
    }
}
// v0 = M^-1 r (start of a restart cycle; only_norm: nothing stored, for Mb = ||M^-1 b||) + per-block sum of squares
__global__ void __launch_bounds__(256) k_gm_start(double* v0, const double* r, const double* ivd, long stride, const GmState* gm, int precond,
                                                  int only_norm, double* part, int nbf) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    if (!gm[b].solve_on) return;
    const long off = (long)b * stride;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        long i = ((long)blockIdx.x * 512 + k * 256 + threadIdx.x) * 2;
        if (i < stride) {
            double2 v = *reinterpret_cast<const double2*>(r + off + i);
            if (precond) {
                double2 d = *reinterpret_cast<const double2*>(ivd + off + i);
                v.x = d.x * v.x;
                v.y = d.y * v.y;
            }
            if (!only_norm) *reinterpret_cast<double2*>(v0 + off + i) = v;
            s += v.x * v.x;
            s += v.y * v.y;
        }
    }
    double bs = block_sum<256>(s, sh);
    if (threadIdx.x == 0) part[(long)b * nbf + blockIdx.x] = bs;
}
// which 0: v0 *= 1/tmp with tmp = ||v0|| finished from k_gm_start's partials by every block; block 0 also resets the cycle state
//          (S = tmp e_0, inner loop on).
// which 1: v[col+1] = w * (1/h1) for the members still inside the inner loop (h1 parked in g.h0 by k_gm_col_tail).
__global__ void __launch_bounds__(256) k_gm_scale(double* dst, const double* src, long stride, GmState* gm, int which, const double* part,
                                                  int nbf) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    GmState& g = gm[b];
    if (!g.solve_on) return;
    double f;
    if (which == 0) {
        const double tmp = sqrt(sum_partials(part + (long)b * nbf, nbf, sh));
        f = 1.0 / tmp;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            g.tmp = tmp;
            for (int i = 0; i <= GM_R; ++i) g.S[i] = 0.0;
            g.S[0] = tmp;
            g.breakdown = 0;
            g.inner_on = 1;
            g.col_final = 0;
        }
    } else {
        if (!g.inner_on) return;
        f = 1.0 / g.h0;
    }
    const long off = (long)b * stride;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        long i = ((long)blockIdx.x * 512 + k * 256 + threadIdx.x) * 2;
        if (i < stride) {
            double2 v = *reinterpret_cast<const double2*>(src + off + i);
            v.x *= f;
            v.y *= f;
            *reinterpret_cast<double2*>(dst + off + i) = v;
        }
    }
}
// w = M^-1 (L v - D v) on the interior + per-block ||w||^2.  2-D grid, block 64x4.
// resid_mode: out = b - (L x - D x) instead (end-of-cycle residual; members solve_on)
__global__ void __launch_bounds__(256) k_gm_matvec(double* out, const double* vin, const double* dterm, const double* ivd,
                                                   const double* rhs, long stride, Lvl L, const GmState* gm, int precond, int resid_mode,
                                                   double* part) {
    __shared__ double sh[4];
    int bx_, by_, b;  // XCD-contiguous block order: the rows above / below a 64x4 block come from the same XCD's L2
    {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        const unsigned lin = xcd_contiguous(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
        bx_ = (int)(lin % gx);
        by_ = (int)((lin / gx) % gy);
        b = (int)(lin / (gx * gy));
    }
    if (!gm[b].solve_on) return;
    if (!resid_mode && !gm[b].inner_on) return;
    const long off = (long)b * stride;
    const int ir = bx_ * 64 + threadIdx.x, iz = by_ * 4 + threadIdx.y;
    double v = 0.0;
    if (iz > 0 && iz < L.nz - 1 && ir > 0 && ir < L.nr - 1) {
        const long c = (long)iz * L.ld + ir;
        const double* x = vin + off;
        double av = lop_at(x, L, iz, ir) - dterm[off + c] * x[c];
        if (resid_mode) v = rhs[off + c] - av;
        else v = precond ? ivd[off + c] * av : av;
        out[off + c] = v;
    }
    const int nblk = gridDim.x * gridDim.y;
    double bs = block_sum_flat(v * v, sh);
    if (threadIdx.x == 0 && threadIdx.y == 0) part[(long)b * nblk + by_ * gridDim.x + bx_] = bs;
}
// ---- finite-difference Jacobian action (opt-in, solver.newton_jacobian = "finite_difference"): the reference's profile
// Jacobian (k_nt_jac_diag) drops the non-local terms of dS/dpsi -- the dependence of psi_axis / psi_bnd and of the I_p
// renormalisation on psi -- so its Newton step needs the fixed 0.5 damping and converges linearly (rate 0.5).  With
//   J v = L*v - (S(psi + eps v) - S(psi)) / eps
// the step is a true Newton step.  eps_b = 1e-7 * sqrt(n_interior) * max(|psi_axis|, 1e-6) / ||v||_2 keeps eps*v at 1e-7 of psi.
// part == nullptr: v is a Krylov basis vector (unit 2-norm by construction), no norm pass was run.  The kernel also copies the
// member record into the scratch table the perturbed topology / current sum are written to (was a device-to-device memcpy).
__global__ void __launch_bounds__(256) k_fd_eps(double* eps, const double* part, int nblk, const Member* st, double n_interior,
                                                const GmState* gm, Member* scratch) {
    __shared__ double sh[4];
    const int b = blockIdx.x;
    if (scratch && threadIdx.x == 64) scratch[b] = st[b];
    if (!gm[b].solve_on) return;
    const double tot = part ? sum_partials(part + (long)b * nblk, nblk, sh) : 1.0;
    if (threadIdx.x == 0) eps[b] = tot > 0.0 ? 1e-7 * sqrt(n_interior) * fmax(fabs(st[b].psi_axis), 1e-6) / sqrt(tot) : 0.0;
}
// out = psi + eps_b * v over the whole member stride (v has a zero ring: the Dirichlet data are untouched)
__global__ void __launch_bounds__(256) k_fd_perturb(Fld outF, Fld psiF, const double* v, long stride, const double* eps, const GmState* gm) {
    const int b = blockIdx.y;
    if (!gm[b].solve_on) return;
    const double e = eps[b];
    const double* p = psiF.at(b);
    double* o = outF.at(b);
    const double* vb = v + (long)b * stride;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2; i < stride; i += (long)gridDim.x * 512) {
        const double2 a = *reinterpret_cast<const double2*>(p + i), c = *reinterpret_cast<const double2*>(vb + i);
        *reinterpret_cast<double2*>(o + i) = double2{a.x + e * c.x, a.y + e * c.y};
    }
}
// out = L*v - (S_pert - S_0)/eps (resid_mode: rhs - that) + per-block sum of squares; grid / block as k_gm_matvec
__global__ void __launch_bounds__(256) k_gm_matvec_fd(double* out, const double* vin, Fld spF, Fld s0F, const double* eps, const double* rhs,
                                                      long stride, Lvl L, const GmState* gm, int resid_mode, double* part) {
    __shared__ double sh[4];
    int bx_, by_, b;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        const unsigned lin = xcd_contiguous(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
        bx_ = (int)(lin % gx);
        by_ = (int)((lin / gx) % gy);
        b = (int)(lin / (gx * gy));
    }
    if (!gm[b].solve_on) return;
    if (!resid_mode && !gm[b].inner_on) return;
    const long off = (long)b * stride;
    const int ir = bx_ * 64 + threadIdx.x, iz = by_ * 4 + threadIdx.y;
    double v = 0.0;
    if (iz > 0 && iz < L.nz - 1 && ir > 0 && ir < L.nr - 1) {
        const long c = (long)iz * L.ld + ir;
        const double e = eps[b];
        const double ds = e != 0.0 ? (spF.at(b)[c] - s0F.at(b)[c]) / e : 0.0;
        const double av = lop_at(vin + off, L, iz, ir) - ds;
        v = resid_mode ? rhs[off + c] - av : av;
        out[off + c] = v;
    }
    const int nblk = gridDim.x * gridDim.y;
    double bs = block_sum_flat(v * v, sh);
    if (threadIdx.x == 0 && threadIdx.y == 0) part[(long)b * nblk + by_ * gridDim.x + bx_] = bs;
}
// one modified Gram-Schmidt step of column `col` (scipy gmres: "for k in range(col+1): tmp = dot(v[k], w); w -= tmp*v[k]"):
//   every block first finishes the PREVIOUS launch's reduction from `part_in` (k == 0: h0 = ||w||, else h[col][k-1] = <v[k-1], w>),
//   block 0 stores it in the member state;
//   k > 0 : w -= h[col][k-1] * v[k-1]
//   k<=col: partial <v[k], w>            k == col+1: partial ||w||^2 (h1)
// V: base of the Krylov basis, vector j of member b at V + (j*B + b)*stride.  flat grid (nbf, B)
__global__ void __launch_bounds__(256) k_gm_mgs(double* w, const double* V, long stride, int B, GmState* gm, int col, int k,
                                                const double* part_in, int n_in, double* part_out, int nbf) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    if (!gm[b].solve_on || !gm[b].inner_on) return;
    GmState& g = gm[b];
    const double prev = sum_partials(part_in + (long)b * n_in, n_in, sh);
    double hprev = 0.0;
    if (k == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) g.h0 = sqrt(prev);
    } else {
        hprev = prev;
        if (blockIdx.x == 0 && threadIdx.x == 0) g.h[col * (GM_R + 1) + (k - 1)] = prev;
    }
    const long off = (long)b * stride;
    const bool tail = k == col + 1;
    const double* vp = k > 0 ? V + ((long)(k - 1) * B + b) * stride : nullptr;
    const double* vk = tail ? nullptr : V + ((long)k * B + b) * stride;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        long i = ((long)blockIdx.x * 512 + q * 256 + threadIdx.x) * 2;
        if (i < stride) {
            double2 wv = *reinterpret_cast<const double2*>(w + off + i);
            if (k > 0) {
                double2 p = *reinterpret_cast<const double2*>(vp + i);
                wv.x -= hprev * p.x;
                wv.y -= hprev * p.y;
                *reinterpret_cast<double2*>(w + off + i) = wv;
            }
            if (tail) { s += wv.x * wv.x; s += wv.y * wv.y; }
            else {
                double2 c = *reinterpret_cast<const double2*>(vk + i);
                s += c.x * wv.x;
                s += c.y * wv.y;
            }
        }
    }
    double bs = block_sum<256>(s, sh);
    if (threadIdx.x == 0) part_out[(long)b * nbf + blockIdx.x] = bs;
}
// column tail, one block per member: h1 = ||w||, breakdown test, rotations, residual estimate, inner-loop exit and -- when the
// inner loop ends -- the triangular solve for y.  The Hessenberg column / rotations are staged through LDS.
__global__ void __launch_bounds__(256) k_gm_col_tail(GmState* gm, int col, const double* part, int nbf, int* n_inner) {
    __shared__ double sh[4];
    __shared__ double hc[GM_R + 1], gv[GM_R * 2];
    const int b = blockIdx.x;
    if (!gm[b].solve_on || !gm[b].inner_on) return;
    GmState& g = gm[b];
    const double tot = sum_partials(part + (long)b * nbf, nbf, sh);
    for (int i = threadIdx.x; i <= col; i += 256) hc[i] = g.h[col * (GM_R + 1) + i];
    for (int i = threadIdx.x; i < 2 * col; i += 256) gv[i] = g.giv[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double eps = 2.220446049250313e-16;
    const double h1 = sqrt(tot);
    hc[col + 1] = h1;
    int breakdown = 0;
    if (h1 <= eps * g.h0) { hc[col + 1] = 0.0; breakdown = 1; }
    g.breakdown = breakdown;
    g.h0 = h1;  // parked for k_gm_scale
    for (int j = 0; j < col; ++j) {
        const double c = gv[2 * j], s_ = gv[2 * j + 1];
        const double n0 = hc[j], n1 = hc[j + 1];
        hc[j] = c * n0 + s_ * n1;
        hc[j + 1] = -s_ * n0 + c * n1;
    }
    double c, s_, mag;
    dlartg(hc[col], hc[col + 1], c, s_, mag);
    g.giv[2 * col] = c;
    g.giv[2 * col + 1] = s_;
    hc[col] = mag;
    hc[col + 1] = 0.0;
    for (int j = 0; j <= col + 1; ++j) g.h[col * (GM_R + 1) + j] = hc[j];
    const double t2 = -s_ * g.S[col];
    g.S[col] = c * g.S[col];
    g.S[col + 1] = t2;
    g.presid = fabs(t2);
    g.inner_total += 1;
    g.col_final = col;
    if (g.presid <= g.ptol || breakdown || col == g.restart - 1) {
        g.inner_on = 0;
        atomicSub(n_inner, 1);
        // y = trsv(h[:col+1,:col+1].T, S[:col+1]) with the pseudo-solve of singular cases
        if (g.h[col * (GM_R + 1) + col] == 0.0) g.S[col] = 0.0;
        for (int j = 0; j <= col; ++j) g.y[j] = g.S[j];
        for (int j = col; j > 0; --j)
            if (g.y[j] != 0.0) {
                g.y[j] /= g.h[j * (GM_R + 1) + j];
                const double t = g.y[j];
                for (int i = 0; i < j; ++i) g.y[i] -= t * g.h[j * (GM_R + 1) + i];
            }
        if (g.y[0] != 0.0) g.y[0] /= g.h[0];
    }
}
// x += y @ v[:col_final+1]  (sequential in k per element).  flat grid (nbf, B)
__global__ void __launch_bounds__(256) k_gm_update_x(double* x, const double* V, long stride, int B, const GmState* gm) {
    const int b = blockIdx.y;
    const GmState& g = gm[b];
    if (!g.solve_on) return;
    const long off = (long)b * stride;
    const int nk = g.col_final + 1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        long i = ((long)blockIdx.x * 512 + q * 256 + threadIdx.x) * 2;
        if (i < stride) {
            double ax = 0.0, ay = 0.0;
            for (int k = 0; k < nk; ++k) {
                const double2 v = *reinterpret_cast<const double2*>(V + ((long)k * B + b) * stride + i);
                ax += g.y[k] * v.x;
                ay += g.y[k] * v.y;
            }
            double2 xv = *reinterpret_cast<double2*>(x + off + i);
            xv.x += ax;
            xv.y += ay;
            *reinterpret_cast<double2*>(x + off + i) = xv;
        }
    }
}
// ---- small grids: one whole restart cycle per launch -----------------------------------------------------------------------
// For grids up to a few 10^4 points the multi-launch pipeline above is launch-bound (~27 launches of ~7 us per Krylov
// column).  Here ONE 1024-thread workgroup per member runs a complete restart cycle -- v0, up to `restart` columns of
// operator + modified Gram-Schmidt + rotation, the triangular solve, x += V y, the true residual and the restart / exit
// logic of scipy's gmres -- with workgroup barriers instead of kernel boundaries.  Vectors stay in global memory (L2).
template <int NT>
__device__ __forceinline__ double block_sum_bcast(double v, double* sh) {  // sum over the NT threads, returned to all
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += sh[i];
    return r;
}
constexpr int GM_SMALL_PTS = 4;  // register variant: up to 4 points per thread (65^2 has 63^2 = 3969 interior points; 5 per thread spills at the
                                 // 128-VGPR cap of a 1024-thread workgroup); larger small grids use k_gm_cycle_lds
template <int MP>
__global__ void __launch_bounds__(1024) k_gm_cycle_small(double* Vb, double* xb, double* rb, const double* rhsb, const double* dtermb,
                                                         const double* ivdb, long stride, int B, Lvl L, GmState* gm, NtMember* nt,
                                                         int precond, int maxiter, int* n_solve) {
    constexpr int NT = 1024;
    __shared__ double sh[NT / 64];
    __shared__ int s_done;
    const int b = blockIdx.x;
    if (!gm[b].solve_on) return;
    GmState& g = gm[b];
    const long off = (long)b * stride;
    double* x = xb + off;
    double* r = rb + off;
    const double* rhs = rhsb + off;
    const double* dterm = dtermb + off;
    const double* ivd = ivdb + off;
    const int nz = L.nz, nr = L.nr, ld = L.ld, ni = nr - 2;
    const long nint = (long)(nz - 2) * ni;
    const double eps = 2.220446049250313e-16;
    auto vec = [&](int j) { return Vb + ((long)j * B + b) * stride; };
    const int tid = threadIdx.x;
    // this thread's grid points (fixed for the whole cycle); w lives in registers, one entry per point.  A point is kept as ONE word:
    // linear index iz * ld + ir in the low 20 bits (the launcher admits grids of at most 20 000 points here), column ir above them --
    // the operator needs both, and recovering (iz, ir) from the index by an integer division per point and column cost the registers
    // that spilled the four-point instantiation (128 VGPRs + 136 B of scratch; now no scratch)
    int cpk[MP];
    double wr[MP];
#pragma unroll
    for (int j = 0; j < MP; ++j) {
        const long q = tid + (long)j * NT;
        cpk[j] = -1;
        wr[j] = 0.0;
        if (q < nint) {
            const int iz = 1 + (int)(q / ni), ir = 1 + (int)(q - (long)(iz - 1) * ni);
            cpk[j] = (iz * ld + ir) | (ir << 20);
        }
    }
#define CIX(j) (cpk[j] & 0xFFFFF)
#define CIR(j) (cpk[j] >> 20)
    // ---- v0 = M^-1 r / ||M^-1 r||
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < MP; ++j)
        if (cpk[j] >= 0) {
            double v = r[CIX(j)];
            if (precond) v = ivd[CIX(j)] * v;
            wr[j] = v;
            s += v * v;
        }
    const double tmp = sqrt(block_sum_bcast<NT>(s, sh));
    {
        const double f = 1.0 / tmp;
        double* v0 = vec(0);
#pragma unroll
        for (int j = 0; j < MP; ++j)
            if (cpk[j] >= 0) v0[CIX(j)] = wr[j] * f;
    }
    if (tid == 0) {
        g.tmp = tmp;
        for (int i = 0; i <= GM_R; ++i) g.S[i] = 0.0;
        g.S[0] = tmp;
        g.breakdown = 0;
        s_done = 0;
    }
    __syncthreads();
    int col = 0;
    for (col = 0; col < g.restart; ++col) {
        const double* vc = vec(col);
        // operator + preconditioner, h0 = ||w||
        s = 0.0;
#pragma unroll
        for (int j = 0; j < MP; ++j)
            if (cpk[j] >= 0) {
                double av = lop_at_lin(vc, L, CIX(j), CIR(j)) - dterm[CIX(j)] * vc[CIX(j)];
                if (precond) av = ivd[CIX(j)] * av;
                wr[j] = av;
                s += av * av;
            }
        const double h0 = sqrt(block_sum_bcast<NT>(s, sh));
        // modified Gram-Schmidt: v_k is read once per step (dot, then the update from the same registers)
        for (int k = 0; k <= col; ++k) {
            const double* vk = vec(k);
            double vr[MP];
            s = 0.0;
#pragma unroll
            for (int j = 0; j < MP; ++j) {
                vr[j] = cpk[j] >= 0 ? vk[CIX(j)] : 0.0;
                s += vr[j] * wr[j];
            }
            const double t = block_sum_bcast<NT>(s, sh);
            if (tid == 0) g.h[col * (GM_R + 1) + k] = t;
#pragma unroll
            for (int j = 0; j < MP; ++j) wr[j] -= t * vr[j];
        }
        s = 0.0;
#pragma unroll
        for (int j = 0; j < MP; ++j) s += wr[j] * wr[j];
        const double h1 = sqrt(block_sum_bcast<NT>(s, sh));
        const bool breakdown = h1 <= eps * h0;
        if (col + 1 < g.restart) {  // v[col+1] = w (scaled unless breakdown); the last column's successor is never used
            const double f = 1.0 / h1;
            double* vn = vec(col + 1);
#pragma unroll
            for (int j = 0; j < MP; ++j)
                if (cpk[j] >= 0) vn[CIX(j)] = breakdown ? wr[j] : wr[j] * f;
        }
        if (tid == 0) {
            double* hc = g.h + col * (GM_R + 1);
            hc[col + 1] = breakdown ? 0.0 : h1;
            g.breakdown = breakdown ? 1 : 0;
            for (int j = 0; j < col; ++j) {
                const double c = g.giv[2 * j], s_ = g.giv[2 * j + 1];
                const double n0 = hc[j], n1 = hc[j + 1];
                hc[j] = c * n0 + s_ * n1;
                hc[j + 1] = -s_ * n0 + c * n1;
            }
            double c, s_, mag;
            dlartg(hc[col], hc[col + 1], c, s_, mag);
            g.giv[2 * col] = c;
            g.giv[2 * col + 1] = s_;
            hc[col] = mag;
            hc[col + 1] = 0.0;
            const double t2 = -s_ * g.S[col];
            g.S[col] = c * g.S[col];
            g.S[col + 1] = t2;
            g.presid = fabs(t2);
            g.inner_total += 1;
            g.col_final = col;
            s_done = (g.presid <= g.ptol || breakdown) ? 1 : 0;
        }
        __syncthreads();
        if (s_done) break;
    }
    if (col == g.restart) col = g.restart - 1;
    // ---- y = trsv(h^T, S) with the pseudo-solve of singular cases, x += V y
    if (tid == 0) {
        if (g.h[col * (GM_R + 1) + col] == 0.0) g.S[col] = 0.0;
        for (int j = 0; j <= col; ++j) g.y[j] = g.S[j];
        for (int j = col; j > 0; --j)
            if (g.y[j] != 0.0) {
                g.y[j] /= g.h[j * (GM_R + 1) + j];
                const double t = g.y[j];
                for (int i = 0; i < j; ++i) g.y[i] -= t * g.h[j * (GM_R + 1) + i];
            }
        if (g.y[0] != 0.0) g.y[0] /= g.h[0];
    }
    __syncthreads();
    {
        double acc[MP];
#pragma unroll
        for (int j = 0; j < MP; ++j) acc[j] = 0.0;
        for (int k = 0; k <= col; ++k) {
            const double yk = g.y[k];
            const double* vk = vec(k);
#pragma unroll
            for (int j = 0; j < MP; ++j)
                if (cpk[j] >= 0) acc[j] += yk * vk[CIX(j)];
        }
#pragma unroll
        for (int j = 0; j < MP; ++j)
            if (cpk[j] >= 0) x[CIX(j)] += acc[j];
    }
    __syncthreads();
    // ---- true residual r = b - A x, restart / exit logic
    s = 0.0;
#pragma unroll
    for (int j = 0; j < MP; ++j)
        if (cpk[j] >= 0) {
            const double v = rhs[CIX(j)] - (lop_at_lin(x, L, CIX(j), CIR(j)) - dterm[CIX(j)] * x[CIX(j)]);
            r[CIX(j)] = v;
            s += v * v;
        }
    const double rnorm = sqrt(block_sum_bcast<NT>(s, sh));
    if (tid == 0) {
        g.rnorm = rnorm;
        g.cycles += 1;
        bool stop = false;
        if (g.rnorm <= g.atol) stop = true;
        else if (g.breakdown) stop = true;
        else if (g.presid <= g.ptol) g.ptol_max_factor = fmax(eps, 0.25 * g.ptol_max_factor);
        else g.ptol_max_factor = fmin(1.0, 1.5 * g.ptol_max_factor);
        if (!stop) g.ptol = g.presid * fmin(g.ptol_max_factor, g.atol / g.rnorm);
        if (stop || g.cycles >= maxiter) {
            g.solve_on = 0;
            g.info = (g.rnorm <= g.atol) ? 0 : maxiter;
            NtMember& m = nt[b];
            m.gm_last_info = g.info;
            if (g.info > 0) m.gm_nonconv += 1;
            m.gm_inner += g.inner_total;
            atomicSub(n_solve, 1);
        }
    }
}
#undef CIX
#undef CIR

// same cycle for grids up to 129^2: the work vector w lives in LDS (one 8-byte slot per interior point, 127 KB at 129^2),
// the loops over the thread's points are rolled (no register arrays)
__global__ void __launch_bounds__(1024) k_gm_cycle_lds(double* Vb, double* xb, double* rb, const double* rhsb, const double* dtermb,
                                                       const double* ivdb, long stride, int B, Lvl L, GmState* gm, NtMember* nt, int precond,
                                                       int maxiter, int* n_solve) {
    constexpr int NT = 1024;
    extern __shared__ __attribute__((aligned(16))) double wl[];  // w, indexed by interior point number q
    __shared__ double sh[NT / 64];
    __shared__ int s_done;
    const int b = blockIdx.x;
    if (!gm[b].solve_on) return;
    GmState& g = gm[b];
    const long off = (long)b * stride;
    double* x = xb + off;
    double* r = rb + off;
    const double* rhs = rhsb + off;
    const double* dterm = dtermb + off;
    const double* ivd = ivdb + off;
    const int nz = L.nz, nr = L.nr, ld = L.ld, ni = nr - 2;
    const int nint = (nz - 2) * ni;
    const double eps = 2.220446049250313e-16;
    auto vec = [&](int j) { return Vb + ((long)j * B + b) * stride; };
    const int tid = threadIdx.x;
#define GM_FOR_OWN for (int q = tid; q < nint; q += NT)
#define GM_CELL const int iz = 1 + q / ni, ir = 1 + (q - (iz - 1) * ni); const long c = (long)iz * ld + ir
    double s = 0.0;
    GM_FOR_OWN {
        GM_CELL;
        double v = r[c];
        if (precond) v = ivd[c] * v;
        wl[q] = v;
        s += v * v;
    }
    const double tmp = sqrt(block_sum_bcast<NT>(s, sh));
    {
        const double f = 1.0 / tmp;
        double* v0 = vec(0);
        GM_FOR_OWN { GM_CELL; v0[c] = wl[q] * f; }
    }
    if (tid == 0) {
        g.tmp = tmp;
        for (int i = 0; i <= GM_R; ++i) g.S[i] = 0.0;
        g.S[0] = tmp;
        g.breakdown = 0;
        s_done = 0;
    }
    __syncthreads();
    int col = 0;
    for (col = 0; col < g.restart; ++col) {
        const double* vc = vec(col);
        s = 0.0;
        GM_FOR_OWN {
            GM_CELL;
            double av = lop_at(vc, L, iz, ir) - dterm[c] * vc[c];
            if (precond) av = ivd[c] * av;
            wl[q] = av;
            s += av * av;
        }
        const double h0 = sqrt(block_sum_bcast<NT>(s, sh));
        for (int k = 0; k <= col; ++k) {
            const double* vk = vec(k);
            s = 0.0;
            GM_FOR_OWN { GM_CELL; s += vk[c] * wl[q]; }
            const double t = block_sum_bcast<NT>(s, sh);
            if (tid == 0) g.h[col * (GM_R + 1) + k] = t;
            GM_FOR_OWN { GM_CELL; wl[q] -= t * vk[c]; }
        }
        s = 0.0;
        GM_FOR_OWN { const double v = wl[q]; s += v * v; }
        const double h1 = sqrt(block_sum_bcast<NT>(s, sh));
        const bool breakdown = h1 <= eps * h0;
        if (col + 1 < g.restart) {
            const double f = 1.0 / h1;
            double* vn = vec(col + 1);
            GM_FOR_OWN { GM_CELL; vn[c] = breakdown ? wl[q] : wl[q] * f; }
        }
        if (tid == 0) {
            double* hc = g.h + col * (GM_R + 1);
            hc[col + 1] = breakdown ? 0.0 : h1;
            g.breakdown = breakdown ? 1 : 0;
            for (int j = 0; j < col; ++j) {
                const double c_ = g.giv[2 * j], s_ = g.giv[2 * j + 1];
                const double n0 = hc[j], n1 = hc[j + 1];
                hc[j] = c_ * n0 + s_ * n1;
                hc[j + 1] = -s_ * n0 + c_ * n1;
            }
            double c_, s_, mag;
            dlartg(hc[col], hc[col + 1], c_, s_, mag);
            g.giv[2 * col] = c_;
            g.giv[2 * col + 1] = s_;
            hc[col] = mag;
            hc[col + 1] = 0.0;
            const double t2 = -s_ * g.S[col];
            g.S[col] = c_ * g.S[col];
            g.S[col + 1] = t2;
            g.presid = fabs(t2);
            g.inner_total += 1;
            g.col_final = col;
            s_done = (g.presid <= g.ptol || breakdown) ? 1 : 0;
        }
        __syncthreads();
        if (s_done) break;
    }
    if (col == g.restart) col = g.restart - 1;
    if (tid == 0) {
        if (g.h[col * (GM_R + 1) + col] == 0.0) g.S[col] = 0.0;
        for (int j = 0; j <= col; ++j) g.y[j] = g.S[j];
        for (int j = col; j > 0; --j)
            if (g.y[j] != 0.0) {
                g.y[j] /= g.h[j * (GM_R + 1) + j];
                const double t = g.y[j];
                for (int i = 0; i < j; ++i) g.y[i] -= t * g.h[j * (GM_R + 1) + i];
            }
        if (g.y[0] != 0.0) g.y[0] /= g.h[0];
    }
    __syncthreads();
    GM_FOR_OWN {
        GM_CELL;
        double acc = 0.0;
        for (int k = 0; k <= col; ++k) acc += g.y[k] * vec(k)[c];
        x[c] += acc;
    }
    __syncthreads();
    s = 0.0;
    GM_FOR_OWN {
        GM_CELL;
        const double v = rhs[c] - (lop_at(x, L, iz, ir) - dterm[c] * x[c]);
        r[c] = v;
        s += v * v;
    }
#undef GM_FOR_OWN
#undef GM_CELL
    const double rnorm = sqrt(block_sum_bcast<NT>(s, sh));
    if (tid == 0) {
        g.rnorm = rnorm;
        g.cycles += 1;
        bool stop = false;
        if (g.rnorm <= g.atol) stop = true;
        else if (g.breakdown) stop = true;
        else if (g.presid <= g.ptol) g.ptol_max_factor = fmax(eps, 0.25 * g.ptol_max_factor);
        else g.ptol_max_factor = fmin(1.0, 1.5 * g.ptol_max_factor);
        if (!stop) g.ptol = g.presid * fmin(g.ptol_max_factor, g.atol / g.rnorm);
        if (stop || g.cycles >= maxiter) {
            g.solve_on = 0;
            g.info = (g.rnorm <= g.atol) ? 0 : maxiter;
            NtMember& m = nt[b];
            m.gm_last_info = g.info;
            if (g.info > 0) m.gm_nonconv += 1;
            m.gm_inner += g.inner_total;
            atomicSub(n_solve, 1);
        }
    }
}

// number of members whose solve is still running (start of a cycle: the host polls this; they all enter the inner loop)
__global__ void k_gm_count(const GmState* gm, int B, int* n_solve, int* n_inner) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int s = 0;
        for (int b = 0; b < B; ++b) s += gm[b].solve_on ? 1 : 0;
        *n_solve = s;
        *n_inner = s;
    }
}

}  // namespace gs

// Time-skewed ("sliding window") form of the fused 3-sweep red-black SOR tile smoother (core/multigrid_solve.py:148-208).
//
// k_smooth_tile holds a 40-row tile in LDS and runs the six colour passes over the whole tile, so it recomputes 12-16 halo
// rows per 24-28 rows it outputs and alternates between a load phase and a sweep phase.  Here a workgroup owns a column strip
// of a LONG row segment (rseg + 12 rows) and pipelines the passes over the rows: in step t
//     wave 6 / 7 (loaders)  commit rows [2t, 2t+2) of psi (+ prolonged coarse correction) / of the source to LDS,
//                           from global loads they issued D steps earlier,
//     wave p = 0..5         runs colour pass p on rows [2t - 4(p+1), +2)          (pass p lags pass p-1 by 4 rows),
//     wave 7 (storer)       writes rows [2t - 28, +2), which pass 5 finished in an earlier step, to global memory,
// then all waves meet at a barrier.  The LDS window is circular (32 rows: row lz lives in slot lz & 31).  Every point
// receives exactly the updates it receives in k_smooth_tile with LZ = rseg + 12 -- red-black SOR only reads the other colour
// inside a pass, so the order in which rows are visited does not change a bit -- but only 12 halo rows per segment are
// recomputed, and loads, passes and stores of different rows overlap all the time.
//
// Why the lag of 4 with 2 rows per step: pass p in step t reads rows up to 2t - 4(p+1) + 2; pass p-1 has finished rows up to
// 2(t-1) - 4p + 1 = 2t - 4(p+1) + 3 in step t-1 (ok), and writes rows >= 2t - 4p = 2t - 4(p+1) + 4 in step t (never read by
// pass p in the same step).  An even lag keeps every wave's first row even, so column parities are compile-time constants.
#pragma once

namespace gs {

constexpr int SL_W = 32;      // rows in the circular LDS window
constexpr int SL_LAG = 4;     // rows between consecutive passes
constexpr int SL_D = 8;       // steps between issuing a global row load and committing it to LDS
constexpr int SL_H = 6;       // halo of three sweeps
constexpr int SL_NT = 512;    // 8 waves: 6 passes + 2 loaders (the second one also stores)

struct SlideArgs {
    Fld out, in, src, coarse;
    int nz_c, nr_c, ld_c;     // PROLONG: coarse-grid shape
    const Member* lazy_st;    // lazy source: src holds J_raw, Source = (-mu0 R) (J_raw * scale) is formed on load (see k_smooth_tile)
    const double* lazy_nm;
    const int* active;
    int rseg;                 // output rows per segment (even)
    double omega, om1;
};

template <bool PROLONG>
__global__ void __launch_bounds__(SL_NT, 4) k_smooth_slide(SlideArgs a, Lvl L) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* A0 = lds;                   // psi, colour 0: (iz + ir) even      [SL_W][64]
    double* A1 = lds + SL_W * 64;       // psi, colour 1
    double* S0 = lds + 2 * SL_W * 64;   // source at the colour-0 points
    double* S1 = lds + 3 * SL_W * 64;
    int bx_, by_, b;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        lin = xcd_contiguous(lin, total);
        bx_ = (int)(lin % gx);
        const unsigned t_ = lin / gx;
        by_ = (int)(t_ % gy);
        b = (int)(t_ / gy);
    }
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index as a scalar: row arithmetic goes to the SALU
    const int nz = L.nz, nr = L.nr, ld = L.ld;
    const int zs = by_ * a.rseg, ze = min(zs + a.rseg, nz);  // output rows [zs, ze)
    const int z_lo = zs - SL_H;                               // even
    const int LZ = (ze - zs) + 2 * SL_H;                      // rows of the (virtual) tile
    const int r0 = bx_ * (128 - 2 * SL_H) - SL_H;             // even
    const int c0 = r0 + 2 * lane;
    const double* in = a.in.at(b);
    double* out = a.out.at(b);
    const bool okx = c0 >= 0 && c0 < nr, oky = c0 + 1 >= 0 && c0 + 1 < nr;
    const int cc = min(max(c0, 0), ld - 2);
    if (a.active && !a.active[b]) {  // frozen member: the window is passed through
        for (int gz = zs + w; gz < ze; gz += SL_NT / 64) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int col = 2 * lane + p, gc = c0 + p;
                if (col >= SL_H && col < 128 - SL_H && gc < nr) out[(long)gz * ld + gc] = in[(long)gz * ld + gc];
            }
        }
        return;
    }
    const int T = (LZ + 7 * SL_LAG + 1) / 2;  // last step stores row LZ - 7 (the last output row is LZ - SL_H - 1)
    // ---- per-role state
    // compute waves (w < 6): coefficients of the lane's even / odd column, column predicates per column parity
    double ae0 = 0.0, ae1 = 0.0, aw0 = 0.0, aw1 = 0.0;
    if (c0 >= 1 && c0 <= nr - 2) { ae0 = L.a_e[c0]; aw0 = L.a_w[c0]; }
    if (c0 + 1 >= 1 && c0 + 1 <= nr - 2) { ae1 = L.a_e[c0 + 1]; aw1 = L.a_w[c0 + 1]; }
    const double a_ns = L.a_ns, a_c = L.a_c, inv_a_c = L.inv_a_c, lo_a_c = L.lo_a_c;
    const bool colok0 = 2 * lane >= 1 && 2 * lane <= 126 && c0 >= 1 && c0 <= nr - 2;              // even column (pr = 0)
    const bool colok1 = 2 * lane + 1 >= 1 && 2 * lane + 1 <= 126 && c0 + 1 >= 1 && c0 + 1 <= nr - 2;  // odd column (pr = 1)
    // loaders: rings of rows in flight
    double2 ring[SL_D][2];         // wave 6: psi rows, wave 7: source rows
    double cring[2][4];            // wave 6, PROLONG: coarse values around the row pair (rows m, m+1 x columns jcc, jcc1), issued 2 steps ahead
    const double* gsrc = a.src.at(b);
    const double* ccp = PROLONG ? a.coarse.at(b) : nullptr;
    const int jc = max(c0, 0) >> 1;
    const int jcc = PROLONG ? min(jc, a.nr_c - 1) : 0, jcc1 = PROLONG ? min(jc + 1, a.nr_c - 1) : 0;
    const int h_end = PROLONG ? min(2 * (a.nr_c - 1), nr - 1) : 0, v_end = PROLONG ? min(2 * (a.nz_c - 1), nz - 1) : 0;
    double lz_sc = 1.0, lz_n0 = 1.0, lz_n1 = 1.0;
    const bool lazy = a.lazy_st != nullptr;
    if (lazy) {
        lz_sc = a.lazy_st[b].jscale;
        lz_n0 = okx ? a.lazy_nm[c0] : 0.0;
        lz_n1 = oky ? a.lazy_nm[c0 + 1] : 0.0;
    }
    auto issue = [&](const int t, const int slot) {  // global loads of rows [2t, 2t+2) into ring slot `slot` (waves 6, 7)
        const double* base = (w == 6) ? in : gsrc;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int lz = 2 * t + k;
            const int gzc = min(max(z_lo + min(lz, LZ - 1), 0), nz - 1);
            ring[slot][k] = *reinterpret_cast<const double2*>(base + (long)gzc * ld + cc);
        }
    };
    auto issue_coarse = [&](const int t, const int slot) {  // wave 6, PROLONG
        // rows 2t (even gz = 2m: uses coarse row m) and 2t+1 (odd gz: rows m and m+1), clamped as in k_smooth_tile
        const int gz0 = min(max(z_lo + min(2 * t, LZ - 1), 0), nz - 1);
        const int m = gz0 >> 1, ica = min(m, a.nz_c - 1), icb = min(m + 1, a.nz_c - 1);
        cring[slot][0] = ccp[(long)ica * a.ld_c + jcc];
        cring[slot][1] = ccp[(long)ica * a.ld_c + jcc1];
        cring[slot][2] = ccp[(long)icb * a.ld_c + jcc];
        cring[slot][3] = ccp[(long)icb * a.ld_c + jcc1];
    };
    auto commit = [&](const int t, const int slot, const int cslot) {  // rows [2t, 2t+2) from the ring to LDS (waves 6, 7)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int lz = 2 * t + k, gz = z_lo + lz;
            if (lz >= LZ) continue;
            const bool rowok = gz >= 0 && gz < nz;
            double vx = (rowok && okx) ? ring[slot][k].x : 0.0, vy = (rowok && oky) ? ring[slot][k].y : 0.0;
            double* P0 = (w == 6) ? A0 : S0;
            double* P1 = (w == 6) ? A1 : S1;
            if (w == 7) {
                if (lazy) {
                    const double tx_ = vx * lz_sc, ty_ = vy * lz_sc;
                    vx = (rowok && okx) ? lz_n0 * (lz_sc != 0.0 ? tx_ : 0.0) : 0.0;
                    vy = (rowok && oky) ? lz_n1 * (lz_sc != 0.0 ? ty_ : 0.0) : 0.0;
                }
            } else if (PROLONG && rowok) {
                // bilinear prolongation, core/multigrid_solve.py:117-143 (same expressions as k_smooth_tile)
                const double ca0 = cring[cslot][0], ca1 = cring[cslot][1], cb0 = cring[cslot][2], cb1 = cring[cslot][3];
                double px, py;
                if (!(gz & 1)) {
                    px = ca0;
                    py = (c0 + 1 >= h_end) ? 0.0 : 0.5 * (ca0 + ca1);
                } else if (gz >= v_end) {
                    px = 0.0;
                    py = 0.0;
                } else {
                    px = 0.5 * (ca0 + cb0);
                    py = (c0 + 1 >= h_end) ? 0.0 : 0.25 * (ca0 + cb0 + ca1 + cb1);
                }
                if (okx) vx = vx + px;
                if (oky) vy = vy + py;
            }
            // even column has colour (gz & 1) == (lz & 1) == k since z_lo, 2t and c0 are even
            const int s = ((lz & (SL_W - 1)) * 64) + lane;
            if (k == 0) { P0[s] = vx; P1[s] = vy; }
            else { P1[s] = vx; P0[s] = vy; }
        }
    };
    // one colour pass on rows [2t - 4(p+1), +2): colour c = p & 1 (compile time), pass index p (wave-uniform)
    auto pass_rows = [&](auto colour, const int t, const int p) {
        constexpr int c = decltype(colour)::value;
        double* mine = c ? A1 : A0;
        const double* oth = c ? A0 : A1;
        const double* sp = c ? S1 : S0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int lz = 2 * t - SL_LAG * (p + 1) + k, gz = z_lo + lz;
            constexpr int dummy = 0;
            (void)dummy;
            const int pr = (c + k) & 1;  // column parity of colour c in this row (lz has the parity of k)
            if (lz >= p + 1 && lz <= LZ - 2 - p && gz >= 1 && gz <= nz - 2) {  // wave-uniform: the window of k_smooth_tile's pass p
                const int sm = ((lz & (SL_W - 1)) * 64) + lane;
                const int sS = (((lz - 1) & (SL_W - 1)) * 64) + lane, sN = (((lz + 1) & (SL_W - 1)) * 64) + lane;
                const double W = oth[sm - 1 + pr], E = oth[sm + pr];
                const double Sv = oth[sS], N = oth[sN];
                const double C = mine[sm];
                const double sv = sp[sm];
                const double ae = pr ? ae1 : ae0, aw = pr ? aw1 : aw0;
                const double gsv = divc4(ae * E + aw * W + a_ns * Sv + a_ns * N - sv, a_c, inv_a_c, lo_a_c);
                const double nw = a.om1 * C + a.omega * gsv;
                if (pr ? colok1 : colok0) mine[sm] = nw;
            }
        }
    };
    auto store_rows = [&](const int t) {  // wave 7: rows [2t - 28, +2)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int lz = 2 * t - 7 * SL_LAG + k, gz = z_lo + lz;
            if (lz >= SL_H && lz < LZ - SL_H && gz < nz) {
                const int s = ((lz & (SL_W - 1)) * 64) + lane;
                const double vx = k ? A1[s] : A0[s], vy = k ? A0[s] : A1[s];
                const int col = 2 * lane;
                const long o = (long)gz * ld + c0;
                if (col >= SL_H && col + 1 < 128 - SL_H && c0 + 1 < nr) {
                    *reinterpret_cast<double2*>(out + o) = double2{vx, vy};
                } else {
                    if (col >= SL_H && col < 128 - SL_H && c0 < nr) out[o] = vx;
                    if (col + 1 >= SL_H && col + 1 < 128 - SL_H && c0 + 1 < nr) out[o + 1] = vy;
                }
            }
        }
    };
    // ---- prologue: the loaders fill their rings
    if (w >= 6) {
#pragma unroll
        for (int u = 0; u < SL_D; ++u) issue(u, u);
        if (PROLONG && w == 6) { issue_coarse(0, 0); issue_coarse(1, 1); }
    }
    // ---- pipeline
#pragma unroll 1
    for (int t0 = 0; t0 < T; t0 += SL_D) {
#pragma unroll
        for (int u = 0; u < SL_D; ++u) {
            const int t = t0 + u;
            if (t < T) {
                if (w >= 6) {
                    commit(t, u, u & 1);
                    issue(t + SL_D, u);
                    if (PROLONG && w == 6) issue_coarse(t + 2, u & 1);
                    if (w == 7) store_rows(t);
                } else if (w & 1) {
                    pass_rows(std::integral_constant<int, 1>{}, t, w);
                } else {
                    pass_rows(std::integral_constant<int, 0>{}, t, w);
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace gs

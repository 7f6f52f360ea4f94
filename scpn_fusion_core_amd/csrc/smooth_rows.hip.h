// smooth_rows.hip.h -- register-resident, temporally blocked red-black SOR smoother (the dominant kernel of the V-cycle).
//
// Reference: mg_smooth x S sweeps (core/multigrid_solve.py:148-208); PROLONG: psi += P e on load (:117-143, :329-330);
// DEFECT: restrict_full_weight(-(L*psi - src)) on store (:211-249, :57-99, :303-306).  Same arithmetic (expression order,
// exact division) as the LDS-tiled k_smooth_tile it replaces -- red-black SOR only reads the other colour inside a colour
// pass, so the execution order inside a pass is free -- but a different mapping to the machine:
//
//   * a workgroup of NW waves owns a tile of LZ = NW*RB rows x 128 columns (halo H = 2S [+2 with DEFECT] on every side);
//     wave w holds RB CONSECUTIVE rows in registers, lane l the column pair (2l, 2l+1): psi and source never touch LDS;
//   * N/S neighbours are the lane's own registers of the adjacent rows, one of E/W is the lane's other column, the other one
//     comes from the neighbouring lane over the DPP network (wave_shr:1 / wave_shl:1, two v_mov_dpp per fp64 value);
//   * per colour pass a wave exchanges only its first / last row with the waves above / below through a 2-KiB LDS mailbox
//     (the slots written after a pass are exactly the ones the next pass reads, so one barrier per pass suffices);
//   * tile ownership is shifted by one row (tile j owns rows j*OWN+1 .. j*OWN+OWN, tile 0 also the ring row 0), so 2^k+1 grids
//     split without a sliver tile, and the colour of a register is a compile-time function of its row index (RB even).
//
// Measured on MI355X against k_smooth_tile (SQ counters in profiles/): the tiled kernel spent 206 VALU + 31 LDS wave-instructions
// per 64 grid points of the 513^2 pre-smoother (VALU 56 % / LDS array 46 % busy, neither saturated, 2.1x redundant halo work);
// this kernel needs no LDS traffic in the sweeps and 1.33 x 1.25 redundant points.
#pragma once
#include "gs_kernels.hip.h"
#include <utility>

namespace gs {

// value of lane-1 / lane+1 (0.0 beyond the wave: those are tile-edge columns whose results are never used)
__device__ __forceinline__ double lane_below(double v) {  // lane i <- lane i-1
    int l = __double2loint(v), h = __double2hiint(v);
    l = __builtin_amdgcn_update_dpp(0, l, 0x138, 0xf, 0xf, true);  // wave_shr:1
    h = __builtin_amdgcn_update_dpp(0, h, 0x138, 0xf, 0xf, true);
    return __hiloint2double(h, l);
}
__device__ __forceinline__ double lane_above(double v) {  // lane i <- lane i+1
    int l = __double2loint(v), h = __double2hiint(v);
    l = __builtin_amdgcn_update_dpp(0, l, 0x130, 0xf, 0xf, true);  // wave_shl:1
    h = __builtin_amdgcn_update_dpp(0, h, 0x130, 0xf, 0xf, true);
    return __hiloint2double(h, l);
}
// verification hook (tests/test_gpu_division.py): the direction of the two wave shifts on this device
__global__ void k_verify_lane_shift(double* below, double* above) {
    const double v = (double)(threadIdx.x + 1);
    below[threadIdx.x] = lane_below(v);
    above[threadIdx.x] = lane_above(v);
}

// m ? a : b for an all-ones / all-zeros lane mask, as two bit-field inserts (never a branch)
__device__ __forceinline__ double bitsel(int m, double a, double b) {
    const int lo = (__double2loint(a) & m) | (__double2loint(b) & ~m), hi = (__double2hiint(a) & m) | (__double2hiint(b) & ~m);
    return __hiloint2double(hi, lo);
}

struct LazySrc {  // source = (-mu0 R) * (J_raw * scale) formed on load (lazy-source Picard iterations); st == nullptr: plain source
    const Member* st = nullptr;
    const double* neg_mu0R = nullptr;
};

// diagnostic build (-DCV_STAMPS): workgroup 1 records s_memrealtime stamps of its first 8 tiles (row = tile, 16 stamps each)
#ifdef CV_STAMPS
#define ROWS_STAMP(i) do { if (DEFECT == STAMP_DEFECT && PROLONG == STAMP_PROLONG && L.nz >= 300 && blockIdx.x == 1 && threadIdx.x == 0 && tile_iter < 8) \
        g_tile_stamps[tile_iter * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ROWS_STAMP(i)
#endif

struct RowsArgs {
    Fld out, in, src, coarse;  // coarse: correction to prolongate (PROLONG) or coarse rhs to produce (DEFECT)
    Lvl L;
    double omega, om1;
    const int* active;
    int nz_c, nr_c, ld_c;
    LazySrc lzs;
    DefectDst dd;
    int xcd_remap;
    int gx, gy, B;  // tiles per row / column of a member, members: tile t = (b * gy + by) * gx + bx
    BlendArgs ba;   // BLEND only
};

// One workgroup works through the tiles t = wg, wg + G, wg + 2G, ... (G = gridDim.x).  PERSIST (G < number of tiles, one
// workgroup per CU, 256-register budget): the global loads of the NEXT tile are issued before the sweeps of the current one
// and land in registers of their own, so the memory system works while the VALU does (a tile is ~6 us of fp64 issue at two
// waves per SIMD against ~4 us of load latency under load).  !PERSIST: G = number of tiles, one tile per workgroup.
//
// BLEND (post-smoother of a Picard iteration, level 0): the tile also performs the tail of the iteration on its owned window --
//   nxt = (1-alpha)*cur + alpha*psi_new (ring from bc), partial sum |psi_new-cur|, non-finite flag        newton_solver.py:515-537
//   partial max|r| / sum (r/max)^2 of r = L*nxt - src over the interior                    fusion_kernel_solver_runtime.py:25-51
//   topology partials of nxt for the NEXT iteration (argmax psi, masked argmin |grad psi|, min psi)       fusion_kernel.py:255-355
// -- the work of k_blend_fused, on registers: psi_new never goes to HBM, cur is read once, and the separate pass disappears.
template <int S, bool PROLONG, bool DEFECT, int RB, int NW, int WPE, bool PERSIST, bool BLEND = false>
__global__ void __launch_bounds__(64 * NW, WPE) k_smooth_rows(RowsArgs A) {
    static_assert(RB % 2 == 0 && RB >= 4, "rows per wave must be even (compile-time colours)");
    static_assert(!(BLEND && (DEFECT || PERSIST)), "the Picard tail is an epilogue of the plain / prolongating one-tile-per-workgroup kernel");
    constexpr int H = 2 * S + ((DEFECT || BLEND) ? 2 : 0), LZ = RB * NW, OWNZ = LZ - 2 * H, OWNR = 128 - 2 * H;
    static_assert(OWNZ >= 2, "tile too short for its halo");
    constexpr int NSLOT = DEFECT ? 6 : 4;  // mailbox rows per wave: first.x first.y last.x last.y [second.x second.y]
    constexpr int NBUF = (PERSIST && DEFECT) ? 2 : 1;  // the defect stage reads the mailbox after the last barrier of a tile
    __shared__ __attribute__((aligned(16))) double mail[NBUF * NW * NSLOT * 64];
    __shared__ double red_d[BLEND ? 4 * NW : 1];  // BLEND: per-wave partials (sum |d psi|, max |r|, min psi, sum r^2 scaled)
    __shared__ VI red_vi[BLEND ? 2 * NW : 1];
    __shared__ int red_bad[BLEND ? NW : 1];

    const Lvl& L = A.L;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nz = L.nz, nr = L.nr, ld = L.ld, nz_c = A.nz_c, nr_c = A.nr_c, ld_c = A.ld_c;
    const double omega = A.omega, om1 = A.om1;
    const unsigned T = (unsigned)A.gx * A.gy * A.B, G = gridDim.x;
    unsigned wg = blockIdx.x;
    if (A.xcd_remap & 1) wg = xcd_contiguous(wg, G);
    const double a_ns = L.a_ns, a_c = L.a_c, inv_a_c = L.inv_a_c, lo_a_c = L.lo_a_c;

    struct Raw {  // what a tile loads from global memory, as loaded (clamped addresses; masks are applied at conversion)
        double2 pv[RB], sv[RB + 1], ae, aw, rs, ri, rl, nm;
        double ca[RB / 2 + 1], cb[RB / 2 + 1];
    };
    struct Pos { int bx, by, b, zw, c0; };
    auto pos_of = [&](unsigned t) __attribute__((always_inline)) {
        Pos q;
        // (readfirstlane: the divisions run on the VALU; the member index must be a scalar so that the slot / scale lookups
        //  stay scalar loads -- a vector load there would make the wave wait for every prefetched load behind it)
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        q.bx = __builtin_amdgcn_readfirstlane((int)(t % A.gx));
        const unsigned u = t / A.gx;
        q.by = __builtin_amdgcn_readfirstlane((int)(u % A.gy));
        q.b = __builtin_amdgcn_readfirstlane((int)(u / A.gy));
        q.zw = q.by * OWNZ + 1 - H + w * RB;  // global row of this wave's register row 0 (odd)
        q.c0 = q.bx * OWNR - H + 2 * lane;    // global column of the lane's x element (even)
        return q;
    };
    // Every load of a tile is unconditional (no load sits behind a branch: a conditional load is a phi, a phi is a register
    // copy, and a copy of a register with a load in flight is a wait for that load).  `real` = false issues the same loads
    // on one row of the member (after the last tile of a workgroup: nothing is prefetched, the registers are simply reloaded).
    // The loads of a tile are numbered j = 0 .. NJ-1 and issued in NPART parts (part k: j % NPART == k).  A CU takes in
    // ~25-30 GB/s, so the 17-23 KiB a wave asks for occupy the memory pipeline for microseconds and a wave that issues them in
    // one go stalls in the ISSUE; spread over the colour passes (one part per pass) the issue never blocks for long.
    struct LoadCtx { const double* in; const double* src; const double* ccp; int cc, zlim, zw, jcc, jcc1, ic0, iclim; };
    auto load_ctx = [&](const Pos& q, const bool real) __attribute__((always_inline)) {
        LoadCtx c;
        c.in = A.in.at_scalar(q.b);
        c.src = A.src.at_scalar(q.b);
        c.ccp = PROLONG ? A.coarse.at_scalar(q.b) : nullptr;
        c.cc = min(max(q.c0, 0), ld - 2);  // clamped 16-B aligned column pair (ld >= nr + 1 for odd nr)
        c.zlim = real ? nz - 1 : 0;
        c.zw = q.zw;
        const int jc = max(q.c0, 0) >> 1;
        c.jcc = min(jc, nr_c - 1);
        c.jcc1 = min(jc + 1, nr_c - 1);
        c.ic0 = (q.zw - 1) >> 1;  // the block's rows zw .. zw+RB-1 (zw odd) touch the coarse rows (zw-1)/2 .. (zw-1)/2 + RB/2
        c.iclim = real ? nz_c - 1 : 0;
        return c;
    };
    constexpr int NSV = RB + (DEFECT ? 1 : 0), NTAB = DEFECT ? 6 : 3, NCO = PROLONG ? 2 * (RB / 2 + 1) : 0;
    constexpr int NJ = RB + NSV + NTAB + NCO;
    auto load_job = [&](const LoadCtx& c, Raw& r, auto job) __attribute__((always_inline)) {  // job index is a compile-time constant: registers, not scratch
        constexpr int j = decltype(job)::value;
        if constexpr (j < RB + NSV) {
            // psi and source rows interleaved (psi 0, src 0, psi 1, ...), the extra source row of DEFECT last
            constexpr bool is_psi = (j < 2 * RB) && !(j & 1);
            constexpr int i = (j < 2 * RB) ? (j >> 1) : RB;
            const int gzc = min(max(c.zw + i, 0), c.zlim);
            const long o = (long)gzc * ld + c.cc;
            if constexpr (is_psi) r.pv[i] = *reinterpret_cast<const double2*>(c.in + o);
            else r.sv[i] = *reinterpret_cast<const double2*>(c.src + o);
        } else if constexpr (j < RB + NSV + NTAB) {
            // per-column tables, padded by the host (build_level_tables): the pair (cc, cc+1) is always readable
            constexpr int k = j - (RB + NSV);
            if constexpr (k == 0) r.ae = *reinterpret_cast<const double2*>(L.a_e + c.cc);
            else if constexpr (k == 1) r.aw = *reinterpret_cast<const double2*>(L.a_w + c.cc);
            else if constexpr (k == 2) r.nm = *reinterpret_cast<const double2*>(A.lzs.neg_mu0R + c.cc);  // (the host always passes a table)
            else if constexpr (k == 3) r.rs = *reinterpret_cast<const double2*>(A.dd.r_safe + c.cc);
            else if constexpr (k == 4) r.ri = *reinterpret_cast<const double2*>(A.dd.r_inv + c.cc);
            else r.rl = *reinterpret_cast<const double2*>(A.dd.r_inv_lo + c.cc);
        } else {
            constexpr int k = (j - (RB + NSV + NTAB)) >> 1;
            const int icc = min(max(c.ic0 + k, 0), c.iclim);
            if constexpr ((j - (RB + NSV + NTAB)) & 1) r.cb[k] = c.ccp[(long)icc * ld_c + c.jcc1];
            else r.ca[k] = c.ccp[(long)icc * ld_c + c.jcc];
        }
    };
    auto issue_part = [&](const LoadCtx& c, Raw& r, auto part, auto nparts) __attribute__((always_inline)) {
        constexpr int K = decltype(part)::value, NP = decltype(nparts)::value;
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (([&] { if constexpr (J % NP == K) load_job(c, r, std::integral_constant<int, J>{}); }()), ...);
        }(std::make_integer_sequence<int, NJ>{});
    };
    auto issue = [&](const Pos& q, Raw& r, const bool real) __attribute__((always_inline)) {
        const LoadCtx c = load_ctx(q, real);
        issue_part(c, r, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    };

    // wave-uniform words of a tile that come from scalar loads: fetched one tile ahead as well (an s_load that misses the
    // scalar cache takes ~0.5 us; eight of them in a row at the top of every tile were 4 us of a 7-us tile)
    struct Meta { int act; double sc; double* out; double* rc; double* ec; };
    auto meta_of = [&](int b) __attribute__((always_inline)) {
        Meta m;
        m.act = A.active ? sload_i32(A.active + b) : 1;
        m.sc = A.lzs.st ? sload_f64(&A.lzs.st[b].jscale) : 1.0;
        m.out = A.out.at_scalar(b);
        m.rc = DEFECT ? A.coarse.at_scalar(b) : nullptr;
        m.ec = DEFECT ? A.dd.e.at_scalar(b) : nullptr;
        return m;
    };
    Raw cur;  // converted into the working registers at the top of a tile, then refilled with the next tile's loads
    if (wg >= T) return;
    issue(pos_of(wg), cur, true);
    Meta mcur = meta_of(pos_of(wg).b);
    int parity = 0;
    [[maybe_unused]] int tile_iter = 0;
    for (unsigned t = wg; t < T; t += G, parity ^= (NBUF - 1), ++tile_iter) {
        const Pos q = pos_of(t);
        const int by_ = q.by, b = q.b, zw = q.zw, c0 = q.c0, r0 = q.bx * OWNR - H;
        double* out = mcur.out;
        // owned window of the tile: local rows [H, LZ-H) (tile row 0 also the ring row H-1), local columns [H, 128-H)
        const int own_lo = (by_ == 0) ? H - 1 : H;
        const bool colx = 2 * lane >= H && 2 * lane < 128 - H && c0 < nr, coly = 2 * lane + 1 >= H && 2 * lane + 1 < 128 - H && c0 + 1 < nr;
        const bool more = PERSIST && t + G < T;
        if (!mcur.act) {  // frozen member: pass the field through (workgroup-uniform)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int lz = w * RB + i, gz = zw + i;
                if (lz >= own_lo && lz < LZ - H && gz >= 0 && gz < nz) {
                    const long o = (long)gz * ld + c0;
                    if (colx) out[o] = cur.pv[i].x;
                    if (coly) out[o + 1] = cur.pv[i].y;
                }
            }
            if (!PERSIST) break;
            const Pos qn = pos_of(more ? t + G : t);
            issue(qn, cur, more);
            mcur = meta_of(qn.b);
            continue;
        }
        ROWS_STAMP(0);
        const bool okx = c0 >= 0 && c0 < nr, oky = c0 + 1 >= 0 && c0 + 1 < nr;
        // Dirichlet ring / out-of-domain columns are never updated (tile-edge columns 0 / 127 may be: their values are halo garbage)
        const bool updx = c0 >= 1 && c0 <= nr - 2, updy = c0 + 1 >= 1 && c0 + 1 <= nr - 2;
        const int mkx = updx ? -1 : 0, mky = updy ? -1 : 0;
        const bool strip_interior = r0 >= 1 && r0 + 127 <= nr - 2;  // workgroup-uniform: every lane's columns are updatable
        double lz_sc = 1.0, lz_n0 = 1.0, lz_n1 = 1.0;
        if (A.lzs.st) {
            lz_sc = mcur.sc;
            lz_n0 = okx ? cur.nm.x : 0.0;
            lz_n1 = oky ? cur.nm.y : 0.0;
        }
        const double aeX = updx ? cur.ae.x : 0.0, awX = updx ? cur.aw.x : 0.0, aeY = updy ? cur.ae.y : 0.0, awY = updy ? cur.aw.y : 0.0;
        // ---- conversion of the loaded rows: out-of-domain values are zeroed, lazy source scaled, coarse correction added
        double px[RB], py[RB], sx[RB + 1], sy[RB + 1];  // (sx/sy[RB]: source of the row below the block, DEFECT only)
        {
            const int h_end = PROLONG ? min(2 * (nr_c - 1), nr - 1) : 0, v_end = PROLONG ? min(2 * (nz_c - 1), nz - 1) : 0;
#pragma unroll
            for (int i = 0; i < RB + (DEFECT ? 1 : 0); ++i) {
                const int gz = zw + i;
                const bool rowok = gz >= 0 && gz < nz;
                double qx = (rowok && okx) ? cur.sv[i].x : 0.0, qy = (rowok && oky) ? cur.sv[i].y : 0.0;
                if (A.lzs.st) {  // wave-uniform; expression of k_source
                    const double tx_ = qx * lz_sc, ty_ = qy * lz_sc;
                    qx = (rowok && okx) ? lz_n0 * (lz_sc != 0.0 ? tx_ : 0.0) : 0.0;
                    qy = (rowok && oky) ? lz_n1 * (lz_sc != 0.0 ? ty_ : 0.0) : 0.0;
                }
                sx[i] = qx;
                sy[i] = qy;
                if (i >= RB) continue;
                double vx = (rowok && okx) ? cur.pv[i].x : 0.0, vy = (rowok && oky) ? cur.pv[i].y : 0.0;
                if (PROLONG && rowok) {
                    // bilinear prolongation, core/multigrid_solve.py:117-143 (same expressions as prolong_at)
                    double ex, ey;
                    if (i & 1) {  // gz even: coarse row gz/2 = ic0 + (i+1)/2
                        const double a0 = cur.ca[(i + 1) / 2], a1 = cur.cb[(i + 1) / 2];
                        ex = a0;
                        ey = (c0 + 1 >= h_end) ? 0.0 : 0.5 * (a0 + a1);
                    } else if (gz >= v_end) {
                        ex = 0.0;
                        ey = 0.0;
                    } else {  // gz odd: coarse rows ic0 + i/2 and + 1
                        const double a0 = cur.ca[i / 2], a1 = cur.cb[i / 2], b0 = cur.ca[i / 2 + 1], b1 = cur.cb[i / 2 + 1];
                        ex = 0.5 * (a0 + b0);
                        ey = (c0 + 1 >= h_end) ? 0.0 : 0.25 * (a0 + b0 + a1 + b1);
                    }
                    if (okx) vx = vx + ex;
                    if (oky) vy = vy + ey;
                }
                px[i] = vx;
                py[i] = vy;
            }
        }
        // the next tile's loads go out now and stay in flight during the sweeps (no load is waited for until the next conversion)
        double rsx = 1.0, rsy = 1.0, rix = 1.0, riy = 1.0, rlx = 0.0, rly = 0.0;
        if (DEFECT) {
            rsx = updx ? cur.rs.x : 1.0; rsy = updy ? cur.rs.y : 1.0; rix = updx ? cur.ri.x : 1.0; riy = updy ? cur.ri.y : 1.0;
            rlx = updx ? cur.rl.x : 0.0; rly = updy ? cur.rl.y : 0.0;
        }
        // prefetch of the next tile (the old contents of `cur` are consumed above this line): its scalar words now, its vector
        // loads in 2S parts behind the barriers of the colour passes
        constexpr int NP = 2 * S;
        LoadCtx nctx{};
        Meta mnext = mcur;
        if (PERSIST) {
            const Pos qn = pos_of(more ? t + G : t);
            nctx = load_ctx(qn, more && !(A.xcd_remap & 4));  // (bit 2: timing experiment, loads of one row only)
            mnext = meta_of(qn.b);
        }
        double* const mbase = mail + parity * (NW * NSLOT * 64);
        double* const mb = mbase + (w * NSLOT) * 64 + lane;                            // this wave's mailbox rows
        const double* const mb_up = mbase + (max(w - 1, 0) * NSLOT) * 64 + lane;       // wave above (rows zw-1 ...)
        const double* const mb_dn = mbase + (min(w + 1, NW - 1) * NSLOT) * 64 + lane;  // wave below (rows zw+RB ...)
        mb[0 * 64] = px[0];
        mb[1 * 64] = py[0];
        mb[2 * 64] = px[RB - 1];
        mb[3 * 64] = py[RB - 1];
        __syncthreads();
        ROWS_STAMP(1);

        // one colour pass.  z0 odd, c0 even, RB even: element x of register row i has colour (i + 1) & 1, element y colour i & 1.
        auto colour_pass = [&](auto colour, const int hs) __attribute__((always_inline)) {
            constexpr int c = decltype(colour)::value;
            // rows still needed for the final window: local rows [hs+1, LZ-2-hs], inside the domain interior [1, nz-2]
            const int ilo = max(hs + 1 - w * RB, 1 - zw), ihi = min(LZ - 2 - hs - w * RB, nz - 2 - zw);
            // colour c: rows i with (i + 1) & 1 == c update x, the others update y.  Row 0 / RB-1 need the mailbox of the neighbour wave
            // (c == 0: row 0 updates y, row RB-1 updates x; c == 1: row 0 updates x, row RB-1 updates y)
            const double s_edge = (c == 0) ? mb_up[3 * 64] : mb_up[2 * 64];
            const double n_edge = (c == 0) ? mb_dn[0 * 64] : mb_dn[1 * 64];
            // MASKED: the strip contains ring / out-of-domain columns, whose lanes keep their value (bit select, no branch).
            // (Dependent fp64 chains issue back to back on gfx950 -- tools/ubench/fp64_issue.hip: 2.5 ns per instruction for one
            // wave alone, dependent or not, 2.05 ns with four waves per SIMD -- so the rows need no manual interleaving.)
            auto row = [&](auto idx, auto masked) __attribute__((always_inline)) {
                constexpr int i = decltype(idx)::value;
                constexpr bool MASKED = decltype(masked)::value;
                if (((i + 1) & 1) == c) {          // x element: W = y of lane-1, E = own y
                    const double W = lane_below(py[i]), E = py[i];
                    const double Sv = (i == 0) ? s_edge : px[i > 0 ? i - 1 : 0], N = (i == RB - 1) ? n_edge : px[i < RB - 1 ? i + 1 : 0];
                    const double gsv = divc4(aeX * E + awX * W + a_ns * Sv + a_ns * N - sx[i], a_c, inv_a_c, lo_a_c);
                    const double nw = om1 * px[i] + omega * gsv;
                    px[i] = MASKED ? bitsel(mkx, nw, px[i]) : nw;
                } else {                           // y element: W = own x, E = x of lane+1
                    const double W = px[i], E = lane_above(px[i]);
                    const double Sv = (i == 0) ? s_edge : py[i > 0 ? i - 1 : 0], N = (i == RB - 1) ? n_edge : py[i < RB - 1 ? i + 1 : 0];
                    const double gsv = divc4(aeY * E + awY * W + a_ns * Sv + a_ns * N - sy[i], a_c, inv_a_c, lo_a_c);
                    const double nw = om1 * py[i] + omega * gsv;
                    py[i] = MASKED ? bitsel(mky, nw, py[i]) : nw;
                }
            };
            auto rows = [&](auto masked) __attribute__((always_inline)) {
                if (ilo <= 0 && ihi >= RB - 1) {  // interior wave: every row takes part; the mailbox rows (0, RB-1) in the last quad
                    [&]<int... I>(std::integer_sequence<int, I...>) { (row(std::integral_constant<int, I + 1>{}, masked), ...); }(std::make_integer_sequence<int, RB - 2>{});
                    row(std::integral_constant<int, 0>{}, masked);
                    row(std::integral_constant<int, RB - 1>{}, masked);
                } else {
                    [&]<int... I>(std::integer_sequence<int, I...>) {
                        ((I >= ilo && I <= ihi ? row(std::integral_constant<int, I>{}, masked) : (void)0), ...);
                    }(std::make_integer_sequence<int, RB>{});
                }
            };
            if (strip_interior) rows(std::false_type{});
            else rows(std::true_type{});
            if (c == 0) { mb[1 * 64] = py[0]; mb[2 * 64] = px[RB - 1]; }
            else { mb[0 * 64] = px[0]; mb[3 * 64] = py[RB - 1]; }
        };
        if constexpr (PERSIST) {
            [&]<int... SW>(std::integer_sequence<int, SW...>) {
                (([&] {
                     colour_pass(std::integral_constant<int, 0>{}, 2 * SW);
                     __syncthreads();
                     ROWS_STAMP(2 + 2 * SW);
                     if (PERSIST) {
                         __builtin_amdgcn_sched_barrier(0);
                         issue_part(nctx, cur, std::integral_constant<int, 2 * SW>{}, std::integral_constant<int, NP>{});
                         __builtin_amdgcn_sched_barrier(0);
                     }
                     colour_pass(std::integral_constant<int, 1>{}, 2 * SW + 1);
                     if (DEFECT && SW == S - 1) { mb[4 * 64] = px[1]; mb[5 * 64] = py[1]; }
                     __syncthreads();
                     ROWS_STAMP(3 + 2 * SW);
                     if (PERSIST) {
                         __builtin_amdgcn_sched_barrier(0);
                         issue_part(nctx, cur, std::integral_constant<int, 2 * SW + 1>{}, std::integral_constant<int, NP>{});
                         __builtin_amdgcn_sched_barrier(0);
                     }
                 }()),
                 ...);
            }(std::make_integer_sequence<int, S>{});
        } else {
#pragma unroll 1
            for (int sw = 0; sw < S; ++sw) {
                colour_pass(std::integral_constant<int, 0>{}, 2 * sw);
                __syncthreads();
                ROWS_STAMP(2 + 2 * sw);
                colour_pass(std::integral_constant<int, 1>{}, 2 * sw + 1);
                if (DEFECT && sw == S - 1) { mb[4 * 64] = px[1]; mb[5 * 64] = py[1]; }
                __syncthreads();
                ROWS_STAMP(3 + 2 * sw);
            }
        }
        if constexpr (BLEND) {
            const BlendArgs& ba = A.ba;
            const double* curp = ba.cur.at_scalar(b);
            const double* bcp = ba.bc.at_scalar(b);
            const int cc2 = min(max(c0, 0), ld - 2);
            double2 cv[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int gzc = min(max(zw + i, 0), nz - 1);
                cv[i] = *reinterpret_cast<const double2*>(curp + (long)gzc * ld + cc2);
            }
            double dsum = 0.0, minpsi = INFINITY, rmx = 0.0;
            int bad = 0;
            VI mx{0.0, -1}, mn{0.0, -1};
            // (1) blend every register row (the rows around the owned window are needed as stencil neighbours); statistics and
            //     the store only for owned points
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int lz = w * RB + i, gz = zw + i;
                if (gz < 0 || gz >= nz) continue;  // wave-uniform (out-of-domain rows hold zeros and stay so)
                const bool rowring = gz == 0 || gz == nz - 1;
                const bool rowown = lz >= own_lo && lz < LZ - H;
                double vx = 0.0, vy = 0.0;
                if (okx) {
                    const bool ring = rowring || c0 == 0 || c0 == nr - 1;
                    const double pn = ring ? bcp[(long)gz * ld + c0] : px[i];
                    const double pc = cv[i].x;
                    vx = ba.om_alpha * pc + ba.alpha * pn;
                    if (rowown && colx) {
                        if (!finite_d(pn)) bad = 1;
                        dsum += fabs(pn - pc);
                        if (vx == vx) mx = vi_max(mx, VI{vx, (long)gz * nr + c0});
                        if (finite_d(vx)) minpsi = fmin(minpsi, vx);
                    }
                }
                if (oky) {
                    const bool ring = rowring || c0 + 1 == 0 || c0 + 1 == nr - 1;
                    const double pn = ring ? bcp[(long)gz * ld + c0 + 1] : py[i];
                    const double pc = cv[i].y;
                    vy = ba.om_alpha * pc + ba.alpha * pn;
                    if (rowown && coly) {
                        if (!finite_d(pn)) bad = 1;
                        dsum += fabs(pn - pc);
                        if (vy == vy) mx = vi_max(mx, VI{vy, (long)gz * nr + c0 + 1});
                        if (finite_d(vy)) minpsi = fmin(minpsi, vy);
                    }
                }
                px[i] = vx;
                py[i] = vy;
                if (rowown) {
                    const long o = (long)gz * ld + c0;
                    if (colx && coly) {
                        *reinterpret_cast<double2*>(out + o) = double2{vx, vy};
                    } else {
                        if (colx) out[o] = vx;
                        if (coly) out[o + 1] = vy;
                    }
                }
            }
            // (2) nxt rows across the wave boundaries (every pass-time reader of the mailbox is behind the last barrier)
            mb[0 * 64] = px[0];
            mb[1 * 64] = py[0];
            mb[2 * 64] = px[RB - 1];
            mb[3 * 64] = py[RB - 1];
            __syncthreads();
            const double xm1 = mb_up[2 * 64], ym1 = mb_up[3 * 64], xr0 = mb_dn[0 * 64], yr0 = mb_dn[1 * 64];
            // (3) residual r = L*nxt - src (kept in the source registers for the rescaled sum) and |grad psi| of the divertor rows
            const double rsx = updx ? A.dd.r_safe[c0] : 1.0, rix = updx ? A.dd.r_inv[c0] : 1.0, rlx = updx ? A.dd.r_inv_lo[c0] : 0.0;
            const double rsy = updy ? A.dd.r_safe[c0 + 1] : 1.0, riy = updy ? A.dd.r_inv[c0 + 1] : 1.0, rly = updy ? A.dd.r_inv_lo[c0 + 1] : 0.0;
            const bool intx = colx && updx, inty = coly && updy;  // owned interior columns
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int lz = w * RB + i, gz = zw + i;
                double r_x = 0.0, r_y = 0.0;
                if (lz >= own_lo && lz < LZ - H && gz >= 0 && gz < nz) {  // wave-uniform: an owned row
                    const double cx = px[i], cy = py[i];
                    const double sxv = (i == 0) ? xm1 : px[i > 0 ? i - 1 : 0], syv = (i == 0) ? ym1 : py[i > 0 ? i - 1 : 0];
                    const double nxv = (i == RB - 1) ? xr0 : px[i < RB - 1 ? i + 1 : 0], nyv = (i == RB - 1) ? yr0 : py[i < RB - 1 ? i + 1 : 0];
                    const double wx = lane_below(cy), ey = lane_above(cx);  // W of x = y of lane-1; E of y = x of lane+1
                    if (gz >= 1 && gz <= nz - 2) {
                        {
                            const double d2r = divc4(cy - 2.0 * cx + wx, A.dd.dr2, A.dd.inv_dr2, A.dd.lo_dr2);
                            const double d1r = divc4(cy - wx, A.dd.two_dr, A.dd.inv_two_dr, A.dd.lo_two_dr);
                            const double d2z = divc4(nxv - 2.0 * cx + sxv, A.dd.dz2, A.dd.inv_dz2, A.dd.lo_dz2);
                            const double lpsi = d2r - divc4(d1r, rsx, rix, rlx) + d2z;
                            const double r = sanitize1(lpsi - sx[i]);
                            if (intx) { r_x = r; rmx = fmax(rmx, fabs(r)); }
                        }
                        {
                            const double d2r = divc4(ey - 2.0 * cy + cx, A.dd.dr2, A.dd.inv_dr2, A.dd.lo_dr2);
                            const double d1r = divc4(ey - cx, A.dd.two_dr, A.dd.inv_two_dr, A.dd.lo_two_dr);
                            const double d2z = divc4(nyv - 2.0 * cy + syv, A.dd.dz2, A.dd.inv_dz2, A.dd.lo_dz2);
                            const double lpsi = d2r - divc4(d1r, rsy, riy, rly) + d2z;
                            const double r = sanitize1(lpsi - sy[i]);
                            if (inty) { r_y = r; rmx = fmax(rmx, fabs(r)); }
                        }
                    }
                    if (gz < ba.n_mask_rows) {  // np.gradient + hypot (fusion_kernel.py:276-288); wave-uniform
                        const double psx = nan_to_num300(cx), psy = nan_to_num300(cy);
                        const double nX = nan_to_num300(nxv), sX = nan_to_num300(sxv), nY = nan_to_num300(nyv), sY = nan_to_num300(syv);
                        const double wX = nan_to_num300(wx), eY = nan_to_num300(ey);
                        double gzx, gzy, grx, gry;
                        if (gz == 0) { gzx = (nX - psx) / ba.dz; gzy = (nY - psy) / ba.dz; }
                        else if (gz == nz - 1) { gzx = (psx - sX) / ba.dz; gzy = (psy - sY) / ba.dz; }
                        else { gzx = (nX - sX) / (2.0 * ba.dz); gzy = (nY - sY) / (2.0 * ba.dz); }
                        // x element: W = y of lane-1, E = own y; y element: W = own x, E = x of lane+1
                        if (c0 == 0) grx = (psy - psx) / ba.dr;
                        else if (c0 == nr - 1) grx = (psx - wX) / ba.dr;
                        else grx = (psy - wX) / (2.0 * ba.dr);
                        if (c0 + 1 == 0) gry = (eY - psy) / ba.dr;
                        else if (c0 + 1 == nr - 1) gry = (psy - psx) / ba.dr;
                        else gry = (eY - psx) / (2.0 * ba.dr);
                        if (colx && okx) {
                            const double bm = hypot(grx, gzx);
                            if (finite_d(bm)) mn = vi_min(mn, VI{bm, (long)gz * nr + c0});
                        }
                        if (coly && oky) {
                            const double bm = hypot(gry, gzy);
                            if (finite_d(bm)) mn = vi_min(mn, VI{bm, (long)gz * nr + c0 + 1});
                        }
                    }
                }
                sx[i] = r_x;
                sy[i] = r_y;
            }
            // (4) workgroup reductions in a fixed order: lanes by DPP tree, then waves 0 .. NW-1
            dsum = wave_sum(dsum);
            rmx = wave_max(rmx);
            minpsi = -wave_max(-minpsi);
            mx = wave_vi<true>(mx);
            mn = wave_vi<false>(mn);
            bad = __any(bad) ? 1 : 0;
            if (lane == 0) { red_d[w] = dsum; red_d[NW + w] = rmx; red_d[2 * NW + w] = minpsi; red_vi[w] = mx; red_vi[NW + w] = mn; red_bad[w] = bad; }
            __syncthreads();
            double bmax = 0.0;
#pragma unroll
            for (int k = 0; k < NW; ++k) bmax = fmax(bmax, red_d[NW + k]);
            // overflow-safe sum of squares, (max, sum (r/max)^2) with the workgroup max as in k_resid_norms (one reciprocal per lane)
            double s2 = 0.0;
            if (bmax > 0.0) {
                const double ib = 1.0 / bmax;
#pragma unroll
                for (int i = 0; i < RB; ++i) { const double qx = sx[i] * ib, qy = sy[i] * ib; s2 += qx * qx; s2 += qy * qy; }
            }
            s2 = wave_sum(s2);
            if (lane == 0) red_d[3 * NW + w] = s2;
            __syncthreads();
            if (threadIdx.x == 0) {
                double a = 0.0, c2 = 0.0, mp = INFINITY;
                VI vmx = red_vi[0], vmn = red_vi[NW];
                int bd = 0;
                for (int k = 0; k < NW; ++k) {
                    a += red_d[k];
                    c2 += red_d[3 * NW + k];
                    mp = fmin(mp, red_d[2 * NW + k]);
                    bd |= red_bad[k];
                    if (k) { vmx = vi_max(vmx, red_vi[k]); vmn = vi_min(vmn, red_vi[NW + k]); }
                }
                const long o = (long)b * ba.nblk + (long)by_ * A.gx + q.bx;
                ba.psum[o] = a;
                ba.pbad[o] = bd;
                ba.rmax[o] = bmax;
                ba.rsum[o] = c2;
                TopoPart tp;
                tp.maxv = vmx.v; tp.maxi = vmx.i; tp.minb = vmn.v; tp.mini = vmn.i; tp.minpsi = mp; tp.pad = 0;
                ba.topo[o] = tp;
            }
        } else {
        // ---- store the owned window
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int lz = w * RB + i, gz = zw + i;
            if (lz >= own_lo && lz < LZ - H && gz < nz) {  // wave-uniform
                const long o = (long)gz * ld + c0;
                if (colx && coly) {
                    *reinterpret_cast<double2*>(out + o) = double2{px[i], py[i]};
                } else {
                    if (colx) out[o] = px[i];
                    if (coly) out[o + 1] = py[i];
                }
            }
        }
        }
        ROWS_STAMP(8);
        if (DEFECT) {
            // (1) defect d = -(L*psi - src) of the block's rows 0 .. RB (row RB belongs to the wave below: the coarse row of register
            //     row RB-1 needs it), both elements, into the source registers; psi rows -1, RB, RB+1 come from the mailboxes
            const double xm1 = mb_up[2 * 64], ym1 = mb_up[3 * 64];
            const double xr0 = mb_dn[0 * 64], yr0 = mb_dn[1 * 64], xr1 = mb_dn[4 * 64], yr1 = mb_dn[5 * 64];
            // tile-edge columns have no E/W neighbour inside the tile: their defect is not used (coarse points own columns [H, 128-H))
            const bool dokx = updx && lane >= 1, doky = updy && lane <= 62;
#pragma unroll
            for (int i = 0; i <= RB; ++i) {
                const int lz = w * RB + i, gz = zw + i;
                double dx = -0.0, dy = -0.0;
                if (lz >= H - 1 && lz <= LZ - H && gz >= 1 && gz <= nz - 2) {  // wave-uniform
                    const double cx = (i < RB) ? px[i < RB ? i : 0] : xr0, cy = (i < RB) ? py[i < RB ? i : 0] : yr0;
                    const double sxv = (i == 0) ? xm1 : ((i - 1 < RB) ? px[(i >= 1 && i - 1 < RB) ? i - 1 : 0] : xr0);
                    const double syv = (i == 0) ? ym1 : ((i - 1 < RB) ? py[(i >= 1 && i - 1 < RB) ? i - 1 : 0] : yr0);
                    const double nxv = (i + 1 < RB) ? px[i + 1 < RB ? i + 1 : 0] : ((i + 1 == RB) ? xr0 : xr1);
                    const double nyv = (i + 1 < RB) ? py[i + 1 < RB ? i + 1 : 0] : ((i + 1 == RB) ? yr0 : yr1);
                    {   // x element
                        const double W = lane_below(cy), E = cy;
                        const double d2r = divc4(E - 2.0 * cx + W, A.dd.dr2, A.dd.inv_dr2, A.dd.lo_dr2);
                        const double d1r = divc4(E - W, A.dd.two_dr, A.dd.inv_two_dr, A.dd.lo_two_dr);
                        const double d2z = divc4(nxv - 2.0 * cx + sxv, A.dd.dz2, A.dd.inv_dz2, A.dd.lo_dz2);
                        const double lpsi = d2r - divc4(d1r, rsx, rix, rlx) + d2z;
                        if (dokx) dx = -(lpsi - sx[i]);
                    }
                    {   // y element
                        const double W = cx, E = lane_above(cx);
                        const double d2r = divc4(E - 2.0 * cy + W, A.dd.dr2, A.dd.inv_dr2, A.dd.lo_dr2);
                        const double d1r = divc4(E - W, A.dd.two_dr, A.dd.inv_two_dr, A.dd.lo_two_dr);
                        const double d2z = divc4(nyv - 2.0 * cy + syv, A.dd.dz2, A.dd.inv_dz2, A.dd.lo_dz2);
                        const double lpsi = d2r - divc4(d1r, rsy, riy, rly) + d2z;
                        if (doky) dy = -(lpsi - sy[i]);
                    }
                }
                sx[i] = dx;
                sy[i] = dy;
            }
            ROWS_STAMP(9);
            // (2) full-weighting restriction at the owned even/even points: even global rows = odd register rows, x elements
            double* rc = mcur.rc;
            double* ec = mcur.ec;
            const int jc = c0 >> 1;
            const bool cown = 2 * lane >= H && 2 * lane < 128 - H && c0 >= 0 && jc < nr_c;
            if (by_ == 0 && w * RB <= H - 1 && H - 1 < w * RB + RB) {  // ring row 0 of the coarse grid (register row H-1 of tile row 0)
                if (cown) { rc[jc] = -0.0; ec[jc] = 0.0; }
            }
#pragma unroll
            for (int i = 1; i < RB; i += 2) {
                const int lz = w * RB + i, gz = zw + i, ic = gz >> 1;
                if (lz >= H && lz < LZ - H && gz >= 0 && ic < nz_c) {  // wave-uniform
                    double v = -0.0;
                    const double lf = lane_below(sy[i]), ul = lane_below(sy[i - 1]), dl = lane_below(sy[i + 1]);
                    if (ic >= 1 && ic <= nz_c - 2 && jc >= 1 && jc <= nr_c - 2)
                        v = fw9(sx[i], sx[i - 1], sx[i + 1], lf, sy[i], ul, sy[i - 1], dl, sy[i + 1]);
                    if (cown) {
                        rc[(long)ic * ld_c + jc] = v;
                        ec[(long)ic * ld_c + jc] = 0.0;
                    }
                }
            }
        }
        ROWS_STAMP(10);
        if (!PERSIST) break;
        mcur = mnext;
    }
}

}  // namespace gs

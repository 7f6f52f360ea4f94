// scpn_hip.hip -- host driver + C ABI of libscpn_hip.so (see include/scpn_hip.h).
//
// Host responsibilities: multigrid hierarchy and its 1-D coefficient tables (reference
// arithmetic, bit-exact), kernel sequencing of the V-cycle and of the Picard iteration, polling the
// device-side convergence word.  All field data and all per-member scalars stay in HBM between
// upload and download.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/scpn_hip.h"
#include "gs_kernels.hip.h"
#include "newton_kernels.hip.h"
#include "comm_rccl.hip.h"

using namespace gs;

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int g_device = 0;
// Every resident handle owns two streams (one per member group) and the HIP runtime multiplexes streams onto 4 hardware queues by
// default: with a second handle alive in the process two group streams land on one queue and stop overlapping (measured: the
// Newton lane 37.5 -> 50.7 ms per ensemble solve).  GPU_MAX_HW_QUEUES=8 avoids that, but it is a process-wide setting read when the
// HIP runtime initialises, so this library does NOT touch the environment of its host application: bench.py sets the variable itself
// before its first device call, and INTEGRATION.md tells an embedding application to do the same.
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) {                                                                        \
            throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e__));              \
        }                                                                                               \
    } while (0)

// every kernel launch is followed by hipGetLastError(): a rejected launch (dynamic LDS over budget, wrong code object, bad
// grid) raises instead of leaving stale fields behind an SCPN_OK; API_END maps the exception to SCPN_E_DEVICE
#define SCPN_LAUNCH(...)                                                                                 \
    do {                                                                                                \
        hipLaunchKernelGGL(__VA_ARGS__);                                                                \
        hipError_t le__ = hipGetLastError();                                                            \
        if (le__ != hipSuccess) throw std::runtime_error(std::string("kernel launch failed: ") + hipGetErrorString(le__)); \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline dim3 grid2d(int nr, int nz, int B) { return dim3(cdiv(nr, 64), cdiv(nz, 4), B); }
static const dim3 BLK2D(64, 4, 1);

// numpy-compatible add.reduce (8192-element chunks of pairwise sums), used for the seed integral
static double np_pairwise(const double* a, long n) {
    if (n < 8) {
        double r = 0.0;
        for (long i = 0; i < n; ++i) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}
static double np_sum(const double* a, long n) {
    double s = 0.0;
    for (long i = 0; i < n; i += 8192) s = s + np_pairwise(a + i, std::min<long>(8192, n - i));
    return s;
}

// ------------------------------------------------------------------------------------------ device memory
template <typename T>
struct DBuf {
    T* p = nullptr;
    size_t n = 0;
    void alloc(size_t count) {
        if (count <= n && p) return;
        release();
        HIP_TRY(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)));
        n = count;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DBuf() { release(); }
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
};

struct LevelHost {
    int nz, nr, ld;
    long stride;
    double dr, dz;
    std::vector<double> r;  // 1-D R of this level (interior columns follow the full-weighting of the finer R)
    DBuf<double> a_e, a_w, r_safe, r_inv, r_inv_lo;
    DBuf<double> psi, psi2, rhs;  // level >= 1: correction e (+ ping-pong twin) and restricted defect
    Lvl dev{};
};
struct Hier {
    std::vector<LevelHost*> lv;
    int lc = -1;          // first level handled by the in-LDS coarse V-cycle kernel (-1: none)
    CoarseV cv{};
    size_t cv_lds = 0;    // dynamic LDS bytes of k_vcycle_coarse
    double* cv_tabs = nullptr;  // the coefficient tables of the levels it holds, packed in its LDS layout (one coalesced copy per launch)
};

static int pitch_of(int nr) { return ((nr + 7) / 8) * 8; }
static long stride_of(int nz, int ld) { return (((long)nz * ld + 31) / 32) * 32; }

// ------------------------------------------------------------------------------------------ solver handle
struct ProfEntry {
    hipEvent_t a, b;
    int id;
    long launches;
};

struct Solver {
    int nr = 0, nz = 0, B = 0, ld = 0;
    int device = 0;  // the HIP device it was created on
    int ensemble_B = 0;  // members of the whole ensemble this stream group belongs to (0: it is the whole ensemble)
    long stride = 0;
    double dr = 0, dz = 0;
    std::vector<double> r_axis, z_axis;
    hipStream_t stream = nullptr;
    bool owns_stream = true;  // false: the shared stream of the stateless operators (TempSolver)

    // multigrid hierarchies keyed by min_grid
    std::map<int, Hier> hier;
    const double* lazy_src = nullptr;  // base of the level-0 source field while it holds J_raw (lazy-source Picard iterations)
    bool lazy_jfly = false;            // ... or, source on the fly (LazySrc::jfly): lazy_src is the CURRENT ITERATE the consumers rebuild J_raw from
    DBuf<double> d_snap, d_fluxpart;    // scpn_gs_snapshot_psi's copy of the iterate / scpn_gs_flux_change's partial sums
    bool snap_valid = false;
    const double* jsum_part = nullptr;  // != nullptr: k_jsum's partial sums wait for the first lazy tile launch to finish them (LazySrc::part)
    int jsum_nblk = 0;
    bool use_fused = true;  // SCPN_HIP_FUSED=0 selects the one-pass-per-launch kernels (A/B testing)
    // Arithmetic of the resident solver's hot kernels (scpn_gs_set_arith; default from SCPN_HIP_ARITH): 0 = exact, every expression
    // in the reference's operation order (bit-exact operators); 1 = contracted: fused multiply-adds and reciprocal multiplies in the
    // tile smoother / coarse V-cycle / Picard tail -- the same iteration to rounding (<= 1e-10 max|psi| on converged equilibria).
    // The stateless operator API (scpn_mg_*, the dispatcher providers, the legacy ABI) is always exact.
    int arith = 0;
    // fields (B members each)
    DBuf<double> psi_slots;  // 3 ping-pong slots
    DBuf<double> W, W2, src, src2, bc, jphi, ext, old, dense, and_psi, and_res, and_part, and_alpha;
    DBuf<int> and_fb;
    // Newton / GMRES workspace (allocated on first use)
    DBuf<double> nt_ws, nt_V, nt_part;
    DBuf<GmState> nt_gm;
    DBuf<NtMember> nt_st;
    DBuf<int> nt_pbad, nt_cnt;
    int nt_V_vecs = 0;
    bool picard_bufs = false;
    DBuf<double> d_r, d_z, d_inv_mu0R, d_neg_mu0R, d_seed;
    DBuf<Member> d_st;
    DBuf<MgMember> d_mg;
    DBuf<int> d_cur, d_nxt, d_best, d_prev, d_active, d_nactive, d_pbad;
    DBuf<int> d_scr_part, d_scr_zc, d_scr_sa, d_scr_mask;  // per-call masks of the Picard / free-boundary drivers
    DBuf<double> d_scr_diff;
    DBuf<Member> d_scr_st;
    DBuf<double> d_part1, d_part2, d_part3, d_jpart;  // reduction partials [B][nblk]
    DBuf<TopoPart> d_topo;
    DBuf<double> d_scal;
    int* h_nactive = nullptr;  // pinned
    unsigned long long* h_word = nullptr;  // pinned + mapped: {sequence, value} posted by k_post_word
    unsigned long long* d_word = nullptr;  // device alias of h_word
    unsigned word_seq = 0;
    DBuf<int> d_iter_count;  // launches of k_picard_final of the running solve (device side of the progress word h_word[1])
    std::vector<double> i_target;
    std::vector<double> vac_r, vac_z, vac_i;
    int n_vac = 0;
    DBuf<double> vac_field;  // calculate_vacuum_field() of the current coil set (valid while vac_valid; see picard_solve_impl)
    bool vac_valid = false;
    double vac_mu0 = 0.0;
    DBuf<double> d_vac_r, d_vac_z, d_vac_i, d_fb_r, d_fb_z, d_fb_i, d_fb_t;
    std::vector<double> seed_host;
    bool seed_user = false, seed_uploaded = false;
    double seed_rc = NAN, seed_integral = 0.0;
    double mu0_tables = NAN;

    std::vector<int> h_seed_act;  // host image of d_scr_sa (uploaded asynchronously by the seed phase)
    std::vector<Member> h_scr_st; // host image of d_scr_st for the final J_phi pass (uploaded asynchronously)
    // profiling
    bool prof_on = false;
    std::vector<ProfEntry> prof;
    std::vector<std::string> prof_names;

    // partial-array capacity: whatever tiling a reduction kernel uses (1024-point blocks, the Picard tail's windows, the column-pair
    // kernels' row bands -- all three move with -D experiment builds), its per-member partial count fits
    int nblk() const { return std::max({(int)cdiv((long)nz * nr, 1024), nblk2(), nblkj()}); }
    int nblkj() const { return (int)(cdiv(nr, 128) * cdiv(nz, JS_ROWS)); }  // workgroups per member of k_jsum / k_source
    int nblk1() const { return cdiv((long)nz * nr, 1024); }
    int nblk2() const { return cdiv(nr, BL_WR) * cdiv(nz, BL_WZ); }  // tiles per member of the fused Picard tail (k_blend_tile: BL_WZ x BL_WR windows)
    int nblk_int() const { return cdiv((long)(nz - 2) * (nr - 2), 1024); }

    Fld slot_fld(const int* slot) { return Fld{psi_slots.p, slot, stride, B}; }
    Fld plain(double* p) { return Fld{p, nullptr, stride, B}; }

    ~Solver() {
        for (auto& kv : hier) {
            for (auto* l : kv.second.lv) delete l;
            if (kv.second.cv_tabs) (void)hipFree(kv.second.cv_tabs);
        }
        for (auto& e : prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        if (h_nactive) (void)hipHostFree(h_nactive);
        if (h_word) (void)hipHostFree(h_word);
        if (stream && owns_stream) (void)hipStreamDestroy(stream);
    }
};

static int prof_id(Solver* h, const char* name) {
    for (size_t i = 0; i < h->prof_names.size(); ++i)
        if (h->prof_names[i] == name) return (int)i;
    h->prof_names.push_back(name);
    return (int)h->prof_names.size() - 1;
}
struct ProfScope {
    Solver* h;
    ProfEntry e{};
    bool on;
    ProfScope(Solver* h_, const char* name, long launches) : h(h_), on(h_->prof_on) {
        if (!on) return;
        HIP_TRY(hipEventCreate(&e.a));
        HIP_TRY(hipEventCreate(&e.b));
        e.id = prof_id(h, name);
        e.launches = launches;
        HIP_TRY(hipEventRecord(e.a, h->stream));
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(e.b, h->stream);
        h->prof.push_back(e);
    }
};

// ------------------------------------------------------------------------------------------ hierarchy
static void build_level_tables(LevelHost* L) {
    // core/multigrid_solve.py:181-189 (dr**2 is libm pow in CPython)
    const double dr2 = std::pow(L->dr, 2.0), dz2 = std::pow(L->dz, 2.0);
    std::vector<double> ae(L->nr, 0.0), aw(L->nr, 0.0), rs(L->nr, 1.0), ri(L->nr, 1.0), rl(L->nr, 0.0);
    // second word of a reciprocal: 1 - d*RN(1/d) is exact in one FMA; divided by d it gives 1/d - RN(1/d) to 53 bits
    auto recip_lo = [](double d) { return std::fma(-d, 1.0 / d, 1.0) / d; };
    for (int ir = 0; ir < L->nr; ++ir) {
        double r_safe = std::fmax(L->r[ir], 1e-10);
        double t = 1.0 / ((2.0 * r_safe) * L->dr);
        ae[ir] = 1.0 / dr2 - t;
        aw[ir] = 1.0 / dr2 + t;
        rs[ir] = r_safe;
        ri[ir] = 1.0 / r_safe;
        rl[ir] = recip_lo(r_safe);
    }
    // (+8: the row-block smoother loads the tables as 16-B pairs (c, c+1) from a clamped even column)
    L->a_e.alloc(L->nr + 8);
    L->a_w.alloc(L->nr + 8);
    L->r_safe.alloc(L->nr + 8);
    L->r_inv.alloc(L->nr + 8);
    L->r_inv_lo.alloc(L->nr + 8);
    HIP_TRY(hipMemcpy(L->r_inv.p, ri.data(), sizeof(double) * L->nr, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(L->r_inv_lo.p, rl.data(), sizeof(double) * L->nr, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(L->a_e.p, ae.data(), sizeof(double) * L->nr, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(L->a_w.p, aw.data(), sizeof(double) * L->nr, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(L->r_safe.p, rs.data(), sizeof(double) * L->nr, hipMemcpyHostToDevice));
    Lvl d;
    d.nz = L->nz; d.nr = L->nr; d.ld = L->ld; d.stride = L->stride;
    d.a_ns = 1.0 / dz2;
    d.a_c = 2.0 / dr2 + 2.0 / dz2;
    d.dr2 = dr2; d.dz2 = dz2; d.two_dr = 2.0 * L->dr; d.dr = L->dr; d.dz = L->dz;
    d.inv_a_c = 1.0 / d.a_c; d.inv_dr2 = 1.0 / dr2; d.inv_dz2 = 1.0 / dz2; d.inv_two_dr = 1.0 / d.two_dr;
    d.lo_a_c = recip_lo(d.a_c); d.lo_dr2 = recip_lo(dr2); d.lo_dz2 = recip_lo(dz2); d.lo_two_dr = recip_lo(d.two_dr);
    d.a_e = L->a_e.p; d.a_w = L->a_w.p; d.r_safe = L->r_safe.p; d.r_inv = L->r_inv.p; d.r_inv_lo = L->r_inv_lo.p;
    L->dev = d;
}

// levels of the V-cycle recursion (core/multigrid_solve.py:289-312): level k is the base case
static Hier& get_hier(Solver* h, int min_grid) {
    auto it = h->hier.find(min_grid);
    if (it != h->hier.end()) return it->second;
    Hier H;
    std::vector<LevelHost*>& v = H.lv;
    auto* L0 = new LevelHost();
    L0->nz = h->nz; L0->nr = h->nr; L0->ld = h->ld; L0->stride = h->stride;
    L0->dr = h->dr; L0->dz = h->dz; L0->r = h->r_axis;
    build_level_tables(L0);
    v.push_back(L0);
    while (!(min_grid >= v.back()->nz || min_grid >= v.back()->nr)) {
        LevelHost* f = v.back();
        auto* c = new LevelHost();
        c->nz = (f->nz + 1) / 2;
        c->nr = (f->nr + 1) / 2;
        c->ld = pitch_of(c->nr);
        c->stride = stride_of(c->nz, c->ld);
        c->dr = f->dr * 2.0;
        c->dz = f->dz * 2.0;
        c->r.assign(c->nr, 0.0);
        // restrict_full_weight(r_grid) of a row-constant grid (multigrid_solve.py:76-97, :307)
        for (int jc = 1; jc < c->nr - 1; ++jc) {
            int j = 2 * jc;
            const double ce = f->r[j], lf = f->r[j - 1], rt = f->r[j + 1];
            c->r[jc] = (4.0 * ce + 2.0 * (ce + ce + lf + rt) + (lf + rt + lf + rt)) / 16.0;
        }
        c->r[0] = f->r[0];
        c->r[c->nr - 1] = f->r[f->nr - 1];
        build_level_tables(c);
        c->psi.alloc((size_t)c->stride * h->B);
        c->psi2.alloc((size_t)c->stride * h->B);
        c->rhs.alloc((size_t)c->stride * h->B);
        HIP_TRY(hipMemsetAsync(c->psi.p, 0, sizeof(double) * c->stride * h->B, h->stream));
        HIP_TRY(hipMemsetAsync(c->psi2.p, 0, sizeof(double) * c->stride * h->B, h->stream));
        HIP_TRY(hipMemsetAsync(c->rhs.p, 0, sizeof(double) * c->stride * h->B, h->stream));
        v.push_back(c);
        if (v.size() > 40) break;
    }
    // in-LDS coarse V-cycle: the deepest suffix of levels (starting at >= 1) that fits 150 KiB
    const int k = (int)v.size() - 1;
    if (k >= 1) {
        const size_t budget = 150 * 1024 / sizeof(double);
        int lc = k;
        auto need = [&](int from) {
            size_t tot = 0;
            for (int l = from; l <= k; ++l) tot += 2 * (size_t)v[l]->nz * v[l]->nr + 4 * (size_t)v[l]->nr + 8;
            return tot + (size_t)v[from]->nz * v[from]->nr + 2;  // + the defect scratch of the top level
        };
        while (lc - 1 >= 1 && (k - (lc - 1) + 1) <= CV_MAX && need(lc - 1) <= budget) --lc;
        if (need(lc) <= budget && (k - lc + 1) <= CV_MAX) {
            H.lc = lc;
            CoarseV& V = H.cv;
            V.n = k - lc + 1;
            int off = 0;
            auto al = [&](int nd) { int o = off; off += (nd + 1) & ~1; return o; };
            // the tables of all levels first, contiguous: they are staged by one coalesced copy of a packed global image
            for (int i = 0; i < V.n; ++i) V.off_tab[i] = al(4 * v[lc + i]->nr);
            V.tab_total = off;
            HIP_TRY(hipMalloc(&H.cv_tabs, sizeof(double) * (size_t)std::max(off, 1)));
            HIP_TRY(hipMemset(H.cv_tabs, 0, sizeof(double) * (size_t)std::max(off, 1)));
            for (int i = 0; i < V.n; ++i) {
                LevelHost* L = v[lc + i];
                const double* srcs[4] = {L->dev.a_e, L->dev.a_w, L->dev.r_safe, L->dev.r_inv};
                for (int t = 0; t < 4; ++t)
                    HIP_TRY(hipMemcpy(H.cv_tabs + V.off_tab[i] + (size_t)t * L->nr, srcs[t], sizeof(double) * L->nr, hipMemcpyDeviceToDevice));
            }
            V.tabs = H.cv_tabs;
            for (int i = 0; i < V.n; ++i) {
                LevelHost* L = v[lc + i];
                V.nz[i] = L->nz; V.nr[i] = L->nr;
                V.off_psi[i] = al(L->nz * L->nr);
                V.off_rhs[i] = al(L->nz * L->nr);
                V.a_ns[i] = L->dev.a_ns; V.a_c[i] = L->dev.a_c; V.dr2[i] = L->dev.dr2; V.dz2[i] = L->dev.dz2; V.two_dr[i] = L->dev.two_dr;
                V.inv_a_c[i] = L->dev.inv_a_c; V.lo_a_c[i] = L->dev.lo_a_c; V.inv_dr2[i] = L->dev.inv_dr2; V.inv_dz2[i] = L->dev.inv_dz2; V.inv_two_dr[i] = L->dev.inv_two_dr;
            }
            V.off_scr = al(v[lc]->nz * v[lc]->nr);
            V.ld0 = v[lc]->ld;
            V.stride0 = v[lc]->stride;
            H.cv_lds = (size_t)off * sizeof(double);
        }
    }
    h->hier[min_grid] = H;
    return h->hier[min_grid];
}
static std::vector<LevelHost*>& get_hierarchy(Solver* h, int min_grid) { return get_hier(h, min_grid).lv; }

// ------------------------------------------------------------------------------------------ smoother / V-cycle drivers
struct ProlongSrc {  // coarse-grid correction to add on load (nullptr base: none)
    Fld c{nullptr, nullptr, 0, 0};
    int nz_c = 0, nr_c = 0, ld_c = 0;
};
struct DefectOut {  // coarse rhs / coarse correction produced by the fused defect + restriction (nullptr base: none)
    Fld rhs{nullptr, nullptr, 0, 0}, e{nullptr, nullptr, 0, 0};
    int nz_c = 0, nr_c = 0, ld_c = 0;
};
static bool gauss_seidel_forms() {
    static const bool on = [] { const char* e = std::getenv("SCPN_HIP_GS1"); return e ? std::atoi(e) != 0 : true; }();
    return on;
}
template <int S, bool PROLONG, bool DEFECT, int LZT, int NWT>
static void launch_tile_g(Solver* h, const Lvl& L, Fld out, Fld in, Fld src, double omega, const int* active, const ProlongSrc& ps,
                          const DefectOut* dfo, bool in_zero) {
    static std::once_flag attr_once;  // (two stream-group threads may reach a first launch at the same time)
    // (+ 3 KiB behind the planes: the residual coefficients the waves of a six-row DEFECT tile share, see SHARE_D in the kernel)
    constexpr bool share_d = DEFECT && LZT / NWT >= 6 && NWT >= 6;
    constexpr size_t lds = (size_t)LZT * 128 * sizeof(double) + (share_d ? 6 * 64 * sizeof(double) : 0);
    auto kern_any = &k_smooth_tile<S, PROLONG, DEFECT, LZT, NWT, TILE_WPE>;
    // the instantiation that also finishes the J_raw sum (LazySrc::part) exists for the V-cycle's first level-0 launch only
    constexpr bool HAS_SUMF = S == 3 && DEFECT && !PROLONG;
    auto kern_sum_any = [] {
        if constexpr (!HAS_SUMF) return (decltype(kern_any))nullptr;
        else return &k_smooth_tile<S, PROLONG, DEFECT, LZT, NWT, TILE_WPE, true>;
    }();
    // omega == 1.0 (plain Gauss-Seidel: the multigrid default and the bench's setting): the instantiations whose relaxation step is
    // one fma, for the three-sweep launches of the V-cycle (SCPN_HIP_GS1=0: the general instantiations whatever omega is)
    constexpr bool HAS_GS1 = S == 3;
    auto kern_gs1 = [] {
        if constexpr (!HAS_GS1) return (decltype(kern_any))nullptr;
        else return &k_smooth_tile<S, PROLONG, DEFECT, LZT, NWT, TILE_WPE, false, true>;
    }();
    auto kern_sum_gs1 = [] {
        if constexpr (!(HAS_GS1 && HAS_SUMF)) return (decltype(kern_any))nullptr;
        else return &k_smooth_tile<S, PROLONG, DEFECT, LZT, NWT, TILE_WPE, true, true>;
    }();
    // contracted arithmetic (Solver::arith == 1): the three-sweep launches only -- one- and two-sweep launches stay exact
    constexpr bool HAS_FAST = S == 3;
    auto kern_fast = [] {
        if constexpr (!HAS_FAST) return (decltype(kern_any))nullptr;
        else return &k_smooth_tile<S, PROLONG, DEFECT, LZT, NWT, TILE_WPE, false, false, true>;
    }();
    auto kern_fast_gs1 = [] {
        if constexpr (!HAS_FAST) return (decltype(kern_any))nullptr;
        else return &k_smooth_tile<S, PROLONG, DEFECT, LZT, NWT, TILE_WPE, false, true, true>;
    }();
    auto kern_fast_sum = [] {
        if constexpr (!(HAS_FAST && HAS_SUMF)) return (decltype(kern_any))nullptr;
        else return &k_smooth_tile<S, PROLONG, DEFECT, LZT, NWT, TILE_WPE, true, false, true>;
    }();
    auto kern_fast_sum_gs1 = [] {
        if constexpr (!(HAS_FAST && HAS_SUMF)) return (decltype(kern_any))nullptr;
        else return &k_smooth_tile<S, PROLONG, DEFECT, LZT, NWT, TILE_WPE, true, true, true>;
    }();
    std::call_once(attr_once, [&] {
        for (auto k : {kern_any, kern_sum_any, kern_gs1, kern_sum_gs1, kern_fast, kern_fast_gs1, kern_fast_sum, kern_fast_sum_gs1})
            if (k) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    });
    const bool gs1 = HAS_GS1 && omega == 1.0 && gauss_seidel_forms();
    const bool fast = HAS_FAST && h->arith == 1;
    auto kern = fast ? (gs1 ? kern_fast_gs1 : kern_fast) : (gs1 ? kern_gs1 : kern_any);
    auto kern_sum = fast ? (gs1 ? kern_fast_sum_gs1 : kern_fast_sum) : (gs1 ? kern_sum_gs1 : kern_sum_any);
    constexpr int H = 2 * S + (DEFECT ? 2 : 0);
    static_assert(LZT - 2 * H >= 2, "tile too short for its halo");
    dim3 g(cdiv(L.nr, 128 - 2 * H), cdiv(L.nz, LZT - 2 * H), h->B);
    DefectDst dd;
    Fld cF = ps.c;
    int nz_c = ps.nz_c, nr_c = ps.nr_c, ld_c = ps.ld_c;
    if (DEFECT) {  // residual constants of this level + where the coarse rhs / zeroed coarse correction go
        dd.dr2 = L.dr2; dd.dz2 = L.dz2; dd.two_dr = L.two_dr; dd.inv_dr2 = L.inv_dr2; dd.inv_dz2 = L.inv_dz2;
        dd.inv_two_dr = L.inv_two_dr; dd.r_safe = L.r_safe; dd.r_inv = L.r_inv;
        dd.lo_dr2 = L.lo_dr2; dd.lo_dz2 = L.lo_dz2; dd.lo_two_dr = L.lo_two_dr; dd.r_inv_lo = L.r_inv_lo;
        dd.kd = L.a_c / omega;
        dd.e = dfo->e;
        cF = dfo->rhs; nz_c = dfo->nz_c; nr_c = dfo->nr_c; ld_c = dfo->ld_c;
    }
    static const int xcd_env = [] { const char* e = std::getenv("SCPN_HIP_XCD"); return (e ? std::atoi(e) : 1) != 0 ? 1 : 0; }();  // any non-zero value = on
    LazySrc lzs;
    if (h->lazy_src && src.base == h->lazy_src) {
        lzs.st = h->d_st.p; lzs.neg_mu0R = h->d_neg_mu0R.p;
        if (h->lazy_jfly) { lzs.jfly = 1; lzs.r_axis = h->d_r.p; lzs.inv_mu0R = h->d_inv_mu0R.p; }
        if (h->jsum_part) {  // the first consumer of J_raw after k_jsum finishes the sum
            if (kern_sum) {
                lzs.part = h->jsum_part; lzs.nblk = h->jsum_nblk; lzs.dr = h->dr; lzs.dz = h->dz; lzs.st_rw = h->d_st.p;
            } else {  // (not the launch shape the fused sum is built for: finish it the ordinary way first)
                SCPN_LAUNCH(k_jsum_final, dim3(h->B), dim3(256), 0, h->stream, h->d_st.p, (const double*)h->jsum_part, h->jsum_nblk, h->dr, h->dz);
            }
            h->jsum_part = nullptr;
        }
    }
    SCPN_LAUNCH(lzs.part ? kern_sum : kern, g, dim3(64 * NWT), lds, h->stream, out, in, src, L, omega, 1.0 - omega, active, cF, nz_c, nr_c,
                ld_c, lzs, dd, (xcd_env & 1) | (in_zero ? 2 : 0));
}
template <int S, bool PROLONG, bool DEFECT>
static void launch_tile(Solver* h, const Lvl& L, Fld out, Fld in, Fld src, double omega, const int* active, const ProlongSrc& ps,
                        const DefectOut* dfo, bool in_zero) {
    // Which tile geometry: by the members the chip works on at once (all stream groups of the ensemble).  More than 12: 48 rows x
    // 8 waves (the least halo redundancy that fits three workgroups per CU; with round 4's one-round-trip load phase it wins from 16
    // members on: 178.8 vs 186.3 us per iteration, and ties at 12 and 8); up to 12 (measured at 129^2 ... 1025^2: the launches
    // are then short chains of workgroup lifetimes whatever the grid): 32 x 8, and 24 x 12 (two rows per wave: the shortest
    // lifetime) while a level-0 launch has at most TILE_TINY tiles (two members at 513^2).  One- and two-sweep launches (sweep
    // counts that are not multiples of three) always take 32 x 8.  SCPN_HIP_TILEGEO=24 / 32 / 48 forces a geometry,
    // SCPN_HIP_TILE_MAXB / SCPN_HIP_TILE_TINY move the limits.
    static const int geo_env = [] { const char* e = std::getenv("SCPN_HIP_TILEGEO"); return e ? std::atoi(e) : 0; }();
    static const int mid_maxb = [] { const char* e = std::getenv("SCPN_HIP_TILE_MAXB"); return e ? std::atoi(e) : 12; }();
    static const long tiny_max = [] { const char* e = std::getenv("SCPN_HIP_TILE_TINY"); return e ? std::atol(e) : 220L; }();
    if constexpr (S == 3) {
        const int members = h->ensemble_B > 0 ? h->ensemble_B : h->B;
        const long tiles0 = (long)cdiv(h->nr, 128 - 16) * cdiv(h->nz, 40 - 16) * members;  // (counted in 24-row windows of 112 columns)
        // (SCPN_HIP_TILEGEO_SUB=24 / 32 / 48: another geometry for the levels below the finest one -- experiments)
        static const int geo_sub = [] { const char* e = std::getenv("SCPN_HIP_TILEGEO_SUB"); return e ? std::atoi(e) : 0; }();
        const bool sub_level = L.nz != h->nz || L.nr != h->nr;
        const int geo = (sub_level && geo_sub) ? geo_sub : geo_env ? geo_env : (members > mid_maxb ? 48 : (tiles0 <= tiny_max ? 24 : 32));
        if (geo == 48) return launch_tile_g<S, PROLONG, DEFECT, TILE_BIG_LZ, TILE_BIG_NW>(h, L, out, in, src, omega, active, ps, dfo, in_zero);
        if (geo == 24) return launch_tile_g<S, PROLONG, DEFECT, TILE_TINY_LZ, TILE_TINY_NW>(h, L, out, in, src, omega, active, ps, dfo, in_zero);
    }
    launch_tile_g<S, PROLONG, DEFECT, TILE_MID_LZ, TILE_MID_NW>(h, L, out, in, src, omega, active, ps, dfo, in_zero);
}
// the three production roles of the tile kernel: plain, + prolongation on load, + defect / restriction on store
template <int S>
static void launch_tile_p(Solver* h, const Lvl& L, Fld out, Fld in, Fld src, double omega, const int* active, const ProlongSrc& ps,
                          const DefectOut* dfo, bool in_zero) {
    if (dfo) launch_tile<S, false, true>(h, L, out, in, src, omega, active, ps, dfo, in_zero);
    else if (ps.c.base) launch_tile<S, true, false>(h, L, out, in, src, omega, active, ps, nullptr, in_zero);
    else launch_tile<S, false, false>(h, L, out, in, src, omega, active, ps, nullptr, in_zero);
}

// n_sweeps red-black SOR sweeps of (in, src).  Returns the field that holds the result: `in` itself for the
// in-place kernels, or `in`/`alt` alternately for the out-of-place LDS-tiled kernel (`alt` must not alias `in`).
static std::string lname(const char* kind, int lvl) { return std::string(kind) + "_L" + std::to_string(lvl); }
static Fld smooth(Solver* h, const Lvl& L, Fld in, Fld alt, Fld src, double omega, int n_sweeps, const int* active,
                  bool clip = false, int lvl = 0, const ProlongSrc* prolong = nullptr, const DefectOut* defect = nullptr,
                  bool* in_zero = nullptr) {
    const bool tile_path = !(n_sweeps <= 0 || L.nz < 3 || L.nr < 3) && !clip && (long)L.nz * L.nr > SMALL_MAX && h->use_fused && alt.base != nullptr;
    if (in_zero && *in_zero && !tile_path) throw std::logic_error("smooth: zero-input flag on a path that reads its input");
    if (n_sweeps <= 0 || L.nz < 3 || L.nr < 3) return in;
    const double om1 = 1.0 - omega;
    if (!clip && (long)L.nz * L.nr <= SMALL_MAX) {
        ProfScope ps(h, lname("small_smooth", lvl).c_str(), 1);
        SCPN_LAUNCH(k_small_smooth, dim3(h->B), dim3(256), 0, h->stream, in, src, L, omega, om1, n_sweeps, active);
        return in;
    }
    if (!clip && h->use_fused && alt.base != nullptr) {
        // family name: level + role (pre = ... + fused defect/restriction, post = fused prolongation + ...)
        const std::string fam = lname("smooth_tile", lvl) + (defect ? "_pre" : (prolong ? "_post" : ""));
        ProfScope ps(h, fam.c_str(), (n_sweeps + 2) / 3);
        Fld a = in, b = alt;
        int left = n_sweeps;
        ProlongSrc pso = prolong ? *prolong : ProlongSrc{};  // only the first launch adds the coarse correction
        // *in_zero (input only): `in` is known to be the zero field and nobody has written it -- the first tile launch starts from
        // +0.0 without reading it.  A path that does read `in` is a hard error with the flag set (the std::logic_error above):
        // the caller has not written the zeros
        bool iz = in_zero && *in_zero;
        while (left > 0) {
            const int S = std::min(3, left);
            const DefectOut* dfo = (defect && left - S <= 0) ? defect : nullptr;  // the LAST launch also produces the coarse rhs
            if (S == 3) launch_tile_p<3>(h, L, b, a, src, omega, active, pso, dfo, iz);
            else if (S == 2) launch_tile_p<2>(h, L, b, a, src, omega, active, pso, dfo, iz);
            else launch_tile_p<1>(h, L, b, a, src, omega, active, pso, dfo, iz);
            iz = false;
            pso = ProlongSrc{};
            std::swap(a, b);
            left -= S;
        }
        return a;
    }
    ProfScope ps(h, lname("rb_half", lvl).c_str(), 2L * n_sweeps);
    int ncol = (L.nr - 2 + 1) / 2;
    dim3 g(cdiv(ncol, 64), cdiv(L.nz - 2, 4), h->B);
    for (int s = 0; s < n_sweeps; ++s)
        for (int parity = 0; parity < 2; ++parity) {
            if (clip) SCPN_LAUNCH(k_rb_half<true>, g, BLK2D, 0, h->stream, in, src, L, omega, om1, parity, active);
            else SCPN_LAUNCH(k_rb_half<false>, g, BLK2D, 0, h->stream, in, src, L, omega, om1, parity, active);
        }
    return in;
}

// One V-cycle (core/multigrid_solve.py:252-335) of (in0, src0) on level 0.  in0 is left untouched; the improved
// estimate is produced in bufA or bufB (returned).  `active` (may be null) freezes members.
// every level-0 consumer of the source inside the V-cycle is a tile kernel (pre-smoother with fused defect, post-smoother)
static bool vcycle_lazy_ok(Solver* h, int pre, int post, int min_grid) {
    Hier& H = get_hier(h, min_grid);
    const Lvl& L0 = H.lv[0]->dev;
    const char* e = std::getenv("SCPN_HIP_DEFECT_MINLVL");
    return h->use_fused && H.lv.size() > 1 && pre > 0 && pre <= 3 && post > 0 && (long)L0.nz * L0.nr > SMALL_MAX && !(e && std::atoi(e) > 0);
}
static Fld vcycle(Solver* h, Fld in0, Fld bufA, Fld bufB, Fld src0, double omega, int pre, int post, int min_grid,
                  const int* active) {
    Hier& H = get_hier(h, min_grid);
    auto& lv = H.lv;
    const int k = (int)lv.size() - 1;
    const int B = h->B;
    const bool fused = h->use_fused;
    auto lvf = [&](DBuf<double>& d, int l) { return Fld{d.p, nullptr, lv[l]->stride, B}; };
    std::vector<Fld> cur(k + 1);
    const int lc = (fused && H.lc >= 1) ? H.lc : k + 1;  // levels >= lc run inside k_vcycle_coarse
    auto rhs_of = [&](int l) { return l == 0 ? src0 : lvf(lv[l]->rhs, l); };
    auto other_of = [&](int l, Fld f) {
        if (l == 0) return f.base == bufA.base && f.slot == bufA.slot ? bufB : bufA;
        return f.base == lv[l]->psi.p ? lvf(lv[l]->psi2, l) : lvf(lv[l]->psi, l);
    };
    // the pre-smoother of a tiled level can also emit the coarse rhs (fused defect + restriction)
    static const int defect_min_level = [] { const char* e = std::getenv("SCPN_HIP_DEFECT_MINLVL"); return e ? std::atoi(e) : 0; }();
    auto defect_fusable = [&](int l) {
        return fused && pre > 0 && l < k && l >= defect_min_level && (long)lv[l]->dev.nz * lv[l]->dev.nr > SMALL_MAX;
    };
    // The correction of level l + 1 starts as the zero field.  Whoever smooths it first need not find the zeros in memory if it
    // starts from +0.0 by itself: the coarse kernel always does (it never reads e), and so does the first tile launch of a
    // pre-smoother (zero-input flag).  Then the fused defect stage of level l leaves e unwritten (SCPN_HIP_ZEROIN=0: written and read).
    static const bool zeroin_env = [] { const char* e = std::getenv("SCPN_HIP_ZEROIN"); return !(e && e[0] == '0'); }();
    auto tile_level = [&](int l) { return fused && (long)lv[l]->dev.nz * lv[l]->dev.nr > SMALL_MAX; };
    auto starts_from_zero = [&](int l) {  // level l (>= 1): its first consumer needs no zeros in memory
        if (!zeroin_env) return false;
        if (l == lc) return true;
        return l < lc && l < k && pre > 0 && tile_level(l);
    };
    auto defect_out = [&](int l) {
        DefectOut d;
        d.rhs = lvf(lv[l + 1]->rhs, l + 1); d.e = lvf(lv[l + 1]->psi, l + 1);
        if (starts_from_zero(l + 1)) d.e.base = nullptr;
        d.nz_c = lv[l + 1]->dev.nz; d.nr_c = lv[l + 1]->dev.nr; d.ld_c = lv[l + 1]->dev.ld;
        return d;
    };
    // level 0: first smoothing pass reads in0 and writes a work buffer
    bool defect_done0 = false;
    {
        const Lvl& L0 = lv[0]->dev;
        const int n0 = (k == 0) ? 50 : pre;  // base case: 50 sweeps (multigrid_solve.py:292-293)
        const bool tile0 = fused && (long)L0.nz * L0.nr > SMALL_MAX && n0 > 0;
        if (tile0) {
            DefectOut d0;
            const bool df = k > 0 && defect_fusable(0);
            if (df) d0 = defect_out(0);
            if (n0 <= 3) {
                cur[0] = smooth(h, L0, in0, bufA, src0, omega, n0, active, false, 0, nullptr, df ? &d0 : nullptr);  // in0 -> bufA
            } else {
                Fld r = smooth(h, L0, in0, bufA, src0, omega, 3, active, false, 0);
                cur[0] = smooth(h, L0, r, bufB, src0, omega, n0 - 3, active, false, 0, nullptr, df ? &d0 : nullptr);
            }
            defect_done0 = df;
        } else {
            ProfScope ps(h, "copy", 1);
            SCPN_LAUNCH(k_copy, grid2d(L0.nr, L0.nz, B), BLK2D, 0, h->stream, bufA, in0, L0.nz, L0.nr, L0.ld, (const int*)nullptr);
            cur[0] = bufA;
            if (n0 > 0) cur[0] = smooth(h, L0, bufA, bufB, src0, omega, n0, active);
        }
        if (k == 0) return cur[0];
    }
    for (int l = 0; l < k && l < lc; ++l) {
        bool done = (l == 0) ? defect_done0 : false;
        if (l > 0) {
            const bool df = defect_fusable(l);
            DefectOut dl;
            if (df) dl = defect_out(l);
            // (the correction was left unwritten by level l - 1's fused defect stage iff that stage ran and this level starts from zero)
            bool in_zero = defect_fusable(l - 1) && (l - 1 > 0 || defect_done0) && starts_from_zero(l);
            cur[l] = smooth(h, lv[l]->dev, cur[l], other_of(l, cur[l]), rhs_of(l), omega, pre, active, false, l, nullptr,
                            df ? &dl : nullptr, &in_zero);
            done = df;
        }
        cur[l + 1] = lvf(lv[l + 1]->psi, l + 1);
        if (done) continue;
        ProfScope ps(h, lname("defect_restrict", l).c_str(), 1);
        const Lvl& Lc = lv[l + 1]->dev;
        SCPN_LAUNCH(k_defect_restrict, grid2d(Lc.nr, Lc.nz, B), BLK2D, 0, h->stream, rhs_of(l + 1), cur[l + 1], cur[l], rhs_of(l),
                           lv[l]->dev, Lc, active);
    }
    int top;  // coarsest level solved so far
    if (lc <= k) {
        static std::once_flag attr_once;
        std::call_once(attr_once, [] {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vcycle_coarse<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vcycle_coarse<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vcycle_coarse<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vcycle_coarse<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        });
        ProfScope ps(h, "vcycle_coarse", 1);
        // (omega == 1.0: the instantiation whose relaxation steps are single fmas, as in the tile smoother)
        const bool gs1 = omega == 1.0 && gauss_seidel_forms();
        auto kern = h->arith == 1 ? (gs1 ? &k_vcycle_coarse<true, true> : &k_vcycle_coarse<false, true>)
                                  : (gs1 ? &k_vcycle_coarse<true> : &k_vcycle_coarse<false>);
        SCPN_LAUNCH(kern, dim3(B), dim3(1024), H.cv_lds, h->stream, cur[lc], rhs_of(lc), H.cv, omega, 1.0 - omega, pre, post, active);
        top = lc;
    } else {
        cur[k] = smooth(h, lv[k]->dev, cur[k], other_of(k, cur[k]), rhs_of(k), omega, 50, active, false, k);
        top = k;
    }
    for (int l = top - 1; l >= 0; --l) {
        const Lvl& Lf = lv[l]->dev;
        const Lvl& Lc = lv[l + 1]->dev;
        const bool tile_post = fused && post > 0 && (long)Lf.nz * Lf.nr > SMALL_MAX;
        if (tile_post) {  // prolongation fused into the post-smoother's tile load
            ProlongSrc pso;
            pso.c = cur[l + 1]; pso.nz_c = Lc.nz; pso.nr_c = Lc.nr; pso.ld_c = Lc.ld;
            cur[l] = smooth(h, Lf, cur[l], other_of(l, cur[l]), rhs_of(l), omega, post, active, false, l, &pso);
        } else {
            {
                ProfScope ps(h, lname("prolong_add", l).c_str(), 1);
                SCPN_LAUNCH(k_prolong<true>, grid2d(Lf.nr, Lf.nz, B), BLK2D, 0, h->stream, cur[l], cur[l + 1], Lf.nz, Lf.nr, Lf.ld,
                                   Lc.nz, Lc.nr, Lc.ld, active);
            }
            cur[l] = smooth(h, Lf, cur[l], other_of(l, cur[l]), rhs_of(l), omega, post, active, false, l);
        }
    }
    return cur[0];
}

static bool same_fld(const Fld& a, const Fld& b) { return a.base == b.base && a.slot == b.slot; }
static void copy_back(Solver* h, Fld dst, Fld res, const int* active) {
    if (same_fld(dst, res)) return;
    SCPN_LAUNCH(k_copy, grid2d(h->nr, h->nz, h->B), BLK2D, 0, h->stream, dst, res, h->nz, h->nr, h->ld, active);
}

// ------------------------------------------------------------------------------------------ transfers
static void upload(Solver* h, double* dst_base, int member, const double* host) {
    // host: dense (nz, nr) [x B when member < 0]; device: pitched rows, member stride
    if (member >= 0) {
        HIP_TRY(hipMemcpy2DAsync(dst_base + (long)member * h->stride, sizeof(double) * h->ld, host, sizeof(double) * h->nr,
                                 sizeof(double) * h->nr, h->nz, hipMemcpyHostToDevice, h->stream));
    } else {
        for (int b = 0; b < h->B; ++b)
            HIP_TRY(hipMemcpy2DAsync(dst_base + (long)b * h->stride, sizeof(double) * h->ld, host + (long)b * h->nz * h->nr,
                                     sizeof(double) * h->nr, sizeof(double) * h->nr, h->nz, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
}
static inline void cpu_relax() {  // a spin-wait hint where the host ISA has one
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}
// Wait for a stream by polling it for a few milliseconds before blocking: a blocking wait sleeps on an interrupt, and on some hosts
// that sleep only ends at the next scheduler tick (a 1-ms call of a stateless operator then takes 20-30 ms: observed on one box of the pool, round 6)
static hipError_t stream_wait(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(st);
        if (q != hipErrorNotReady) return q;
        cpu_relax();
        if ((spins & 0x3ffu) == 0x3ffu && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) return hipStreamSynchronize(st);
    }
}
static void download(Solver* h, const double* src_base, int member, double* host) {
    if (member >= 0) {
        HIP_TRY(hipMemcpy2DAsync(host, sizeof(double) * h->nr, src_base + (long)member * h->stride, sizeof(double) * h->ld,
                                 sizeof(double) * h->nr, h->nz, hipMemcpyDeviceToHost, h->stream));
    } else {
        for (int b = 0; b < h->B; ++b)
            HIP_TRY(hipMemcpy2DAsync(host + (long)b * h->nz * h->nr, sizeof(double) * h->nr, src_base + (long)b * h->stride,
                                     sizeof(double) * h->ld, sizeof(double) * h->nr, h->nz, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(stream_wait(h->stream));
}
// psi lives in ping-pong slots: member b is at slot cur[b]
static std::vector<int> host_cur(Solver* h) {
    std::vector<int> cur(h->B, 0);
    HIP_TRY(hipMemcpyAsync(cur.data(), h->d_cur.p, sizeof(int) * h->B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return cur;
}
static double* psi_member_ptr(Solver* h, const std::vector<int>& cur, int b) {
    return h->psi_slots.p + ((long)cur[b] * h->B + b) * h->stride;
}

static void ensure_picard_bufs(Solver* h) {
    if (h->picard_bufs) return;
    size_t n = (size_t)h->stride * h->B;
    h->src2.alloc(n);
    h->bc.alloc(n);
    h->jphi.alloc(n);
    h->ext.alloc(n);
    h->old.alloc(n);
    HIP_TRY(hipMemsetAsync(h->bc.p, 0, sizeof(double) * n, h->stream));
    HIP_TRY(hipMemsetAsync(h->jphi.p, 0, sizeof(double) * n, h->stream));
    HIP_TRY(hipMemsetAsync(h->ext.p, 0, sizeof(double) * n, h->stream));
    h->d_topo.alloc((size_t)h->B * h->nblk());
    h->d_part3.alloc((size_t)h->B * h->nblk());
    h->d_pbad.alloc((size_t)h->B * h->nblk());
    h->d_jpart.alloc((size_t)h->B * h->nblk());
    h->picard_bufs = true;
}

static void set_tables_mu0(Solver* h, double mu0) {
    if (h->mu0_tables == mu0) return;
    std::vector<double> inv(h->nr), neg(h->nr);
    for (int ir = 0; ir < h->nr; ++ir) {
        inv[ir] = 1.0 / (mu0 * h->r_axis[ir]);  // core/fusion_kernel.py:431
        neg[ir] = (-mu0) * h->r_axis[ir];       // core/fusion_kernel_newton_solver.py:513
    }
    h->d_inv_mu0R.alloc(h->nr + 8);
    h->d_neg_mu0R.alloc(h->nr + 8);
    HIP_TRY(hipMemcpy(h->d_inv_mu0R.p, inv.data(), sizeof(double) * h->nr, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_neg_mu0R.p, neg.data(), sizeof(double) * h->nr, hipMemcpyHostToDevice));
    h->mu0_tables = mu0;
}

// ------------------------------------------------------------------------------------------ creation
static Solver* create_impl(int nr, int nz, int batch, const double* r_axis, const double* z_axis, double dr, double dz) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available");
    HIP_TRY(hipSetDevice(g_device));
    auto* h = new Solver();
    try {
        h->nr = nr; h->nz = nz; h->B = batch; h->device = g_device;
        h->ld = pitch_of(nr);
        h->stride = stride_of(nz, h->ld);
        h->dr = dr; h->dz = dz;
        h->r_axis.assign(r_axis, r_axis + nr);
        if (z_axis) h->z_axis.assign(z_axis, z_axis + nz);
        else { h->z_axis.assign(nz, 0.0); for (int i = 0; i < nz; ++i) h->z_axis[i] = i * dz; }
        HIP_TRY(hipStreamCreate(&h->stream));
        size_t n = (size_t)h->stride * batch;
        h->psi_slots.alloc(3 * n);
        h->W.alloc(n);
        h->W2.alloc(n);
        h->src.alloc(n);
        {
            const char* e = std::getenv("SCPN_HIP_FUSED");
            h->use_fused = !(e && e[0] == '0');
        }
        HIP_TRY(hipMemsetAsync(h->psi_slots.p, 0, sizeof(double) * 3 * n, h->stream));
        HIP_TRY(hipMemsetAsync(h->W.p, 0, sizeof(double) * n, h->stream));
        HIP_TRY(hipMemsetAsync(h->W2.p, 0, sizeof(double) * n, h->stream));
        HIP_TRY(hipMemsetAsync(h->src.p, 0, sizeof(double) * n, h->stream));
        h->d_r.alloc(nr);
        h->d_z.alloc(nz);
        HIP_TRY(hipMemcpy(h->d_r.p, h->r_axis.data(), sizeof(double) * nr, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->d_z.p, h->z_axis.data(), sizeof(double) * nz, hipMemcpyHostToDevice));
        h->d_st.alloc(batch);
        h->d_mg.alloc(batch);
        h->d_cur.alloc(batch); h->d_nxt.alloc(batch); h->d_best.alloc(batch); h->d_prev.alloc(batch);
        h->d_active.alloc(batch);
        h->d_nactive.alloc(1);
        HIP_TRY(hipMemset(h->d_cur.p, 0, sizeof(int) * batch));
        HIP_TRY(hipMemset(h->d_nxt.p, 0, sizeof(int) * batch));
        HIP_TRY(hipMemset(h->d_best.p, 0, sizeof(int) * batch));
        HIP_TRY(hipMemset(h->d_prev.p, 0, sizeof(int) * batch));
        h->d_part1.alloc((size_t)batch * h->nblk());
        h->d_part2.alloc((size_t)batch * h->nblk());
        h->d_scal.alloc(8);
        HIP_TRY(hipHostMalloc((void**)&h->h_nactive, sizeof(int) * 4, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void**)&h->h_word, 64, hipHostMallocMapped));
        std::memset(h->h_word, 0, 64);  // word 0: poll_word; 1 / 2: progress / done words of the run-ahead Picard loop
        HIP_TRY(hipHostGetDevicePointer((void**)&h->d_word, h->h_word, 0));
        h->i_target.assign(batch, 0.0);
        HIP_TRY(hipStreamSynchronize(h->stream));
    } catch (...) {
        delete h;
        throw;
    }
    return h;
}

#define API_BEGIN try {
#define API_END                                    \
    }                                              \
    catch (const std::exception& e) {              \
        return fail(SCPN_E_DEVICE, e.what());      \
    }                                              \
    return SCPN_OK;

static bool omega_ok(double w) { return std::isfinite(w) && w >= 1.0 && w < 2.0; }

// Value of the device int at `dptr` once everything enqueued on the handle's stream so far has run (the stream is NOT
// synchronised through the runtime: see k_post_word).  SCPN_HIP_POLL=memcpy restores hipMemcpyAsync + hipStreamSynchronize.
static int poll_word(Solver* h, const int* dptr) {
    static const bool use_memcpy = [] { const char* e = std::getenv("SCPN_HIP_POLL"); return e && std::string(e) == "memcpy"; }();
    hipStream_t st = h->stream;
    if (use_memcpy || !h->d_word) {
        HIP_TRY(hipMemcpyAsync(h->h_nactive, dptr, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return h->h_nactive[0];
    }
    const unsigned seq = ++h->word_seq;
    SCPN_LAUNCH(k_post_word, dim3(1), dim3(1), 0, st, dptr, h->d_word, seq);
    // Spin on the host-mapped word: tight for the first ~50 us (the usual wait is a few kernel times), then yielding the core
    // between looks (two stream-group threads per rank and eight ranks per node must not pin sixteen cores for a long wait); the
    // runtime is asked every ~2 ms of WALL time whether the stream has failed -- a device fault would otherwise leave the word
    // unposted for ever.
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    auto next_query = t0 + std::chrono::milliseconds(2);
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long w = __atomic_load_n(h->h_word, __ATOMIC_ACQUIRE);
        if ((unsigned)(w >> 32) == seq) return (int)(unsigned)(w & 0xffffffffull);
        cpu_relax();
        if ((spins & 0xffu) != 0xffu) continue;
        const auto now = clk::now();
        if (now - t0 > std::chrono::microseconds(50)) std::this_thread::yield();
        if (now < next_query) continue;
        next_query = now + std::chrono::milliseconds(2);
        const hipError_t qs = hipStreamQuery(st);
        if (qs == hipSuccess) {
            const unsigned long long w2 = __atomic_load_n(h->h_word, __ATOMIC_ACQUIRE);
            if ((unsigned)(w2 >> 32) == seq) return (int)(unsigned)(w2 & 0xffffffffull);
            throw std::runtime_error("progress word was not posted although the stream is idle");
        }
        if (qs != hipErrorNotReady) throw std::runtime_error(std::string("stream failed while polling: ") + hipGetErrorString(qs));
        if (now - t0 > std::chrono::seconds(120)) HIP_TRY(hipStreamSynchronize(st));
    }
}

// ------------------------------------------------------------------------------------------ Picard
static void picard_solve_impl(Solver* h, const scpn_picard_params* p, scpn_picard_result* results, double* hist_diff,
                              double* hist_gs, double* trace, const std::vector<int>* member_mask) {
    ensure_picard_bufs(h);
    set_tables_mu0(h, p->mu0);
    const int B = h->B, nz = h->nz, nr = h->nr, ld = h->ld;
    const int nblk1 = h->nblk1(), nblk2 = h->nblk2(), nblk_r = h->nblk_int();
    auto& lv = get_hierarchy(h, 5);
    const Lvl& L0 = lv[0]->dev;
    hipStream_t st = h->stream;
    const int max_iter = std::max(p->max_iter, 0);

    // ---- member state
    std::vector<int> cur = host_cur(h);
    std::vector<Member> ms(B);
    int n_act = 0;
    std::vector<int> zero_current(B, 0);
    for (int b = 0; b < B; ++b) {
        Member& m = ms[b];
        std::memset(&m, 0, sizeof(m));
        m.i_target = h->i_target[b];
        m.diff_best = 1e9;
        m.gs_best = INFINITY;
        m.gs = INFINITY;
        m.cur = cur[b];
        m.best = (cur[b] + 1) % 3;  // holds the pre-seed copy (Psi_best = Psi.copy(), newton_solver.py:484)
        m.nxt = (cur[b] + 2) % 3;
        m.prev = cur[b];
        m.x_idx = -1;
        bool masked_out = member_mask && !(*member_mask)[b];
        // zero-current short-circuit (newton_solver.py:448-465)
        // (the Newton dispatch, method 4 = its Picard warm-up, has no such short-circuit: newton_solver.py:435-439)
        if (!masked_out && p->method != 4 && std::fabs(m.i_target) < 1e-12 && !p->preserve_initial_state) zero_current[b] = 1;
        m.active = (!masked_out && !zero_current[b] && max_iter > 0) ? 1 : 0;
        if (zero_current[b]) m.converged = 1;
        n_act += m.active;
    }
    HIP_TRY(hipMemcpyAsync(h->d_st.p, ms.data(), sizeof(Member) * B, hipMemcpyHostToDevice, st));
    SCPN_LAUNCH(k_publish_slots, dim3(cdiv(B, 64)), dim3(64), 0, st, h->d_st.p, h->d_cur.p, h->d_nxt.p, h->d_best.p,
                       h->d_prev.p, h->d_active.p, B);
    // "all members that take part" mask for the preparation steps (active or zero-current)
    std::vector<int> part(B);
    for (int b = 0; b < B; ++b) part[b] = (member_mask ? (*member_mask)[b] : 1) && !zero_current[b];
    // (per-call scratch lives in the handle: hipFree synchronises the whole device, i.e. also the other stream group)
    DBuf<int>& d_part = h->d_scr_part;
    d_part.alloc(B);
    HIP_TRY(hipMemcpyAsync(d_part.p, part.data(), sizeof(int) * B, hipMemcpyHostToDevice, st));

    Fld curF = h->slot_fld(h->d_cur.p), nxtF = h->slot_fld(h->d_nxt.p), bestF = h->slot_fld(h->d_best.p);
    Fld prevF = h->slot_fld(h->d_prev.p);
    Fld WF = h->plain(h->W.p), W2F = h->plain(h->W2.p), srcF = h->plain(h->src.p), src2F = h->plain(h->src2.p), bcF = h->plain(h->bc.p);
    Fld jF = h->plain(h->jphi.p);
    dim3 g2 = grid2d(nr, nz, B);
    dim3 gring(cdiv(2L * (nz + nr), 256), B);
    dim3 gjs(cdiv(nr, 128), cdiv(nz, JS_ROWS), B);  // column-pair kernels (k_jsum, k_source)
    const int nblkj = (int)(gjs.x * gjs.y);

    // ---- _prepare_initial_flux (iterative_solver.py:412-451)
    bool need_vac = !p->has_boundary_flux && !p->preserve_initial_state;
    bool any_zero = false;
    for (int b = 0; b < B; ++b) any_zero |= zero_current[b] != 0;
    if (need_vac) {
        if (h->n_vac > 0) {
            // calculate_vacuum_field() of unchanged coils is the same field every call (a transport loop re-solves with the same
            // coil set each step): evaluated once per coil set and permeability into a field kept by the handle -- the elliptic
            // integrals of every grid point and coil are 1.8 ms for 32 members at 513^2 -- and copied from there
            if (!(h->vac_valid && h->vac_mu0 == p->mu0)) {
                h->vac_field.alloc((size_t)h->stride * B);
                SCPN_LAUNCH(k_coil_flux, g2, BLK2D, 0, st, h->plain(h->vac_field.p), nz, nr, ld, h->d_r.p, h->d_z.p, h->n_vac, h->d_vac_r.p,
                                   h->d_vac_z.p, h->d_vac_i.p, (const double*)nullptr, p->mu0, 0);
                h->vac_valid = true;
                h->vac_mu0 = p->mu0;
            }
            SCPN_LAUNCH(k_copy, g2, BLK2D, 0, st, bcF, h->plain(h->vac_field.p), nz, nr, ld, (const int*)nullptr);
        } else
            SCPN_LAUNCH(k_fill, g2, BLK2D, 0, st, bcF, 0.0, nz, nr, ld, (const int*)d_part.p);
    } else if (!p->has_boundary_flux) {  // preserve_initial_state: the current psi is the boundary map
        SCPN_LAUNCH(k_copy, g2, BLK2D, 0, st, bcF, curF, nz, nr, ld, (const int*)d_part.p);
    }
    if (p->preserve_initial_state) {
        SCPN_LAUNCH(k_apply_bc, gring, dim3(256), 0, st, curF, bcF, nz, nr, ld, (const int*)d_part.p);
    } else {
        SCPN_LAUNCH(k_copy, g2, BLK2D, 0, st, curF, bcF, nz, nr, ld, (const int*)d_part.p);
    }
    if (any_zero) {  // zero-current members: psi = calculate_vacuum_field() (newton_solver.py:450)
        if (h->n_vac > 0)
            SCPN_LAUNCH(k_coil_flux, g2, BLK2D, 0, st, WF, nz, nr, ld, h->d_r.p, h->d_z.p, h->n_vac, h->d_vac_r.p, h->d_vac_z.p,
                               h->d_vac_i.p, (const double*)nullptr, p->mu0, 0);
        else
            SCPN_LAUNCH(k_fill, g2, BLK2D, 0, st, WF, 0.0, nz, nr, ld, (const int*)nullptr);
        DBuf<int>& d_zc = h->d_scr_zc;
        d_zc.alloc(B);
        HIP_TRY(hipMemcpyAsync(d_zc.p, zero_current.data(), sizeof(int) * B, hipMemcpyHostToDevice, st));
        SCPN_LAUNCH(k_copy, g2, BLK2D, 0, st, curF, WF, nz, nr, ld, (const int*)d_zc.p);
        HIP_TRY(hipStreamSynchronize(st));
    }
    // Psi_best = Psi.copy() before seeding
    SCPN_LAUNCH(k_copy, g2, BLK2D, 0, st, bestF, curF, nz, nr, ld, (const int*)h->d_active.p);

    // ---- _seed_plasma (iterative_solver.py:384-410)
    int n_part = 0;
    for (int b = 0; b < B; ++b) n_part += part[b];
    if (!p->skip_seed && n_part > 0) {
        // the Gaussian seed and its integral depend only on the grid and r_center: built once per handle (263k exp + a 2 MB
        // pageable upload per Picard call otherwise stall the stream for ~1 ms at 513^2)
        const bool seed_stale = h->seed_user ? !h->seed_uploaded : !(h->seed_uploaded && h->seed_rc == p->r_center);
        if (seed_stale) {
            if (!h->seed_user) {
                h->seed_host.assign((size_t)nz * nr, 0.0);
                for (int iz = 0; iz < nz; ++iz)
                    for (int ir = 0; ir < nr; ++ir) {
                        double a = h->r_axis[ir] - p->r_center;
                        double dist_sq = a * a + h->z_axis[iz] * h->z_axis[iz];
                        h->seed_host[(size_t)iz * nr + ir] = std::exp(-dist_sq / 2.0);
                    }
            }
            h->seed_integral = (np_sum(h->seed_host.data(), (long)nz * nr) * h->dr) * h->dz;
            h->d_seed.alloc((size_t)nz * nr);
            HIP_TRY(hipMemcpyAsync(h->d_seed.p, h->seed_host.data(), sizeof(double) * nz * nr, hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            h->seed_uploaded = true;
            h->seed_rc = p->r_center;
        }
        const double i_seed = h->seed_integral;
        SCPN_LAUNCH(k_seed_j, g2, BLK2D, 0, st, jF, (const double*)h->d_seed.p, (const Member*)h->d_st.p, i_seed, nz, nr, ld,
                           (const int*)d_part.p);
        // members with |I_target| < 1e-12 keep psi untouched (their J is zero)
        std::vector<int>& seed_act = h->h_seed_act;  // (lives in the handle: its upload is not waited for here)
        seed_act.assign(B, 0);
        for (int b = 0; b < B; ++b) seed_act[b] = part[b] && std::fabs(ms[b].i_target) >= 1e-12;
        DBuf<int>& d_sa = h->d_scr_sa;
        d_sa.alloc(B);
        HIP_TRY(hipMemcpyAsync(d_sa.p, seed_act.data(), sizeof(int) * B, hipMemcpyHostToDevice, st));
        SCPN_LAUNCH(k_source_from_j, g2, BLK2D, 0, st, srcF, jF, nz, nr, ld, (const double*)h->d_neg_mu0R.p, (const int*)d_sa.p);
        if (h->use_fused && nz >= 16 && nr >= 16) {  // 50 steps = 10 launches of 5 temporally blocked steps
            ProfScope ps(h, "seed_jacobi", 10);
            const dim3 gj(cdiv(nr, 116), cdiv(nz, 36), B);  // (window 36 x 116 of a 48 x 128 register tile)
            for (int s = 0; s < 5; ++s) {
                auto kj = h->arith == 1 ? &k_jacobi_reg<5, false, true> : &k_jacobi_reg<5, false, false>;
                SCPN_LAUNCH(kj, gj, dim3(512), 0, st, WF, curF, srcF, L0, (const int*)d_sa.p, 1.0, 0.0);
                SCPN_LAUNCH(kj, gj, dim3(512), 0, st, curF, WF, srcF, L0, (const int*)d_sa.p, 1.0, 0.0);
            }
        } else {
            ProfScope ps(h, "seed_jacobi", 50);
            for (int s = 0; s < 25; ++s) {
                SCPN_LAUNCH(k_jacobi, g2, BLK2D, 0, st, WF, curF, srcF, L0, (const int*)d_sa.p);
                SCPN_LAUNCH(k_jacobi, g2, BLK2D, 0, st, curF, WF, srcF, L0, (const int*)d_sa.p);
            }
        }
        // (the relaxation stays in the stream: the host needs nothing from it; seed_act / d_sa are per-handle buffers)
    }
    if (p->external_profile_mode) {
        SCPN_LAUNCH(k_source_from_j, g2, BLK2D, 0, st, srcF, jF, nz, nr, ld, (const double*)h->d_neg_mu0R.p, (const int*)h->d_active.p);
    }

    // ---- Picard loop (newton_solver.py:499-582)
    Prof pf;
    pf.hmode = p->hmode;
    for (int i = 0; i < 4; ++i) { pf.ped_p[i] = p->ped_p[i]; pf.ped_ff[i] = p->ped_ff[i]; }
    int n_mask_rows = 0;  // rows with Z < 0.5*Z_min (fusion_kernel.py:279); Z ascending
    for (int iz = 0; iz < nz; ++iz) if ((p->z_min_cfg * 0.5) > h->z_axis[iz]) n_mask_rows = iz + 1;
    h->h_nactive[0] = n_act;
    HIP_TRY(hipMemcpyAsync(h->d_nactive.p, h->h_nactive, sizeof(int), hipMemcpyHostToDevice, st));
    DBuf<double> d_hist_diff, d_hist_gs, d_trace;
    if (hist_diff) { d_hist_diff.alloc((size_t)B * max_iter); HIP_TRY(hipMemsetAsync(d_hist_diff.p, 0, sizeof(double) * B * max_iter, st)); }
    if (hist_gs) { d_hist_gs.alloc((size_t)B * max_iter); HIP_TRY(hipMemsetAsync(d_hist_gs.p, 0, sizeof(double) * B * max_iter, st)); }
    if (trace) { d_trace.alloc((size_t)B * max_iter * 6); HIP_TRY(hipMemsetAsync(d_trace.p, 0, sizeof(double) * B * max_iter * 6, st)); }
    const int check_every = std::max(1, p->check_every);
    // SCPN_HIP_RUNAHEAD=n: iterations the host may enqueue beyond the one whose final kernel the device has started (default 2;
    // 0 = the blocking poll every check_every iterations of rounds 1-4)
    static const int run_ahead_env = [] { const char* e = std::getenv("SCPN_HIP_RUNAHEAD"); return e ? std::max(0, std::atoi(e)) : 2; }();
    const int run_ahead = h->d_word ? run_ahead_env : 0;
    const unsigned solve_id = ++h->word_seq;
    if (run_ahead > 0) {
        h->d_iter_count.alloc(1);
        HIP_TRY(hipMemsetAsync(h->d_iter_count.p, 0, sizeof(int), h->stream));
    }
    const double om_alpha = 1.0 - p->alpha;
    const int* act = h->d_active.p;
    int topo_count = nblk2;
    int and_start = 0, and_len = 0, iter_k = 0;  // Anderson history ring (uniform across members: lock-step iterations)
    if (p->method == 3) {
        const size_t hs = (size_t)h->stride * B, cap = (size_t)p->anderson_m + 3;
        h->and_psi.alloc(hs * cap);
        h->and_res.alloc(hs * cap);
        h->and_part.alloc((size_t)B * nblk1 * 20);
        h->and_alpha.alloc((size_t)B * AND_MAX);
        h->and_fb.alloc(B);
    }
    // lazy source: k_jsum keeps J_raw in the source field and the level-0 consumers (tile pre / post smoother, fused tail)
    // apply (-mu0 R) * (J_raw * scale) on load -- the k_source pass over the fine grid disappears (bit-identical; -1.1 % per
    // iteration on MI355X).  SCPN_HIP_LAZYSRC=0 restores the separate source pass
    static const bool lazy_env = [] { const char* e = std::getenv("SCPN_HIP_LAZYSRC"); return !(e && e[0] == '0'); }();
    const bool lazy = lazy_env && h->use_fused && p->method == 0 && !p->external_profile_mode &&
                      vcycle_lazy_ok(h, 3, 3, 5);
    struct LazyGuard {
        Solver* h;
        ~LazyGuard() { h->lazy_src = nullptr; h->jsum_part = nullptr; h->lazy_jfly = false; }
    } lazy_guard{h};
    // Source on the fly (contracted arithmetic, L-mode; LazySrc::jfly): J_raw is never stored -- the level-0 consumers rebuild it from
    // the current iterate, which the pre-smoother and the tail load anyway.  SCPN_HIP_JFLY=0 keeps the stored J_raw field.
    static const bool jfly_env = [] { const char* e = std::getenv("SCPN_HIP_JFLY"); return !(e && e[0] == '0'); }();
    const bool jfly = jfly_env && lazy && h->arith == 1 && !p->hmode;
    const Fld src0F = jfly ? curF : srcF;  // what the V-cycle's level-0 kernels are handed as their source field
    if (lazy) { h->lazy_src = src0F.base; h->lazy_jfly = jfly; }
    // the fused tail (k_blend_tile) leaves the topology partials of the next iterate, and k_picard_final finishes them for the
    // members that stay active: from the second iteration on there is no topology launch at all
    const bool topo_in_final = h->use_fused && p->method != 3 && !p->xpoint_saddle;
    // The J_raw sum finished by its first consumer: in lazy-source mode the V-cycle's first level-0 tile launch adds the per-
    // workgroup partial sums itself (every wave, redundantly: <= 256 values) and forms the scale, so k_jsum_final and its kernel
    // boundary leave the chain -- no ticket, no atomics (the "last workgroup finishes" form of rounds 3-4 is archived in
    // profiles/r04_negative/last_workgroup_finishes.md: slower every time it was measured).  Measured (same box): ONE 513^2 equilibrium
    // 122.7 -> 121.1 us per iteration, 2 x 16 members 347.6 -> 349.4: like every fused scalar tail before it, it pays only where
    // the iteration is a chain of short launches -- default for stream groups of up to 14 members (the register-tile rule);
    // SCPN_HIP_SUMFUSE=0 / 1 forces it off / on.
    static const int sumfuse_env = [] { const char* e = std::getenv("SCPN_HIP_SUMFUSE"); return e ? std::atoi(e) : -1; }();
    const bool sum_fuse = (sumfuse_env == 1 || (sumfuse_env < 0 && B <= 14)) && lazy && nblkj <= 256;
    auto iteration = [&](bool first) {
        if (first || !topo_in_final) {
            ProfScope ps(h, "topology", first || !h->use_fused || p->method == 3 || p->xpoint_saddle ? 2 : 1);
            int ntopo = topo_count;  // partial count produced by the previous iteration's fused tail
            if (first || !h->use_fused || p->method == 3 || p->xpoint_saddle) {  // later iterations get their partials from the fused blend kernel
                ntopo = nblk1;
                SCPN_LAUNCH(k_topology, dim3(nblk1, B), dim3(256), 0, st, curF, (const Member*)h->d_st.p, nz, nr, ld, h->dr, h->dz,
                                   n_mask_rows, h->d_topo.p, nblk1);
            }
            SCPN_LAUNCH(k_topology_final, dim3(B), dim3(256), 0, st, curF, h->d_st.p, (const TopoPart*)h->d_topo.p, ntopo, nr, ld,
                               (const double*)h->d_r.p, (const double*)h->d_z.p);
            if (p->xpoint_saddle)
                SCPN_LAUNCH(k_xpoint_saddle, dim3(B), dim3(1024), 0, st, curF, h->d_st.p, nz, nr, ld, h->dr, h->dz, n_mask_rows,
                                   (const double*)h->d_r.p, (const double*)h->d_z.p);
        }
        if (!p->external_profile_mode) {
            ProfScope ps(h, "source", (lazy ? 2 : 3) - (sum_fuse ? 1 : 0));
            SCPN_LAUNCH((jfly ? k_jsum<0, true> : pf.hmode ? k_jsum<1> : k_jsum<0>), gjs, dim3(256), 0, st, curF, (const Member*)h->d_st.p, nz, nr, ld, (const double*)h->d_r.p,
                               (const double*)h->d_inv_mu0R.p, pf, h->d_jpart.p, nblkj, (lazy && !jfly) ? srcF : Fld{nullptr, nullptr, 0, 0});
            if (sum_fuse) { h->jsum_part = h->d_jpart.p; h->jsum_nblk = nblkj; }  // (the V-cycle's first level-0 launch finishes the sum)
            else
                SCPN_LAUNCH(k_jsum_final, dim3(B), dim3(256), 0, st, h->d_st.p, (const double*)h->d_jpart.p, nblkj, h->dr, h->dz);
            if (!lazy)  // (lazy: the V-cycle's level-0 tiles and the fused tail form the source from J_raw on load)
                SCPN_LAUNCH(k_source, gjs, dim3(256), 0, st, srcF, jF, 0, curF, (const Member*)h->d_st.p, nz, nr, ld,
                                   (const double*)h->d_r.p, (const double*)h->d_inv_mu0R.p, (const double*)h->d_neg_mu0R.p, pf);
        }
        // _elliptic_solve (iterative_solver.py:361-380)
        Fld newF = WF;
        // (source generation inside the pre-smoother and blend / norms / topology inside the post-smoother were implemented,
        //  bit-exact and measured slower or unsafe in rounds 1-2: profiles/r02_negative/tile_kernel_r02_variants.hip.h.txt)
        int nb_d = nblk1, nb_r = nblk_r;
        if (p->method == 0) {
            newF = vcycle(h, curF, WF, W2F, src0F, p->sor_omega, 3, 3, 5, act);
        } else if (p->method == 1 || p->method == 3 || p->method == 4) {  // "sor", "anderson" and any other method string (iterative_solver.py:376-378)
            SCPN_LAUNCH(k_sanitize, g2, BLK2D, 0, st, WF, curF, nz, nr, ld);
            SCPN_LAUNCH(k_sanitize, g2, BLK2D, 0, st, src2F, srcF, nz, nr, ld);
            smooth(h, L0, WF, Fld{nullptr, nullptr, 0, 0}, src2F, p->sor_omega, 1, act, true);
        } else {
            SCPN_LAUNCH(k_jacobi, g2, BLK2D, 0, st, WF, curF, srcF, L0, act);
        }
        PicardCtl ctl;
        ctl.tol = p->tol; ctl.gs_tol = p->gs_tol; ctl.require_gs = p->require_gs_residual; ctl.max_iter = max_iter;
        ctl.iter = 0; ctl.pad = (p->method == 4) ? 1 : 0; /* pad = keep the state on divergence (Newton warm-up) */ ctl.n_total = (long)nz * nr; ctl.n_interior = (long)(nz - 2) * (nr - 2);
        PicardTail pt;
        pt.st = h->d_st.p; pt.psum = h->d_part1.p; pt.pbad = h->d_pbad.p; pt.rmax = h->d_part2.p; pt.rsum = h->d_part3.p;
        pt.ctl = ctl; pt.hist_diff = d_hist_diff.p; pt.hist_gs = d_hist_gs.p; pt.trace = d_trace.p; pt.n_active = h->d_nactive.p;
        pt.f_cur = h->d_cur.p; pt.f_nxt = h->d_nxt.p; pt.f_best = h->d_best.p; pt.f_prev = h->d_prev.p; pt.f_active = h->d_active.p;
        pt.nr = nr; pt.r_axis = h->d_r.p; pt.z_axis = h->d_z.p;
        if (run_ahead > 0) { pt.host_prog = h->d_word + 1; pt.host_done = h->d_word + 2; pt.iter_count = h->d_iter_count.p; pt.solve_id = solve_id; }
        if (h->use_fused && p->method != 3) {
            ProfScope ps(h, "blend_fused", 1);
            nb_d = nb_r = nblk2;
            auto recip_lo = [](double d) { return std::fma(-d, 1.0 / d, 1.0) / d; };
            GradRecip gq;
            gq.dz = L0.dz; gq.inv_dz = 1.0 / L0.dz; gq.lo_dz = recip_lo(L0.dz);
            gq.two_dz = 2.0 * L0.dz; gq.inv_two_dz = 1.0 / gq.two_dz; gq.lo_two_dz = recip_lo(gq.two_dz);
            gq.dr = L0.dr; gq.inv_dr = 1.0 / L0.dr; gq.lo_dr = recip_lo(L0.dr);
            static const int xcd_blend = [] { const char* e = std::getenv("SCPN_HIP_XCD_BLEND"); return e ? std::atoi(e) : 1; }();
            SCPN_LAUNCH((h->arith == 1 ? k_blend_tile<true> : k_blend_tile<false>), dim3(cdiv(nr, BL_WR), cdiv(nz, BL_WZ), B), dim3(64 * BL_NW), 0, st, nxtF, curF, newF, bcF, srcF,
                               (const Member*)h->d_st.p, L0, p->alpha, om_alpha, n_mask_rows, h->d_part1.p, h->d_pbad.p, h->d_part2.p,
                               h->d_part3.p, h->d_topo.p, nblk2, lazy ? (const double*)h->d_neg_mu0R.p : (const double*)nullptr, gq, xcd_blend,
                               jfly ? (const double*)h->d_r.p : (const double*)nullptr, jfly ? (const double*)h->d_inv_mu0R.p : (const double*)nullptr);
        } else {
            ProfScope ps(h, "blend_norms", 2);
            SCPN_LAUNCH(k_blend, dim3(nblk1, B), dim3(256), 0, st, nxtF, curF, newF, bcF, (const Member*)h->d_st.p, nz, nr, ld, p->alpha,
                               om_alpha, h->d_part1.p, h->d_pbad.p, nblk1);
            if (p->method == 3) {  // Anderson acceleration (newton_solver.py:540-550)
                const int cap = p->anderson_m + 3;
                const long hs = h->stride * (long)B;
                const int slot = (and_start + and_len) % cap;
                SCPN_LAUNCH(k_and_push, g2, BLK2D, 0, st, h->and_psi.p + slot * hs, h->and_res.p + slot * hs, h->stride, nxtF, newF, bcF,
                                   nz, nr, ld, act);
                and_len += 1;
                if (and_len >= 3 && iter_k % 3 == 0) {
                    const int mk = std::min(p->anderson_m, and_len);
                    if (mk >= 2) {
                        AndFields F{}, P{};
                        F.mk = P.mk = mk; F.stride = P.stride = h->stride;
                        for (int j = 0; j < mk; ++j) {
                            const int sl = (and_start + and_len - mk + j) % cap;
                            F.f[j] = h->and_res.p + sl * hs;
                            P.f[j] = h->and_psi.p + sl * hs;
                        }
                        SCPN_LAUNCH(k_and_gram, dim3(nblk1, B), dim3(256), 0, st, F, nz, nr, ld, h->and_part.p, nblk1, act);
                        SCPN_LAUNCH(k_and_final, dim3(cdiv(B, 64)), dim3(64), 0, st, (const double*)h->and_part.p, nblk1, mk, h->and_alpha.p,
                                           h->and_fb.p, B, act);
                        SCPN_LAUNCH(k_and_mix, g2, BLK2D, 0, st, nxtF, P, (const double*)h->and_alpha.p, (const int*)h->and_fb.p, bcF, nz, nr,
                                           ld, act);
                    } else {  // mk < 2: the latest iterate with the boundary re-applied
                        SCPN_LAUNCH(k_apply_bc, gring, dim3(256), 0, st, nxtF, bcF, nz, nr, ld, act);
                    }
                }
                if (and_len > p->anderson_m + 2) { and_start = (and_start + 1) % cap; and_len -= 1; }
            }
            SCPN_LAUNCH(k_resid_norms, dim3(nblk_r, B), dim3(256), 0, st, nxtF, srcF, L0, h->d_part2.p, h->d_part3.p, nblk_r, act);
        }
        topo_count = nb_d;
        {
            ProfScope ps(h, "picard_final", 1);
            pt.nblk = nb_d; pt.nblk_r = nb_r; pt.ntopo = nb_d;
            pt.topo = topo_in_final ? (const TopoPart*)h->d_topo.p : (const TopoPart*)nullptr;
            SCPN_LAUNCH(k_picard_final, dim3(B), dim3(256), 0, st, pt);
        }
    };
    int k = 0;
    if (run_ahead > 0) {
        // Run-ahead loop: the host never blocks on the device's queue running empty.  It enqueues iteration k + 1 as soon as the
        // device has STARTED the final kernel of iteration k + 1 - run_ahead (progress word), and stops the moment the device says
        // that the last member has left the iteration (done word).  At most `run_ahead` iterations are enqueued beyond the one
        // that froze the last member (their kernels see frozen members only), where the blocking poll every `check_every`
        // iterations ran on average check_every / 2 + the drain of the queue at every poll.
        using clk = std::chrono::steady_clock;
        bool done = n_act <= 0;
        while (!done && k < max_iter) {
            iter_k = k;
            iteration(k == 0);
            ++k;
            const auto t0 = clk::now();
            auto next_query = t0 + std::chrono::milliseconds(2);
            for (unsigned spins = 0;; ++spins) {
                const unsigned long long wd = __atomic_load_n(h->h_word + 2, __ATOMIC_ACQUIRE);
                if ((unsigned)(wd >> 32) == solve_id) { done = true; break; }
                const unsigned long long wp = __atomic_load_n(h->h_word + 1, __ATOMIC_ACQUIRE);
                const int started = (unsigned)(wp >> 32) == solve_id ? (int)(unsigned)(wp & 0xffffffffull) : 0;
                if (k - started < run_ahead) break;
                cpu_relax();
                if ((spins & 0xffu) != 0xffu) continue;
                const auto now = clk::now();
                if (now - t0 > std::chrono::microseconds(200)) std::this_thread::yield();
                if (now < next_query) continue;
                next_query = now + std::chrono::milliseconds(2);
                const hipError_t qs = hipStreamQuery(st);
                if (qs == hipSuccess) {  // everything enqueued has run: the words are final
                    const unsigned long long w2 = __atomic_load_n(h->h_word + 2, __ATOMIC_ACQUIRE);
                    done = (unsigned)(w2 >> 32) == solve_id;
                    break;
                }
                if (qs != hipErrorNotReady) throw std::runtime_error(std::string("stream failed inside the Picard loop: ") + hipGetErrorString(qs));
                if (now - t0 > std::chrono::seconds(120)) HIP_TRY(hipStreamSynchronize(st));
            }
        }
    } else {
        if (n_act > 0 && max_iter > 0) {
            iter_k = 0;
            iteration(true);
            ++k;
            if (k % check_every == 0 || k >= max_iter) {
                n_act = poll_word(h, h->d_nactive.p);
            }
        }
        while (n_act > 0 && k < max_iter) {
            iter_k = k;
            iteration(false);
            ++k;
            if (k % check_every == 0 || k >= max_iter) {
                n_act = poll_word(h, h->d_nactive.p);
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    // ---- J_phi of the last source update (recomputed from the pre-blend psi kept in slot `prev`)
    if (!p->external_profile_mode) {
        std::vector<Member> fin(B);
        HIP_TRY(hipMemcpyAsync(fin.data(), h->d_st.p, sizeof(Member) * B, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<Member>& tmp = h->h_scr_st;  // (lives in the handle: its asynchronous upload below outlives this block)
        tmp = fin;
        bool any = false;
        for (int b = 0; b < B; ++b) { tmp[b].active = (fin[b].iterations > 0) ? 1 : 0; any |= tmp[b].active != 0; }
        if (any) {
            DBuf<Member>& d_tmp = h->d_scr_st;
            d_tmp.alloc(B);
            HIP_TRY(hipMemcpyAsync(d_tmp.p, tmp.data(), sizeof(Member) * B, hipMemcpyHostToDevice, st));
            SCPN_LAUNCH(k_source, gjs, dim3(256), 0, st, src2F, jF, 1, prevF, (const Member*)d_tmp.p, nz, nr, ld, (const double*)h->d_r.p,
                               (const double*)h->d_inv_mu0R.p, (const double*)h->d_neg_mu0R.p, pf);
            if (lazy)  // the source field held J_raw during the iterations: leave the last source in it, as the eager path does
                SCPN_LAUNCH(k_source, gjs, dim3(256), 0, st, srcF, jF, 0, prevF, (const Member*)d_tmp.p, nz, nr, ld, (const double*)h->d_r.p,
                                   (const double*)h->d_inv_mu0R.p, (const double*)h->d_neg_mu0R.p, pf);
            // (the host image `tmp` is the handle's h_scr_st: nothing here depends on when the runtime stages a pageable upload; the
            //  results read-back below waits for these launches)
        }
        for (int b = 0; b < B; ++b)
            if (zero_current[b]) {
                // J_phi = 0 for the vacuum short-circuit
                HIP_TRY(hipMemsetAsync(h->jphi.p + (long)b * h->stride, 0, sizeof(double) * h->stride, st));
            }
    }
    // ---- results
    std::vector<Member> fin(B);
    HIP_TRY(hipMemcpyAsync(fin.data(), h->d_st.p, sizeof(Member) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (results) {
        for (int b = 0; b < B; ++b) {
            const Member& m = fin[b];
            scpn_picard_result& r = results[b];
            std::memset(&r, 0, sizeof(r));
            r.converged = m.converged;
            r.iterations = zero_current[b] ? 0 : std::max(m.iterations, (member_mask && !(*member_mask)[b]) ? 0 : 1);
            r.diverged = m.diverged;
            r.residual = zero_current[b] ? 0.0 : m.diff_best;
            r.gs_residual = zero_current[b] ? 0.0 : m.gs;
            r.gs_residual_best = zero_current[b] ? 0.0 : m.gs_best;
            r.psi_axis = m.psi_axis; r.psi_bnd = m.psi_bnd; r.x_r = m.x_r; r.x_z = m.x_z; r.i_current = m.i_current;
        }
    }
    if (hist_diff) HIP_TRY(hipMemcpy(hist_diff, d_hist_diff.p, sizeof(double) * B * max_iter, hipMemcpyDeviceToHost));
    if (hist_gs) HIP_TRY(hipMemcpy(hist_gs, d_hist_gs.p, sizeof(double) * B * max_iter, hipMemcpyDeviceToHost));
    if (trace) HIP_TRY(hipMemcpy(trace, d_trace.p, sizeof(double) * B * max_iter * 6, hipMemcpyDeviceToHost));
}

// ------------------------------------------------------------------------------------------ Newton-Kantorovich + GMRES
// core/fusion_kernel_newton_solver.py:104-372.  Phase A = picard_solve_impl in warm-up mode (method 4); phase B below.
static void newton_solve_impl(Solver* h, const scpn_picard_params* p, const scpn_newton_opts* o, scpn_newton_result* results,
                              double* hist_out, double* gs_hist_out, const std::vector<int>* member_mask = nullptr) {
    const int B = h->B, nz = h->nz, nr = h->nr, ld = h->ld;
    const long stride = h->stride;
    const int max_iter = std::max(p->max_iter, 0);
    const int warmup = o->warmup_steps >= 0 ? std::min(o->warmup_steps, max_iter) : std::min(15, max_iter / 2);
    const int cap = max_iter + 2;
    hipStream_t st = h->stream;

    // ---- phase A: Picard warm-up
    scpn_picard_params pa = *p;
    pa.method = 4;
    pa.max_iter = warmup;
    pa.skip_seed = 0;
    std::vector<scpn_picard_result> ra(B);
    std::vector<double> hd((size_t)B * std::max(warmup, 1), 0.0), hg((size_t)B * std::max(warmup, 1), 0.0);
    picard_solve_impl(h, &pa, ra.data(), warmup > 0 ? hd.data() : nullptr, warmup > 0 ? hg.data() : nullptr, nullptr, member_mask);

    // ---- workspace
    const long n = stride * B;
    const int restart = std::max(1, std::min({o->gmres_restart > 0 ? o->gmres_restart : GM_R, GM_R, (nz - 2) * (nr - 2)}));
    enum { RK = 0, RHS, RES, WV, XV, DG, IVD, SRT, PRV, ZERO, TMPV, PSIP, NBUF };  // ZERO / TMPV: multigrid preconditioner; PSIP: psi + eps v
    if (!h->nt_ws.p) {
        h->nt_ws.alloc((size_t)n * NBUF);
        HIP_TRY(hipMemsetAsync(h->nt_ws.p, 0, sizeof(double) * n * NBUF, st));
        h->nt_gm.alloc(B);
        h->nt_st.alloc(B);
        HIP_TRY(hipMemsetAsync(h->nt_gm.p, 0, sizeof(GmState) * B, st));
        const int nbmax = std::max(cdiv(stride, 1024), (int)(cdiv(nr, 64) * cdiv(nz, 4)));
        h->nt_part.alloc((size_t)2 * B * nbmax);
        h->nt_pbad.alloc((size_t)B * nbmax);
        h->nt_cnt.alloc(4);
    }
    if (h->nt_V_vecs < restart) {
        h->nt_V.alloc((size_t)n * restart);
        HIP_TRY(hipMemsetAsync(h->nt_V.p, 0, sizeof(double) * n * restart, st));
        h->nt_V_vecs = restart;
    }
    auto buf = [&](int i) { return h->nt_ws.p + (long)i * n; };
    Fld rkF = h->plain(buf(RK)), dxF = h->plain(buf(XV)), dgF = h->plain(buf(DG)), ivF = h->plain(buf(IVD)), srtF = h->plain(buf(SRT)),
        prvF = h->plain(buf(PRV));
    Fld curF = h->slot_fld(h->d_cur.p), srcF = h->plain(h->src.p), src2F = h->plain(h->src2.p), bcF = h->plain(h->bc.p),
        jF = h->plain(h->jphi.p);
    auto& lv = get_hierarchy(h, 5);
    const Lvl& L0 = lv[0]->dev;
    const dim3 g2 = grid2d(nr, nz, B);
    const int nb2 = (int)(g2.x * g2.y);
    const int nbf = cdiv(stride, 1024);
    const dim3 gflat(nbf, B);
    const int nblk1 = h->nblk1();
    const dim3 gjs(cdiv(nr, 128), cdiv(nz, JS_ROWS), B);
    const int nblkj = (int)(gjs.x * gjs.y);
    Prof pf;
    pf.hmode = p->hmode;
    for (int i = 0; i < 4; ++i) { pf.ped_p[i] = p->ped_p[i]; pf.ped_ff[i] = p->ped_ff[i]; }
    int n_mask_rows = 0;
    for (int iz = 0; iz < nz; ++iz) if ((p->z_min_cfg * 0.5) > h->z_axis[iz]) n_mask_rows = iz + 1;

    // ---- per-member Newton state from the warm-up results
    std::vector<NtMember> nm(B);
    DBuf<double> d_hist, d_gs;
    d_hist.alloc((size_t)B * cap);
    d_gs.alloc((size_t)B * cap);
    std::vector<double> hh((size_t)B * cap, 0.0), hgs((size_t)B * cap, 0.0);
    int n_on = 0;
    for (int b = 0; b < B; ++b) {
        NtMember& m = nm[b];
        std::memset(&m, 0, sizeof(m));
        const int its = warmup > 0 ? ra[b].iterations : 0;
        const int pushed = std::max(0, its - (ra[b].diverged ? 1 : 0));
        for (int i = 0; i < pushed; ++i) { hh[(size_t)b * cap + i] = hd[(size_t)b * warmup + i]; hgs[(size_t)b * cap + i] = hg[(size_t)b * warmup + i]; }
        m.n_hist = m.n_gs_hist = pushed;
        m.last_hist = pushed > 0 ? hh[(size_t)b * cap + pushed - 1] : INFINITY;
        m.gs = pushed > 0 ? hgs[(size_t)b * cap + pushed - 1] : INFINITY;
        m.gs_best = pushed > 0 ? ra[b].gs_residual_best : INFINITY;
        m.converged = ra[b].converged;
        m.iterations = warmup > 0 ? ra[b].iterations : 1;
        m.on = (!m.converged && warmup < max_iter && !(member_mask && !(*member_mask)[b])) ? 1 : 0;
        n_on += m.on;
    }
    HIP_TRY(hipMemcpyAsync(h->nt_st.p, nm.data(), sizeof(NtMember) * B, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_hist.p, hh.data(), sizeof(double) * B * cap, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_gs.p, hgs.data(), sizeof(double) * B * cap, hipMemcpyHostToDevice, st));
    int* d_non = h->nt_cnt.p;          // members still iterating (phase B)
    int* d_nsolve = h->nt_cnt.p + 1;   // members whose GMRES is still running
    int* d_ninner = h->nt_cnt.p + 2;   // members inside the current inner loop
    int hcnt[4] = {n_on, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(d_non, hcnt, sizeof(int) * 4, hipMemcpyHostToDevice, st));
    auto flags = [&](int op, int k = 0) {
        SCPN_LAUNCH(k_nt_flags, dim3(cdiv(B, 64)), dim3(64), 0, st, h->d_st.p, h->nt_st.p, h->d_active.p, B, op, o->newton_alpha, k);
    };
    flags(3);
    if (p->external_profile_mode)
        SCPN_LAUNCH(k_source_from_j, g2, BLK2D, 0, st, srcF, jF, nz, nr, ld, (const double*)h->d_neg_mu0R.p, (const int*)h->d_active.p);

    NtCtl ctl;
    ctl.tol = p->tol; ctl.gs_tol = p->gs_tol; ctl.require_gs = p->require_gs_residual; ctl.line_search = o->line_search;
    ctl.hist_cap = cap; ctl.pad = 0; ctl.n_interior = (long)(nz - 2) * (nr - 2);
    const double diag_lap = -2.0 / std::pow(h->dr, 2.0) - 2.0 / std::pow(h->dz, 2.0);

    // axis / X-point / (source) on the psi in `cur` for the members whose Member.active is set
    auto topo_source = [&](Fld dstSrc) {
        {
            ProfScope ps(h, "topology", 2);
            SCPN_LAUNCH(k_topology, dim3(nblk1, B), dim3(256), 0, st, curF, (const Member*)h->d_st.p, nz, nr, ld, h->dr, h->dz,
                               n_mask_rows, h->d_topo.p, nblk1);
            SCPN_LAUNCH(k_topology_final, dim3(B), dim3(256), 0, st, curF, h->d_st.p, (const TopoPart*)h->d_topo.p, nblk1, nr, ld,
                               (const double*)h->d_r.p, (const double*)h->d_z.p);
            if (p->xpoint_saddle)
                SCPN_LAUNCH(k_xpoint_saddle, dim3(B), dim3(1024), 0, st, curF, h->d_st.p, nz, nr, ld, h->dr, h->dz, n_mask_rows,
                                   (const double*)h->d_r.p, (const double*)h->d_z.p);
        }
        if (!p->external_profile_mode) {
            ProfScope ps(h, "source", 3);
            SCPN_LAUNCH((pf.hmode ? k_jsum<1> : k_jsum<0>), gjs, dim3(256), 0, st, curF, (const Member*)h->d_st.p, nz, nr, ld, (const double*)h->d_r.p,
                               (const double*)h->d_inv_mu0R.p, pf, h->d_jpart.p, nblkj, Fld{nullptr, nullptr, 0, 0});
            SCPN_LAUNCH(k_jsum_final, dim3(B), dim3(256), 0, st, h->d_st.p, (const double*)h->d_jpart.p, nblkj, h->dr, h->dz);
            SCPN_LAUNCH(k_source, gjs, dim3(256), 0, st, dstSrc, jF, 1, curF, (const Member*)h->d_st.p, nz, nr, ld,
                               (const double*)h->d_r.p, (const double*)h->d_inv_mu0R.p, (const double*)h->d_neg_mu0R.p, pf);
        }
    };
    double* P0 = h->nt_part.p;                        // reduction partials, ping-pong halves
    double* P1 = h->nt_part.p + (size_t)B * h->nt_part.n / (2 * (size_t)B);
    auto resid = [&](Fld outF, Fld sF, int mode) {
        ProfScope ps(h, "newton_resid", 2);
        SCPN_LAUNCH(k_nt_resid, g2, BLK2D, 0, st, outF, curF, sF, L0, (const NtMember*)h->nt_st.p, mode == 1 ? 1 : 0, P0);
        SCPN_LAUNCH(k_nt_resid_final, dim3(B), dim3(256), 0, st, h->d_st.p, h->nt_st.p, ctl, mode, o->line_search_c, d_hist.p, d_gs.p,
                           (const double*)P0, nb2, h->d_active.p, d_non);
    };
    auto poll = [&](int* dptr) { return poll_word(h, dptr); };
    // restarted GMRES on (L - diag) x = -r_k for the members that are `on` (scipy gmres, see newton_kernels.hip.h).
    // Preconditioner modes: 0 none, 1 diagonal (the reference's), 2 multigrid: z = M^-1 v is ONE V-cycle(3,3) of the GS* operator
    // from a zero start with v as right-hand side on the zero-ring Krylov grid -- M^-1 (L - D) = I - M^-1 D, the Jacobian's
    // profile term is a bounded perturbation, so the Krylov space needs a handful of vectors instead of thousands.
    auto gmres = [&]() {
        const int mg = o->precond == 2 ? 1 : 0;
        const int precond = (o->precond && !mg) ? 1 : 0;   // what the kernels apply themselves (the diagonal scaling)
        const int pflag = o->precond ? 1 : 0;              // what the tolerance logic sees (a preconditioned norm exists)
        const int maxit = o->gmres_maxiter > 0 ? o->gmres_maxiter : 100;
        const int pe = std::max(1, o->poll_every);
        double* V = h->nt_V.p;
        Fld zeroF = h->plain(buf(ZERO)), tmpF = h->plain(buf(TMPV)), WbufF = h->plain(h->W.p), W2bufF = h->plain(h->W2.p);
        // dst <- M^-1 src (multigrid): returns the device pointer that holds the result (one of the two work buffers)
        auto apply_mg = [&](const double* src_field) -> const double* {
            ProfScope ps(h, "gmres_mg_precond", 1);
            Fld srcv = h->plain(const_cast<double*>(src_field));
            Fld z = vcycle(h, zeroF, WbufF, W2bufF, srcv, 1.0, 3, 3, 5, nullptr);
            return z.base;
        };
        // A v (resid_mode 0, into `out`, with the partial sums of |out|^2 in P0) or rhs - A x (resid_mode 1)
        const bool fd = o->reserved == 1 && !p->external_profile_mode;
        auto matvec = [&](double* out, const double* vin, int resid_mode, bool unit_norm = false) {
            if (!fd) {
                SCPN_LAUNCH(k_gm_matvec, g2, BLK2D, 0, st, out, vin, (const double*)buf(DG), (const double*)buf(IVD), (const double*)buf(RHS), stride,
                                   L0, (const GmState*)h->nt_gm.p, precond, resid_mode, P0);
                return;
            }
            // finite-difference Jacobian action: ||v|| -> eps, psi + eps v, topology + source of the perturbed state on a scratch
            // copy of the member table, then L v - (S_pert - S_0) / eps
            Fld pF = h->plain(buf(PSIP));
            DBuf<Member>& scr = h->d_scr_st;
            scr.alloc(B);
            h->d_scr_diff.alloc(B);
            if (!unit_norm)  // (Krylov basis vectors have unit norm: no pass over v for them)
                SCPN_LAUNCH(k_gm_start, gflat, dim3(256), 0, st, (double*)nullptr, vin, (const double*)buf(IVD), stride, (const GmState*)h->nt_gm.p,
                                   0, 1, P1, nbf);
            SCPN_LAUNCH(k_fd_eps, dim3(B), dim3(256), 0, st, h->d_scr_diff.p, unit_norm ? (const double*)nullptr : (const double*)P1, nbf,
                               (const Member*)h->d_st.p, (double)((long)(nz - 2) * (nr - 2)), (const GmState*)h->nt_gm.p, scr.p);
            SCPN_LAUNCH(k_fd_perturb, dim3(std::min(cdiv(stride, 512), 1024), B), dim3(256), 0, st, pF, curF, vin, stride,
                               (const double*)h->d_scr_diff.p, (const GmState*)h->nt_gm.p);
            SCPN_LAUNCH(k_topology, dim3(nblk1, B), dim3(256), 0, st, pF, (const Member*)scr.p, nz, nr, ld, h->dr, h->dz, n_mask_rows,
                               h->d_topo.p, nblk1);
            SCPN_LAUNCH(k_topology_final, dim3(B), dim3(256), 0, st, pF, scr.p, (const TopoPart*)h->d_topo.p, nblk1, nr, ld,
                               (const double*)h->d_r.p, (const double*)h->d_z.p);
            SCPN_LAUNCH((pf.hmode ? k_jsum<1> : k_jsum<0>), gjs, dim3(256), 0, st, pF, (const Member*)scr.p, nz, nr, ld, (const double*)h->d_r.p,
                               (const double*)h->d_inv_mu0R.p, pf, h->d_jpart.p, nblkj, Fld{nullptr, nullptr, 0, 0});
            SCPN_LAUNCH(k_jsum_final, dim3(B), dim3(256), 0, st, scr.p, (const double*)h->d_jpart.p, nblkj, h->dr, h->dz);
            SCPN_LAUNCH(k_source, gjs, dim3(256), 0, st, src2F, jF, 0, pF, (const Member*)scr.p, nz, nr, ld, (const double*)h->d_r.p,
                               (const double*)h->d_inv_mu0R.p, (const double*)h->d_neg_mu0R.p, pf);
            SCPN_LAUNCH(k_gm_matvec_fd, g2, BLK2D, 0, st, out, vin, src2F, srcF, (const double*)h->d_scr_diff.p, (const double*)buf(RHS), stride, L0,
                               (const GmState*)h->nt_gm.p, resid_mode, P0);
        };
        auto tail = [&](int mode, const double* part, int nblk) {
            SCPN_LAUNCH(k_gm_tail, dim3(B), dim3(256), 0, st, h->nt_gm.p, h->nt_st.p, mode, o->gmres_atol, o->gmres_rtol, restart, pflag,
                               maxit, part, nblk, d_nsolve);
        };
        {
            ProfScope ps(h, "gmres_init", pflag ? 4 : 2);
            SCPN_LAUNCH(k_gm_init, gflat, dim3(256), 0, st, buf(RHS), buf(XV), buf(RES), (const double*)buf(RK), stride, h->nt_gm.p,
                               (const NtMember*)h->nt_st.p, P0, nbf);
            tail(0, P0, nbf);
            if (pflag) {  // norm of the preconditioned right-hand side
                const double* pb = mg ? apply_mg(buf(RHS)) : (const double*)buf(RHS);
                // multigrid: x0 = 0, so M^-1 b is also the unscaled first Krylov vector of cycle 0 -- store it, one V-cycle saved
                SCPN_LAUNCH(k_gm_start, gflat, dim3(256), 0, st, mg ? V : (double*)nullptr, pb, (const double*)buf(IVD), stride,
                                   (const GmState*)h->nt_gm.p, precond, mg ? 0 : 1, P0, nbf);
                tail(1, P0, nbf);
            }
        }
        // small grids: one launch per restart cycle (k_gm_cycle_small); SCPN_HIP_GMRES_SMALL=<max points> (default 20000, 0 = off)
        static const long small_max = [] { const char* e = std::getenv("SCPN_HIP_GMRES_SMALL"); return e ? std::atol(e) : 20000L; }();
        // (one workgroup per member: always for <= 65^2; up to 129^2 when there are enough members to occupy several CUs)
        const long npts = (long)nz * nr;
        if (!mg && !fd && npts <= std::min(small_max, 20000L) && (npts <= 4500 || B >= 4)) {
            SCPN_LAUNCH(k_gm_count, dim3(1), dim3(64), 0, st, (const GmState*)h->nt_gm.p, B, d_nsolve, d_ninner);
            int left = poll(d_nsolve);
            for (int cyc = 0; cyc < maxit && left > 0; ++cyc) {
                ProfScope ps(h, "gmres_cycle_small", 1);
                const long nint = (long)(nz - 2) * (nr - 2);
                auto launch = [&](auto kern) {
                    SCPN_LAUNCH(kern, dim3(B), dim3(1024), 0, st, V, buf(XV), buf(RES), (const double*)buf(RHS), (const double*)buf(DG),
                                       (const double*)buf(IVD), stride, B, L0, h->nt_gm.p, h->nt_st.p, precond, maxit, d_nsolve);
                };
                // (the register variants keep (point index, column) of a point in ONE signed word -- index in bits 0..19, column from
                //  bit 20 up: a thin grid whose columns reach 2048 or whose pitched size reaches 2^20 takes the LDS variant instead)
                const bool packs = nr < 2048 && (long)nz * ld < (1L << 20);
                if (packs && nint <= 1024L * 2) launch(k_gm_cycle_small<2>);
                else if (packs && nint <= 1024L * GM_SMALL_PTS) launch(k_gm_cycle_small<GM_SMALL_PTS>);
                else {
                    static std::once_flag once;
                    std::call_once(once, [] {
                        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gm_cycle_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    160 * 1024 - 256));
                    });
                    SCPN_LAUNCH(k_gm_cycle_lds, dim3(B), dim3(1024), sizeof(double) * nint, st, V, buf(XV), buf(RES), (const double*)buf(RHS),
                                       (const double*)buf(DG), (const double*)buf(IVD), stride, B, L0, h->nt_gm.p, h->nt_st.p, precond, maxit,
                                       d_nsolve);
                }
                if ((cyc & 3) == 3 || cyc + 1 == maxit) left = poll(d_nsolve);  // finished members return at once; poll every 4 cycles
            }
            return;
        }
        const int pe_mg = mg ? 1 : pe;  // a preconditioned solve takes a handful of columns: look at the inner flag after every one
        for (int cyc = 0; cyc < maxit; ++cyc) {
            SCPN_LAUNCH(k_gm_count, dim3(1), dim3(64), 0, st, (const GmState*)h->nt_gm.p, B, d_nsolve, d_ninner);
            // (multigrid lane, first cycle: no host round trip -- members with nothing to solve are masked on the device anyway)
            if (!(mg && cyc == 0) && poll(d_nsolve) == 0) break;
            if (!(mg && cyc == 0)) {  // (multigrid, cycle 0: V[0] and its partial sums are already there from the prologue)
                const double* pr = mg ? apply_mg(buf(RES)) : (const double*)buf(RES);
                SCPN_LAUNCH(k_gm_start, gflat, dim3(256), 0, st, V, pr, (const double*)buf(IVD), stride,
                                   (const GmState*)h->nt_gm.p, precond, 0, P0, nbf);
            }
            SCPN_LAUNCH(k_gm_scale, gflat, dim3(256), 0, st, V, (const double*)V, stride, h->nt_gm.p, 0, (const double*)P0, nbf);
            for (int col = 0; col < restart; ++col) {
                int nb_first = nb2;  // number of partial sums of |w|^2 the first MGS launch reads
                {
                    ProfScope ps(h, "gmres_matvec", 1);
                    matvec(mg ? buf(TMPV) : buf(WV), (const double*)(V + (long)col * n), 0, true);
                }
                if (mg) {  // w = M^-1 (A v): V-cycle, then the copy into w together with the partial sums of |w|^2
                    const double* z = apply_mg(buf(TMPV));
                    SCPN_LAUNCH(k_gm_start, gflat, dim3(256), 0, st, buf(WV), z, (const double*)buf(IVD), stride, (const GmState*)h->nt_gm.p, 0, 0,
                                       P0, nbf);
                    nb_first = nbf;
                }
                {
                    ProfScope ps(h, "gmres_mgs", col + 3);
                    for (int k = 0; k <= col + 1; ++k)  // launch k reads the partials of the previous launch, writes the other half
                        SCPN_LAUNCH(k_gm_mgs, gflat, dim3(256), 0, st, buf(WV), (const double*)V, stride, B, h->nt_gm.p, col, k,
                                           (const double*)((k & 1) ? P1 : P0), k == 0 ? nb_first : nbf, (k & 1) ? P0 : P1, nbf);
                    SCPN_LAUNCH(k_gm_col_tail, dim3(B), dim3(256), 0, st, h->nt_gm.p, col, (const double*)(((col + 1) & 1) ? P0 : P1), nbf,
                                       d_ninner);
                }
                if (col + 1 < restart)
                    SCPN_LAUNCH(k_gm_scale, gflat, dim3(256), 0, st, V + (long)(col + 1) * n, (const double*)buf(WV), stride, h->nt_gm.p, 1,
                                       (const double*)P0, nbf);
                if ((col + 1) % pe_mg == 0 && col + 1 < restart && poll(d_ninner) == 0) break;
            }
            {
                ProfScope ps(h, "gmres_update", 3);
                SCPN_LAUNCH(k_gm_update_x, gflat, dim3(256), 0, st, buf(XV), (const double*)V, stride, B, (const GmState*)h->nt_gm.p);
                if (mg && cyc + 1 == maxit) {
                    tail(3, P0, nb2);  // capped inexact solve: the last cycle exits on the Arnoldi estimate, no extra operator application
                } else {
                    matvec(buf(RES), (const double*)buf(XV), 1);
                    tail(2, P0, nb2);
                }
            }
        }
    };
    auto step_check = [&]() {
        SCPN_LAUNCH(k_nt_step_final, dim3(B), dim3(256), 0, st, h->d_st.p, h->nt_st.p, (const int*)h->nt_pbad.p, nb2, h->d_active.p, d_non);
    };

    // ---- phase B: Newton iterations (newton_solver.py:183-306)
    for (int k = warmup; k < max_iter && n_on > 0; ++k) {
        flags(4, k);
        topo_source(srcF);
        resid(rkF, srcF, 0);
        n_on = poll(d_non);
        if (n_on == 0) break;
        // (finite-difference operator + multigrid preconditioner: neither the profile-Jacobian diagonal nor its inverse is read)
        if (!(o->precond == 2 && o->reserved == 1 && !p->external_profile_mode)) {
            ProfScope ps(h, "newton_jacobian", 2);
            if (!p->external_profile_mode)
                SCPN_LAUNCH(k_nt_jac_sum, dim3(nblk1, B), dim3(256), 0, st, curF, (const Member*)h->d_st.p, (const NtMember*)h->nt_st.p, nz, nr,
                                   ld, (const double*)h->d_r.p, P0, nblk1);
            SCPN_LAUNCH(k_nt_jac_diag, g2, BLK2D, 0, st, dgF, ivF, curF, jF, (const Member*)h->d_st.p, (const NtMember*)h->nt_st.p, nz, nr, ld,
                               (const double*)h->d_r.p, (const double*)h->d_neg_mu0R.p, p->external_profile_mode, o->precond == 1 ? 1 : 0, diag_lap,
                               h->dr, h->dz, (const double*)P0, nblk1);
        }
        gmres();
        if (!o->line_search) {
            SCPN_LAUNCH(k_nt_step, g2, BLK2D, 0, st, curF, curF, dxF, bcF, (const NtMember*)h->nt_st.p, nz, nr, ld, 0, o->newton_alpha,
                               h->nt_pbad.p);
            step_check();
        } else {
            SCPN_LAUNCH(k_nt_copy, g2, BLK2D, 0, st, prvF, curF, (const NtMember*)h->nt_st.p, 0, nz, nr, ld);
            SCPN_LAUNCH(k_nt_copy, g2, BLK2D, 0, st, srtF, srcF, (const NtMember*)h->nt_st.p, 0, nz, nr, ld);
            flags(0);
            for (int bt = 0; bt < o->max_backtracks; ++bt) {
                SCPN_LAUNCH(k_nt_step, g2, BLK2D, 0, st, curF, prvF, dxF, bcF, (const NtMember*)h->nt_st.p, nz, nr, ld, 1, 0.0, (int*)nullptr);
                if (!p->external_profile_mode) { flags(1); topo_source(srtF); }
                resid(Fld{nullptr, nullptr, 0, 0}, srtF, 1);
            }
            // every backtrack rejected: minimum step, J_phi refreshed, Source kept (:277-291)
            SCPN_LAUNCH(k_nt_step, g2, BLK2D, 0, st, curF, prvF, dxF, bcF, (const NtMember*)h->nt_st.p, nz, nr, ld, 2, 0.0, (int*)nullptr);
            if (!p->external_profile_mode) { flags(2); topo_source(src2F); }
            SCPN_LAUNCH(k_nt_copy, g2, BLK2D, 0, st, srcF, srtF, (const NtMember*)h->nt_st.p, 1, nz, nr, ld);
            flags(3);
            resid(Fld{nullptr, nullptr, 0, 0}, srcF, 2);
            // NaN / Inf check of the accepted state (alpha 0 leaves psi as it is)
            SCPN_LAUNCH(k_nt_step, g2, BLK2D, 0, st, curF, curF, dxF, bcF, (const NtMember*)h->nt_st.p, nz, nr, ld, 0, 0.0, h->nt_pbad.p);
            step_check();
        }
        // (no poll here: the one after the next iteration's residual sees the members this step switched off)
    }
    HIP_TRY(hipStreamSynchronize(st));

    // ---- results
    HIP_TRY(hipMemcpy(nm.data(), h->nt_st.p, sizeof(NtMember) * B, hipMemcpyDeviceToHost));
    if (hist_out) HIP_TRY(hipMemcpy(hist_out, d_hist.p, sizeof(double) * B * cap, hipMemcpyDeviceToHost));
    if (gs_hist_out) HIP_TRY(hipMemcpy(gs_hist_out, d_gs.p, sizeof(double) * B * cap, hipMemcpyDeviceToHost));
    if (results)
        for (int b = 0; b < B; ++b) {
            const NtMember& m = nm[b];
            scpn_newton_result& r = results[b];
            std::memset(&r, 0, sizeof(r));
            r.converged = m.converged;
            r.iterations = m.iterations;
            r.diverged = m.diverged || ra[b].diverged;
            r.n_hist = std::min(m.n_hist, cap);
            r.n_gs_hist = std::min(m.n_gs_hist, cap);
            r.ls_attempts = m.ls_attempts; r.ls_accepts = m.ls_accepts; r.ls_rejects = m.ls_rejects;
            r.gmres_nonconverged = m.gm_nonconv; r.gmres_breakdown = m.gm_breakdown; r.gmres_last_info = m.gm_last_info;
            r.gmres_inner = m.gm_inner;
            r.residual = m.n_hist > 0 ? m.last_hist : INFINITY;
            r.gs_residual = m.n_gs_hist > 0 ? m.gs : INFINITY;
            r.gs_residual_best = m.n_gs_hist > 0 ? m.gs_best : INFINITY;
        }
}

// ------------------------------------------------------------------------------------------ legacy FastSolver handle
struct FastSolverHip {
    int nr, nz, ld;
    double rmin, rmax, zmin, zmax, dr, dz;
    double boundary_value = 0.0;
    DBuf<double> psi, j, Rtab, c_rp, c_rm, d_last;
    DBuf<unsigned long long> d_bits;
    DBuf<int> d_perf, d_done;
    hipStream_t stream = nullptr;
    ~FastSolverHip() { if (stream) (void)hipStreamDestroy(stream); }
};

static void fast_sweeps(FastSolverHip* s, const double* j_host, double* psi_host, int iterations, double omega, bool use_tol,
                        double tol, int* performed, double* last_delta) {
    hipStream_t st = s->stream;
    const int nr = s->nr, nz = s->nz, ld = s->ld;
    HIP_TRY(hipMemcpy2DAsync(s->j.p, sizeof(double) * ld, j_host, sizeof(double) * nr, sizeof(double) * nr, nz, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(s->d_bits.p, 0, sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(s->d_perf.p, 0, sizeof(int), st));
    HIP_TRY(hipMemsetAsync(s->d_done.p, 0, sizeof(int), st));
    HIP_TRY(hipMemsetAsync(s->d_last.p, 0, sizeof(double), st));
    const double dr_sq = s->dr * s->dr, dz_sq = s->dz * s->dz;
    const double c_z = 1.0 / dz_sq, center = 2.0 / dr_sq + 2.0 / dz_sq, om1 = 1.0 - omega;
    int ncol = (nr - 2 + 1) / 2;
    dim3 g(cdiv(std::max(ncol, 1), 64), cdiv(std::max(nz - 2, 1), 4));
    dim3 gr(cdiv(2L * (nz + nr), 256));
    for (int it = 0; it < iterations; ++it) {
        SCPN_LAUNCH(k_fast_ring, gr, dim3(256), 0, st, s->psi.p, nz, nr, ld, s->boundary_value, (const int*)s->d_done.p);
        if (nz > 2 && nr > 2)
            for (int parity = 0; parity < 2; ++parity)
                SCPN_LAUNCH(k_fast_half, g, dim3(256), 0, st, s->psi.p, (const double*)s->j.p, nz, nr, ld, (const double*)s->Rtab.p,
                                   (const double*)s->c_rp.p, (const double*)s->c_rm.p, c_z, center, omega, om1, parity, s->d_bits.p,
                                   (const int*)s->d_done.p);
        SCPN_LAUNCH(k_fast_ctl, dim3(1), dim3(1), 0, st, s->d_bits.p, s->d_last.p, s->d_perf.p, s->d_done.p, tol, use_tol ? 1 : 0);
    }
    HIP_TRY(hipMemcpy2DAsync(psi_host, sizeof(double) * nr, s->psi.p, sizeof(double) * ld, sizeof(double) * nr, nz, hipMemcpyDeviceToHost, st));
    int perf = 0;
    double last = 0.0;
    HIP_TRY(hipMemcpyAsync(&perf, s->d_perf.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&last, s->d_last.p, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (performed) *performed = perf;
    if (last_delta) *last_delta = last;
}

// ========================================================================================== C ABI
extern "C" {

const char* scpn_last_error(void) { return g_err.c_str(); }
const char* scpn_version(void) { return "scpn_hip 0.1 (gfx950, fp64, fp-contract=off)"; }

int scpn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int scpn_set_device(int device) {
    int n = scpn_device_count();
    if (device < 0 || device >= n) return fail(SCPN_E_DEVICE, "device index out of range");
    g_device = device;
    return hipSetDevice(device) == hipSuccess ? SCPN_OK : fail(SCPN_E_DEVICE, "hipSetDevice failed");
}
int scpn_device_info(char* buf, int buflen) {
    if (!buf || buflen <= 0) return fail(SCPN_E_ARG, "bad buffer");
    if (scpn_device_count() <= 0) return fail(SCPN_E_DEVICE, "no HIP device");
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, g_device) != hipSuccess) return fail(SCPN_E_DEVICE, "hipGetDeviceProperties failed");
    // (the marketing name is empty on boxes without the amdgpu.ids table)
    snprintf(buf, (size_t)buflen, "%s | %s | %d CUs | %.0f GiB | HIP", pr.name[0] ? pr.name : "AMD GPU", pr.gcnArchName, pr.multiProcessorCount,
             (double)pr.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    return SCPN_OK;
}

// ---- resident handle
static int S_create(Solver** out, int nr, int nz, int batch, const double* r_axis, const double* z_axis, double dr, double dz) {
    if (!out || !r_axis || nr < 2 || nz < 2 || batch < 1) return fail(SCPN_E_ARG, "scpn_gs_create: need nr,nz >= 2, batch >= 1");
    if (!(dr > 0.0) || !(dz > 0.0)) return fail(SCPN_E_ARG, "scpn_gs_create: dr, dz must be > 0");
    API_BEGIN
    *out = create_impl(nr, nz, batch, r_axis, z_axis, dr, dz);
    API_END
}
static void S_destroy(Solver* h) {
    if (!h) return;
    (void)hipStreamSynchronize(h->stream);
    delete h;
}
static int check_member(Solver* h, int member) {
    if (!h) return fail(SCPN_E_ARG, "null handle");
    if (member < -1 || member >= h->B) return fail(SCPN_E_ARG, "member out of range");
    return SCPN_OK;
}
static int set_plain(Solver* h, DBuf<double>& buf, int member, const double* host) {
    if (check_member(h, member) || !host) return fail(SCPN_E_ARG, "bad member / null buffer");
    API_BEGIN
    ensure_picard_bufs(h);
    upload(h, buf.p, member, host);
    API_END
}
static int get_plain(Solver* h, DBuf<double>& buf, int member, double* host) {
    if (check_member(h, member) || !host) return fail(SCPN_E_ARG, "bad member / null buffer");
    API_BEGIN
    ensure_picard_bufs(h);
    download(h, buf.p, member, host);
    API_END
}
static int S_set_psi(Solver* h, int member, const double* psi) {
    if (check_member(h, member) || !psi) return fail(SCPN_E_ARG, "bad member / null buffer");
    API_BEGIN
    std::vector<int> cur = host_cur(h);
    for (int b = 0; b < h->B; ++b) {
        if (member >= 0 && b != member) continue;
        const double* src = member >= 0 ? psi : psi + (long)b * h->nz * h->nr;
        HIP_TRY(hipMemcpy2DAsync(psi_member_ptr(h, cur, b), sizeof(double) * h->ld, src, sizeof(double) * h->nr, sizeof(double) * h->nr,
                                 h->nz, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    API_END
}
static int S_get_psi(Solver* h, int member, double* psi) {
    if (check_member(h, member) || !psi) return fail(SCPN_E_ARG, "bad member / null buffer");
    API_BEGIN
    std::vector<int> cur = host_cur(h);
    for (int b = 0; b < h->B; ++b) {
        if (member >= 0 && b != member) continue;
        double* dst = member >= 0 ? psi : psi + (long)b * h->nz * h->nr;
        HIP_TRY(hipMemcpy2DAsync(dst, sizeof(double) * h->nr, psi_member_ptr(h, cur, b), sizeof(double) * h->ld, sizeof(double) * h->nr,
                                 h->nz, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    API_END
}
static int S_get_jphi(Solver* h, int member, double* jphi) { return get_plain(h, h->jphi, member, jphi); }
static int S_set_jphi(Solver* h, int member, const double* jphi) { return set_plain(h, h->jphi, member, jphi); }
static int S_get_source(Solver* h, int member, double* source) { return get_plain(h, h->src, member, source); }
static int S_set_source(Solver* h, int member, const double* source) { return set_plain(h, h->src, member, source); }
static int S_set_boundary_flux(Solver* h, int member, const double* bc) { return set_plain(h, h->bc, member, bc); }
static int S_get_boundary_flux(Solver* h, int member, double* bc) { return get_plain(h, h->bc, member, bc); }
static int S_set_current_target(Solver* h, int member, const double* it) {
    if (check_member(h, member) || !it) return fail(SCPN_E_ARG, "bad member / null buffer");
    if (member >= 0) h->i_target[member] = *it;
    else for (int b = 0; b < h->B; ++b) h->i_target[b] = it[b];
    return SCPN_OK;
}
static int S_set_vacuum_coils(Solver* h, int n, const double* r, const double* z, const double* i) {
    if (!h || n < 0 || (n > 0 && (!r || !z || !i))) return fail(SCPN_E_ARG, "bad coil arrays");
    API_BEGIN
    h->n_vac = n;
    h->vac_valid = false;  // (the cached vacuum field belongs to the previous coil set)
    if (n > 0) {
        size_t c = (size_t)n * h->B;
        h->d_vac_r.alloc(c); h->d_vac_z.alloc(c); h->d_vac_i.alloc(c);
        HIP_TRY(hipMemcpy(h->d_vac_r.p, r, sizeof(double) * c, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->d_vac_z.p, z, sizeof(double) * c, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->d_vac_i.p, i, sizeof(double) * c, hipMemcpyHostToDevice));
    }
    API_END
}
static int S_set_seed(Solver* h, const double* seed) {
    if (!h) return fail(SCPN_E_ARG, "null handle");
    h->seed_uploaded = false;  // the cached device copy / integral belong to the previous seed
    if (!seed) { h->seed_user = false; return SCPN_OK; }
    h->seed_host.assign(seed, seed + (size_t)h->nz * h->nr);
    h->seed_user = true;
    return SCPN_OK;
}

static int S_picard_solve(Solver* h, const scpn_picard_params* p, scpn_picard_result* results, double* hist_diff, double* hist_gs,
                         double* trace) {
    if (!h || !p) return fail(SCPN_E_ARG, "null handle / params");
    if (p->method < 0 || p->method > 3) return fail(SCPN_E_ARG, "method must be 0 (multigrid), 1 (sor), 2 (jacobi) or 3 (anderson)");
    if (p->method == 3 && (p->anderson_m < 1 || p->anderson_m > 5)) return fail(SCPN_E_ARG, "anderson_depth must be in [1, 5]");
    if (p->method != 2 && !omega_ok(p->sor_omega)) return fail(SCPN_E_ARG, "omega must be finite and satisfy 1.0 <= omega < 2.0");
    if (p->require_gs_residual && !(p->gs_tol > 0.0)) return fail(SCPN_E_ARG, "solver.gs_residual_threshold must be > 0");
    if (h->nz < 3 || h->nr < 3) return fail(SCPN_E_ARG, "grid must be at least 3x3");
    API_BEGIN
    picard_solve_impl(h, p, results, hist_diff, hist_gs, trace, nullptr);
    if (p->fail_on_diverge && results)
        for (int b = 0; b < h->B; ++b)
            if (results[b].diverged) return fail(SCPN_E_DIVERGED, "Equilibrium solver diverged");
    API_END
}

static int S_newton_solve(Solver* h, const scpn_picard_params* p, const scpn_newton_opts* o, scpn_newton_result* results, double* hist,
                          double* gs_hist) {
    if (!h || !p || !o) return fail(SCPN_E_ARG, "null handle / params");
    if (!omega_ok(p->sor_omega)) return fail(SCPN_E_ARG, "omega must be finite and satisfy 1.0 <= omega < 2.0");
    if (p->require_gs_residual && !(p->gs_tol > 0.0)) return fail(SCPN_E_ARG, "solver.gs_residual_threshold must be > 0");
    if (o->reserved < 0 || o->reserved > 1) return fail(SCPN_E_ARG, "solver.newton_jacobian must be profile or finite_difference");
    if (o->precond < 0 || o->precond > 2) return fail(SCPN_E_ARG, "solver.gmres_preconditioner must be none, diagonal or multigrid on the device");
    if (o->line_search && (!(o->line_search_c > 0.0) || !(o->line_search_c < 1.0)))
        return fail(SCPN_E_ARG, "solver.newton_line_search_c must be finite and in (0, 1)");
    if (o->line_search && o->max_backtracks <= 0) return fail(SCPN_E_ARG, "solver.newton_line_search_max_backtracks must be >= 1");
    if (o->gmres_restart > GM_R) return fail(SCPN_E_ARG, "gmres_restart must be <= 50");
    if (h->nz < 3 || h->nr < 3) return fail(SCPN_E_ARG, "grid must be at least 3x3");
    API_BEGIN
    newton_solve_impl(h, p, o, results, hist, gs_hist);
    if (p->fail_on_diverge && results)
        for (int b = 0; b < h->B; ++b)
            if (results[b].diverged) return fail(SCPN_E_DIVERGED, "Newton solver diverged");
    API_END
}

static int S_free_boundary_solve(Solver* h, const scpn_picard_params* p, int n_coils, const double* coil_r, const double* coil_z,
                                const double* coil_i, const double* coil_turns, int max_outer, double tol, int* outer_iterations,
                                double* final_diff, int* picard_iters, scpn_picard_result* results, const scpn_newton_opts* newton = nullptr) {
    if (!h || !p) return fail(SCPN_E_ARG, "null handle / params");
    if (max_outer < 1) return fail(SCPN_E_ARG, "max_outer_iter must be >= 1.");
    if (!std::isfinite(tol) || tol < 0.0) return fail(SCPN_E_ARG, "tol must be finite and >= 0.");
    if (n_coils < 0 || (n_coils > 0 && (!coil_r || !coil_z || !coil_i))) return fail(SCPN_E_ARG, "bad coil arrays");
    if (p->method != 2 && !omega_ok(p->sor_omega)) return fail(SCPN_E_ARG, "omega must be finite and satisfy 1.0 <= omega < 2.0");
    API_BEGIN
    ensure_picard_bufs(h);
    const int B = h->B, nz = h->nz, nr = h->nr, ld = h->ld;
    hipStream_t st = h->stream;
    dim3 g2 = grid2d(nr, nz, B);
    dim3 gring(cdiv(2L * (nz + nr), 256), B);
    // psi_ext = compute_external_flux (free_boundary.py:646) -> kept in `ext`, used as boundary_flux
    size_t c = (size_t)std::max(n_coils, 1) * B;
    h->d_fb_r.alloc(c); h->d_fb_z.alloc(c); h->d_fb_i.alloc(c); h->d_fb_t.alloc(c);
    if (n_coils > 0) {
        HIP_TRY(hipMemcpyAsync(h->d_fb_r.p, coil_r, sizeof(double) * n_coils * B, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(h->d_fb_z.p, coil_z, sizeof(double) * n_coils * B, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(h->d_fb_i.p, coil_i, sizeof(double) * n_coils * B, hipMemcpyHostToDevice, st));
        if (coil_turns) HIP_TRY(hipMemcpyAsync(h->d_fb_t.p, coil_turns, sizeof(double) * n_coils * B, hipMemcpyHostToDevice, st));
    }
    Fld extF = h->plain(h->ext.p), bcF = h->plain(h->bc.p), oldF = h->plain(h->old.p);
    {
        ProfScope ps(h, "coil_flux", 1);
        // only the Dirichlet ring of psi_ext is consumed (free_boundary.py:652-655 and the Picard boundary map)
        SCPN_LAUNCH(k_coil_flux, g2, BLK2D, 0, st, extF, nz, nr, ld, h->d_r.p, h->d_z.p, n_coils, h->d_fb_r.p, h->d_fb_z.p,
                           h->d_fb_i.p, coil_turns ? (const double*)h->d_fb_t.p : (const double*)nullptr, 0.0, 1, 1);
    }
    SCPN_LAUNCH(k_copy, g2, BLK2D, 0, st, bcF, extF, nz, nr, ld, (const int*)nullptr);
    scpn_picard_params q = *p;
    q.preserve_initial_state = 1;
    q.has_boundary_flux = 1;
    std::vector<int> mask(B, 1), outer_it(B, 0);
    std::vector<double> fdiff(B, INFINITY), hdiff(B);
    std::vector<scpn_picard_result> res(B), last(B);
    DBuf<double>& d_diff = h->d_scr_diff;  // (scratch in the handle: hipFree would synchronise the other stream group too)
    d_diff.alloc(B);
    const int nblk = h->nblk();
    for (int outer = 0; outer < max_outer; ++outer) {
        bool any = false;
        for (int b = 0; b < B; ++b) any |= mask[b] != 0;
        if (!any) break;
        DBuf<int>& d_mask = h->d_scr_mask;
        d_mask.alloc(B);
        HIP_TRY(hipMemcpyAsync(d_mask.p, mask.data(), sizeof(int) * B, hipMemcpyHostToDevice, st));
        Fld curF = h->slot_fld(h->d_cur.p);
        SCPN_LAUNCH(k_apply_bc, gring, dim3(256), 0, st, curF, extF, nz, nr, ld, (const int*)d_mask.p);
        SCPN_LAUNCH(k_copy, g2, BLK2D, 0, st, oldF, curF, nz, nr, ld, (const int*)d_mask.p);
        // (no host wait here: the inner solve starts with a read-back on the same stream)
        if (newton) {  // inner solve = Newton-Kantorovich (solver_method "newton"): same outer loop, same Dirichlet data
            std::vector<scpn_newton_result> nres(B);
            newton_solve_impl(h, &q, newton, nres.data(), nullptr, nullptr, &mask);
            for (int b = 0; b < B; ++b) {
                std::memset(&res[b], 0, sizeof(res[b]));
                res[b].converged = nres[b].converged; res[b].iterations = nres[b].iterations; res[b].diverged = nres[b].diverged;
                res[b].residual = nres[b].residual; res[b].gs_residual = nres[b].gs_residual; res[b].gs_residual_best = nres[b].gs_residual_best;
            }
        } else {
            picard_solve_impl(h, &q, res.data(), nullptr, nullptr, nullptr, &mask);
        }
        curF = h->slot_fld(h->d_cur.p);
        SCPN_LAUNCH(k_maxabsdiff, dim3(nblk, B), dim3(256), 0, st, curF, oldF, nz, nr, ld, h->d_part1.p, nblk);
        SCPN_LAUNCH(k_max_final, dim3(B), dim3(256), 0, st, d_diff.p, (const double*)h->d_part1.p, nblk);
        HIP_TRY(hipMemcpyAsync(hdiff.data(), d_diff.p, sizeof(double) * B, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int b = 0; b < B; ++b) {
            if (!mask[b]) continue;
            outer_it[b] = outer + 1;
            fdiff[b] = hdiff[b];
            last[b] = res[b];
            if (picard_iters) picard_iters[(long)b * max_outer + outer] = res[b].iterations;
            if (hdiff[b] < tol) mask[b] = 0;
        }
    }
    for (int b = 0; b < B; ++b) {
        if (outer_iterations) outer_iterations[b] = outer_it[b];
        if (final_diff) final_diff[b] = fdiff[b];
        if (results) results[b] = last[b];
    }
    API_END
}

static int S_rb_sor(Solver* h, double omega, int n_sweeps) {
    if (!h) return fail(SCPN_E_ARG, "null handle");
    if (!omega_ok(omega)) return fail(SCPN_E_ARG, "omega must be finite and satisfy 1.0 <= omega < 2.0");
    API_BEGIN
    auto& lv = get_hierarchy(h, 5);
    Fld curF = h->slot_fld(h->d_cur.p);
    Fld res = smooth(h, lv[0]->dev, curF, h->plain(h->W.p), h->plain(h->src.p), omega, n_sweeps, nullptr);
    copy_back(h, curF, res, nullptr);
    HIP_TRY(hipStreamSynchronize(h->stream));
    API_END
}
static int S_vcycle(Solver* h, double omega, int pre, int post, int min_grid) {
    if (!h) return fail(SCPN_E_ARG, "null handle");
    if (!omega_ok(omega)) return fail(SCPN_E_ARG, "omega must be finite and satisfy 1.0 <= omega < 2.0");
    API_BEGIN
    Fld curF = h->slot_fld(h->d_cur.p);
    Fld res = vcycle(h, curF, h->plain(h->W.p), h->plain(h->W2.p), h->plain(h->src.p), omega, pre, post, min_grid, nullptr);
    copy_back(h, curF, res, nullptr);
    HIP_TRY(hipStreamSynchronize(h->stream));
    API_END
}
static int S_compute_b_field(Solver* h, int member, double* b_r, double* b_z) {
    if (!h || member < -1 || member >= h->B || !b_r || !b_z) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    const int nb = member < 0 ? h->B : 1;
    size_t n = (size_t)h->nz * h->nr * nb;
    DBuf<double> dbr, dbz;
    dbr.alloc(n);
    dbz.alloc(n);
    SCPN_LAUNCH(k_b_field, grid2d(h->nr, h->nz, nb), BLK2D, 0, h->stream, dbr.p, dbz.p, h->slot_fld(h->d_cur.p), member, h->nz, h->nr,
                       h->ld, h->dr, h->dz, (const double*)h->d_r.p);
    HIP_TRY(hipMemcpyAsync(b_r, dbr.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(b_z, dbz.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    API_END
}
// map_profiles_to_2d of every member (core/integrated_transport_solver_coupling.py:39-73): device J_phi <- profiles
static int S_map_profiles(Solver* h, double z_min_cfg, int n_rho, const double* rho, const double* p1d, const double* jbs1d,
                          double* p2d_out, double* i_current_out) {
    if (!h || !rho || !p1d || !jbs1d) return fail(SCPN_E_ARG, "null arguments");
    if (n_rho < 2 || n_rho > MAP_MAX_RHO) return fail(SCPN_E_ARG, "n_rho must be in [2, 1024]");
    API_BEGIN
    ensure_picard_bufs(h);
    const int B = h->B, nz = h->nz, nr = h->nr, ld = h->ld;
    const int nblk1 = h->nblk1();
    hipStream_t st = h->stream;
    std::vector<Member> ms(B);
    for (int b = 0; b < B; ++b) { std::memset(&ms[b], 0, sizeof(Member)); ms[b].active = 1; ms[b].i_target = h->i_target[b]; }
    DBuf<Member> d_ms;
    d_ms.alloc(B);
    HIP_TRY(hipMemcpyAsync(d_ms.p, ms.data(), sizeof(Member) * B, hipMemcpyHostToDevice, st));
    DBuf<double> d_rho, d_p, d_j, d_p2, d_part;
    d_rho.alloc(n_rho); d_p.alloc((size_t)n_rho * B); d_j.alloc((size_t)n_rho * B); d_part.alloc((size_t)B * nblk1);
    HIP_TRY(hipMemcpyAsync(d_rho.p, rho, sizeof(double) * n_rho, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_p.p, p1d, sizeof(double) * n_rho * B, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_j.p, jbs1d, sizeof(double) * n_rho * B, hipMemcpyHostToDevice, st));
    if (p2d_out) d_p2.alloc((size_t)B * nz * nr);
    int n_mask_rows = 0;
    for (int iz = 0; iz < nz; ++iz) if ((z_min_cfg * 0.5) > h->z_axis[iz]) n_mask_rows = iz + 1;
    Fld curF = h->slot_fld(h->d_cur.p), jF = h->plain(h->jphi.p);
    {
        ProfScope ps(h, "map_profiles", 4);
        SCPN_LAUNCH(k_topology, dim3(nblk1, B), dim3(256), 0, st, curF, (const Member*)d_ms.p, nz, nr, ld, h->dr, h->dz, n_mask_rows,
                           h->d_topo.p, nblk1);
        SCPN_LAUNCH(k_map_topo_final, dim3(B), dim3(256), 0, st, curF, d_ms.p, (const TopoPart*)h->d_topo.p, nblk1, nr, ld);
        SCPN_LAUNCH(k_map_profiles, dim3(nblk1, B), dim3(256), 0, st, jF, p2d_out ? d_p2.p : (double*)nullptr, curF, (const Member*)d_ms.p,
                           nz, nr, ld, (const double*)h->d_r.p, n_rho, (const double*)d_rho.p, (const double*)d_p.p, (const double*)d_j.p, d_part.p,
                           nblk1);
        SCPN_LAUNCH(k_map_scale, dim3(nblk1, B), dim3(256), 0, st, jF, d_ms.p, nz, nr, ld, h->dr, h->dz, (const double*)d_part.p, nblk1);
    }
    if (p2d_out) HIP_TRY(hipMemcpyAsync(p2d_out, d_p2.p, sizeof(double) * B * nz * nr, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(ms.data(), d_ms.p, sizeof(Member) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (i_current_out) for (int b = 0; b < B; ++b) i_current_out[b] = ms[b].i_current;
    API_END
}
// interp_psi of every member at n_pts points (core/fusion_kernel_free_boundary.py:156-159, 562-580); out [B][n_pts]
static int S_sample_psi(Solver* h, int n_pts, const double* pts_r, const double* pts_z, double* out) {
    if (!h || n_pts < 1 || !pts_r || !pts_z || !out) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    DBuf<double> dr_, dz_, dout;
    dr_.alloc(n_pts); dz_.alloc(n_pts); dout.alloc((size_t)n_pts * h->B);
    HIP_TRY(hipMemcpyAsync(dr_.p, pts_r, sizeof(double) * n_pts, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dz_.p, pts_z, sizeof(double) * n_pts, hipMemcpyHostToDevice, h->stream));
    SCPN_LAUNCH(k_sample_psi, dim3(cdiv(n_pts, 128), h->B), dim3(128), 0, h->stream, dout.p, h->slot_fld(h->d_cur.p), h->nz, h->nr, h->ld,
                       h->dr, h->dz, (const double*)h->d_r.p, (const double*)h->d_z.p, n_pts, (const double*)dr_.p, (const double*)dz_.p);
    HIP_TRY(hipMemcpyAsync(out, dout.p, sizeof(double) * n_pts * h->B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    API_END
}
// the transport coupling's relative flux change without field downloads: a device copy of the current iterate, and later the two
// norms of (current, copy) from per-workgroup partial sums added on the host in index order
static int S_snapshot_psi(Solver* h) {
    if (!h) return fail(SCPN_E_ARG, "null handle");
    API_BEGIN
    h->d_snap.alloc((size_t)h->stride * h->B);
    SCPN_LAUNCH(k_copy, grid2d(h->nr, h->nz, h->B), BLK2D, 0, h->stream, h->plain(h->d_snap.p), h->slot_fld(h->d_cur.p), h->nz, h->nr, h->ld,
                       (const int*)nullptr);
    h->snap_valid = true;
    API_END
}
static int S_flux_change(Solver* h, double* rel_change) {
    if (!h || !rel_change) return fail(SCPN_E_ARG, "bad arguments");
    if (!h->snap_valid) return fail(SCPN_E_ARG, "scpn_gs_flux_change: no snapshot (call scpn_gs_snapshot_psi first)");
    API_BEGIN
    const int nblk = cdiv((long)h->nz * h->nr, 1024);
    h->d_fluxpart.alloc((size_t)h->B * nblk * 2);
    SCPN_LAUNCH(k_flux_change, dim3(nblk, h->B), dim3(256), 0, h->stream, h->slot_fld(h->d_cur.p), h->plain(h->d_snap.p), h->nz, h->nr, h->ld,
                       h->d_fluxpart.p, nblk);
    std::vector<double> part((size_t)h->B * nblk * 2);
    HIP_TRY(hipMemcpyAsync(part.data(), h->d_fluxpart.p, sizeof(double) * part.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int b = 0; b < h->B; ++b) {
        double so = 0.0, sd = 0.0;
        for (int i = 0; i < nblk; ++i) { so += part[((size_t)b * nblk + i) * 2]; sd += part[((size_t)b * nblk + i) * 2 + 1]; }
        double norm_old = std::sqrt(so);
        if (norm_old < 1e-30) norm_old = 1.0;  // integrated_transport_solver_coupling.py:133-134
        rel_change[b] = std::sqrt(sd) / norm_old;
    }
    API_END
}
static int S_fill_psi(Solver* h, double value) {
    if (!h) return fail(SCPN_E_ARG, "null handle");
    API_BEGIN
    SCPN_LAUNCH(k_fill, grid2d(h->nr, h->nz, h->B), BLK2D, 0, h->stream, h->slot_fld(h->d_cur.p), value, h->nz, h->nr, h->ld,
                       (const int*)nullptr);
    HIP_TRY(hipStreamSynchronize(h->stream));
    API_END
}
static int S_pack_psi_into(Solver* h, double* dst, bool wait = true) {  // dst: device buffer [B][nz][nr] (dense)
    if (!h || !dst) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    size_t n = (size_t)h->nz * h->nr;
    std::vector<int> cur = host_cur(h);
    for (int b = 0; b < h->B; ++b)
        HIP_TRY(hipMemcpy2DAsync(dst + n * b, sizeof(double) * h->nr, psi_member_ptr(h, cur, b), sizeof(double) * h->ld,
                                 sizeof(double) * h->nr, h->nz, hipMemcpyDeviceToDevice, h->stream));
    if (wait) HIP_TRY(hipStreamSynchronize(h->stream));  // (otherwise the caller orders its consumer behind h->stream with an event)
    API_END
}
static int S_sync(Solver* h) {
    if (!h) return fail(SCPN_E_ARG, "null handle");
    API_BEGIN
    HIP_TRY(hipStreamSynchronize(h->stream));
    API_END
}
static int S_profile(Solver* h, int enable) {
    if (!h) return fail(SCPN_E_ARG, "null handle");
    API_BEGIN
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto& e : h->prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    h->prof.clear();
    h->prof_on = enable != 0;
    API_END
}
static int S_profile_names(Solver* h, char* buf, int buflen) {
    if (!h || !buf || buflen <= 0) return fail(SCPN_E_ARG, "bad arguments");
    std::string all;
    for (auto& n : h->prof_names) { if (!all.empty()) all += ","; all += n; }
    snprintf(buf, (size_t)buflen, "%s", all.c_str());
    return SCPN_OK;
}
static int S_profile_read(Solver* h, const char* name, double* total_ms, long* launches) {
    if (!h || !name) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    HIP_TRY(hipStreamSynchronize(h->stream));
    double tot = 0.0;
    long n = 0;
    for (auto& e : h->prof) {
        if (h->prof_names[e.id] != name) continue;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
        tot += ms;
        n += e.launches;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    API_END
}

// ---- public handle: the ensemble is split into independent groups, each a Solver with its own HIP stream
//      driven by its own host thread, so the latency-bound coarse-level kernels of one group overlap the
//      bandwidth-bound fine-level kernels of another (SCPN_HIP_GROUPS overrides the group count).
}  // extern "C" (reopened below)
struct scpn_gs {
    std::vector<Solver*> kids;
    std::vector<int> off;  // first member of each group; off.back() == B
    int B = 0, nr = 0, nz = 0;
    DBuf<double> dense;
    ~scpn_gs() { for (auto* k : kids) { if (k) { (void)hipStreamSynchronize(k->stream); delete k; } } }
};
template <class F>
static int run_groups(scpn_gs* g, F f) {
    const int n = (int)g->kids.size();
    bool profiling = false;  // kernel timing wants isolated kernels: run the groups one after the other
    for (auto* k : g->kids) profiling |= k->prof_on;
    if (n == 1 || profiling) {
        for (int gi = 0; gi < n; ++gi) {
            int rc = f(gi, g->kids[gi]);
            if (rc) return rc;
        }
        return SCPN_OK;
    }
    std::vector<int> rc(n, 0);
    std::vector<std::string> err(n);
    std::vector<std::thread> th;
    for (int gi = 0; gi < n; ++gi)
        th.emplace_back([&, gi] {
            (void)hipSetDevice(g_device);
            rc[gi] = f(gi, g->kids[gi]);
            if (rc[gi]) err[gi] = g_err;
        });
    for (auto& t : th) t.join();
    for (int gi = 0; gi < n; ++gi)
        if (rc[gi]) return fail(rc[gi], err[gi]);
    return SCPN_OK;
}
template <class F>
static int each_group(scpn_gs* g, F f) {  // sequential
    for (size_t gi = 0; gi < g->kids.size(); ++gi) {
        int rc = f((int)gi, g->kids[gi]);
        if (rc) return rc;
    }
    return SCPN_OK;
}
// member -> (group, local member); member == -1 handled by the callers
static int locate(scpn_gs* g, int member, int* local) {
    for (size_t gi = 0; gi + 1 < g->off.size(); ++gi)
        if (member < g->off[gi + 1]) { *local = member - g->off[gi]; return (int)gi; }
    return -1;
}
template <class FM, class FA>
static int field_op(scpn_gs* g, int member, size_t per_member, FM one, FA /*unused*/) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    if (member < -1 || member >= g->B) return fail(SCPN_E_ARG, "member out of range");
    if (member >= 0) {
        int local = 0, gi = locate(g, member, &local);
        return one(g->kids[gi], local, (size_t)0);
    }
    return each_group(g, [&](int gi, Solver* k) { return one(k, -1, per_member * (size_t)g->off[gi]); });
}
extern "C" {

int scpn_gs_create(scpn_gs** out, int nr, int nz, int batch, const double* r_axis, const double* z_axis, double dr, double dz) {
    if (!out || !r_axis || nr < 2 || nz < 2 || batch < 1) return fail(SCPN_E_ARG, "scpn_gs_create: need nr,nz >= 2, batch >= 1");
    if (!(dr > 0.0) || !(dz > 0.0)) return fail(SCPN_E_ARG, "scpn_gs_create: dr, dz must be > 0");
    int ng = batch >= 16 ? 2 : 1;
    if (const char* e = std::getenv("SCPN_HIP_GROUPS")) ng = std::max(1, std::atoi(e));
    ng = std::min(ng, batch);
    auto* g = new scpn_gs();
    g->B = batch; g->nr = nr; g->nz = nz;
    int done = 0;
    for (int gi = 0; gi < ng; ++gi) {
        int cnt = (batch - done + (ng - gi) - 1) / (ng - gi);
        Solver* k = nullptr;
        int rc = S_create(&k, nr, nz, cnt, r_axis, z_axis, dr, dz);
        if (rc) { delete g; return rc; }
        k->ensemble_B = batch;
        // (the environment default applies to resident handles only: the stateless operator API builds its solvers through
        //  create_impl and is always exact)
        const char* a = std::getenv("SCPN_HIP_ARITH");
        k->arith = (a && (!std::strcmp(a, "1") || !std::strcmp(a, "contracted") || !std::strcmp(a, "fast"))) ? 1 : 0;
        g->kids.push_back(k);
        g->off.push_back(done);
        done += cnt;
    }
    g->off.push_back(done);
    *out = g;
    return SCPN_OK;
}
void scpn_gs_destroy(scpn_gs* g) { delete g; }

#define FIELD_SETTER(NAME, SNAME)                                                                                      \
    int NAME(scpn_gs* g, int member, const double* a) {                                                                \
        if (!a) return fail(SCPN_E_ARG, "null buffer");                                                                \
        size_t pm = g ? (size_t)g->nz * g->nr : 0;                                                                     \
        return field_op(g, member, pm, [&](Solver* k, int m, size_t o) { return SNAME(k, m, a + o); }, 0);             \
    }
#define FIELD_GETTER(NAME, SNAME)                                                                                      \
    int NAME(scpn_gs* g, int member, double* a) {                                                                      \
        if (!a) return fail(SCPN_E_ARG, "null buffer");                                                                \
        size_t pm = g ? (size_t)g->nz * g->nr : 0;                                                                     \
        return field_op(g, member, pm, [&](Solver* k, int m, size_t o) { return SNAME(k, m, a + o); }, 0);             \
    }
FIELD_SETTER(scpn_gs_set_psi, S_set_psi)
FIELD_GETTER(scpn_gs_get_psi, S_get_psi)
FIELD_SETTER(scpn_gs_set_jphi, S_set_jphi)
FIELD_GETTER(scpn_gs_get_jphi, S_get_jphi)
FIELD_SETTER(scpn_gs_set_source, S_set_source)
FIELD_GETTER(scpn_gs_get_source, S_get_source)
FIELD_SETTER(scpn_gs_set_boundary_flux, S_set_boundary_flux)
FIELD_GETTER(scpn_gs_get_boundary_flux, S_get_boundary_flux)

int scpn_gs_set_current_target(scpn_gs* g, int member, const double* it) {
    if (!g || !it) return fail(SCPN_E_ARG, "bad arguments");
    if (member < -1 || member >= g->B) return fail(SCPN_E_ARG, "member out of range");
    if (member >= 0) { int local = 0, gi = locate(g, member, &local); return S_set_current_target(g->kids[gi], local, it); }
    return each_group(g, [&](int gi, Solver* k) { return S_set_current_target(k, -1, it + g->off[gi]); });
}
int scpn_gs_set_vacuum_coils(scpn_gs* g, int n, const double* r, const double* z, const double* i) {
    if (!g || n < 0 || (n > 0 && (!r || !z || !i))) return fail(SCPN_E_ARG, "bad coil arrays");
    return each_group(g, [&](int gi, Solver* k) {
        size_t o = (size_t)g->off[gi] * n;
        return S_set_vacuum_coils(k, n, n ? r + o : nullptr, n ? z + o : nullptr, n ? i + o : nullptr);
    });
}
int scpn_gs_set_seed(scpn_gs* g, const double* seed) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    return each_group(g, [&](int, Solver* k) { return S_set_seed(k, seed); });
}
int scpn_gs_picard_solve(scpn_gs* g, const scpn_picard_params* p, scpn_picard_result* results, double* hist_diff, double* hist_gs,
                         double* trace) {
    if (!g || !p) return fail(SCPN_E_ARG, "null handle / params");
    const size_t mi = (size_t)std::max(p->max_iter, 0);
    return run_groups(g, [&](int gi, Solver* k) {
        const size_t o = (size_t)g->off[gi];
        return S_picard_solve(k, p, results ? results + o : nullptr, hist_diff ? hist_diff + o * mi : nullptr,
                              hist_gs ? hist_gs + o * mi : nullptr, trace ? trace + o * mi * 6 : nullptr);
    });
}
int scpn_gs_newton_solve(scpn_gs* g, const scpn_picard_params* p, const scpn_newton_opts* o, scpn_newton_result* results, double* hist,
                         double* gs_hist) {
    if (!g || !p || !o) return fail(SCPN_E_ARG, "null handle / params");
    const size_t cap = (size_t)std::max(p->max_iter, 0) + 2;
    return run_groups(g, [&](int gi, Solver* k) {
        const size_t of = (size_t)g->off[gi];
        return S_newton_solve(k, p, o, results ? results + of : nullptr, hist ? hist + of * cap : nullptr, gs_hist ? gs_hist + of * cap : nullptr);
    });
}
int scpn_gs_free_boundary_solve(scpn_gs* g, const scpn_picard_params* p, int n_coils, const double* coil_r, const double* coil_z,
                                const double* coil_i, const double* coil_turns, int max_outer, double tol, int* outer_iterations,
                                double* final_diff, int* picard_iters, scpn_picard_result* results) {
    if (!g || !p) return fail(SCPN_E_ARG, "null handle / params");
    if (max_outer < 1) return fail(SCPN_E_ARG, "max_outer_iter must be >= 1.");
    return run_groups(g, [&](int gi, Solver* k) {
        const size_t o = (size_t)g->off[gi], oc = o * (size_t)std::max(n_coils, 0);
        return S_free_boundary_solve(k, p, n_coils, coil_r ? coil_r + oc : nullptr, coil_z ? coil_z + oc : nullptr,
                                     coil_i ? coil_i + oc : nullptr, coil_turns ? coil_turns + oc : nullptr, max_outer, tol,
                                     outer_iterations ? outer_iterations + o : nullptr, final_diff ? final_diff + o : nullptr,
                                     picard_iters ? picard_iters + o * max_outer : nullptr, results ? results + o : nullptr);
    });
}
int scpn_gs_free_boundary_solve_newton(scpn_gs* g, const scpn_picard_params* p, const scpn_newton_opts* o, int n_coils, const double* coil_r,
                                       const double* coil_z, const double* coil_i, const double* coil_turns, int max_outer, double tol,
                                       int* outer_iterations, double* final_diff, int* inner_iters, scpn_picard_result* results) {
    if (!g || !p || !o) return fail(SCPN_E_ARG, "null handle / params");
    if (max_outer < 1) return fail(SCPN_E_ARG, "max_outer_iter must be >= 1.");
    if (o->precond < 0 || o->precond > 2 || o->reserved < 0 || o->reserved > 1) return fail(SCPN_E_ARG, "bad Newton options");
    return run_groups(g, [&](int gi, Solver* k) {
        const size_t of = (size_t)g->off[gi], oc = of * (size_t)std::max(n_coils, 0);
        return S_free_boundary_solve(k, p, n_coils, coil_r ? coil_r + oc : nullptr, coil_z ? coil_z + oc : nullptr,
                                     coil_i ? coil_i + oc : nullptr, coil_turns ? coil_turns + oc : nullptr, max_outer, tol,
                                     outer_iterations ? outer_iterations + of : nullptr, final_diff ? final_diff + of : nullptr,
                                     inner_iters ? inner_iters + of * max_outer : nullptr, results ? results + of : nullptr, o);
    });
}
int scpn_gs_rb_sor(scpn_gs* g, double omega, int n_sweeps) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    return run_groups(g, [&](int, Solver* k) { return S_rb_sor(k, omega, n_sweeps); });
}
int scpn_gs_vcycle(scpn_gs* g, double omega, int pre, int post, int min_grid) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    return run_groups(g, [&](int, Solver* k) { return S_vcycle(k, omega, pre, post, min_grid); });
}
int scpn_gs_compute_b_field(scpn_gs* g, int member, double* b_r, double* b_z) {
    if (!g || member < -1 || member >= g->B) return fail(SCPN_E_ARG, "bad arguments");
    if (member < 0) {
        const size_t n = (size_t)g->nz * g->nr;
        return each_group(g, [&](int gi, Solver* k) { return S_compute_b_field(k, -1, b_r + n * g->off[gi], b_z + n * g->off[gi]); });
    }
    int local = 0, gi = locate(g, member, &local);
    return S_compute_b_field(g->kids[gi], local, b_r, b_z);
}
int scpn_gs_map_profiles(scpn_gs* g, double z_min_cfg, int n_rho, const double* rho, const double* pressure_1d, const double* jbs_1d,
                         double* pressure_2d_out, double* i_current_out) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    const size_t n = (size_t)g->nz * g->nr;
    return each_group(g, [&](int gi, Solver* k) {
        const size_t o = (size_t)g->off[gi];
        return S_map_profiles(k, z_min_cfg, n_rho, rho, pressure_1d ? pressure_1d + o * n_rho : nullptr, jbs_1d ? jbs_1d + o * n_rho : nullptr,
                              pressure_2d_out ? pressure_2d_out + o * n : nullptr, i_current_out ? i_current_out + o : nullptr);
    });
}
int scpn_gs_sample_psi(scpn_gs* g, int n_pts, const double* pts_r, const double* pts_z, double* out) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    return each_group(g, [&](int gi, Solver* k) { return S_sample_psi(k, n_pts, pts_r, pts_z, out ? out + (size_t)n_pts * g->off[gi] : nullptr); });
}
#ifdef CV_STAMPS
int scpn_debug_tile_stamps(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -2;
}
int scpn_debug_cv_stamps(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cv_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -2;
}
#endif
int scpn_gs_num_groups(scpn_gs* g) { return g ? (int)g->kids.size() : 0; }
int scpn_gs_set_arith(scpn_gs* g, int mode) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    if (mode != 0 && mode != 1) return fail(SCPN_E_ARG, "scpn_gs_set_arith: mode must be 0 (exact) or 1 (contracted)");
    for (auto* k : g->kids) k->arith = mode;
    return SCPN_OK;
}
int scpn_gs_get_arith(scpn_gs* g) { return (g && !g->kids.empty()) ? g->kids[0]->arith : 0; }
int scpn_gs_snapshot_psi(scpn_gs* g) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    return each_group(g, [&](int, Solver* k) { return S_snapshot_psi(k); });
}
int scpn_gs_flux_change(scpn_gs* g, double* rel_change) {
    if (!g || !rel_change) return fail(SCPN_E_ARG, "bad arguments");
    return each_group(g, [&](int gi, Solver* k) { return S_flux_change(k, rel_change + g->off[gi]); });
}
int scpn_gs_fill_psi(scpn_gs* g, double value) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    return each_group(g, [&](int, Solver* k) { return S_fill_psi(k, value); });
}
int scpn_gs_pack_psi_device(scpn_gs* g, void** device_ptr) {
    if (!g || !device_ptr) return fail(SCPN_E_ARG, "bad arguments");
    try {
        g->dense.alloc((size_t)g->B * g->nz * g->nr);
    } catch (const std::exception& e) { return fail(SCPN_E_DEVICE, e.what()); }
    int rc = each_group(g, [&](int gi, Solver* k) { return S_pack_psi_into(k, g->dense.p + (size_t)g->off[gi] * g->nz * g->nr); });
    if (rc) return rc;
    *device_ptr = g->dense.p;
    return SCPN_OK;
}
// ------------------------------------------------------------------------------------------ ensemble gather over RCCL
#define RCCL_TRY(expr)                                                                                              \
    do {                                                                                                            \
        ncclResult_t r__ = (expr);                                                                                  \
        if (r__ != ncclSuccess)                                                                                     \
            throw std::runtime_error(std::string(#expr) + ": " + (comm::state().api.GetErrorString ? comm::state().api.GetErrorString(r__) : "rccl error")); \
    } while (0)

int scpn_comm_unique_id(char* id128) {
    if (!id128) return fail(SCPN_E_ARG, "scpn_comm_unique_id: null buffer");
    API_BEGIN
    comm::State& c = comm::state();
    if (const char* miss = comm::load(c.api)) throw std::runtime_error(miss);
    ncclUniqueId id;
    static_assert(sizeof(id) == SCPN_COMM_ID_BYTES, "ncclUniqueId size");
    RCCL_TRY(c.api.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    API_END
}
// Everything scpn_comm_init does BEFORE its collective call -- librccl loaded and complete, this rank's device selectable, NCCL 2.x
// ABI -- without entering the collective: the host exchanges the ranks' verdicts on this step first, so that a rank that cannot
// even get this far never leaves its peers blocked inside ncclCommInitRank (scpn_fusion_core_amd/distributed.py).
static void comm_preflight(comm::State& c, int device, int* version) {
    if (const char* miss = comm::load(c.api)) throw std::runtime_error(miss);
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev)
        throw std::runtime_error("device " + std::to_string(device) + " is not visible to this rank (" + std::to_string(n_dev) + " device(s))");
    HIP_TRY(hipSetDevice(device));
    g_device = device;
    int ver = 0;
    RCCL_TRY(c.api.GetVersion(&ver));
    if (ver < 20000) throw std::runtime_error("librccl reports version " + std::to_string(ver) + ": the NCCL 2.x ABI is required");
    if (version) *version = ver;
}
int scpn_comm_prepare(int device, int* rccl_version) {
    API_BEGIN
    comm_preflight(comm::state(), device, rccl_version);
    API_END
}
int scpn_comm_init(int rank, int world, const char* id128, int device) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(SCPN_E_ARG, "scpn_comm_init: need 0 <= rank < world and an id");
    API_BEGIN
    comm::State& c = comm::state();
    if (c.comm) throw std::runtime_error("scpn_comm_init: communicator already initialised");
    comm_preflight(c, device, nullptr);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    RCCL_TRY(c.api.CommInitRank(&c.comm, world, id, rank));
    HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void**)&c.d_scalar, 2 * sizeof(double)));
    c.rank = rank; c.world = world; c.device = device;
    API_END
}
int scpn_comm_info(int* rank, int* world, int* rccl_version) {
    API_BEGIN
    comm::State& c = comm::state();
    if (!c.comm) throw std::runtime_error("communicator not initialised");
    int r = -1, w = 0, v = 0;
    RCCL_TRY(c.api.CommUserRank(c.comm, &r));
    RCCL_TRY(c.api.CommCount(c.comm, &w));
    RCCL_TRY(c.api.GetVersion(&v));
    if (rank) *rank = r;
    if (world) *world = w;
    if (rccl_version) *rccl_version = v;
    API_END
}
// pack the local members' psi densely and all-gather the blocks of all ranks into recv_device ([world][per][nz][nr] doubles).
// `per` (>= this handle's batch) is the block size EVERY rank sends: a rank that owns fewer members (the short last block of an
// uneven ensemble) pads with zeros, so the collective counts agree on all ranks and recv_device holds `world` whole blocks.
// The pack copies run on the solver groups' streams and the collective on the communicator's own stream, ordered behind them by
// events; the only host wait is the read-back of the members' slot indices on each group's stream (S_pack_psi_into), so the
// gather itself overlaps the next solve (scpn_comm_wait).  The send block is the communicator's OWN buffer: it is only ever
// grown after the communicator's stream has drained, and nothing else (scpn_gs_pack_psi_device stages in the handle's buffer)
// writes to it, so a pack or a larger gather issued while a collective is in flight cannot touch what that collective reads.
int scpn_comm_allgather_psi(scpn_gs* g, int per, void* recv_device) {
    if (!g || !recv_device) return fail(SCPN_E_ARG, "scpn_comm_allgather_psi: null handle / buffer");
    if (per < g->B) return fail(SCPN_E_ARG, "scpn_comm_allgather_psi: the common block is smaller than this rank's batch");
    API_BEGIN
    comm::State& c = comm::state();
    if (!c.comm) throw std::runtime_error("communicator not initialised");
    const size_t n = (size_t)g->nz * g->nr;
    // the previous gather still reads the send block: it has to finish before the block is refilled (or re-allocated)
    HIP_TRY(hipStreamSynchronize(c.stream));
    if (c.send_cap < (size_t)per * n) {
        if (c.send) { HIP_TRY(hipFree(c.send)); c.send = nullptr; c.send_cap = 0; }
        HIP_TRY(hipSetDevice(c.device));
        HIP_TRY(hipMalloc((void**)&c.send, sizeof(double) * (size_t)per * n));
        c.send_cap = (size_t)per * n;
    }
    while (c.ready.size() < g->kids.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c.ready.push_back(e);
    }
    if (per > g->B) HIP_TRY(hipMemsetAsync(c.send + (size_t)g->B * n, 0, sizeof(double) * (per - g->B) * n, c.stream));
    for (size_t gi = 0; gi < g->kids.size(); ++gi) {
        if (int rc = S_pack_psi_into(g->kids[gi], c.send + (size_t)g->off[gi] * n, false)) return rc;
        HIP_TRY(hipEventRecord(c.ready[gi], g->kids[gi]->stream));
        HIP_TRY(hipStreamWaitEvent(c.stream, c.ready[gi], 0));
    }
    RCCL_TRY(c.api.AllGather(c.send, recv_device, (size_t)per * n, ncclDouble, c.comm, c.stream));
    API_END
}
int scpn_comm_wait(void) {
    API_BEGIN
    comm::State& c = comm::state();
    if (c.stream) HIP_TRY(hipStreamSynchronize(c.stream));
    API_END
}
// in-place all-reduce (max) of one host double over the ranks; doubles as the barrier of the timed region
int scpn_comm_allreduce_max(double* value) {
    if (!value) return fail(SCPN_E_ARG, "scpn_comm_allreduce_max: null value");
    API_BEGIN
    comm::State& c = comm::state();
    if (!c.comm) throw std::runtime_error("communicator not initialised");
    HIP_TRY(hipMemcpyAsync(c.d_scalar, value, sizeof(double), hipMemcpyHostToDevice, c.stream));
    RCCL_TRY(c.api.AllReduce(c.d_scalar, c.d_scalar + 1, 1, ncclDouble, ncclMax, c.comm, c.stream));
    HIP_TRY(hipMemcpyAsync(value, c.d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    API_END
}
int scpn_comm_finalize(void) {
    API_BEGIN
    comm::State& c = comm::state();
    if (c.stream) { (void)hipStreamSynchronize(c.stream); }
    if (c.comm) { (void)c.api.CommDestroy(c.comm); c.comm = nullptr; }
    if (c.d_scalar) { (void)hipFree(c.d_scalar); c.d_scalar = nullptr; }
    if (c.send) { (void)hipFree(c.send); c.send = nullptr; c.send_cap = 0; }
    for (hipEvent_t e : c.ready) (void)hipEventDestroy(e);
    c.ready.clear();
    if (c.stream) { (void)hipStreamDestroy(c.stream); c.stream = nullptr; }
    c.rank = -1; c.world = 0;
    API_END
}
// plain device buffers for the gathered ensemble (the caller owns them)
int scpn_device_alloc(long bytes, void** ptr) {
    if (!ptr || bytes < 0) return fail(SCPN_E_ARG, "scpn_device_alloc: bad arguments");
    API_BEGIN
    HIP_TRY(hipSetDevice(g_device));
    HIP_TRY(hipMalloc(ptr, (size_t)std::max<long>(bytes, 1)));
    API_END
}
int scpn_device_free(void* ptr) {
    API_BEGIN
    if (ptr) HIP_TRY(hipFree(ptr));
    API_END
}
int scpn_device_download(void* host, const void* device_ptr, long bytes) {
    if (!host || !device_ptr || bytes < 0) return fail(SCPN_E_ARG, "scpn_device_download: bad arguments");
    API_BEGIN
    HIP_TRY(hipMemcpy(host, device_ptr, (size_t)bytes, hipMemcpyDeviceToHost));
    API_END
}

int scpn_gs_sync(scpn_gs* g) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    return each_group(g, [&](int, Solver* k) { return S_sync(k); });
}
int scpn_gs_profile(scpn_gs* g, int enable) {
    if (!g) return fail(SCPN_E_ARG, "null handle");
    return each_group(g, [&](int, Solver* k) { return S_profile(k, enable); });
}
int scpn_gs_profile_names(scpn_gs* g, char* buf, int buflen) {
    if (!g || !buf || buflen <= 0) return fail(SCPN_E_ARG, "bad arguments");
    std::string all;
    for (auto* k : g->kids)
        for (auto& n : k->prof_names)
            if (("," + all + ",").find("," + n + ",") == std::string::npos) { if (!all.empty()) all += ","; all += n; }
    snprintf(buf, (size_t)buflen, "%s", all.c_str());
    return SCPN_OK;
}
int scpn_gs_profile_read(scpn_gs* g, const char* name, double* total_ms, long* launches) {
    if (!g || !name) return fail(SCPN_E_ARG, "bad arguments");
    double tot = 0.0;
    long n = 0;
    int rc = each_group(g, [&](int, Solver* k) {
        double t = 0.0;
        long c = 0;
        int r = S_profile_read(k, name, &t, &c);
        tot += t;
        n += c;
        return r;
    });
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return rc;
}

// ---- operator library on host buffers -------------------------------------------------------
// The stateless operators run on a solver object that lives for one call.  Building one costs ~5 ms (a dozen device allocations,
// the coefficient tables of every multigrid level) against 0.1-1 ms of work for a 257^2 V-cycle, so the last few are kept and
// handed to the next call on the SAME grid -- the reference's GPU-tier provider keeps its PyGpuSolver instances the same way, keyed
// by (nr, nz, r_left, r_right, z_bottom, z_top) (core/_multi_compat_providers.py:720,755-792).  A pooled solver is only ever used by
// one call at a time (it leaves the pool while in use); every call uploads all of its inputs, so nothing of a previous call is
// read.  SCPN_HIP_OPCACHE=0 turns the pool off; objects above 4 GB are never kept.
static std::atomic<long> g_pool_hits{0}, g_pool_misses{0};
struct TempSolver {
    Solver* h = nullptr;
    static std::mutex& mu() { static std::mutex m; return m; }
    static std::vector<Solver*>& idle() { static std::vector<Solver*> v; return v; }
    static bool enabled() {
        static const bool on = [] { const char* e = std::getenv("SCPN_HIP_OPCACHE"); return !(e && e[0] == '0'); }();
        return on;
    }
    static bool same_grid(const Solver* k, int nr, int nz, int batch, const double* r_axis, const double* z_axis, double dr, double dz) {
        if (k->nr != nr || k->nz != nz || k->B != batch || k->dr != dr || k->dz != dz || k->device != g_device) return false;
        if (std::memcmp(k->r_axis.data(), r_axis, sizeof(double) * nr) != 0) return false;
        for (int i = 0; i < nz; ++i)
            if (k->z_axis[i] != (z_axis ? z_axis[i] : i * dz)) return false;
        return true;
    }
    TempSolver(int nr, int nz, int batch, const double* r_axis, const double* z_axis, double dr, double dz) {
        if (enabled()) {
            std::lock_guard<std::mutex> lk(mu());
            auto& v = idle();
            for (size_t i = 0; i < v.size(); ++i)
                if (same_grid(v[i], nr, nz, batch, r_axis, z_axis, dr, dz)) { h = v[i]; v.erase(v.begin() + (long)i); break; }
        }
        (h ? g_pool_hits : g_pool_misses)++;
        if (h) {  // back to the state create_impl leaves (the fields are overwritten by the call's uploads)
            h->arith = 0; h->prof_on = false; h->lazy_src = nullptr; h->jsum_part = nullptr; h->snap_valid = false;
            h->vac_valid = false; h->n_vac = 0;
            std::fill(h->i_target.begin(), h->i_target.end(), 0.0);
        } else {
            h = create_impl(nr, nz, batch, r_axis, z_axis, dr, dz);
        }
        // The stateless operators share ONE stream (created once per process and device): every live stream is mapped onto one of the
        // few hardware queues, and a stream per pooled solver made the stream groups of a resident handle share theirs (BASELINE config 4
        // inside bench.py: 1 083 -> 693 member-steps/s with four parked streams) while creating / destroying a stream per call costs
        // milliseconds.  Concurrent stateless calls from several host threads are serialised on it.
        static const bool share_env = [] { const char* e = std::getenv("SCPN_HIP_OPSTREAM"); return !(e && e[0] == '0'); }();
        if (enabled() && share_env) {
            static std::mutex sm;
            static hipStream_t shared = nullptr;
            static int shared_dev = -1;
            std::lock_guard<std::mutex> lk(sm);
            if (!shared && hipStreamCreate(&shared) == hipSuccess) shared_dev = g_device;
            if (shared && shared_dev == g_device && h->stream != shared) {
                if (h->stream && h->owns_stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
                h->stream = shared;
                h->owns_stream = false;
            }
        }
    }
    ~TempSolver() {
        if (!h) return;
        const bool healthy = stream_wait(h->stream) == hipSuccess;
        const bool small = (double)h->stride * h->B * sizeof(double) * 12.0 < 4.0e9;
        if (enabled() && healthy && small && std::uncaught_exceptions() == 0) {
            Solver* evict = nullptr;
            {
                std::lock_guard<std::mutex> lk(mu());
                auto& v = idle();
                v.push_back(h);
                if (v.size() > 4) { evict = v.front(); v.erase(v.begin()); }
            }
            delete evict;
        } else {
            delete h;
        }
    }
};
static int check_grid(const void* a, const void* b, const double* r_axis, int nz, int nr, int batch, double dr, double dz) {
    if (!a || !b || !r_axis) return fail(SCPN_E_ARG, "null buffer");
    if (nz < 2 || nr < 2 || batch < 1) return fail(SCPN_E_ARG, "need nz, nr >= 2 and batch >= 1");
    if (!(dr > 0.0) || !(dz > 0.0) || !std::isfinite(dr) || !std::isfinite(dz)) return fail(SCPN_E_ARG, "dr, dz must be finite and > 0");
    return SCPN_OK;
}

// pool statistics of the stateless operators (tests / tools): hits, misses, idle solvers
int scpn_operator_pool_stats(long* hits, long* misses, int* idle) {
    if (hits) *hits = g_pool_hits.load();
    if (misses) *misses = g_pool_misses.load();
    if (idle) { std::lock_guard<std::mutex> lk(TempSolver::mu()); *idle = (int)TempSolver::idle().size(); }
    return SCPN_OK;
}
int scpn_mg_smooth(double* psi, const double* source, const double* r_axis, int nz, int nr, int batch, double dr, double dz, double omega,
                   int n_sweeps) {
    if (!omega_ok(omega)) return fail(SCPN_E_ARG, "omega must be finite and satisfy 1.0 <= omega < 2.0");
    if (check_grid(psi, source, r_axis, nz, nr, batch, dr, dz)) return SCPN_E_ARG;
    if (n_sweeps < 0) return fail(SCPN_E_ARG, "n_sweeps must be >= 0");
    API_BEGIN
    TempSolver t(nr, nz, batch, r_axis, nullptr, dr, dz);
    upload(t.h, t.h->psi_slots.p, -1, psi);
    upload(t.h, t.h->src.p, -1, source);
    auto& lv = get_hierarchy(t.h, 1 << 30);
    Fld res = smooth(t.h, lv[0]->dev, t.h->slot_fld(nullptr), t.h->plain(t.h->W.p), t.h->plain(t.h->src.p), omega, n_sweeps, nullptr);
    copy_back(t.h, t.h->slot_fld(nullptr), res, nullptr);
    download(t.h, t.h->psi_slots.p, -1, psi);
    API_END
}
int scpn_mg_residual(const double* psi, const double* source, const double* r_axis, int nz, int nr, int batch, double dr, double dz,
                     double* out) {
    if (check_grid(psi, source, r_axis, nz, nr, batch, dr, dz) || !out) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    TempSolver t(nr, nz, batch, r_axis, nullptr, dr, dz);
    upload(t.h, t.h->psi_slots.p, -1, psi);
    upload(t.h, t.h->src.p, -1, source);
    auto& lv = get_hierarchy(t.h, 1 << 30);
    SCPN_LAUNCH(k_residual, grid2d(nr, nz, batch), BLK2D, 0, t.h->stream, t.h->plain(t.h->W.p), t.h->slot_fld(nullptr),
                       t.h->plain(t.h->src.p), lv[0]->dev);
    download(t.h, t.h->W.p, -1, out);
    API_END
}
static void resid_norms_host(Solver* h, Fld psi, Fld src, std::vector<double>& mx, std::vector<double>& ss) {
    auto& lv = get_hierarchy(h, 1 << 30);
    const int nb = std::max(h->nblk_int(), 1), B = h->B;
    mx.assign(B, 0.0);
    ss.assign(B, 0.0);
    if (h->nz < 3 || h->nr < 3) return;
    SCPN_LAUNCH(k_resid_norms, dim3(nb, B), dim3(256), 0, h->stream, psi, src, lv[0]->dev, h->d_part1.p, h->d_part2.p, nb, (const int*)nullptr);
    std::vector<double> pm((size_t)B * nb), ps((size_t)B * nb);
    HIP_TRY(hipMemcpyAsync(pm.data(), h->d_part1.p, sizeof(double) * B * nb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(ps.data(), h->d_part2.p, sizeof(double) * B * nb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int b = 0; b < B; ++b) {  // (m_b, s_b = sum (r/m_b)^2) -> M = max m_b, ss = sum (r/M)^2
        for (int i = 0; i < nb; ++i) mx[b] = std::fmax(mx[b], pm[(size_t)b * nb + i]);
        if (mx[b] > 0.0)
            for (int i = 0; i < nb; ++i) {
                const double f = pm[(size_t)b * nb + i] / mx[b];
                ss[b] += ps[(size_t)b * nb + i] * (f * f);
            }
    }
}
int scpn_residual_linf(const double* psi, const double* source, const double* r_axis, int nz, int nr, int batch, double dr, double dz,
                       double* linf) {
    if (check_grid(psi, source, r_axis, nz, nr, batch, dr, dz) || !linf) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    TempSolver t(nr, nz, batch, r_axis, nullptr, dr, dz);
    upload(t.h, t.h->psi_slots.p, -1, psi);
    upload(t.h, t.h->src.p, -1, source);
    std::vector<double> mx, ss;
    resid_norms_host(t.h, t.h->slot_fld(nullptr), t.h->plain(t.h->src.p), mx, ss);
    for (int b = 0; b < batch; ++b) linf[b] = mx[b];
    API_END
}
int scpn_gs_residual_rms(const double* psi, const double* source, const double* r_axis, int nz, int nr, int batch, double dr, double dz,
                         double* rms) {
    if (check_grid(psi, source, r_axis, nz, nr, batch, dr, dz) || !rms) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    TempSolver t(nr, nz, batch, r_axis, nullptr, dr, dz);
    upload(t.h, t.h->psi_slots.p, -1, psi);
    upload(t.h, t.h->src.p, -1, source);
    std::vector<double> mx, ss;
    resid_norms_host(t.h, t.h->slot_fld(nullptr), t.h->plain(t.h->src.p), mx, ss);
    long ni = (long)(nz - 2) * (nr - 2);
    for (int b = 0; b < batch; ++b) rms[b] = (ni <= 0 || mx[b] <= 0.0) ? 0.0 : mx[b] * std::sqrt(ss[b] / (double)ni);  // stable_rms
    API_END
}
int scpn_restrict_full_weight(const double* fine, int nz_f, int nr_f, int batch, double* coarse) {
    if (!fine || !coarse || nz_f < 2 || nr_f < 2 || batch < 1) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    HIP_TRY(hipSetDevice(g_device));
    int nz_c = (nz_f + 1) / 2, nr_c = (nr_f + 1) / 2;
    size_t nf = (size_t)nz_f * nr_f, nc = (size_t)nz_c * nr_c;
    DBuf<double> df, dc;
    df.alloc(nf * batch);
    dc.alloc(nc * batch);
    HIP_TRY(hipMemcpy(df.p, fine, sizeof(double) * nf * batch, hipMemcpyHostToDevice));
    SCPN_LAUNCH(k_restrict, grid2d(nr_c, nz_c, batch), BLK2D, 0, 0, Fld{dc.p, nullptr, (long)nc, batch},
                       Fld{df.p, nullptr, (long)nf, batch}, nz_f, nr_f, nr_f, nz_c, nr_c, nr_c);
    HIP_TRY(hipMemcpy(coarse, dc.p, sizeof(double) * nc * batch, hipMemcpyDeviceToHost));
    API_END
}
int scpn_prolongate_bilinear(const double* coarse, int nz_c, int nr_c, int nz_f, int nr_f, int batch, double* fine) {
    if (!fine || !coarse || nz_f < 1 || nr_f < 1 || batch < 1) return fail(SCPN_E_ARG, "bad arguments");
    if (nz_c != (nz_f + 1) / 2 || nr_c != (nr_f + 1) / 2)
        return fail(SCPN_E_ARG, "coarse shape must be ((nz_f+1)//2, (nr_f+1)//2)");
    API_BEGIN
    HIP_TRY(hipSetDevice(g_device));
    size_t nf = (size_t)nz_f * nr_f, nc = (size_t)nz_c * nr_c;
    DBuf<double> df, dc;
    df.alloc(nf * batch);
    dc.alloc(nc * batch);
    HIP_TRY(hipMemcpy(dc.p, coarse, sizeof(double) * nc * batch, hipMemcpyHostToDevice));
    SCPN_LAUNCH(k_prolong<false>, grid2d(nr_f, nz_f, batch), BLK2D, 0, 0, Fld{df.p, nullptr, (long)nf, batch},
                       Fld{dc.p, nullptr, (long)nc, batch}, nz_f, nr_f, nr_f, nz_c, nr_c, nr_c, (const int*)nullptr);
    HIP_TRY(hipMemcpy(fine, df.p, sizeof(double) * nf * batch, hipMemcpyDeviceToHost));
    API_END
}
int scpn_multigrid_vcycle(double* psi, const double* source, const double* r_axis, int nz, int nr, int batch, double dr, double dz,
                          double omega, int pre, int post, int min_grid) {
    if (!omega_ok(omega)) return fail(SCPN_E_ARG, "omega must be finite and satisfy 1.0 <= omega < 2.0");
    if (check_grid(psi, source, r_axis, nz, nr, batch, dr, dz)) return SCPN_E_ARG;
    if (pre < 0 || post < 0) return fail(SCPN_E_ARG, "smoothing counts must be >= 0");
    API_BEGIN
    TempSolver t(nr, nz, batch, r_axis, nullptr, dr, dz);
    upload(t.h, t.h->psi_slots.p, -1, psi);
    upload(t.h, t.h->src.p, -1, source);
    Fld res = vcycle(t.h, t.h->slot_fld(nullptr), t.h->plain(t.h->W.p), t.h->plain(t.h->W2.p), t.h->plain(t.h->src.p), omega, pre, post,
                     min_grid, nullptr);
    copy_back(t.h, t.h->slot_fld(nullptr), res, nullptr);
    download(t.h, t.h->psi_slots.p, -1, psi);
    API_END
}
int scpn_multigrid_solve(const double* source, double* psi, const double* r_axis, const double* z_axis, int nr, int nz, int batch,
                         double tol, int max_cycles, double omega, int pre, int post, int min_grid, double* residual, int* n_cycles,
                         int* converged) {
    if (!source || !psi || !r_axis || !z_axis) return fail(SCPN_E_ARG, "null buffer");
    if (!(std::isfinite(tol) && tol > 0.0)) return fail(SCPN_E_ARG, "tol must be finite and > 0.");
    if (max_cycles < 1) return fail(SCPN_E_ARG, "max_cycles must be >= 1.");
    if (!omega_ok(omega)) return fail(SCPN_E_ARG, "omega must be finite and satisfy 1.0 <= omega < 2.0");
    if (nr < 2 || nz < 2 || batch < 1) return fail(SCPN_E_ARG, "need nz, nr >= 2 and batch >= 1");
    API_BEGIN
    // multigrid_solve.py:431-432: dr = r_axis[1]-r_axis[0]
    double dr = r_axis[1] - r_axis[0], dz = z_axis[1] - z_axis[0];
    TempSolver t(nr, nz, batch, r_axis, z_axis, dr, dz);
    Solver* h = t.h;
    ensure_picard_bufs(h);
    upload(h, h->psi_slots.p, -1, psi);
    upload(h, h->src.p, -1, source);
    hipStream_t st = h->stream;
    const int B = batch;
    Fld psiF = h->slot_fld(nullptr), srcF = h->plain(h->src.p), bcF = h->plain(h->bc.p);
    SCPN_LAUNCH(k_copy, grid2d(nr, nz, B), BLK2D, 0, st, bcF, psiF, nz, nr, h->ld, (const int*)nullptr);
    auto& lv0 = get_hierarchy(h, 1 << 30);
    const int nb = std::max(h->nblk_int(), 1);
    std::vector<MgMember> ms(B);
    std::vector<int> ones(B, 1);
    for (auto& m : ms) { m.residual = 0; m.n_cycles = 0; m.converged = 0; m.active = 1; m.pad = 0; }
    HIP_TRY(hipMemcpyAsync(h->d_mg.p, ms.data(), sizeof(MgMember) * B, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->d_active.p, ones.data(), sizeof(int) * B, hipMemcpyHostToDevice, st));
    h->h_nactive[0] = B;
    HIP_TRY(hipMemcpyAsync(h->d_nactive.p, h->h_nactive, sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(h->d_part1.p, 0, sizeof(double) * B * h->nblk(), st));
    auto norms = [&](int count_cycle) {
        if (nz >= 3 && nr >= 3)
            SCPN_LAUNCH(k_resid_norms, dim3(nb, B), dim3(256), 0, st, psiF, srcF, lv0[0]->dev, h->d_part1.p, h->d_part2.p, nb,
                               (const int*)h->d_active.p);
        SCPN_LAUNCH(k_mg_final, dim3(B), dim3(256), 0, st, h->d_mg.p, (const double*)h->d_part1.p, nb, tol, max_cycles, count_cycle,
                           h->d_active.p, h->d_nactive.p);
    };
    norms(0);
    int n_act = B;
    for (int c = 0; c < max_cycles; ++c) {
        HIP_TRY(hipMemcpyAsync(h->h_nactive, h->d_nactive.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        n_act = h->h_nactive[0];
        if (n_act <= 0) break;
        Fld res = vcycle(h, psiF, h->plain(h->W.p), h->plain(h->W2.p), srcF, omega, pre, post, min_grid, h->d_active.p);
        copy_back(h, psiF, res, (const int*)h->d_active.p);
        SCPN_LAUNCH(k_apply_bc, dim3(cdiv(2L * (nz + nr), 256), B), dim3(256), 0, st, psiF, bcF, nz, nr, h->ld, (const int*)h->d_active.p);
        norms(1);
    }
    HIP_TRY(hipMemcpyAsync(ms.data(), h->d_mg.p, sizeof(MgMember) * B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b) {
        if (residual) residual[b] = ms[b].residual;
        if (n_cycles) n_cycles[b] = ms[b].n_cycles;
        if (converged) converged[b] = ms[b].converged;
    }
    download(h, h->psi_slots.p, -1, psi);
    API_END
}
int scpn_jacobi_step(const double* psi, const double* source, const double* r_axis, int nz, int nr, int batch, double dr, double dz,
                     double* out) {
    if (check_grid(psi, source, r_axis, nz, nr, batch, dr, dz) || !out) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    TempSolver t(nr, nz, batch, r_axis, nullptr, dr, dz);
    upload(t.h, t.h->psi_slots.p, -1, psi);
    upload(t.h, t.h->src.p, -1, source);
    auto& lv = get_hierarchy(t.h, 1 << 30);
    SCPN_LAUNCH(k_jacobi, grid2d(nr, nz, batch), BLK2D, 0, t.h->stream, t.h->plain(t.h->W.p), t.h->slot_fld(nullptr),
                       t.h->plain(t.h->src.p), lv[0]->dev, (const int*)nullptr);
    download(t.h, t.h->W.p, -1, out);
    API_END
}
int scpn_sor_step(const double* psi, const double* source, const double* r_axis, int nz, int nr, int batch, double dr, double dz,
                  double omega, double* out) {
    if (!omega_ok(omega)) return fail(SCPN_E_ARG, "omega must be finite and satisfy 1.0 <= omega < 2.0");
    if (check_grid(psi, source, r_axis, nz, nr, batch, dr, dz) || !out) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    TempSolver t(nr, nz, batch, r_axis, nullptr, dr, dz);
    Solver* h = t.h;
    ensure_picard_bufs(h);
    upload(h, h->psi_slots.p, -1, psi);
    upload(h, h->src.p, -1, source);
    auto& lv = get_hierarchy(h, 1 << 30);
    dim3 g2 = grid2d(nr, nz, batch);
    SCPN_LAUNCH(k_sanitize, g2, BLK2D, 0, h->stream, h->plain(h->W.p), h->slot_fld(nullptr), nz, nr, h->ld);
    SCPN_LAUNCH(k_sanitize, g2, BLK2D, 0, h->stream, h->plain(h->src2.p), h->plain(h->src.p), nz, nr, h->ld);
    smooth(h, lv[0]->dev, h->plain(h->W.p), Fld{nullptr, nullptr, 0, 0}, h->plain(h->src2.p), omega, 1, nullptr, true);
    download(h, h->W.p, -1, out);
    API_END
}
int scpn_coil_flux(const double* r_axis, const double* z_axis, int nz, int nr, int batch, int n_coils, const double* coil_r,
                   const double* coil_z, const double* coil_i, const double* coil_turns, double mu0, int mode, double* out) {
    if (!r_axis || !z_axis || !out || nz < 1 || nr < 1 || batch < 1 || n_coils < 0) return fail(SCPN_E_ARG, "bad arguments");
    if (n_coils > 0 && (!coil_r || !coil_z || !coil_i)) return fail(SCPN_E_ARG, "null coil arrays");
    if (mode != 0 && mode != 1) return fail(SCPN_E_ARG, "mode must be 0 or 1");
    API_BEGIN
    HIP_TRY(hipSetDevice(g_device));
    size_t n = (size_t)nz * nr, c = (size_t)std::max(n_coils, 1) * batch;
    // Scratch of the calling thread on this device, grown on demand and kept: the shape-optimisation loop calls this once per outer
    // iteration, and seven hipMalloc / hipFree pairs were ~7 of the 8 ms of a 513 x 513 call.  (Never destroyed: no hipFree from a
    // thread-exit destructor after the runtime has shut down.)
    struct Scratch { DBuf<double> dout, dr_, dz_, cr, cz, ci, ct; };
    static thread_local std::map<int, Scratch*>* pools = new std::map<int, Scratch*>();
    Scratch*& sp = (*pools)[g_device];
    if (!sp) sp = new Scratch();
    DBuf<double>&dout = sp->dout, &dr_ = sp->dr_, &dz_ = sp->dz_, &cr = sp->cr, &cz = sp->cz, &ci = sp->ci, &ct = sp->ct;
    dout.alloc(n * batch); dr_.alloc(nr); dz_.alloc(nz); cr.alloc(c); cz.alloc(c); ci.alloc(c); ct.alloc(c);
    HIP_TRY(hipMemcpy(dr_.p, r_axis, sizeof(double) * nr, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dz_.p, z_axis, sizeof(double) * nz, hipMemcpyHostToDevice));
    if (n_coils > 0) {
        HIP_TRY(hipMemcpy(cr.p, coil_r, sizeof(double) * n_coils * batch, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(cz.p, coil_z, sizeof(double) * n_coils * batch, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ci.p, coil_i, sizeof(double) * n_coils * batch, hipMemcpyHostToDevice));
        if (coil_turns) HIP_TRY(hipMemcpy(ct.p, coil_turns, sizeof(double) * n_coils * batch, hipMemcpyHostToDevice));
    }
    SCPN_LAUNCH(k_coil_flux, grid2d(nr, nz, batch), BLK2D, 0, 0, Fld{dout.p, nullptr, (long)n, batch}, nz, nr, nr, (const double*)dr_.p,
                       (const double*)dz_.p, n_coils, (const double*)cr.p, (const double*)cz.p, (const double*)ci.p,
                       coil_turns ? (const double*)ct.p : (const double*)nullptr, mu0, mode);
    HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * n * batch, hipMemcpyDeviceToHost));
    API_END
}
// Device layout of the batched one-thread-per-system kernels (k_gtsv, k_current_diffusion): [n][batch] (default; the ABI's
// [batch][n] arrays are transposed on the device on the way in and out) or, with SCPN_HIP_F4_LAYOUT=rows, the [batch][n] layout
// of rounds 1-5 (every access of a wave strided by n doubles) -- kept for the before / after timing of tools/config4_split.py.
static bool f4_transposed() {
    static const bool t = [] { const char* e = std::getenv("SCPN_HIP_F4_LAYOUT"); return !(e && std::string(e) == "rows"); }();
    return t;
}
static void f4_transpose(double* dst, const double* src, int rows, int cols, int arrays = 1) {  // dst[c][r] = src[r][c], `arrays` of them back to back
    SCPN_LAUNCH(k_transpose, dim3(cdiv(cols, 32), cdiv(rows, 32), arrays), dim3(256), 0, 0, dst, src, rows, cols);
}
int scpn_tridiag_solve(int batch, int n, const double* dl, const double* d, const double* du, const double* b, double* x, int* info) {
    if (batch < 1 || n < 1 || !d || !b || !x || (n > 1 && (!dl || !du))) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available");
    HIP_TRY(hipSetDevice(g_device));
    const size_t tot = (size_t)batch * n;
    DBuf<double> w;  // dl, d, du, b padded to n per system
    DBuf<int> di;
    w.alloc(4 * tot);
    di.alloc(batch);
    HIP_TRY(hipMemset(w.p, 0, sizeof(double) * 4 * tot));
    if (n > 1) {
        HIP_TRY(hipMemcpy2D(w.p, sizeof(double) * n, dl, sizeof(double) * (n - 1), sizeof(double) * (n - 1), batch, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy2D(w.p + 2 * tot, sizeof(double) * n, du, sizeof(double) * (n - 1), sizeof(double) * (n - 1), batch, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(w.p + tot, d, sizeof(double) * tot, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(w.p + 3 * tot, b, sizeof(double) * tot, hipMemcpyHostToDevice));
    if (f4_transposed()) {  // [batch][n] -> [n][batch]: lane s of a wave reads consecutive addresses at every step of the recurrence
        DBuf<double> wt;
        wt.alloc(4 * tot);
        f4_transpose(wt.p, w.p, batch, n, 4);  // dl, d, du, b in one launch
        SCPN_LAUNCH(k_gtsv, dim3(cdiv(batch, 64)), dim3(64), 0, 0, batch, n, wt.p, wt.p + tot, wt.p + 2 * tot, wt.p + 3 * tot, di.p, 1L, (long)batch);
        f4_transpose(w.p + 3 * tot, wt.p + 3 * tot, n, batch);
        HIP_TRY(hipDeviceSynchronize());
    } else {
        SCPN_LAUNCH(k_gtsv, dim3(cdiv(batch, 64)), dim3(64), 0, 0, batch, n, w.p, w.p + tot, w.p + 2 * tot, w.p + 3 * tot, di.p, (long)n, 1L);
    }
    HIP_TRY(hipMemcpy(x, w.p + 3 * tot, sizeof(double) * tot, hipMemcpyDeviceToHost));
    if (info) HIP_TRY(hipMemcpy(info, di.p, sizeof(int) * batch, hipMemcpyDeviceToHost));
    API_END
}
int scpn_current_diffusion_step(int batch, int n, const double* rho, double* psi, double r0, double a, double b0, double dt, const double* te,
                                const double* ne, double z_eff, const double* j_bs, const double* j_cd, const double* j_ext, int* info) {
    (void)b0;
    (void)ne;  // B0 and n_e enter only q / the collisionality, which the reference evaluates and discards
    if (batch < 1 || n < 3 || !rho || !psi || !te || !j_bs || !j_cd) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available");
    HIP_TRY(hipSetDevice(g_device));
    const size_t tot = (size_t)batch * n;
    DBuf<double> drho_, dpsi, dte, djb, djc, dje, scratch;
    DBuf<int> di;
    drho_.alloc(n); dpsi.alloc(tot); dte.alloc(tot); djb.alloc(tot); djc.alloc(tot); scratch.alloc(4 * tot); di.alloc(batch);
    HIP_TRY(hipMemcpy(drho_.p, rho, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dpsi.p, psi, sizeof(double) * tot, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dte.p, te, sizeof(double) * tot, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(djb.p, j_bs, sizeof(double) * tot, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(djc.p, j_cd, sizeof(double) * tot, hipMemcpyHostToDevice));
    if (j_ext) { dje.alloc(tot); HIP_TRY(hipMemcpy(dje.p, j_ext, sizeof(double) * tot, hipMemcpyHostToDevice)); }
    CdConst c;
    const double mu0 = (4.0 * 3.141592653589793) * 1e-7;  // core/current_diffusion.py:20
    c.r0 = r0; c.a = a; c.dt = dt; c.z_eff = z_eff; c.mu0_a2 = mu0 * (a * a);
    const dim3 ga(cdiv((long)tot, 256));
    if (f4_transposed()) {  // rows assembled straight into the [n][batch] scratch; the solved rhs rows come back through one transpose
        SCPN_LAUNCH(k_cd_assemble, ga, dim3(256), 0, 0, batch, n, (const double*)drho_.p, (const double*)dpsi.p, c, (const double*)dte.p,
                           (const double*)djb.p, (const double*)djc.p, j_ext ? (const double*)dje.p : (const double*)nullptr, scratch.p, 1L, (long)batch);
        SCPN_LAUNCH(k_cd_solve, dim3(cdiv(batch, 64)), dim3(64), 0, 0, batch, n, scratch.p, di.p, 1L, (long)batch);
        f4_transpose(dpsi.p, scratch.p + 3 * tot, n, batch);
    } else {
        SCPN_LAUNCH(k_cd_assemble, ga, dim3(256), 0, 0, batch, n, (const double*)drho_.p, (const double*)dpsi.p, c, (const double*)dte.p,
                           (const double*)djb.p, (const double*)djc.p, j_ext ? (const double*)dje.p : (const double*)nullptr, scratch.p, (long)n, 1L);
        SCPN_LAUNCH(k_cd_solve, dim3(cdiv(batch, 64)), dim3(64), 0, 0, batch, n, scratch.p, di.p, (long)n, 1L);
        HIP_TRY(hipMemcpy(dpsi.p, scratch.p + 3 * tot, sizeof(double) * tot, hipMemcpyDeviceToDevice));
    }
    HIP_TRY(hipMemcpy(psi, dpsi.p, sizeof(double) * tot, hipMemcpyDeviceToHost));
    if (info) HIP_TRY(hipMemcpy(info, di.p, sizeof(int) * batch, hipMemcpyDeviceToHost));
    API_END
}
int scpn_flux_evolution(int batch, int n, int n_steps, double dt, const double* rate, const double* source, const double* psi0,
                        double* psi, double* inc, double* drop, double* resid) {
    if (batch < 1 || n < 1 || n_steps < 1 || !rate || !source || !psi0 || !psi) return fail(SCPN_E_ARG, "bad arguments");
    if (!(dt > 0.0) || !std::isfinite(dt)) return fail(SCPN_E_ARG, "dt must be positive");
    API_BEGIN
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available");
    HIP_TRY(hipSetDevice(g_device));
    const size_t n1 = (size_t)batch * (n_steps + 1) * n, n0 = (size_t)batch * n_steps * n;
    DBuf<double> d_rate, d_src, d_psi, d_inc, d_drop, d_res;
    d_rate.alloc(n1); d_src.alloc(n1); d_psi.alloc(n1); d_inc.alloc(n0); d_drop.alloc(n0); d_res.alloc(n0);
    HIP_TRY(hipMemcpy(d_rate.p, rate, sizeof(double) * n1, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_src.p, source, sizeof(double) * n1, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy2D(d_psi.p, sizeof(double) * (n_steps + 1) * n, psi0, sizeof(double) * n, sizeof(double) * n, batch, hipMemcpyHostToDevice));
    SCPN_LAUNCH(k_flux_evolution, dim3(cdiv(n, 64), batch), dim3(64), 0, 0, batch, n, n_steps, dt, (const double*)d_rate.p,
                       (const double*)d_src.p, d_psi.p, d_inc.p, d_drop.p, d_res.p);
    HIP_TRY(hipMemcpy(psi, d_psi.p, sizeof(double) * n1, hipMemcpyDeviceToHost));
    if (inc) HIP_TRY(hipMemcpy(inc, d_inc.p, sizeof(double) * n0, hipMemcpyDeviceToHost));
    if (drop) HIP_TRY(hipMemcpy(drop, d_drop.p, sizeof(double) * n0, hipMemcpyDeviceToHost));
    if (resid) HIP_TRY(hipMemcpy(resid, d_res.p, sizeof(double) * n0, hipMemcpyDeviceToHost));
    API_END
}
int scpn_probe_response(int n_coils, const double* coil_r, const double* coil_z, const double* coil_turns, int n_flux, const double* flux_r,
                        const double* flux_z, int n_b, const double* b_r, const double* b_z, const int* b_dir, double* out) {
    if (n_coils < 1 || !coil_r || !coil_z || !out) return fail(SCPN_E_ARG, "CoilSet.positions must contain at least one coil.");
    if (n_flux < 0 || n_b < 0 || n_flux + n_b < 1) return fail(SCPN_E_ARG, "At least one flux point or B probe point must be provided.");
    if ((n_flux > 0 && (!flux_r || !flux_z)) || (n_b > 0 && (!b_r || !b_z || !b_dir))) return fail(SCPN_E_ARG, "null probe arrays");
    for (int i = 0; i < n_b; ++i)
        if (b_dir[i] != 0 && b_dir[i] != 1) return fail(SCPN_E_ARG, "b_probe_directions entries must be 'R' or 'Z'.");
    API_BEGIN
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available");
    HIP_TRY(hipSetDevice(g_device));
    const int rows = n_flux + n_b;
    DBuf<double> cr, cz, ct, fr, fz, br, bz, dout;
    DBuf<int> bd;
    auto up = [&](DBuf<double>& d, const double* src, int n) { d.alloc(std::max(n, 1)); if (n > 0) HIP_TRY(hipMemcpy(d.p, src, sizeof(double) * n, hipMemcpyHostToDevice)); };
    up(cr, coil_r, n_coils); up(cz, coil_z, n_coils);
    if (coil_turns) up(ct, coil_turns, n_coils);
    up(fr, flux_r, n_flux); up(fz, flux_z, n_flux); up(br, b_r, n_b); up(bz, b_z, n_b);
    bd.alloc(std::max(n_b, 1));
    if (n_b > 0) HIP_TRY(hipMemcpy(bd.p, b_dir, sizeof(int) * n_b, hipMemcpyHostToDevice));
    dout.alloc((size_t)rows * n_coils);
    SCPN_LAUNCH(k_probe_response, dim3(cdiv(rows, 64), n_coils), dim3(64), 0, 0, dout.p, n_coils, (const double*)cr.p, (const double*)cz.p,
                       coil_turns ? (const double*)ct.p : (const double*)nullptr, n_flux, (const double*)fr.p, (const double*)fz.p, n_b,
                       (const double*)br.p, (const double*)bz.p, (const int*)bd.p);
    HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * rows * n_coils, hipMemcpyDeviceToHost));
    API_END
}
int scpn_green_points(int n_coils, const double* coil_r, const double* coil_z, int n_pts, const double* pts_r, const double* pts_z,
                      double* out) {
    if (n_coils < 1 || n_pts < 1 || !coil_r || !coil_z || !pts_r || !pts_z || !out) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    HIP_TRY(hipSetDevice(g_device));
    DBuf<double> cr, cz, pr, pz, dout;
    cr.alloc(n_coils); cz.alloc(n_coils); pr.alloc(n_pts); pz.alloc(n_pts); dout.alloc((size_t)n_coils * n_pts);
    HIP_TRY(hipMemcpy(cr.p, coil_r, sizeof(double) * n_coils, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(cz.p, coil_z, sizeof(double) * n_coils, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pr.p, pts_r, sizeof(double) * n_pts, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pz.p, pts_z, sizeof(double) * n_pts, hipMemcpyHostToDevice));
    SCPN_LAUNCH(k_green_points, dim3(cdiv(n_pts, 256), n_coils), dim3(256), 0, 0, dout.p, n_coils, (const double*)cr.p,
                       (const double*)cz.p, n_pts, (const double*)pr.p, (const double*)pz.p);
    HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * n_coils * n_pts, hipMemcpyDeviceToHost));
    API_END
}
int scpn_gs_solve_np(double r_min, double r_max, double z_min, double z_max, int nr, int nz, const double* r_axis, const double* z_axis,
                     const double* psi_seed, double ip_target, double mu0, int n_picard, int n_jacobi, double alpha, double omega_j,
                     double beta_mix, double* psi_out) {
    (void)r_max; (void)z_min; (void)z_max;
    if (!r_axis || !z_axis || !psi_out || nr < 3 || nz < 3 || n_picard < 0 || n_jacobi < 0) return fail(SCPN_E_ARG, "bad arguments");
    API_BEGIN
    double dr = r_axis[1] - r_axis[0], dz = z_axis[1] - z_axis[0];
    TempSolver t(nr, nz, 1, r_axis, z_axis, dr, dz);
    Solver* h = t.h;
    ensure_picard_bufs(h);
    std::vector<double> seed((size_t)nz * nr);
    if (psi_seed) std::copy(psi_seed, psi_seed + (size_t)nz * nr, seed.begin());
    else {
        double r_center = 0.5 * (r_min + r_axis[nr - 1]);
        for (int iz = 0; iz < nz; ++iz)
            for (int ir = 0; ir < nr; ++ir) {
                double a = r_axis[ir] - r_center;
                bool ring = iz == 0 || iz == nz - 1 || ir == 0 || ir == nr - 1;
                seed[(size_t)iz * nr + ir] = ring ? 0.0 : std::exp(-(a * a) / 0.5) * 0.01;
            }
    }
    upload(h, h->psi_slots.p, 0, seed.data());
    // jax_gs_solver.py:67-74 uses dR*dR (not dR**2); identical to pow(dR, 2) in IEEE arithmetic
    auto& lv = get_hierarchy(h, 1 << 30);
    Lvl L = lv[0]->dev;
    hipStream_t st = h->stream;
    Fld psiF = h->slot_fld(nullptr), srcF = h->plain(h->src.p), aF = h->plain(h->W.p), bF = h->plain(h->src2.p);
    dim3 g2 = grid2d(nr, nz, 1);
    const int nb = h->nblk(), nbi = std::max(h->nblk_int(), 1);
    for (int it = 0; it < n_picard; ++it) {
        SCPN_LAUNCH(k_max_interior, dim3(nbi, 1), dim3(256), 0, st, psiF, nz, nr, h->ld, h->d_part1.p, nbi);
        SCPN_LAUNCH(k_np_final, dim3(1), dim3(256), 0, st, h->d_scal.p, (const double*)h->d_part1.p, nbi, 0, dr, dz, ip_target);
        SCPN_LAUNCH(k_np_jsum, dim3(nb, 1), dim3(256), 0, st, psiF, (const double*)h->d_scal.p, nz, nr, h->ld, (const double*)h->d_r.p, mu0,
                           beta_mix, h->d_part2.p, nb);
        SCPN_LAUNCH(k_np_final, dim3(1), dim3(256), 0, st, h->d_scal.p, (const double*)h->d_part2.p, nb, 1, dr, dz, ip_target);
        SCPN_LAUNCH(k_np_source, g2, BLK2D, 0, st, srcF, psiF, (const double*)h->d_scal.p, nz, nr, h->ld, (const double*)h->d_r.p, mu0,
                           beta_mix);
        SCPN_LAUNCH(k_copy, g2, BLK2D, 0, st, aF, psiF, nz, nr, h->ld, (const int*)nullptr);
        Fld in = aF, out = bF;
        int j = 0;
        if (nz >= 16 && nr >= 16) {  // five temporally blocked steps per launch (register tile, as the seed relaxation): 40 launches for 200 steps
            const dim3 gj(cdiv(nr, 116), cdiv(nz, 36), 1);
            for (; j + 5 <= n_jacobi; j += 5) {
                SCPN_LAUNCH((k_jacobi_reg<5, true>), gj, dim3(512), 0, st, out, in, srcF, L, (const int*)nullptr, omega_j, 1.0 - omega_j);
                std::swap(in, out);
            }
        }
        for (; j < n_jacobi; ++j) {
            SCPN_LAUNCH(k_jacobi_damped, g2, BLK2D, 0, st, out, in, srcF, L, omega_j, 1.0 - omega_j);
            std::swap(in, out);
        }
        SCPN_LAUNCH(k_axpby, g2, BLK2D, 0, st, psiF, psiF, in, 1.0 - alpha, alpha, nz, nr, h->ld);
    }
    download(h, h->psi_slots.p, 0, psi_out);
    API_END
}

// The device's division-by-reciprocal sequences on caller-supplied operands (the reciprocal words are formed here exactly as
// build_level_tables forms them); lets the tests compare them with IEEE division on hard-to-round cases.
int scpn_verify_division(const double* x, const double* d, int n, double* q5, double* q4) {
    if (!x || !d || !q5 || !q4 || n < 1) return fail(SCPN_E_ARG, "scpn_verify_division: null buffer or n < 1");
    API_BEGIN
    std::vector<double> r(n), rl(n);
    for (int i = 0; i < n; ++i) { r[i] = 1.0 / d[i]; rl[i] = std::fma(-d[i], r[i], 1.0) / d[i]; }
    DBuf<double> dx, dd_, dr_, dl, o5, o4;
    dx.alloc(n); dd_.alloc(n); dr_.alloc(n); dl.alloc(n); o5.alloc(n); o4.alloc(n);
    HIP_TRY(hipMemcpy(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dd_.p, d, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dr_.p, r.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dl.p, rl.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    SCPN_LAUNCH(k_verify_div, dim3(cdiv(n, 256)), dim3(256), 0, nullptr, (const double*)dx.p, (const double*)dd_.p, (const double*)dr_.p,
                       (const double*)dl.p, n, o5.p, o4.p);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(q5, o5.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(q4, o4.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    API_END
}

// device bandwidth of this GPU, HIP events around `reps` launches on `bytes` of HBM: mode 0 = copy (read + written bytes per
// second, best of three grid sizes), mode 1 = read-only sum
int scpn_copy_bandwidth_probe(long bytes, int reps, double* gb_per_s) { return scpn_bandwidth_probe(0, bytes, reps, gb_per_s); }
// mode 0: grid-stride copy (4 loads in flight); 1: read-only sum (8 in flight); 2: contiguous 32-KiB chunks per workgroup, 8 loads
// in flight; 3: the same with non-temporal stores; 4: hipMemcpyDtoDAsync; 5: three read streams + one write stream (the mix of the
// fused Picard tail).  Copies count read + written bytes.  1 GiB per
// buffer by default (4x the Infinity Cache); best over several grid sizes.
int scpn_bandwidth_probe(int mode, long bytes, int reps, double* gb_per_s) {
    if (!gb_per_s || bytes < (1L << 20) || reps < 1 || mode < 0 || mode > 5) return fail(SCPN_E_ARG, "scpn_bandwidth_probe: mode 0..5, bytes >= 1 MiB, reps >= 1");
    API_BEGIN
    HIP_TRY(hipSetDevice(g_device));
    const long n16 = bytes / 16;
    DBuf<double2> a, b, c2, c3;
    a.alloc((size_t)n16);
    if (mode != 1) b.alloc((size_t)n16); else b.alloc(16);
    HIP_TRY(hipMemset(a.p, 1, (size_t)n16 * 16));
    if (mode == 5) {  // three read streams + one write stream (four buffers of `bytes`)
        c2.alloc((size_t)n16); c3.alloc((size_t)n16);
        HIP_TRY(hipMemset(c2.p, 1, (size_t)n16 * 16));
        HIP_TRY(hipMemset(c3.p, 1, (size_t)n16 * 16));
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    double best = 0.0;
    for (int wg_per_cu : {4, 8, 16, 32}) {
        const dim3 g(256 * wg_per_cu);
        auto launch = [&] {
            switch (mode) {
                case 0: SCPN_LAUNCH(k_copy_probe, g, dim3(256), 0, nullptr, b.p, (const double2*)a.p, n16); break;
                case 1: SCPN_LAUNCH(k_read_probe, g, dim3(256), 0, nullptr, reinterpret_cast<double*>(b.p), (const double2*)a.p, n16); break;
                case 2: SCPN_LAUNCH(k_copy_probe_chunk<false>, g, dim3(256), 0, nullptr, b.p, (const double2*)a.p, n16); break;
                case 3: SCPN_LAUNCH(k_copy_probe_chunk<true>, g, dim3(256), 0, nullptr, b.p, (const double2*)a.p, n16); break;
                case 5: SCPN_LAUNCH(k_mix31_probe, g, dim3(256), 0, nullptr, b.p, (const double2*)a.p, (const double2*)c2.p, (const double2*)c3.p, n16); break;
                default: HIP_TRY(hipMemcpyDtoDAsync(b.p, a.p, (size_t)n16 * 16, nullptr)); break;
            }
        };
        launch();  // warm-up
        HIP_TRY(hipEventRecord(e0, nullptr));
        for (int r = 0; r < reps; ++r) launch();
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        best = std::max(best, (mode == 1 ? 1.0 : (mode == 5 ? 4.0 : 2.0)) * (double)n16 * 16.0 * reps / ((double)ms * 1e-3) / 1e9);
        if (mode == 4) break;  // no grid to vary
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *gb_per_s = best;
    API_END
}

// polyglot lane diagnostics (core/jax_gs_solver.py:196-316)
int scpn_np_delta_star(const double* psi, int nz, int nr, const double* r_axis, double dr, double dz, double* out) {
    if (!psi || !r_axis || !out || nr < 3 || nz < 3) return fail(SCPN_E_ARG, "psi shape must be at least 3x3 for centered GS diagnostics");
    API_BEGIN
    TempSolver t(nr, nz, 1, r_axis, nullptr, dr, dz);
    Solver* h = t.h;
    upload(h, h->psi_slots.p, 0, psi);
    DBuf<double> d;
    d.alloc((size_t)nz * nr);
    SCPN_LAUNCH(k_np_delta_star, grid2d(nr, nz, 1), BLK2D, 0, h->stream, d.p, h->slot_fld(nullptr), nz, nr, h->ld, dr, dz, (const double*)h->d_r.p);
    HIP_TRY(hipMemcpyAsync(out, d.p, sizeof(double) * nz * nr, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    API_END
}
int scpn_np_equation_residual(const double* psi, int nz, int nr, const double* r_axis, const double* z_axis, double ip_target, double mu0,
                              double beta_mix, double* out3) {
    if (!psi || !r_axis || !z_axis || !out3 || nr < 3 || nz < 3) return fail(SCPN_E_ARG, "Grad-Shafranov residual requires at least a 3 x 3 grid");
    API_BEGIN
    const double dr = r_axis[1] - r_axis[0], dz = z_axis[1] - z_axis[0];
    TempSolver t(nr, nz, 1, r_axis, z_axis, dr, dz);
    Solver* h = t.h;
    ensure_picard_bufs(h);
    upload(h, h->psi_slots.p, 0, psi);
    auto& lv = get_hierarchy(h, 1 << 30);
    Lvl L = lv[0]->dev;
    hipStream_t st = h->stream;
    Fld psiF = h->slot_fld(nullptr), srcF = h->plain(h->src.p);
    const int nb = h->nblk(), nbi = std::max(h->nblk_int(), 1);
    // _compute_source_np (jax_gs_solver.py:88-123): psi_axis = max interior, L-mode profile, sum J renormalised to Ip
    SCPN_LAUNCH(k_max_interior, dim3(nbi, 1), dim3(256), 0, st, psiF, nz, nr, h->ld, h->d_part1.p, nbi);
    SCPN_LAUNCH(k_np_final, dim3(1), dim3(256), 0, st, h->d_scal.p, (const double*)h->d_part1.p, nbi, 0, dr, dz, ip_target);
    SCPN_LAUNCH(k_np_jsum, dim3(nb, 1), dim3(256), 0, st, psiF, (const double*)h->d_scal.p, nz, nr, h->ld, (const double*)h->d_r.p, mu0, beta_mix,
                       h->d_part2.p, nb);
    SCPN_LAUNCH(k_np_final, dim3(1), dim3(256), 0, st, h->d_scal.p, (const double*)h->d_part2.p, nb, 1, dr, dz, ip_target);
    SCPN_LAUNCH(k_np_source, grid2d(nr, nz, 1), BLK2D, 0, st, srcF, psiF, (const double*)h->d_scal.p, nz, nr, h->ld, (const double*)h->d_r.p, mu0,
                       beta_mix);
    DBuf<double> part;
    part.alloc((size_t)4 * nbi);
    SCPN_LAUNCH(k_np_eq_residual, dim3(nbi, 1), dim3(256), 0, st, psiF, srcF, L, part.p, nbi);
    std::vector<double> hp((size_t)4 * nbi);
    HIP_TRY(hipMemcpyAsync(hp.data(), part.p, sizeof(double) * 4 * nbi, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    double rm = 0, r2 = 0, sm = 0, s2 = 0;
    for (int i = 0; i < nbi; ++i) { rm = std::fmax(rm, hp[i]); r2 += hp[nbi + i]; sm = std::fmax(sm, hp[2 * nbi + i]); s2 += hp[3 * nbi + i]; }
    out3[0] = rm;
    out3[1] = rm / (sm + 1.0e-30);
    out3[2] = std::sqrt(r2) / (std::sqrt(s2) + 1.0e-30);
    API_END
}

// ---- legacy libscpn_solver symbols (hpc/solver.cpp:200-335) ---------------------------------
void* create_solver(int nr, int nz, double rmin, double rmax, double zmin, double zmax) {
    if (nr < 2 || nz < 2) return nullptr;
    if (rmin >= rmax || zmin >= zmax) return nullptr;
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nullptr;
        HIP_TRY(hipSetDevice(g_device));
        auto* s = new FastSolverHip();
        s->nr = nr; s->nz = nz; s->ld = pitch_of(nr);
        s->rmin = rmin; s->rmax = rmax; s->zmin = zmin; s->zmax = zmax;
        s->dr = (rmax - rmin) / (nr - 1);  // solver.cpp:73-74
        s->dz = (zmax - zmin) / (nz - 1);
        HIP_TRY(hipStreamCreate(&s->stream));
        size_t n = (size_t)nz * s->ld;
        s->psi.alloc(n); s->j.alloc(n);
        HIP_TRY(hipMemset(s->psi.p, 0, sizeof(double) * n));
        HIP_TRY(hipMemset(s->j.p, 0, sizeof(double) * n));
        std::vector<double> R(nr), crp(nr), crm(nr);
        const double dr_sq = s->dr * s->dr;
        for (int r = 0; r < nr; ++r) {
            R[r] = rmin + r * s->dr;  // solver.cpp:85
            crp[r] = 1.0 / dr_sq - 1.0 / (2.0 * R[r] * s->dr);  // solver.cpp:170-171
            crm[r] = 1.0 / dr_sq + 1.0 / (2.0 * R[r] * s->dr);
        }
        s->Rtab.alloc(nr); s->c_rp.alloc(nr); s->c_rm.alloc(nr);
        HIP_TRY(hipMemcpy(s->Rtab.p, R.data(), sizeof(double) * nr, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->c_rp.p, crp.data(), sizeof(double) * nr, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->c_rm.p, crm.data(), sizeof(double) * nr, hipMemcpyHostToDevice));
        s->d_bits.alloc(1); s->d_perf.alloc(1); s->d_done.alloc(1); s->d_last.alloc(1);
        return s;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void set_boundary_dirichlet(void* solver, double v) {
    if (!solver) return;
    auto* s = static_cast<FastSolverHip*>(solver);
    s->boundary_value = v;
    try {
        SCPN_LAUNCH(k_fast_ring, dim3(cdiv(2L * (s->nz + s->nr), 256)), dim3(256), 0, s->stream, s->psi.p, s->nz, s->nr, s->ld, v,
                           (const int*)nullptr);
        HIP_TRY(hipStreamSynchronize(s->stream));
    } catch (const std::exception& e) { g_err = e.what(); }
}
void run_step(void* solver, const double* j, double* psi, int size, int iterations) {
    if (!solver || !j || !psi || size <= 0) return;
    auto* s = static_cast<FastSolverHip*>(solver);
    if ((long)size != (long)s->nr * s->nz) return;
    try {
        fast_sweeps(s, j, psi, std::max(iterations, 1), 1.8, false, 0.0, nullptr, nullptr);
    } catch (const std::exception& e) { g_err = e.what(); }
}
int run_step_converged(void* solver, const double* j, double* psi, int size, int max_iterations, double omega, double tolerance,
                       double* final_delta_out) {
    if (!solver || !j || !psi || size <= 0) { if (final_delta_out) *final_delta_out = 0.0; return 0; }
    auto* s = static_cast<FastSolverHip*>(solver);
    if ((long)size != (long)s->nr * s->nz) { if (final_delta_out) *final_delta_out = 0.0; return 0; }
    const int n_iter = std::max(max_iterations, 1);
    const double omega_safe = std::isfinite(omega) ? omega : 1.8;
    const double tol_safe = std::isfinite(tolerance) ? tolerance : 0.0;
    const double omega_c = std::min(std::max(omega_safe, 0.1), 1.99);
    const double tol = std::max(tol_safe, 0.0);
    int performed = 0;
    double last = 0.0;
    try {
        fast_sweeps(s, j, psi, n_iter, omega_c, true, tol, &performed, &last);
    } catch (const std::exception& e) { g_err = e.what(); return 0; }
    if (final_delta_out) *final_delta_out = last;
    return performed;
}
void destroy_solver(void* solver) {
    if (!solver) return;
    auto* s = static_cast<FastSolverHip*>(solver);
    (void)hipStreamSynchronize(s->stream);
    delete s;
}
void delete_solver(void* solver) { destroy_solver(solver); }

}  // extern "C"

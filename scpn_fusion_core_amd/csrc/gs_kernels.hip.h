// gs_kernels.hip.h -- CDNA4 (gfx950) device kernels of the Grad-Shafranov hot path.
//
// Conventions
//   * fp64 everywhere; compiled with -ffp-contract=off: every expression keeps the
//     operation order of the reference NumPy code so that the linear elliptic operators are
//     bit-identical to it (red-black SOR is order-independent inside one colour).
//   * a field is (nz, nr) with row pitch `ld` doubles (ld >= nr, rows 64-B aligned) and
//     `stride` doubles between ensemble members; blockIdx.z (or .y) is the member index.
//   * per-level 1-D coefficient tables a_e[ir], a_w[ir], r_safe[ir] are built on the host with
//     the reference's arithmetic (core/multigrid_solve.py:184-189) -- no R-grid traffic.
//   * wave = 64 lanes; reductions are fixed-shape trees (run-to-run deterministic).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

namespace gs {

constexpr double CAP = 1.0e250;  // core/fusion_kernel_numerics.py:16

struct Lvl {
    int nz, nr, ld;
    long stride;
    double a_ns, a_c, dr2, dz2, two_dr, dr, dz;
    double inv_a_c, inv_dr2, inv_dz2, inv_two_dr;  // correctly rounded reciprocals (host IEEE division) for divc()
    double lo_a_c, lo_dr2, lo_dz2, lo_two_dr;      // second words of the reciprocals (1/d - RN(1/d)) for divc4()
    const double* a_e;
    const double* a_w;
    const double* r_safe;
    const double* r_inv;     // RN(1/r_safe[ir])
    const double* r_inv_lo;  // 1/r_safe[ir] - r_inv[ir]
};

// x / d for a divisor whose correctly rounded reciprocal r = RN(1/d) is known.  q0 = RN(x*r) is within
// 1.5 ulp; one FMA correction makes it faithful (< 1 ulp), and Markstein's theorem (Handbook of Floating-Point
// Arithmetic, Newton-Raphson division with FMA) then makes the second correction the correctly rounded
// quotient -- bit-identical to IEEE division for all finite x away from under/overflow (the only difference
// is the sign of a zero result, and NaN instead of Inf for infinite x).  5 full-rate instructions instead
// of the ~14 of the IEEE sequence; verified against x/d on 2e9 random + near-midpoint inputs.
__device__ __forceinline__ double divc(double x, double d, double r) {
    double q = x * r;
    double e = __builtin_fma(-q, d, x);
    q = __builtin_fma(e, r, q);
    e = __builtin_fma(-q, d, x);
    return __builtin_fma(e, r, q);
}

// The same quotient in 4 instructions for the hot loops: with the reciprocal as a double-double (r + rl = 1/d to ~2^-106),
// q0 = RN(x*r + RN(x*rl)) is already within 0.5 ulp + 2^-105 of x/d, i.e. faithful, so ONE Markstein correction with the
// correctly rounded r gives the correctly rounded quotient (same theorem as the last step of divc).
__device__ __forceinline__ double divc4(double x, double d, double r, double rl) {
    double q = __builtin_fma(x, r, x * rl);
    double e = __builtin_fma(-q, d, x);
    return __builtin_fma(e, r, q);
}

// verification hook (tests/test_gpu_division.py): both sequences against IEEE division on arbitrary (x, d) pairs
__global__ void k_verify_div(const double* x, const double* d, const double* r, const double* rl, int n, double* q5, double* q4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q5[i] = divc(x[i], d[i], r[i]);
    q4[i] = divc4(x[i], d[i], r[i], rl[i]);
}

// Workgroups are dealt to the 8 XCDs round-robin in linear-id order (id i runs on XCD i % 8).  xcd_contiguous maps the
// hardware id to a logical id such that XCD k works through one contiguous run of logical ids (a bijection for any total):
// XCD k owns ceil((total - k) / 8) ids starting at k*(total/8) + min(k, total%8).
__device__ __forceinline__ unsigned xcd_contiguous(unsigned lin, unsigned total) {
    const unsigned k = lin & 7u, q = total >> 3, r = total & 7u;
    return k * q + (k < r ? k : r) + (lin >> 3);
}

// A batched field whose member b may live in one of several ping-pong slots.
// constants a branch-free load can point at when its real source is absent (see slot_ptr)
// (never written; not declared const so that they live in the global address space like the pointers they stand in for -- a
//  select between a global and a constant-address-space pointer would make the load a flat one)
__device__ int g_const_i32[2] = {0, 1};
__device__ double g_const_f64[2] = {0.0, 1.0};
struct Fld {
    double* base;
    const int* slot;  // nullptr -> slot 0
    long stride;
    int nb;  // members per slot
    __device__ __forceinline__ double* at(int b) const {
        int k = slot ? slot[b] : 0;
        return base + ((long)k * nb + b) * stride;
    }
    // the two halves of at(): where the slot index of member b is read from (a constant 0 if the field has one slot), so that a
    // kernel can put the slot words of all its fields in flight with ONE wait instead of a dependent load per field
    __device__ __forceinline__ const int* slot_ptr(int b) const { return slot ? slot + b : &g_const_i32[0]; }
    __device__ __forceinline__ double* at_slot(int b, int k) const { return base + ((long)k * nb + b) * stride; }
};

// per-member solver state kept on the device for the whole Picard solve
struct Member {
    double psi_axis, psi_bnd, denom, inv_denom;  // topology of the current iteration
    double x_r, x_z;
    long x_idx;
    double i_current, jscale;
    double i_target;
    double diff, diff_best, gs, gs_best;
    int active, converged, diverged, iterations;
    int cur, best, nxt, prev;  // ping-pong slots of psi
    int improved, pad;
};

__device__ __forceinline__ double sanitize1(double x) {  // fusion_kernel_numerics.py:19-25
    if (x != x) return 0.0;
    if (x > CAP) return CAP;
    if (x < -CAP) return -CAP;
    return x;
}
// the same as selects (no branches): for straight-line stencil code
__device__ __forceinline__ double sanitize_sel(double x) {
    const double c = fmin(fmax(x, -CAP), CAP);  // fmin / fmax return the non-NaN operand: a NaN input gives CAP here ...
    return x != x ? 0.0 : c;                    // ... and is replaced by 0 as sanitize1 does
}
__device__ __forceinline__ double nan_to_num300(double x) {  // fusion_kernel.py:273
    if (x != x) return 0.0;
    if (x > 1.79e308) return 1e300;
    if (x < -1.79e308) return -1e300;
    return x;
}
__device__ __forceinline__ bool finite_d(double x) { return fabs(x) <= 1.7976931348623157e308; }

template <typename T>
__device__ __forceinline__ T ld_val(const T* p) { return *p; }

// ---------------------------------------------------------------- block reductions (256 thr)
// Wave reductions on the DPP cross-lane network (row_shr within the 16-lane rows, then row_bcast:15 / row_bcast:31 across
// rows; the total lands in lane 63 and is broadcast through an SGPR).  A __shfl_down butterfly costs two ds_bpermute
// round trips per step for a double; the DPP moves are ordinary VALU instructions, which matters in the fused Picard tail
// where every workgroup reduces seven quantities.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double keep, double v) {  // lanes without a valid source (or masked off) get `keep`
    int kl = __double2loint(keep), kh = __double2hiint(keep), l = __double2loint(v), h = __double2hiint(v);
    l = __builtin_amdgcn_update_dpp(kl, l, CTRL, ROW_MASK, BANK_MASK, false);
    h = __builtin_amdgcn_update_dpp(kh, h, CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(h, l);
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ long dpp_i64(long keep, long v) {
    int kl = (int)(keep & 0xffffffffL), kh = (int)(keep >> 32), l = (int)(v & 0xffffffffL), h = (int)(v >> 32);
    l = __builtin_amdgcn_update_dpp(kl, l, CTRL, ROW_MASK, BANK_MASK, false);
    h = __builtin_amdgcn_update_dpp(kh, h, CTRL, ROW_MASK, BANK_MASK, false);
    return (long)(((unsigned long)(unsigned)h << 32) | (unsigned long)(unsigned)l);
}
__device__ __forceinline__ double lane63_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// DPP controls: 0x111/0x112/0x114/0x118 = row_shr:1/2/4/8, 0x142 = row_bcast:15, 0x143 = row_bcast:31
#define SCPN_DPP_REDUCE(X, OP, KEEP)                                       \
    X = OP(X, (dpp_f64<0x111, 0xf, 0xf>(KEEP(X), X)));                     \
    X = OP(X, (dpp_f64<0x112, 0xf, 0xf>(KEEP(X), X)));                     \
    X = OP(X, (dpp_f64<0x114, 0xf, 0xf>(KEEP(X), X)));                     \
    X = OP(X, (dpp_f64<0x118, 0xf, 0xf>(KEEP(X), X)));                     \
    X = OP(X, (dpp_f64<0x142, 0xa, 0xf>(KEEP(X), X)));                     \
    X = OP(X, (dpp_f64<0x143, 0xc, 0xf>(KEEP(X), X)));
#define SCPN_OP_ADD(a, b) ((a) + (b))
#define SCPN_KEEP_ZERO(x) 0.0
#define SCPN_KEEP_SELF(x) (x)
// total of the wave in every lane
__device__ __forceinline__ double wave_sum(double v) {
    SCPN_DPP_REDUCE(v, SCPN_OP_ADD, SCPN_KEEP_ZERO)
    return lane63_f64(v);
}
__device__ __forceinline__ double wave_max(double v) {
    SCPN_DPP_REDUCE(v, fmax, SCPN_KEEP_SELF)
    return lane63_f64(v);
}
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) r += sh[i];
    }
    __syncthreads();
    return r;  // valid in thread 0
}
template <int NT>
__device__ __forceinline__ double block_max(double v, double* sh) {
    v = wave_max(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        r = sh[0];
#pragma unroll
        for (int i = 1; i < NT / 64; ++i) r = fmax(r, sh[i]);
    }
    __syncthreads();
    return r;
}
// block max returned to every thread
template <int NT>
__device__ __forceinline__ double block_max_all(double v, double* sh) {
    v = wave_max(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = sh[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) r = fmax(r, sh[i]);
    return r;
}
// (value, index) reductions with numpy's first-occurrence tie rule
struct VI {
    double v;
    long i;
};
__device__ __forceinline__ VI vi_max(VI a, VI b) {
    bool tb = (b.i >= 0) && (a.i < 0 || b.v > a.v || (b.v == a.v && b.i < a.i));
    return tb ? b : a;
}
__device__ __forceinline__ VI vi_min(VI a, VI b) {
    bool tb = (b.i >= 0) && (a.i < 0 || b.v < a.v || (b.v == a.v && b.i < a.i));
    return tb ? b : a;
}
template <bool MAX, int CTRL, int ROW_MASK>
__device__ __forceinline__ VI vi_dpp_step(VI x) {
    VI y;
    y.v = dpp_f64<CTRL, ROW_MASK, 0xf>(x.v, x.v);  // invalid / masked lanes see themselves: the operations are idempotent
    y.i = dpp_i64<CTRL, ROW_MASK, 0xf>(x.i, x.i);
    return MAX ? vi_max(x, y) : vi_min(x, y);
}
template <bool MAX>
__device__ __forceinline__ VI wave_vi(VI x) {  // result in every lane
    x = vi_dpp_step<MAX, 0x111, 0xf>(x);
    x = vi_dpp_step<MAX, 0x112, 0xf>(x);
    x = vi_dpp_step<MAX, 0x114, 0xf>(x);
    x = vi_dpp_step<MAX, 0x118, 0xf>(x);
    x = vi_dpp_step<MAX, 0x142, 0xa>(x);
    x = vi_dpp_step<MAX, 0x143, 0xc>(x);
    VI r;
    r.v = lane63_f64(x.v);
    const int il = (int)(x.i & 0xffffffffL), ih = (int)(x.i >> 32);
    r.i = (long)(((unsigned long)(unsigned)__builtin_amdgcn_readlane(ih, 63) << 32) | (unsigned long)(unsigned)__builtin_amdgcn_readlane(il, 63));
    return r;
}
template <bool MAX, int NT>
__device__ __forceinline__ VI block_vi(VI x, VI* sh) {
    x = wave_vi<MAX>(x);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = x;
    __syncthreads();
    VI r = x;
    if (threadIdx.x == 0) {
        r = sh[0];
#pragma unroll
        for (int i = 1; i < NT / 64; ++i) r = MAX ? vi_max(r, sh[i]) : vi_min(r, sh[i]);
    }
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------- host-visible progress word
// The host drivers poll a device counter every few iterations / Krylov columns.  A hipMemcpyAsync + stream synchronise costs
// ~45 us of stream idle per poll (copy engine kernel, interrupt, relaunch latency: profiles/r03e_newton_timeline.txt); this
// one-thread kernel instead posts {sequence number, value} as ONE aligned 8-byte store into host-mapped pinned memory, and the
// host spins on that word (scpn_hip.hip: poll_word).
__global__ void k_post_word(const int* value, unsigned long long* host_word, unsigned seq) {
    const unsigned long long w = ((unsigned long long)seq << 32) | (unsigned long long)(unsigned)(*value);
    __hip_atomic_store(host_word, w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------- elementwise / copies
// read-only bandwidth probe: grid-stride sum, 16 B per lane, eight independent loads in flight per lane
__global__ void __launch_bounds__(256) k_read_probe(double* __restrict__ out, const double2* __restrict__ src, long n16) {
    const long step = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    for (; i + 7 * step < n16; i += 8 * step) {
        const double2 a = src[i], b = src[i + step], c = src[i + 2 * step], d = src[i + 3 * step];
        const double2 e = src[i + 4 * step], f = src[i + 5 * step], g = src[i + 6 * step], h = src[i + 7 * step];
        acc += (a.x + b.x) + (c.x + d.x) + (e.x + f.x) + (g.x + h.x) + (a.y + b.y) + (c.y + d.y) + (e.y + f.y) + (g.y + h.y);
    }
    for (; i < n16; i += step) acc += src[i].x + src[i].y;
    if (acc == 123.456) out[0] = acc;  // keeps the loads alive without a store per lane
}
// device-copy bandwidth probe (bench.py: the measured denominator reported next to the nominal 8 TB/s): grid-stride copy,
// 16 B per lane, four independent loads in flight per lane
__global__ void __launch_bounds__(256) k_copy_probe(double2* __restrict__ dst, const double2* __restrict__ src, long n16) {
    const long step = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n16; i += 4 * step) {
        const double2 a = src[i], b = src[i + step], c = src[i + 2 * step], d = src[i + 3 * step];
        dst[i] = a; dst[i + step] = b; dst[i + 2 * step] = c; dst[i + 3 * step] = d;
    }
    for (; i < n16; i += step) dst[i] = src[i];
}
// the same copy in the shape the hardware guide measures its 6.3 TB/s with: every workgroup walks CONTIGUOUS 32-KiB chunks
// (256 lanes x 8 x 16 B), eight independent 16-B loads in flight per lane, optionally non-temporal stores
template <bool NT>
__global__ void __launch_bounds__(256) k_copy_probe_chunk(double2* __restrict__ dst, const double2* __restrict__ src, long n16) {
    constexpr long CH = 256 * 8;  // 16-B elements per chunk
    const long nch = n16 / CH;
    for (long c = blockIdx.x; c < nch; c += gridDim.x) {
        const double2* s = src + c * CH + threadIdx.x;
        double2* d = dst + c * CH + threadIdx.x;
        double2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = s[256 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (NT) {
                __builtin_nontemporal_store(v[k].x, &d[256 * k].x);
                __builtin_nontemporal_store(v[k].y, &d[256 * k].y);
            } else {
                d[256 * k] = v[k];
            }
        }
    }
    for (long i = nch * CH + (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) dst[i] = src[i];
}
// three read streams + one write stream (the traffic mix of the fused Picard tail: cur, new, source in, next out), flat chunks
__global__ void __launch_bounds__(256) k_mix31_probe(double2* __restrict__ dst, const double2* __restrict__ a, const double2* __restrict__ b,
                                                     const double2* __restrict__ c, long n16) {
    constexpr long CH = 256 * 4;
    const long nch = n16 / CH;
    for (long ch = blockIdx.x; ch < nch; ch += gridDim.x) {
        const long o = ch * CH + threadIdx.x;
        double2 va[4], vb[4], vc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { va[k] = a[o + 256 * k]; vb[k] = b[o + 256 * k]; vc[k] = c[o + 256 * k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[o + 256 * k] = double2{va[k].x + vb[k].x + vc[k].x, va[k].y + vb[k].y + vc[k].y};
    }
}

// dst <- src over the logical (nz, nr) window; grid (ceil(nr/64), ceil(nz/4), B), block (64,4)
__global__ void k_copy(Fld dst, Fld src, int nz, int nr, int ld, const int* active) {
    int b = blockIdx.z;
    if (active && !active[b]) return;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    dst.at(b)[(long)iz * ld + ir] = src.at(b)[(long)iz * ld + ir];
}
__global__ void k_fill(Fld dst, double v, int nz, int nr, int ld, const int* active) {
    int b = blockIdx.z;
    if (active && !active[b]) return;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    dst.at(b)[(long)iz * ld + ir] = v;
}
// Dirichlet ring from bc (core/fusion_kernel_iterative_solver.py:316-329); grid (ceil(2*(nz+nr)/256), B)
__global__ void k_apply_bc(Fld psi, Fld bc, int nz, int nr, int ld, const int* active) {
    int b = blockIdx.y;
    if (active && !active[b]) return;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int iz, ir;
    if (t < nr) { iz = 0; ir = t; }
    else if (t < 2 * nr) { iz = nz - 1; ir = t - nr; }
    else if (t < 2 * nr + nz) { iz = t - 2 * nr; ir = 0; }
    else if (t < 2 * nr + 2 * nz) { iz = t - 2 * nr - nz; ir = nr - 1; }
    else return;
    psi.at(b)[(long)iz * ld + ir] = bc.at(b)[(long)iz * ld + ir];
}

// ---------------------------------------------------------------- red-black SOR (a1)
// core/multigrid_solve.py:193-206: one colour pass.  Threads are colour-compacted: thread t of
// row iz owns column ir = ir0(iz,parity) + 2t, so no lane idles on the other colour.
// grid (ceil(ncol/64), ceil((nz-2)/4), B), block (64,4)
template <bool CLIP>
__global__ void k_rb_half(Fld psiF, Fld srcF, Lvl L, double omega, double om1, int parity,
                          const int* active) {
    int b = blockIdx.z;
    if (active && !active[b]) return;
    int iz = 1 + blockIdx.y * 4 + threadIdx.y;
    int t = blockIdx.x * 64 + threadIdx.x;
    int ir = 1 + ((iz + 1 + parity) & 1) + 2 * t;
    if (iz >= L.nz - 1 || ir >= L.nr - 1) return;
    double* psi = psiF.at(b);
    const double* src = srcF.at(b);
    long c = (long)iz * L.ld + ir;
    double num = L.a_e[ir] * psi[c + 1] + L.a_w[ir] * psi[c - 1] + L.a_ns * psi[c - L.ld] + L.a_ns * psi[c + L.ld] - src[c];
    double gs = CLIP ? num / L.a_c : divc(num, L.a_c, L.inv_a_c);  // the clipped (_sor_step) form keeps IEEE Inf semantics
    double v = om1 * psi[c] + omega * gs;
    if (CLIP) v = fmin(fmax(v, -CAP), CAP);
    psi[c] = v;
}

// Whole small level in LDS: n_sweeps full red-black sweeps by ONE workgroup per member
// (levels up to SMALL_MAX points: the multigrid base case and the coarse levels).
constexpr int SMALL_MAX = 1156;  // 34*34
__global__ void __launch_bounds__(256) k_small_smooth(Fld psiF, Fld srcF, Lvl L, double omega,
                                                      double om1, int n_sweeps, const int* active) {
    __shared__ double sp[SMALL_MAX];
    __shared__ double ss[SMALL_MAX];
    int b = blockIdx.x;
    if (active && !active[b]) return;
    double* psi = psiF.at(b);
    const double* src = srcF.at(b);
    const int nz = L.nz, nr = L.nr, n = nz * nr;
    for (int i = threadIdx.x; i < n; i += 256) {
        int iz = i / nr, ir = i - iz * nr;
        sp[i] = psi[(long)iz * L.ld + ir];
        ss[i] = src[(long)iz * L.ld + ir];
    }
    __syncthreads();
    const int ni = nr - 2, nint = (nz - 2) * ni;
    for (int s = 0; s < 2 * n_sweeps; ++s) {
        int parity = s & 1;
        for (int k = threadIdx.x; k < nint; k += 256) {
            int iz = 1 + k / ni, ir = 1 + (k - (iz - 1) * ni);
            if (((iz + ir) & 1) != parity) continue;
            int c = iz * nr + ir;
            double gs = divc(L.a_e[ir] * sp[c + 1] + L.a_w[ir] * sp[c - 1] + L.a_ns * sp[c - nr] + L.a_ns * sp[c + nr] - ss[c],
                             L.a_c, L.inv_a_c);
            sp[c] = om1 * sp[c] + omega * gs;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += 256) {
        int iz = i / nr, ir = i - iz * nr;
        psi[(long)iz * L.ld + ir] = sp[i];
    }
}

// ---------------------------------------------------------------- residual (a2, a3)
__device__ __forceinline__ double resid_at(const double* psi, const double* src, const Lvl& L,
                                           int iz, int ir) {
    // core/multigrid_solve.py:243-248
    long c = (long)iz * L.ld + ir;
    double p = psi[c], e = psi[c + 1], w = psi[c - 1], n = psi[c + L.ld], s = psi[c - L.ld];
    double d2r = divc(e - 2.0 * p + w, L.dr2, L.inv_dr2);
    double d1r = divc(e - w, L.two_dr, L.inv_two_dr);
    double d2z = divc(n - 2.0 * p + s, L.dz2, L.inv_dz2);
    double lpsi = d2r - divc(d1r, L.r_safe[ir], L.r_inv[ir]) + d2z;
    return lpsi - src[c];
}
// out = L*psi - src (zero ring); grid (ceil(nr/64), ceil(nz/4), B)
__global__ void k_residual(Fld outF, Fld psiF, Fld srcF, Lvl L) {
    int b = blockIdx.z;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= L.nr || iz >= L.nz) return;
    double v = 0.0;
    if (iz > 0 && iz < L.nz - 1 && ir > 0 && ir < L.nr - 1) v = resid_at(psiF.at(b), srcF.at(b), L, iz, ir);
    outF.at(b)[(long)iz * L.ld + ir] = v;
}
// partial max|r| and overflow-safe partial sum over the interior: grid (nblk, B), 256 threads x 4 points.
// stable_rms (core/fusion_kernel_numerics.py:28-37) is max * sqrt(mean((r/max)^2)); each block stores its own max m_b and
// s_b = sum (r/m_b)^2, the consumer combines M = max m_b, sum s_b (m_b/M)^2 -- no r^2 is ever formed (|r| up to 1e250).
__global__ void __launch_bounds__(256) k_resid_norms(Fld psiF, Fld srcF, Lvl L, double* pmax,
                                                     double* psum, int nblk, const int* active) {
    __shared__ double sh[4];
    int b = blockIdx.y;
    if (active && !active[b]) return;
    const double* psi = psiF.at(b);
    const double* src = srcF.at(b);
    const int ni = L.nr - 2;
    const long nint = (long)(L.nz - 2) * ni;
    double m = 0.0, rv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        rv[k] = 0.0;
        if (q < nint) {
            int iz = 1 + (int)(q / ni), ir = 1 + (int)(q - (long)(iz - 1) * ni);
            rv[k] = sanitize1(resid_at(psi, src, L, iz, ir));
            m = fmax(m, fabs(rv[k]));
        }
    }
    const double bm = block_max_all<256>(m, sh);
    double s = 0.0;
    if (bm > 0.0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double t = rv[k] / bm; s += t * t; }
    }
    __syncthreads();
    double bs = block_sum<256>(s, sh);
    if (threadIdx.x == 0) {
        pmax[(long)b * nblk + blockIdx.x] = bm;
        psum[(long)b * nblk + blockIdx.x] = bs;
    }
}

// ---------------------------------------------------------------- restriction (a4)
__device__ __forceinline__ double fw9(double c, double up, double dn, double lf, double rt,
                                      double ul, double ur, double dl, double dr_) {
    // core/multigrid_solve.py:76-91
    return (4.0 * c + 2.0 * (up + dn + lf + rt) + (ul + ur + dl + dr_)) / 16.0;
}
// plain full weighting of a field; grid (ceil(nr_c/64), ceil(nz_c/4), B)
__global__ void k_restrict(Fld cF, Fld fF, int nz_f, int nr_f, int ld_f, int nz_c, int nr_c,
                           int ld_c) {
    int b = blockIdx.z;
    int jc = blockIdx.x * 64 + threadIdx.x, ic = blockIdx.y * 4 + threadIdx.y;
    if (jc >= nr_c || ic >= nz_c) return;
    const double* f = fF.at(b);
    double v;
    // injection on the ring; columns override rows at the corners (multigrid_solve.py:94-97)
    if (jc == 0) v = f[(long)(2 * ic) * ld_f];
    else if (jc == nr_c - 1) v = f[(long)(2 * ic) * ld_f + nr_f - 1];
    else if (ic == 0) v = f[2 * jc];
    else if (ic == nz_c - 1) v = f[(long)(nz_f - 1) * ld_f + 2 * jc];
    else {
        const double* p = f + (long)(2 * ic) * ld_f + 2 * jc;
        v = fw9(p[0], p[-ld_f], p[ld_f], p[-1], p[1], p[-ld_f - 1], p[-ld_f + 1], p[ld_f - 1], p[ld_f + 1]);
    }
    cF.at(b)[(long)ic * ld_c + jc] = v;
}
// fused: coarse rhs = restrict_full_weight(-(L*psi - src))  (multigrid_solve.py:303-306) and
// coarse e = 0 (multigrid_solve.py:315).  Each coarse thread evaluates its 9 fine residuals.
__global__ void k_defect_restrict(Fld rhsC, Fld eC, Fld psiF, Fld srcF, Lvl Lf, Lvl Lc,
                                  const int* active) {
    int b = blockIdx.z;
    if (active && !active[b]) return;
    int jc = blockIdx.x * 64 + threadIdx.x, ic = blockIdx.y * 4 + threadIdx.y;
    if (jc >= Lc.nr || ic >= Lc.nz) return;
    double v = -0.0;  // injected ring of the negated (zero-ring) residual
    if (jc > 0 && jc < Lc.nr - 1 && ic > 0 && ic < Lc.nz - 1) {
        const double* psi = psiF.at(b);
        const double* src = srcF.at(b);
        int i = 2 * ic, j = 2 * jc;
        double c = -resid_at(psi, src, Lf, i, j);
        double up = -resid_at(psi, src, Lf, i - 1, j), dn = -resid_at(psi, src, Lf, i + 1, j);
        double lf = -resid_at(psi, src, Lf, i, j - 1), rt = -resid_at(psi, src, Lf, i, j + 1);
        double ul = -resid_at(psi, src, Lf, i - 1, j - 1), ur = -resid_at(psi, src, Lf, i - 1, j + 1);
        double dl = -resid_at(psi, src, Lf, i + 1, j - 1), dr_ = -resid_at(psi, src, Lf, i + 1, j + 1);
        v = fw9(c, up, dn, lf, rt, ul, ur, dl, dr_);
    }
    rhsC.at(b)[(long)ic * Lc.ld + jc] = v;
    eC.at(b)[(long)ic * Lc.ld + jc] = 0.0;
}

// ---------------------------------------------------------------- prolongation (a5)
__device__ __forceinline__ double prolong_at(const double* c, int ld_c, int nz_c, int nr_c,
                                             int nz_f, int nr_f, int i, int j) {
    // core/multigrid_solve.py:117-143 with nz_c == (nz_f+1)/2, nr_c == (nr_f+1)/2
    int h_end = min(2 * (nr_c - 1), nr_f - 1), v_end = min(2 * (nz_c - 1), nz_f - 1);
    bool ie = !(i & 1), je = !(j & 1);
    int ic = i >> 1, jc = j >> 1;
    if (ie && je) return c[(long)ic * ld_c + jc];
    if (ie) {  // even row, odd col
        if (j >= h_end) return 0.0;
        return 0.5 * (c[(long)ic * ld_c + jc] + c[(long)ic * ld_c + jc + 1]);
    }
    if (i >= v_end) return 0.0;
    if (je) return 0.5 * (c[(long)ic * ld_c + jc] + c[(long)(ic + 1) * ld_c + jc]);
    if (j >= h_end) return 0.0;
    return 0.25 * (c[(long)ic * ld_c + jc] + c[(long)(ic + 1) * ld_c + jc] + c[(long)ic * ld_c + jc + 1] +
                   c[(long)(ic + 1) * ld_c + jc + 1]);
}
// ---------------------------------------------------------------- tile smoother: S sweeps per launch, the tile in registers
// S full red-black sweeps per launch with temporal blocking.  A workgroup owns an LZ x 128 tile of psi (halo H = 2S, + 2 for
// DEFECT); each colour pass invalidates one more cell from every interior tile edge; after 2S passes the centre
// (LZ - 2H) x (128 - 2H) window is exact and is written to `out` (out-of-place: neighbouring tiles read the halo from `in`).
// Results are bit-identical to the one-pass-per-launch kernel because a colour pass only reads the other colour.
// Inactive members are passed through (out = in).  grid (tiles_r, tiles_z, B), dynamic LDS = LZ KiB.
struct Prof {
    int hmode;
    double ped_p[4], ped_ff[4];
};
__device__ __forceinline__ double mtanh1(double x, const double* p) {  // core/fusion_kernel.py:376-390
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
// HM: 1 / 0 = the profile mode as a compile-time constant (the L-mode form carries no pedestal code and half the registers), -1 = read pf.hmode
template <int HM = -1>
__device__ __forceinline__ double j_raw_at(double psi, double R, double inv_mu0R, double psi_axis,
                                           double denom, double inv_denom, const Prof& pf) {
    // core/fusion_kernel.py:419-434
    double pn = divc(psi - psi_axis, denom, inv_denom);
    double p = 0.0, ff = 0.0;
    if (pn >= 0.0 && pn < 1.0) {
        if (HM < 0 ? pf.hmode != 0 : HM != 0) { p = mtanh1(pn, pf.ped_p); ff = mtanh1(pn, pf.ped_ff); }
        else { p = 1.0 - pn; ff = p; }
    }
    double j_p = R * p;
    double j_f = inv_mu0R * ff;
    return 0.5 * j_p + 0.5 * j_f;
}
// lazy-source mode: the source field holds J_raw of k_jsum and Source = (-mu0 R) * (J_raw * scale) is formed on load with the
// expression of k_source, so the separate source pass over the fine grid disappears (st == nullptr: the field is the source).
struct LazySrc {
    const Member* st = nullptr;
    const double* neg_mu0R = nullptr;
    // part != nullptr: this launch is the FIRST consumer of J_raw after k_jsum and finishes the sum itself -- every wave adds the
    // nblk (<= 256) per-workgroup partial sums of its member in the order of jsum_final_body and forms the scale, tile (0, 0) of
    // the member also stores I and the scale for the later consumers: the k_jsum_final launch and its kernel boundary leave the
    // dependent chain of an iteration
    const double* part = nullptr;
    int nblk = 0;
    double dr = 0, dz = 0;
    Member* st_rw = nullptr;
    // Source ON THE FLY (contracted arithmetic, L-mode): the "source" field IS the current iterate psi -- J_raw is not stored at
    // all.  Every consumer forms J_raw = (0.5 R + 0.5 / (mu0 R)) * shape(psi_N) from psi on load (jraw_shape below: the
    // expression of j_raw_at with the division by the normalisation span as a reciprocal multiply), so k_jsum becomes a
    // read-only pass, the pre-smoother (whose input is that same psi) loads one field instead of two and the Picard tail,
    // which reads the current iterate anyway, none for the source: 76 instead of 100 bytes per point and iteration on level 0.
    int jfly = 0;
    const double* r_axis = nullptr;
    const double* inv_mu0R = nullptr;
};
// shape(psi_N) of the L-mode profile (core/fusion_kernel.py:419-429: p = ff = 1 - psi_N inside 0 <= psi_N < 1, else 0)
__device__ __forceinline__ double jraw_shape(double psi, double psi_axis, double inv_denom) {
    const double pn = (psi - psi_axis) * inv_denom;
    return (pn >= 0.0 && pn < 1.0) ? 1.0 - pn : 0.0;
}
// the scale of member b from the partial sums (core/fusion_kernel.py:436-438, the arithmetic of jsum_final_body: wave totals of
// 64-value chunks added in order to +0.0); x0 = this lane's value of the first chunk, already loaded
__device__ __forceinline__ double lazy_scale_from_parts(const LazySrc& z, int b, int lane, double x0, double itg, bool store) {
    double tot = 0.0;
    tot += wave_sum(x0);
    for (int k = 1; k * 64 < z.nblk; ++k) {  // (uniform; grids above ~1000^2 only)
        const int i = k * 64 + lane;
        tot += wave_sum(i < z.nblk ? z.part[(long)b * z.nblk + i] : 0.0);
    }
    const double ic = (tot * z.dr) * z.dz;
    const double sc = fabs(ic) > 1e-9 ? itg / ic : 0.0;
    if (store) { z.st_rw[b].i_current = ic; z.st_rw[b].jscale = sc; }
    return sc;
}
#ifdef CV_STAMPS
__device__ unsigned long long g_cv_stamps[64];
__device__ unsigned long long g_tile_stamps[8 * 16];  // 8 sampled workgroups (dispatch-order ids 1, 511, 1021, ...) x 16 stamps
#define CV_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_cv_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#ifndef STAMP_DEFECT
#define STAMP_DEFECT false
#endif
#ifndef STAMP_PROLONG
#define STAMP_PROLONG false
#endif
#ifndef STAMP_NZ_MIN
#define STAMP_NZ_MIN 300  // which level is stamped: rows of the level in [STAMP_NZ_MIN, STAMP_NZ_MAX]
#endif
#ifndef STAMP_NZ_MAX
#define STAMP_NZ_MAX 100000
#endif
#ifndef STAMP_STRIDE
#define STAMP_STRIDE 510  // every STAMP_STRIDE-th workgroup (dispatch order) is sampled, eight in all
#endif
#define TILE_STAMP(i) do { if (DEFECT != STAMP_DEFECT || PROLONG != STAMP_PROLONG || L.nz < STAMP_NZ_MIN || L.nz > STAMP_NZ_MAX) break; const unsigned lin__ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
        if (threadIdx.x == 0 && lin__ % STAMP_STRIDE == 1 && lin__ / STAMP_STRIDE < 8) g_tile_stamps[(lin__ / STAMP_STRIDE) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define BLEND_STAMP(i) do { const unsigned lin__ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
        if (threadIdx.x == 0 && gridDim.y > 20 && lin__ % 577 == 1 && lin__ / 577 < 8) g_tile_stamps[(lin__ / 577) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CV_STAMP(i)
#define TILE_STAMP(i)
#define BLEND_STAMP(i)
#endif
// Tile geometries (rows x waves; rows / waves = an even number of consecutive rows per wave), chosen per launch by the size of
// the ensemble (scpn_hip.hip, launch_tile): BIG for ensembles that fill the chip (six rows per wave, the sources of a wave's
// inner rows in LDS slots), MID for ensembles of up to 12 members (four rows per wave), TINY while a level-0 launch has at most ~220 tiles
// (two rows per wave: the shortest workgroup lifetime).  -D overrides build other geometries for experiments.
#ifndef TILE_BIG_LZ
#define TILE_BIG_LZ 48
#endif
#ifndef TILE_BIG_NW
#define TILE_BIG_NW 8
#endif
#ifndef TILE_MID_LZ
#define TILE_MID_LZ 32
#endif
#ifndef TILE_MID_NW
#define TILE_MID_NW 8
#endif
#ifndef TILE_TINY_LZ
#define TILE_TINY_LZ 24
#endif
#ifndef TILE_TINY_NW
#define TILE_TINY_NW 12
#endif
#ifndef TILE_WPE
#define TILE_WPE 6  // minimum waves per SIMD the register allocation must allow: <= 80 VGPRs (three 8-wave workgroups per CU)
#endif
// rows per wave from which on a wave whose rows drop out of the passes skips them (see half_sweep)
#ifndef TILE_SKIP_MIN_ROWS
#define TILE_SKIP_MIN_ROWS 4
#endif
// PROLONG with fewer than six rows per wave: fine rows whose psi / source loads are in flight together behind the coarse rows
// (register budget; with six rows per wave the inner-row sources travel by LDS-DMA and all rows are in flight at once).
// (Round 4's A/B switches of the load phase -- TILE_PWIDE, TILE_LG, TILE_GLDS, TILE_SHARE_D -- are retired: the forms they
// switched back to are archived as diffs under profiles/r04_negative/ and profiles/r05_negative/retired_tile_switches.md.)
#ifndef TILE_LG_WIDE
#define TILE_LG_WIDE 2
#endif
// PROLONG: the tile is loaded as in + P(coarse) (bilinear prolongation of the coarse-grid correction,
// multigrid_solve.py:329-330) -- the separate prolong-add pass over the fine grid disappears.
// DEFECT: after the sweeps the tile also produces the coarse-grid right-hand side restrict_full_weight(-(L*psi - src))
// (core/multigrid_solve.py:303-306) and zeroes the coarse correction: the halo grows by 2 (defect ring + restriction
// stencil) and the separate defect/restrict pass over the fine grid disappears.  (coarseF = coarse rhs, dd.e = coarse e.)
struct TopoPart {
    double maxv;
    long maxi;
    double minb;
    long mini;
    double minpsi;
    double psi_at_min;  // psi at `mini` (the X-point candidate): the final reduction needs no read of the field
};
struct DefectDst {
    Fld e{nullptr, nullptr, 0, 0};
    double dr2 = 1, dz2 = 1, two_dr = 1, inv_dr2 = 1, inv_dz2 = 1, inv_two_dr = 1, lo_dr2 = 0, lo_dz2 = 0, lo_two_dr = 0;
    const double* r_safe = nullptr;
    const double* r_inv = nullptr;
    const double* r_inv_lo = nullptr;
    double kd = 1;  // contracted arithmetic: a_c / omega (the defect is kd * (omega * psi - [omega-scaled Gauss-Seidel value]))
};
// Where the tile lives: a wave owns NROWS = LZ / NW CONSECUTIVE rows (NROWS even, so the parity of a row is a compile-time
// property of its index i), and a lane the column pair (2l, 2l+1) of those rows: psi and source of the pair stay in registers
// from the global load to the store.  Of the four neighbours of an update, one is the lane's other column, one belongs to the
// neighbouring LANE (a whole 128-column row sits in one wave: a wave_shr / wave_shl DPP move, two per fp64 value) and two are
// the rows above / below, which are registers too except for the first / last row of the wave.  LDS carries only those boundary
// rows between waves (two 512-B writes and two reads per wave and colour pass), the inner-row sources of the six-row geometry
// and, for DEFECT, the defect planes the restriction stencil is taken from.  (Rounds 1-3 kept the tile in two LDS colour planes,
// five reads and a write per ROW update: profiles/r03_negative/lds_plane_tile_smoother.hip.h.txt, same bits, slower.)
//
// Out-of-domain cells.  A tile that overhangs the grid loads CLAMPED copies of in-domain values into its out-of-domain cells (row
// index clamped on the scalar unit, the lane's column pair clamped once) and never masks them: a red-black update is only KEPT at
// interior points (1 <= gz <= nz-2, 1 <= gc <= nr-2), whose four neighbours are all in the domain, the defect is kept under the
// same condition (-0.0 elsewhere) and stores are confined to the in-domain part of the window -- so no out-of-domain value ever
// reaches an output.  That keeps the load phase free of selects: per row two 16-B loads from (scalar member base) + (32-bit byte
// offset: row part on the scalar unit, one vector add for the lane) and the lazy-source products.
template <int CTRL>
__device__ __forceinline__ double wave_shift(double x) {  // 0x138: wave_shr:1 = value of lane-1, 0x130: wave_shl:1 = value of lane+1
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);  // (bound_ctrl: lane 0 / 63 receive 0, no prior value needed)
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// FAST (contracted arithmetic, Solver::arith == 1; every other instantiation keeps the reference's operation order bit for bit):
// the update (a_E E + a_W W + a_NS (S + N) - src) / a_C relaxed by omega becomes ONE chain of four (omega == 1) or five fused
// multiply-adds -- the column coefficients and the source are scaled by omega / a_C once per tile (the lazy-source factors
// (-mu0 R) * scale fold into the same product), no division sequence, no separate products -- and the defect
// src - L*psi = a_C (psi - gs) is read off the same chain (six operations instead of 27).  With omega == 1 the column predicate
// folds into the operands too: a column that is never updated carries zero coefficients and -psi as its scaled source, so
// the chain returns psi itself (exactly: 0 * x + psi) and the sweeps carry no select at all; the row test is a wave-uniform branch.
// Results differ from the exact instantiations by rounding only (tests/test_gpu_fast_arith.py: <= 1e-10 max|psi| on converged equilibria).
template <int S, bool PROLONG, bool DEFECT, int LZT, int NWT, int WPEU, bool SUMF = false, bool GS1T = false, bool FAST = false>
__global__ void __launch_bounds__(64 * NWT, WPEU) k_smooth_tile(Fld outF, Fld inF, Fld srcF, Lvl L, double omega, double om1,
                                                    const int* active, Fld coarseF, int nz_c, int nr_c, int ld_c, LazySrc lzs,
                                                    DefectDst dd, int xcd_remap) {
    constexpr int LZ = LZT, H = 2 * S + (DEFECT ? 2 : 0), NW = NWT, NROWS = LZ / NW;
    static_assert(LZ % NW == 0 && NROWS % 2 == 0 && NROWS >= 2, "a wave owns an even number of consecutive rows");
    // With six or more rows per wave the sources of the wave's INNER rows (1 .. NROWS-2) live in the lane's own slots of the colour
    // planes (written once, read back by the same lane: no conflict, no barrier) instead of in registers; the first / last row's
    // slots carry psi to the neighbouring waves, so their sources stay in registers.
    constexpr bool SRCL = NROWS >= 6;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* A0 = lds;            // colour 0: (lz + column) even; cell (lz, col) at [lz * 64 + col / 2]
    double* A1 = lds + LZ * 64;  // colour 1
    int bx_, by_, b;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
        const unsigned total = gx * gy * gz;
        unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        if (xcd_remap & 1) lin = xcd_contiguous(lin, total);
        bx_ = (int)(lin % gx);
        const unsigned t = lin / gx;
        by_ = (int)(t % gy);
        b = (int)(t / gy);
    }
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lzb = w * NROWS;  // first tile row of this wave (even)
    const int z0 = by_ * (LZ - 2 * H) - H, r0 = bx_ * (128 - 2 * H) - H;  // both even
    const int nz = L.nz, nr = L.nr, ld = L.ld;
    const int c0 = r0 + 2 * lane;
    // Everything that depends only on (member, column) is requested here, unconditionally and at once -- the member's activity
    // word, the slot words of the fields, the lazy-source scale, the column coefficients -- so that the row loads wait for ONE
    // memory round trip, not for a chain of them (absent sources point at a constant).
    const bool lazy = lzs.st != nullptr;  // uniform
    // bit 1 of xcd_remap: `in` is the zero field (the correction of a level >= 1 before its pre-smoother: the level above leaves
    // it unwritten, see DefectDst::e) -- the tile starts from +0.0 without loading it
    const bool in_zero = (xcd_remap & 2) != 0;
    const int act_w = *(active ? active + b : &g_const_i32[1]);
    const int k_in = *inF.slot_ptr(b), k_out = *outF.slot_ptr(b), k_src = *srcF.slot_ptr(b);
    const int k_cc = (PROLONG || DEFECT) ? *coarseF.slot_ptr(b) : 0;  // PROLONG: the coarse correction read; DEFECT: the coarse rhs written
    const int k_ce = DEFECT ? *dd.e.slot_ptr(b) : 0;
    const bool in0 = c0 >= 1 && c0 <= nr - 2, in1 = c0 + 1 >= 1 && c0 + 1 <= nr - 2;
    constexpr bool lz_sum = SUMF;  // this instantiation finishes the J_raw sum (see LazySrc; lzs.part is set)
    double lz_sc = *((lazy && !lz_sum) ? &lzs.st[b].jscale : &g_const_f64[1]);
    const double lz_itg = *(lz_sum ? &lzs.st[b].i_target : &g_const_f64[1]);
    const double lz_x0 = *((lz_sum && lane < lzs.nblk) ? lzs.part + ((long)b * lzs.nblk + lane) : &g_const_f64[0]);
    double lz_n0 = *(lazy ? lzs.neg_mu0R + min(max(c0, 0), nr - 1) : &g_const_f64[1]);
    double lz_n1 = *(lazy ? lzs.neg_mu0R + min(max(c0 + 1, 0), nr - 1) : &g_const_f64[1]);
    // source on the fly (FAST instantiations only; uniform): the member's normalisation and the column factor 0.5 R + 0.5 / (mu0 R)
    const bool jfly = FAST && lazy && lzs.jfly != 0;
    const double jf_pa = *(jfly ? &lzs.st[b].psi_axis : &g_const_f64[0]), jf_id = *(jfly ? &lzs.st[b].inv_denom : &g_const_f64[1]);
    if (jfly) {
        const int ca = min(max(c0, 0), nr - 1), cb = min(max(c0 + 1, 0), nr - 1);
        lz_n0 = lz_n0 * (0.5 * lzs.r_axis[ca] + 0.5 * lzs.inv_mu0R[ca]);
        lz_n1 = lz_n1 * (0.5 * lzs.r_axis[cb] + 0.5 * lzs.inv_mu0R[cb]);
    }
    // (PROLONG, wide load: the stencil coefficients are first used by the sweeps -- they are requested behind the LAST group of row
    //  loads, so that they hold no registers while the earlier groups are in flight)
    constexpr bool COEF_LATE = PROLONG;
    const double* const pe0 = in0 ? L.a_e + c0 : &g_const_f64[0];
    const double* const pw0 = in0 ? L.a_w + c0 : &g_const_f64[0];
    const double* const pe1 = in1 ? L.a_e + c0 + 1 : &g_const_f64[0];
    const double* const pw1 = in1 ? L.a_w + c0 + 1 : &g_const_f64[0];
    double ae0 = 0.0, aw0 = 0.0, ae1 = 0.0, aw1 = 0.0;  // even / odd column of the pair
    if (!COEF_LATE) { ae0 = *pe0; aw0 = *pw0; ae1 = *pe1; aw1 = *pw1; }
    // DEFECT: r_safe / its reciprocal (two words) of the lane's even / odd column, requested with the rest (the defect stage
    // then starts without a memory round trip)
    double rs0 = 1.0, rs1 = 1.0, ri0 = 1.0, ri1 = 1.0, rl0 = 0.0, rl1 = 0.0;
    constexpr bool FOLD = FAST && GS1T;  // (see the template header: column predicate folded into the operands)
    constexpr bool EARLY_D = DEFECT && !FAST && NROWS < 6;  // (six rows per wave: no registers to spare through the sweeps)
    // Six rows per wave: the six values are the same for all waves of the tile (same columns), so wave k requests the k-th of them
    // here -- one double in flight per lane -- parks it in LDS behind the tile's planes, and the defect stage reads all six from
    // there (an LDS access instead of a memory round trip in the middle of the tile's life).
    constexpr bool SHARE_D = DEFECT && !FAST && !EARLY_D && NW >= 6;
    double* const CO = lds + LZ * 128;  // [6][64] doubles (allocated by the launcher for SHARE_D instantiations)
    double co_x = 0.0;
    if (SHARE_D) {
        const double* const arr = (w % 3) == 0 ? dd.r_safe : ((w % 3) == 1 ? dd.r_inv : dd.r_inv_lo);  // (w < 6 used; scalar selects)
        const bool odd = w >= 3, okc = odd ? in1 : in0;
        co_x = *((okc && w < 6) ? arr + c0 + (odd ? 1 : 0) : &g_const_f64[(w % 3) == 2 ? 0 : 1]);
    }
    if (EARLY_D) {
        rs0 = *(in0 ? dd.r_safe + c0 : &g_const_f64[1]); ri0 = *(in0 ? dd.r_inv + c0 : &g_const_f64[1]); rl0 = *(in0 ? dd.r_inv_lo + c0 : &g_const_f64[0]);
        rs1 = *(in1 ? dd.r_safe + c0 + 1 : &g_const_f64[1]); ri1 = *(in1 ? dd.r_inv + c0 + 1 : &g_const_f64[1]); rl1 = *(in1 ? dd.r_inv_lo + c0 + 1 : &g_const_f64[0]);
    }
    const double* in = inF.at_slot(b, k_in);
    double* out = outF.at_slot(b, k_out);
    const bool pair_own = lane >= H / 2 && lane < 64 - H / 2 && c0 < nr;
    const bool pair_full = c0 + 1 < nr;
    const unsigned ld8 = (unsigned)ld * 8u;
    if (!act_w) {  // frozen member: out = in on the window
#pragma unroll
        for (int i = 0; i < NROWS; ++i) {
            const int lz = lzb + i, gz = z0 + lz;
            if (lz >= H && lz < LZ - H && gz < nz && pair_own) {
                const long o = (long)gz * ld + c0;
                if (pair_full) *reinterpret_cast<double2*>(out + o) = *reinterpret_cast<const double2*>(in + o);
                else out[o] = in[o];
            }
        }
        return;
    }
    TILE_STAMP(0);
    const double* src = srcF.at_slot(b, k_src);
    const double a_ns = L.a_ns, a_c = L.a_c, inv_a_c = L.inv_a_c, lo_a_c = L.lo_a_c;
    // column predicate of the even / odd column: inside the tile's updatable columns and the grid's interior
    const bool colok0 = lane >= 1 && c0 >= 1 && c0 <= nr - 2;
    const bool colok1 = lane <= 62 && c0 + 1 >= 1 && c0 + 1 <= nr - 2;
    // v[c][i]: the colour-c element of row i of the pair, s[c][i] its source.  Row i (tile row lzb + i, parity i & 1) carries
    // colour c in the column of parity (c + i) & 1.
    double v0[NROWS], v1[NROWS], s0[NROWS], s1[NROWS];
    double* const b0 = A0 + lzb * 64 + lane;  // the lane's slot of the wave's first row in the colour-0 / colour-1 plane
    double* const b1 = A1 + lzb * 64 + lane;
    // FAST: the factor every source value is multiplied by on load (omega / a_C, times the lazy-source factors once the scale is known)
    const double fsc = FAST ? omega * inv_a_c : 1.0;
    double sf0 = fsc, sf1 = fsc;
    // ---- load phase (out-of-domain cells hold clamped copies, see above)
    const unsigned lane_off = (unsigned)min(max(c0, 0), ld - 2) * 8u;
#ifdef TILE_FAKE_LOADS
    const char* const in_b = reinterpret_cast<const char*>(inF.base);
    const char* const src_b = reinterpret_cast<const char*>(srcF.base);
#else
    const char* const in_b = reinterpret_cast<const char*>(in);
    const char* const src_b = reinterpret_cast<const char*>(src);
#endif
    // PROLONG: the NROWS consecutive rows of a wave (first row even) need only NROWS / 2 + 1 DISTINCT coarse rows -- fine rows
    // 2k and 2k + 1 both start from coarse row k, odd rows add coarse row k + 1 -- so the coarse correction is fetched once per
    // distinct row (two 8-B loads: columns jc, jc + 1) IN FRONT of the row loads instead of two / four loads per fine row inside a
    // row-at-a-time loop: 8 instead of 18 coarse loads per lane at six rows per wave, and the tile's load phase is one memory
    // round trip instead of a chain of NROWS of them (rounds 1-3 loaded per fine row: profiles/r05_negative/retired_tile_switches.md).
    constexpr bool PWIDE = PROLONG;
    // PROLONG with six or more rows per wave: the sources of the wave's INNER rows never need registers -- they go straight to
    // the lane's LDS slots by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per row, lane-linear: the slot layout of
    // these rows is the lane's (even column, odd column) PAIR here, two rows per KiB-pair of the wave's own inner plane rows), and
    // with them out of the register budget ALL rows of the wave are in flight at once: one memory round trip per tile.
    // The lazy-source factors are applied in place once the data has landed (the issuing wave's own vmcnt orders its reads).
    constexpr bool GLDS = PWIDE && SRCL;
    constexpr int LG = PROLONG ? (GLDS ? NROWS : (TILE_LG_WIDE < NROWS ? TILE_LG_WIDE : NROWS)) : NROWS;
    constexpr int GHALF = (NROWS - 2) / 2 > 0 ? (NROWS - 2) / 2 : 1;  // inner rows per plane chunk (GLDS)
    // wave-uniform base (doubles) of the pair row of inner row i: rows 1 .. GHALF in the wave's inner rows of plane 0, the rest in plane 1
    auto pair_base = [&](int i) { return (i - 1 < GHALF ? A0 : A1) + (lzb + 1) * 64 + ((i - 1) % GHALF) * 128; };
    const char* const cc_b = PROLONG ? reinterpret_cast<const char*>(coarseF.at_slot(b, k_cc)) : nullptr;
    const unsigned ldc8 = (unsigned)ld_c * 8u;
    const int jc = max(c0, 0) >> 1;
    const unsigned jc_off0 = PROLONG ? (unsigned)min(jc, nr_c - 1) * 8u : 0u, jc_off1 = PROLONG ? (unsigned)min(jc + 1, nr_c - 1) * 8u : 0u;
    const int h_end = PROLONG ? min(2 * (nr_c - 1), nr - 1) : 0, v_end = PROLONG ? min(2 * (nz_c - 1), nz - 1) : 0;
    const bool odd_is_zero = c0 + 1 >= h_end;
    constexpr int NCR = PWIDE ? NROWS / 2 + 1 : 1;
    double cr0[NCR], cr1[NCR];  // distinct coarse rows cbase .. cbase + NROWS/2 at columns jc / jc + 1 (clamped into the coarse grid)
    const int cbase = (z0 + lzb) >> 1;  // (z0 + lzb is even; negative above the grid: clamped, those rows are never used)
#pragma unroll
    for (int g0 = 0; g0 < NROWS; g0 += LG) {
        double2 pv[LG], sv2[LG];
        double ca0[LG], ca1[LG], cb0[LG], cb1[LG];
        int gzs[LG];
        if (PWIDE) {
            // the coarse rows this group of fine rows needs and no earlier group has fetched: rows g0 .. g0 + LG - 1 read coarse
            // rows g0 >> 1 .. (g0 + LG) >> 1 (requested together with the group's row loads: one round trip per group)
#pragma unroll
            for (int k = 0; k < NCR; ++k) {
                if (k >= (g0 == 0 ? 0 : (g0 >> 1) + 1) && k <= ((g0 + LG) >> 1)) {
                    const unsigned ra = (unsigned)min(max(cbase + k, 0), nz_c - 1) * ldc8;
                    cr0[k] = *reinterpret_cast<const double*>(cc_b + (ra + jc_off0));
                    cr1[k] = *reinterpret_cast<const double*>(cc_b + (ra + jc_off1));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < LG; ++j) {
            const int i = g0 + j;
            const int gzc = min(max(z0 + lzb + i, 0), nz - 1);  // scalar
            gzs[j] = gzc;
#ifdef TILE_FAKE_LOADS
            const unsigned vo = (unsigned)(lzb + i) * ld8 + (unsigned)lane * 16u;
#else
            const unsigned vo = (unsigned)gzc * ld8 + lane_off;
#endif
            // (in_zero, uniform: the input is the all-zero correction a coarser level starts from -- nothing is loaded)
            if (in_zero) pv[j] = double2{0.0, 0.0};
            else pv[j] = *reinterpret_cast<const double2*>(in_b + vo);
            if (GLDS && i > 0 && i < NROWS - 1)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_b + vo),
                                                 (__attribute__((address_space(3))) void*)pair_base(i), 16, 0, 0);
            else if (jfly && src_b == in_b && !in_zero)
                sv2[j] = pv[j];  // (source on the fly in the pre-smoother: its input IS the field the source is formed from)
            else
                sv2[j] = *reinterpret_cast<const double2*>(src_b + vo);
            if (PWIDE) {
                // (an in-domain row z0 + lzb + i >= 0 has coarse row (z0 + lzb + i) >> 1 = cbase + (i >> 1); rows clamped into the
                //  domain take some other in-domain coarse value -- never used)
                ca0[j] = cr0[i >> 1];
                ca1[j] = cr1[i >> 1];
                if (i & 1) { cb0[j] = cr0[(i >> 1) + 1]; cb1[j] = cr1[(i >> 1) + 1]; }
            }
        }
        if (COEF_LATE && g0 + LG >= NROWS) { ae0 = *pe0; aw0 = *pw0; ae1 = *pe1; aw1 = *pw1; }
        if constexpr (lz_sum) if (g0 == 0) lz_sc = lazy_scale_from_parts(lzs, b, lane, lz_x0, lz_itg, bx_ == 0 && by_ == 0 && threadIdx.x == 0);
        if (FAST && lazy && g0 == 0) {  // Source * omega / a_C = J_raw * [(-mu0 R) * scale * omega / a_C]: one product per value
            sf0 = lz_sc != 0.0 ? (lz_n0 * lz_sc) * fsc : 0.0;
            sf1 = lz_sc != 0.0 ? (lz_n1 * lz_sc) * fsc : 0.0;
        }
#pragma unroll
        for (int j = 0; j < LG; ++j) {
            const int i = g0 + j;
            double vx = pv[j].x, vy = pv[j].y;
            if (PROLONG) {  // bilinear prolongation, core/multigrid_solve.py:117-143 (same expressions as prolong_at)
                const int gzc = gzs[j];
                double px, py;
                if (!(i & 1)) {
                    px = ca0[j];
                    py = odd_is_zero ? 0.0 : 0.5 * (ca0[j] + ca1[j]);
                } else if (gzc >= v_end) {
                    px = 0.0;
                    py = 0.0;
                } else {
                    px = 0.5 * (ca0[j] + cb0[j]);
                    py = odd_is_zero ? 0.0 : 0.25 * (ca0[j] + cb0[j] + ca1[j] + cb1[j]);
                }
                vx = vx + px;
                vy = vy + py;
            }
            if (i & 1) { v1[i] = vx; v0[i] = vy; } else { v0[i] = vx; v1[i] = vy; }
            const bool dma_row = GLDS && i > 0 && i < NROWS - 1;
            double2* const pr = dma_row ? reinterpret_cast<double2*>(pair_base(i)) + lane : nullptr;
            if (dma_row) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's DMA has landed; everything else is needed by now anyway)
            double sx = dma_row ? pr->x : sv2[j].x, sy = dma_row ? pr->y : sv2[j].y;  // source of the even / odd column
            if (FAST) {
                if (jfly) { sx = jraw_shape(sx, jf_pa, jf_id); sy = jraw_shape(sy, jf_pa, jf_id); }
                sx = sx * sf0; sy = sy * sf1;
                if (FOLD) { sx = colok0 ? sx : -vx; sy = colok1 ? sy : -vy; }  // a column that is never updated: the chain returns psi
            } else if (lazy) {  // Source = (-mu0 R) * (scale != 0 ? J_raw * scale : 0), the expression of k_source (uniform branches)
                if (lz_sc != 0.0) { sx = lz_n0 * (sx * lz_sc); sy = lz_n1 * (sy * lz_sc); }
                else { sx = lz_n0 * 0.0; sy = lz_n1 * 0.0; }
            }
            if (dma_row) {
                if (lazy || FAST) *pr = double2{sx, sy};  // (in place; a plain right-hand side is already what the exact sweeps read)
            } else if (SRCL && i > 0 && i < NROWS - 1) {
                b0[i * 64] = (i & 1) ? sy : sx;
                b1[i * 64] = (i & 1) ? sx : sy;
            } else if (i & 1) { s1[i] = sx; s0[i] = sy; }
            else { s0[i] = sx; s1[i] = sy; }
        }
        if (PROLONG) __builtin_amdgcn_sched_barrier(0);
    }
    if (SHARE_D) { if (w < 6) CO[w * 64 + lane] = co_x; }
    // FAST: the stencil coefficients scaled by omega / a_C (FOLD: zero in a column that is never updated)
    const double ans_f = a_ns * fsc;
    double cn0 = ans_f, cn1 = ans_f;
    if (FAST) {
        ae0 = ae0 * fsc; aw0 = aw0 * fsc; ae1 = ae1 * fsc; aw1 = aw1 * fsc;
        if (FOLD) {
            ae0 = colok0 ? ae0 : 0.0; aw0 = colok0 ? aw0 : 0.0; cn0 = colok0 ? ans_f : 0.0;
            ae1 = colok1 ? ae1 : 0.0; aw1 = colok1 ? aw1 : 0.0; cn1 = colok1 ? ans_f : 0.0;
        }
    }
    // the wave's first and last row, both colours, for the waves above and below
    b0[0] = v0[0];
    b1[0] = v1[0];
    b0[(NROWS - 1) * 64] = v0[NROWS - 1];
    b1[(NROWS - 1) * 64] = v1[NROWS - 1];
    // rows of the neighbouring waves (clamped at the tile's first / last row, which is never updated)
    const int o_s = (w > 0 ? -64 : 0), o_n = (w < NW - 1 ? NROWS * 64 : (NROWS - 1) * 64);
    __syncthreads();
    TILE_STAMP(1);
    // row i takes part in pass hs (hs = 0 .. 2S-1) iff it is an interior row of the grid and at least hs + 1 rows away from the
    // tile's first / last row: hs <= hmax[i]
    int hmax[NROWS];
#pragma unroll
    for (int i = 0; i < NROWS; ++i) {
        const int lz = lzb + i, gz = z0 + lz;
        hmax[i] = (gz >= 1 && gz <= nz - 2) ? min(lz - 1, LZ - 2 - lz) : -1;
    }
    int hmax_low = hmax[0];  // (wave-uniform: the first pass index at which some row of this wave is no longer updated, minus one)
#pragma unroll
    for (int i = 1; i < NROWS; ++i) hmax_low = min(hmax_low, hmax[i]);
    // one update of row i, colour c (both compile-time): mine / oth are the register rows of colour c / 1 - c
    // gauss_seidel (compile-time): omega == 1.0, i.e. om1 == 0.0 -- the relaxed value (1 - omega) * old + omega * gs is then
    // fma(0.0, old, gs) in EVERY case (the product 0 * old is exact: +-0, or NaN for a non-finite old, and 1.0 * gs is gs), one
    // instruction instead of three and a dependent chain of 10 instead of 11
    auto update = [&](auto gauss_seidel, auto colour, auto row, const int hs, double (&mine)[NROWS], const double (&oth)[NROWS],
                      const double Sb, const double Nb) {
        constexpr int c = decltype(colour)::value, i = decltype(row)::value, p = (c + i) & 1;
        constexpr bool GS1 = decltype(gauss_seidel)::value;
        // The row test (wave-uniform: hs <= hmax[i]) is NOT a branch: the update is evaluated for every row and kept under
        // (row test & column predicate), so the dependent 11-instruction chains of the wave's rows are independent straight-line
        // code the scheduler interleaves (a lone chain issues one fp64 instruction every ~13 cycles).  A row outside keeps its value.
        const double O = oth[i];
        const double W = p ? O : wave_shift<0x138>(O), E = p ? wave_shift<0x130>(O) : O;
        const double Sv = i > 0 ? oth[i > 0 ? i - 1 : 0] : Sb, N = i < NROWS - 1 ? oth[i < NROWS - 1 ? i + 1 : 0] : Nb;
        const double ae = p ? ae1 : ae0, aw = p ? aw1 : aw0;
        const double sv = (GLDS && i > 0 && i < NROWS - 1) ? pair_base(i)[2 * lane + p]
                          : ((SRCL && i > 0 && i < NROWS - 1) ? (c ? b1 : b0)[i * 64] : (c ? s1[i] : s0[i]));
        if constexpr (FAST) {
            const double cn = FOLD ? (p ? cn1 : cn0) : ans_f;
            const double base = GS1 ? -sv : __builtin_fma(om1, mine[i], -sv);
            double nw = __builtin_fma(ae, E, __builtin_fma(aw, W, __builtin_fma(cn, Sv, __builtin_fma(cn, N, base))));
            if constexpr (FOLD) {
                mine[i] = nw;  // (the caller tests the row, wave-uniformly; the column predicate is in the operands)
            } else {
                asm volatile("" : "+v"(nw));
                const bool keep = (hs <= hmax[i]) & (p ? colok1 : colok0);
                mine[i] = keep ? nw : mine[i];
            }
        } else {
        const double gsv = divc4(ae * E + aw * W + a_ns * Sv + a_ns * N - sv, a_c, inv_a_c, lo_a_c);
        double nw = GS1 ? __builtin_fma(om1, mine[i], gsv) : om1 * mine[i] + omega * gsv;
        asm volatile("" : "+v"(nw));  // (an opaque use: keeps the compiler from sinking the chain into an exec-masked branch)
        const bool keep = (hs <= hmax[i]) & (p ? colok1 : colok0);
        mine[i] = keep ? nw : mine[i];
        }
    };
    auto half_sweep = [&](auto gauss_seidel, auto colour, const int hs) {
        constexpr int c = decltype(colour)::value;
        double(&mine)[NROWS] = c ? v1 : v0;
        const double(&oth)[NROWS] = c ? v0 : v1;
        const double* const ob = c ? b0 : b1;  // boundary rows of the other colour
        const double Sb = ob[o_s], Nb = ob[o_n];
        if ((FOLD || NROWS >= TILE_SKIP_MIN_ROWS) && hmax_low < 2 * S - 1) {
            // a wave some of whose rows drop out before the last pass -- the tile's first / last wave (one row per pass, all of them
            // by the last pass) and waves that overhang the grid: a row is SKIPPED under its wave-uniform test -- 15-20 % of a
            // six-row tile's pass arithmetic; the rows left are short of the other waves' six interleaved chains anyway
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((hs <= hmax[I] ? update(gauss_seidel, colour, std::integral_constant<int, I>{}, hs, mine, oth, Sb, Nb) : (void)0), ...);
            }(std::make_integer_sequence<int, NROWS>{});
        } else {
            // rows that need no LDS value first
            [&]<int... I>(std::integer_sequence<int, I...>) {
                (update(gauss_seidel, colour, std::integral_constant<int, I + 1>{}, hs, mine, oth, Sb, Nb), ...);
            }(std::make_integer_sequence<int, NROWS - 2>{});
            update(gauss_seidel, colour, std::integral_constant<int, 0>{}, hs, mine, oth, Sb, Nb);
            update(gauss_seidel, colour, std::integral_constant<int, NROWS - 1>{}, hs, mine, oth, Sb, Nb);
        }
        double* const mb = c ? b1 : b0;
        mb[0] = mine[0];
        mb[(NROWS - 1) * 64] = mine[NROWS - 1];
    };
    auto sweeps = [&](auto gauss_seidel) {
#pragma unroll 1
        for (int sw = 0; sw < S; ++sw) {
            half_sweep(gauss_seidel, std::integral_constant<int, 0>{}, 2 * sw);
            __syncthreads();
            TILE_STAMP(2 + 2 * sw);
            half_sweep(gauss_seidel, std::integral_constant<int, 1>{}, 2 * sw + 1);
            __syncthreads();
            TILE_STAMP(3 + 2 * sw);
        }
    };
    sweeps(std::integral_constant<bool, GS1T>{});
    // ---- store the window
#pragma unroll
    for (int i = 0; i < NROWS; ++i) {
        const int lz = lzb + i, gz = z0 + lz;
        if (lz >= H && lz < LZ - H && gz < nz && pair_own) {  // (row test wave-uniform)
            const double vx = (i & 1) ? v1[i] : v0[i], vy = (i & 1) ? v0[i] : v1[i];
            char* const o = reinterpret_cast<char*>(out) + ((unsigned)gz * ld8 + (unsigned)c0 * 8u);
            if (pair_full) *reinterpret_cast<double2*>(o) = double2{vx, vy};
            else *reinterpret_cast<double*>(o) = vx;
        }
    }
    TILE_STAMP(8);
    if (DEFECT) {
        // (1) defect d = -(L*psi - src) of this wave's rows, both colours, into the source registers
        if (SHARE_D) {
            rs0 = CO[lane]; ri0 = CO[64 + lane]; rl0 = CO[128 + lane]; rs1 = CO[192 + lane]; ri1 = CO[256 + lane]; rl1 = CO[320 + lane];
        } else if (!EARLY_D) {
            rs0 = *(in0 ? dd.r_safe + c0 : &g_const_f64[1]); ri0 = *(in0 ? dd.r_inv + c0 : &g_const_f64[1]); rl0 = *(in0 ? dd.r_inv_lo + c0 : &g_const_f64[0]);
            rs1 = *(in1 ? dd.r_safe + c0 + 1 : &g_const_f64[1]); ri1 = *(in1 ? dd.r_inv + c0 + 1 : &g_const_f64[1]); rl1 = *(in1 ? dd.r_inv_lo + c0 + 1 : &g_const_f64[0]);
        }
        const double Sb0 = b0[o_s], Nb0 = b0[o_n], Sb1 = b1[o_s], Nb1 = b1[o_n];
        auto defect = [&](auto colour, auto row) {
            constexpr int c = decltype(colour)::value, i = decltype(row)::value, p = (c + i) & 1;
            const double(&mine)[NROWS] = c ? v1 : v0;
            const double(&oth)[NROWS] = c ? v0 : v1;
            // evaluated for every row (straight-line, as the sweeps) and kept where the row lies in the defect ring of the window
            // (H - 1 <= lz <= LZ - H, an interior row of the grid: hmax[i] >= H - 2) and the column is updatable; -0.0 elsewhere
            const double O = oth[i];
            const double W = p ? O : wave_shift<0x138>(O), E = p ? wave_shift<0x130>(O) : O;
            const double Sv = i > 0 ? oth[i > 0 ? i - 1 : 0] : (c ? Sb0 : Sb1);
            const double N = i < NROWS - 1 ? oth[i < NROWS - 1 ? i + 1 : 0] : (c ? Nb0 : Nb1);
            const double C = mine[i];
            constexpr bool inner = SRCL && i > 0 && i < NROWS - 1;
            const double sv = inner ? (c ? b1 : b0)[i * 64] : (c ? s1[i] : s0[i]);
            double dv;
            if constexpr (FAST) {
                // src - L*psi = a_C (psi - gs): u is the omega-scaled Gauss-Seidel value of the point (the chain of the sweeps)
                const double ae = p ? ae1 : ae0, aw = p ? aw1 : aw0, cn = FOLD ? (p ? cn1 : cn0) : ans_f;
                const double u = __builtin_fma(ae, E, __builtin_fma(aw, W, __builtin_fma(cn, Sv, __builtin_fma(cn, N, -sv))));
                dv = dd.kd * (GS1T ? C - u : __builtin_fma(omega, C, -u));
            } else {
            const double rs = p ? rs1 : rs0, ri = p ? ri1 : ri0, rl = p ? rl1 : rl0;
            const double d2r = divc4(E - 2.0 * C + W, dd.dr2, dd.inv_dr2, dd.lo_dr2);
            const double d1r = divc4(E - W, dd.two_dr, dd.inv_two_dr, dd.lo_two_dr);
            const double d2z = divc4(N - 2.0 * C + Sv, dd.dz2, dd.inv_dz2, dd.lo_dz2);
            const double lpsi = d2r - divc4(d1r, rs, ri, rl) + d2z;
            dv = -(lpsi - sv);
            }
            asm volatile("" : "+v"(dv));
            // (FOLD: a column that is never updated has u == psi, i.e. dv == 0 by itself)
            const bool keep = FOLD ? (hmax[i] >= H - 2) : ((hmax[i] >= H - 2) & (p ? colok1 : colok0));
            const double d = keep ? dv : -0.0;
            if (inner) (c ? b1 : b0)[i * 64] = d;  // (the slot its source came from: the planes become the defect planes)
            else if (c) s1[i] = d;
            else s0[i] = d;
        };
        int hmax_top = hmax[0];
#pragma unroll
        for (int i = 1; i < NROWS; ++i) hmax_top = max(hmax_top, hmax[i]);
        if (hmax_top >= H - 2) {  // (wave-uniform: the tile's first / last wave may have no row in the defect ring at all)
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ((defect(std::integral_constant<int, 0>{}, std::integral_constant<int, I>{}),
                  defect(std::integral_constant<int, 1>{}, std::integral_constant<int, I>{})), ...);
            }(std::make_integer_sequence<int, NROWS>{});
        } else {
#pragma unroll
            for (int i = 0; i < NROWS; ++i) {
                if (!(SRCL && i > 0 && i < NROWS - 1)) { s0[i] = -0.0; s1[i] = -0.0; }
                else { b0[i * 64] = -0.0; b1[i * 64] = -0.0; }
            }
        }
        __syncthreads();  // every wave has read its neighbours' boundary rows
        // (2) the planes take the defect
#pragma unroll
        for (int i = 0; i < NROWS; ++i)
            if (!(SRCL && i > 0 && i < NROWS - 1)) { b0[i * 64] = s0[i]; b1[i * 64] = s1[i]; }
        __syncthreads();
        TILE_STAMP(9);
        // (3) full-weighting restriction of the window's even/even points; coarse ring = injected -0.0.  A coarse point sits at
        // tile cell (lz, col) = (H + 2cz, H + 2cr), both even: the centre and the four diagonal neighbours are colour 0, the
        // N/S/W/E neighbours colour 1, and with q = lz*64 + col/2 every address is q + constant.
        constexpr int CZ = (LZ - 2 * H) / 2, CR = (128 - 2 * H) / 2;
        double* rc = coarseF.at_slot(b, k_cc);
        double* ec = dd.e.at_slot(b, k_ce);
        const int ic0 = (z0 + H) >> 1, jc0 = (r0 + H) >> 1;
        for (int t = threadIdx.x; t < CZ * CR; t += 64 * NW) {
            const int cz = t / CR, cr = t - cz * CR;
            const int ic = ic0 + cz, jcn = jc0 + cr;
            if (ic >= nz_c || jcn >= nr_c) continue;
            double v = -0.0;
            if (ic >= 1 && ic <= nz_c - 2 && jcn >= 1 && jcn <= nr_c - 2) {
                const int q = (H + 2 * cz) * 64 + (H / 2 + cr);
                v = fw9(A0[q], A1[q - 64], A1[q + 64], A1[q - 1], A1[q], A0[q - 65], A0[q - 64], A0[q + 63], A0[q + 64]);
            }
            rc[(long)ic * ld_c + jcn] = v;
            if (dd.e.base) ec[(long)ic * ld_c + jcn] = 0.0;  // (nullptr: the consumer of the coarse level starts from zero by itself)
        }
    }
    TILE_STAMP(10);
}


// ---------------------------------------------------------------- whole coarse V-cycle in LDS
// One 1024-thread workgroup per member runs the complete recursion from level `lc` down to the base
// case and back (core/multigrid_solve.py:289-335) with every level resident in LDS: e_lc = Vcycle(0, rhs_lc).
constexpr int CV_MAX = 8;
#ifndef CV_BLOCKS
#define CV_BLOCKS 1  // 17^2 .. 65^2 levels of the coarse kernel: 2x2 register blocks per lane (1) or column pairs with three LDS reads per update (0)
#endif
struct CoarseV {
    int n;                 // number of levels held (last one is the base case)
    int nz[CV_MAX], nr[CV_MAX];
    int off_psi[CV_MAX], off_rhs[CV_MAX], off_tab[CV_MAX];  // LDS offsets (doubles); tab = a_e, a_w, r_safe, r_inv (4*nr)
    int off_scr;           // scratch of nz[0]*nr[0] doubles: the defect field between its evaluation and its restriction
    double a_ns[CV_MAX], a_c[CV_MAX], dr2[CV_MAX], dz2[CV_MAX], two_dr[CV_MAX];
    double inv_a_c[CV_MAX], inv_dr2[CV_MAX], inv_dz2[CV_MAX], inv_two_dr[CV_MAX], lo_a_c[CV_MAX];
    int ld0;       // pitch of the global level-lc arrays
    long stride0;
    const double* tabs;  // the tables of all levels packed as they lie in LDS from off_tab[0] on (tab_total doubles)
    int tab_total;
};
// a double moved along its 16-lane DPP row (CTRL 0x101+n-1: row_shl:n = value of lane+n; 0x111+n-1: row_shr:n = value of lane-n;
// lanes whose source is outside the row receive 0)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// the relaxed value (1 - omega) * old + omega * gs; GS1 (omega == 1.0, om1 == 0.0): fma(0.0, old, gs), the same value in every case
// (the product is exact; see k_smooth_tile) in one instruction instead of three
template <bool GS1>
__device__ __forceinline__ double relaxed(double om1, double old, double omega, double gs) {
    return GS1 ? __builtin_fma(om1, old, gs) : om1 * old + omega * gs;
}
struct CvC { double dr2, dz2, two_dr, inv_dr2, inv_dz2, inv_two_dr; double kd = 1.0, omega = 1.0; };
// One relaxed update of the coarse kernel.  Exact: (1 - omega) old + omega (a_E E + a_W W + a_NS S + a_NS N - src) / a_C in the
// reference's operation order (DIV5: the five-instruction quotient of the general forms).  FAST (contracted arithmetic, see
// k_smooth_tile): the caller's coefficients and source are already scaled by omega / a_C -- one chain of four / five FMAs.
template <bool GS1, bool FAST, bool DIV5 = false>
__device__ __forceinline__ double cv_upd(double ae, double E, double aw, double W, double a_ns, double S, double N, double sv, double old,
                                         double a_c, double inv_a_c, double lo_a_c, double omega, double om1) {
    if constexpr (FAST) {
        return __builtin_fma(ae, E, __builtin_fma(aw, W, __builtin_fma(a_ns, S, __builtin_fma(a_ns, N, GS1 ? -sv : __builtin_fma(om1, old, -sv)))));
    } else {
        const double x = ae * E + aw * W + a_ns * S + a_ns * N - sv;
        return relaxed<GS1>(om1, old, omega, DIV5 ? divc(x, a_c, inv_a_c) : divc4(x, a_c, inv_a_c, lo_a_c));
    }
}
// FAST: the defect src - L*psi = a_C (psi - gs) = kd (omega psi - u), u the omega-scaled Gauss-Seidel value from scaled operands
template <bool GS1>
__device__ __forceinline__ double cv_defect_fast(double p, double ae, double E, double aw, double W, double a_ns, double S, double N, double sv,
                                                 const CvC& k) {
    const double u = __builtin_fma(ae, E, __builtin_fma(aw, W, __builtin_fma(a_ns, S, __builtin_fma(a_ns, N, -sv))));
    return k.kd * (GS1 ? p - u : __builtin_fma(k.omega, p, -u));
}
// L*psi - rhs at one point from its five values (core/multigrid_solve.py:211-249, the expression of cv_resid / mg_residual)
__device__ __forceinline__ double cv_resid_pt(double p, double e, double w, double n, double s, double rhs_c, double r_safe, double r_inv,
                                              const CvC& k) {
    const double d2r = divc(e - 2.0 * p + w, k.dr2, k.inv_dr2);
    const double d1r = divc(e - w, k.two_dr, k.inv_two_dr);
    const double d2z = divc(n - 2.0 * p + s, k.dz2, k.inv_dz2);
    const double lpsi = d2r - divc(d1r, r_safe, r_inv) + d2z;
    return lpsi - rhs_c;
}
// the levels the 2x2-block form of cv_smooth takes (uniform)
__device__ __forceinline__ bool cv_block_level(int nz, int nr) {
    const int nbx = (nr - 1) >> 1, nby = (nz - 1) >> 1, nblk = nbx * nby;
    return CV_BLOCKS && (nr & 1) && (nz & 1) && nbx >= 8 && nbx <= 32 && (nbx & (nbx - 1)) == 0 && nblk <= (int)blockDim.x && (nblk & 63) == 0;
}
// add_c / nrc: (block levels only) the coarse-grid correction whose bilinear prolongation is added to psi while the blocks are loaded
// (psi += P e, the loop of k_vcycle_coarse); dscr / kk: (block levels only) where the defect -(L*psi - rhs) of the smoothed level goes
template <bool GS1, bool FAST = false>
__device__ __forceinline__ void cv_smooth(double* psi, const double* rhs, const double* tab, int nz, int nr, double a_ns,
                                          double a_c, double inv_a_c, double lo_a_c, double omega, double om1, int n_sweeps,
                                          const double* add_c = nullptr, int nrc = 0, double* dscr = nullptr, const CvC* kk = nullptr) {
    const int hw = (nr - 2 + 1) / 2, tot = (nz - 2) * hw;
    const double* a_e = tab;
    const double* a_w = tab + nr;
    const int wi = nr - 2, hi_ = nz - 2;  // interior width / height
    if (wi <= 0 || hi_ <= 0) { __syncthreads(); return; }
    if (wi * hi_ <= 64) {
        // The interior fits one wave (9x9 level, 5x5 base case with its 50 sweeps): one interior point per lane, its value in
        // a register for all passes, the four neighbours from the neighbouring lanes, ring neighbours as per-lane constants --
        // no LDS round trip and no barrier per colour pass.  Interiors of at most 4 rows x 3 columns sit in one 16-lane DPP row
        // (lane = 4 * row + column): the neighbours come by row_shr / row_shl moves instead of ds_bpermute.
        if (CV_BLOCKS && nr == 9 && (nz & 1) && nz >= 5 && nz <= 9) {
            // 9x9 (and 9 columns x 5 / 7 rows) as 2x2 blocks, see the 17x17 .. 65x65 form below: 4 x 4 blocks are the 16 lanes of ONE
            // DPP row, so all four out-of-block neighbours come by row moves (row_shr / row_shl by 1 and by 4) -- no LDS and no
            // ds_bpermute inside the passes.  Lanes without a block hold the ring's +0.0, which is also the moves' fill.
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x, nby = (nz - 1) >> 1;
                const bool part = lane < 4 * nby;
                const int by = part ? lane >> 2 : 0, bx = lane & 3;
                double* const p = psi + (2 * by) * nr + 2 * bx;
                const double* const q = rhs + (2 * by) * nr + 2 * bx;
                double v00 = part ? p[0] : 0.0, v01 = part ? p[1] : 0.0, v10 = part ? p[nr] : 0.0, v11 = part ? p[nr + 1] : 0.0;
                const double s00 = part ? q[0] : 0.0, s01 = part ? q[1] : 0.0, s10 = part ? q[nr] : 0.0, s11 = part ? q[nr + 1] : 0.0;
                const double ae0 = a_e[2 * bx], aw0 = a_w[2 * bx], ae1 = a_e[2 * bx + 1], aw1 = a_w[2 * bx + 1];
                const bool k_tl = part && by >= 1 && bx >= 1, k_tr = part && by >= 1, k_bl = part && bx >= 1, k_br = part;
                for (int s = 0; s < n_sweeps; ++s) {
                    {  // top-left and bottom-right
                        const double W_tl = dpp_f64<0x111>(v01), S_tl = dpp_f64<0x114>(v10);  // lane-1's right column, lane-4's bottom row
                        const double E_br = dpp_f64<0x101>(v10), N_br = dpp_f64<0x104>(v01);  // lane+1's left column, lane+4's top row
                        const double n0 = cv_upd<GS1, FAST>(ae0, v01, aw0, W_tl, a_ns, S_tl, v10, s00, v00, a_c, inv_a_c, lo_a_c, omega, om1);
                        const double n1 = cv_upd<GS1, FAST>(ae1, E_br, aw1, v10, a_ns, v01, N_br, s11, v11, a_c, inv_a_c, lo_a_c, omega, om1);
                        v00 = k_tl ? n0 : v00;
                        v11 = k_br ? n1 : v11;
                    }
                    {  // top-right and bottom-left
                        const double E_tr = dpp_f64<0x101>(v00), S_tr = dpp_f64<0x114>(v11);
                        const double W_bl = dpp_f64<0x111>(v11), N_bl = dpp_f64<0x104>(v00);
                        const double n0 = cv_upd<GS1, FAST>(ae1, E_tr, aw1, v00, a_ns, S_tr, v11, s01, v01, a_c, inv_a_c, lo_a_c, omega, om1);
                        const double n1 = cv_upd<GS1, FAST>(ae0, v11, aw0, W_bl, a_ns, v00, N_bl, s10, v10, a_c, inv_a_c, lo_a_c, omega, om1);
                        v01 = k_tr ? n0 : v01;
                        v10 = k_bl ? n1 : v10;
                    }
                }
                if (part) { p[0] = v00; p[1] = v01; p[nr] = v10; p[nr + 1] = v11; }
            }
            __syncthreads();
            return;
        }
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const bool dpp = wi <= 3 && hi_ <= 4;  // (at most three columns: the fourth lane of a row is the spare one that holds the ring's 0)
            const int lw = dpp ? 4 : wi;  // lanes per interior row
            const int rz = lane / lw, rr = lane - rz * lw;
            const bool in = rz < hi_ && rr < wi;
            const int iz = in ? rz + 1 : 1, ir = in ? rr + 1 : 1, c = iz * nr + ir;
            double v = in ? psi[c] : 0.0;
            const double ae = in ? a_e[ir] : 0.0, aw = in ? a_w[ir] : 0.0, sv = in ? rhs[c] : 0.0;
            const bool hasE = in && ir < nr - 2, hasW = in && ir > 1, hasN = in && iz < nz - 2, hasS = in && iz > 1;
            const double rE = (in && !hasE) ? psi[c + 1] : 0.0, rW = (in && !hasW) ? psi[c - 1] : 0.0;
            const double rN = (in && !hasN) ? psi[c + nr] : 0.0, rS = (in && !hasS) ? psi[c - nr] : 0.0;
            const int par = (iz + ir) & 1;
            if (dpp) {
                for (int s = 0; s < 2 * n_sweeps; ++s) {
                    // (the moves run with every lane enabled: a DPP read from a disabled lane returns 0)
                    // (the ring is +0.0 in every level of k_vcycle_coarse, and so is what the moves deliver for it: the spare lane
                    //  of a 4-lane row, a lane without a point, or the fill outside the 16-lane row -- no select per neighbour)
                    const double E = dpp_f64<0x101>(v), W = dpp_f64<0x111>(v);  // row_shl:1 / row_shr:1
                    const double N = dpp_f64<0x104>(v), S = dpp_f64<0x114>(v);  // row_shl:4 / row_shr:4
                    const double nw = cv_upd<GS1, FAST>(ae, E, aw, W, a_ns, S, N, sv, v, a_c, inv_a_c, lo_a_c, omega, om1);
                    if (in && par == (s & 1)) v = nw;
                }
            } else {
                for (int s = 0; s < 2 * n_sweeps; ++s) {
                    const double Es = __shfl(v, lane + 1, 64), Ws = __shfl(v, lane - 1, 64);
                    const double Ns = __shfl(v, lane + lw, 64), Ss = __shfl(v, lane - lw, 64);
                    const double E = hasE ? Es : rE, W = hasW ? Ws : rW, N = hasN ? Ns : rN, S = hasS ? Ss : rS;
                    const double nw = cv_upd<GS1, FAST>(ae, E, aw, W, a_ns, S, N, sv, v, a_c, inv_a_c, lo_a_c, omega, om1);
                    if (in && par == (s & 1)) v = nw;
                }
            }
            if (in) psi[c] = v;
        }
        __syncthreads();
        return;
    }
    // 2^k+1 levels from 17x17 to 65x65 (CV_BLOCKS): a lane owns the 2x2 BLOCK of points (2by..2by+1, 2bx..2bx+1), values, right-hand
    // sides and column coefficients in registers for all passes.  Block origins are even/even, so a colour pass updates the same two
    // points of every block -- top-left + bottom-right, then top-right + bottom-left: two INDEPENDENT chains per lane, compiled per
    // colour without selects on the operands.  Of a point's four neighbours two are the lane's own registers, one is the
    // neighbouring lane's (a block row is nbx <= 32 consecutive lanes of one wave: wave_shr / wave_shl moves) and one comes from
    // the level's plane in LDS, where every updated value is also written for the rows above / below, the defect and the
    // prolongation: 1 LDS read + 1 write per update instead of 3 + 1, and a pass is one chain deep on 65x65 instead of two.
    // The ring is +0.0 in every use (the levels of k_vcycle_coarse start from e = 0 and the ring is never updated): the lane move
    // across the end of a block row (last block's E neighbour = ring column nr-1) delivers the next row's ring column 0, or the
    // move's 0 fill at the wave's end -- the same +0.0.  17x17 is one wave: no workgroup barrier in its passes.
    {
        const int nbx = (nr - 1) >> 1, nby = (nz - 1) >> 1, nblk = nbx * nby;
        if (cv_block_level(nz, nr)) {
            const int t = threadIdx.x;
            const bool part = t < nblk;  // (wave-uniform: nblk is a multiple of 64)
            const bool one_wave = nblk == 64;
            if (part) {
                const int by = t / nbx, bx = t - by * nbx;
                double* const p = psi + (2 * by) * nr + 2 * bx;  // block origin
                const double* const pS = by >= 1 ? p - nr : p;    // row above the block (first block row: never used)
                const double* const q = rhs + (2 * by) * nr + 2 * bx;
                double v00 = p[0], v01 = p[1], v10 = p[nr], v11 = p[nr + 1];
                const double s00 = q[0], s01 = q[1], s10 = q[nr], s11 = q[nr + 1];
                const double ae0 = a_e[2 * bx], aw0 = a_w[2 * bx], ae1 = a_e[2 * bx + 1], aw1 = a_w[2 * bx + 1];
                const bool k_tl = by >= 1 && bx >= 1, k_tr = by >= 1, k_bl = bx >= 1;  // (the bottom-right point is always interior)
                if (add_c) {
                    // psi += P e on load: the block (2by.., 2bx..) is exactly the fine cell of coarse cell (by, bx) -- the coincident
                    // point, the two edge midpoints and the centre, expressions of prolong_at (core/multigrid_solve.py:117-143; on
                    // a 2^k+1 level none of its zero fills applies inside the blocks, and the fine ring row / column outside them
                    // would receive + 0.0: the coarse ring is +0.0)
                    const double* const c = add_c + by * nrc + bx;
                    const double c00 = c[0], c01 = c[1], c10 = c[nrc], c11 = c[nrc + 1];
                    v00 = v00 + c00;
                    v01 = v01 + 0.5 * (c00 + c01);
                    v10 = v10 + 0.5 * (c00 + c10);
                    v11 = v11 + 0.25 * (c00 + c10 + c01 + c11);
                    p[0] = v00; p[1] = v01; p[nr] = v10; p[nr + 1] = v11;
                    if (one_wave) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    else __syncthreads();
                }
                for (int s = 0; s < n_sweeps; ++s) {
                    {  // colour 0, (row + column) even: top-left (2by, 2bx) and bottom-right (2by+1, 2bx+1)
                        const double S_tl = pS[0], N_br = p[2 * nr + 1];
                        const double W_tl = wave_shift<0x138>(v01), E_br = wave_shift<0x130>(v10);
                        const double n0 = cv_upd<GS1, FAST>(ae0, v01, aw0, W_tl, a_ns, S_tl, v10, s00, v00, a_c, inv_a_c, lo_a_c, omega, om1);
                        const double n1 = cv_upd<GS1, FAST>(ae1, E_br, aw1, v10, a_ns, v01, N_br, s11, v11, a_c, inv_a_c, lo_a_c, omega, om1);
                        v00 = k_tl ? n0 : v00;
                        v11 = n1;
                        p[0] = v00;
                        p[nr + 1] = v11;
                    }
                    if (one_wave) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // (the wave's own LDS writes before its reads)
                    else __syncthreads();
                    {  // colour 1: top-right (2by, 2bx+1) and bottom-left (2by+1, 2bx)
                        const double S_tr = pS[1], N_bl = p[2 * nr];
                        const double E_tr = wave_shift<0x130>(v00), W_bl = wave_shift<0x138>(v11);
                        const double n0 = cv_upd<GS1, FAST>(ae1, E_tr, aw1, v00, a_ns, S_tr, v11, s01, v01, a_c, inv_a_c, lo_a_c, omega, om1);
                        const double n1 = cv_upd<GS1, FAST>(ae0, v11, aw0, W_bl, a_ns, v00, N_bl, s10, v10, a_c, inv_a_c, lo_a_c, omega, om1);
                        v01 = k_tr ? n0 : v01;
                        v10 = k_bl ? n1 : v10;
                        p[1] = v01;
                        p[nr] = v10;
                    }
                    if (one_wave) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    else __syncthreads();
                }
                if (dscr) {
                    // the defect -(L*psi - rhs) of the block's points from the registers (four independent chains; the rows above /
                    // below from the plane, the columns left / right by lane moves), kept at interior points
                    const double* const rs = tab + 2 * nr;  // r_safe, then its reciprocals
                    const double rs0 = rs[2 * bx], ri0 = rs[nr + 2 * bx], rs1 = rs[2 * bx + 1], ri1 = rs[nr + 2 * bx + 1];
                    const double S0 = pS[0], S1 = pS[1], N0 = p[2 * nr], N1 = p[2 * nr + 1];
                    const double W0 = wave_shift<0x138>(v01), W1 = wave_shift<0x138>(v11), E0 = wave_shift<0x130>(v00), E1 = wave_shift<0x130>(v10);
                    double* const d = dscr + (2 * by) * nr + 2 * bx;
                    double d00, d01, d10, d11;
                    if constexpr (FAST) {
                        d00 = cv_defect_fast<GS1>(v00, ae0, v01, aw0, W0, a_ns, S0, v10, s00, *kk);
                        d01 = cv_defect_fast<GS1>(v01, ae1, E0, aw1, v00, a_ns, S1, v11, s01, *kk);
                        d10 = cv_defect_fast<GS1>(v10, ae0, v11, aw0, W1, a_ns, v00, N0, s10, *kk);
                        d11 = cv_defect_fast<GS1>(v11, ae1, E1, aw1, v10, a_ns, v01, N1, s11, *kk);
                    } else {
                        d00 = -cv_resid_pt(v00, v01, W0, v10, S0, s00, rs0, ri0, *kk);
                        d01 = -cv_resid_pt(v01, E0, v00, v11, S1, s01, rs1, ri1, *kk);
                        d10 = -cv_resid_pt(v10, v11, W1, N0, v00, s10, rs0, ri0, *kk);
                        d11 = -cv_resid_pt(v11, E1, v10, N1, v01, s11, rs1, ri1, *kk);
                    }
                    if (k_tl) d[0] = d00;
                    if (k_tr) d[1] = d01;
                    if (k_bl) d[nr] = d10;
                    d[nr + 1] = d11;
                }
            } else if (!one_wave) {
                for (int s = 0; s < 2 * n_sweeps + (add_c ? 1 : 0); ++s) __syncthreads();  // (the waves without blocks keep the barrier count)
            }
            if (one_wave) __syncthreads();
            return;
        }
    }
    // Up to 2 column pairs per thread (65x65 on 1024 threads): the pair's coefficients, right-hand sides and OWN values stay in
    // registers for all passes.  A pass updates one column of the pair, whose E or W neighbour is the thread's other column, so
    // an update costs three LDS reads (S, N and the neighbour outside the pair) and one write.  Rows are dealt to the threads
    // odd rows first, then even rows: the column a lane updates in a pass (e = row parity ^ pass parity) is then the same
    // for a whole wave (for 2^k+1 grids even for the whole workgroup) and the update is compiled for e = 0 and e = 1 without
    // per-lane selects; a wave that straddles the odd / even boundary of an odd-sized grid takes the select form.
    if (tot <= 2 * (int)blockDim.x) {
        constexpr int NJ = 2;
        const int n_odd = (nz - 2 + 1) / 2;  // interior rows 1, 3, 5, ...
        bool ok0[NJ], ok1[NJ];
        int fl[NJ];
        double ae0[NJ], ae1[NJ], aw0[NJ], aw1[NJ], s0[NJ], s1[NJ], v0[NJ], v1[NJ];
        double* pc[NJ];  // &psi[row][col] of the pair's first column
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int t = threadIdx.x + j * blockDim.x;
            ok0[j] = t < tot;
            const int rs = ok0[j] ? t / hw : 0, k = ok0[j] ? t - rs * hw : 0, col = 1 + 2 * k;
            const int iz = rs < n_odd ? 1 + 2 * rs : 2 + 2 * (rs - n_odd);
            ok1[j] = ok0[j] && col + 1 <= nr - 2;
            const int c0 = iz * nr + col;
            pc[j] = psi + c0;
            fl[j] = (iz + 1) & 1;
            ae0[j] = ok0[j] ? a_e[col] : 0.0; aw0[j] = ok0[j] ? a_w[col] : 0.0;
            ae1[j] = ok1[j] ? a_e[col + 1] : 0.0; aw1[j] = ok1[j] ? a_w[col + 1] : 0.0;
            s0[j] = ok0[j] ? rhs[c0] : 0.0; s1[j] = ok1[j] ? rhs[c0 + 1] : 0.0;
            v0[j] = ok0[j] ? psi[c0] : 0.0; v1[j] = ok0[j] ? psi[c0 + 1] : 0.0;  // (col + 1 <= nr - 1: ring or interior)
        }
        const int njs = tot <= (int)blockDim.x ? 1 : NJ;
        for (int s = 0; s < 2 * n_sweeps; ++s) {
            const int parity = s & 1;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j >= njs) continue;
                const int e = fl[j] ^ parity;
                const unsigned long long mv = __ballot(ok0[j]), m1 = __ballot(ok0[j] && e);
                double* const p = pc[j];
                if (m1 == 0) {  // the wave updates the first column of its pairs: E is the lane's own second column
                    if (ok0[j]) {
                        const double nw = cv_upd<GS1, FAST>(ae0[j], v1[j], aw0[j], p[-1], a_ns, p[-nr], p[nr], s0[j], v0[j], a_c, inv_a_c, lo_a_c, omega, om1);
                        v0[j] = nw;
                        p[0] = nw;
                    }
                } else if (m1 == mv) {  // the second column: W is the lane's own first column
                    if (ok1[j]) {
                        const double nw = cv_upd<GS1, FAST>(ae1[j], p[2], aw1[j], v0[j], a_ns, p[1 - nr], p[1 + nr], s1[j], v1[j], a_c, inv_a_c, lo_a_c, omega, om1);
                        v1[j] = nw;
                        p[1] = nw;
                    }
                } else if (e ? ok1[j] : ok0[j]) {  // mixed wave (odd-sized grids only)
                    const double oth = p[e ? 2 : -1];
                    const double Sv = p[e - nr], Nv = p[e + nr];
                    const double E = e ? oth : v1[j], W = e ? v0[j] : oth, C = e ? v1[j] : v0[j];
                    const double nw = cv_upd<GS1, FAST>(e ? ae1[j] : ae0[j], E, e ? aw1[j] : aw0[j], W, a_ns, Sv, Nv, e ? s1[j] : s0[j], C, a_c, inv_a_c,
                                                        lo_a_c, omega, om1);
                    if (e) v1[j] = nw; else v0[j] = nw;
                    p[e] = nw;
                }
            }
            __syncthreads();
        }
        return;
    }
    // thread -> point mapping is the same in every pass: hoist the integer division out of the pass loop
    constexpr int MAXIT = 4;
    int cbase[MAXIT], flag[MAXIT], kcol[MAXIT];
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
        const int t = threadIdx.x + j * blockDim.x;
        cbase[j] = -1;
        flag[j] = 0;
        kcol[j] = 0;
        if (t < tot) {
            const int iz = 1 + t / hw, k = t - (iz - 1) * hw;
            cbase[j] = iz * nr + 1 + 2 * k;
            flag[j] = (iz + 1) & 1;
            kcol[j] = 1 + 2 * k;
        }
    }
    if (tot <= MAXIT * (int)blockDim.x) {
        for (int s = 0; s < 2 * n_sweeps; ++s) {
            const int parity = s & 1;
#pragma unroll
            for (int j = 0; j < MAXIT; ++j) {
                if (cbase[j] >= 0) {
                    const int off = flag[j] ^ parity, ir = kcol[j] + off;
                    if (ir <= nr - 2) {
                        const int c = cbase[j] + off;
                        psi[c] = cv_upd<GS1, FAST, true>(a_e[ir], psi[c + 1], a_w[ir], psi[c - 1], a_ns, psi[c - nr], psi[c + nr], rhs[c], psi[c], a_c, inv_a_c,
                                                         lo_a_c, omega, om1);
                    }
                }
            }
            __syncthreads();
        }
        return;
    }
    for (int s = 0; s < 2 * n_sweeps; ++s) {
        const int parity = s & 1;
        for (int t = threadIdx.x; t < tot; t += blockDim.x) {
            int iz = 1 + t / hw, k = t - (iz - 1) * hw;
            int ir = 1 + ((iz + 1 + parity) & 1) + 2 * k;
            if (ir <= nr - 2) {
                int c = iz * nr + ir;
                psi[c] = cv_upd<GS1, FAST, true>(a_e[ir], psi[c + 1], a_w[ir], psi[c - 1], a_ns, psi[c - nr], psi[c + nr], rhs[c], psi[c], a_c, inv_a_c, lo_a_c,
                                                 omega, om1);
            }
        }
        __syncthreads();
    }
}
__device__ __forceinline__ double cv_resid(const double* psi, const double* rhs, const double* rs, int nr, const CvC& k, int iz,
                                           int ir) {
    int c = iz * nr + ir;  // rs = r_safe table, rs + nr = its reciprocals
    double p = psi[c], e = psi[c + 1], w = psi[c - 1], n = psi[c + nr], s = psi[c - nr];
    double d2r = divc(e - 2.0 * p + w, k.dr2, k.inv_dr2);
    double d1r = divc(e - w, k.two_dr, k.inv_two_dr);
    double d2z = divc(n - 2.0 * p + s, k.dz2, k.inv_dz2);
    double lpsi = d2r - divc(d1r, rs[ir], rs[nr + ir]) + d2z;
    return lpsi - rhs[c];
}
// FAST (contracted arithmetic, see k_smooth_tile): every level's coefficient tables and right-hand side live in LDS scaled by
// omega / a_C of the level (the tables as they are staged, the top-level rhs as it is loaded, a restricted defect as it is
// stored), the updates are FMA chains (cv_upd) and the defect is read off the same chain (cv_defect_fast).
template <bool GS1, bool FAST = false>
__global__ void __launch_bounds__(1024) k_vcycle_coarse(Fld eF, Fld rhsF, CoarseV V, double omega, double om1, int pre,
                                                        int post, const int* active) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x;
    const int act_w = *(active ? active + b : &g_const_i32[1]);  // activity and the two slot words in one round trip
    const int k_rhs = *rhsF.slot_ptr(b), k_e = *eF.slot_ptr(b);
    if (!act_w) return;
    CV_STAMP(0);
    // stage the top-level rhs and the coefficient tables of every level.  The first five rhs values of a thread (all of a 65x65
    // level) are requested before the table loops run, so their round trip to memory overlaps the tables'
    {
        const double* g = rhsF.at_slot(b, k_rhs);
        double* rhs = lds + V.off_rhs[0];
        double* psi = lds + V.off_psi[0];
        const int nz = V.nz[0], nr = V.nr[0], n = nz * nr;
        constexpr int RB = 5;
        double tmp[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int i = min((int)threadIdx.x + k * (int)blockDim.x, n - 1);
            const int iz = i / nr, ir = i - iz * nr;
            tmp[k] = g[(long)iz * V.ld0 + ir];
        }
        if constexpr (FAST) {
            // (a_e, a_w of level l scaled by omega / a_C; r_safe and its reciprocals are not used by the contracted forms)
            for (int i = threadIdx.x; i < V.tab_total; i += blockDim.x) {
                int l = 0;
                while (l + 1 < V.n && i >= V.off_tab[l + 1] - V.off_tab[0]) ++l;
                const double f = (i - (V.off_tab[l] - V.off_tab[0]) < 2 * V.nr[l]) ? omega * V.inv_a_c[l] : 1.0;
                lds[V.off_tab[0] + i] = V.tabs[i] * f;
            }
        } else {
            for (int i = threadIdx.x; i < V.tab_total; i += blockDim.x) lds[V.off_tab[0] + i] = V.tabs[i];  // (one round trip for all levels)
        }
        const double f0 = FAST ? omega * V.inv_a_c[0] : 1.0;
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const int i = threadIdx.x + k * (int)blockDim.x;
            if (i < n) { rhs[i] = FAST ? tmp[k] * f0 : tmp[k]; psi[i] = 0.0; }
        }
        for (int i = threadIdx.x + RB * blockDim.x; i < n; i += blockDim.x) {  // levels beyond 5120 points
            const int iz = i / nr, ir = i - iz * nr;
            const double x = g[(long)iz * V.ld0 + ir];
            rhs[i] = FAST ? x * f0 : x;
            psi[i] = 0.0;
        }
    }
    __syncthreads();
    CV_STAMP(1);
    const double inv_om = FAST ? 1.0 / omega : 1.0;
    for (int l = 0; l < V.n - 1; ++l) {
        double* psi = lds + V.off_psi[l];
        const double* rhs = lds + V.off_rhs[l];
        const double* tab = lds + V.off_tab[l];
        const CvC kk{V.dr2[l], V.dz2[l], V.two_dr[l], V.inv_dr2[l], V.inv_dz2[l], V.inv_two_dr[l], FAST ? V.a_c[l] * inv_om : 1.0, omega};
        double* scr = lds + V.off_scr;
        const bool blk = cv_block_level(V.nz[l], V.nr[l]);  // (block levels: the smoother leaves the defect in the scratch field itself)
        const double ans_l = FAST ? V.a_ns[l] * (omega * V.inv_a_c[l]) : V.a_ns[l];  // (FAST: scaled like the tables)
        cv_smooth<GS1, FAST>(psi, rhs, tab, V.nz[l], V.nr[l], ans_l, V.a_c[l], V.inv_a_c[l], V.lo_a_c[l], omega, om1, pre, nullptr, 0,
                             blk ? scr : nullptr, &kk);
        CV_STAMP(2 + 2 * l);
        // coarse rhs = restrict_full_weight(-(L*psi - rhs)); coarse e = 0
        double* rc = lds + V.off_rhs[l + 1];
        double* pc = lds + V.off_psi[l + 1];
        const int nzc = V.nz[l + 1], nrc = V.nr[l + 1], nrf = V.nr[l], nc = nzc * nrc;
        const double* rs = tab + 2 * nrf;
        // (1) the defect -(L*psi - rhs) of every fine interior point, once, into the scratch field (the restriction stencils of
        //     neighbouring coarse points overlap: evaluating it per coarse point costs 2.25x the arithmetic and LDS reads)
        if (!blk) {
            const int wi = nrf - 2, ni = (V.nz[l] - 2) * wi;
            for (int i = threadIdx.x; i < ni; i += blockDim.x) {
                const int iz = 1 + i / wi, ir = 1 + (i - (iz - 1) * wi);
                if constexpr (FAST) {
                    const int c = iz * nrf + ir;
                    scr[c] = cv_defect_fast<GS1>(psi[c], tab[ir], psi[c + 1], tab[nrf + ir], psi[c - 1], ans_l, psi[c - nrf], psi[c + nrf], rhs[c], kk);
                } else {
                    scr[iz * nrf + ir] = -cv_resid(psi, rhs, rs, nrf, kk, iz, ir);
                }
            }
        }
        __syncthreads();
        // (2) full weighting on the coarse interior (one point per thread up to 33x33), -0.0 on the coarse ring, coarse e = 0
        {
            const int wi = nrc - 2, ni = (nzc - 2) * wi;
            for (int i = threadIdx.x; i < ni; i += blockDim.x) {
                const int ic = 1 + i / wi, jc = 1 + (i - (ic - 1) * wi);
                const double* d = scr + (2 * ic) * nrf + 2 * jc;
                const double rv = fw9(d[0], d[-nrf], d[nrf], d[-1], d[1], d[-nrf - 1], d[-nrf + 1], d[nrf - 1], d[nrf + 1]);
                rc[ic * nrc + jc] = FAST ? rv * (omega * V.inv_a_c[l + 1]) : rv;
            }
            for (int i = threadIdx.x; i < nc; i += blockDim.x) {
                const int ic = i / nrc, jc = i - ic * nrc;
                if (!(jc > 0 && jc < nrc - 1 && ic > 0 && ic < nzc - 1)) rc[i] = -0.0;
                pc[i] = 0.0;
            }
        }
        __syncthreads();
        CV_STAMP(3 + 2 * l);
    }
    {
        const int l = V.n - 1;  // base case: 50 sweeps (multigrid_solve.py:293)
        cv_smooth<GS1, FAST>(lds + V.off_psi[l], lds + V.off_rhs[l], lds + V.off_tab[l], V.nz[l], V.nr[l],
                             FAST ? V.a_ns[l] * (omega * V.inv_a_c[l]) : V.a_ns[l], V.a_c[l], V.inv_a_c[l], V.lo_a_c[l], omega, om1, 50);
        CV_STAMP(20);
    }
    for (int l = V.n - 2; l >= 0; --l) {
        double* psi = lds + V.off_psi[l];
        const double* pc = lds + V.off_psi[l + 1];
        const int nzf = V.nz[l], nrf = V.nr[l], nzc = V.nz[l + 1], nrc = V.nr[l + 1], n = nzf * nrf;
        // psi += P e, one thread per coarse cell: its four coarse corners are read once and serve the coincident fine point, the
        // two edge midpoints and the cell centre (expressions and zero fill of prolong_at, core/multigrid_solve.py:117-143)
        (void)n;
        // (block levels: the post-smoother adds the prolonged correction while it loads its blocks)
        const bool blk = cv_block_level(nzf, nrf) && 2 * (nrc - 1) == nrf - 1 && 2 * (nzc - 1) == nzf - 1;
        if (!blk) {
            const int h_end = min(2 * (nrc - 1), nrf - 1), v_end = min(2 * (nzc - 1), nzf - 1);
            for (int i = threadIdx.x; i < nzc * nrc; i += blockDim.x) {
                const int ic = i / nrc, jc = i - ic * nrc;
                const int fi = 2 * ic, fj = 2 * jc;
                const bool jn = jc + 1 < nrc, in_ = ic + 1 < nzc;
                const double c00 = pc[i], c01 = jn ? pc[i + 1] : 0.0, c10 = in_ ? pc[i + nrc] : 0.0, c11 = (jn && in_) ? pc[i + nrc + 1] : 0.0;
                double* f = psi + fi * nrf + fj;
                f[0] = f[0] + c00;
                if (fj + 1 < nrf) f[1] = f[1] + ((fj + 1 >= h_end) ? 0.0 : 0.5 * (c00 + c01));
                if (fi + 1 < nzf) {
                    f[nrf] = f[nrf] + ((fi + 1 >= v_end) ? 0.0 : 0.5 * (c00 + c10));
                    if (fj + 1 < nrf) f[nrf + 1] = f[nrf + 1] + ((fi + 1 >= v_end || fj + 1 >= h_end) ? 0.0 : 0.25 * (c00 + c10 + c01 + c11));
                }
            }
            __syncthreads();
        }
        CV_STAMP(21 + 2 * l);
        cv_smooth<GS1, FAST>(psi, lds + V.off_rhs[l], lds + V.off_tab[l], nzf, nrf, FAST ? V.a_ns[l] * (omega * V.inv_a_c[l]) : V.a_ns[l], V.a_c[l],
                             V.inv_a_c[l], V.lo_a_c[l], omega, om1, post, blk ? pc : nullptr, nrc);
        CV_STAMP(22 + 2 * l);
    }
    {
        double* g = eF.at_slot(b, k_e);
        const double* psi = lds + V.off_psi[0];
        const int nz = V.nz[0], nr = V.nr[0], n = nz * nr;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            int iz = i / nr, ir = i - iz * nr;
            g[(long)iz * V.ld0 + ir] = psi[i];
        }
    }
    CV_STAMP(40);
}

// ADD: psi += P e (multigrid_solve.py:329-330) else fine = P c; grid (ceil(nr_f/64), ceil(nz_f/4), B)
template <bool ADD>
__global__ void k_prolong(Fld fF, Fld cF, int nz_f, int nr_f, int ld_f, int nz_c, int nr_c, int ld_c,
                          const int* active) {
    int b = blockIdx.z;
    if (active && !active[b]) return;
    int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (j >= nr_f || i >= nz_f) return;
    double p = prolong_at(cF.at(b), ld_c, nz_c, nr_c, nz_f, nr_f, i, j);
    double* f = fF.at(b) + (long)i * ld_f + j;
    if (ADD) *f = *f + p;
    else *f = p;
}

// ---------------------------------------------------------------- Jacobi / SOR steps of the Picard driver
// core/fusion_kernel_iterative_solver.py:54-95 (sanitize, update, clip); out != psi
__global__ void k_jacobi(Fld outF, Fld psiF, Fld srcF, Lvl L, const int* active) {
    int b = blockIdx.z;
    if (active && !active[b]) return;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= L.nr || iz >= L.nz) return;
    const double* psi = psiF.at(b);
    const double* src = srcF.at(b);
    long c = (long)iz * L.ld + ir;
    double v = sanitize1(psi[c]);
    if (iz > 0 && iz < L.nz - 1 && ir > 0 && ir < L.nr - 1) {
        v = (L.a_e[ir] * sanitize1(psi[c + 1]) + L.a_w[ir] * sanitize1(psi[c - 1]) +
             L.a_ns * sanitize1(psi[c - L.ld]) + L.a_ns * sanitize1(psi[c + L.ld]) - sanitize1(src[c])) /
            L.a_c;
        v = fmin(fmax(v, -CAP), CAP);
    }
    outF.at(b)[c] = v;
}
// S Jacobi steps per launch (temporal blocking of the 50-step seed relaxation, core/fusion_kernel_iterative_solver.py:408-410) in
// the register-tile form of k_smooth_tile (round 4; the LDS-tile form of rounds 1-3: profiles/r04_negative/): a 48 x 128 tile, eight waves,
// a wave owns six consecutive rows and a lane the column pair (2l, 2l+1) of those rows -- psi, source and the column coefficients
// stay in VGPRs from the load to the store.  A step updates every cell from the OLD values of its four neighbours: E / W of the
// pair's outer sides by wave_shl / wave_shr DPP moves, N / S from the registers of the neighbouring rows (a two-row rolling copy
// of the old values while the rows are overwritten in place), the wave's first / last row from the neighbouring waves through
// LDS (double-buffered by step parity: ONE barrier per step).  Halo 6 (even: pairs stay aligned) >= S, window 36 x 116.
// Out-of-domain cells hold clamped copies that no kept update or store reads, as in the smoother.  Same arithmetic as k_jacobi
// (sanitised inputs, x / a_c through divc4, clip), bit-identical for finite x.
// DAMPED: the damped-Jacobi step of the polyglot lane (core/jax_gs_solver.py:58-86, the arithmetic of k_jacobi_damped below:
// a_NS * (S + N) association, no sanitising, no clip, psi <- (1 - omega_j) psi + omega_j * update) in the same blocked form --
// Jacobi reads only OLD values, so S steps per launch are bit-identical to S launches of one step.
// FAST (contracted arithmetic, the seed relaxation of a resident handle in that mode; never with DAMPED: the polyglot lane is a stateless
// operator and stays exact): the step as a chain of four FMAs on coefficients / sources scaled by 1 / a_C once per tile, then the clip.
template <int S, bool DAMPED = false, bool FAST = false>
__global__ void __launch_bounds__(512, 6) k_jacobi_reg(Fld outF, Fld psiF, Fld srcF, Lvl L, const int* active, double omega_j = 1.0,
                                                       double om1_j = 0.0) {
    static_assert(!(FAST && DAMPED), "the damped form is the polyglot lane's: exact only");
    constexpr int LZ = 48, NW = 8, NROWS = LZ / NW, H = 6;
    static_assert(S <= H, "the halo must cover the steps of one launch");
    __shared__ __attribute__((aligned(16))) double2 xch[2][NW][2][64];  // [step parity][wave][first / last row][lane]
    int bx_, by_, b;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        const unsigned lin = xcd_contiguous(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
        bx_ = (int)(lin % gx);
        by_ = (int)((lin / gx) % gy);
        b = (int)(lin / (gx * gy));
    }
    const int act_w = *(active ? active + b : &g_const_i32[1]);
    const int k_in = *psiF.slot_ptr(b), k_out = *outF.slot_ptr(b), k_src = *srcF.slot_ptr(b);
    if (!act_w) return;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lzb = w * NROWS;
    const int nz = L.nz, nr = L.nr, ld = L.ld;
    const int z0 = by_ * (LZ - 2 * H) - H, r0 = bx_ * (128 - 2 * H) - H;  // both even
    const int c0 = r0 + 2 * lane;
    const bool in0 = c0 >= 1 && c0 <= nr - 2, in1 = c0 + 1 >= 1 && c0 + 1 <= nr - 2;
    const double ae0 = *(in0 ? L.a_e + c0 : &g_const_f64[0]), aw0 = *(in0 ? L.a_w + c0 : &g_const_f64[0]);
    const double ae1 = *(in1 ? L.a_e + c0 + 1 : &g_const_f64[0]), aw1 = *(in1 ? L.a_w + c0 + 1 : &g_const_f64[0]);
    const char* const in_b = reinterpret_cast<const char*>(psiF.at_slot(b, k_in));
    const char* const src_b = reinterpret_cast<const char*>(srcF.at_slot(b, k_src));
    double* const out = outF.at_slot(b, k_out);
    const unsigned ld8 = (unsigned)ld * 8u;
    const unsigned lane_off = (unsigned)min(max(c0, 0), ld - 2) * 8u;
    double2 v[NROWS], sv[NROWS];
#pragma unroll
    for (int i = 0; i < NROWS; ++i) {
        const int gzc = min(max(z0 + lzb + i, 0), nz - 1);  // scalar
        const unsigned vo = (unsigned)gzc * ld8 + lane_off;
        v[i] = *reinterpret_cast<const double2*>(in_b + vo);
        sv[i] = *reinterpret_cast<const double2*>(src_b + vo);
    }
    if (!DAMPED) {
#pragma unroll
        for (int i = 0; i < NROWS; ++i) {
            v[i].x = sanitize1(v[i].x); v[i].y = sanitize1(v[i].y);
            sv[i].x = sanitize1(sv[i].x); sv[i].y = sanitize1(sv[i].y);
        }
    }
    // a cell is updated iff it is an interior grid point and not on the tile's rim (rows: wave-uniform tests)
    const bool colok0 = lane >= 1 && in0, colok1 = lane <= 62 && in1;
    bool rowok[NROWS];
#pragma unroll
    for (int i = 0; i < NROWS; ++i) {
        const int lz = lzb + i, gz = z0 + lz;
        rowok[i] = gz >= 1 && gz <= nz - 2 && lz >= 1 && lz <= LZ - 2;
    }
    const double a_ns = L.a_ns, a_c = L.a_c, inv_a_c = L.inv_a_c, lo_a_c = L.lo_a_c;
    const double fe0 = ae0 * inv_a_c, fw0 = aw0 * inv_a_c, fe1 = ae1 * inv_a_c, fw1 = aw1 * inv_a_c, fns = a_ns * inv_a_c;  // (FAST)
    if (FAST) {
#pragma unroll
        for (int i = 0; i < NROWS; ++i) { sv[i].x = sv[i].x * inv_a_c; sv[i].y = sv[i].y * inv_a_c; }
    }
    const int w_s = w > 0 ? w - 1 : 0, w_n = w < NW - 1 ? w + 1 : NW - 1;  // (the tile's first / last row is never updated)
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        const int pb = s & 1;
        xch[pb][w][0][lane] = v[0];
        xch[pb][w][1][lane] = v[NROWS - 1];
        __syncthreads();
        double2 old_m1 = xch[pb][w_s][1][lane];           // last row of the wave above in the tile (row index - 1)
        const double2 nb = xch[pb][w_n][0][lane];         // first row of the wave below
#pragma unroll
        for (int i = 0; i < NROWS; ++i) {
            const double2 o = v[i];
            const double2 up = i < NROWS - 1 ? v[i < NROWS - 1 ? i + 1 : 0] : nb;
            const double Wx = wave_shift<0x138>(o.y), Ey = wave_shift<0x130>(o.x);
            double cx, cy;
            if constexpr (DAMPED) {
                const double ux = divc4(ae0 * o.y + aw0 * Wx + a_ns * (old_m1.x + up.x) - sv[i].x, a_c, inv_a_c, lo_a_c);
                const double uy = divc4(ae1 * Ey + aw1 * o.x + a_ns * (old_m1.y + up.y) - sv[i].y, a_c, inv_a_c, lo_a_c);
                cx = om1_j * o.x + omega_j * ux;
                cy = om1_j * o.y + omega_j * uy;
            } else if constexpr (FAST) {
                const double nx = __builtin_fma(fe0, o.y, __builtin_fma(fw0, Wx, __builtin_fma(fns, old_m1.x, __builtin_fma(fns, up.x, -sv[i].x))));
                const double ny = __builtin_fma(fe1, Ey, __builtin_fma(fw1, o.x, __builtin_fma(fns, old_m1.y, __builtin_fma(fns, up.y, -sv[i].y))));
                cx = fmin(fmax(nx, -CAP), CAP);
                cy = fmin(fmax(ny, -CAP), CAP);
            } else {
                const double nx = divc4(ae0 * o.y + aw0 * Wx + a_ns * old_m1.x + a_ns * up.x - sv[i].x, a_c, inv_a_c, lo_a_c);
                const double ny = divc4(ae1 * Ey + aw1 * o.x + a_ns * old_m1.y + a_ns * up.y - sv[i].y, a_c, inv_a_c, lo_a_c);
                cx = fmin(fmax(nx, -CAP), CAP);
                cy = fmin(fmax(ny, -CAP), CAP);
            }
            v[i].x = (rowok[i] & colok0) ? cx : o.x;
            v[i].y = (rowok[i] & colok1) ? cy : o.y;
            old_m1 = o;
        }
    }
    const bool pair_own = lane >= H / 2 && lane < 64 - H / 2 && c0 < nr;
    const bool pair_full = c0 + 1 < nr;
#pragma unroll
    for (int i = 0; i < NROWS; ++i) {
        const int lz = lzb + i, gz = z0 + lz;
        if (lz >= H && lz < LZ - H && gz < nz && pair_own) {  // (row test wave-uniform)
            char* const o = reinterpret_cast<char*>(out) + ((unsigned)gz * ld8 + (unsigned)c0 * 8u);
            if (pair_full) *reinterpret_cast<double2*>(o) = v[i];
            else *reinterpret_cast<double*>(o) = v[i].x;
        }
    }
}
// damped Jacobi of the polyglot lane, core/jax_gs_solver.py:58-86 (a_NS*(S+N) association)
__global__ void k_jacobi_damped(Fld outF, Fld psiF, Fld srcF, Lvl L, double omega_j, double om1) {
    int b = blockIdx.z;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= L.nr || iz >= L.nz) return;
    const double* psi = psiF.at(b);
    long c = (long)iz * L.ld + ir;
    double v = psi[c];
    if (iz > 0 && iz < L.nz - 1 && ir > 0 && ir < L.nr - 1) {
        double upd = (L.a_e[ir] * psi[c + 1] + L.a_w[ir] * psi[c - 1] + L.a_ns * (psi[c - L.ld] + psi[c + L.ld]) -
                      srcF.at(b)[c]) /
                     L.a_c;
        v = om1 * v + omega_j * upd;
    }
    outF.at(b)[c] = v;
}
__global__ void k_sanitize(Fld dst, Fld src, int nz, int nr, int ld) {
    int b = blockIdx.z;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    dst.at(b)[(long)iz * ld + ir] = sanitize1(src.at(b)[(long)iz * ld + ir]);
}

// ---------------------------------------------------------------- Picard: topology (a8)
// argmax psi, argmin |grad psi| inside the divertor rows [0, n_mask_rows), min finite psi.
// np.gradient semantics (2nd-order centred inside, 1st-order one-sided on the array edges),
// core/fusion_kernel.py:276-288, 342-355.  grid (nblk, B), 256 threads x 4 points.
__global__ void __launch_bounds__(256) k_topology(Fld psiF, const Member* st, int nz, int nr, int ld,
                                                  double dr, double dz, int n_mask_rows,
                                                  TopoPart* part, int nblk) {
    __shared__ VI shv[4];
    __shared__ double shd[4];
    int b = blockIdx.y;
    if (!st[b].active) return;
    const double* psi = psiF.at(b);
    const long n = (long)nz * nr;
    VI mx{0.0, -1}, mn{0.0, -1};
    double minpsi = INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (q >= n) continue;
        int iz = (int)(q / nr), ir = (int)(q - (long)iz * nr);
        long c = (long)iz * ld + ir;
        double p = psi[c];
        if (p == p) mx = vi_max(mx, VI{p, q});
        if (finite_d(p)) minpsi = fmin(minpsi, p);
        if (iz < n_mask_rows) {
            double gz, gr;
            if (iz == 0) gz = (nan_to_num300(psi[c + ld]) - nan_to_num300(p)) / dz;
            else if (iz == nz - 1) gz = (nan_to_num300(p) - nan_to_num300(psi[c - ld])) / dz;
            else gz = (nan_to_num300(psi[c + ld]) - nan_to_num300(psi[c - ld])) / (2.0 * dz);
            if (ir == 0) gr = (nan_to_num300(psi[c + 1]) - nan_to_num300(p)) / dr;
            else if (ir == nr - 1) gr = (nan_to_num300(p) - nan_to_num300(psi[c - 1])) / dr;
            else gr = (nan_to_num300(psi[c + 1]) - nan_to_num300(psi[c - 1])) / (2.0 * dr);
            double bm = hypot(gr, gz);
            if (finite_d(bm)) mn = vi_min(mn, VI{bm, q});
        }
    }
    VI rmx = block_vi<true, 256>(mx, shv);
    VI rmn = block_vi<false, 256>(mn, shv);
    double rmp = -block_max<256>(-minpsi, shd);
    if (threadIdx.x == 0) {
        TopoPart t;
        t.maxv = rmx.v; t.maxi = rmx.i; t.minb = rmn.v; t.mini = rmn.i; t.minpsi = rmp;
        t.psi_at_min = rmn.i >= 0 ? psi[(rmn.i / nr) * (long)ld + (rmn.i % nr)] : 0.0;
        part[(long)b * nblk + blockIdx.x] = t;
    }
}
// one block per member: finish the reductions and apply the outer-loop heuristics
// (core/fusion_kernel.py:350-355, 288-293, 339-340; core/fusion_kernel_newton_solver.py:505-506)
__global__ void __launch_bounds__(256) k_topology_final(Fld psiF, Member* st, const TopoPart* part, int nblk,
                                                        int nr, int ld, const double* r_axis,
                                                        const double* z_axis) {
    // One block per member on the dependent chain of every Picard iteration: the activity flag and the partials are requested
    // together (one round trip), the three reductions share one barrier (same combination order as block_vi / block_max).
    __shared__ VI shx[4], shn[4];
    __shared__ double shd[4];
    const int b = blockIdx.x;
    const int act = st[b].active;
    VI mx{0.0, -1}, mn{0.0, -1};
    double minpsi = INFINITY;
    if ((int)threadIdx.x < nblk) {
        const TopoPart t = part[(long)b * nblk + threadIdx.x];
        mx = VI{t.maxv, t.maxi};
        mn = VI{t.minb, t.mini};
        minpsi = t.minpsi;
        if (mx.i < 0) mx = VI{0.0, -1};  // (what vi_max / vi_min against the empty start value would keep)
        if (mn.i < 0) mn = VI{0.0, -1};
    }
    if (!act) return;
    for (int i = threadIdx.x + 256; i < nblk; i += 256) {
        const TopoPart t = part[(long)b * nblk + i];
        mx = vi_max(mx, VI{t.maxv, t.maxi});
        mn = vi_min(mn, VI{t.minb, t.mini});
        minpsi = fmin(minpsi, t.minpsi);
    }
    mx = wave_vi<true>(mx);
    mn = wave_vi<false>(mn);
    const double wmp = wave_max(-minpsi);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { shx[w] = mx; shn[w] = mn; shd[w] = wmp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        VI rmx = shx[0], rmn = shn[0];
        double rmp = shd[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) { rmx = vi_max(rmx, shx[i]); rmn = vi_min(rmn, shn[i]); rmp = fmax(rmp, shd[i]); }
        rmp = -rmp;
        Member& m = st[b];
        const double* psi = psiF.at(b);
        double pa = rmx.i >= 0 ? rmx.v : 0.0;
        if (fabs(pa) < 1e-6) pa = 1e-6;
        double pb, xr = 0.0, xz = 0.0;
        if (rmn.i >= 0) {
            int iz = (int)(rmn.i / nr), ir = (int)(rmn.i - (long)iz * nr);
            pb = psi[(long)iz * ld + ir];
            xr = r_axis[ir];  // (the three loads are independent: one more round trip, not three)
            xz = z_axis[iz];
            if (!finite_d(pb)) pb = nan_to_num300(pb);
        } else {
            pb = rmp;
        }
        if (fabs(pa - pb) < 0.1) pb = pa * 0.1;
        double denom = pb - pa;  // core/fusion_kernel.py:415-417
        if (fabs(denom) < 1e-9) denom = 1e-9;
        m.psi_axis = pa; m.psi_bnd = pb; m.denom = denom; m.inv_denom = 1.0 / denom; m.x_idx = rmn.i; m.x_r = xr; m.x_z = xz;
    }
}

// solver.xpoint_use_saddle_detection (core/fusion_kernel.py:284-329): after k_topology_final picked the minimum of the masked
// |grad psi|, walk the (up to) 16 smallest candidates in ascending (|grad psi|, index) order and take the first one that is an
// interior saddle (det Hessian < 0); none -> the minimum stays.  One block of 1024 threads per member.
__global__ void __launch_bounds__(1024) k_xpoint_saddle(Fld psiF, Member* st, int nz, int nr, int ld, double dr, double dz,
                                                        int n_mask_rows, const double* r_axis, const double* z_axis) {
    __shared__ VI shv[16];
    __shared__ VI cand;
    __shared__ int done;
    const int b = blockIdx.x;
    if (!st[b].active || st[b].x_idx < 0) return;
    const double* psi = psiF.at(b);
    const long nm = (long)n_mask_rows * nr;  // masked rows are iz < n_mask_rows (Z ascending)
    double last_b = -1.0;
    long last_q = -1;
    for (int k = 0; k < 16; ++k) {
        VI mn{0.0, -1};
        for (long q = threadIdx.x; q < nm; q += 1024) {
            const int iz = (int)(q / nr), ir = (int)(q - (long)iz * nr);
            const long c = (long)iz * ld + ir;
            const double p = psi[c];
            double gz, gr;
            if (iz == 0) gz = (nan_to_num300(psi[c + ld]) - nan_to_num300(p)) / dz;
            else if (iz == nz - 1) gz = (nan_to_num300(p) - nan_to_num300(psi[c - ld])) / dz;
            else gz = (nan_to_num300(psi[c + ld]) - nan_to_num300(psi[c - ld])) / (2.0 * dz);
            if (ir == 0) gr = (nan_to_num300(psi[c + 1]) - nan_to_num300(p)) / dr;
            else if (ir == nr - 1) gr = (nan_to_num300(p) - nan_to_num300(psi[c - 1])) / dr;
            else gr = (nan_to_num300(psi[c + 1]) - nan_to_num300(psi[c - 1])) / (2.0 * dr);
            const double bm = hypot(gr, gz);
            if (!finite_d(bm)) continue;
            if (bm < last_b || (bm == last_b && q <= last_q)) continue;  // already visited
            mn = vi_min(mn, VI{bm, q});
        }
        VI r = block_vi<false, 1024>(mn, shv);
        if (threadIdx.x == 0) {
            cand = r;
            done = 0;
            if (r.i < 0) done = 1;  // fewer finite candidates than 16
            else {
                const int iz = (int)(r.i / nr), ir = (int)(r.i - (long)iz * nr);
                if (iz > 0 && iz < nz - 1 && ir > 0 && ir < nr - 1) {
                    const long c = (long)iz * ld + ir;
#define PS(o) nan_to_num300(psi[c + (o)])
                    const double d2r = (PS(1) - 2.0 * PS(0) + PS(-1)) / (dr * dr);
                    const double d2z = (PS(ld) - 2.0 * PS(0) + PS(-ld)) / (dz * dz);
                    const double drz = (PS(ld + 1) - PS(ld - 1) - PS(-ld + 1) + PS(-ld - 1)) / (4.0 * dr * dz);
#undef PS
                    const double det = d2r * d2z - drz * drz;
                    if (finite_d(det) && det < 0.0) {
                        Member& m = st[b];
                        double pb = psi[c];
                        if (!finite_d(pb)) pb = nan_to_num300(pb);
                        const double pa = m.psi_axis;
                        if (fabs(pa - pb) < 0.1) pb = pa * 0.1;
                        double denom = pb - pa;
                        if (fabs(denom) < 1e-9) denom = 1e-9;
                        m.psi_bnd = pb; m.denom = denom; m.inv_denom = 1.0 / denom; m.x_idx = r.i; m.x_r = r_axis[ir]; m.x_z = z_axis[iz];
                        done = 1;
                    }
                }
            }
        }
        __syncthreads();
        if (done) return;
        last_b = cand.v;
        last_q = cand.i;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- Picard: source (a9)
// partial sums of J_raw.  Each thread owns a column pair (16-B loads) of 4 rows: block = 64 pairs x 4 row-lanes,
// 16 rows per block.  grid (ceil(nr/128), ceil(nz/16), B); partial index = blockIdx.y * gridDim.x + blockIdx.x
// (16 rows -- four per wave -- are equal for the 32-member ensemble and -1.6 % per Picard iteration for ONE equilibrium, but regroup
//  the partial sums of J_raw: the last bits of the current integral move and the finite-difference Newton lane, which is sensitive to
//  them, takes 33 instead of 22 inner iterations on the DIII-D-like case; 64 stays: profiles/r04_experiments.md)
#ifndef JS_ROWS_N
#define JS_ROWS_N 64
#endif
constexpr int JS_ROWS = JS_ROWS_N;  // rows per workgroup of the column-pair kernels (k_jsum, k_source)
// finishes the J_raw sum of member b: I = (sum * dr) * dz, scale = I_target / I  (core/fusion_kernel.py:436-438); all 256 threads
__device__ __forceinline__ void jsum_final_body(int b, Member* st, const double* part, int nblk, double dr, double dz, double* sh,
                                                bool commit = true) {
    const double itg = st[b].i_target;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += ld_val(part + (long)b * nblk + i);
    double tot = block_sum<256>(s, sh);
    if (threadIdx.x == 0 && commit) {
        double ic = (tot * dr) * dz;
        st[b].i_current = ic;
        st[b].jscale = fabs(ic) > 1e-9 ? itg / ic : 0.0;
    }
}
// FLY (source on the fly, see LazySrc): J_raw in the form its consumers rebuild it in -- (0.5 R + 0.5 / (mu0 R)) * shape(psi_N) -- so that
// the sum normalises exactly the current density the smoothers and the tail will see; nothing is stored (rawF.base == nullptr).
template <int HM, bool FLY = false>
__global__ void __launch_bounds__(256) k_jsum(Fld psiF, const Member* st, int nz, int nr, int ld,
                                              const double* r_axis, const double* inv_mu0R, Prof pf,
                                              double* part, int nblk, Fld rawF) {
    __shared__ double sh[4];
    int b = blockIdx.z;
    // the member's words in one round trip (activity, slots, normalisation), as in the tile kernels
    const int act_w = st[b].active;
    const int k_psi = *psiF.slot_ptr(b), k_raw = *rawF.slot_ptr(b);
    const double pa = st[b].psi_axis, den = st[b].denom, iden = st[b].inv_denom;
    if (!act_w) return;
    const double* psi = psiF.at_slot(b, k_psi);
    double* raw = rawF.base ? rawF.at_slot(b, k_raw) : nullptr;  // lazy-source mode: J_raw is kept, consumers apply scale and -mu0 R
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + tx) * 2;
    double s = 0.0;
    if (c0 < nr) {
        const bool two = c0 + 1 < nr;
        const double R0 = r_axis[c0], R1 = two ? r_axis[c0 + 1] : 1.0;
        const double m0 = inv_mu0R[c0], m1 = two ? inv_mu0R[c0 + 1] : 0.0;
        // rows in batches of JS_BATCH: all loads of a batch are issued before the first J_raw store (the compiler may not move
        // a load above a store through another pointer, so load / compute / store per row would serialise the memory round trips)
        constexpr int JS_BATCH = JS_ROWS / 4 < 8 ? JS_ROWS / 4 : 8;
#pragma unroll
        for (int k0 = 0; k0 < JS_ROWS / 4; k0 += JS_BATCH) {
            double2 v[JS_BATCH];
#pragma unroll
            for (int j = 0; j < JS_BATCH; ++j) {
                const int iz = min(blockIdx.y * JS_ROWS + ty + 4 * (k0 + j), nz - 1);  // clamped: rows beyond nz are skipped below
                v[j] = *reinterpret_cast<const double2*>(psi + (long)iz * ld + c0);
            }
#pragma unroll
            for (int j = 0; j < JS_BATCH; ++j) {
                const int iz = blockIdx.y * JS_ROWS + ty + 4 * (k0 + j);
                if (k0 + j < JS_ROWS / 4 && iz < nz) {
                    const double j0 = FLY ? (0.5 * R0 + 0.5 * m0) * jraw_shape(v[j].x, pa, iden) : j_raw_at<HM>(v[j].x, R0, m0, pa, den, iden, pf);
                    s += j0;
                    double j1 = 0.0;
                    if (two) { j1 = FLY ? (0.5 * R1 + 0.5 * m1) * jraw_shape(v[j].y, pa, iden) : j_raw_at<HM>(v[j].y, R1, m1, pa, den, iden, pf); s += j1; }
                    if (raw) {
                        if (two) *reinterpret_cast<double2*>(raw + (long)iz * ld + c0) = double2{j0, j1};
                        else raw[(long)iz * ld + c0] = j0;
                    }
                }
            }
        }
    }
    double bs = block_sum<256>(s, sh);
    if (threadIdx.x == 0) part[(long)b * nblk + (long)blockIdx.y * gridDim.x + blockIdx.x] = bs;
}
__global__ void __launch_bounds__(256) k_jsum_final(Member* st, const double* part, int nblk, double dr,
                                                    double dz) {
    __shared__ double sh[4];
    const int b = blockIdx.x;
    // (a frozen member's sum is formed and dropped: the activity word travels with the partial sums instead of in front of them)
    jsum_final_body(b, st, part, nblk, dr, dz, sh, st[b].active != 0);
}
// Source = (-mu0*R) * (J_raw * scale)  (core/fusion_kernel.py:440, newton_solver.py:513); optionally stores J_phi.
// Column pair per thread (16-B loads / stores); grid (ceil(nr/128), ceil(nz/16), B), 256 threads
__global__ void __launch_bounds__(256) k_source(Fld srcF, Fld jF, int write_j, Fld psiF, const Member* st, int nz, int nr, int ld,
                                                const double* r_axis, const double* inv_mu0R, const double* neg_mu0R, Prof pf) {
    int b = blockIdx.z;
    if (!st[b].active) return;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + tx) * 2;
    if (c0 >= nr) return;
    const bool two = c0 + 1 < nr;
    const double pa = st[b].psi_axis, den = st[b].denom, iden = st[b].inv_denom, sc = st[b].jscale;
    const double R0 = r_axis[c0], R1 = two ? r_axis[c0 + 1] : 1.0;
    const double m0 = inv_mu0R[c0], m1 = two ? inv_mu0R[c0 + 1] : 0.0;
    const double n0 = neg_mu0R[c0], n1 = two ? neg_mu0R[c0 + 1] : 0.0;
    const double* psi = psiF.at(b);
    double* so = srcF.at(b);
    double* jo = jF.at(b);
#pragma unroll
    for (int k = 0; k < JS_ROWS / 4; ++k) {
        const int iz = blockIdx.y * JS_ROWS + ty + 4 * k;
        if (iz >= nz) continue;
        const long c = (long)iz * ld + c0;
        const double2 v = *reinterpret_cast<const double2*>(psi + c);
        const double jr0 = j_raw_at(v.x, R0, m0, pa, den, iden, pf);
        const double j0 = sc != 0.0 ? jr0 * sc : 0.0;
        if (two) {
            const double jr1 = j_raw_at(v.y, R1, m1, pa, den, iden, pf);
            const double j1 = sc != 0.0 ? jr1 * sc : 0.0;
            *reinterpret_cast<double2*>(so + c) = double2{n0 * j0, n1 * j1};
            if (write_j) *reinterpret_cast<double2*>(jo + c) = double2{j0, j1};
        } else {
            so[c] = n0 * j0;
            if (write_j) jo[c] = j0;
        }
    }
}
// Source from a stored J_phi (seed / external profile mode)
__global__ void k_source_from_j(Fld srcF, Fld jF, int nz, int nr, int ld, const double* neg_mu0R,
                                const int* active) {
    int b = blockIdx.z;
    if (active && !active[b]) return;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    long c = (long)iz * ld + ir;
    srcF.at(b)[c] = neg_mu0R[ir] * jF.at(b)[c];
}
// J_phi = seed * (I_target / I_seed) per member (iterative_solver.py:404-406)
__global__ void k_seed_j(Fld jF, const double* seed, const Member* st, double i_seed, int nz, int nr, int ld, const int* active) {
    int b = blockIdx.z;
    if (active && !active[b]) return;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    double s = seed[(long)iz * nr + ir];
    double it = st[b].i_target;
    double v;
    if (fabs(it) < 1e-12) v = 0.0;
    else v = i_seed > 0 ? s * (it / i_seed) : s;
    jF.at(b)[(long)iz * ld + ir] = v;
}

// ---------------------------------------------------------------- Picard: blend + norms (a10)
// nxt = (1-alpha)*cur + alpha*new with new's ring taken from bc; partial sum |new-cur|, non-finite flag
// (core/fusion_kernel_newton_solver.py:515-537).  grid (nblk, B), 256 x 4 points
__global__ void __launch_bounds__(256) k_blend(Fld nxtF, Fld curF, Fld newF, Fld bcF, const Member* st, int nz,
                                               int nr, int ld, double alpha, double om_alpha, double* psum,
                                               int* pbad, int nblk) {
    __shared__ double sh[4];
    __shared__ int shbad;
    int b = blockIdx.y;
    if (!st[b].active) return;
    if (threadIdx.x == 0) shbad = 0;
    __syncthreads();
    const double* cur = curF.at(b);
    const double* nw = newF.at(b);
    const double* bc = bcF.at(b);
    double* nxt = nxtF.at(b);
    const long n = (long)nz * nr;
    double s = 0.0;
    int bad = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (q < n) {
            int iz = (int)(q / nr), ir = (int)(q - (long)iz * nr);
            long c = (long)iz * ld + ir;
            bool ring = iz == 0 || iz == nz - 1 || ir == 0 || ir == nr - 1;
            double pn = ring ? bc[c] : nw[c];
            double pc = cur[c];
            if (!finite_d(pn)) bad = 1;
            s += fabs(pn - pc);
            nxt[c] = om_alpha * pc + alpha * pn;
        }
    }
    if (bad) atomicOr(&shbad, 1);
    double bs = block_sum<256>(s, sh);
    if (threadIdx.x == 0) {
        psum[(long)b * nblk + blockIdx.x] = bs;
        pbad[(long)b * nblk + blockIdx.x] = shbad;
    }
}

// Fused tail of a Picard iteration (one pass over the fine grid instead of three):
//   nxt = (1-alpha)*cur + alpha*new (new's ring from bc)            newton_solver.py:515-537
//   partial sum |new-cur|, non-finite flag                             newton_solver.py:518,535
//   partial max|r|, sum r^2 of r = L*nxt - src over the interior       fusion_kernel_solver_runtime.py:25-51
//   topology partials of nxt for the NEXT iteration (argmax psi, masked argmin |grad psi|, min psi)
// (kernel: k_blend_tile below.)  All seven workgroup reductions share two barriers: wave-level DPP reductions -> LDS -> barrier ->
// (block max, rescaled sum of squares) -> LDS -> barrier -> thread 0.
struct PicardCtl {
    double tol, gs_tol;
    int require_gs, max_iter, iter, pad;
    long n_total, n_interior;
};
// everything k_picard_final needs besides the member index
struct PicardTail {
    Member* st = nullptr;
    const double* psum = nullptr;
    const int* pbad = nullptr;
    const double* rmax = nullptr;
    const double* rsum = nullptr;
    int nblk = 0, nblk_r = 0;
    PicardCtl ctl{};
    double* hist_diff = nullptr;
    double* hist_gs = nullptr;
    double* trace = nullptr;
    int* n_active = nullptr;
    // run-ahead control of the host loop (scpn_hip.hip, picard_solve_impl): two host-mapped words, each written by ONE thread of
    // the whole launch as one aligned 8-byte system-scope store {solve id, value} -- host_prog: the number of k_picard_final launches
    // of this solve that have started (block 0, from the device counter iter_count: kernel arguments stay constant over the
    // iterations); host_done: set by the workgroup whose atomicSub takes n_active to zero (atomics on one address are totally ordered:
    // exactly one workgroup sees the old value 1).  nullptr: the blocking poll every check_every iterations is used instead.
    unsigned long long* host_prog = nullptr;
    unsigned long long* host_done = nullptr;
    int* iter_count = nullptr;
    unsigned solve_id = 0;
    int* f_cur = nullptr;
    int* f_nxt = nullptr;
    int* f_best = nullptr;
    int* f_prev = nullptr;
    int* f_active = nullptr;
    const TopoPart* topo = nullptr;  // != nullptr: also finish the axis / X-point search of the NEXT iteration (k_topology_final's work)
    int ntopo = 0;
    int nr = 0;
    const double* r_axis = nullptr;
    const double* z_axis = nullptr;
};
// one member leaves the iteration (one thread per member): count it out and, if it was the last one, tell the host
__device__ __forceinline__ void picard_member_frozen(const PicardTail& t) {
    if (atomicSub(t.n_active, 1) == 1 && t.host_done)
        __hip_atomic_store(t.host_done, ((unsigned long long)t.solve_id << 32) | 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// shared scratch of picard_final_body (declared by the calling kernel)
struct PicardShared {
    double sh[4], sh2[4], shd[4], shp[4];
    int shbad[4];
    VI shx[4], shn[4];
};
// One 256-thread workgroup finishes member b: diff / GS RMS, best / ping-pong slots, convergence
// (core/fusion_kernel_newton_solver.py:518-571) and, with t.topo, the axis / X-point of the next iterate.  The partial of index i is
// combined by thread i % 256 in ascending i, then lanes, then waves 0..3.
__device__ __forceinline__ void picard_final_body(const int b, const PicardTail& t, PicardShared& S) {
    Member* st = t.st;
    const int nblk = t.nblk, nblk_r = t.nblk_r, ntopo = t.ntopo;
    const PicardCtl& ctl = t.ctl;
    VI tmx{0.0, -1}, tmn{0.0, -1};
    double tminpsi = INFINITY, tpsi = 0.0;  // tpsi: psi at this thread's arg-min candidate
    const int m_it = st[b].iterations, m_cur = st[b].cur, m_nxt = st[b].nxt, m_best = st[b].best, m_prev = st[b].prev;
    const double m_gs_best = st[b].gs_best, m_diff_best = st[b].diff_best;
    double s = 0.0, rs = 0.0, rm = 0.0;
    int bad = 0;
    // Every partial this thread combines FIRST (index threadIdx.x of each array: all there is up to 256 tiles per member) is
    // requested here, before anything is computed -- one memory round trip for the kernel instead of one per reduction stage
    // (sum / flags, maximum, rescaled squares behind the maximum's barrier, topology candidates).  Absent entries read a constant.
    const int i0 = threadIdx.x;
    const bool in_d = i0 < nblk, in_r = i0 < nblk_r, in_t = t.topo != nullptr && i0 < ntopo;
    const double f_ps = ld_val(in_d ? t.psum + (long)b * nblk + i0 : &g_const_f64[0]);
    const int f_pb = ld_val(in_d ? t.pbad + (long)b * nblk + i0 : &g_const_i32[0]);
    const double f_rm = ld_val(in_r ? t.rmax + (long)b * nblk_r + i0 : &g_const_f64[0]);
    const double f_rs = ld_val(in_r ? t.rsum + (long)b * nblk_r + i0 : &g_const_f64[0]);
    const TopoPart* const q0 = in_t ? t.topo + (long)b * ntopo + i0 : nullptr;
    const double f_maxv = in_t ? ld_val(&q0->maxv) : 0.0, f_minb = in_t ? ld_val(&q0->minb) : 0.0;
    const long f_maxi = in_t ? ld_val(&q0->maxi) : -1L, f_mini = in_t ? ld_val(&q0->mini) : -1L;
    const double f_minpsi = in_t ? ld_val(&q0->minpsi) : INFINITY, f_pam = in_t ? ld_val(&q0->psi_at_min) : 0.0;
    s += f_ps;
    bad |= f_pb;
    for (int i = threadIdx.x + 256; i < nblk; i += 256) {
        s += ld_val(t.psum + (long)b * nblk + i);
        bad |= ld_val(t.pbad + (long)b * nblk + i);
    }
    rm = fmax(rm, f_rm);
    for (int i = threadIdx.x + 256; i < nblk_r; i += 256) rm = fmax(rm, ld_val(t.rmax + (long)b * nblk_r + i));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wbad = __any(bad) ? 1 : 0;
    if (lane == 0) S.shbad[w] = wbad;
    const double rmx_all = block_max_all<256>(rm, S.sh);  // M = max_b m_b  (its barriers also publish shbad)
    if (rmx_all > 0.0) {
        { const double f = f_rm / rmx_all; rs += f_rs * (f * f); }
        for (int i = threadIdx.x + 256; i < nblk_r; i += 256) {
            const double f = ld_val(t.rmax + (long)b * nblk_r + i) / rmx_all;
            rs += ld_val(t.rsum + (long)b * nblk_r + i) * (f * f);
        }
    }
    if (t.topo) {
        auto take = [&](const VI cmx, const VI cmn, const double cpsi, const double cminpsi) {
            tmx = vi_max(tmx, cmx);
            const VI nmn = vi_min(tmn, cmn);
            if (nmn.i != tmn.i) tpsi = cpsi;
            tmn = nmn;
            tminpsi = fmin(tminpsi, cminpsi);
        };
        if (in_t) take(VI{f_maxv, f_maxi}, VI{f_minb, f_mini}, f_pam, f_minpsi);
        for (int i = threadIdx.x + 256; i < ntopo; i += 256) {
            const TopoPart* q = t.topo + (long)b * ntopo + i;
            take(VI{ld_val(&q->maxv), ld_val(&q->maxi)}, VI{ld_val(&q->minb), ld_val(&q->mini)},
                 ld_val(&q->psi_at_min), ld_val(&q->minpsi));
        }
        const VI own = tmn;
        tmx = wave_vi<true>(tmx);
        tmn = wave_vi<false>(tmn);
        tminpsi = wave_max(-tminpsi);
        if (tmn.i >= 0 && own.i == tmn.i) S.shp[w] = tpsi;  // a grid index lives in exactly one partial: one lane per wave at most
    }
    // the sums (and the topology candidates) share one barrier (block_sum's combination order: wave totals added 0..3 by thread 0)
    s = wave_sum(s);
    rs = wave_sum(rs);
    __syncthreads();  // (sh is still being read by block_max_all above)
    if (lane == 0) { S.sh[w] = s; S.sh2[w] = rs; S.shx[w] = tmx; S.shn[w] = tmn; S.shd[w] = tminpsi; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tot = 0.0, rtot = 0.0;
    int anybad = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { tot += S.sh[i]; rtot += S.sh2[i]; anybad |= S.shbad[i]; }
    const double rmx = rmx_all;
    Member& m = st[b];
    const int it = m_it;  // iteration index lives on the device: kernel arguments stay constant
    m.iterations = it + 1;
    if (anybad) {  // diverged: revert to best, stop (newton_solver.py:518-532); the Newton warm-up keeps Psi (:163-166)
        int cur = m_cur, prev = m_prev;
        if (!ctl.pad) {
            prev = m_cur;
            cur = m_best;
            m.prev = prev;
            m.cur = cur;
        }
        m.diverged = 1;
        m.active = 0;
        picard_member_frozen(t);
        t.f_cur[b] = cur; t.f_nxt[b] = m_nxt; t.f_best[b] = m_best; t.f_prev[b] = prev; t.f_active[b] = 0;
        return;
    }
    double diff = tot / (double)ctl.n_total;
    // stable_rms (fusion_kernel_numerics.py:28-37): max * sqrt(mean((r/max)^2)), overflow-safe up to |r| = 1e250
    double gs = rmx <= 0.0 ? 0.0 : rmx * sqrt(rtot / (double)ctl.n_interior);
    m.diff = diff;
    m.gs = gs;
    if (gs < m_gs_best) m.gs_best = gs;
    const int prev = m_cur, cur = m_nxt;
    int best = m_best;
    m.prev = prev;
    m.cur = cur;
    if (diff < m_diff_best) { m.diff_best = diff; best = cur; m.best = best; }
    int nx = 0;
    for (int k = 0; k < 3; ++k) if (k != cur && k != best) { nx = k; break; }
    m.nxt = nx;
    if (t.hist_diff) t.hist_diff[(long)b * ctl.max_iter + it] = diff;
    if (t.hist_gs) t.hist_gs[(long)b * ctl.max_iter + it] = gs;
    if (t.trace) {
        double* tr = t.trace + ((long)b * ctl.max_iter + it) * 6;
        tr[0] = m.psi_axis; tr[1] = m.psi_bnd; tr[2] = (double)m.x_idx; tr[3] = m.i_current; tr[4] = diff; tr[5] = gs;
    }
    bool upd = diff < ctl.tol;
    bool gsc = (!ctl.require_gs) || (gs < ctl.gs_tol);
    int active = 1;
    if (upd && gsc) { m.converged = 1; active = 0; }
    else if (it + 1 >= ctl.max_iter) active = 0;
    if (!active) { m.active = 0; picard_member_frozen(t); }
    // publish the slot indices / activity the field kernels index (was a separate k_publish_slots launch)
    t.f_cur[b] = cur; t.f_nxt[b] = nx; t.f_best[b] = best; t.f_prev[b] = prev; t.f_active[b] = active;
    if (t.topo && active) {  // k_topology_final for the next iteration, on the new current iterate
        VI wmx = S.shx[0], wmn = S.shn[0];
        double rmp = S.shd[0], pmin = S.shp[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            wmx = vi_max(wmx, S.shx[i]);
            const VI n2 = vi_min(wmn, S.shn[i]);
            if (n2.i != wmn.i) pmin = S.shp[i];
            wmn = n2;
            rmp = fmax(rmp, S.shd[i]);
        }
        rmp = -rmp;
        double pa = wmx.i >= 0 ? wmx.v : 0.0;
        if (fabs(pa) < 1e-6) pa = 1e-6;
        double pb, xr = 0.0, xz = 0.0;
        if (wmn.i >= 0) {
            int iz = (int)(wmn.i / t.nr), ir = (int)(wmn.i - (long)iz * t.nr);
            pb = pmin;  // psi of the new iterate at the arg-min, carried with the partials
            xr = t.r_axis[ir];
            xz = t.z_axis[iz];
            if (!finite_d(pb)) pb = nan_to_num300(pb);
        } else {
            pb = rmp;
        }
        if (fabs(pa - pb) < 0.1) pb = pa * 0.1;
        double denom = pb - pa;  // core/fusion_kernel.py:415-417
        if (fabs(denom) < 1e-9) denom = 1e-9;
        m.psi_axis = pa; m.psi_bnd = pb; m.denom = denom; m.inv_denom = 1.0 / denom; m.x_idx = wmn.i; m.x_r = xr; m.x_z = xz;
    }
}
__global__ void __launch_bounds__(256) k_picard_final(PicardTail t) {
    __shared__ PicardShared S;
    const int b = blockIdx.x;
    if (b == 0 && threadIdx.x == 0 && t.host_prog) {  // progress word: this launch of the solve has started (see PicardTail)
        const int k = *t.iter_count + 1;
        *t.iter_count = k;
        __hip_atomic_store(t.host_prog, ((unsigned long long)t.solve_id << 32) | (unsigned long long)(unsigned)k, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!t.st[b].active) return;
    picard_final_body(b, t, S);
}
// reciprocals (two words) of dz, 2 dz, dr for the exact divisions of the divertor-row gradient (host IEEE division)
struct GradRecip { double dz, inv_dz, lo_dz, two_dz, inv_two_dz, lo_two_dz, dr, inv_dr, lo_dr; };

// ---------------------------------------------------------------- fused Picard tail, row-pair form (round 3)
// One launch per Picard iteration: nxt = (1-alpha) cur + alpha new with the ring from bc, sum |new - cur|, non-finite flag,
// (max, sum (r/max)^2) of r = L* nxt - Source, argmax / divertor-row argmin |grad| / min of nxt per tile.  Structured like the
// tile smoother: a 256-thread workgroup owns a 14-row x 124-column window of a 16 x 128 tile; a lane owns the column pair
// (2l, 2l+1) of its wave's four rows, so cur / new / source arrive as 16-B loads from (scalar row pointer) + (lane offset), the
// blended pair stays in registers for its own stencil and only the N/S pairs and the two out-of-pair neighbours are read from
// the LDS tile (five LDS reads per two points).  (Round 2's form -- one element per thread, 32 x 64 tiles -- ran at the same speed with
// the same instruction count: the kernel is bound by tiles in flight x tile lifetime, profiles/r03_experiments.md.)
#ifndef BL_WZ_N
#define BL_WZ_N 14  // window rows (tile rows = + 2, a multiple of the four waves): 14 fits 80 VGPRs, 18 needs 81
#endif
#ifndef BL_WPE
#define BL_WPE 6  // minimum waves per SIMD the register allocation must allow (a workgroup puts one wave on each SIMD): six workgroups per CU
#endif
constexpr int BL_WZ = BL_WZ_N, BL_LB = BL_WZ + 2, BL_WR = 124, BL_NW = 4, BL_NROWS = BL_LB / BL_NW;
static_assert(BL_LB % BL_NW == 0, "rows are dealt to the four waves evenly");
__device__ __forceinline__ int blend_tiles_x(int nr) { return (nr + BL_WR - 1) / BL_WR; }
// Round 4: the per-point bookkeeping taken out of the point loop (round 3's form: profiles/r04_negative/blend_tile_round3.hip.h.txt).  What a LANE accumulates is kept per COLUMN
// of its pair (sum |new - cur|, min, running max + the row it came from, max |r|), so the lane's column predicates (window lane,
// inside the grid, interior column) are applied ONCE per lane after the rows instead of as selects per point; the non-finite
// check of the elliptic solve's output is read off the partial sum of |new - cur|.  CONTRACT of that verdict: a non-finite `new`
// ALWAYS makes its owner's sum non-finite, whatever `cur` holds (inf - inf = NaN, inf - x = inf), so nothing the reference's
// `isnan(Psi_new).any() or isinf(Psi_new).any()` flags is missed; the verdict is a superset only where `new` is finite and either
// `cur` is non-finite or the difference overflows (|new - cur| > 1.8e308).  Neither occurs on the reference's paths: `cur` is an
// accepted iterate, and a preserved start is sanitised in its interior before the first iteration (_prepare_initial_flux) while a
// non-finite RING value makes `new` non-finite too (the ring is Dirichlet data of the elliptic solve) --
// tests/test_gpu_parity.py::test_non_finite_entry_in_a_preserved_start.  The min / arg-max candidates take -inf as a minimum and
// never as a maximum, as np.min / np.argmax order it.  The arg-max candidate travels as (value, row tag) and becomes a grid index once per
// lane; the (value, index) wave reductions are a value reduction plus a 32-bit index reduction among the lanes that hold the
// extreme value; the residual is sanitised on a cold path (a wave that sees |r| > CAP or a NaN -- never in a healthy run) and its
// sum of squares is scaled by a power of two taken from the WAVE's own largest |r| (exact scaling, no division, no workgroup
// barrier between the maximum and the squares).  Outputs are those of round 3's kernel to rounding:
// rmax is the power of two the tile's sum is relative to.
__device__ __forceinline__ double pow2_floor_or_zero(double m) {  // 2^floor(log2 m) for finite m > 0 (subnormals: 2^-1022), 0 for m == 0
    const unsigned long b = (unsigned long)__double_as_longlong(m);
    const unsigned long ex = (b >> 52) & 0x7ffUL;
    return m > 0.0 ? __longlong_as_double((long long)((ex ? ex : 1UL) << 52)) : 0.0;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {  // result in every lane
#define SCPN_U32_STEP(CTRL, RM) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, RM, 0xf, false));
    SCPN_U32_STEP(0x111, 0xf) SCPN_U32_STEP(0x112, 0xf) SCPN_U32_STEP(0x114, 0xf) SCPN_U32_STEP(0x118, 0xf) SCPN_U32_STEP(0x142, 0xa) SCPN_U32_STEP(0x143, 0xc)
#undef SCPN_U32_STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// FAST (contracted arithmetic, see k_smooth_tile): the blend is one product + one FMA, and the residual is the stencil form
// a_E E + a_W W + a_NS (N + S) - a_C psi - Source as one FMA chain (six operations per point instead of 31; the waves fetch
// a_e / a_w of the lane's columns where the exact form fetches r_safe and its reciprocal).  The divertor-row gradient keeps the
// exact quotients: it decides a discrete quantity (the arg-min of |grad psi|).
template <bool FAST = false>
__global__ void __launch_bounds__(64 * BL_NW, BL_WPE) k_blend_tile(Fld nxtF, Fld curF, Fld newF, Fld bcF, Fld srcF, const Member* st, Lvl L,
                                                             double alpha, double om_alpha, int n_mask_rows, double* psum, int* pbad,
                                                             double* rmax, double* rsum, TopoPart* topo, int nblk, const double* lazy_nm,
                                                             GradRecip gq, int xcd_remap, const double* jfly_r, const double* jfly_m) {
    __shared__ __attribute__((aligned(16))) double t[BL_LB][128];
    __shared__ VI sh_mx[BL_NW], sh_mn[BL_NW];
    __shared__ double sh_dsum[BL_NW], sh_p2[BL_NW], sh_minp[BL_NW], sh_s2[BL_NW];
    __shared__ double sh_coef[2 * BL_NW][64];  // r_safe, 1/r_safe (two words), -mu0 R of the lane's two columns: loaded by one wave each
    int bx_, by_, b;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        if (xcd_remap) lin = xcd_contiguous(lin, total);
        bx_ = (int)(lin % gx);
        const unsigned t_ = lin / gx;
        by_ = (int)(t_ % gy);
        b = (int)(t_ / gy);
    }
    // the member's words -- activity, the slot of every field, the lazy-source scale -- in ONE round trip (see k_smooth_tile)
    const int act_w = st[b].active;
    const int k_cur = *curF.slot_ptr(b), k_new = *newF.slot_ptr(b), k_bc = *bcF.slot_ptr(b), k_src = *srcF.slot_ptr(b), k_nxt = *nxtF.slot_ptr(b);
    const double lz_sc = *(lazy_nm ? &st[b].jscale : &g_const_f64[1]);
    // source on the fly (FAST; jfly_r = r_axis, jfly_m = 1 / (mu0 R); uniform): the source is rebuilt from `cur`, which this kernel loads anyway
    const bool jfly = FAST && lazy_nm != nullptr && jfly_r != nullptr;
    const double jf_pa = *(jfly ? &st[b].psi_axis : &g_const_f64[0]), jf_id = *(jfly ? &st[b].inv_denom : &g_const_f64[1]);
    if (!act_w) return;
    BLEND_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nz = L.nz, nr = L.nr, ld = L.ld;
    // tile row lz <-> grid row z0 + lz, tile column lc <-> grid column r0 + lc; owned: lz in [1, BL_WZ], lc in [2, 125]
    const int z0 = by_ * BL_WZ - 1, r0 = bx_ * BL_WR - 2;  // r0 even: the lane's pair is 16-B aligned
    const int c0 = r0 + 2 * lane;
    const bool pair_win = lane >= 1 && lane <= 62;
    const bool own_x = pair_win && c0 < nr, own_y = pair_win && c0 + 1 < nr;  // (c0 >= 0 for window lanes)
    const char* const cur_b = reinterpret_cast<const char*>(curF.at_slot(b, k_cur));
    const char* const new_b = reinterpret_cast<const char*>(newF.at_slot(b, k_new));
    const char* const bc_b = reinterpret_cast<const char*>(bcF.at_slot(b, k_bc));
    const char* const src_b = reinterpret_cast<const char*>(srcF.at_slot(b, k_src));
    char* const nxt_b = reinterpret_cast<char*>(nxtF.at_slot(b, k_nxt));
    const unsigned ld8 = (unsigned)ld * 8u;
    const unsigned lane_off = (unsigned)min(max(c0, 0), ld - 2) * 8u;
    // ring columns (grid column 0 / nr-1) take their value from bc; only tiles that touch them look (wave-uniform)
    const bool edge_cols = r0 <= 0 || r0 + 127 >= nr - 1;
    const bool cring_x = c0 == 0 || c0 == nr - 1, cring_y = c0 + 1 == nr - 1;
    double2 pc[BL_NROWS], pn[BL_NROWS], sv[BL_NROWS];
#pragma unroll
    for (int i = 0; i < BL_NROWS; ++i) {
        const int gzc = min(max(z0 + w + BL_NW * i, 0), nz - 1);  // scalar
        const unsigned vo = (unsigned)gzc * ld8 + lane_off;
        const bool rowring = gzc == 0 || gzc == nz - 1;
        pc[i] = *reinterpret_cast<const double2*>(cur_b + vo);
        pn[i] = *reinterpret_cast<const double2*>((rowring ? bc_b : new_b) + vo);
        if (!jfly) sv[i] = *reinterpret_cast<const double2*>(src_b + vo);
    }
    if (edge_cols) {
#pragma unroll
        for (int i = 0; i < BL_NROWS; ++i) {
            const int gzc = min(max(z0 + w + BL_NW * i, 0), nz - 1);
            const unsigned vo = (unsigned)gzc * ld8 + lane_off;
            if (cring_x) pn[i].x = *reinterpret_cast<const double*>(bc_b + vo);
            if (cring_y) pn[i].y = *reinterpret_cast<const double*>(bc_b + vo + 8u);
        }
    }
    // column coefficients of the residual stage: wave k fetches the k-th of (r_safe, 1/r_safe, its low word, -mu0 R) for the lane's
    // two columns now, behind the row loads, and hands them to the other waves through LDS at the barrier below
    const bool int_x = own_x && c0 >= 1 && c0 <= nr - 2, int_y = own_y && c0 + 1 <= nr - 2;  // (c0 + 1 >= 1 always)
    {
        const double* const arr = FAST ? (w == 0 ? L.a_e : (w == 1 ? L.a_w : (w == 2 ? (jfly ? jfly_r : nullptr) : lazy_nm)))
                                       : (w == 0 ? L.r_safe : (w == 1 ? L.r_inv : (w == 2 ? L.r_inv_lo : lazy_nm)));
        // r_safe, 1/r_safe: 1.0; low word: 0.0; -mu0 R: 0.0 outside the interior, 1.0 when the source field already holds the source
        const double* const dflt = &g_const_f64[(w <= 1 || (w == 3 && !lazy_nm)) ? 1 : 0];
        const bool have = arr != nullptr;
        double cx = *((have && int_x) ? arr + c0 : dflt), cy = *((have && int_y) ? arr + c0 + 1 : dflt);
        if (jfly && w == 2) {  // the column factor of J_raw: 0.5 R + 0.5 / (mu0 R)
            cx = int_x ? 0.5 * cx + 0.5 * jfly_m[c0] : 0.0;
            cy = int_y ? 0.5 * cy + 0.5 * jfly_m[c0 + 1] : 0.0;
        }
        sh_coef[2 * w][lane] = cx;
        sh_coef[2 * w + 1][lane] = cy;
    }
    // ---- stage 1: blend, store, stage the tile; per-COLUMN accumulators, no per-point predicates (rows: wave-uniform tests)
    double dsx = 0.0, dsy = 0.0, mnx = INFINITY, mny = INFINITY, bvx = -INFINITY, bvy = -INFINITY;
    int btx = -1, bty = -1;  // row slot i of the running maximum (ascending i = ascending grid row: `>` keeps the first occurrence)
#pragma unroll
    for (int i = 0; i < BL_NROWS; ++i) {
        const int lz = w + BL_NW * i, gz = z0 + lz;
        double2 vi;
        if constexpr (FAST) {
            vi.x = __builtin_fma(alpha, pn[i].x, om_alpha * pc[i].x);
            vi.y = __builtin_fma(alpha, pn[i].y, om_alpha * pc[i].y);
        } else {
            vi.x = om_alpha * pc[i].x + alpha * pn[i].x;
            vi.y = om_alpha * pc[i].y + alpha * pn[i].y;
        }
        if (lz >= 1 && lz <= BL_WZ && gz < nz) {  // owned row (wave-uniform)
            dsx += fabs(pn[i].x - pc[i].x);
            dsy += fabs(pn[i].y - pc[i].y);
            mnx = fmin(mnx, vi.x);
            mny = fmin(mny, vi.y);
            const bool gx = vi.x > bvx, gy = vi.y > bvy;  // (a NaN never wins)
            bvx = gx ? vi.x : bvx; btx = gx ? i : btx;
            bvy = gy ? vi.y : bvy; bty = gy ? i : bty;
            char* const o = nxt_b + ((unsigned)gz * ld8 + (unsigned)c0 * 8u);
            if (own_y) *reinterpret_cast<double2*>(o) = vi;
            else if (own_x) *reinterpret_cast<double*>(o) = vi.x;
        }
        *reinterpret_cast<double2*>(&t[lz][2 * lane]) = vi;
    }
    BLEND_STAMP(1);
    __syncthreads();
    BLEND_STAMP(2);
    // ---- stage 2: residual r = L* nxt - Source of the owned interior rows: own pair, N / S pairs and the two out-of-pair neighbours
    // from the tile; lanes whose column is not an interior grid column compute on clamped copies and are dropped after the rows
    const double rs0 = sh_coef[0][lane], rs1 = sh_coef[1][lane], ri0 = sh_coef[2][lane], ri1 = sh_coef[3][lane];
    const double rl0 = sh_coef[4][lane], rl1 = sh_coef[5][lane], nm0 = sh_coef[6][lane], nm1 = sh_coef[7][lane];
    const double fnm0 = jfly ? (nm0 * lz_sc) * rl0 : nm0 * lz_sc, fnm1 = jfly ? (nm1 * lz_sc) * rl1 : nm1 * lz_sc;  // (jfly: rl = the column factor)
    double2 rv[BL_NROWS];
#pragma unroll
    for (int i = 0; i < BL_NROWS; ++i) {
        const int lz = w + BL_NW * i, gz = z0 + lz;
        rv[i] = double2{0.0, 0.0};
        if (!(lz >= 1 && lz <= BL_WZ && gz >= 1 && gz <= nz - 2)) continue;  // wave-uniform
        const double2 N = *reinterpret_cast<const double2*>(&t[lz + 1][2 * lane]);
        const double2 S = *reinterpret_cast<const double2*>(&t[lz - 1][2 * lane]);
        const double Wn = (&t[lz][0])[2 * lane - 1], En = (&t[lz][0])[2 * lane + 2];
        const double2 P = *reinterpret_cast<const double2*>(&t[lz][2 * lane]);
        const double px = P.x, py = P.y;
        // Source = (-mu0 R) * (J_raw * scale) when the source field holds J_raw (lazy source, see k_smooth_tile); otherwise the
        // factors are 1.0 (exact identity).  Branch- and select-free: for scale == 0 the product is +-0 where k_source writes
        // (-mu0 R) * 0.0 -- the same |r| (only the sign of a zero source can differ, for finite J_raw).
        if constexpr (FAST) {
            // (rs0 / ri0 hold a_e / a_w of the even column, rs1 / ri1 of the odd one; fnm = (-mu0 R) * scale)
            const double nsx = N.x + S.x, nsy = N.y + S.y;
            const double jx = jfly ? jraw_shape(pc[i].x, jf_pa, jf_id) : sv[i].x, jy = jfly ? jraw_shape(pc[i].y, jf_pa, jf_id) : sv[i].y;
            rv[i].x = __builtin_fma(rs0, py, __builtin_fma(ri0, Wn, __builtin_fma(L.a_ns, nsx, __builtin_fma(-fnm0, jx, -L.a_c * px))));
            rv[i].y = __builtin_fma(rs1, En, __builtin_fma(ri1, px, __builtin_fma(L.a_ns, nsy, __builtin_fma(-fnm1, jy, -L.a_c * py))));
            continue;
        }
        const double sx = nm0 * (sv[i].x * lz_sc), sy = nm1 * (sv[i].y * lz_sc);
        const double d2rx = divc4(py - 2.0 * px + Wn, L.dr2, L.inv_dr2, L.lo_dr2);
        const double d2ry = divc4(En - 2.0 * py + px, L.dr2, L.inv_dr2, L.lo_dr2);
        const double d1rx = divc4(py - Wn, L.two_dr, L.inv_two_dr, L.lo_two_dr);
        const double d1ry = divc4(En - px, L.two_dr, L.inv_two_dr, L.lo_two_dr);
        const double d2zx = divc4(N.x - 2.0 * px + S.x, L.dz2, L.inv_dz2, L.lo_dz2);
        const double d2zy = divc4(N.y - 2.0 * py + S.y, L.dz2, L.inv_dz2, L.lo_dz2);
        rv[i].x = (d2rx - divc4(d1rx, rs0, ri0, rl0) + d2zx) - sx;
        rv[i].y = (d2ry - divc4(d1ry, rs1, ri1, rl1) + d2zy) - sy;
    }
    // largest |r| of the lane's interior columns; a NaN is invisible to fmax and shows up in the squares below
    auto col_max = [&](bool on_x, bool on_y) {
        double ax = 0.0, ay = 0.0;
#pragma unroll
        for (int i = 0; i < BL_NROWS; ++i) { ax = fmax(ax, fabs(rv[i].x)); ay = fmax(ay, fabs(rv[i].y)); }
        return fmax(on_x ? ax : 0.0, on_y ? ay : 0.0);
    };
    double w_m = wave_max(col_max(int_x, int_y));
    double p2 = pow2_floor_or_zero(w_m);  // (wave-uniform) the power of two the wave's squares are relative to
    auto sum_sq = [&](double inv_p2) {
        double qx = 0.0, qy = 0.0;
#pragma unroll
        for (int i = 0; i < BL_NROWS; ++i) {
            const double ux = rv[i].x * inv_p2, uy = rv[i].y * inv_p2;  // exact (power of two), |u| < 2
            qx += ux * ux;
            qy += uy * uy;
        }
        return (int_x ? qx : 0.0) + (int_y ? qy : 0.0);
    };
    double s2 = p2 > 0.0 ? sum_sq(1.0 / p2) : 0.0;  // (1 / 2^k is exact; the branch is wave-uniform)
    if (__any(!(w_m <= CAP) || s2 != s2)) {
        // cold path: sanitise as the reference does (fusion_kernel_numerics.py:19-25: NaN -> 0, clip to +-CAP) and redo the norms
#pragma unroll
        for (int i = 0; i < BL_NROWS; ++i) { rv[i].x = sanitize_sel(rv[i].x); rv[i].y = sanitize_sel(rv[i].y); }
        w_m = wave_max(col_max(int_x, int_y));
        p2 = pow2_floor_or_zero(w_m);
        s2 = p2 > 0.0 ? sum_sq(1.0 / p2) : 0.0;
    }
    BLEND_STAMP(3);
    // ---- wave-level reductions (results in every lane), one LDS hand-over per wave
    const double w_dsum = wave_sum((own_x ? dsx : 0.0) + (own_y ? dsy : 0.0));
    const double w_minp = -wave_max(-fmin(own_x ? mnx : INFINITY, own_y ? mny : INFINITY));
    const double w_s2 = wave_sum(s2);
    // arg-max candidate of the wave: the largest value, then the smallest flattened index among the lanes that hold it
    // (local index = tile row * 128 + tile column: ascending in the grid index within a tile)
    const bool hx = own_x && btx >= 0, hy = own_y && bty >= 0;
    const bool y_wins = hy && (!hx || bvy > bvx || (bvy == bvx && bty < btx));  // (same row: x comes first)
    const double lv = y_wins ? bvy : (hx ? bvx : -INFINITY);
    const unsigned lloc = (hx || hy) ? (unsigned)((w + BL_NW * (y_wins ? bty : btx)) * 128 + 2 * lane + (y_wins ? 1 : 0)) : 0xffffffffu;
    const double w_bv = wave_max(lv);
    const unsigned w_loc = wave_min_u32((lv == w_bv) ? lloc : 0xffffffffu);
    VI w_mx{0.0, -1};
    if (w_loc != 0xffffffffu) { w_mx.v = w_bv; w_mx.i = (long)(z0 + (int)(w_loc >> 7)) * nr + (r0 + (int)(w_loc & 127u)); }
    // Divertor rows (a quarter of the tiles): np.gradient + hypot + masked arg-min (fusion_kernel.py:276-288), from the LDS tile
    VI w_mn{0.0, -1};
    if (z0 + 1 < n_mask_rows) {  // (uniform: the tile owns at least one masked row)
        double gvx = INFINITY, gvy = INFINITY;
        int gtx = -1, gty = -1;
#pragma unroll
        for (int i = 0; i < BL_NROWS; ++i) {
            const int lz = w + BL_NW * i, gz = z0 + lz;
            if (!(lz >= 1 && lz <= BL_WZ && gz < nz) || gz >= n_mask_rows) continue;  // wave-uniform
            const double2 N = *reinterpret_cast<const double2*>(&t[lz + 1][2 * lane]);
            const double2 S = *reinterpret_cast<const double2*>(&t[lz - 1][2 * lane]);
            const double Wn = (&t[lz][0])[2 * lane - 1], En = (&t[lz][0])[2 * lane + 2];
            const double2 P = *reinterpret_cast<const double2*>(&t[lz][2 * lane]);
            const double px = P.x, py = P.y;
            auto grad = [&](double p, double n_, double s_, double e_, double w_, int gc) {
                if constexpr (FAST) {
                    // contracted: reciprocal multiplies and the SQUARED magnitude (the arg-min of |grad psi|^2 is the arg-min of |grad psi|;
                    // a non-finite operand makes the value non-finite, which never wins -- the reference's nan_to_num only matters in a
                    // state the non-finite check has already flagged)
                    double gzv, grv;
                    if (gz == 0) gzv = (n_ - p) * gq.inv_dz;  // (row tests are wave-uniform)
                    else if (gz == nz - 1) gzv = (p - s_) * gq.inv_dz;
                    else gzv = (n_ - s_) * gq.inv_two_dz;
                    grv = (e_ - w_) * L.inv_two_dr;
                    if (edge_cols) {
                        const double g0 = (e_ - p) * gq.inv_dr, g1 = (p - w_) * gq.inv_dr;
                        grv = gc == 0 ? g0 : (gc == nr - 1 ? g1 : grv);
                    }
                    return __builtin_fma(grv, grv, gzv * gzv);
                }
                const double ps = nan_to_num300(p), nn = nan_to_num300(n_), ss = nan_to_num300(s_), ee = nan_to_num300(e_), ww = nan_to_num300(w_);
                double gzv, grv;
                if (gz == 0) gzv = divc4(nn - ps, gq.dz, gq.inv_dz, gq.lo_dz);  // (row tests are wave-uniform)
                else if (gz == nz - 1) gzv = divc4(ps - ss, gq.dz, gq.inv_dz, gq.lo_dz);
                else gzv = divc4(nn - ss, gq.two_dz, gq.inv_two_dz, gq.lo_two_dz);
                grv = divc4(ee - ww, L.two_dr, L.inv_two_dr, L.lo_two_dr);
                if (edge_cols) {  // wave-uniform: only tiles that touch grid column 0 / nr-1 evaluate the one-sided forms
                    const double g0 = divc4(ee - ps, gq.dr, gq.inv_dr, gq.lo_dr), g1 = divc4(ps - ww, gq.dr, gq.inv_dr, gq.lo_dr);
                    grv = gc == 0 ? g0 : (gc == nr - 1 ? g1 : grv);
                }
                return hypot(grv, gzv);
            };
            const double bmx = grad(px, N.x, S.x, py, Wn, c0), bmy = grad(py, N.y, S.y, En, px, c0 + 1);
            const bool lx = bmx < gvx, ly = bmy < gvy;  // (a NaN / an overflowed hypot never wins: +inf < +inf is false)
            gvx = lx ? bmx : gvx; gtx = lx ? i : gtx;
            gvy = ly ? bmy : gvy; gty = ly ? i : gty;
        }
        const bool kx = own_x && gtx >= 0 && finite_d(gvx), ky = own_y && gty >= 0 && finite_d(gvy);
        const bool yw = ky && (!kx || gvy < gvx || (gvy == gvx && gty < gtx));
        const double gl = yw ? gvy : (kx ? gvx : INFINITY);
        const unsigned gloc = (kx || ky) ? (unsigned)((w + BL_NW * (yw ? gty : gtx)) * 128 + 2 * lane + (yw ? 1 : 0)) : 0xffffffffu;
        const double w_gv = -wave_max(-gl);
        const unsigned w_gl = wave_min_u32((gl == w_gv) ? gloc : 0xffffffffu);
        if (w_gl != 0xffffffffu) { w_mn.v = w_gv; w_mn.i = (long)(z0 + (int)(w_gl >> 7)) * nr + (r0 + (int)(w_gl & 127u)); }
    }
    if (lane == 0) { sh_dsum[w] = w_dsum; sh_p2[w] = p2; sh_s2[w] = w_s2; sh_minp[w] = w_minp; sh_mx[w] = w_mx; sh_mn[w] = w_mn; }
    __syncthreads();
    BLEND_STAMP(4);
    const long o = (long)b * nblk + (long)by_ * gridDim.x + bx_;
    double bsum = 0.0, bs2 = 0.0, bp2 = 0.0, bminp = INFINITY, pmin = 0.0;
    VI rmx{0.0, -1}, rmn{0.0, -1};
    if (tid == 0) {
        bsum = sh_dsum[0] + sh_dsum[1] + sh_dsum[2] + sh_dsum[3];
        bp2 = fmax(fmax(sh_p2[0], sh_p2[1]), fmax(sh_p2[2], sh_p2[3]));
        if (bp2 > 0.0) {
            const double ip = 1.0 / bp2;  // exact
#pragma unroll
            for (int i = 0; i < BL_NW; ++i) { const double f = sh_p2[i] * ip; bs2 += sh_s2[i] * (f * f); }  // exact rescaling
        }
        bminp = fmin(fmin(sh_minp[0], sh_minp[1]), fmin(sh_minp[2], sh_minp[3]));
        rmx = sh_mx[0]; rmn = sh_mn[0];
#pragma unroll
        for (int i = 1; i < BL_NW; ++i) { rmx = vi_max(rmx, sh_mx[i]); rmn = vi_min(rmn, sh_mn[i]); }
        // psi of the blended iterate at this tile's arg-min candidate (still in the LDS tile)
        if (rmn.i >= 0) {
            const int iz = (int)(rmn.i / nr), ir = (int)(rmn.i - (long)iz * nr);
            pmin = t[iz - z0][ir - r0];
        }
    }
    BLEND_STAMP(5);
    if (tid == 0) {
        psum[o] = bsum;
        pbad[o] = finite_d(bsum) ? 0 : 1;  // (a non-finite psi_new somewhere in the tile: see the header)
        rmax[o] = bp2;
        rsum[o] = bs2;
        TopoPart tp;
        tp.maxv = rmx.v; tp.maxi = rmx.i; tp.minb = rmn.v; tp.mini = rmn.i; tp.minpsi = bminp; tp.psi_at_min = pmin;
        topo[o] = tp;
    }
}
// ---------------------------------------------------------------- Anderson mixing (a10, method "anderson")
// core/fusion_kernel_newton_solver.py:540-550 and core/fusion_kernel_iterative_solver.py:248-314
constexpr int AND_MAX = 8;  // max mixing depth m
struct AndFields {
    const double* f[AND_MAX];  // base pointers of the last mk history fields (oldest first), member stride = stride
    long stride;
    int mk;
};
// psi_hist <- blended psi; res_hist <- Psi_new - psi (Psi_new carries the Dirichlet ring of bc)
__global__ void k_and_push(double* psi_h, double* res_h, long hstride, Fld nxtF, Fld newF, Fld bcF, int nz, int nr, int ld,
                           const int* active) {
    int b = blockIdx.z;
    if (active && !active[b]) return;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    long c = (long)iz * ld + ir;
    bool ring = iz == 0 || iz == nz - 1 || ir == 0 || ir == nr - 1;
    double pn = ring ? bcF.at(b)[c] : newF.at(b)[c];
    double p = nxtF.at(b)[c];
    psi_h[(long)b * hstride + c] = p;
    res_h[(long)b * hstride + c] = pn - p;
}
// partial sums of G[a][b] = sum dF_a dF_b and v[a] = sum dF_a F_last, dF_a = F_{a+1} - F_a; part[b][blk][20]
__global__ void __launch_bounds__(256) k_and_gram(AndFields F, int nz, int nr, int ld, double* part, int nblk, const int* active) {
    __shared__ double sh[4];
    int b = blockIdx.y;
    if (active && !active[b]) return;
    const int q = F.mk - 1;
    double acc[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) acc[i] = 0.0;
    const long n = (long)nz * nr;
    for (int k = 0; k < 4; ++k) {
        long p = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (p < n) {
            int iz = (int)(p / nr), ir = (int)(p - (long)iz * nr);
            long c = (long)b * F.stride + (long)iz * ld + ir;
            double fv[AND_MAX], d[AND_MAX];
#pragma unroll
            for (int j = 0; j < AND_MAX; ++j) fv[j] = j < F.mk ? F.f[j][c] : 0.0;
#pragma unroll
            for (int j = 0; j < AND_MAX - 1; ++j) d[j] = fv[j + 1] - fv[j];
            const double last = fv[F.mk - 1];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2)
                    if (a < q && c2 < q) acc[a * 4 + c2] += d[a] * d[c2];
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < q) acc[16 + a] += d[a] * last;
        }
    }
    for (int i = 0; i < 20; ++i) {
        double r = block_sum<256>(acc[i], sh);
        if (threadIdx.x == 0) part[((long)b * nblk + blockIdx.x) * 20 + i] = r;
    }
}
// one thread per member: finish the sums, regularise, solve (LU, partial pivoting), mixing weights alpha[mk] (or fallback)
__global__ void __launch_bounds__(64) k_and_final(const double* part, int nblk, int mk, double* alpha_out, int* fallback, int B, const int* active) {
    // the member's 4x4 system lives in LDS: run-time indexed private arrays would go to scratch memory
    __shared__ double sG[64][4][4], sv[64][4], sgamma[64][4], salpha[64][AND_MAX];
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (active && !active[b]) return;
    const int q = mk - 1;
    double (*G)[4] = sG[threadIdx.x];
    double* v = sv[threadIdx.x];
    double* gamma = sgamma[threadIdx.x];
    double* alpha = salpha[threadIdx.x];
    for (int a = 0; a < 4; ++a) { v[a] = 0.0; gamma[a] = 0.0; for (int c = 0; c < 4; ++c) G[a][c] = 0.0; }
    for (int i = 0; i < nblk; ++i) {
        const double* p = part + ((long)b * nblk + i) * 20;
        for (int a = 0; a < q; ++a) { for (int c = 0; c < q; ++c) G[a][c] += p[a * 4 + c]; v[a] += p[16 + a]; }
    }
    for (int a = 0; a < q; ++a) G[a][a] += 1e-10;
    bool sing = false;
    for (int c = 0; c < q && !sing; ++c) {
        int piv = c;
        for (int r = c + 1; r < q; ++r) if (fabs(G[r][c]) > fabs(G[piv][c])) piv = r;
        if (G[piv][c] == 0.0 || !(G[piv][c] == G[piv][c])) { sing = true; break; }
        if (piv != c) { for (int k = 0; k < q; ++k) { double t = G[c][k]; G[c][k] = G[piv][k]; G[piv][k] = t; } double t = v[c]; v[c] = v[piv]; v[piv] = t; }
        for (int r = c + 1; r < q; ++r) {
            double f = G[r][c] / G[c][c];
            for (int k = c; k < q; ++k) G[r][k] -= f * G[c][k];
            v[r] -= f * v[c];
        }
    }
    int fb = sing ? 1 : 0;
    if (!sing) {
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
        if (fabs(asum) < 1e-12 || !(asum == asum)) fb = 1;
        else for (int a = 0; a < mk; ++a) alpha_out[(long)b * AND_MAX + a] = alpha[a] / asum;
    }
    fallback[b] = fb;
}
// psi <- sum alpha_j psi_hist_j (or the latest iterate on fallback), Dirichlet ring from bc
__global__ void k_and_mix(Fld nxtF, AndFields P, const double* alpha, const int* fallback, Fld bcF, int nz, int nr, int ld,
                          const int* active) {
    int b = blockIdx.z;
    if (active && !active[b]) return;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    long c = (long)iz * ld + ir;
    bool ring = iz == 0 || iz == nz - 1 || ir == 0 || ir == nr - 1;
    double v;
    if (ring) v = bcF.at(b)[c];
    else if (fallback[b]) v = P.f[P.mk - 1][(long)b * P.stride + c];
    else {
        v = 0.0;
        for (int j = 0; j < P.mk; ++j) v += alpha[(long)b * AND_MAX + j] * P.f[j][(long)b * P.stride + c];
    }
    nxtF.at(b)[c] = v;
}

// copies the per-member slot indices into flat int arrays the kernels index (cur/nxt/best/prev/active)
__global__ void k_publish_slots(const Member* st, int* cur, int* nxt, int* best, int* prev, int* active, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    cur[b] = st[b].cur; nxt[b] = st[b].nxt; best[b] = st[b].best; prev[b] = st[b].prev; active[b] = st[b].active;
}

// ---------------------------------------------------------------- linear multigrid_solve control (a7)
struct MgMember {
    double residual;
    int n_cycles, converged, active, pad;
};
__global__ void __launch_bounds__(256) k_mg_final(MgMember* st, const double* rmax, int nblk, double tol,
                                                  int max_cycles, int count_cycle, int* active, int* n_active) {
    __shared__ double sh[4];
    int b = blockIdx.x;
    if (!st[b].active) return;
    double rm = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) rm = fmax(rm, rmax[(long)b * nblk + i]);
    double r = block_max<256>(rm, sh);
    if (threadIdx.x == 0) {
        MgMember& m = st[b];
        if (count_cycle) m.n_cycles += 1;
        m.residual = r;
        m.converged = r < tol;
        if (m.converged || m.n_cycles >= max_cycles) { m.active = 0; active[b] = 0; atomicSub(n_active, 1); }
    }
}

// max |a-b| over the field (free-boundary outer diff, fusion_kernel_free_boundary.py:708); grid (nblk,B)
__global__ void __launch_bounds__(256) k_maxabsdiff(Fld aF, Fld bF, int nz, int nr, int ld, double* pmax, int nblk) {
    __shared__ double sh[4];
    int b = blockIdx.y;
    const double* a = aF.at(b);
    const double* bb = bF.at(b);
    const long n = (long)nz * nr;
    double m = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (q < n) {
            int iz = (int)(q / nr), ir = (int)(q - (long)iz * nr);
            long c = (long)iz * ld + ir;
            m = fmax(m, fabs(a[c] - bb[c]));
        }
    }
    double bm = block_max<256>(m, sh);
    if (threadIdx.x == 0) pmax[(long)b * nblk + blockIdx.x] = bm;
}
// per-workgroup partial sums of old^2 and (cur - old)^2 of member blockIdx.y: the two norms of the transport coupling's relative
// flux change ||psi - psi_old|| / ||psi_old|| (core/integrated_transport_solver_coupling.py:131-146) without a download of the
// fields.  part: [B][nblk][2]; the host adds the partials of a member in index order.
__global__ void __launch_bounds__(256) k_flux_change(Fld curF, Fld oldF, int nz, int nr, int ld, double* part, int nblk) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const double* a = curF.at(b);
    const double* o = oldF.at(b);
    const long n = (long)nz * nr;
    double so = 0.0, sd = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (q < n) {
            const int iz = (int)(q / nr), ir = (int)(q - (long)iz * nr);
            const long c = (long)iz * ld + ir;
            const double vo = o[c], d = a[c] - vo;
            so += vo * vo;
            sd += d * d;
        }
    }
    const double bo = block_sum<256>(so, sh);
    const double bd = block_sum<256>(sd, sh);
    if (threadIdx.x == 0) {
        part[((long)b * nblk + blockIdx.x) * 2] = bo;
        part[((long)b * nblk + blockIdx.x) * 2 + 1] = bd;
    }
}
__global__ void __launch_bounds__(256) k_max_final(double* out, const double* pmax, int nblk) {
    __shared__ double sh[4];
    int b = blockIdx.x;
    double m = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) m = fmax(m, pmax[(long)b * nblk + i]);
    double r = block_max<256>(m, sh);
    if (threadIdx.x == 0) out[b] = r;
}
// polyglot lane: psi_axis = max over the interior (jax_gs_solver.py:99); grid (nblk,B)
__global__ void __launch_bounds__(256) k_max_interior(Fld aF, int nz, int nr, int ld, double* pmax, int nblk) {
    __shared__ double sh[4];
    int b = blockIdx.y;
    const double* a = aF.at(b);
    const int ni = nr - 2;
    const long n = (long)(nz - 2) * ni;
    double m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (q < n) {
            int iz = 1 + (int)(q / ni), ir = 1 + (int)(q - (long)(iz - 1) * ni);
            m = fmax(m, a[(long)iz * ld + ir]);
        }
    }
    double bm = -0.0;
    bm = block_max<256>(m, sh);
    if (threadIdx.x == 0) pmax[(long)b * nblk + blockIdx.x] = bm;
}

// ---------------------------------------------------------------- coil Green's functions (a11)
__device__ __forceinline__ void ellip_ke(double m, double& K, double& E) {
    // arithmetic-geometric mean, A&S 17.6.3-4 (scipy.special.ellipk/ellipe, parameter m)
    double a = 1.0, b = sqrt(1.0 - m), s = 0.5 * m, pw = 0.5;
    for (int it = 0; it < 40; ++it) {
        double c = 0.5 * (a - b);
        double an = 0.5 * (a + b);
        b = sqrt(a * b);
        a = an;
        pw *= 2.0;
        s += pw * c * c;
        if (fabs(c) <= 1e-17 * a) break;
    }
    double k = 3.14159265358979323846 / (2.0 * a);
    K = k;
    E = k * (1.0 - s);
}
constexpr double MU0_SI = 4e-7 * 3.14159265358979323846;
__device__ __forceinline__ double green_si(double r_src, double z_src, double r_obs, double z_obs) {
    // core/fusion_kernel_free_boundary.py:58-80
    double dr = r_obs - r_src, dz = z_obs - z_src;
    bool self_mask = (dr * dr + dz * dz) < 1e-24;
    double denom = (r_obs + r_src) * (r_obs + r_src) + dz * dz;
    double k2 = denom > 1e-30 ? ((4.0 * r_obs) * r_src) / fmax(denom, 1e-30) : 0.0;
    k2 = fmin(fmax(k2, 1e-12), 1.0 - 1e-12);
    double k = sqrt(k2), K, E;
    ellip_ke(k2, K, E);
    double prefactor = (MU0_SI / (2.0 * 3.14159265358979323846)) * sqrt(r_obs * r_src);
    double flux = (prefactor * ((2.0 - k2) * K - 2.0 * E)) / k;
    return self_mask ? 0.0 : flux;
}
// mode 0: calculate_vacuum_field (core/fusion_kernel.py:232-249); mode 1: compute_external_flux
// ring_only: evaluate only the Dirichlet ring (the only part the free-boundary loop consumes)
__global__ void k_coil_flux(Fld outF, int nz, int nr, int ld, const double* r_axis, const double* z_axis,
                            int n_coils, const double* coil_r, const double* coil_z, const double* coil_i,
                            const double* coil_turns, double mu0, int mode, int ring_only = 0) {
    int b = blockIdx.z;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    if (ring_only && !(iz == 0 || iz == nz - 1 || ir == 0 || ir == nr - 1)) {  // only the Dirichlet ring is consumed
        outF.at(b)[(long)iz * ld + ir] = 0.0;
        return;
    }
    double R = r_axis[ir], Z = z_axis[iz];
    double acc = 0.0;
    for (int c = 0; c < n_coils; ++c) {
        long ci = (long)b * n_coils + c;
        double rc = coil_r[ci], zc = coil_z[ci], cur = coil_i[ci];
        if (mode == 0) {
            double dZ = Z - zc;
            double rp = (R + rc) * (R + rc);
            double k2 = ((4.0 * R) * rc) / (rp + dZ * dZ);
            k2 = fmin(fmax(k2, 1e-12), 1.0 - 1e-12);
            double K, E;
            ellip_ke(k2, K, E);
            double prefactor = (mu0 * cur) / (2.0 * 3.14159265358979323846);
            double term = ((2.0 - k2) * K - 2.0 * E) / sqrt(k2);
            acc += (prefactor * sqrt(R * rc)) * term;
        } else {
            double ieff = cur * (coil_turns ? coil_turns[ci] : 1.0);
            acc += ieff * green_si(rc, zc, R, Z);
        }
    }
    outF.at(b)[(long)iz * ld + ir] = acc;
}
__global__ void k_green_points(double* out, int n_coils, const double* coil_r, const double* coil_z, int n_pts,
                               const double* pr, const double* pz) {
    int p = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (p >= n_pts) return;
    double rs = coil_r[c], zs = coil_z[c], ro = pr[p], zo = pz[p];
    double v;
    if (!finite_d(rs) || !finite_d(zs) || !finite_d(ro) || !finite_d(zo) || rs <= 0.0 || ro <= 0.0) v = NAN;
    else v = green_si(rs, zs, ro, zo);
    out[(long)c * n_pts + p] = v;
}

// ---------------------------------------------------------------- polyglot lane source (a13)
// core/jax_gs_solver.py:89-123 with psi_axis from k_max_interior; two passes (sum, then source)
__device__ __forceinline__ double np_jraw(double psi, double R, double mu0, double psi_axis, double denom,
                                          double beta_mix) {
    double x = (psi - psi_axis) / denom;
    x = fmin(fmax(x, 0.0), 1.0);
    double prof = (x >= 0.0 && x < 1.0) ? 1.0 - x : 0.0;
    double r_safe = fmax(R, 1e-10);
    double j_p = R * prof;
    double j_f = prof / (mu0 * r_safe);
    return beta_mix * j_p + (1.0 - beta_mix) * j_f;
}
__global__ void __launch_bounds__(256) k_np_jsum(Fld psiF, const double* scal, int nz, int nr, int ld,
                                                 const double* r_axis, double mu0, double beta_mix, double* part,
                                                 int nblk) {
    __shared__ double sh[4];
    int b = blockIdx.y;
    const double* psi = psiF.at(b);
    double pa = scal[0], den = scal[1];
    const long n = (long)nz * nr;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (q < n) {
            int iz = (int)(q / nr), ir = (int)(q - (long)iz * nr);
            s += np_jraw(psi[(long)iz * ld + ir], r_axis[ir], mu0, pa, den, beta_mix);
        }
    }
    double bs = block_sum<256>(s, sh);
    if (threadIdx.x == 0) part[(long)b * nblk + blockIdx.x] = bs;
}
// scal[0]=psi_axis, scal[1]=denom, scal[2]=scale.  stage 0: from pmax -> axis/denom; stage 1: from psum -> scale
__global__ void __launch_bounds__(256) k_np_final(double* scal, const double* part, int nblk, int stage, double dr,
                                                  double dz, double ip_target) {
    __shared__ double sh[4];
    if (stage == 0) {
        double m = -INFINITY;
        for (int i = threadIdx.x; i < nblk; i += 256) m = fmax(m, part[i]);
        double r = block_max<256>(m, sh);
        if (threadIdx.x == 0) {
            double denom = 0.0 - r;
            if (fabs(denom) < 1e-9) denom = denom != 0.0 ? (denom > 0 ? 1e-9 : -1e-9) : 1e-9;
            scal[0] = r;
            scal[1] = denom;
        }
    } else {
        double s = 0.0;
        for (int i = threadIdx.x; i < nblk; i += 256) s += part[i];
        double tot = block_sum<256>(s, sh);
        if (threadIdx.x == 0) {
            double ic = (tot * dr) * dz;
            scal[2] = ip_target / fmax(fabs(ic), 1e-9);
        }
    }
}
__global__ void k_np_source(Fld srcF, Fld psiF, const double* scal, int nz, int nr, int ld, const double* r_axis,
                            double mu0, double beta_mix) {
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    long c = (long)iz * ld + ir;
    double jr = np_jraw(psiF.at(0)[c], r_axis[ir], mu0, scal[0], scal[1], beta_mix);
    srcF.at(0)[c] = ((-mu0) * r_axis[ir]) * (jr * scal[2]);
}
__global__ void k_axpby(Fld outF, Fld aF, Fld bF, double ca, double cb, int nz, int nr, int ld) {
    int b = blockIdx.z;
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    long c = (long)iz * ld + ir;
    outF.at(b)[c] = ca * aF.at(b)[c] + cb * bF.at(b)[c];
}

// ---------------------------------------------------------------- legacy FastSolver ABI (a14)
// hpc/solver.cpp:93-127,164-188: colour pass with source = -1.0*R*J built in-kernel, constant ring,
// running max|delta| (non-negative doubles order like their bit patterns -> integer atomicMax)
__global__ void k_fast_ring(double* psi, int nz, int nr, int ld, double v, const int* done) {
    if (done && *done) return;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int iz, ir;
    if (t < nr) { iz = 0; ir = t; }
    else if (t < 2 * nr) { iz = nz - 1; ir = t - nr; }
    else if (t < 2 * nr + nz) { iz = t - 2 * nr; ir = 0; }
    else if (t < 2 * nr + 2 * nz) { iz = t - 2 * nr - nz; ir = nr - 1; }
    else return;
    psi[(long)iz * ld + ir] = v;
}
__global__ void __launch_bounds__(256) k_fast_half(double* psi, const double* j, int nz, int nr, int ld,
                                                   const double* Rtab, const double* c_rp, const double* c_rm,
                                                   double c_z, double center, double omega, double om1, int parity,
                                                   unsigned long long* delta_bits, const int* done) {
    __shared__ double sh[4];
    if (done && *done) return;
    int iz = 1 + blockIdx.y * 4 + (threadIdx.x >> 6);
    int t = blockIdx.x * 64 + (threadIdx.x & 63);
    int ir = 1 + ((iz + 1 + parity) & 1) + 2 * t;
    double d = 0.0;
    if (iz < nz - 1 && ir < nr - 1) {
        long c = (long)iz * ld + ir;
        double source = -1.0 * Rtab[ir] * j[c];
        double p_gs = (source + c_z * (psi[c + ld] + psi[c - ld]) + c_rp[ir] * psi[c + 1] + c_rm[ir] * psi[c - 1]) /
                      center;
        double old = psi[c];
        double nw = om1 * old + omega * p_gs;
        psi[c] = nw;
        d = fabs(nw - old);
        if (!(d == d)) d = 0.0;
    }
    double bm = block_max<256>(d, sh);
    if (threadIdx.x == 0 && bm > 0.0) atomicMax(delta_bits, (unsigned long long)__double_as_longlong(bm));
}
// after a sweep: record delta, decide stop (solver.cpp:303-310)
__global__ void k_fast_ctl(unsigned long long* delta_bits, double* last_delta, int* performed, int* done, double tol,
                           int use_tol) {
    if (*done) return;
    double d = __longlong_as_double((long long)*delta_bits);
    *last_delta = d;
    *performed += 1;
    *delta_bits = 0ull;
    if (use_tol && d <= tol) *done = 1;
}

// compute_b_field, core/fusion_kernel.py:450-456
// member >= 0: that member into br/bz (nz*nr); member < 0: every member b = blockIdx.z into br/bz + b*nz*nr
__global__ void k_b_field(double* br, double* bz, Fld psiF, int member, int nz, int nr, int ld, double dr, double dz,
                          const double* r_axis) {
    int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    if (member < 0) {
        member = blockIdx.z;
        br += (long)member * nz * nr;
        bz += (long)member * nz * nr;
    }
    const double* psi = psiF.at(member);
    long c = (long)iz * ld + ir;
    double gz, gr;
    if (iz == 0) gz = (psi[c + ld] - psi[c]) / dz;
    else if (iz == nz - 1) gz = (psi[c] - psi[c - ld]) / dz;
    else gz = (psi[c + ld] - psi[c - ld]) / (2.0 * dz);
    if (ir == 0) gr = (psi[c + 1] - psi[c]) / dr;
    else if (ir == nr - 1) gr = (psi[c] - psi[c - 1]) / dr;
    else gr = (psi[c + 1] - psi[c - 1]) / (2.0 * dr);
    double rs = fmax(r_axis[ir], 1e-6);
    br[(long)iz * nr + ir] = -(1.0 / rs) * gz;
    bz[(long)iz * nr + ir] = (1.0 / rs) * gr;
}


// polyglot lane diagnostics (core/jax_gs_solver.py:196-316).
// gs_delta_star_np: d2/dR2 - (1/R) d/dR + d2/dZ2 with THIS lane's association ((E-W)/(2 dR R), dR*dR); zero ring
__global__ void k_np_delta_star(double* out, Fld psiF, int nz, int nr, int ld, double dr, double dz, const double* r_axis) {
    const int ir = blockIdx.x * 64 + threadIdx.x, iz = blockIdx.y * 4 + threadIdx.y;
    if (ir >= nr || iz >= nz) return;
    const double* psi = psiF.at(0);
    double v = 0.0;
    if (iz >= 1 && iz <= nz - 2 && ir >= 1 && ir <= nr - 2) {
        const long c = (long)iz * ld + ir;
        const double d2r = (psi[c + 1] - 2.0 * psi[c] + psi[c - 1]) / (dr * dr);
        const double d1r = (psi[c + 1] - psi[c - 1]) / ((2.0 * dr) * r_axis[ir]);
        const double d2z = (psi[c + ld] - 2.0 * psi[c] + psi[c - ld]) / (dz * dz);
        v = d2r - d1r + d2z;
    }
    out[(long)iz * nr + ir] = v;
}
// gs_equation_residual_np: a_E psi_E + a_W psi_W + a_NS (psi_S + psi_N) - a_C psi_C - source on the interior; per-block
// max|res|, sum res^2, max|source|, sum source^2.  grid (nblk, 1), 256 threads x 4 interior points
__global__ void __launch_bounds__(256) k_np_eq_residual(Fld psiF, Fld srcF, Lvl L, double* part, int nblk) {
    __shared__ double sh[4];
    const double* psi = psiF.at(0);
    const double* src = srcF.at(0);
    const int ni = L.nr - 2;
    const long nint = (long)(L.nz - 2) * ni;
    double rm = 0.0, r2 = 0.0, sm = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (q < nint) {
            const int iz = 1 + (int)(q / ni), ir = 1 + (int)(q - (long)(iz - 1) * ni);
            const long c = (long)iz * L.ld + ir;
            const double sv = src[c];
            const double r = L.a_e[ir] * psi[c + 1] + L.a_w[ir] * psi[c - 1] + L.a_ns * (psi[c - L.ld] + psi[c + L.ld]) - L.a_c * psi[c] - sv;
            rm = fmax(rm, fabs(r));
            r2 += r * r;
            sm = fmax(sm, fabs(sv));
            s2 += sv * sv;
        }
    }
    const double brm = block_max<256>(rm, sh), br2 = block_sum<256>(r2, sh), bsm = block_max<256>(sm, sh), bs2 = block_sum<256>(s2, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = brm;
        part[nblk + blockIdx.x] = br2;
        part[2 * nblk + blockIdx.x] = bsm;
        part[3 * nblk + blockIdx.x] = bs2;
    }
}

// ---------------------------------------------------------------- f3: flux sampling and diagnostic response matrices
// numpy.searchsorted(a, x) (side="left") on an ascending axis: first index i with a[i] >= x
__device__ __forceinline__ int searchsorted_left(const double* a, int n, double x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// interp_psi (core/fusion_kernel_free_boundary.py:562-580) at n_pts points for every member; out [B][n_pts]
__global__ void k_sample_psi(double* out, Fld psiF, int nz, int nr, int ld, double dr, double dz, const double* r_axis,
                             const double* z_axis, int n_pts, const double* pr, const double* pz) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (p >= n_pts) return;
    const double R = pr[p], Z = pz[p];
    int ir = searchsorted_left(r_axis, nr, R) - 1, iz = searchsorted_left(z_axis, nz, Z) - 1;
    ir = max(0, min(ir, nr - 2));
    iz = max(0, min(iz, nz - 2));
    double tr = (R - r_axis[ir]) / dr, tz = (Z - z_axis[iz]) / dz;
    tr = fmax(0.0, fmin(1.0, tr));
    tz = fmax(0.0, fmin(1.0, tz));
    const double* psi = psiF.at(b);
    const long c = (long)iz * ld + ir;
    double v = ((1 - tr) * (1 - tz)) * psi[c];
    v = v + (tr * (1 - tz)) * psi[c + 1];
    v = v + ((1 - tr) * tz) * psi[c + ld];
    v = v + (tr * tz) * psi[c + ld + 1];
    out[(long)b * n_pts + p] = v;
}
// build_magnetic_probe_response_matrix (core/fusion_kernel_free_boundary.py:285-373): out[row][coil]; rows 0..n_flux-1 are
// flux loops (turns * G), rows n_flux.. are B probes by central differences of the same Green's function
// (dir 0: B_R = -(dG/dZ)/R, dir 1: B_Z = (dG/dR)/R).  grid (ceil(rows/64), n_coils)
__global__ void k_probe_response(double* out, int n_coils, const double* coil_r, const double* coil_z, const double* coil_turns,
                                 int n_flux, const double* fr, const double* fz, int n_b, const double* br, const double* bz,
                                 const int* bdir) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (row >= n_flux + n_b) return;
    const double rc = coil_r[c], zc = coil_z[c], turns = coil_turns ? coil_turns[c] : 1.0;
    double v;
    if (row < n_flux) {
        v = turns * green_si(rc, zc, fr[row], fz[row]);
    } else {
        const int q = row - n_flux;
        const double r_obs = br[q], z_obs = bz[q];
        const double eps_r = fmax(1.0e-5, 1.0e-5 * fabs(r_obs));
        const double eps_z = fmax(1.0e-5, 1.0e-5 * (1.0 + fabs(z_obs)));
        const double r_safe = fmax(r_obs, eps_r);
        if (bdir[q] == 0) {
            const double plus = turns * green_si(rc, zc, r_safe, z_obs + eps_z);
            const double minus = turns * green_si(rc, zc, r_safe, z_obs - eps_z);
            v = -(plus - minus) / ((2.0 * eps_z) * r_safe);
        } else {
            const double plus = turns * green_si(rc, zc, r_safe + eps_r, z_obs);
            const double minus = turns * green_si(rc, zc, r_safe - eps_r, z_obs);
            v = (plus - minus) / ((2.0 * eps_r) * r_safe);
        }
    }
    out[(long)row * n_coils + c] = v;
}


// ---------------------------------------------------------------- f4: transport coupling
// numpy.interp at one abscissa (compiled_base.c arr_interp): xp ascending (m knots), left = fp[0], right = fp[m-1]
__device__ __forceinline__ int interp_interval(const double* xp, int m, double x) {  // largest j with xp[j] <= x (x within range)
    int lo = 0, hi = m;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ double np_interp1(const double* xp, const double* fp, int m, double x, int j) {
    if (x != x) return x;
    if (x > xp[m - 1]) return fp[m - 1];
    if (x < xp[0]) return fp[0];
    if (j == m - 1) return fp[j];
    if (xp[j] == x) return fp[j];
    const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    double r = slope * (x - xp[j]) + fp[j];
    if (r != r) {
        r = slope * (x - xp[j + 1]) + fp[j + 1];
        if (r != r && fp[j] == fp[j + 1]) r = fp[j];
    }
    return r;
}
// map_profiles_to_2d topology (core/integrated_transport_solver_coupling.py:41-52): psi_axis = max psi (no floor), psi_edge =
// X-point flux, replaced by min psi when within 1.0 of the axis value; denom floored at 1e-9.  one block per member
__global__ void __launch_bounds__(256) k_map_topo_final(Fld psiF, Member* st, const TopoPart* part, int nblk, int nr, int ld) {
    __shared__ VI shv[4];
    __shared__ double shd[4];
    const int b = blockIdx.x;
    VI mx{0.0, -1}, mn{0.0, -1};
    double minpsi = INFINITY;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        TopoPart t = part[(long)b * nblk + i];
        mx = vi_max(mx, VI{t.maxv, t.maxi});
        mn = vi_min(mn, VI{t.minb, t.mini});
        minpsi = fmin(minpsi, t.minpsi);
    }
    VI rmx = block_vi<true, 256>(mx, shv);
    VI rmn = block_vi<false, 256>(mn, shv);
    double rmp = -block_max<256>(-minpsi, shd);
    if (threadIdx.x == 0) {
        Member& m = st[b];
        const double* psi = psiF.at(b);
        const double pa = rmx.i >= 0 ? rmx.v : 0.0;
        double pe;
        if (rmn.i >= 0) {
            int iz = (int)(rmn.i / nr), ir = (int)(rmn.i - (long)iz * nr);
            pe = psi[(long)iz * ld + ir];
            if (!finite_d(pe)) pe = nan_to_num300(pe);
        } else pe = rmp;
        if (fabs(pe - pa) < 1.0) pe = rmp;
        double denom = pe - pa;
        if (fabs(denom) < 1e-9) denom = 1e-9;
        m.psi_axis = pa; m.psi_bnd = pe; m.denom = denom; m.x_idx = rmn.i;
    }
}
// rho_2d = sqrt(clip((psi - psi_axis)/denom, 0, 1)); Pressure_2D = interp(rho_2d, rho, p1d); J = Pressure_2D*R + interp(rho_2d,
// rho, jbs1d) (:53-68); per-block partial sums of J.  knots staged in LDS (n_rho <= MAP_MAX_RHO).  grid (nblk, B)
constexpr int MAP_MAX_RHO = 1024;
__global__ void __launch_bounds__(256) k_map_profiles(Fld jF, double* p2d_out, Fld psiF, const Member* st, int nz, int nr, int ld,
                                                      const double* r_axis, int n_rho, const double* rho, const double* p1d,
                                                      const double* jbs1d, double* part, int nblk) {
    __shared__ double sh[4];
    __shared__ double s_rho[MAP_MAX_RHO], s_p[MAP_MAX_RHO], s_j[MAP_MAX_RHO];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < n_rho; i += 256) {
        s_rho[i] = rho[i];
        s_p[i] = p1d[(long)b * n_rho + i];
        s_j[i] = jbs1d[(long)b * n_rho + i];
    }
    __syncthreads();
    const double* psi = psiF.at(b);
    double* jo = jF.at(b);
    const double pa = st[b].psi_axis, denom = st[b].denom;
    const long n = (long)nz * nr;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (q < n) {
            int iz = (int)(q / nr), ir = (int)(q - (long)iz * nr);
            double pn = (psi[(long)iz * ld + ir] - pa) / denom;
            pn = fmin(fmax(pn, 0.0), 1.0);
            if (pn != pn) pn = NAN;
            const double x = sqrt(pn);
            const int j = (x == x && x <= s_rho[n_rho - 1] && x >= s_rho[0]) ? interp_interval(s_rho, n_rho, x) : 0;
            const double pv = np_interp1(s_rho, s_p, n_rho, x, j);
            const double jb = np_interp1(s_rho, s_j, n_rho, x, j);
            const double jv = (pv * r_axis[ir]) + jb;
            jo[(long)iz * ld + ir] = jv;
            if (p2d_out) p2d_out[(long)b * n + q] = pv;
            s += jv;
        }
    }
    double bs = block_sum<256>(s, sh);
    if (threadIdx.x == 0) part[(long)b * nblk + blockIdx.x] = bs;
}
// J *= I_target / I_curr when |I_curr| > 1e-9 (:70-72); every block finishes I_curr = sum(J)*dR*dZ from the partials
__global__ void __launch_bounds__(256) k_map_scale(Fld jF, Member* st, int nz, int nr, int ld, double dr, double dz, const double* part,
                                                   int nblk) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[(long)b * nblk + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const double ic = (((sh[0] + sh[1]) + (sh[2] + sh[3])) * dr) * dz;
    if (blockIdx.x == 0 && threadIdx.x == 0) { st[b].i_current = ic; st[b].jscale = fabs(ic) > 1e-9 ? st[b].i_target / ic : 1.0; }
    if (!(fabs(ic) > 1e-9)) return;
    const double sc = st[b].i_target / ic;
    double* jo = jF.at(b);
    const long n = (long)nz * nr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long q = (long)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (q < n) {
            int iz = (int)(q / nr), ir = (int)(q - (long)iz * nr);
            jo[(long)iz * ld + ir] *= sc;
        }
    }
}

// LAPACK dgtsv for one right-hand side (what scipy.linalg.solve_banded((1,1), ...) calls): Gaussian elimination with partial
// pivoting; dl, d, du, b are overwritten (b <- x); returns info.  Element i of an array lies at [i * st]: st = 1 for a system
// stored contiguously, st = batch for the [n][batch] device layout of the batched kernels below, in which the lanes of a wave
// (consecutive systems) read consecutive addresses.
//
// Same operations in the same order as the reference routine, but the recurrence runs in REGISTERS: step i needs the current
// d[i], du[i], b[i] (carried from step i - 1) and the ORIGINAL dl[i], d[i+1], du[i+1], b[i+1], which no earlier step has touched --
// so the originals of GT_K steps are requested together in front of the steps that use them (one memory round trip per GT_K
// steps instead of a store -> load round trip through memory inside every step, which is what bounded rounds 1-5's form at
// ~1.1 us per row whatever the layout), and the finished rows are stored behind them.  The back substitution carries x[i+1], x[i+2]
// the same way.
constexpr int GT_K = 8;
__device__ __forceinline__ int gtsv1(int n, double* __restrict__ dl, double* __restrict__ d, double* __restrict__ du, double* __restrict__ b, long st = 1) {
#define GT(a, i) (a)[(long)(i) * st]
    if (n == 0) return 0;
    double di = GT(d, 0), dui = n > 1 ? GT(du, 0) : 0.0, bi = GT(b, 0);  // current values of row i
    // originals of rows i0 + k (dl) and i0 + k + 1 (d, du, b); clamped reads beyond the end are never used.  The chunk behind the
    // one being eliminated is already in flight (double buffer): a wave of systems is alone on its SIMD, nothing else hides the latency
    auto fetch = [&](int i0, double (&o_dl)[GT_K], double (&o_d1)[GT_K], double (&o_du1)[GT_K], double (&o_b1)[GT_K]) {
#pragma unroll
        for (int k = 0; k < GT_K; ++k) {
            const int i = min(i0 + k, n - 2);
            o_dl[k] = GT(dl, i);
            o_d1[k] = GT(d, i + 1);
            o_du1[k] = GT(du, min(i + 1, n - 2));
            o_b1[k] = GT(b, i + 1);
        }
    };
    auto eliminate = [&](int i0, const double (&o_dl)[GT_K], const double (&o_d1)[GT_K], const double (&o_du1)[GT_K], const double (&o_b1)[GT_K]) -> int {
#pragma unroll
        for (int k = 0; k < GT_K; ++k) {
            const int i = i0 + k;
            if (i >= n - 1) break;
            const bool last = i == n - 2;
            const double dli = o_dl[k];
            double d1 = o_d1[k], du1 = o_du1[k], b1 = o_b1[k];
            if (fabs(di) >= fabs(dli)) {
                if (di != 0.0) {
                    const double fact = dli / di;
                    d1 = d1 - fact * dui;
                    b1 = b1 - fact * bi;
                } else {
                    GT(d, i) = di; GT(du, i) = dui; GT(b, i) = bi;  // (the state the reference leaves behind a zero pivot)
                    return i + 1;
                }
                GT(d, i) = di; GT(du, i) = dui; GT(b, i) = bi;
                if (!last) GT(dl, i) = 0.0;
            } else {
                const double fact = di / dli;
                GT(d, i) = dli;
                const double temp = d1;
                d1 = dui - fact * temp;
                if (!last) {
                    GT(dl, i) = du1;
                    du1 = -fact * du1;
                }
                GT(du, i) = temp;
                GT(b, i) = b1;
                b1 = bi - fact * b1;
            }
            di = d1; dui = du1; bi = b1;
        }
        return 0;
    };
    if (n > 1) {
        double a_dl[GT_K], a_d1[GT_K], a_du1[GT_K], a_b1[GT_K], c_dl[GT_K], c_d1[GT_K], c_du1[GT_K], c_b1[GT_K];
        fetch(0, a_dl, a_d1, a_du1, a_b1);
        for (int i0 = 0; i0 < n - 1; i0 += 2 * GT_K) {
            // (a chunk's rows are only ever modified by the elimination of the chunk in front of it: rows up to i0 + GT_K by the
            //  first call below, whose stores touch rows < i0 + GT_K of d / du / b and dl -- the second fetch reads rows > i0 + GT_K)
            if (i0 + GT_K < n - 1) fetch(i0 + GT_K, c_dl, c_d1, c_du1, c_b1);
            int info = eliminate(i0, a_dl, a_d1, a_du1, a_b1);
            if (info) return info;
            if (i0 + GT_K >= n - 1) break;
            if (i0 + 2 * GT_K < n - 1) fetch(i0 + 2 * GT_K, a_dl, a_d1, a_du1, a_b1);
            info = eliminate(i0 + GT_K, c_dl, c_d1, c_du1, c_b1);
            if (info) return info;
        }
    }
    GT(d, n - 1) = di; GT(b, n - 1) = bi;
    if (di == 0.0) return n;
    double x1 = bi / di, x2 = 0.0;  // x[i + 1], x[i + 2]
    GT(b, n - 1) = x1;
    if (n > 1) {
        const double x0 = (GT(b, n - 2) - GT(du, n - 2) * x1) / GT(d, n - 2);
        GT(b, n - 2) = x0;
        x2 = x1; x1 = x0;
    }
    auto fetch_back = [&](int i0, double (&r_b)[GT_K], double (&r_du)[GT_K], double (&r_dl)[GT_K], double (&r_d)[GT_K]) {
#pragma unroll
        for (int k = 0; k < GT_K; ++k) {
            const int i = max(i0 - k, 0);
            r_b[k] = GT(b, i); r_du[k] = GT(du, i); r_dl[k] = GT(dl, i); r_d[k] = GT(d, i);
        }
    };
    auto substitute = [&](int i0, const double (&r_b)[GT_K], const double (&r_du)[GT_K], const double (&r_dl)[GT_K], const double (&r_d)[GT_K]) {
#pragma unroll
        for (int k = 0; k < GT_K; ++k) {
            const int i = i0 - k;
            if (i < 0) break;
            const double x0 = (r_b[k] - r_du[k] * x1 - r_dl[k] * x2) / r_d[k];
            GT(b, i) = x0;
            x2 = x1; x1 = x0;
        }
    };
    if (n > 2) {
        double a_b[GT_K], a_du[GT_K], a_dl[GT_K], a_d[GT_K], c_b[GT_K], c_du[GT_K], c_dl[GT_K], c_d[GT_K];
        fetch_back(n - 3, a_b, a_du, a_dl, a_d);
        for (int i0 = n - 3; i0 >= 0; i0 -= 2 * GT_K) {  // (a chunk stores x into rows it has already read; the chunk below is in flight meanwhile)
            if (i0 - GT_K >= 0) fetch_back(i0 - GT_K, c_b, c_du, c_dl, c_d);
            substitute(i0, a_b, a_du, a_dl, a_d);
            if (i0 - GT_K < 0) break;
            if (i0 - 2 * GT_K >= 0) fetch_back(i0 - 2 * GT_K, a_b, a_du, a_dl, a_d);
            substitute(i0 - GT_K, c_b, c_du, c_dl, c_d);
        }
    }
    return 0;
#undef GT
}
// dst[c][r] = src[r][c] for a (rows x cols) array: the [batch][n] host layout of the C ABI <-> the [n][batch] device layout of the
// batched one-thread-per-system kernels.  32 x 32 tiles through LDS (+1 column: no bank conflicts); grid (ceil(cols/32), ceil(rows/32))
__global__ void __launch_bounds__(256) k_transpose(double* __restrict__ dst, const double* __restrict__ src, int rows, int cols) {
    __shared__ double t[32][33];
    dst += (long)blockIdx.z * rows * cols;  // (gridDim.z arrays of rows x cols lying behind each other, transposed in one launch)
    src += (long)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < rows && c < cols) t[ty + 8 * k][tx] = src[(long)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) dst[(long)c * rows + r] = t[tx][ty + 8 * k];
    }
}
// batched tridiagonal solve: one thread per system.  Work arrays: element i of system s at [s * sys + i * el] -- (sys, el) =
// (1, batch) for the [n][batch] layout (coalesced: the default), (n, 1) for [batch][n] (dl, du use n-1 entries)
__global__ void __launch_bounds__(64) k_gtsv(int batch, int n, double* dl, double* d, double* du, double* b, int* info, long sys, long el) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    info[s] = gtsv1(n, dl + s * sys, d + s * sys, du + s * sys, b + s * sys, el);
}
// CurrentDiffusionSolver.step (core/current_diffusion.py:277-353) for `batch` independent profiles: Sauter resistivity
// (:175-207), Crank-Nicolson assembly, dgtsv.  one thread per member; scratch [batch][4n]
struct CdConst { double r0, a, dt, z_eff, mu0_a2; };
// Two launches: the Crank-Nicolson rows are independent of each other -- k_cd_assemble builds them with one thread per (member,
// rho point) (the pow / sqrt of the Sauter resistivity were 60 % of rounds 1-5's one-thread-per-member kernel) -- and
// k_cd_solve runs the dgtsv recurrence, one thread per member, and leaves the new psi in the rhs rows.
// Element i of member s of every profile / scratch row at [s * sys + i * el] (see k_gtsv); scratch = sub, diag, sup, rhs of
// batch * n doubles each.
__global__ void __launch_bounds__(256) k_cd_assemble(int batch, int n, const double* rho, const double* psi, CdConst c, const double* te,
                                                    const double* j_bs, const double* j_cd, const double* j_ext, double* scratch, long sys, long el) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, tot = (long)batch * n;
    if (t >= tot) return;
    // consecutive threads walk the contiguous axis of the SCRATCH layout; the profiles are read where the ABI left them ([batch][n]:
    // with the [n][batch] scratch these reads are strided -- 5 values per thread of a 10-us kernel -- which saves five transposes)
    const int i = el == 1 ? (int)(t % n) : (int)(t / batch), s = el == 1 ? (int)(t / n) : (int)(t % batch);
    const long base = s * sys, ibase = (long)s * n;
#define AT(a, i) (a)[base + (long)(i) * el]
#define IN(a, i) (a)[ibase + (i)]
    double* sub = scratch;
    double* diag = scratch + tot;
    double* sup = scratch + 2 * tot;
    double* rhs = scratch + 3 * tot;
    const double drho = rho[1] - rho[0];
    const double drho2 = drho * drho;
    const double alpha = c.dt / 2.0;
    // q_from_psi (:210-233) only feeds the collisionality nu_star_e, which the reference computes and discards (:199, noqa F841)
    double tk = fmax(IN(te, i), 1e-3), eps = fmax(rho[i] * c.a / c.r0, 1e-6);
    const double eta_sp = 1.65e-9 * c.z_eff * 17.0 / pow(tk, 1.5);
    double ft = 1.0 - ((1.0 - eps) * (1.0 - eps)) / (sqrt(1.0 - eps * eps) * (1.0 + 1.46 * sqrt(eps)));
    ft = fmax(0.0, fmin(ft, 1.0));
    const double cr = 1.0 - (1.0 + 0.36 / c.z_eff) * ft + (0.59 / c.z_eff) * (ft * ft);
    const double eta = fmax(eta_sp / (1.0 - ft) * cr, eta_sp);
    const double D = eta / c.mu0_a2;
    const double jt = (IN(j_bs, i) + IN(j_cd, i)) + (j_ext ? IN(j_ext, i) : 0.0);
    if (i == 0) {
        AT(sub, 0) = 0.0;
        AT(diag, 0) = 1.0 + alpha * 4.0 * D / drho2;
        AT(rhs, 0) = IN(psi, 0) + alpha * 4.0 * D * (IN(psi, 1) - IN(psi, 0)) / drho2 + c.dt * c.r0 * eta * jt;
        AT(sup, 0) = -alpha * 4.0 * D / drho2;
    } else if (i < n - 1) {
        const double r = rho[i];
        const double cp = D * (1.0 / drho2 - 1.0 / (2.0 * r * drho));
        const double cc = D * (-2.0 / drho2);
        const double cn = D * (1.0 / drho2 + 1.0 / (2.0 * r * drho));
        AT(sub, i) = -alpha * cp;
        AT(diag, i) = 1.0 - alpha * cc;
        const double l = cp * IN(psi, i - 1) + cc * IN(psi, i) + cn * IN(psi, i + 1);
        AT(rhs, i) = IN(psi, i) + alpha * l + c.dt * c.r0 * eta * jt;
        AT(sup, i) = -alpha * cn;
    } else {
        AT(diag, n - 1) = 1.0;
        AT(sub, n - 1) = 0.0;
        AT(sup, n - 1) = 0.0;
        AT(rhs, n - 1) = IN(psi, n - 1);
    }
#undef AT
#undef IN
}
__global__ void __launch_bounds__(64) k_cd_solve(int batch, int n, double* scratch, int* info, long sys, long el) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    const long tot = (long)batch * n, base = s * sys;
    info[s] = gtsv1(n, scratch + base + el, scratch + tot + base, scratch + 2 * tot + base, scratch + 3 * tot + base, el);
}


// time recurrence of solve_flux_evolution_nonadiabatic (core/current_diffusion.py:141-158) for `batch` profiles: one thread per
// (profile, rho) walks the n_steps steps; rate / source are [batch][n_steps+1][n], psi [batch][n_steps+1][n] (row 0 = psi0),
// inc / drop / resid [batch][n_steps][n].  Consecutive lanes own consecutive rho: every row access is coalesced.
__global__ void k_flux_evolution(int batch, int n, int n_steps, double dt, const double* rate, const double* source, double* psi,
                                 double* inc, double* drop, double* resid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= n || b >= batch) return;
    const long o1 = (long)b * (n_steps + 1) * n + i, o0 = (long)b * n_steps * n + i;
    double p = psi[o1];
    double r0 = rate[o1], s0 = source[o1];
    for (int k = 0; k < n_steps; ++k) {
        const double r1 = rate[o1 + (long)(k + 1) * n], s1 = source[o1 + (long)(k + 1) * n];
        const double gamma = 0.5 * (r0 + r1), s = 0.5 * (s0 + s1);
        const double decay = exp(-gamma * dt);
        const double di = gamma > 0.0 ? s * (1.0 - decay) / gamma : dt * s;
        const double dd = p * (1.0 - decay);
        const double nx = p - dd + di;
        psi[o1 + (long)(k + 1) * n] = nx;
        inc[o0 + (long)k * n] = di;
        drop[o0 + (long)k * n] = dd;
        resid[o0 + (long)k * n] = nx - nx;
        p = nx; r0 = r1; s0 = s1;
    }
}

}  // namespace gs

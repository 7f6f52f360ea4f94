// Micro-benchmark: issue rate / dependent latency of fp64 VALU instructions and DPP moves on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_issue tools/ubench/fp64_issue.hip && /tmp/fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* cyc, double a, double b, int iters) {
    double x0 = a + threadIdx.x, x1 = a * 2 + threadIdx.x, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 8 independent fma chains
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                               "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));)
        } else if (MODE == 1) {  // one dependent fma chain (128 per iteration)
            REP16(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                               "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                               : "+v"(x0) : "v"(b), "v"(a));)
        } else if (MODE == 2) {  // 8 independent mul
            REP16(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                               "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));)
        } else if (MODE == 3) {  // 8 independent add
            REP16(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                               "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));)
        } else if (MODE == 4) {  // dependent add chain
            REP16(asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                               "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                               : "+v"(x0) : "v"(b));)
        } else if (MODE == 5) {  // 4 x (2 dpp movs of a double) independent
            REP16(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %4, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_mov_b32_dpp %5, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               : "+v"(((int*)&x0)[0]), "+v"(((int*)&x0)[1]), "+v"(((int*)&x1)[0]), "+v"(((int*)&x1)[1]), "+v"(((int*)&x2)[0]), "+v"(((int*)&x2)[1]),
                                 "+v"(((int*)&x3)[0]), "+v"(((int*)&x3)[1]));)
        } else if (MODE == 6) {  // 8 independent fma with an SGPR operand (as the compiler emits for uniform constants)
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                               "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(b), "v"(a));)
        } else if (MODE == 7) {  // fp32 fma for comparison
            float f0 = (float)x0, f1 = (float)x1, f2 = (float)x2, f3 = (float)x3, fb = (float)b, fa = (float)a;
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                               "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fb), "v"(fa));)
            x0 += f0 + f1 + f2 + f3;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
    const int iters = 200, nblk = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * nblk * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * nblk);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nblk, 256>>>(out, cyc, 1.0000001, 0.9999999, iters);
    hipEventRecord(e0);
    k<MODE><<<nblk, 256>>>(out, cyc, 1.0000001, 0.9999999, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nblk, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= nblk;
    const double ninst = 128.0 * iters;
    printf("%-34s waves/SIMD %d: %7.2f s_memtime-cycles per instr per wave, %7.2f per SIMD-slot; kernel %.1f us -> %.2f ns per instr per SIMD\n", name, waves_per_simd,
           avg / ninst, avg / ninst / waves_per_simd, ms * 1e3, ms * 1e6 / (ninst * waves_per_simd));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("fma_f64 x8 independent", w);
        run<1>("fma_f64 dependent chain", w);
        run<2>("mul_f64 x8 independent", w);
        run<3>("add_f64 x8 independent", w);
        run<4>("add_f64 dependent chain", w);
        run<5>("mov_b32_dpp wave_shr/shl", w);
        run<6>("fma_f64 x8 indep, SGPR operand", w);
        run<7>("fma_f32 x4 chains", w);
    }
    return 0;
}

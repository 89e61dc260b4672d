// valu_bench.hip -- issue rate of the VALU instruction classes the encode kernels are made of (gfx950).
// Each kernel runs ITERS x 32 independent instructions of one class per wave; `waves per SIMD` is swept by the
// launch geometry.  Prints cycles per wave-instruction per SIMD (at the clock implied by s_memtime... here: wall time
// x nominal 2.4 GHz, and the ratio to v_fma_f32).
//   hipcc --offload-arch=gfx950 -O3 tools/bench/valu_bench.hip -o gpurun_out/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 2048

#define REP8(X) X X X X X X X X
#define BODY32(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    for (int i = 0; i < ITERS; i++) {
        if constexpr (KIND == 0) {  // v_fma_f32, 8 independent chains x 4
            asm volatile(BODY32("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 1) {  // v_mul_f32 with a 32-bit literal
            asm volatile(BODY32("v_mul_f32 %0, 0x3f7ff000, %0\n v_mul_f32 %1, 0x3f7ff000, %1\n v_mul_f32 %2, 0x3f7ff000, %2\n v_mul_f32 %3, 0x3f7ff000, %3\n"
                                "v_mul_f32 %4, 0x3f7ff000, %4\n v_mul_f32 %5, 0x3f7ff000, %5\n v_mul_f32 %6, 0x3f7ff000, %6\n v_mul_f32 %7, 0x3f7ff000, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 2) {  // v_rcp_f32
            asm volatile(BODY32("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                                "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 3) {  // v_pk_fma_f32 (4 chains of pairs, 16 pk instructions = 32 fma lanes-ops)
            asm volatile(BODY32("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        } else if constexpr (KIND == 4) {  // v_cmp + v_cndmask pairs
            asm volatile(BODY32("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
                                "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
        } else if constexpr (KIND == 5) {  // v_med3_f32 (VOP3, two literals not allowed: use regs)
            asm volatile(BODY32("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                                "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 6) {  // dependent chain: ONE accumulator
            asm volatile(BODY32("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                                "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n")
                         : "+v"(a0) : "v"(b), "v"(c));
        } else if constexpr (KIND == 7) {  // v_add_u32 / integer
            asm volatile(BODY32("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                "v_lshrrev_b32 %4, 1, %4\n v_and_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 9) {  // v_minimum3_f32 / v_maximum3_f32 (NaN-propagating, gfx950)
            asm volatile(BODY32("v_minimum3_f32 %0, %0, %8, %8\n v_maximum3_f32 %1, %1, %9, %9\n v_minimum3_f32 %2, %2, %8, %8\n v_maximum3_f32 %3, %3, %9, %9\n"
                                "v_minimum3_f32 %4, %4, %8, %8\n v_maximum3_f32 %5, %5, %9, %9\n v_minimum3_f32 %6, %6, %8, %8\n v_maximum3_f32 %7, %7, %9, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 10) {  // v_max_f32 / v_min_f32
            asm volatile(BODY32("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %9\n"
                                "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_min_f32 %6, %6, %8\n v_max_f32 %7, %7, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 11) {  // v_cmp only (to SGPR pairs other than vcc)
            asm volatile(BODY32("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n"
                                "v_cmp_lt_f32 s[28:29], %4, %8\n v_cmp_lt_f32 s[30:31], %5, %8\n v_cmp_lt_f32 s[32:33], %6, %8\n v_cmp_lt_f32 s[34:35], %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
                         : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");
        } else if constexpr (KIND == 12) {  // v_cndmask only (vcc fixed)
            asm volatile(BODY32("v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n"
                                "v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
        } else if constexpr (KIND == 13) {  // v_fmac_f32 (VOP2)
            asm volatile(BODY32("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                                "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 14) {  // v_floor / v_cvt
            asm volatile(BODY32("v_floor_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_floor_f32 %2, %2\n v_cvt_f32_u32 %3, %3\n"
                                "v_floor_f32 %4, %4\n v_cvt_u32_f32 %5, %5\n v_floor_f32 %6, %6\n v_cvt_f32_u32 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 8) {  // v_fma_f64
            double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = b, dc = c;
            asm volatile(BODY32("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));
            a0 = (float)d0; a1 = (float)d1; a2 = (float)d2; a3 = (float)d3;
        }
    }
    if constexpr (KIND == 3) { a0 = p0.x + p0.y; a1 = p1.x + p1.y; a2 = p2.x + p2.y; a3 = p3.x + p3.y; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
static void run(const char *name, int per_iter, float *d)
{
    for (int wps : {2, 8}) {
        const int blocks = 256 * wps;  // 256-thread blocks = 4 waves = one per SIMD; wps blocks per CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<KIND><<<blocks, 256>>>(d, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(d, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)ITERS * per_iter * wps;
        printf("%-28s waves/SIMD %d : %.3f ms -> %.2f cycles per wave-instruction per SIMD @2.4 GHz\n", name, wps, ms,
               ms * 1e-3 * 2.4e9 / instr_per_simd);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32 (8 chains)", 32 * 8, d);
    run<6>("v_fma_f32 (1 dependent chain)", 32 * 8, d);
    run<1>("v_mul_f32 literal", 32 * 8, d);
    run<2>("v_rcp_f32", 32 * 8, d);
    run<3>("v_pk_fma_f32", 32 * 4, d);
    run<4>("v_cmp+v_cndmask", 32 * 8, d);
    run<5>("v_med3_f32", 32 * 8, d);
    run<7>("int add/shift/and", 32 * 8, d);
    run<8>("v_fma_f64", 32 * 4, d);
    run<9>("v_minimum3/maximum3_f32", 32 * 8, d);
    run<10>("v_min/max_f32", 32 * 8, d);
    run<11>("v_cmp_lt_f32 -> sgpr", 32 * 8, d);
    run<12>("v_cndmask_b32 (vcc)", 32 * 8, d);
    run<13>("v_fmac_f32", 32 * 8, d);
    run<14>("v_floor/v_cvt", 32 * 8, d);
    return 0;
}

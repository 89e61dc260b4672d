// tools/bench/powf_variants.cpp -- VERDICT r05 item 3: can the fp64 chain of the YCbCr kernels' powf be shortened without losing
// bit-equality with glibc's powf?  Host-side, exhaustive over the argument ranges the DECODE kernel proves for its two powers
// (luma_device.hpp pq_decode_r):
//     A  Vp = val^(1/m),  val in [2^-21, 1]              (every float: 176 160 769 arguments)
//     B  t^(1/n),         t in [3e-9, 6.4]                (every float: ~260 M arguments)
// Each candidate evaluates the same real-valued function as e_powf.c with fewer or cheaper operations; what counts is whether the
// FLOAT it returns equals the exact chain's (lh::powf_regular = glibc's operations in glibc's order) for every argument.
//
//   g++ -O2 -ffp-contract=off -std=c++17 -pthread -o /tmp/powf_variants tools/bench/powf_variants.cpp && /tmp/powf_variants
//
// Candidates (fp64 instructions of the chain: 17 in the exact form):
//   fold_y     the exponent folded into the log2 polynomial: y*A0 .. y*A4 and y*(logc + k) precomputed, so ylogx comes out of the
//              last fma (16: one v_mul_f64 less).  Same real function, different roundings.
//   f32_tail   the r^4 (A0 r + A1) term -- 2^-21 of the result -- evaluated in fp32 and widened (15 fp64 + 3 fp32 + 1 cvt).
//   deg4       that term's A0 r dropped altogether (a degree-4 polynomial): how much slack the LAST term has.
//   near_rn    not a candidate: counts the arguments whose exact double result lies within 2^-41 (relative) of a float rounding
//              boundary -- the arguments a "cheap chain + fall back when close to a boundary" scheme would have to redo.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../lumahdrv_amd/csrc/pow_glibc.hpp"

using namespace lh;

static const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1,
                    A4 = 0x1.71547652ab82bp0;
static const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;

static inline float exp2_stage(double ylogx, const PowfTables &T, double *e_out = nullptr)
{
    const double SHIFT = 0x1.8p+52 / 32;
    double kd = ylogx + SHIFT;
    const uint64_t ki = pw_asuint64(kd);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    const double s = pw_asdouble(T.exp2_tab[ki % 32] + (ki << 47));
    const double zz = __builtin_fma(C0, rr, C1);
    const double rr2 = rr * rr;
    double e = __builtin_fma(C2, rr, 1.0);
    e = __builtin_fma(zz, rr2, e);
    e = e * s;
    if (e_out)
        *e_out = e;
    return (float)e;
}

struct Red {
    double r, y0;
};
static inline Red reduce(float x, const PowfTables &T)
{
    const uint32_t ix = pw_asuint(x), tmp = ix - 0x3f330000u;
    const int i = (tmp >> 19) % 16;
    const uint32_t top = tmp & 0xff800000u;
    const int k = (int32_t)top >> 23;
    const double z = (double)pw_asfloat(ix - top);
    return {__builtin_fma(z, T.log2_tab[i][0], -1.0), T.log2_tab[i][1] + (double)k};
}

static float v_exact(float x, float y, const PowfTables &T, double *e = nullptr)
{
    const Red q = reduce(x, T);
    const double r = q.r, r2 = r * r, r4 = r2 * r2;
    double yy = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3);
    double t = __builtin_fma(A4, r, q.y0);
    t = __builtin_fma(p, r2, t);
    yy = __builtin_fma(yy, r4, t);
    return exp2_stage((double)y * yy, T, e);
}
static float v_fold_y(float x, float y, const PowfTables &T)
{
    const Red q = reduce(x, T);
    const double Y = (double)y, r = q.r, r2 = r * r, r4 = r2 * r2;
    double yy = __builtin_fma(Y * A0, r, Y * A1);
    const double p = __builtin_fma(Y * A2, r, Y * A3);
    double t = __builtin_fma(Y * A4, r, Y * q.y0);   // (Y * y0: one more table column on the device)
    t = __builtin_fma(p, r2, t);
    return exp2_stage(__builtin_fma(yy, r4, t), T);
}
static float v_f32_tail(float x, float y, const PowfTables &T)
{
    const Red q = reduce(x, T);
    const double r = q.r, r2 = r * r;
    const float rf = (float)r;
    const float tail = (rf * rf) * (rf * rf) * __builtin_fmaf((float)A0, rf, (float)A1);
    const double p = __builtin_fma(A2, r, A3);
    double t = __builtin_fma(A4, r, q.y0);
    t = __builtin_fma(p, r2, t);
    return exp2_stage((double)y * (t + (double)tail), T);
}
static float v_deg4(float x, float y, const PowfTables &T)
{
    const Red q = reduce(x, T);
    const double r = q.r, r2 = r * r, r4 = r2 * r2;
    const double p = __builtin_fma(A2, r, A3);
    double t = __builtin_fma(A4, r, q.y0);
    t = __builtin_fma(p, r2, t);
    return exp2_stage((double)y * __builtin_fma(A1, r4, t), T);
}

int main()
{
    static const PowfTables T = kPowfTablesHost;
    const float m = 78.8438f, n = 0.1593f;
    volatile float one = 1.0f;
    struct Dom {
        const char *name;
        float y, lo, hi;
    } doms[2] = {{"A: val^(1/m), val in [2^-21, 1]", one / m, 0x1p-21f, 1.0f}, {"B: t^(1/n), t in [3e-9, 6.4]", one / n, 3e-9f, 6.4f}};
    const unsigned nt = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 4;
    for (const Dom &d : doms) {
        const uint32_t b0 = pw_asuint(d.lo), b1 = pw_asuint(d.hi);
        std::atomic<uint64_t> bad_fold{0}, bad_tail{0}, bad_deg4{0}, near{0}, vs_libm{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                uint64_t f = 0, a = 0, g = 0, nr = 0, lm = 0;
                for (uint64_t b = (uint64_t)b0 + t; b <= b1; b += nt) {
                    const float x = pw_asfloat((uint32_t)b);
                    double e;
                    const float ex = v_exact(x, d.y, T, &e);
                    lm += pw_asuint(ex) != pw_asuint(powf(x, d.y));
                    f += pw_asuint(v_fold_y(x, d.y, T)) != pw_asuint(ex);
                    a += pw_asuint(v_f32_tail(x, d.y, T)) != pw_asuint(ex);
                    g += pw_asuint(v_deg4(x, d.y, T)) != pw_asuint(ex);
                    // distance of e from the nearest float rounding boundary, in units of its own ulp(double): the 29 dropped bits
                    const uint64_t low = pw_asuint64(e) & ((1ull << 29) - 1);
                    const uint64_t dist = low > (1ull << 28) ? low - (1ull << 28) : (1ull << 28) - low;
                    nr += dist < (1ull << 12);   // within 2^12 * 2^-52 = 2^-40 .. 2^-41 relative
                }
                bad_fold += f;
                bad_tail += a;
                bad_deg4 += g;
                near += nr;
                vs_libm += lm;
            });
        for (auto &x : th)
            x.join();
        const double N = (double)b1 - b0 + 1;
        printf("%s: %.0f arguments; exact chain vs host libm powf: %llu differ\n", d.name, N, (unsigned long long)vs_libm.load());
        printf("   fold_y   (16 fp64)                 : %llu floats differ (%.2e of the arguments)\n", (unsigned long long)bad_fold.load(), bad_fold / N);
        printf("   f32_tail (15 fp64 + 3 fp32 + 1 cvt): %llu floats differ (%.2e)\n", (unsigned long long)bad_tail.load(), bad_tail / N);
        printf("   deg4     (15 fp64)                 : %llu floats differ (%.2e)\n", (unsigned long long)bad_deg4.load(), bad_deg4 / N);
        printf("   near_rn  (within 2^-41 of a rounding boundary): %llu arguments (%.2e) -> with %d such powf per wave and unit, %.1f %% of the units redo\n",
               (unsigned long long)near.load(), near / N, 64 * 8 * 3, 100.0 * (1.0 - pow(1.0 - near / N, 64.0 * 8 * 3)));
    }
    return 0;
}

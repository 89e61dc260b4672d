// tools/verify_powf.cpp -- exhaustive check of lumahdrv_amd/csrc/pow_glibc.hpp against the host libm.
//
//   g++ -O2 -ffp-contract=off -std=c++17 -pthread -o /tmp/verify_powf tools/verify_powf.cpp -lm && /tmp/verify_powf [stride]
//
// For each of the four exponents the reference's transformPQ uses (src/luma_quantizer.cpp:485-501:
// n = 0.1593f, m = 78.8438f, 1.0f/m, 1.0f/n) every non-negative float bit pattern 0 .. 0x7f800000 (zero,
// subnormals, normals, +inf) plus NaNs and a sweep of negatives is evaluated with the restatement and
// with libm powf; results must be bit-identical (NaN == NaN).  stride > 1 samples every stride-th pattern.
// The straight-line form the kernels use (powf_regular with its "redo with the complete function" flag) is checked
// on the same arguments with both table layouts: the 16-entry log2 table and the wide LDS table (one entry per
// binary exponent in [-64, 63] x table index), filled by the function the kernels fill it with.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../lumahdrv_amd/csrc/pow_glibc.hpp"

static inline bool same(float a, float b)
{
    if (a != a && b != b)
        return true;
    uint32_t x, y;
    memcpy(&x, &a, 4);
    memcpy(&y, &b, 4);
    return x == y;
}

int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const float m = 78.8438, n = 0.1593;
    volatile float one = 1.0f;
    const float ys[4] = {n, m, one / m, one / n};
    const unsigned nt = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 4;
    uint64_t total_bad = 0, total = 0;
    static lh::PowfTablesWide wide;
    {
        const double lt[16][2] = LH_POWF_LOG2_TAB;
        const uint64_t et[32] = LH_POWF_EXP2_TAB;
        memcpy(wide.log2_tab, lt, sizeof lt);
        memcpy(wide.exp2_tab, et, sizeof et);
        for (int e = 0; e < 2048; e++)
            lh::pw_wide_entry(e, lt, wide.wide[e][0], wide.wide[e][1]);
        for (int e = 0; e < lh::FOLD_A_LEN; e++)
            lh::pw_fold_entry<0>(e, lt, wide.foldA[e][0], wide.foldA[e][1]);
        for (int e = 0; e < 16; e++)
            lh::pw_fold_entry<1>(e, lt, wide.foldC[e][0], wide.foldC[e][1]);
    }
    std::atomic<uint64_t> fast16{0}, fastw{0};
    for (int e = 0; e < 4; e++) {
        const float y = ys[e];
        std::atomic<uint64_t> bad{0}, cnt{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                uint64_t b = 0, c = 0, f1 = 0, f2 = 0;
                for (uint64_t u = t * stride; u <= 0xffffffffull; u += nt * stride) {
                    // all non-negative patterns; of the negative / NaN half only every 4096th
                    if (u > 0x7f800000ull && (u & 0xfff) != 0 && stride == 1)
                        continue;
                    float x = lh::pw_asfloat((uint32_t)u);
                    float a = lh::powf_glibc(x, y, lh::kPowfTablesHost);
                    float r = powf(x, y);
                    c++;
                    if (!same(a, r)) {
                        if (b < 5)
                            fprintf(stderr, "MISMATCH y=%a x=%a (0x%08x): restated %a libm %a\n", y, x, (unsigned)u, a, r);
                        b++;
                    }
                    // straight-line form: wherever it does not raise `slow` its result must be libm's
                    bool s16 = false, sw = false;
                    const float f16 = lh::powf_regular<true, true, true>(x, y, lh::kPowfTablesHost, s16);
                    const float fw = lh::powf_regular<true, true, true>(x, y, wide, sw);
                    if ((!s16 && !same(f16, r)) || (!sw && !same(fw, r))) {
                        if (b < 5)
                            fprintf(stderr, "MISMATCH (straight-line) y=%a x=%a (0x%08x): %a / %a libm %a\n", y, x, (unsigned)u, f16, fw, r);
                        b++;
                    }
                    f1 += !s16;
                    f2 += !sw;
                }
                bad += b;
                cnt += c;
                fast16 += f1;
                fastw += f2;
            });
        for (auto &x : th)
            x.join();
        printf("y=%-14a (%.9g): %llu arguments, %llu mismatches\n", y, y, (unsigned long long)cnt.load(),
               (unsigned long long)bad.load());
        total_bad += bad;
        total += cnt;
    }
    // the folded form (round 6) over everything its tables reach -- the binades' split point is OFF = 0x3f330000 = 0.69921875:
    // val^(1/m) for val in [OFF 2^-21, 2 OFF) -- the kernels hand it [2^-21, 1] -- and q^m for q in [OFF, 2 OFF) -- the kernels
    // hand it [0.8359, 1.0088].  Every float unless stride > 7.
    {
        const uint64_t fs = stride > 7 ? 7 : stride;
        const struct {
            int which;
            float y, lo, hi;
        } dom[2] = {{0, ys[2], 0.69921875f * 0x1p-21f, 1.3984375f}, {1, ys[1], 0.69921875f, 1.3984375f}};
        for (const auto &d : dom) {
            const uint32_t b0 = lh::pw_asuint(d.lo), b1 = lh::pw_asuint(d.hi) - 1;
            std::atomic<uint64_t> bad{0}, cnt{0};
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&, t]() {
                    uint64_t b = 0, c = 0;
                    for (uint64_t u = (uint64_t)b0 + t * fs; u <= b1; u += nt * fs) {
                        const float x = lh::pw_asfloat((uint32_t)u);
                        const float a = d.which == 0 ? lh::powf_folded<0>(x, wide) : lh::powf_folded<1>(x, wide);
                        c++;
                        if (!same(a, powf(x, d.y))) {
                            if (b < 5)
                                fprintf(stderr, "MISMATCH (folded %d) x=%a (0x%08x): %a libm %a\n", d.which, x, (unsigned)u, a, powf(x, d.y));
                            b++;
                        }
                    }
                    bad += b;
                    cnt += c;
                });
            for (auto &x : th)
                x.join();
            printf("folded form, y=%-14a (%.9g), x in [%g, %g): %llu arguments, %llu mismatches\n", d.y, d.y, d.lo, d.hi,
                   (unsigned long long)cnt.load(), (unsigned long long)bad.load());
            total_bad += bad;
            total += cnt;
        }
    }
    // EMAX (pow_glibc.hpp): an argument of +0 must stay far below every bound on |y log2 x| (56 is the smallest in use) through
    // the wide table for y = 1/n -- the decode kernels test that bound on a running maximum that +0 arguments take part in
    for (int e = 3; e < 4; e++) {
        bool slow = false;
        uint32_t emax = 0;
        const float r0 = lh::powf_regular<true, false, 0>(0.0f, ys[e], wide, slow, &emax);
        if (r0 != 0.0f || slow || emax >= lh::pw_emax_limit(32)) {
            fprintf(stderr, "MISMATCH (EMAX of +0) y=%a: result %a, high word of |ylogx| 0x%08x\n", ys[e], r0, emax);
            total_bad++;
        }
    }
    printf("straight-line form applied to %llu (16-entry table) / %llu (wide table) of them\n",
           (unsigned long long)fast16.load(), (unsigned long long)fastw.load());
    printf("TOTAL %llu arguments, %llu mismatches\n", (unsigned long long)total, (unsigned long long)total_bad);
    return total_bad ? 1 : 0;
}

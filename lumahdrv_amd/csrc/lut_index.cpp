// lut_index.cpp -- see lut_index.hpp
#include "lut_index.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>

namespace lh {

static inline int32_t fbits(float f)
{
    int32_t b;
    std::memcpy(&b, &f, 4);
    return b;
}

static inline float from_bits(int64_t b)
{
    int32_t x = (int32_t)b;
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

int quantize_literal_host(float v, const float *lut, int maxVal)
{
    int l = 0, r = maxVal;
    while (l + 1 < r) {
        const int m = (l + r) / 2;
        if (v < lut[m])
            r = m;
        else
            l = m;
    }
    return ((v - lut[l]) < (lut[r] - v)) ? l : r;
}

int thresh_lookup_host(const ThreshIndex &ix, float v)
{
    const int32_t b = fbits(v);
    const int khi = ix.kmin + ix.nbuckets - 1;
    const int k = std::min(std::max(b >> ix.shift, ix.kmin), khi);
    const uint32_t mask = ix.shift >= 32 ? 0xffffffffu : ((1u << ix.shift) - 1u);
    return (int)((ix.rec[(size_t)(k - ix.kmin)] + ((uint32_t)b & mask)) >> ix.shift);
}

ThreshIndex build_thresh_index(const float *lut, int n, int max_buckets)
{
    const int maxVal = n - 1;
    if (n < 2 || n > 65536)
        return ThreshIndex();
    for (int i = 0; i < n; i++) {
        if (lut[i] != lut[i] || std::isinf(lut[i]))
            return ThreshIndex();
        if (i && lut[i] < lut[i - 1])
            return ThreshIndex();
    }
    return build_thresh_index_fn([&](float v) { return quantize_literal_host(v, lut, maxVal); }, maxVal, max_buckets, false);
}

// T[j] = the smallest bit pattern whose code is >= c0 + 1 + j, c0 = code(+0): false when the step function does not qualify
// (everything at or below +0 must share one code -- then every threshold is a positive float and bit patterns of the
// non-negative half sort like the values; a caller that never presents negative values says so --, code(+inf) must be maxVal,
// and the code must not jump by two at one float: duplicate table entries)
static bool find_thresholds(const std::function<int(float)> &code, int maxVal, bool nonneg_only, int &c0, std::vector<uint32_t> &T)
{
    auto f = [&](uint32_t bits) { return code(from_bits((int64_t)bits)); };
    const uint32_t PINF = 0x7f800000u;
    c0 = f(0u);
    if ((!nonneg_only && code(-__builtin_inff()) != c0) || f(PINF) != maxVal)
        return false;
    T.clear();
    T.reserve((size_t)(maxVal - c0));
    uint32_t prev = 1u;
    for (int c = c0 + 1; c <= maxVal; c++) {
        uint32_t lo = prev, hi = PINF;  // f(hi) = maxVal >= c; answer in [lo, hi]
        if (f(lo) >= c) {
            hi = lo;
        } else {
            while (hi - lo > 1) {  // invariant f(lo) < c <= f(hi)
                const uint32_t mid = lo + (hi - lo) / 2;
                if (f(mid) >= c)
                    hi = mid;
                else
                    lo = mid;
            }
        }
        if (!T.empty() && hi == T.back())
            return false;
        T.push_back(hi);
        prev = hi;
    }
    return !T.empty();
}

static inline uint32_t lin_key(float v, float kscale, int nbuckets)
{
    float p = v * kscale;                       // (-ffp-contract=off: one rounded fp32 product, as v_mul_f32)
    p = std::fmin(p, (float)(nbuckets - 1));    // minNum: a NaN product becomes the top bucket (v_min_f32)
    return p > 0.0f ? (uint32_t)p : 0u;         // v_cvt_u32_f32: truncation, negatives (and -0) to 0
}

int lin_lookup_host(const LinIndex &ix, float v)
{
    const uint32_t k = lin_key(v, ix.kscale, ix.nbuckets);
    return (int)ix.rec[2 * (size_t)k + 1] + (fbits(v) > (int32_t)ix.rec[2 * (size_t)k] ? 1 : 0);
}

LinIndex build_lin_index(const float *lut, int n, int max_buckets)
{
    LinIndex ix;
    const int maxVal = n - 1;
    if (n < 2 || n > 65536)
        return ix;
    for (int i = 0; i < n; i++)
        if (lut[i] != lut[i] || std::isinf(lut[i]) || (i && lut[i] < lut[i - 1]))
            return ix;
    const std::function<int(float)> code = [&](float v) { return quantize_literal_host(v, lut, maxVal); };
    int c0 = 0;
    std::vector<uint32_t> T;
    if (!find_thresholds(code, maxVal, false, c0, T))
        return ix;
    if (T.size() < 2)
        return ix;
    double gap = __builtin_inf();     // smallest distance between two thresholds (bucket 0 may hold T[0] whatever its value)
    for (size_t j = 1; j < T.size(); j++)
        gap = std::min(gap, (double)from_bits((int64_t)T[j]) - (double)from_bits((int64_t)T[j - 1]));
    const double top = (double)from_bits((int64_t)T.back());
    if (!(gap > 0.0) || top / gap + 3.0 > (double)max_buckets)
        return ix;
    const uint32_t PINF = 0x7f800000u;
    // bucket width just under the smallest gap; rounding of the product can still put two thresholds that are exactly one
    // gap apart into one bucket, hence the retries with a slightly larger scale
    for (int attempt = 0; attempt < 24 && !ix.ok; attempt++) {
        const float kscale = (float)((1.0 + 0.002 * (attempt + 1)) / gap);
        if (!(kscale > 0.0f) || std::isinf(kscale))
            return ix;
        const int64_t nb = (int64_t)lin_key(from_bits((int64_t)T.back()), kscale, 0x7fffff00) + 2;
        if (nb > max_buckets || nb < 2)
            return ix;
        bool unique = true;
        for (size_t j = 1; j < T.size() && unique; j++)
            unique = lin_key(from_bits((int64_t)T[j]), kscale, (int)nb) != lin_key(from_bits((int64_t)T[j - 1]), kscale, (int)nb);
        if (!unique)
            continue;
        ix.kscale = kscale;
        ix.nbuckets = (int)nb;
        ix.rec.assign(2 * (size_t)nb, 0u);
        size_t j = 0;
        for (int64_t k = 0; k < nb; k++) {
            ix.rec[2 * k + 1] = (uint32_t)(c0 + (int)j);     // thresholds in earlier buckets
            ix.rec[2 * k] = 0x7fffffffu;
            if (j < T.size() && (int64_t)lin_key(from_bits((int64_t)T[j]), kscale, (int)nb) == k)
                ix.rec[2 * k] = T[j++] - 1u;   // (T >= 1: code(+0) is c0, so no threshold sits at +0)
        }
        if (j != T.size())
            continue;   // (a threshold beyond the top bucket: cannot happen, nb was sized from the last one)
        // self-check against the literal loop: every threshold and its neighbours, both ends of every bucket (found by
        // bisection over bit patterns on key()), the special values
        bool ok = true;
        auto same = [&](uint32_t bits) { return lin_lookup_host(ix, from_bits((int64_t)bits)) == code(from_bits((int64_t)bits)); };
        for (size_t t = 0; t < T.size() && ok; t++)
            ok = same(T[t]) && same(T[t] - 1u) && (T[t] >= PINF || same(T[t] + 1u));
        uint32_t lo = 0u;
        for (int64_t k = 1; k < nb && ok; k++) {   // first float of bucket k: smallest pattern with key >= k
            uint32_t a = lo, b = PINF;
            if (lin_key(from_bits((int64_t)b), kscale, (int)nb) < (uint32_t)k)
                break;
            while (b - a > 1) {
                const uint32_t mid = a + (b - a) / 2;
                if (lin_key(from_bits((int64_t)mid), kscale, (int)nb) >= (uint32_t)k)
                    b = mid;
                else
                    a = mid;
            }
            ok = same(b) && same(b - 1u);
            lo = b;
        }
        ok = ok && same(0u) && same(PINF) && same(0x7fc00000u) && same(0xffc00000u) && same(0x7f800001u) && same(0xff800001u) && same(0x7fffffffu) && same(0xffffffffu) &&
             same(0x80000000u) && same(0xbf800000u) && same(0xff800000u) && same(0x00000001u) && same(0x80000001u);
        if (ok)
            ix.ok = true;
        else
            ix.rec.clear();
    }
    return ix;
}

ThreshIndex build_thresh_index_fn(const std::function<int(float)> &code, int maxVal, int max_buckets, bool nonneg_only)
{
    ThreshIndex ix;
    if (maxVal < 1 || maxVal > 65535)
        return ix;
    auto f = [&](uint32_t bits) { return code(from_bits((int64_t)bits)); };
    const uint32_t PINF = 0x7f800000u;
    int c0 = 0;
    std::vector<uint32_t> T;
    if (!find_thresholds(code, maxVal, nonneg_only, c0, T))
        return ix;
    int codebits = 1;
    while ((1 << codebits) <= maxVal)
        codebits++;
    for (int B = 0; B <= 23; B++) {
        const int shift = 23 - B;
        if (codebits + shift > 32)
            continue;
        const int64_t kfirst = (int64_t)(T.front() >> shift), klast = (int64_t)(T.back() >> shift);
        const int64_t nb = klast - kfirst + 3;  // one bucket below the first threshold's, one above the last's
        if (nb > max_buckets)
            break;
        if (kfirst < 1)
            continue;  // kmin must be >= 0 (the kernels also clamp the key as an unsigned number)
        bool unique = true;
        for (size_t j = 1; j < T.size() && unique; j++)
            unique = (T[j] >> shift) != (T[j - 1] >> shift);
        if (!unique)
            continue;
        ix.mant_bits = B;
        ix.shift = shift;
        ix.kmin = (int)kfirst - 1;
        ix.nbuckets = (int)nb;
        ix.rec.assign((size_t)nb, 0u);
        const uint32_t lowmask = (1u << shift) - 1u;  // shift <= 23 here
        size_t j = 0;
        for (int64_t k = kfirst - 1; k <= klast + 1; k++) {
            // thresholds at or below the bucket's first float are already counted in its start code
            const int64_t first = k << shift;
            while (j < T.size() && (int64_t)T[j] <= first)
                j++;
            uint32_t start = (uint32_t)(c0 + (int)j), u = 0;
            if (j < T.size() && (int64_t)(T[j] >> shift) == k)
                u = (1u << shift) - (T[j] & lowmask);  // low(T) != 0 here (T > first)
            ix.rec[(size_t)(k - (kfirst - 1))] = (start << shift) | u;
        }
        // self-check against the literal loop at every threshold, its predecessor and both ends of every bucket
        ix.ok = true;
        auto same = [&](uint32_t bits) { return thresh_lookup_host(ix, from_bits((int64_t)bits)) == f(bits); };
        for (size_t t = 0; t < T.size() && ix.ok; t++)
            ix.ok = same(T[t]) && same(T[t] - 1u) && (T[t] >= PINF || same(T[t] + 1u));
        for (int64_t k = std::max<int64_t>(kfirst - 1, 0); k <= klast + 1 && ix.ok; k++) {
            const int64_t a = k << shift, b = ((k + 1) << shift) - 1;
            if (a <= (int64_t)PINF)
                ix.ok = same((uint32_t)a);
            if (ix.ok && b <= (int64_t)PINF)
                ix.ok = same((uint32_t)b);
        }
        ix.ok = ix.ok && same(0u) && same(PINF) && same(0x7fc00000u) &&
                (nonneg_only || (thresh_lookup_host(ix, -0.0f) == c0 && thresh_lookup_host(ix, -1.0f) == c0 &&
                                 thresh_lookup_host(ix, -__builtin_inff()) == c0));
        // a step function that is not monotone between its thresholds cannot be told from one that is by the checks above.
        // The table search is monotone by proof (lut_index.hpp); a caller-supplied function gets a strided sweep of every
        // bucket (32 floats per bucket) against the literal function
        for (int64_t k = std::max<int64_t>(kfirst - 1, 0); nonneg_only && k <= klast + 1 && ix.ok; k++) {
            const int64_t a = k << shift, step = std::max<int64_t>(1, ((int64_t)1 << shift) / 32);
            for (int64_t b2 = a; b2 < a + ((int64_t)1 << shift) && ix.ok; b2 += step)
                if (b2 <= (int64_t)PINF)
                    ix.ok = same((uint32_t)b2);
        }
        if (!ix.ok)
            ix.rec.clear();
        return ix;
    }
    return ix;
}

}  // namespace lh

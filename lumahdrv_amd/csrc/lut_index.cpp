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

ThreshIndex build_thresh_index_fn(const std::function<int(float)> &code, int maxVal, int max_buckets, bool nonneg_only)
{
    ThreshIndex ix;
    if (maxVal < 1 || maxVal > 65535)
        return ix;
    auto f = [&](uint32_t bits) { return code(from_bits((int64_t)bits)); };
    const uint32_t PINF = 0x7f800000u;
    const int c0 = f(0u);
    // everything at or below +0 must share one code (then every threshold is a positive float and bit patterns
    // of the non-negative half sort like the values); a caller that never presents negative values says so
    if ((!nonneg_only && code(-__builtin_inff()) != c0) || f(PINF) != maxVal)
        return ix;
    std::vector<uint32_t> T;  // T[j] = smallest bit pattern whose code is >= c0 + 1 + j
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
            return ix;  // the code jumps by two at one float (duplicate table entries): not representable
        T.push_back(hi);
        prev = hi;
    }
    if (T.empty())
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

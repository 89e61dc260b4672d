// lut_index.cpp -- see lut_index.hpp
#include "lut_index.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

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

LutIndex build_lut_index(const float *lut, int n, int max_lds_bitdepth)
{
    LutIndex idx;
    const int maxVal = n - 1;
    if (n > (1 << max_lds_bitdepth)) {
        idx.mode = LUT_LITERAL_GLOBAL;
        return idx;
    }
    idx.mode = LUT_LITERAL_LDS;
    if (n < 3)
        return idx;
    // the closed form of the reference's bisection needs a non-decreasing, NaN-free table whose last
    // entry is a positive normal finite float (so that fp32 bit patterns of the positive part sort)
    for (int i = 0; i < n; i++) {
        if (lut[i] != lut[i])
            return idx;
        if (i && lut[i] < lut[i - 1])
            return idx;
    }
    if (!(lut[maxVal] >= FLT_MIN) || std::isinf(lut[maxVal]))
        return idx;
    int lo = 0;
    while (lo < n && !(lut[lo] >= FLT_MIN))
        lo++;

    auto p = [&](float v) {  // clamp((#entries <= v) - 1, 0, maxVal-1)
        int c = (int)(std::upper_bound(lut, lut + n, v) - lut) - 1;
        return std::min(std::max(c, 0), maxVal - 1);
    };

    bool have = false;
    LutIndex best;
    for (int B = 0; B <= 10; B++) {
        const int shift = 23 - B;
        const int kmin = fbits(lut[lo]) >> shift, kmax = fbits(lut[maxVal]) >> shift;
        const int K = kmax - kmin + 1;
        if (K + 1 > 8192)
            break;
        std::vector<uint16_t> start(K + 1);
        // one bucket past the table's last key: every value there is > map[maxVal] (and +inf / sign-clear NaNs land
        // there too); starting the search AT maxVal gives the reference's answer maxVal directly
        start[K] = (uint16_t)(4 * maxVal);
        int maxspan = 0;
        for (int k = 0; k < K; k++) {
            // bucket 0 also receives every value below it (key clamp), the last bucket every value above
            int s = (k == 0) ? 0 : p(from_bits((int64_t)(kmin + k) << shift));
            int e = (k == K - 1) ? maxVal - 1 : p(from_bits(((int64_t)(kmin + k + 1) << shift) - 1));
            start[k] = (uint16_t)(4 * s);  // byte offset into the table
            maxspan = std::max(maxspan, e - s);
        }
        int S = 0;
        while ((1 << S) - 1 < maxspan)
            S++;
        if (!have || S < best.steps) {
            best.mode = LUT_BUCKET_LDS;
            best.mant_bits = B;
            best.shift = shift;
            best.kmin = kmin;
            best.nbuckets = K + 1;
            best.steps = S;
            best.pad = (1 << S) + 1;
            best.start = std::move(start);
            have = true;
        }
        if (S <= 1)
            break;
    }
    if (have)
        return best;
    return idx;
}

}  // namespace lh

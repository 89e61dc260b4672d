// lut_index.hpp -- host-side construction of the luminance search index ("threshold records").
//
// The reference quantizes a luminance by bisection over the whole transfer-function table
// (LumaQuantizer::quantize, src/luma_quantizer.cpp:222-235): l=0, r=maxVal, halve until r==l+1, then
// pick the nearer of map[l], map[r] by two rounded fp32 subtractions.  The kernels replace that by ONE
// 4-byte gather per value (below).  Tables that do not qualify (NaNs, decreasing entries -- possible when a
// decoder is handed an arbitrary attachment-434 table) get mode LITERAL and the kernels run the reference's
// bisection as is.
#pragma once

#include <cstdint>
#include <functional>
#include <vector>

namespace lh {

enum LutMode : int {
    LUT_LITERAL_LDS = 0,    // reference bisection, table staged in LDS
    LUT_LITERAL_GLOBAL = 2, // reference bisection on the table in global memory (bitdepth > 12)
    LUT_THRESH_LDS = 3,     // threshold records staged in LDS: ONE 4-byte LDS read per value
    LUT_THRESH_GLOBAL = 4,  // threshold records in global memory / L2 (record table too large for LDS)
    // (5, 6: the YCbCr composite records / + the half-input table; kernel-side numbers only, luma_kernels.hpp)
    LUT_LINKEY_LDS = 7      // value-keyed records staged in LDS (evenly spaced tables: PTF_LINEAR), see LinIndex below
};

// ---------------------------------------------------------------------------------------------------
// Threshold records (the search the shipped kernels run for every monotone table).
//
// For a non-decreasing, NaN-free, finite table the reference's quantize(v) -- bisection, then "nearest of two" by
// the two rounded subtractions (v - map[l]) < (map[r] - v) -- is a NON-DECREASING step function of v:
//   * l(v) = clamp(#{entries <= v} - 1, 0, maxVal-1) is non-decreasing;
//   * inside one interval, fl(v - map[l]) is non-decreasing and fl(map[r] - v) non-increasing in v (rounding is
//     monotone), so once the comparison is false it stays false: the code switches from l to l+1 exactly once.
// Hence quantize(v) = c0 + #{c : T[c] <= v} for thresholds T[c0+1] <= ... <= T[maxVal], T[c] = the smallest float
// whose code is >= c.  The T[c] are found on the host by bisection over fp32 bit patterns with the reference's
// loop evaluated LITERALLY (quantize_literal_host), so rounding ties and the table's irregularities are baked in.
//
// Record table: key = bits(v) >> shift (exponent + B mantissa bits), clamped to [kmin, kmin+nbuckets-1]; B is the
// smallest value for which no key holds two thresholds.  rec[key-kmin] = (start << shift) | u with
//   start = code of the first float of the bucket,  u = 0 (no threshold inside) or 2^shift - low(T) (one inside),
// so that          code(v) = (rec[key] + (bits(v) & (2^shift - 1))) >> shift
// -- the carry out of the low field is the "v >= T" test.  Bucket kmin (start c0, u 0) also receives everything
// below it through the key clamp (negatives, -0, -inf: all code c0); the last bucket (start maxVal, u 0) receives
// everything above the last threshold, +inf and sign-clear NaNs (the reference returns maxVal for any NaN; a
// caller that may see negative values tests for NaN explicitly, one that cannot clamps the key as an unsigned number,
// which sends sign-set NaNs to the top bucket as well; kmin >= 0 always).  tests/test_gpu_exhaustive.py compares this with
// the literal bisection for all 2^32 bit patterns through the encode kernels themselves.
struct ThreshIndex {
    bool ok = false;
    int mant_bits = 0;  // B
    int shift = 0;      // 23 - B
    int kmin = 0;
    int nbuckets = 0;
    std::vector<uint32_t> rec;
};

// ---------------------------------------------------------------------------------------------------
// Value-keyed records, for tables whose entries are (about) evenly spaced in VALUE -- PTF_LINEAR,
// src/luma_quantizer.cpp:200-203: map[i] = Lmax * i / maxVal.  Their thresholds are evenly spaced too, so a key made of
// float bits needs as many mantissa bits at EVERY exponent as the top exponent does (LINEAR-12: 12 bits x 14 exponents =
// 57 000 buckets, 229 KiB: not in LDS).  Keyed by value there are about maxVal + 2 buckets:
//     key(v) = cvt_u32(min(v * kscale, nbuckets - 1))        (one fp32 product, rounded; NaN -> top bucket through the min;
//                                                             the conversion truncates and sends negatives to 0)
//     rec[key] = {P, start}:  code(v) = start + ((int)bits(v) > (int)P),   P = bits(T) - 1
// key() is non-decreasing in v (rounding is monotone), so a bucket is a contiguous range of floats, every threshold with
// a smaller key than v's is <= v and every one with a larger key is > v: `start` = c0 + the thresholds in earlier buckets,
// P = the bit pattern just below the one threshold T inside the bucket (0x7fffffff: none; the SIGNED comparison is then
// false for every float -- the all-ones NaN included, which is why the record holds T - 1 and the test is strict -- and it is
// false for every negative v in bucket 0, whose patterns are negative integers).  kscale is chosen from the
// smallest gap between two thresholds so that no bucket holds two; the top bucket (everything beyond the last threshold,
// +inf, every NaN) has start maxVal.  8 bytes per bucket: LINEAR-12 32 KiB, LINEAR-14 128 KiB.
struct LinIndex {
    bool ok = false;
    float kscale = 0.0f;
    int nbuckets = 0;
    std::vector<uint32_t> rec;   // 2 words per bucket: P = bits(T) - 1 (0x7fffffff: no threshold), start
};
int lin_lookup_host(const LinIndex &ix, float v);   // device-equivalent evaluation
LinIndex build_lin_index(const float *lut, int n, int max_buckets);

// the reference's loop, literally (src/luma_quantizer.cpp:222-235)
int quantize_literal_host(float v, const float *lut, int maxVal);
// device-equivalent evaluation of a record table (v must not be a sign-set NaN)
int thresh_lookup_host(const ThreshIndex &ix, float v);
ThreshIndex build_thresh_index(const float *lut, int n, int max_buckets);
// The same records for ANY non-decreasing step function code(v) on the non-negative floats with code(+inf) == maxVal
// (nonneg_only: the kernels never present a negative value, so code() need not be constant below +0).  Used for the
// composite "Y' -> luminance code" function of the YCbCr encode kernels (luma_device.hpp, ycbcr_luma_code).
ThreshIndex build_thresh_index_fn(const std::function<int(float)> &code, int maxVal, int max_buckets, bool nonneg_only);

}  // namespace lh

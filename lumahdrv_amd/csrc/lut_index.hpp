// lut_index.hpp -- host-side construction of the bucketed LUT search index.
//
// The reference quantizes a luminance by bisection over the whole transfer-function table
// (LumaQuantizer::quantize, src/luma_quantizer.cpp:222-235): l=0, r=maxVal, halve until r==l+1, then
// pick the nearer of map[l], map[r] by two rounded fp32 subtractions.  For a table that is
// non-decreasing and NaN-free the loop's result is a pure function of v:
//
//      l = clamp( (number of entries <= v) - 1, 0, maxVal-1 ),   r = l + 1
//
// (invariant map[l] <= v < map[r] with the two ends never tested).  The kernels exploit that: the top
// bits of v's IEEE-754 encoding (exponent + B mantissa bits) select a bucket whose first entry index
// start[k] was precomputed here; the bucket spans at most 2^S-1 further entries, so S compare-and-step
// probes (instead of log2(2^bits) = 10..12) reach the same l, and the final nearest-of-two decision is
// then evaluated literally.  Tables that are not monotone (possible when a decoder is handed an
// arbitrary attachment-434 table) get mode LITERAL and the kernels run the reference's bisection as is.
#pragma once

#include <cstdint>
#include <vector>

namespace lh {

enum LutMode : int {
    LUT_LITERAL_LDS = 0,    // reference bisection, table staged in LDS
    LUT_BUCKET_LDS = 1,     // bucketed search, table + bucket starts staged in LDS
    LUT_LITERAL_GLOBAL = 2  // reference bisection on the table in global memory (bitdepth > 12)
};

struct LutIndex {
    int mode = LUT_LITERAL_LDS;
    int mant_bits = 0;  // B
    int shift = 0;      // 23 - B
    int kmin = 0;       // key of bucket 0
    int nbuckets = 0;   // K
    int steps = 0;      // S
    int pad = 1;        // NaN floats appended after the table: probes reach index maxVal + 2^S
    std::vector<uint16_t> start;  // K entries: BYTE offset (4 * first candidate index) per bucket
};

// lut has n = maxVal+1 entries
LutIndex build_lut_index(const float *lut, int n, int max_lds_bitdepth = 12);

}  // namespace lh

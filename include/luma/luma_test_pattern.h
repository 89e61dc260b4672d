// luma_test_pattern.h -- the synthetic test frame of the reference's ExrInterface::testFrame
// (src/exr_interface.cpp:50-70 there; `lumaenc -i __test__` and test_simple_enc without arguments use it):
// top 10 % of the rows a quadratic ramp, next 10 % a 20-step staircase, the rest a 20x30 checkerboard in R
// and row/column-modulated quadratic ramps in G and B, all within [0, 10000] cd/m2.
#ifndef LUMA_HIP_TEST_PATTERN_H
#define LUMA_HIP_TEST_PATTERN_H

#include "luma_exception.h"
#include "luma_frame.h"

inline bool lumaTestFrame(LumaFrame &frame, unsigned int w = 1280, unsigned int h = 720)
{
    frame.width = w;
    frame.height = h;
    frame.channels = 3;
    if (!frame.init())
        throw LumaException("Cannot allocate memory for input frame");
    const size_t W = w, H = h;
    float *R = frame.getChannel(0), *G = frame.getChannel(1), *B = frame.getChannel(2);
    for (size_t y = 0; y < H; y++) {
        const bool ramp = y < H / 10, stairs = y < H / 5;
        const size_t band = (20 * y / H) % 2;
        for (size_t x = 0; x < W; x++) {
            const size_t i = x + y * W;
            if (stairs) {
                const float v = ramp ? 10000.0f * ((float)(x * x)) / (W * W) : 10000.0f * ((20 * x) / W) / 20.0f;
                R[i] = G[i] = B[i] = v;
            } else {
                R[i] = 10000.0f * (band ^ ((30 * x / W) % 2));
                G[i] = 10000.0f * band * ((float)(y * y)) / (H * H);
                B[i] = 10000.0f * band * ((float)(x * x)) / (W * W);
            }
        }
    }
    return true;
}

#endif
